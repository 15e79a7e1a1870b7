#!/bin/bash
# dev aid (GPU box): kernel time of several variant libraries in ONE call, interleaved twice (same box, same thermal state)
#   tools/abx.sh name1 name2 ...     (names under build/var/lib_<name>.so; "shipped" = the in-tree library)
mkdir -p gpurun_out/abx
for rep in 1 2; do
for n in "$@"; do
    lib=build/var/lib_$n.so; [ "$n" = shipped ] && lib=""
    MFX_LIB=$lib timeout -k 10 120 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline ${ABX_ARGS:-} > gpurun_out/abx/$n.$rep.json 2> gpurun_out/abx/$n.$rep.err || { echo "$n FAILED"; tail -3 gpurun_out/abx/$n.$rep.err; continue; }
    python3 - $n $rep gpurun_out/abx/$n.$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("%-14s rep %s  step %.4f ms  kernel %.4f ms  value %.4g" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["value"]))
PY
done
done
