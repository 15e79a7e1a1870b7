#!/bin/bash
# dev aid: A/B the shipped libmfcchip.so against experimental builds of the same ABI (MFX_LIB override)
#   tools/ab.sh [lib.so ...]      -> ms_per_step + dominant kernel ms per variant
mkdir -p gpurun_out/exp
for v in "" "$@"; do
    name=$(basename "${v:-shipped}" .so)
    MFX_LIB=$v timeout -k 10 200 python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline \
        > gpurun_out/exp/ab_$name.json 2> gpurun_out/exp/ab_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/exp/ab_$name.err; exit 1; }
    python3 - "$name" gpurun_out/exp/ab_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-16s step %.4f ms  value %.4g  kernel %s" % (sys.argv[1], d["ms_per_step"], d["value"], {k: r[k] for k in r if "ms" in k or k in ("achieved", "frac")}))
PY
done
