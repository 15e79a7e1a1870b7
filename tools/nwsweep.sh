#!/bin/bash
# dev aid (GPU box): C5 kernel time of variant libraries x waves per block:  tools/nwsweep.sh "lib:nw lib:nw ..."
mkdir -p gpurun_out/nw
for spec in $1; do
  n=${spec%%:*}; nw=${spec##*:}
  lib=build/var/lib_$n.so; [ "$n" = shipped ] && lib=""
  MFX_REG_NW=$nw MFX_LIB=$lib timeout -k 10 120 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload ${WL:-C5} > gpurun_out/nw/$n.$nw.json 2> gpurun_out/nw/$n.$nw.err || { echo "$spec FAILED"; tail -3 gpurun_out/nw/$n.$nw.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/nw/$n.$nw.json').read().strip().splitlines()[-1])
print('%-10s nw %-3s kernel %.4f ms value %.4g'%('$n','$nw',d['roofline']['kernel_avg_ms'],d['value']))"
done
