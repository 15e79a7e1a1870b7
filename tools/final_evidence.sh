#!/bin/bash
# Collects the round's evidence on the GPU box: smoke, rocprofv3 kernel stats of the bench, PMC passes, the
# default bench line (with the CPU baseline), C3/C5 lines.   tools/final_evidence.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-final}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" && timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 500 --warmup 50 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err" || exit 1
cd "$R" && tools/pmc_run.sh "$TAG/pmc" > "$O/pmc_stdout.log" 2>&1 || exit 1
cp "$O/pmc/traffic.json" profiles/traffic_latest.json
timeout -k 10 400 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
for w in C3 C5; do timeout -k 10 300 python3 bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline > "$O/bench_$w.json" 2> "$O/bench_$w.err" || exit 1; done
tail -1 "$O/smoke.log"; tail -c 1400 "$O/bench_default.json"; head -3 "$O"/stats/*/*kernel_stats.csv | cut -c1-150
