#!/bin/bash
# dev aid: rocprofv3 kernel stats of a bench run, first lines (run on the GPU box)
#   tools/kstats.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-kstats}; shift || true
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$TAG" -- python3 "$R/bench.py" --steps 500 --warmup 50 --no-cpu-baseline "$@" > "$R/gpurun_out/$TAG.json" 2> "$R/gpurun_out/$TAG.err"
head -6 "$R"/gpurun_out/$TAG/*/*kernel_stats.csv | cut -c1-170
