#!/bin/bash
# dev aid: rocprofv3 kernel stats of an arbitrary python script (run on the GPU box): tools/kstats_cmd.sh <tag> <script.py>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$TAG" -- python3 "$R/$1" > "$R/gpurun_out/$TAG.log" 2>&1
grep "mfx::" "$R"/gpurun_out/$TAG/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
