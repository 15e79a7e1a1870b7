#!/bin/bash
# dev aid: one PMC group over the bench for a given library (run on the GPU box): tools/pmc_one.sh <tag> <lib|""> "<counters>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LIB=$2; CNT=$3
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/$TAG"
MFX_LIB=$LIB timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$R/gpurun_out/$TAG" -- python3 "$R/bench.py" --steps 5 --warmup 1 --settle-ms 0 --no-cpu-baseline > "$R/gpurun_out/$TAG.log" 2>&1
python3 - "$R/gpurun_out/$TAG" "$TAG" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_front512" in row.get("Kernel_Name", ""): agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
