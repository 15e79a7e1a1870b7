#!/bin/bash
# Copies the judged summaries of a tools/final_evidence.sh run from gpurun_out/<tag>/ into profiles/<round>/ with a prefix.
#   tools/collect_profiles.sh <tag> <round dir> <prefix>
R=$(cd "$(dirname "$0")/.." && pwd); T=$R/gpurun_out/$1; D=$R/profiles/$2; P=$3
mkdir -p "$D"
for w in C2 C3 C5 R; do
  [ -f "$T/${w}_kernel_stats.csv" ] && cp "$T/${w}_kernel_stats.csv" "$D/${P}_${w}_kernel_stats.csv"
  [ -f "$T/bench_${w}_under_rocprof.json" ] && cp "$T/bench_${w}_under_rocprof.json" "$D/${P}_${w}_bench_under_rocprof.json"
  [ -f "$T/pmc_$w/pmc_summary.json" ] && cp "$T/pmc_$w/pmc_summary.json" "$D/${P}_${w}_pmc_summary.json"
  [ -f "$T/pmc_$w/traffic.json" ] && cp "$T/pmc_$w/traffic.json" "$D/${P}_${w}_traffic.json"
done
for f in bench_C2_default bench_C2_steps20_warmup5 bench_C3 bench_C5 bench_R bench_C4 bench_C4_strong_n1 bench_C2_gpus2_one_device_gloo; do
  [ -f "$T/$f.json" ] && cp "$T/$f.json" "$D/${P}_$f.json"
done
for f in stream_bench afet_bench host_batch_bench melcep_sweep stamps2048; do [ -f "$T/$f.txt" ] && cp "$T/$f.txt" "$D/${P}_$f.txt"; done
# HBM traffic per workload for bench.py's roofline.traffic (labelled there as read from this file)
python3 - "$R/profiles/traffic_latest.json" "$T" <<'PY'
import json, os, sys
path, t = sys.argv[1:3]
cur = {}
for wl in ("C2", "C3", "C5", "R"):
    f = os.path.join(t, "pmc_" + wl, "traffic.json")
    if os.path.exists(f):
        e = json.load(open(f)); e["workload"] = wl
        e["source"] = "rocprofv3 --pmc passes (tools/pmc_run.sh): FETCH_SIZE x2 + WRITE_SIZE, KiB, per launch of the front-end kernel"
        cur[wl] = e
if cur: json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
PY
ls "$D" | grep "^$P" | wc -l
