#!/bin/bash
# Collects rocprofv3 PMC passes for the bench (one counter group per run; --pmc is never combined
# with trace domains other than the kernel trace).  Usage on the GPU box:  tools/pmc_run.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}; shift || true
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --steps 5 --warmup 1 --settle-ms 0 --no-cpu-baseline "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "mfx::" not in k: continue
        short = next((n for n in ("k_front2048", "k_front512", "k_front1024", "k_front_reg", "k_front_wave", "k_delta16", "k_delta4", "k_delta", "k_melcep", "k_norm_seg", "k_norm_stats", "k_norm_finalize", "k_norm_apply") if n in k), k[:40])
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
# HBM traffic of the dominant kernel per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE reports half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM;
# confirmed for this kernel's 4-byte-per-lane loads by tools/ubench/fetch_calib.hip: 1 GiB read ->
# 524 299 KiB), WRITE_SIZE is exact at 32-byte sector granularity (same calibration).
wl = "C2"
for i, a in enumerate(sys.argv):
    if a == "--workload" and i + 1 < len(sys.argv): wl = sys.argv[i + 1]
kn = next((n for n in ("k_front2048", "k_front512", "k_front1024", "k_front_reg", "k_front_wave") if n in res), "k_front512")
k = res.get(kn) or {}
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    traffic = {"kernel": kn, "workload": wl,
               "fetch_size_kib_raw": k["FETCH_SIZE"], "write_size_kib": k["WRITE_SIZE"],
               "hbm_read_bytes_per_launch": 2 * 1024 * k["FETCH_SIZE"], "hbm_write_bytes_per_launch": 1024 * k["WRITE_SIZE"],
               "hbm_bytes_per_launch": 2 * 1024 * k["FETCH_SIZE"] + 1024 * k["WRITE_SIZE"],
               "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1"}
    # every kernel of the step (front end + delta (+ normaliser)): bench.py's roofline.whole_path.traffic
    nfront = max(len(agg[kn]["FETCH_SIZE"]), 1)
    traffic["step_kernels"] = {n: {"hbm_bytes_per_launch": 2 * 1024 * d["FETCH_SIZE"] + 1024 * d["WRITE_SIZE"],
                                   "hbm_read_bytes_per_launch": 2 * 1024 * d["FETCH_SIZE"],
                                   "hbm_write_bytes_per_launch": 1024 * d["WRITE_SIZE"],
                                   "launches_per_step": len(agg[n]["FETCH_SIZE"]) / nfront}
                               for n, d in res.items() if n.startswith("k_") and "FETCH_SIZE" in d and "WRITE_SIZE" in d}
    json.dump(traffic, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
