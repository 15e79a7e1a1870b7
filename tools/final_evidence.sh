#!/bin/bash
# Collects a round's evidence on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   smoke; C2: rocprofv3 kernel stats of the default bench, PMC passes + HBM traffic, default line with CPU baseline,
#   the driver's short protocol (--steps 20 --warmup 5); C3 / C5: line with CPU baseline, kernel stats, PMC passes +
#   traffic; C4 line; the 2-rank self-launch rehearsed on one GPU.      tools/final_evidence.sh <tag> [parts]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-final}; PARTS=${2:-smoke,c2,c3,c5,r,c4,ranks,stream,afet,hostbatch,sweep,stamps}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
has() { case ",$PARTS," in *",$1,"*) return 0;; esac; return 1; }
stats() { # <name> <bench args...>
  local n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$n" -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench_${n}_under_rocprof.json" 2> "$O/rocprof_$n.err") || return 1
  cp "$O"/stats_$n/*/*kernel_stats.csv "$O/${n}_kernel_stats.csv"
}
merge_traffic() { # <workload> <traffic.json>
  python3 - "$R/profiles/traffic_latest.json" "$1" "$2" <<'PY'
import json, sys
path, wl, src = sys.argv[1:4]
try: cur = json.load(open(path))
except Exception: cur = {}
if "kernel" in cur: cur = {cur.get("workload", "C2"): cur}      # older single-entry form
t = json.load(open(src)); t["source"] = "rocprofv3 --pmc passes (tools/pmc_run.sh), FETCH_SIZE x2 + WRITE_SIZE, KiB"
cur[wl] = t
json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
PY
}
cd "$R"
if has smoke; then timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }; tail -1 "$O/smoke.log"; fi
if has c2; then
  stats C2 --steps 500 --warmup 50 || exit 1
  tools/pmc_run.sh "$TAG/pmc_C2" > "$O/pmc_C2_stdout.log" 2>&1 || exit 1
  merge_traffic C2 "$O/pmc_C2/traffic.json"
  timeout -k 10 400 python3 bench.py > "$O/bench_C2_default.json" 2> "$O/bench_C2_default.err" || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_C2_steps20_warmup5.json" 2>> "$O/bench_C2_default.err" || exit 1
  tail -c 900 "$O/bench_C2_default.json"; echo; head -3 "$O/C2_kernel_stats.csv" | cut -c1-150
fi
for w in C3 C5 R; do
  lw=$(echo $w | tr 'A-Z' 'a-z')
  if has $lw; then
    stats $w --workload $w --steps 100 --warmup 20 || exit 1
    tools/pmc_run.sh "$TAG/pmc_$w" --workload $w > "$O/pmc_${w}_stdout.log" 2>&1 || exit 1
    merge_traffic $w "$O/pmc_$w/traffic.json"
    timeout -k 10 400 python3 bench.py --workload $w --steps 100 --warmup 20 > "$O/bench_$w.json" 2> "$O/bench_$w.err" || exit 1
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('cpu_baseline',{}).get('value'))" "$O/bench_$w.json" $w
  fi
done
if has c4; then
  timeout -k 10 400 python3 bench.py --workload C4 --steps 30 --warmup 5 --no-cpu-baseline > "$O/bench_C4.json" 2> "$O/bench_C4.err" || exit 1; python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('C4', d['value'], d['ms_per_step'])" "$O/bench_C4.json"
  # BASELINE configs[3] as ONE job (100 000 utterances, 32 GB of PCM) on this one GPU: the N = 1 point of --scaling strong
  timeout -k 10 500 python3 bench.py --workload C4 --scaling strong --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_C4_strong_n1.json" 2> "$O/bench_C4_strong.err" || exit 1; python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('C4 strong N=1', d['value'], d['ms_per_step'], d['config']['utterances_total'])" "$O/bench_C4_strong_n1.json"
fi
if has ranks; then MFX_BENCH_DEVICE=0 MFX_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline > "$O/bench_C2_gpus2_one_device_gloo.json" 2> "$O/bench_gpus2.err" || exit 1; python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('gpus2 (one device)', d['n_gpus'], d['value'], d['ms_per_step'])" "$O/bench_C2_gpus2_one_device_gloo.json"; fi
if has stream; then timeout -k 10 300 python3 tools/stream_bench.py > "$O/stream_bench.txt" 2>&1 || exit 1; grep "C ABI" "$O/stream_bench.txt" | head -2; fi
if has afet; then tools/afet_bench2.sh 2048 > "$O/afet_bench.txt" 2>&1 || exit 1; grep "files/s" "$O/afet_bench.txt"; fi
if has sweep; then timeout -k 10 300 python3 tools/melcep_sweep_bench.py 2>&1 | grep -v amdgpu.ids > "$O/melcep_sweep.txt" || exit 1; cat "$O/melcep_sweep.txt"; fi
if has stamps && [ -f build/var/lib_stamps.so ]; then MFX_LIB=build/var/lib_stamps.so timeout -k 10 300 python3 tools/stamps2048.py 2>&1 | grep -v amdgpu.ids > "$O/stamps2048.txt"; head -3 "$O/stamps2048.txt"; fi
if has hostbatch; then timeout -k 10 300 python3 tools/host_batch_bench.py > "$O/host_batch_bench.txt" 2>&1 || exit 1; grep "host buffers" "$O/host_batch_bench.txt"; fi
