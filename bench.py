#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MFCC hot path on MI355X.

Metric (BASELINE.json): audio frames/sec at 16 kHz, 25 ms / 10 ms, 512-pt FFT, 40 mel, 13 MFCC
+ delta + delta-delta.  Workload at every N: BASELINE configs[1] per GPU -- 1000 synthetic 10 s
utterances (998 000 frames) resident in HBM; one "step" = one pass of the whole hot path
(int16 PCM -> [frames][39] float features) over that batch, outputs left in HBM.  Ranks own
independent utterance shards (weak scaling, no data-path collective; torch.distributed is used
only for the barrier and the max-over-ranks of the timing).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

  --workload C2|C3|C4|C5|R|T   (R = the reference main()'s own defaults, ASR_OCL.cpp:560: 15 banks, 12 MFCC + c0, CVN)
  --scaling weak|strong         weak (default): every rank runs the workload's utterance count; strong: ONE job of
                                `n_utt_total` utterances (C4: BASELINE configs[3], 100 000) sharded round-robin over the
                                ranks by asr-featext-opencl_amd/sharding.py, every utterance's PCM a function of its index
                                only, so that N = 1, 2, 4, 8 compute the same job

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3
RCCL_PROBE_TIMEOUT_S = 60   # --collective rccl: a slow bring-up must not eat the driver's time limit

# What the counters say limits each front-end kernel, and the committed profile the reading comes from (DESIGN.md
# section 7).  roofline.bound stays the contract's nominal bound for this path ("hbm"); a kernel without a collected
# profile gets no diagnosis (the label was hardcoded for every workload).
BOUND_DIAGNOSED = {
    "k_front512": ("valu_issue", "profiles/r04/v3_C2_pmc_summary.json: 182 VALU wave-instr per frame, ~0.8 of the issue slots at 4 waves/SIMD"),
    "k_front1024": ("valu_issue+lds_latency", "profiles/r04/v3_C3_pmc_summary.json: ~0.68 of the issue slots at 4 waves/SIMD, LDS array "
                    "0.55 busy (eight LDS phases per frame)"),
    "k_front2048": ("valu_issue+lds_latency", "profiles/r04/v3_C5_pmc_summary.json: ~0.69 of the issue slots at 3 waves/SIMD "
                    "(register- and LDS-capped occupancy)"),
}

WORKLOADS = {
    # name: (n_utt, utt_samples, sample_rate, W, S, fft, nb, nc, dyn)
    "C2": dict(n_utt=1000, utt_samples=160000, sr=16000.0, W=400, S=160, fft=0, nb=40, nc=13, dyn=2,
               desc="1000 synthetic 16 kHz utterances x 10 s, 25 ms/10 ms, 512-pt FFT, 40 mel, 13 MFCC + d + dd"),
    "C4": dict(n_utt=12500, n_utt_total=100000, utt_samples=160000, sr=16000.0, W=400, S=160, fft=0, nb=40, nc=13, dyn=2,
               desc="per-GPU share of 100 000 synthetic 16 kHz utterances x 10 s sharded round-robin over 8 GPUs "
                    "(12 500 utterances, 4 GB of PCM per GPU), 512-pt FFT, 40 mel, 13 MFCC + d + dd",
               desc_strong="100 000 synthetic 16 kHz utterances x 10 s sharded round-robin over the GPUs, 512-pt FFT, "
                           "40 mel, 13 MFCC + d + dd"),
    "T": dict(n_utt=8, n_utt_total=13, utt_samples=16000, sr=16000.0, W=400, S=160, fft=0, nb=40, nc=13, dyn=2,
              desc="tiny test workload: 8 synthetic 16 kHz utterances x 1 s (launch-path tests only)",
              desc_strong="tiny test job: 13 synthetic 16 kHz utterances x 1 s sharded round-robin over the GPUs"),
    # the reference main()'s own defaults (ASR_OCL.cpp:560): 15 banks, 12 MFCC + c0, CVN, no deltas -- the configuration in
    # which a quarter of the reference's CPU profile is the normaliser (output_files/out.txt)
    "R": dict(n_utt=1000, utt_samples=160000, sr=16000.0, W=400, S=160, fft=0, nb=15, nc=12, c0=True, dyn=0, norm=2,
              desc="reference main() defaults (ASR_OCL.cpp:560): 1000 synthetic 16 kHz utterances x 10 s, 512-pt FFT, "
                   "15 mel, 12 MFCC + c0, CVN per utterance block, no deltas"),
    "C3": dict(n_utt=1, utt_samples=57600000, sr=16000.0, W=400, S=160, fft=1024, nb=80, nc=13, dyn=0,
               desc="one 1-hour 16 kHz stream, 1024-pt FFT, 80 mel, 13 MFCC"),
    "C5": dict(n_utt=200, utt_samples=441000, sr=44100.0, W=1102, S=441, fft=0, nb=128, nc=40, dyn=2, channels=2,
               desc="200 synthetic 44.1 kHz STEREO utterances x 10 s (downmix (L+R)>>1 in the kernel), 2048-pt FFT, "
                    "128 mel, 40 MFCC + d + dd"),
}


def synth_pcm_torch(torch, n_utt, utt_samples, sr, seed, device):
    """int16 PCM [n_utt, utt_samples]: clip16(round(3000*N(0,1) + 6000*sin(2*pi*f_u*n/sr))),
    f_u = 100 + 37*(u mod 64) Hz.  Generated on the device, in slabs."""
    g = torch.Generator(device=device)
    g.manual_seed(0x5EED0000 + seed)
    pcm = torch.empty((n_utt, utt_samples), dtype=torch.int16, device=device)
    n = torch.arange(utt_samples, device=device, dtype=torch.float32)
    slab = max(1, min(n_utt, (1 << 26) // max(utt_samples, 1)))
    for u0 in range(0, n_utt, slab):
        u1 = min(n_utt, u0 + slab)
        f = 100.0 + 37.0 * (torch.arange(u0, u1, device=device) % 64).to(torch.float32)
        x = 3000.0 * torch.randn((u1 - u0, utt_samples), generator=g, device=device, dtype=torch.float32)
        x += 6000.0 * torch.sin((2.0 * np.pi / sr) * f[:, None] * n[None, :])
        pcm[u0:u1] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
        del x
    return pcm


def synth_pcm_by_index(torch, utt_ids, utt_samples, sr, device):
    """The same signal model with every utterance a function of its GLOBAL index only (counter-based: a 32-bit integer hash
    of (utterance, sample) -> two uniforms -> Box-Muller), so that any sharding of one job sees the same data.
    utt_ids: int64 tensor/array of global utterance indices; returns int16 [len(utt_ids), utt_samples]."""
    ids = torch.as_tensor(np.asarray(utt_ids, dtype=np.int64), device=device)
    pcm = torch.empty((ids.numel(), utt_samples), dtype=torch.int16, device=device)
    n = torch.arange(utt_samples, device=device, dtype=torch.int64)
    nf = n.to(torch.float32)
    M = 0xFFFFFFFF

    def mix(x):   # lowbias32 (public domain integer hash), on int64 lanes masked to 32 bits
        x = x & M
        x = ((x ^ (x >> 16)) * 0x7FEB352D) & M
        x = ((x ^ (x >> 15)) * 0x846CA68B) & M
        return x ^ (x >> 16)

    slab = max(1, min(ids.numel(), (1 << 24) // max(utt_samples, 1)))
    for u0 in range(0, ids.numel(), slab):
        u = ids[u0:u0 + slab]
        key = mix((0x5EED0000 + u) & M)[:, None]
        h1 = mix(key ^ mix(2 * n + 1)[None, :])
        h2 = mix(key ^ mix(2 * n + 2)[None, :])
        u1 = (h1.to(torch.float32) + 1.0) * (1.0 / 4294967296.0)
        u2 = h2.to(torch.float32) * (1.0 / 4294967296.0)
        g = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * np.pi) * u2)
        f = 100.0 + 37.0 * (u % 64).to(torch.float32)
        x = 3000.0 * g + 6000.0 * torch.sin((2.0 * np.pi / sr) * f[:, None] * nf[None, :])
        pcm[u0:u0 + slab] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
        del h1, h2, u1, u2, g, x
    return pcm


def plan_job(wl, rank, world, scaling, sharding):
    """Which utterances this rank runs and where they lie in its PCM array: (global ids, offsets, lengths, total samples).
    weak: the workload's n_utt utterances per rank (ids rank*n_utt ..., fixed per-GPU work); strong: the rank's round-robin
    share of ONE job of n_utt_total utterances (sharding.shard_layout: rank r owns r, r + world, ...)."""
    if scaling == "strong":
        total = wl.get("n_utt_total", wl["n_utt"])
        lengths = np.full(total, wl["utt_samples"], dtype=np.int64)
        return sharding.shard_layout(lengths, rank, world)
    ids = np.arange(wl["n_utt"], dtype=np.int64) + rank * wl["n_utt"]
    off = np.arange(wl["n_utt"], dtype=np.int64) * wl["utt_samples"]
    ln = np.full(wl["n_utt"], wl["utt_samples"], dtype=np.int64)
    return ids, off, ln, int(wl["n_utt"] * wl["utt_samples"])


def work_model(wl, width, channels):
    """SURVEY 8(d) per-frame models: algorithmic flops, the fused path's bytes, and the bytes of a STAGED pipeline (one
    kernel per reference stage, every intermediate through HBM) for diagnosis."""
    W, S, nb = wl["W"], wl["S"], wl["nb"]
    W2 = wl["fft"] or (1 << (W - 1).bit_length())
    dl = wl["nc"] + (1 if wl.get("c0") else 0)
    cols = dl if wl["nc"] > 0 else nb
    lg = W2.bit_length() - 1
    f_front = W + 2.5 * W2 * lg + 6 * (W2 // 2 + 1) + (2 * nb * dl if wl["nc"] > 0 else 0)
    f_delta = 6 * cols * wl["dyn"]
    f_norm = 5 * cols * (1 + wl["dyn"]) if wl.get("norm") else 0
    staged = 2 * S * channels + 2 * 4 * W2 + 2 * 8 * (W2 // 2 + 1) + 2 * 4 * nb + 2 * 4 * cols * (1 + wl["dyn"]) + \
        (3 * 4 * width if wl.get("norm") else 0)
    return dict(flops_front=f_front, flops_delta=f_delta, flops_norm=f_norm, flops=f_front + f_delta + f_norm,
                staged_bytes=staged, fused_bytes=2 * S * channels + 4 * width)


def host_cpu_share():
    """Threads this process may really use: the cgroup CPU quota when there is one (a 1-GPU box
    exposes all host CPUs in the affinity mask but grants a share of them), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / float(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / float(per)))))
            break
        except Exception:
            continue
    how = "min(affinity mask, cgroup CPU quota) = %d" % n
    env = os.environ.get("MFX_CPU_THREADS")
    if env:
        n = max(1, int(env))
        how = "MFX_CPU_THREADS=%d" % n
    elif n > 64:
        # no quota visible on a many-core host: a 1-GPU box is granted 16 CPUs of the host (the pool's documented share)
        how = "capped to 16 (the 1-GPU box's CPU share): no cgroup quota visible, %d CPUs in the affinity mask" % n
        n = 16
    return n, how


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(orc, wl, pcm_host, window, budget_s=12.0):
    """Oracle (kind "port") timed on this host's cores over a bounded sample of the same workload."""
    import ctypes
    cores, cores_how = host_cpu_share()
    # a -march=native build of the same source if the compiler is here; else the portable one
    libpath = None
    try:
        import subprocess
        import tempfile
        d = tempfile.mkdtemp(prefix="orc_native_")
        out = os.path.join(d, "liboracle_native.so")
        r = subprocess.run(["gcc", "-std=c99", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared", "-o", out,
                            os.path.join(ROOT, "oracle", "mfcc_oracle.c"), "-lm"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode == 0:
            libpath = out
    except Exception:
        libpath = None
    cfg = orc.make_config(wl["utt_samples"] + 1000, window_size=wl["W"], shift=wl["S"], num_banks=wl["nb"],
                          sample_rate=wl["sr"], high_freq=wl["sr"] / 2, ceps_len=wl["nc"], want_c0=bool(wl.get("c0")),
                          norm=wl.get("norm", 0), dyn=wl["dyn"], delta_l1=3, delta_l2=3, fft_mode=1)
    n_avail = pcm_host.shape[0]

    def run(n_utt, threads, reps):
        frames, sec = orc.bench_batch(cfg, pcm_host[:n_utt], window, n_threads=threads, reps=reps, libpath=libpath)
        return frames / sec, sec, frames

    # single thread first (the reference itself is single-threaded: ASR_OCL.cpp:365-366)
    r1, _, _ = run(min(n_avail, 2), 1, 1)
    fpu = wl_frames(wl) // wl["n_utt"]
    n1 = int(max(1, min(n_avail, (budget_s * 0.25) * r1 / fpu)))
    r1, dt1, f1 = run(n1, 1, 1)
    # all host threads: repeat the resident sample until ~budget seconds of work
    n_all = n_avail
    probe_rate, _, _ = run(n_all, cores, 1)        # short probe so that the timed sample fits the budget
    reps = int(max(1, (budget_s * 0.75) * probe_rate / (fpu * n_all)))
    rall, dtall, fall = run(n_all, cores, reps)
    # FFTW itself is not installed: as an FFTW-class proxy, the FFT stage ALONE through scipy's
    # pocketfft (float32 real FFT of the workload's length over the same number of host threads)
    fft_proxy = None
    try:
        import scipy.fft
        W2 = wl["fft"] or (1 << (wl["W"] - 1).bit_length())
        x = np.random.default_rng(0).standard_normal((1 << 16, W2)).astype(np.float32)
        scipy.fft.rfft(x[:1024], axis=1, workers=cores)
        t0 = time.perf_counter()
        n_pass = 0
        while time.perf_counter() - t0 < 1.5:
            scipy.fft.rfft(x, axis=1, workers=cores)
            n_pass += 1
        fft_proxy = {"what": "scipy.fft.rfft (pocketfft), float32, %d-point, %d workers: the FFT stage alone" % (W2, cores),
                     "frames_per_s": n_pass * x.shape[0] / (time.perf_counter() - t0)}
    except Exception:
        fft_proxy = None
    return {
        "fft_stage_proxy": fft_proxy,
        "value": rall, "unit": "frames/s", "cores": cores, "kind": "port",
        "cores_derivation": cores_how, "os_cpu_count": os.cpu_count(), "cpu_model": cpu_model(),
        "sample": "%d utterances of the workload x %d passes = %d frames on all %d host threads (OpenMP, one "
                  "extractor per thread, setup untimed), %.1f s; oracle/mfcc_oracle.c with its float32 FFT, %s build"
                  % (n_all, reps, fall, cores, dtall, "-march=native" if libpath else "portable -mavx2"),
        "single_thread_value": r1, "single_thread_sample": "%d utterances, %.1f s" % (n1, dt1),
    }


KERNEL_SHORT_NAMES = ("k_front2048", "k_front512", "k_front1024", "k_front_reg", "k_front_wave", "k_delta16", "k_delta4", "k_delta",
                      "k_melcep", "k_norm_seg", "k_norm_stats", "k_norm_finalize", "k_norm_apply")


def measure_traffic_live(argv_tail, timeout_s=75):
    """HBM bytes per launch of every kernel of the step, measured NOW: two child runs of this same bench (3 steps, no settle, no
    CPU baseline) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter per pass, kernel trace only -- the form
    MI355X_MICROARCH.md prescribes), after the timed region.  Corrections as tools/pmc_run.sh: both counters are in KiB; on
    gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read (x2), WRITE_SIZE is exact at 32-byte sectors.
    Returns {kernel: {read, write, total, launches}} or None (no rocprofv3, a pass failed or ran out of time: the caller
    then falls back to the tracked profiles/traffic_latest.json and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="mfx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", MFX_BENCH_LIVE_TRAFFIC="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [tool, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--settle-ms", "0", "--no-cpu-baseline"] + argv_tail
            # (its own process group: on a timeout the profiler AND the benchmark under it are killed, by exact group id)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None
            if rc != 0:
                return None
            rows = 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "mfx::" not in name or row.get("Counter_Name") != counter:
                        continue
                    short = next((n for n in KERNEL_SHORT_NAMES if n in name), None)
                    if short is None:
                        continue
                    e = out.setdefault(short, {"FETCH_SIZE": [], "WRITE_SIZE": []})
                    e[counter].append(float(row["Counter_Value"]))
                    rows += 1
            if rows == 0:
                return None
    except Exception:   # noqa: BLE001 -- timeouts, a missing tool, an unreadable file: the tracked file is the fallback
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k, e in out.items():
        if not e["FETCH_SIZE"] or not e["WRITE_SIZE"]:
            return None
        rd = 2 * 1024 * sum(e["FETCH_SIZE"]) / len(e["FETCH_SIZE"])
        wr = 1024 * sum(e["WRITE_SIZE"]) / len(e["WRITE_SIZE"])
        res[k] = {"read": rd, "write": wr, "total": rd + wr, "launches": len(e["FETCH_SIZE"])}
    return res or None


def wl_frames(wl):
    per = (wl["utt_samples"] - (wl["W"] - wl["S"])) // wl["S"]
    return per * wl["n_utt"]


def spawn_ranks(args):
    """`bench.py --gpus N` without an external launcher: start the N ranks as children (one process per GPU under
    torch.distributed.run), relay rank 0's JSON line and return the children's status.  Runs BEFORE anything in
    this process touches the GPU (torch is not even imported yet)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout[-4000:])
        return r.returncode or 1
    if json.loads(line).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: the launched job reported n_gpus != --gpus\n")
        return 1
    print(line)
    return 0


def resolve_backend(collective, environ):
    """torch.distributed backend that carries the stopwatch's barrier for N > 1: "gloo" (default) or "nccl" (= RCCL, on
    request: --collective rccl); MFX_BENCH_BACKEND overrides the flag (tests)."""
    return environ.get("MFX_BENCH_BACKEND", {"rccl": "nccl"}.get(collective, collective))


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # The headline does not depend on these two: an untimed settle phase (--settle-ms of GPU work, reported as
    # settle_ms) runs before the warm-up steps, because for the first ~10 ms of sustained load an MI355X is still
    # raising its clocks (a bare 5 + 20 step run reads 15 % slower than 50 + 500; profiles/r01/README.md).
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="run untimed steps until this much wall time of back-to-back GPU work has passed (0: off)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: one job of the workload's n_utt_total utterances sharded round-robin over the ranks")
    ap.add_argument("--collective", default="gloo", choices=["gloo", "rccl"],
                    help="N > 1: what carries the barrier and the max-over-ranks (no collective is on the data path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the tracked profiles/traffic_latest.json instead of measuring it after the "
                         "timed region (two child runs under rocprofv3 --pmc, ~10 s each; N = 1 only)")
    ap.add_argument("--engine", type=int, default=0, help="mfx_config.engine bits (A/B of equivalent kernels; 0 = the library's choice)")
    ap.add_argument("--tail-split", type=int, default=0, help="mfx_config.tail_split (0 = default, -1 = off)")
    ap.add_argument("--overlap", action="store_true",
                    help="let the delta tail of a step overlap the next step's front end (mfx_batch_overlap); measured "
                         "neutral on MI355X -- the front end's waves fill the register file -- so off by default")
    return ap


def main():
    args = build_parser().parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    # test knob (rehearsing the N>1 path on a 1-GPU box): MFX_BENCH_DEVICE pins every rank to one device.
    # --collective (MFX_BENCH_BACKEND overrides it): what carries the stopwatch's barrier and the max-over-ranks.  The data
    # path has no collective (utterance shards are independent; north_star: "no RCCL needed"; the reference's own file
    # queue has none either, ASR_OCL.cpp:340-368), so the default is gloo on CPU scalars beside torch.cuda.synchronize():
    # an RCCL bring-up over 8 devices can take minutes of the driver's time limit and buys this benchmark nothing.
    # `--collective rccl` uses device tensors over RCCL, probed with a 60 s limit and an agreed fallback to gloo.
    dev_index = int(os.environ.get("MFX_BENCH_DEVICE", local_rank))
    backend = resolve_backend(args.collective, os.environ)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    group_dev = None   # the RCCL group (device tensors); None: the default gloo group carries the barriers
    if world > 1:
        import datetime
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        if backend == "nccl":
            # RCCL for the device-side barrier / reductions, gloo beside it: should RCCL fail to come up on some rank,
            # all ranks agree over gloo to finish on CPU tensors, and the line says so (config.collective_backend)
            bad = 0
            try:
                group_dev = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=RCCL_PROBE_TIMEOUT_S))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe, group=group_dev)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    bad = 1
            except Exception as e:   # noqa: BLE001 -- whatever RCCL raised, the benchmark itself does not need it
                sys.stderr.write("bench.py rank %d: RCCL did not initialise (%s); falling back to gloo\n"
                                 % (rank, " ".join(str(e).split())[:300]))
                bad = 1
            flag = torch.tensor([bad], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                backend = "gloo"
        elif backend != "gloo":
            raise SystemExit("bench.py: unknown collective backend %r" % backend)

    def barrier():   # device work drained first, then the ranks meet (device tensor over RCCL, or CPU tensor over gloo)
        torch.cuda.synchronize()
        t = torch.zeros(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, group=group_dev if backend == "nccl" else None)
        if backend == "nccl":
            torch.cuda.synchronize()

    pkg = G.load_package()
    wl = WORKLOADS[args.workload]
    W, S = wl["W"], wl["S"]
    window = pkg.reference_window(W)

    # ---- extractor and batch plan first (they need only the sizes), then the synthetic input
    channels = wl.get("channels", 1)
    m = pkg.MfccHip(wl["utt_samples"] + 1000, W, S, wl["nb"], wl["sr"], 64.0, wl["sr"] / 2, wl["nc"], bool(wl.get("c0")),
                    22.0, wl.get("norm", pkg.NORM_NONE), wl["dyn"], 3, 3, True, device=dev_index, fft_size=wl["fft"],
                    channels=channels, engine=args.engine, tail_split=args.tail_split)
    m.set_window(window)
    strong = args.scaling == "strong"
    if strong and "n_utt_total" not in wl:
        raise SystemExit("bench.py: --scaling strong needs a workload with a job size (C4, T)")
    utt_ids, offsets, lengths, n_samples_plan = plan_job(wl, rank, world, args.scaling, pkg.sharding)
    n_utt_rank = int(utt_ids.size)
    if args.overlap:
        m.batch_overlap(True)   # consecutive steps pipeline: tail of step i beside the front end of step i+1
    rows, total_rows = m.batch_plan(offsets, lengths)
    width = m.get_output_data_width()
    out = torch.empty((max(total_rows, 1), width), dtype=torch.float32, device=device)
    # synthetic input resident in HBM (per rank: its own shard of utterances)
    if strong:
        pcm = synth_pcm_by_index(torch, utt_ids, wl["utt_samples"], wl["sr"], device)
    else:
        pcm = synth_pcm_torch(torch, wl["n_utt"], wl["utt_samples"], wl["sr"], seed=rank, device=device)
    if channels == 2:   # interleaved L/R: the right channel is the left one of the neighbouring utterance
        pcm = torch.stack((pcm, torch.roll(pcm, 1, dims=0)), dim=2).contiguous()
    n_samples = pcm.numel() // channels     # per channel
    assert n_samples >= n_samples_plan
    torch.cuda.synchronize()

    def step():
        m.batch_run_device(pcm.data_ptr(), n_samples, out.data_ptr())

    # settle: untimed back-to-back steps until the clocks have stopped rising (time based, so the headline does not
    # depend on the caller's step counts); batches of steps are queued without a host sync in between
    settle_ms, settle_steps = 0.0, 0
    if args.settle_ms > 0:
        step()
        m.synchronize()
        t_s = time.perf_counter()
        step()
        m.synchronize()
        one = max(time.perf_counter() - t_s, 1e-5)
        burst = int(max(1, min(1000, 0.01 / one)))       # ~10 ms of queued work per host sync
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            for _ in range(burst):
                step()
            m.synchronize()
            settle_steps += burst
        settle_ms = (time.perf_counter() - t_s) * 1e3
    for _ in range(args.warmup):
        step()
    m.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        barrier()
    m.profile_enable(True)
    m.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    m.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group_dev if backend == "nccl" else None)
        elapsed = float(t.item())
    launches, kernel_ms = m.profile_read(reset=True)
    m.profile_enable(False)

    frames_rank = int(total_rows)
    if dist is not None:   # strong scaling: the shards differ by one utterance at most, the total is the sum over ranks
        t = torch.tensor([frames_rank], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group_dev if backend == "nccl" else None)
        frames_all = int(t.item())
    else:
        frames_all = frames_rank
    ms_per_step = 1e3 * elapsed / args.steps
    value = frames_all / (elapsed / args.steps)

    # ---- roofline of the dominant kernel (HIP events on the handle's stream around that kernel)
    bytes_in = 2 * S * channels
    cols = (wl["nc"] + (1 if wl.get("c0") else 0)) if wl["nc"] > 0 else wl["nb"]
    kname = m.dominant_kernel_name()
    model = work_model(wl, width, channels)
    # the front-end kernel reads each PCM sample once and writes the static coefficients once;
    # the (small) delta kernel adds the remaining 4*2*cols B/frame of output
    kernel_bytes_per_frame = bytes_in + 4 * cols
    path_bytes_per_frame = bytes_in + 4 * width
    avg_kernel_ms = kernel_ms / max(launches, 1)
    launches_per_step = max(launches, 1) / args.steps
    achieved = (kernel_bytes_per_frame * frames_rank / launches_per_step) / (avg_kernel_ms * 1e-3) / 1e9 if avg_kernel_ms > 0 else 0.0
    # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, collected in their own rocprofv3 --pmc runs of
    # this same command and corrected as MI355X_MICROARCH.md prescribes): NOT measured in this run -- read from the
    # tracked profiles/traffic_latest.json, per workload and kernel, and labelled so
    traffic, traffic_source, step_traffic = None, None, None
    live = None
    if (world == 1 and not args.no_live_traffic and not args.no_cpu_baseline and os.environ.get("MFX_BENCH_LIVE_TRAFFIC", "1") != "0"):
        # (the full default line only: A/B loops, profiler runs and the multi-rank path keep the tracked file)
        tail = ["--workload", args.workload, "--engine", str(args.engine), "--tail-split", str(args.tail_split)]
        live = measure_traffic_live(tail)
    if live and kname in live:
        front = max(live[kname]["launches"], 1)
        traffic = live[kname]["total"]
        traffic_source = ("measured in this run, after the timed region: two child runs of this command (3 steps) under "
                          "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (x2 / x1, KiB; MI355X_MICROARCH.md), per launch")
        step_traffic = {"bytes_per_step": sum(v["total"] * v["launches"] / front for v in live.values()),
                        "kernels": {k: v["total"] for k, v in live.items()}, "source": traffic_source}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if traffic is None and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(args.workload) if isinstance(tj.get(args.workload), dict) else tj
            if ent.get("workload", args.workload) == args.workload and ent.get("kernel") == kname:
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic_latest.json (%s)" % ent.get("source", "rocprofv3 --pmc passes")
                # every kernel of the step (front end + delta (+ normaliser)): counter bytes per launch x launches per step
                sk = ent.get("step_kernels")
                if isinstance(sk, dict) and sk:
                    step_traffic = {"bytes_per_step": sum(v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1) for v in sk.values()),
                                    "kernels": {k: v["hbm_bytes_per_launch"] for k, v in sk.items()},
                                    "source": traffic_source}
        except Exception:
            traffic = None
    frames_per_launch = frames_rank / launches_per_step
    kernel_tflops = model["flops_front"] * frames_per_launch / (avg_kernel_ms * 1e-3) / 1e12 if avg_kernel_ms > 0 else 0.0
    path_tflops = model["flops"] * frames_rank / (ms_per_step * 1e-3) / 1e12
    staged_gbs = model["staged_bytes"] * frames_rank / (ms_per_step * 1e-3) / 1e9
    roofline = {
        # `achieved` / `peak` / `frac` price the dominant kernel against HBM (the contract's nominal bound for this path).
        # What the counters say limits it is vector-instruction issue + dependent LDS round trips:
        # fully fused, the path has ~32 flop/B against a machine balance of ~20 flop/B, so 60 % of the HBM roofline
        # would need ~100 % of the FP32 vector peak.  Both fractions are reported.
        "bound": "hbm",
        "bound_diagnosed": ({"label": BOUND_DIAGNOSED[kname][0], "source": BOUND_DIAGNOSED[kname][1]}
                            if kname in BOUND_DIAGNOSED else None),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "kernel": kname, "kernel_avg_ms": avg_kernel_ms, "kernel_launches_per_step": launches_per_step,
        "algorithmic_bytes_per_frame": kernel_bytes_per_frame,
        "flops_per_frame": model["flops"], "kernel_flops_per_frame": model["flops_front"],
        "achieved_tflops": kernel_tflops, "peak_fp32_tflops": FP32_PEAK_TFLOPS,
        "frac_fp32_peak": kernel_tflops / FP32_PEAK_TFLOPS,
        "arithmetic_intensity_flop_per_byte": model["flops"] / model["fused_bytes"],
        "machine_balance_flop_per_byte": FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9),
        "staged_bytes_per_frame": model["staged_bytes"],
        "staged_pipeline_equivalent": {"achieved": staged_gbs, "frac": staged_gbs / HBM_PEAK_GBS,
                                       "what": "HBM bytes a one-kernel-per-reference-stage pipeline would move per "
                                               "frame x this run's frame rate (diagnostic, not traffic)"},
        "whole_path": {"bytes_per_frame": path_bytes_per_frame,
                       "achieved": path_bytes_per_frame * frames_rank / (ms_per_step * 1e-3) / 1e9,
                       "frac": path_bytes_per_frame * frames_rank / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "achieved_tflops": path_tflops, "frac_fp32_peak": path_tflops / FP32_PEAK_TFLOPS,
                       # counter bytes of ALL kernels of a step against the algorithmic bytes of the step
                       "traffic": step_traffic["bytes_per_step"] if step_traffic else None,
                       "traffic_kernels": step_traffic["kernels"] if step_traffic else None,
                       "traffic_over_algorithmic": (step_traffic["bytes_per_step"] / (path_bytes_per_frame * frames_rank)
                                                    if step_traffic else None)},
    }

    result = {
        "metric": "audio frames/sec (16 kHz, 25 ms/10 ms, 512-pt FFT, 40 mel, 13 MFCC + delta + delta-delta)"
                  if args.workload == "C2" else "audio frames/sec (%s)" % wl["desc"],
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle_ms": settle_ms, "settle_steps": settle_steps,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("%s: %s; one job over all GPUs, resident in HBM" % (args.workload, wl["desc_strong"])) if strong
                               else "%s: %s; per GPU, resident in HBM" % (args.workload, wl["desc"]),
                   "frames_per_step": frames_all, "frames_rank0_per_step": frames_rank, "utterances_rank0": n_utt_rank,
                   "utterances_total": wl["n_utt_total"] if strong else wl["n_utt"] * world,
                   "utterance_ids_rank0_head": [int(v) for v in utt_ids[:4]],
                   "collective_backend": ("rccl" if backend == "nccl" else backend) if dist is not None else None,
                   "sharding": "round-robin utterance shards (rank r owns r, r + N, ...), no collective" if strong
                               else "independent utterance shards per rank, no collective",
                   "step_pipelining": "delta tail of step i overlaps front end of step i+1" if args.overlap else "off"},
        "roofline": roofline,
    }

    if rank == 0 and not args.no_cpu_baseline:   # rank 0 only, the same bounded sample at every N
        orc = G.load_oracle()
        n_host = min(n_utt_rank, 512)
        pcm_host = pcm[:n_host].cpu().numpy()
        wl_cpu, cut = wl, False
        if wl["n_utt"] == 1 and wl["utt_samples"] >= 64 * 160000:
            # one long stream: the CPU threads each take 10 s pieces of it (independent extractors, as for the
            # utterance workloads; the W - S samples of overlap lost at each cut are 0.2 % of the frames)
            seg = 160000
            pcm_host = pcm_host[:, :(pcm_host.shape[1] // seg) * seg].reshape(-1, seg, *pcm_host.shape[2:])[:512]
            wl_cpu, cut = dict(wl, n_utt=pcm_host.shape[0], utt_samples=seg), True
        elif pcm_host.shape[0] != wl["n_utt"]:
            wl_cpu = dict(wl, n_utt=pcm_host.shape[0])
        if channels == 2:   # the same downmix the kernel applies, done before the timed CPU passes
            pcm_host = ((pcm_host[..., 0].astype(np.int32) + pcm_host[..., 1].astype(np.int32)) >> 1).astype(np.int16)
        result["cpu_baseline"] = cpu_baseline(orc, wl_cpu, pcm_host, window)
        if cut:
            result["cpu_baseline"]["sample"] = "the stream cut into 10 s pieces: " + result["cpu_baseline"]["sample"]
        result["cpu_baseline"]["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    m.close()
    if rank == 0:
        print(json.dumps(result), flush=True)   # (before the teardown: a hiccup there must not lose the line)
    if dist is not None:
        try:
            barrier()
            dist.destroy_process_group()
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench.py rank %d: process-group teardown: %s\n" % (rank, str(e)[:200]))


if __name__ == "__main__":
    main()
