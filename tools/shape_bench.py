"""Dev aid: batch-entry throughput of a few configurations outside BASELINE's five (which kernel serves them, frames/s)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
import torch
SHAPES = [  # name, sr, W, S, nb, nc, channels, seconds, n_utt
    ("16 kHz fbank-80: 25/10 ms, 512-pt, 80 log mel energies, no deltas", 16000.0, 400, 160, 80, 0, 1, 10, 1000),
    ("16 kHz fbank-40 + d + dd", 16000.0, 400, 160, 40, 0, 1, 10, 1000),
    ("22.05 kHz fbank-80 (1024-pt, 25/10 ms), no deltas", 22050.0, 551, 220, 80, 0, 1, 10, 500),
    ("8 kHz telephony, 25/10 ms, 256-pt, 23 mel, 13 MFCC + d + dd", 8000.0, 200, 80, 23, 13, 1, 10, 2000),
    ("11.025 kHz, 25/10 ms, 512-pt, 26 mel", 11025.0, 275, 110, 26, 13, 1, 10, 1000),
    ("16 kHz (C2)", 16000.0, 400, 160, 40, 13, 1, 10, 1000),
    ("22.05 kHz, 25/10 ms, 1024-pt, 40 mel", 22050.0, 551, 220, 40, 13, 1, 10, 500),
    ("32 kHz, 25/10 ms, 1024-pt, 64 mel", 32000.0, 800, 320, 64, 13, 1, 10, 400),
    ("44.1 kHz mono, 25/10 ms, 2048-pt, 128 mel, 40 MFCC", 44100.0, 1102, 441, 128, 40, 1, 10, 200),
    ("48 kHz mono, 25/10 ms, 2048-pt, 128 mel, 40 MFCC", 48000.0, 1200, 480, 128, 40, 1, 10, 200),
    ("44.1 kHz mono, n_fft = win = 2048, hop 512, 128 mel, 40 MFCC", 44100.0, 2048, 512, 128, 40, 1, 10, 200),
    ("8 kHz STEREO (downmix in the kernel), 256-pt, 23 mel", 8000.0, 200, 80, 23, 13, 2, 10, 2000),
    ("16 kHz STEREO (downmix in the kernel), 512-pt, 40 mel", 16000.0, 400, 160, 40, 13, 2, 10, 1000),
    ("44.1 kHz STEREO (C5)", 44100.0, 1102, 441, 128, 40, 2, 10, 200),
]
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
for name, sr, W, S, nb, nc, ch, sec, n_utt in SHAPES:
    if ONLY and not name.startswith(ONLY):
        continue
    n = int(sr * sec)
    n += n & 1
    dyn = 0 if "no deltas" in name else 2
    m = pkg.MfccHip(n + 1000, W, S, nb, sr, 64.0, sr / 2, nc, False, 22.0, 0, dyn, 3, 3, True, channels=ch)
    m.set_window(pkg.reference_window(W))
    rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
    shape = (n_utt, n) if ch == 1 else (n_utt, n, 2)
    pcm = (3000 * torch.randn(shape, device="cuda")).to(torch.int16)
    out = torch.empty((total, m.get_output_data_width()), device="cuda")
    for _ in range(20):
        m.batch_run_device(pcm.data_ptr(), n_utt * n, out.data_ptr())
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        m.batch_run_device(pcm.data_ptr(), n_utt * n, out.data_ptr())
    m.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print("%-62s %-12s %7.4f ms per %8d frames = %6.3f G frames/s (%5.0f x real time per GPU-second: %.0f h of audio)" % (
        name, m.dominant_kernel_name(), dt * 1e3, total, total / dt / 1e9, total * S / sr / dt, total * S / sr / dt / 3600))
    m.close()
