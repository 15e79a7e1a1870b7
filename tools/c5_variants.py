"""Dev aid: the C5 shape with different input layouts (paired 32-bit loads / single samples / stereo)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
dev = torch.device("cuda", 0)
n_utt, n = 200, 441000
for name, S, ch in (("mono, shift 440 (paired loads)", 440, 1), ("mono, shift 441 (single samples)", 441, 1),
                    ("stereo, shift 441", 441, 2)):
    pcm = (3000.0 * torch.randn((n_utt, n, ch), device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    m = pkg.MfccHip(n + 1000, 1102, S, 128, 44100.0, 64.0, 22050.0, 40, False, 22.0, 0, 2, 3, 3, True, channels=ch)
    m.set_window(pkg.reference_window(1102))
    rows, total = m.batch_plan(np.arange(n_utt, dtype=np.int64) * n, np.full(n_utt, n, dtype=np.int64))
    out = torch.empty((total, 120), dtype=torch.float32, device=dev)
    for _ in range(30):
        m.batch_run_device(pcm.data_ptr(), pcm.numel() // ch, out.data_ptr())
    m.synchronize()
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        m.batch_run_device(pcm.data_ptr(), pcm.numel() // ch, out.data_ptr())
    m.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-36s %7.3f ms/step  %.1f M frames/s" % (name, dt * 1e3, total / dt / 1e6))
    m.close()
