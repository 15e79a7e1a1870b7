"""Dev aid: the C2 shape with even / odd shifts and utterance offsets (aligned / unaligned builds of k_front512)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
dev = torch.device("cuda", 0)
n_utt, n = 1000, 160000
pcm = (3000.0 * torch.randn((n_utt * n + 16,), device=dev)).round().clamp(-32768, 32767).to(torch.int16)
for name, W, S, off in (("shift 160, even offsets", 400, 160, 0), ("shift 161 (odd)", 400, 161, 0), ("shift 160, odd offsets", 400, 160, 1),
                        ("512 taps, shift 160", 512, 160, 0)):
    m = pkg.MfccHip(n + 1000, W, S, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, 0, 2, 3, 3, True)
    m.set_window(pkg.reference_window(W))
    rows, total = m.batch_plan(np.arange(n_utt, dtype=np.int64) * n + off, np.full(n_utt, n - 2, dtype=np.int64))
    out = torch.empty((total, m.get_output_data_width()), dtype=torch.float32, device=dev)
    for _ in range(30):
        m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
    m.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
    m.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-26s %-12s %7.4f ms/step  %.1f M frames/s" % (name, m.dominant_kernel_name(), dt * 1e3, total / dt / 1e6))
    m.close()
