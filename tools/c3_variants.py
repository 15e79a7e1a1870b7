"""Dev aid: the C3 shape (1024 points, 400-tap window, 80 mel, 13 MFCC) with even / odd shifts and offsets, through
k_front1024 and through k_front_reg (mfx_config.engine = MFX_ENGINE_NO_FRONT1024)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
dev = torch.device("cuda", 0)
n = 57_600_000
pcm = (3000.0 * torch.randn((n,), device=dev)).round().clamp(-32768, 32767).to(torch.int16)
for name, W, S, off in (("shift 160, even offset", 400, 160, 0), ("shift 161 (odd)", 400, 161, 0), ("shift 160, odd offset", 400, 160, 1),
                        ("512 taps, shift 160", 512, 160, 0), ("768 taps, shift 160", 768, 160, 0), ("1024 taps, shift 160", 1024, 160, 0)):
    for no in ("0", "1"):
        m = pkg.MfccHip(n + 1000, W, S, 80, 16000.0, 64.0, 8000.0, 13, False, 22.0, 0, 0, 3, 3, True, fft_size=1024, engine=int(no))
        m.set_window(pkg.reference_window(W))
        rows, total = m.batch_plan(np.array([off], dtype=np.int64), np.array([n - off], dtype=np.int64))
        out = torch.empty((total, m.get_output_data_width()), dtype=torch.float32, device=dev)
        for _ in range(30):
            m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
        m.synchronize()
        t0 = time.perf_counter()
        K = 100
        for _ in range(K):
            m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
        m.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("%-26s %-12s %7.4f ms/step  %.1f M frames/s" % (name, m.dominant_kernel_name(), dt * 1e3, total / dt / 1e6))
        m.close()
