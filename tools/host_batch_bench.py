"""Dev aid: PCIe-inclusive rate of the batch entry (mfx_batch_run_host: host int16 in, host float32 out), C2 shape."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
n_utt, n = 1000, 160000
for pinned in (False, True):
    t_in = torch.empty((n_utt, n), dtype=torch.int16, pin_memory=pinned)
    t_in.copy_((3000.0 * torch.randn((n_utt, n))).round().clamp(-32768, 32767).to(torch.int16))
    m = pkg.MfccHip(n + 1000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, 0, 2, 3, 3, True)
    m.set_window(pkg.reference_window(400))
    rows, total = m.batch_plan(np.arange(n_utt, dtype=np.int64) * n, np.full(n_utt, n, dtype=np.int64))
    t_out = torch.empty((total, 39), dtype=torch.float32, pin_memory=pinned)
    import ctypes as C
    L = pkg.load_library()
    def run():
        rc = L.mfx_batch_run_host(m._h, C.cast(t_in.data_ptr(), C.POINTER(C.c_int16)), n_utt * n,
                                  C.cast(t_out.data_ptr(), C.POINTER(C.c_float)))
        assert rc == 0, rc
    for _ in range(3):
        run()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        run()
    dt = (time.perf_counter() - t0) / K
    print("%s host buffers: %.2f ms per 998000-frame batch = %.1f M frames/s (%.1f GB/s of in + out)" % (
        "pinned" if pinned else "pageable", dt * 1e3, total / dt / 1e6, (n_utt * n * 2 + total * 156) / dt / 1e9))
    m.close()
