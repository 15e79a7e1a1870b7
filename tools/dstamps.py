"""Dev aid: C2 workload on a -DMFX_DSTAMPS build (MFX_LIB=...): where the fused kernel's delta wave spends its life."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
n_utt, n = 1000, 160000
dev = torch.device("cuda", 0)
pcm = (3000.0 * torch.randn((n_utt, n), device=dev)).round().clamp(-32768, 32767).to(torch.int16)
m = pkg.MfccHip(n + 1000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, 0, 2, 3, 3, True)
m.set_window(pkg.reference_window(400))
rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
out = torch.empty((total, 39), dtype=torch.float32, device=dev)
for _ in range(3):
    m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
m.synchronize()
raw = m.debug_read(4)[:256 * 16].view(np.uint64).reshape(256, 8).astype(np.float64)
us = raw[:, :3] / 100.0
print("delta wave per block (us): wait mean %.1f  work mean %.1f  life mean %.1f max %.1f; tiles mean %.1f" % (
    us[:, 0].mean(), us[:, 1].mean(), us[:, 2].mean(), us[:, 2].max(), raw[:, 3].mean()))
print("work per tile: %.2f us" % (us[:, 1].sum() / raw[:, 3].sum()))
ph = raw[:, 4:8].sum(0) / raw[:, 3].sum() / 100.0
print("phases per tile (us): issue loads %.2f | loads land + stage + statics out %.2f | delta %.2f | accel %.2f" % tuple(ph))
