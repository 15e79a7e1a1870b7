"""Dev aid: run the C2 workload on a -DMFX_STAMPS build (MFX_LIB=...libmfcchip_stamps.so) and print
the share of wave time per kernel phase."""
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
raw = m.debug_read(4)[:4096 * 16].view(np.uint64).reshape(4096, 8).astype(np.float64)
names = ["loop-top/prefetch", "load+cvt+window", "passA+twiddle", "transpose", "passB", "split+mag", "mel+dct+store", "-"]
tot = raw[:, :7].sum(1)
rt = raw[:, 7]
print("shader clock over the wave's life: median %.3f GHz (min %.3f, max %.3f); wave life median %.1f us, max %.1f us" % (
    np.median(tot / rt) * 0.1, (tot / rt).min() * 0.1, (tot / rt).max() * 0.1, np.median(rt) / 100.0, rt.max() / 100.0))
q = np.percentile(rt / 100.0, [0, 5, 25, 50, 75, 95, 100])
print("wave life percentiles (us): min %.0f p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f" % tuple(q))
print("waves with data:", int((tot > 0).sum()), "mean cycles per wave:", tot[tot > 0].mean())
for i in range(7):
    print("%-18s %6.1f %%   %9.0f cycles/wave" % (names[i], 100 * raw[:, i].sum() / tot.sum(), raw[:, i][tot > 0].mean()))
