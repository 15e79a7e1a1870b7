"""Dev aid: run the C5 workload on a -DMFX_STAMPS build (MFX_LIB=build/var/lib_stamps.so) and print the share of wave
time per phase of k_front2048."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
pkg = G.load_package()
n_utt, n = 200, 441000
dev = torch.device("cuda", 0)
pcm = (3000.0 * torch.randn((n_utt, n, 2), device=dev)).round().clamp(-32768, 32767).to(torch.int16)
m = pkg.MfccHip(n + 1000, 1102, 441, 128, 44100.0, 64.0, 22050.0, 40, False, 22.0, 0, 2, 3, 3, True, channels=2)
m.set_window(pkg.reference_window(1102))
rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
out = torch.empty((total, 120), dtype=torch.float32, device=dev)
for _ in range(3):
    m.batch_run_device(pcm.data_ptr(), n_utt * n, out.data_ptr())
m.synchronize()
NW = 256 * 12
raw = m.debug_read(4)[:NW * 28].view(np.uint64).reshape(NW, 14).astype(np.float64)
names = ["loop top", "convert + window", "pass 1 + twiddle", "transposition", "pass 2", "(after the prefetch issue: sync)",
         "mel walk + log", "DCT + store", "chunk switch", None, "split: partner fetch + pairs + magnitude writes", "bin 512 + prefetch issue", "wait for the prefetched samples", None]
idx = [0, 12, 1, 2, 3, 4, 10, 11, 5, 6, 7, 8]
tot = raw[:, idx].sum(1)
rt = raw[:, 9]
ok = tot > 0
print("shader clock over the wave's life: median %.3f GHz; wave life median %.1f us, max %.1f us" % (
    np.median(tot[ok] / rt[ok]) * 0.1, np.median(rt[ok]) / 100.0, rt[ok].max() / 100.0))
q = np.percentile(rt[ok] / 100.0, [0, 5, 25, 50, 75, 95, 100])
print("wave life percentiles (us): min %.0f p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f" % tuple(q))
iters = total / 2.0 / ok.sum()
print("waves with data:", int(ok.sum()), " mean cycles per wave: %.0f  (%.1f iterations per wave -> %.0f cycles per iteration)" % (
    tot[ok].mean(), iters, tot[ok].mean() / iters))
for i in idx:
    print("%-38s %6.1f %%   %9.0f cycles/wave  %7.0f cycles/iteration" % (names[i], 100 * raw[:, i].sum() / tot.sum(), raw[:, i][ok].mean(),
                                                                          raw[:, i][ok].mean() / iters))
