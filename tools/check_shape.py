"""Dev aid (GPU box): ONE configuration through the batch entry and the streaming interface, each column group reported three ways --
product against the CPU checker (the north-star bar: 1e-4 of the group's scale, 1e-5 relative L2), the checker against the same
formulas in float64 on the same tables (float32's own noise for this shape), and the product against that float64 result.  A
fuzz finding on an exotic shape (many narrow filters, a large c0 beside small deltas) is a defect only if the product is further
from float64 than the checker is.

    python tools/check_shape.py W S sample_rate num_banks ceps_len c0 dyn l1 l2 [fft_size] [frames] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402
from fuzz_common import truth64  # noqa: E402

pkg = G.load_package()
orc = G.load_oracle()
a = sys.argv[1:]
W, S, sr, nb, nc, c0, dyn, l1, l2 = int(a[0]), int(a[1]), float(a[2]), int(a[3]), int(a[4]), bool(int(a[5])), int(a[6]), int(a[7]), int(a[8])
fft_size = int(a[9]) if len(a) > 9 else 0
T = int(a[10]) if len(a) > 10 else 200
rng = np.random.default_rng(int(a[11]) if len(a) > 11 else 1)
n = (T - 1) * S + W
pcm = (4000.0 * rng.standard_normal(n)).round().clip(-32768, 32767).astype(np.int16)
window = pkg.reference_window(W)
Wo = fft_size or W
wo = np.zeros(Wo, np.float32)
wo[:W] = window
seg = np.concatenate([pcm, np.zeros(Wo - W, np.int16)]) if fft_size else pcm
groups = 1 + dyn
cfg = orc.make_config(n + 4 * Wo, window_size=Wo, shift=S, num_banks=nb, sample_rate=sr, high_freq=sr / 2, ceps_len=nc, want_c0=c0,
                      dyn=dyn, delta_l1=l1, delta_l2=l2)
want = orc.run_utterance(cfg, seg, wo, bug_compat=False).astype(np.float64)
truth = truth64(pkg, seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, 1.0, 64.0, sr / 2)[:want.shape[0]]
m = pkg.MfccHip(n + 4 * W, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, 0, dyn, l1, l2, True, fft_size=fft_size, bug_compat=False)
m.set_window(window)
rows, total = m.batch_plan([0], [n])
got = m.batch_run_host(pcm)[:want.shape[0]].astype(np.float64)
print("kernel", m.dominant_kernel_name(), "rows", want.shape)
m.close()
# the streaming interface, three blocks
ms = pkg.MfccHip(n // 3 + 2 * W, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, 0, dyn, l1, l2, True, fft_size=fft_size, bug_compat=True)
ms.set_window(window)
parts, pos, ibs = [], 0, ms.get_input_buffer_size()
while pos < n:
    k = ms.set_input(pcm[pos:pos + ibs])
    pos += ibs
    if k > 0:
        ms.apply()
        parts.append(ms.get_output_data(k))
k = ms.flush()
if k > 0:
    ms.apply()
    parts.append(ms.get_output_data(k))
ms.close()
st = np.concatenate(parts, 0)[:want.shape[0]].astype(np.float64)
wdt = want.shape[1] // groups


def dist(x, y, sl):
    scale = max(np.abs(want[:, sl]).max(), 1e-30)
    return np.abs(x[:, sl] - y[:, sl]).max() / scale, np.linalg.norm(x[:, sl] - y[:, sl]) / max(np.linalg.norm(want[:, sl]), 1e-30)


for g in range(groups):
    sl = slice(g * wdt, (g + 1) * wdt)
    print("group %d (scale %.3g):" % (g, np.abs(want[:, sl]).max()))
    for label, x, y in (("batch   vs checker", got, want), ("stream  vs checker", st, want), ("checker vs float64", want, truth),
                        ("batch   vs float64", got, truth), ("stream  vs float64", st, truth)):
        if x.shape != y.shape:
            print("   %s: shapes %s %s" % (label, x.shape, y.shape))
            continue
        e, l2n = dist(x, y, sl)
        print("   %s: max %.2e of scale, rel L2 %.2e" % (label, e, l2n))
