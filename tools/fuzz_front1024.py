"""Dev aid: random 1024-point short-window configurations through k_front1024 and through k_front_reg
(mfx_config.engine = MFX_ENGINE_NO_FRONT1024), ragged utterances at random offsets; prints the worst relative differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
worst = 0.0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
routed = 0
for case in range(n_cases):
    W = int(rng.integers(40, 1025))
    S = int(rng.integers(max(8, W // 6), W + 1))
    if W > 512:
        S += S & 1   # long windows run on k_front1024 for aligned frames only
    nb = int(rng.integers(6, 81))
    c0 = bool(rng.integers(0, 2))
    nc = int(rng.integers(2, min(nb, 15 if c0 else 16) + 1))
    sr = float(rng.choice([16000.0, 22050.0, 44100.0, 8000.0]))
    dyn = int(rng.integers(0, 3))
    alpha = float(rng.choice([1.0, 1.0, 0.88, 1.12]))
    frames = [int(x) for x in rng.integers(1, 90, size=int(rng.integers(1, 9)))]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    even = W > 512
    if even:
        lens = [n + (n & 1) for n in lens]
    offs, pos = [], 0 if even else int(rng.integers(0, 3))
    for n in lens:
        offs.append(pos)
        pos += n + (2 * int(rng.integers(0, 3)) if even else int(rng.integers(0, 5)))
    pcm = (4000.0 * rng.standard_normal(pos)).round().clip(-32768, 32767).astype(np.int16)
    outs = []
    for no in ("0", "1"):
        m = pkg.MfccHip(max(lens) + 2000, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, 0, dyn, 2, 2, True, fft_size=1024, engine=int(no))
        m.set_window(pkg.reference_window(W))
        if alpha != 1.0:
            m.set_alpha(alpha)
        name = m.dominant_kernel_name()
        m.batch_plan(offs, lens)
        outs.append((name, m.batch_run_host(pcm)))
        m.close()
    (n0, a), (n1, b) = outs
    routed += n0 == "k_front1024"
    assert a.shape == b.shape
    scale = max(1.0, float(np.abs(b).max()))
    d = float(np.abs(a - b).max()) / scale
    l2 = float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
    worst = max(worst, d)
    flag = "" if (d < 1e-4 and l2 < 1e-5) else "   <-- CHECK"
    print("case %2d W %3d S %3d nb %2d nc %2d c0 %d dyn %d a %.2f utts %d  %s vs %s: max %.2e l2 %.2e%s" % (
        case, W, S, nb, nc, c0, dyn, alpha, len(lens), n0, n1, d, l2, flag))
print("routed to k_front1024: %d of %d; worst max-diff / scale %.2e" % (routed, n_cases, worst))
