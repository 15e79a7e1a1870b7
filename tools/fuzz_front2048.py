"""Dev aid: random 2048-point configurations through k_front2048 and through k_front_reg
(mfx_config.engine = MFX_ENGINE_NO_FRONT2048), mono and stereo, ragged utterances; prints the worst relative differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
worst, routed = 0.0, 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for case in range(n_cases):
    ch = int(rng.integers(1, 3))
    W = int(rng.integers(600, 2049))
    S = int(rng.integers(max(8, W // 6), W + 1))
    even = ch == 1 and bool(rng.integers(0, 2))     # mono: half of the cases on aligned sample pairs, half at any alignment
    if even:
        W += W & 1
        S += S & 1
    nb = int(rng.choice([8, 15, 26, 40, 64, 80, 128, 200, 256, int(rng.integers(6, 257))]))
    c0 = bool(rng.integers(0, 2))
    nc = int(rng.integers(2, min(nb, 70) + 1 - (1 if c0 else 0)))
    if rng.integers(0, 4) == 0:
        nc = 0          # no DCT: log mel energies
        c0 = False
    sr = float(rng.choice([44100.0, 48000.0, 32000.0]))
    dyn = int(rng.integers(0, 3))
    alpha = float(rng.choice([1.0, 1.0, 0.88, 1.12]))
    frames = [int(x) for x in rng.integers(1, 70, size=int(rng.integers(1, 9)))]
    D = 0 if dyn == 0 else 2 if dyn == 1 else 4
    frames = [max(T, 2 * D + 1) for T in frames]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    if even:
        lens = [n + (n & 1) for n in lens]
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos += n + (2 * int(rng.integers(0, 3)) if (even or ch == 2) else int(rng.integers(0, 5)))
    pcm = (4000.0 * rng.standard_normal(pos * ch)).round().clip(-32768, 32767).astype(np.int16)
    outs = []
    try:
        for eng in (0, pkg.mfcc.ENGINE_NO_FRONT2048):
            m = pkg.MfccHip(max(lens) + 4000, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, 0, dyn, 2, 2, True, fft_size=2048,
                            channels=ch, engine=eng)
            m.set_window(pkg.reference_window(W))
            if alpha != 1.0:
                m.set_alpha(alpha)
            name = m.dominant_kernel_name()
            m.batch_plan(offs, lens)
            outs.append((name, m.batch_run_host(pcm)))
            m.close()
    except pkg.mfcc.MfxError as e:   # e.g. 8 filters over 1025 bins: refused (MFX_ERR_CONFIG), at create or when alpha widens them
        print("case %2d ch %d W %4d S %4d nb %3d nc %2d a %.2f: refused (%s)" % (case, ch, W, S, nb, nc, alpha, e))
        continue
    (n0, a), (n1, b) = outs
    routed += n0 == "k_front2048"
    assert a.shape == b.shape and np.isfinite(a).all()
    scale = max(1.0, float(np.abs(b).max()))
    d = float(np.abs(a - b).max()) / scale
    l2 = float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
    worst = max(worst, d)
    flag = "" if (d < 1e-4 and l2 < 1e-5) else "   <-- CHECK"
    print("case %2d ch %d W %4d S %4d nb %3d nc %2d c0 %d dyn %d a %.2f utts %d  %s vs %s: max %.2e l2 %.2e%s" % (
        case, ch, W, S, nb, nc, c0, dyn, alpha, len(lens), n0, n1, d, l2, flag))
print("routed to k_front2048: %d of %d; worst max-diff / scale %.2e" % (routed, n_cases, worst))
