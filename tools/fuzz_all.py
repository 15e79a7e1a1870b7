"""Dev aid (GPU box): random configurations over EVERY transform size and front-end kernel -- window, shift, filters, columns, c0,
dyn, normalisation, VTLN warp, mono / stereo, aligned / odd offsets, ragged utterances -- through the batch entry AND the streaming
interface (random block lengths), each against the CPU checker (oracle/, test infrastructure) at the north-star bar
(1e-4 of the column group's scale, 1e-5 relative L2) -- or, where float32 itself is noisier than that (exotic shapes: 128 filters
on 257 bins put a c0 of 500 beside deltas of 5), at 4 x the checker's own distance from the same arithmetic in float64 on the
same tables.  Normalised configurations are run twice: with the normaliser off (the bar above) and on (same shapes; the largest difference
and the number of positions finite on one side only are printed, not judged -- a one- or two-row block's 1 / sigma is 0 / 0 or
amplifies float32 noise without bound; the exact criterion is the three-part check of tests/conftest.py).
Prints one line per case and a summary per kernel; exits non-zero on a failure.

    python tools/fuzz_all.py [seed] [cases] [wide | big | k1024]     (k1024: every case inside k_front1024's envelope)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
orc = G.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"   # wider ranges: band edges, warp, delta orders, filter / column counts, tiny shifts
K1024 = len(sys.argv) > 3 and sys.argv[3] == "k1024"  # every case inside k_front1024's envelope (1024 points, mono, <= 80 filters, <= 16 columns or no DCT)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"     # many long utterances per plan (whole grids, the tail split, the 4-frame pieces): a sample of them is checked
rng = np.random.default_rng(seed)
LOW, HIGH_CUT = 64.0, 0.0


def rel_err(a, b, groups):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    w = b.shape[1] // groups
    emax = el2 = 0.0
    for g in range(groups):
        x, y = a[:, g * w:(g + 1) * w], b[:, g * w:(g + 1) * w]
        scale = max(np.abs(y).max(), 1e-30)
        emax = max(emax, np.abs(x - y).max() / scale)
        el2 = max(el2, np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30))
    return emax, el2


from fuzz_common import judge, truth64 as _truth64  # noqa: E402


def truth64(seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, alpha):
    return _truth64(pkg, seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, alpha, LOW, sr / 2 - HIGH_CUT)


by_kernel, failures = {}, 0
for case in range(n_cases):
    W2 = int(rng.choice([64, 128, 256, 512, 512, 512, 1024, 1024, 2048, 2048, 4096]))
    W = int(rng.integers(max(W2 // 2 + 1, 24), W2 + 1))
    fft_size = 0
    if rng.integers(0, 3) == 0 and W2 >= 512:       # a short window zero padded to the transform (BASELINE configs[2] style)
        W = int(rng.integers(W2 // 8, W2 // 2 + 1))
        fft_size = W2
    S = int(rng.integers(max(8, W // 6), W + 1))
    ch = 2 if (rng.integers(0, 4) == 0) else 1
    sr = float(rng.choice([8000.0, 16000.0, 22050.0, 44100.0, 48000.0]))
    nb = int(rng.choice([6, 15, 23, 26, 40, 64, 80, 128, int(rng.integers(2, 140))]))
    nb = min(nb, W2 // 4)
    c0 = bool(rng.integers(0, 2))
    nc = int(rng.integers(1, min(nb, 40) + 1 - (1 if c0 else 0))) if nb > 1 else 0
    if rng.integers(0, 5) == 0 or nb <= 1:
        nc, c0 = 0, False
    dyn = int(rng.integers(0, 3))
    norm = int(rng.choice([0, 0, 0, 1, 2, 3]))
    nad = bool(rng.integers(0, 2))
    l1, l2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    alpha = float(rng.choice([1.0, 1.0, 0.88, 0.95, 1.12]))
    if WIDE:
        nb = int(rng.integers(1, min(256, W2 // 4) + 1))
        nc = int(rng.integers(0, min(nb, 100) + 1 - (1 if c0 else 0))) if nb > 1 else 0
        if nc == 0:
            c0 = False
        l1, l2 = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        alpha = float(rng.uniform(0.8, 1.25))
        S = int(rng.integers(4, W + 1))
        LOW, HIGH_CUT = float(rng.integers(0, 300)), float(rng.integers(0, int(sr / 8)))
    if K1024:
        W2, ch = 1024, 1
        if rng.integers(0, 2):      # a short window zero padded to 1024 points (the 16-wave builds), or any window up to 1024
            W, fft_size = int(rng.integers(64, 513)), 1024
        else:
            W, fft_size = int(rng.integers(513, 1025)), 0
        S = int(rng.integers(max(8, W // 6), W + 1))
        if rng.integers(0, 2):
            S &= ~1                 # even shifts at even offsets: the aligned builds
        nb = int(rng.integers(1, 81))
        c0 = bool(rng.integers(0, 2))
        nc = 0 if (rng.integers(0, 4) == 0 or nb < 2) else int(rng.integers(1, min(nb, 16 - (1 if c0 else 0)) + 1))
        if nc == 0:
            c0 = False
        alpha = float(rng.uniform(0.8, 1.25))
    D = (l1 if dyn else 0) + (l2 if dyn == 2 else 0)
    frames = [max(int(x), 2 * D + 2) for x in rng.integers(2, 50, size=int(rng.integers(1, 6)))]
    if BIG:
        frames = [max(int(x), 2 * D + 2) for x in rng.integers(100, 1500, size=int(rng.integers(40, 300)))]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    odd_ok = ch == 1 and not (K1024 and S % 2 == 0 and rng.integers(0, 4) > 0)
    offs, pos = [], int(rng.integers(0, 3)) if odd_ok else 0
    for n in lens:
        offs.append(pos)
        pos += n + (int(rng.integers(0, 5)) if odd_ok else 2 * int(rng.integers(0, 3)))
        if K1024 and not odd_ok:
            pos += pos & 1          # every utterance at an even offset
    pcm = (4000.0 * rng.standard_normal(pos * ch)).round().clip(-32768, 32767).astype(np.int16)
    mono = pcm if ch == 1 else ((pcm[0::2].astype(np.int32) + pcm[1::2].astype(np.int32)) >> 1).astype(np.int16)
    window = pkg.reference_window(W)
    ibs = max(lens) + 4 * W
    what = "case %3d W2 %4d W %4d S %4d ch %d sr %5.0f nb %3d nc %2d c0 %d dyn %d l %d%d norm %d nad %d a %.2f utts %d" % (
        case, W2, W, S, ch, sr, nb, nc, c0, dyn, l1, l2, norm, nad, alpha, len(lens))
    Wo = fft_size or W          # the checker ties the transform to the window: a zero-padded window expresses fft_size
    wo = np.zeros(Wo, np.float32)
    wo[:W] = window
    groups = 1 + dyn
    pad = (lambda x: np.concatenate([x, np.zeros(Wo - W, np.int16)])) if fft_size else (lambda x: x)   # samples the zero taps meet
    mkcfg = lambda ibs_, nrm: orc.make_config(ibs_, window_size=Wo, shift=S, num_banks=nb, sample_rate=sr, low_freq=LOW, high_freq=sr / 2 - HIGH_CUT,
                                              ceps_len=nc, want_c0=c0, norm=nrm, dyn=dyn, delta_l1=l1, delta_l2=l2, norm_after_dyn=nad)
    worst = [0.0, 0.0]
    worst_batch = worst_stream = worst_norm = 0.0
    n_pattern = 0
    ok, name, refused = True, "?", None
    blk = None
    for nrm in ([0] if norm == 0 else [0, norm]):      # a normalised configuration also runs with the normaliser off
        try:
            m = pkg.MfccHip(ibs, W, S, nb, sr, LOW, sr / 2 - HIGH_CUT, nc, c0, 22.0, nrm, dyn, l1, l2, nad, fft_size=fft_size, channels=ch,
                            bug_compat=False)
        except pkg.MfxError as e:
            refused = "at create (%s)" % e
            break
        try:
            m.set_window(window)
            m.set_alpha(alpha)
            rows, total = m.batch_plan(offs, lens)
            got = m.batch_run_host(pcm)
            name = m.dominant_kernel_name()
            cfg = mkcfg(ibs + Wo, nrm)
            check = range(len(lens)) if not BIG else sorted(set([0, len(lens) - 1] + [int(x) for x in rng.integers(0, len(lens), size=6)]))
            for u in check:
                n = lens[u]
                seg = pad(mono[offs[u]:offs[u] + n])
                want = orc.run_utterance(cfg, seg, wo, alpha=alpha, bug_compat=False)
                g = got[rows[u]:rows[u] + want.shape[0]]
                if g.shape != want.shape:
                    ok = False
                    print(what + ": utt %d shape %s vs %s" % (u, g.shape, want.shape))
                    break
                if nrm == 0:
                    if not np.isfinite(g).all():
                        ok = False
                        print(what + ": utt %d non-finite" % u)
                        break
                    good, e0, e1 = judge(g, want, truth64(seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, alpha), groups)
                    ok = ok and good
                    worst = [max(worst[0], e0), max(worst[1], e1)]
                    worst_batch = max(worst_batch, e0)
                else:
                    # (reported, not judged: one- and two-row blocks and zero-variance columns make the reference's CVN 0 / 0 or
                    # sqrt(0 / eps) with eps the rounding of a float product -- NaN on one side, 0 on the other, by luck)
                    fin = np.isfinite(want) & np.isfinite(g)
                    n_pattern += int((np.isfinite(want) != np.isfinite(g)).sum())
                    if fin.any():
                        e0 = np.abs(np.where(fin, g - want, 0.0)).max() / max(np.abs(want[fin]).max(), 1e-30)
                        worst_norm = max(worst_norm, e0)   # (reported, not judged: a 2-row block's 1 / sigma amplifies float32 noise without bound)
            # streaming interface (mono handles only, as the reference's): the longest utterance in random blocks
            if ok and ch == 1:
                u = int(np.argmax(lens))
                seg = mono[offs[u]:offs[u] + lens[u]]
                if blk is None:   # (first block >= 2 D frames: DESIGN.md B13)
                    blk = int(rng.integers((2 * D + 2) * S + W, max(lens[u] + S, (2 * D + 3) * S + W)))
                ms = pkg.MfccHip(blk, W, S, nb, sr, LOW, sr / 2 - HIGH_CUT, nc, c0, 22.0, nrm, dyn, l1, l2, nad, fft_size=fft_size,
                                 bug_compat=True)
                ms.set_window(window)
                try:
                    gs = ms.process_stream(seg, alpha=alpha)
                finally:
                    ms.close()
                if not fft_size:   # (with a padded window the checker's block structure differs: the batch comparison covers it)
                    ws = orc.run_utterance(mkcfg(blk, nrm), seg, wo, alpha=alpha, bug_compat=True)
                    if gs.shape != ws.shape:
                        ok = False
                        print(what + ": streaming shape %s vs %s" % (gs.shape, ws.shape))
                    elif gs.size and nrm == 0:
                        # (the checker's float32 noise from the batch formulas: same rows except B1's six, which are compared all the same)
                        good, e0, e1 = judge(gs, ws, None, groups)
                        tr = truth64(seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, alpha)
                        if not good and tr.shape == ws.shape:
                            wb = orc.run_utterance(mkcfg(ibs + Wo, 0), seg, wo, alpha=alpha, bug_compat=False)
                            fl = judge(wb, tr, None, groups)
                            good = e0 <= max(1e-4, 4 * fl[1]) and e1 <= max(1e-5, 4 * fl[2])
                        ok = ok and good
                        worst = [max(worst[0], e0), max(worst[1], e1)]
                        worst_stream = e0
                    elif gs.size:
                        fin = np.isfinite(ws) & np.isfinite(gs)
                        n_pattern += int((np.isfinite(ws) != np.isfinite(gs)).sum())
                        if fin.any():
                            e0 = np.abs(np.where(fin, gs - ws, 0.0)).max() / max(np.abs(ws[fin]).max(), 1e-30)
                            worst_norm = max(worst_norm, e0)
        except pkg.MfxError as e:
            refused = "later (%s)" % e
            m.close()
            break
        m.close()
    if refused:
        print(what + ": refused " + refused)
        continue
    failures += not ok
    k = by_kernel.setdefault(name, [0, 0.0])
    k[0] += 1
    k[1] = max(k[1], worst[0])
    print("%s  %-12s max %.2e l2 %.2e (batch %.1e stream %.1e%s)%s" % (what, name, worst[0], worst[1], worst_batch, worst_stream,
                                                                      ", normalised %.1e, %d non-finite mismatches" % (worst_norm, n_pattern) if norm else "",
                                                                      "" if ok else "   <-- FAIL"))
print("seed %d: %d cases, %d failures; per kernel (cases, worst un-normalised max-diff / scale): %s" % (
    seed, n_cases, failures, {k: (v[0], "%.1e" % v[1]) for k, v in sorted(by_kernel.items())}))
sys.exit(1 if failures else 0)
