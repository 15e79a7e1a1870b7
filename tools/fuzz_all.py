"""Dev aid (GPU box): random configurations over EVERY transform size and front-end kernel -- window, shift, filters, columns, c0,
dyn, normalisation, VTLN warp, mono / stereo, aligned / odd offsets, ragged utterances -- through the batch entry AND the streaming
interface (random block lengths), each against the CPU checker (oracle/, test infrastructure) at the north-star bar
(1e-4 of the column group's scale, 1e-5 relative L2; normalised outputs: 2e-3 of the group's scale, their exact bound is the
three-part check of tests/conftest.py).  Prints one line per case and a summary per kernel; exits non-zero on a failure.

    python tools/fuzz_all.py [seed] [cases]
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
rng = np.random.default_rng(seed)


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
    D = (l1 if dyn else 0) + (l2 if dyn == 2 else 0)
    frames = [max(int(x), 2 * D + 2) for x in rng.integers(2, 50, size=int(rng.integers(1, 6)))]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    odd_ok = ch == 1
    offs, pos = [], int(rng.integers(0, 3)) if odd_ok else 0
    for n in lens:
        offs.append(pos)
        pos += n + (int(rng.integers(0, 5)) if odd_ok else 2 * int(rng.integers(0, 3)))
    pcm = (4000.0 * rng.standard_normal(pos * ch)).round().clip(-32768, 32767).astype(np.int16)
    mono = pcm if ch == 1 else ((pcm[0::2].astype(np.int32) + pcm[1::2].astype(np.int32)) >> 1).astype(np.int16)
    window = pkg.reference_window(W)
    ibs = max(lens) + 4 * W
    what = "case %3d W2 %4d W %4d S %4d ch %d sr %5.0f nb %3d nc %2d c0 %d dyn %d l %d%d norm %d nad %d a %.2f utts %d" % (
        case, W2, W, S, ch, sr, nb, nc, c0, dyn, l1, l2, norm, nad, alpha, len(lens))
    try:
        m = pkg.MfccHip(ibs, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, norm, dyn, l1, l2, nad, fft_size=fft_size, channels=ch,
                        bug_compat=False)
    except pkg.MfxError as e:
        print(what + ": refused at create (%s)" % e)
        continue
    m.set_window(window)
    m.set_alpha(alpha)
    # the checker ties the transform to the window: a zero-padded window expresses fft_size
    Wo = fft_size or W
    wo = np.zeros(Wo, np.float32)
    wo[:W] = window
    cfg = orc.make_config(ibs + Wo, window_size=Wo, shift=S, num_banks=nb, sample_rate=sr, high_freq=sr / 2, ceps_len=nc,
                          want_c0=c0, norm=norm, dyn=dyn, delta_l1=l1, delta_l2=l2, norm_after_dyn=nad)
    groups = 1 + dyn
    tol = (1e-4, 1e-5) if norm == 0 else (2e-3, 2e-3)
    worst = (0.0, 0.0)
    worst_batch = worst_stream = 0.0
    ok = True
    try:
        rows, total = m.batch_plan(offs, lens)
        got = m.batch_run_host(pcm)
        name = m.dominant_kernel_name()
        for u, n in enumerate(lens):
            seg = mono[offs[u]:offs[u] + n]
            if fft_size:   # (the checker's frames are Wo long: give it the samples the zero taps meet)
                seg = np.concatenate([seg, np.zeros(Wo - W, np.int16)])
            want = orc.run_utterance(cfg, seg, wo, alpha=alpha, bug_compat=False)
            T = want.shape[0]
            g = got[rows[u]:rows[u] + T]
            if g.shape != want.shape or not np.isfinite(g).all():
                ok = False
                print(what + ": utt %d shape %s vs %s / non-finite" % (u, g.shape, want.shape))
                break
            e = rel_err(g, want, groups)
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
            worst_batch = max(worst_batch, e[0])
        # streaming interface (mono handles only, as the reference's): the longest utterance in random blocks
        if ok and ch == 1:
            u = int(np.argmax(lens))
            seg = mono[offs[u]:offs[u] + lens[u]]
            blk = int(rng.integers((2 * D + 2) * S + W, max(lens[u] + S, (2 * D + 3) * S + W)))   # (first block >= 2 D frames: DESIGN.md B13)
            ms = pkg.MfccHip(blk, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, norm, dyn, l1, l2, nad, fft_size=fft_size,
                             bug_compat=True)
            ms.set_window(window)
            try:
                gs = ms.process_stream(seg, alpha=alpha)
            finally:
                ms.close()
            segp = np.concatenate([seg, np.zeros(Wo - W, np.int16)]) if fft_size else seg
            cfgs = orc.make_config(blk, window_size=Wo, shift=S, num_banks=nb,
                                   sample_rate=sr, high_freq=sr / 2, ceps_len=nc, want_c0=c0, norm=norm, dyn=dyn, delta_l1=l1,
                                   delta_l2=l2, norm_after_dyn=nad)
            if not fft_size:   # (with a padded window the checker's block structure differs: batch comparison above covers it)
                ws = orc.run_utterance(cfgs, segp, wo, alpha=alpha, bug_compat=True)
                if gs.shape != ws.shape:
                    ok = False
                    print(what + ": streaming shape %s vs %s" % (gs.shape, ws.shape))
                elif gs.size:
                    e = rel_err(gs, ws, groups)
                    worst = (max(worst[0], e[0]), max(worst[1], e[1]))
                    worst_stream = e[0]
                    if e[0] > tol[0] and os.environ.get("FUZZ_DUMP"):
                        np.savez(os.path.join(os.environ["FUZZ_DUMP"], "case%d_seed%d.npz" % (case, seed)), seg=seg, gs=gs, ws=ws, blk=blk)
    except pkg.MfxError as e:
        print(what + ": refused later (%s)" % e)
        m.close()
        continue
    m.close()
    bad = (not ok) or worst[0] > tol[0] or worst[1] > tol[1]
    failures += bad
    k = by_kernel.setdefault(name, [0, 0.0])
    k[0] += 1
    if norm == 0:
        k[1] = max(k[1], worst[0])
    print("%s  %-12s max %.2e l2 %.2e (batch %.1e stream %.1e)%s" % (what, name, worst[0], worst[1], worst_batch, worst_stream,
                                                                    "   <-- FAIL" if bad else ""))
print("seed %d: %d cases, %d failures; per kernel (cases, worst un-normalised max-diff / scale): %s" % (
    seed, n_cases, failures, {k: (v[0], "%.1e" % v[1]) for k, v in sorted(by_kernel.items())}))
sys.exit(1 if failures else 0)
