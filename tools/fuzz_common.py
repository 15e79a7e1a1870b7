"""Shared by tools/fuzz_all.py and tools/check_shape.py: the batch formulas in float64 on the product's own float32 tables, and the
judgement of a result against the CPU checker with float32's own noise floor taken into account."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def truth64(pkg, seg, wo, S, nb, sr, nc, c0, dyn, l1, l2, alpha, low, high):
    """The batch formulas in float64 ON THE PRODUCT'S OWN float32 TABLES (mel weights, edges, DCT matrix): what the checker and
    the kernels would both give without rounding.  (oracle/np_restatement.py builds its tables in float64: an edge can move.)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import np_restatement as NP
    Wo = wo.size
    W2 = 1 << int(np.ceil(np.log2(Wo)))
    T = NP.ewc(seg.size, Wo, S)
    idx = np.arange(T)[:, None] * S + np.arange(Wo)[None, :]
    x = np.zeros((T, W2))
    x[:, :Wo] = seg.astype(np.float64)[idx] * wo.astype(np.float64)[None, :]
    v = np.abs(np.fft.rfft(x, axis=1)) / W2
    Tm, beg = pkg.host_mel_table(nb, W2, sr, low, high, alpha)
    E = np.empty((T, nb))
    for m_ in range(nb):
        E[:, m_] = v[:, beg[m_]:beg[m_ + 2]] @ Tm[m_ % 2, beg[m_]:beg[m_ + 2]].astype(np.float64)
    mel = np.log(np.maximum(E, 1e-30))
    c = mel @ pkg.host_dct_matrix(nb, nc, c0, 22.0).astype(np.float64) if nc > 0 else mel
    if dyn == 0:
        return c
    if dyn == 1:
        l2 = 0
    D_ = l1 + l2
    cp = np.concatenate([np.repeat(c[:1], D_, 0), c, np.repeat(c[-1:], D_, 0)], 0)
    d_ext = NP.regress(cp, l1)
    out = [c, d_ext[l2:l2 + T]]
    if dyn == 2:
        out.append(NP.regress(d_ext, l2))
    return np.concatenate(out, 1)


def judge(g, want, truth, groups):
    """(ok, max err / scale, rel L2): per column group, against max(the bar, 4 x the checker's own float32 noise)."""
    g, want = np.asarray(g, np.float64), np.asarray(want, np.float64)
    wdt = want.shape[1] // groups
    ok, emax, el2 = True, 0.0, 0.0
    for k in range(groups):
        sl = slice(k * wdt, (k + 1) * wdt)
        scale = max(np.abs(want[:, sl]).max(), 1e-30)
        nrm = max(np.linalg.norm(want[:, sl]), 1e-30)
        a = np.abs(g[:, sl] - want[:, sl]).max() / scale
        b = np.linalg.norm(g[:, sl] - want[:, sl]) / nrm
        fa = fb = 0.0
        if truth is not None:
            fa = np.abs(want[:, sl] - truth[:, sl]).max() / scale
            fb = np.linalg.norm(want[:, sl] - truth[:, sl]) / nrm
        ok = ok and a <= max(1e-4, 4 * fa) and b <= max(1e-5, 4 * fb)
        emax, el2 = max(emax, a), max(el2, b)
    return ok, emax, el2
