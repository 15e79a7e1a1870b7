"""Independent float64 numpy restatement of the *batch* semantics of the reference CPU path
(TEST INFRASTRUCTURE ONLY -- a second reading of the same reference files, used to cross-check
oracle/mfcc_oracle.c; it shares no code with it).

Whole-utterance formulas (what a multi-block streaming run of the reference produces):
  frames   x[t][j] = w[j] * pcm[t*S + j], zero padded to W2          segmentercpu.cpp:17-28
  spectrum |rfft(x)| / W2                                            mfcccpu.cpp:187-190,203
  mel      log(max(sum_j T[m%2][j] * v[j], 1e-30))                   mfcccpu.cpp:24-60,192-220
  cepstra  mel @ M                                                   mfcccpu.cpp:118-136,222-232
  delta    regression over replicate-padded statics; the second
           order is the regression of the first-order *buffer*       mfcccpu.cpp:234-263, deltacpu.cpp:16-29
"""
import numpy as np


def ewc(samples, W, S):
    return int(np.floor(np.float32(samples - (W - S)) / np.float32(S)))  # parambase.cpp:16-19


def mel_tables(nb, W2, sr, low, high, alpha=1.0):
    hz2mel = lambda f: 1127.0 * np.log(f / 700.0 + 1.0)
    mel2hz = lambda m: 700.0 * (np.exp(m / 1127.0) - 1.0)
    lo, hi = hz2mel(low), hz2mel(high)
    i = np.arange(nb + 2)
    f = mel2hz(i / float(nb + 1) * (hi - lo) + lo)
    o = 2 * np.pi * f / sr
    o = o + 2 * np.arctan(((1 - alpha) * np.sin(o)) / (1 - (1 - alpha) * np.cos(o)))
    centers = sr * o / (2 * np.pi)
    beg = np.floor(centers * W2 / sr + 0.5).astype(int)
    T = np.zeros((2, W2))
    for m in range(nb):
        cl, cc, cr = centers[m], centers[m + 1], centers[m + 2]
        for j in range(beg[m], beg[m + 2]):
            fj = j * sr / W2
            T[m % 2, j] = max(0.0, min((fj - cl) / (cc - cl), (fj - cr) / (cc - cr)))
    return T, beg


def dct_matrix(nb, nc, want_c0, lift):
    dl = nc + (1 if want_c0 else 0)
    M = np.zeros((nb, dl))
    k = np.arange(nb)
    nf = np.sqrt(2.0 / nb)
    for i in range(1, nc + 1):
        M[:, i - 1] = (1 + lift / 2 * np.sin(np.pi * i / lift)) * nf * np.cos(np.pi * i * (k + 0.5) / nb)
    if want_c0:
        M[:, nc] = nf
    return M


def regress(x_padded, L):
    """deltacpu.cpp:16-29 on an input that already carries L rows of context on both sides."""
    n = x_padded.shape[0] - 2 * L
    num = np.zeros((n, x_padded.shape[1]))
    for l in range(1, L + 1):
        num += l * (x_padded[L + l:L + l + n] - x_padded[L - l:L - l + n])
    return num / (2.0 * sum(l * l for l in range(1, L + 1)))


def mfcc_batch(pcm, window, W, S, nb, sr, low, high, nc, want_c0, lift, dyn, l1, l2, alpha=1.0):
    pcm = np.asarray(pcm, dtype=np.float64)
    T = ewc(pcm.size, W, S)
    W2 = 1 << int(np.ceil(np.log2(W)))
    idx = np.arange(T)[:, None] * S + np.arange(W)[None, :]
    x = np.zeros((T, W2))
    x[:, :W] = pcm[idx] * np.asarray(window, dtype=np.float64)[None, :]
    v = np.abs(np.fft.rfft(x, axis=1)) / W2
    Tm, beg = mel_tables(nb, W2, sr, low, high, alpha)
    E = np.empty((T, nb))
    for m in range(nb):
        E[:, m] = v[:, beg[m]:beg[m + 2]] @ Tm[m % 2, beg[m]:beg[m + 2]]
    mel = np.log(np.maximum(E, 1e-30))
    c = mel @ dct_matrix(nb, nc, want_c0, lift) if nc > 0 else mel
    if dyn == 0:
        return c
    if dyn == 1:
        l2 = 0
    D = l1 + l2
    cp = np.concatenate([np.repeat(c[:1], D, 0), c, np.repeat(c[-1:], D, 0)], 0)
    d_ext = regress(cp, l1)            # T + 2*l2 rows (mfcccpu.cpp:259)
    out = [c, d_ext[l2:l2 + T]]
    if dyn == 2:
        out.append(regress(d_ext, l2))  # over the delta BUFFER (mfcccpu.cpp:260-262)
    return np.concatenate(out, 1)
