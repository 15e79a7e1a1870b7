"""Dev aid (GPU box): random CALL SEQUENCES on the streaming interface (the reference's ParamBase protocol, parambase.h:23-32,
ASR_OCL.cpp:227-301), legal steps mirrored on the CPU checker, illegal ones thrown in on the product only:

  legal    set_input(block) -> [set_alpha] -> apply [-> apply with another alpha] -> get_output_data(n) [again, or fewer rows]
           ... -> flush -> apply -> get_output_data -> (a new stream on the same handle)
  illegal  apply / get_output_data with no block, get_output_data before apply or for more rows than the block has, blocks longer
           than get_input_buffer_size(), empty blocks, flush twice, flush on a fresh handle, negative counts, NULL pointers

The product must answer every illegal step with a status code (or rows nobody specified) -- never crash, never hang -- and the
legal steps that follow must still deliver the checker's rows: state is not corrupted by misuse.

    python tools/fuzz_calls.py [seed] [handles]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
orc = G.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_handles = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
failures = 0
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
sp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int16))

for hcase in range(n_handles):
    W2 = int(rng.choice([256, 512, 512, 1024, 2048]))
    W = int(rng.integers(W2 // 2 + 1, W2 + 1))
    S = int(rng.integers(max(8, W // 5), W // 2 + 1))
    sr = float(rng.choice([8000.0, 16000.0, 44100.0]))
    nb, nc, c0 = int(rng.choice([15, 26, 40])), int(rng.integers(2, 14)), bool(rng.integers(0, 2))
    dyn = int(rng.integers(0, 3))
    norm = int(rng.choice([0, 0, 1]))
    l1, l2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    D = (l1 if dyn else 0) + (l2 if dyn == 2 else 0)
    groups = 1 + dyn
    blk = int(rng.integers((2 * D + 3) * S + W, (2 * D + 40) * S + W))
    window = pkg.reference_window(W)
    what = "handle %3d W %4d S %3d sr %5.0f nb %2d nc %2d c0 %d dyn %d l %d%d norm %d blk %d" % (hcase, W, S, sr, nb, nc, c0, dyn, l1, l2, norm, blk)
    m = pkg.MfccHip(blk, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, norm, dyn, l1, l2, True, bug_compat=True)
    L, h = m._L, m._h
    cfg = orc.make_config(blk, window_size=W, shift=S, num_banks=nb, sample_rate=sr, high_freq=sr / 2, ceps_len=nc, want_c0=c0,
                          norm=norm, dyn=dyn, delta_l1=l1, delta_l2=l2, norm_after_dyn=True)
    o = orc.OracleMfcc(cfg, window)
    width, ibs = m.get_output_data_width(), m.get_input_buffer_size()
    cap = m.max_frames_out() + 64
    out = np.zeros(cap * width, np.float32)
    nfr = C.c_int32()
    notes = []

    def misuse():
        """one illegal call on the product only; returns a description when the answer is not a status code"""
        k = int(rng.integers(0, 9))
        if k == 0:
            rc = L.mfx_apply(h)
        elif k == 1:
            rc = L.mfx_get_output_data(h, fp(out), int(rng.integers(1, cap)))
        elif k == 2:
            big = np.zeros(ibs + int(rng.integers(1, 500)), np.int16)
            rc = L.mfx_set_input(h, sp(big), big.size, C.byref(nfr))
            if rc == 0:
                return "a block longer than get_input_buffer_size() was accepted"
        elif k == 3:
            rc = L.mfx_set_input(h, sp(np.zeros(4, np.int16)), -1, C.byref(nfr))
            if rc == 0:
                return "a negative sample count was accepted"
        elif k == 4:
            rc = L.mfx_set_input(h, None, 100, C.byref(nfr))
            if rc == 0:
                return "a NULL block was accepted"
        elif k == 5:
            rc = L.mfx_get_output_data(h, None, 3)
            if rc == 0:
                return "a NULL output buffer was accepted"
        elif k == 6:
            rc = L.mfx_get_output_data(h, fp(out), -2)
            if rc == 0:
                return "a negative row count was accepted"
        elif k == 7:
            rc = L.mfx_get_output_data(h, fp(out), 1 << 28)
            if rc == 0:
                return "2^28 rows were accepted"
        else:
            rc = L.mfx_apply_alphas(h, None, 3)
            if rc == 0:
                return "a NULL alpha list was accepted"
        return None

    try:
        m.set_window(window)
        if rng.integers(0, 3) == 0:   # misuse on a fresh handle (flush before any block: the reference's flush() on an empty segmenter)
            L.mfx_flush(h, C.byref(nfr))
            o2 = orc.OracleMfcc(cfg, window)   # (the checker's flush on a fresh object is defined: 0 frames)
            o2.close()
            for _ in range(int(rng.integers(1, 4))):
                w_ = misuse()
                if w_:
                    notes.append(w_)
        for stream in range(int(rng.integers(1, 4))):     # several files on one handle
            n_total = int(rng.integers(1, 5)) * ibs + int(rng.integers(0, ibs))
            pcm = (4000.0 * rng.standard_normal(n_total)).round().clip(-32768, 32767).astype(np.int16)
            pos = 0
            scale = [0.0] * groups
            while True:
                last = pos >= pcm.size
                if rng.integers(0, 4) == 0 and not last:
                    w_ = misuse()     # between blocks: must not disturb the stream (a refused set_input leaves the state alone)
                    if w_:
                        notes.append(w_)
                if last:
                    a, b = m.flush(), o.flush()
                else:
                    piece = pcm[pos:pos + ibs]
                    a, b = m.set_input(piece), o.set_input(piece)
                    pos += ibs
                if a != b:
                    notes.append("frame counts %d vs %d" % (a, b))
                    break
                if a > 0:
                    alphas = [1.0] if rng.integers(0, 3) else [float(rng.choice([0.9, 1.1])), 1.0]
                    for al in alphas:      # apply may be repeated with another alpha on the same block
                        m.set_alpha(al)
                        o.set_alpha(al)
                        m.apply()
                        o.apply()
                    want = o.get_output_data(a)
                    for k in ([a] if rng.integers(0, 2) else [a, max(1, a // 2)]):   # read again, fewer rows
                        y = m.get_output_data(k)
                        ref = want[:k]
                        if norm == 0:
                            w = ref.shape[1] // groups
                            for g_ in range(groups):
                                x_, y_ = y[:, g_ * w:(g_ + 1) * w].astype(np.float64), ref[:, g_ * w:(g_ + 1) * w].astype(np.float64)
                                # (the bar is 1e-4 of the column group's scale over the STREAM so far: a flush block of D rows
                                # of delta-deltas can lie a hundred times below the stream's scale)
                                scale[g_] = max(scale[g_], np.abs(y_).max())
                                if np.abs(x_ - y_).max() > 1e-4 * max(scale[g_], 1e-30) or not np.isfinite(x_).all():
                                    notes.append("rows differ from the checker's after %s (group %d)" % ("flush" if last else "a block", g_))
                        elif y.shape != ref.shape:
                            notes.append("row shape")
                    if rng.integers(0, 4) == 0:
                        w_ = misuse()
                        if w_:
                            notes.append(w_)
                if last or notes:
                    break
            if notes:
                break
            if rng.integers(0, 2):     # a second flush: nothing left (the reference returns the same rows again)
                L.mfx_flush(h, C.byref(nfr))
                o.flush()
    except pkg.MfxError as e:
        notes.append("MfxError on a legal step: %s" % e)
    except RuntimeError as e:
        notes.append("checker refused a legal step: %s" % e)
    m.close()
    o.close()
    failures += bool(notes)
    print("%s: %s" % (what, "ok" if not notes else "FAIL -- " + "; ".join(sorted(set(notes)))), flush=True)
print("seed %d: %d handles, %d failures" % (seed, n_handles, failures))
sys.exit(1 if failures else 0)
