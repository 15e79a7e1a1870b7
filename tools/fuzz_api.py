"""Dev aid (GPU box): random configurations against the library's OWN equivalence claims and the awkward corners of the C ABI
(tools/fuzz_all.py compares values with the CPU checker; this one compares the library with itself and with the checker where
the claim is "the same bits" or "the same rows"):

  1. engine bits that promise THE SAME BITS: MFX_ENGINE_NORM_TWO_KERNELS, MFX_ENGINE_DMA_SMALL_BLOCKS, MFX_ENGINE_FUSE_DELTA, MFX_ENGINE_FRONT1024_12_WAVES
     (where it applies), and MFX_ENGINE_STREAM_KERNELS on the batch entry == the streaming interface's rows;
  2. mfx_apply_alphas == rounds of mfx_set_alpha + mfx_apply (same bits), each against the checker;
  3. one handle, several files: set_input* -> flush -> set_input* ... == a fresh handle per file;
  4. ragged batches with utterances of 0 frames, of fewer than 2 D frames (whole-utterance formulas: the float64 restatement)
     and ordinary ones in one plan; empty plans; plans replaced on a live handle;
  5. the device entry: consecutive batches with mfx_batch_overlap on == off (same bits), == the host entry.

    python tools/fuzz_api.py [seed] [cases]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
orc = G.load_oracle()
import np_restatement as NP  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
E = pkg.mfcc
failures = 0


def stream(m, seg, alpha=1.0):
    return m.process_stream(seg, alpha=alpha)


# (against the checker: the north-star bar on the maximum; 5e-5 on the relative L2 -- delta groups of many-filter shapes, 64
# filters on 129 or 257 bins, sit at 1 - 2e-5 by float32 noise alone: tools/fuzz_all.py measures that floor case by case)
def close_enough(a, b, groups, tol=(1e-4, 5e-5)):
    if a.shape != b.shape:
        return False, "shape %s vs %s" % (a.shape, b.shape)
    if a.size == 0:
        return True, ""
    if not np.isfinite(a).all():
        return False, "non-finite"
    w = b.shape[1] // groups
    for g in range(groups):
        x, y = a[:, g * w:(g + 1) * w].astype(np.float64), b[:, g * w:(g + 1) * w].astype(np.float64)
        e = np.abs(x - y).max() / max(np.abs(y).max(), 1e-30)
        l2 = np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30)
        if e > tol[0] or l2 > tol[1]:
            return False, "group %d max %.2e l2 %.2e" % (g, e, l2)
    return True, ""


for case in range(n_cases):
    W2 = int(rng.choice([256, 512, 512, 512, 1024, 2048]))
    W = int(rng.integers(W2 // 2 + 1, W2 + 1))
    S = int(rng.integers(max(8, W // 5), W // 2 + 1))
    sr = float(rng.choice([8000.0, 16000.0, 44100.0]))
    nb = min(int(rng.choice([15, 23, 26, 40, 64])), W2 // 8)   # (denser banks put float32's own noise above the bar on the delta groups: those
                                                                # shapes are judged against float64 by tools/fuzz_all.py, not here)
    c0 = bool(rng.integers(0, 2))
    nc = int(rng.integers(2, 14))
    dyn = int(rng.integers(0, 3))
    norm = int(rng.choice([0, 0, 1, 2, 3]))
    nad = bool(rng.integers(0, 2))
    l1, l2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    D = (l1 if dyn else 0) + (l2 if dyn == 2 else 0)
    groups = 1 + dyn
    window = pkg.reference_window(W)
    what = "case %3d W2 %4d W %4d S %3d sr %5.0f nb %2d nc %2d c0 %d dyn %d l %d%d norm %d nad %d" % (
        case, W2, W, S, sr, nb, nc, c0, dyn, l1, l2, norm, nad)
    mk = lambda ibs, engine=0, bug_compat=True, nrm=norm: pkg.MfccHip(ibs, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, nrm, dyn, l1, l2,
                                                                    nad, engine=engine, bug_compat=bug_compat)
    mkcfg = lambda ibs, nrm=norm: orc.make_config(ibs, window_size=W, shift=S, num_banks=nb, sample_rate=sr, high_freq=sr / 2,
                                                  ceps_len=nc, want_c0=c0, norm=nrm, dyn=dyn, delta_l1=l1, delta_l2=l2,
                                                  norm_after_dyn=nad)
    notes = []
    try:
        # ---- files of this case
        n_files = int(rng.integers(2, 5))
        files = []
        for _ in range(n_files):
            T = int(rng.integers(2 * D + 3, 60))
            n = (T - 1) * S + W + int(rng.integers(0, S))
            files.append((4000.0 * rng.standard_normal(n)).round().clip(-32768, 32767).astype(np.int16))
        blk = int(rng.integers((2 * D + 2) * S + W, max(len(f) for f in files) + S))

        # 1a. streaming: engine bits that promise the same bits
        ref_rows = []
        m0 = mk(blk)
        m0.set_window(window)
        for f in files:                                    # 3. one handle, several files
            ref_rows.append(stream(m0, f))
        m0.close()
        for f, r in zip(files, ref_rows):                  # ... == a fresh handle per file
            mf = mk(blk)
            mf.set_window(window)
            y = stream(mf, f)
            mf.close()
            if not (y.shape == r.shape and np.array_equal(y, r, equal_nan=True)):
                notes.append("handle reuse after flush != fresh handle")
        for bit, label in ((E.ENGINE_NORM_TWO_KERNELS, "NORM_TWO_KERNELS"), (E.ENGINE_DMA_SMALL_BLOCKS, "DMA_SMALL_BLOCKS")):
            mb = mk(blk, engine=bit)
            mb.set_window(window)
            for f, r in zip(files, ref_rows):
                y = stream(mb, f)
                if not (y.shape == r.shape and np.array_equal(y, r, equal_nan=True)):
                    notes.append("streaming, engine %s: bits differ" % label)
                    break
            mb.close()
        # streaming rows against the checker (un-normalised twin when normalised: the exact criterion lives in the tests)
        if norm == 0:
            for f, r in zip(files, ref_rows):
                okc, why = close_enough(r, orc.run_utterance(mkcfg(blk), f, window, bug_compat=True), groups)
                if not okc:
                    notes.append("streaming vs checker: " + why)
                    break

        # 1b. the batch entry on the streaming interface's kernels delivers the streaming interface's bits (what afet_hip's
        # whole-file batches rely on): files that fit ONE block, B1 applied by the caller as afet_hip does (bug_compat off here
        # on both sides: the flush rows at their correct place)
        big = max(len(f) for f in files) + 4 * W
        ms = mk(big, engine=E.ENGINE_STREAM_KERNELS, bug_compat=False)
        ms.set_window(window)
        offs, pos = [], 0
        for f in files:
            offs.append(pos)
            pos += len(f) + (len(f) & 1)
        pcm = np.zeros(pos, np.int16)
        for o_, f in zip(offs, files):
            pcm[o_:o_ + len(f)] = f
        rows, total = ms.batch_plan(offs, [len(f) for f in files])
        got = ms.batch_run_host(pcm)
        ms.close()
        one = mk(big, bug_compat=False)
        one.set_window(window)
        for u, f in enumerate(files):
            y = stream(one, f)
            g = got[rows[u]:rows[u] + y.shape[0]]
            if not (g.shape == y.shape and np.array_equal(g, y, equal_nan=True)):
                notes.append("batch on STREAM_KERNELS != streaming bits (file %d)" % u)
                break
        one.close()

        # 1c. the fused kernels' batch entry: NORM_TWO_KERNELS and FUSE_DELTA the same bits; values against the checker
        outs = {}
        for bit, label in ((0, "default"), (E.ENGINE_NORM_TWO_KERNELS, "NORM_TWO_KERNELS"), (E.ENGINE_FUSE_DELTA, "FUSE_DELTA"),
                           (E.ENGINE_FRONT1024_12_WAVES, "FRONT1024_12_WAVES")):
            mb = mk(big, engine=bit, bug_compat=False)
            mb.set_window(window)
            mb.batch_plan(offs, [len(f) for f in files])
            outs[label] = mb.batch_run_host(pcm)
            if bit == 0:   # 4. the plan replaced on a live handle: a shorter plan, then the first one again
                mb.batch_plan(offs[:1], [len(files[0])])
                first = mb.batch_run_host(pcm)
                if not np.array_equal(first, outs[label][:first.shape[0]], equal_nan=True):
                    notes.append("a replaced plan changed the first utterance's rows")
                mb.batch_plan(offs, [len(f) for f in files])
                if not np.array_equal(mb.batch_run_host(pcm), outs[label], equal_nan=True):
                    notes.append("the first plan again: different rows")
            mb.close()
        for label in ("NORM_TWO_KERNELS", "FUSE_DELTA", "FRONT1024_12_WAVES"):
            if not np.array_equal(outs[label], outs["default"], equal_nan=True):
                notes.append("batch, engine %s: bits differ" % label)
        if norm == 0:
            for u, f in enumerate(files):
                want = orc.run_utterance(mkcfg(big), f, window, bug_compat=False)
                okc, why = close_enough(outs["default"][rows[u]:rows[u] + want.shape[0]], want, groups)
                if not okc:
                    notes.append("batch vs checker (file %d): %s" % (u, why))
                    break

        # 1d. device entry: consecutive batches with the delta tail of batch i overlapping the front end of batch i + 1
        # (mfx_batch_overlap: second stream, double-buffered statics) -- every batch the bits of the plain run, different PCM
        # per batch (a race between a tail and the next front end would mix them)
        import torch
        dev = torch.device("cuda", 0)
        mo = mk(big, bug_compat=False)
        mo.set_window(window)
        rows_o, total_o = mo.batch_plan(offs, [len(f) for f in files])
        pcms = [torch.from_numpy(np.roll(pcm, 17 * i).copy()).to(dev) for i in range(4)]
        outs_plain, outs_ovl = [], []
        for mode, sink in ((False, outs_plain), (True, outs_ovl)):
            mo.batch_overlap(mode)
            bufs = [torch.zeros((max(total_o, 1), mo.get_output_data_width()), dtype=torch.float32, device=dev) for _ in range(4)]
            for i in range(4):
                mo.batch_run_device(pcms[i].data_ptr(), pcm.size, bufs[i].data_ptr())
            mo.synchronize()
            sink.extend(b.cpu().numpy() for b in bufs)
        mo.batch_overlap(False)
        mo.close()
        for i in range(4):
            if not np.array_equal(outs_plain[i], outs_ovl[i], equal_nan=True):
                notes.append("overlapped batches: batch %d differs from the plain run" % i)
                break
        if not np.array_equal(outs_plain[0], outs["default"], equal_nan=True):
            notes.append("device entry != host entry")

        # 2. alpha sweep in one call == rounds of set_alpha + apply (same bits), against the checker
        alphas = [float(a) for a in rng.choice([0.85, 0.9, 0.95, 1.0, 1.05, 1.1, 1.15], size=int(rng.integers(1, 5)), replace=False)]
        ma = mk(big, bug_compat=True)
        ma.set_window(window)
        n = ma.set_input(files[0][:ma.get_input_buffer_size()])
        ma.apply_alphas(alphas)
        sweep = [ma.get_output_data_alpha(i, n) for i in range(len(alphas))]
        for i, a in enumerate(alphas):
            ma.set_alpha(a)
            ma.apply()
            y = ma.get_output_data(n)
            if not np.array_equal(y, sweep[i], equal_nan=True):
                notes.append("apply_alphas != set_alpha + apply (alpha %.2f)" % a)
                break
            if norm == 0:
                oc = orc.OracleMfcc(mkcfg(big), window)
                oc.set_input(files[0][:ma.get_input_buffer_size()])
                oc.set_alpha(a)
                oc.apply()
                okc, why = close_enough(y, oc.get_output_data(n), groups)
                oc.close()
                if not okc:
                    notes.append("sweep vs checker (alpha %.2f): %s" % (a, why))
                    break
        ma.close()

        # 4. ragged plan: utterances of 0 frames, of 1 .. 2 D frames, and ordinary ones; then an empty plan
        lens = [int(rng.integers(0, W)), W, W + S * int(rng.integers(0, max(2 * D, 1))), len(files[0]), int(rng.integers(0, 5))]
        lens = [n + (n & 1) for n in lens]
        offs2, pos = [], 0
        for n in lens:
            offs2.append(pos)
            pos += n
        pcm2 = (4000.0 * rng.standard_normal(max(pos, 2))).round().clip(-32768, 32767).astype(np.int16)
        mr = mk(big, bug_compat=False, nrm=0)
        mr.set_window(window)
        rows2, total2 = mr.batch_plan(offs2, lens)
        got2 = mr.batch_run_host(pcm2)
        exp_frames = [max(0, pkg.host_frame_count(n, W, S)) for n in lens]
        if total2 != sum(exp_frames) or got2.shape[0] != total2:
            notes.append("ragged plan: %d rows, expected %d" % (total2, sum(exp_frames)))
        else:
            for u, n in enumerate(lens):
                T = exp_frames[u]
                if T == 0:
                    continue
                want = NP.mfcc_batch(pcm2[offs2[u]:offs2[u] + n], window, W, S, nb, sr, 64.0, sr / 2, nc, c0, 22.0, dyn, l1, l2)
                okc, why = close_enough(got2[rows2[u]:rows2[u] + T], want, groups, tol=(2e-4, 2e-4))   # (float64 tables: looser)
                if not okc:
                    notes.append("ragged plan, utterance of %d frames vs the float64 restatement: %s" % (T, why))
                    break
        r0, t0 = mr.batch_plan([], [])
        if t0 != 0 or mr.batch_run_host(pcm2).shape[0] != 0:
            notes.append("empty plan delivered rows")
        mr.close()
    except pkg.MfxError as e:
        notes.append("MfxError: %s" % e)
    failures += bool(notes)
    print("%s: %s" % (what, "ok" if not notes else "FAIL -- " + "; ".join(notes)), flush=True)
print("seed %d: %d cases, %d failures" % (seed, n_cases, failures))
sys.exit(1 if failures else 0)
