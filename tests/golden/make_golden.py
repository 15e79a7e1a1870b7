#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/.  Run in the build container (needs
/root/reference for the reference stage objects in oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

Files
  a0001.wav, a1.wav        data files of the reference (soundfiles/), copied verbatim as inputs.
  ref_stage_vectors.npz    inputs and outputs of the REAL reference objects that build without
                           FFTW (SegmenterCPU, DeltaCPU, NormalizerCPU, ParamBase/MfccBase), driven
                           through oracle/_ref/libref_stages.so on seeded inputs.
  ref_mfcccpu_vectors.npz  outputs of the REAL MfccCpu member functions (refresh_filters, filter, dct, do_delta, normalize,
                           apply, get_output_data: /root/reference/mfcccpu.cpp compiled in place into
                           oracle/_ref/libref_mfcccpu.so, its FFTW-calling constructor / fft() dropped at link time; the
                           transform at that call site is the double-precision DFT rounded to float) for every case of
                           tests/refcases.py: rows, frames per call, mel tables, and filter()/dct() on synthetic spectra.
                           g++ binds the reference's unqualified libm calls to the double functions; the vectors are
                           therefore those of a g++ build of the reference (see oracle/mfcc_oracle.h, orc_set_libm_binding).
  ref_mfcccpu_vectors_f32.npz  the same cases from oracle/_ref/libref_mfcccpu_f32.so: the same reference sources compiled with
                           `-include math.h -include stdlib.h`, which makes the unqualified log/exp/atan/sin/cos/sqrt/abs
                           calls pick the FLOAT overloads -- the selection the reference's own toolchain (MSVC) makes.  The
                           checker's DEFAULT binding is bit-identical to these, MINMAX included; the HIP path is compared
                           with them.
  c1_a0001_oracle.npz      features of BASELINE config C1 (a0001.wav, 26 mel, 13 MFCC + d + dd)
                           from this repo's oracle (oracle/mfcc_oracle.c).  NOT reference output:
                           mfcccpu.cpp needs libfftw3f and cannot be built here, and the reference
                           ships no feature vectors.  It pins the oracle against regressions and gives
                           the GPU tests a committed target.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
sp = lambda a: a.ctypes.data_as(C.POINTER(C.c_short))


def ref_stage_vectors():
    R = O.ref()
    rng = np.random.default_rng(20260104)
    out = {}

    # ---- SegmenterCPU: a streaming session, blocks of uneven size, then flush
    W, S, D = 400, 160, 6
    window_limit = 40
    blocks = [3000, 1700, 900, 2500]
    pcm = rng.integers(-20000, 20000, size=sum(blocks), dtype=np.int16)
    window = O.reference_window(W)
    seg = R.ref_seg_new(W, S, window_limit, D)
    R.ref_seg_set_window(seg, fp(window))
    data = np.zeros((window_limit, 512), dtype=np.float32)
    pos, log, frames = 0, [], []
    for b in blocks:
        wc, wcnd = C.c_int(0), C.c_int(0)
        rc = R.ref_seg_set_input(seg, sp(pcm[pos:pos + b]), fp(data), b, C.byref(wc), C.byref(wcnd))
        pos += b
        log.append([rc, wc.value, wcnd.value, R.ref_seg_remaining(seg), R.ref_seg_samples(seg),
                    R.ref_seg_is_flushed(seg), R.ref_seg_was_flushed(seg)])
        frames.append(data[:max(wcnd.value, 0)].copy() if wc.value > 0 else np.zeros((0, 512), np.float32))
    wc, wcnd = C.c_int(0), C.c_int(0)
    R.ref_seg_flush(seg, fp(data), C.byref(wc), C.byref(wcnd))
    log.append([0, wc.value, wcnd.value, R.ref_seg_remaining(seg), R.ref_seg_samples(seg),
                R.ref_seg_is_flushed(seg), R.ref_seg_was_flushed(seg)])
    frames.append(data[:max(wcnd.value, 0)].copy())
    R.ref_seg_free(seg)
    out.update(seg_params=np.array([W, S, window_limit, D]), seg_blocks=np.array(blocks), seg_pcm=pcm,
               seg_window=window, seg_log=np.array(log, dtype=np.int64))
    for i, f in enumerate(frames):
        out["seg_frames_%d" % i] = f

    # ---- SegmenterCPU error path: first block too short
    seg = R.ref_seg_new(W, S, window_limit, D)
    R.ref_seg_set_window(seg, fp(window))
    wc, wcnd = C.c_int(0), C.c_int(0)
    out["seg_short_rc"] = np.array([R.ref_seg_set_input(seg, sp(pcm[:1000]), fp(data), 1000, C.byref(wc), C.byref(wcnd))])
    R.ref_seg_free(seg)

    # ---- DeltaCPU
    for L in (1, 2, 3):
        dim, wc_ = 13, 37
        x = rng.standard_normal((wc_ + 2 * L, dim)).astype(np.float32) * 10
        d = R.ref_delta_new(dim, wc_ + 4, L)
        R.ref_delta_apply(d, fp(x), wc_)
        y = np.ctypeslib.as_array(R.ref_delta_output(d), shape=(wc_ * dim,)).reshape(wc_, dim).copy()
        R.ref_delta_free(d)
        out["delta_in_%d" % L], out["delta_out_%d" % L] = x, y

    # ---- NormalizerCPU (CMN, CVN; MINMAX separately -- see test_oracle.py on B4)
    for nt in (1, 2, 3):
        dim, n = 13, 53
        x = (rng.standard_normal((n, dim)) * 7 + 3).astype(np.float32)
        x2 = (rng.standard_normal((11, dim)) * 7 + 3).astype(np.float32)
        nz = R.ref_norm_new(nt, dim)
        a = x.copy()
        R.ref_norm_normalize(nz, fp(a), n, 0)
        b = x2.copy()
        R.ref_norm_normalize(nz, fp(b), 11, 1)  # use_last_stats
        R.ref_norm_free(nz)
        out["norm_in_%d" % nt], out["norm_out_%d" % nt] = x, a
        out["norm_in2_%d" % nt], out["norm_out2_%d" % nt] = x2, b
        # the same NormalizerCPU built with the float overload binding of its unqualified abs() (normalizercpu.cpp:66;
        # oracle/Makefile ref_f32): MINMAX without the int truncation of the plain g++ build
        if O.ref_available(f32=True):
            RF = O.ref(f32=True)
            nz = RF.ref_norm_new(nt, dim)
            a, b = x.copy(), x2.copy()
            RF.ref_norm_normalize(nz, fp(a), n, 0)
            RF.ref_norm_normalize(nz, fp(b), 11, 1)
            RF.ref_norm_free(nz)
            out["norm_out_%d_f32" % nt], out["norm_out2_%d_f32" % nt] = a, b

    # ---- ParamBase / MfccBase arithmetic
    rows = []
    for (ibs, W_, S_) in ((32000, 400, 160), (10000000, 400, 160), (8000, 1102, 441), (5000, 256, 100), (16000, 512, 128)):
        for (nb, nc, c0, dyn) in ((26, 13, 0, 2), (40, 13, 1, 1), (15, 12, 1, 0), (80, 0, 0, 2)):
            p = R.ref_base_new(ibs, W_, S_, nb, 16000.0, 64.0, 8000.0, nc, c0, 22.0, 0, dyn, 3, 3, 1)
            rows.append([ibs, W_, S_, nb, nc, c0, dyn, R.ref_base_input_buffer_size(p), R.ref_base_output_width(p)] +
                        [R.ref_base_ewc(p, s) for s in (0, 239, 240, 399, 400, 559, 560, 16000, 114000, 9999999)])
            R.ref_base_free(p)
    out["base_rows"] = np.array(rows, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_stage_vectors.npz"), **out)
    print("wrote ref_stage_vectors.npz with", len(out), "arrays")


def c1_oracle():
    pcm, sr = O.read_wav_pcm16(os.path.join(HERE, "a0001.wav"))
    pcm = pcm[:, 0]
    multi = O.run_utterance(O.make_config(32000, num_banks=26, ceps_len=13), pcm)
    single = O.run_utterance(O.make_config(10000000, num_banks=26, ceps_len=13), pcm, bug_compat=True)
    pcm1, _ = O.read_wav_pcm16(os.path.join(HERE, "a1.wav"))
    # the reference main()'s own defaults (ASR_OCL.cpp:560): 15 banks, 12 ceps + c0, CVN, no dyn
    dflt = O.run_utterance(O.make_config(32000, num_banks=15, ceps_len=12, want_c0=True, norm=O.NORM_CVN,
                                         dyn=O.DYN_NONE), pcm1[:, 0])
    np.savez_compressed(os.path.join(HERE, "c1_a0001_oracle.npz"), multi_block=multi, single_block_bug=single,
                        a1_main_defaults=dflt)
    print("wrote c1_a0001_oracle.npz", multi.shape, single.shape, dflt.shape)


def ref_mfcccpu_vectors(f32=False):
    """f32 = False: the plain g++ build (unqualified libm names -> C double functions, abs -> int abs(int));
    f32 = True: oracle/_ref/libref_mfcccpu_f32.so, the same sources with the float overloads selected, as the
    reference's own toolchain selects them (oracle/Makefile ref_f32) -> ref_mfcccpu_vectors_f32.npz."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refcases as RC
    out = {}
    for name, c in RC.cases().items():
        pcm = RC.load_pcm(c["pcm"])
        m = O.RefMfccCpu(RC.make_cfg(O, c), RC.case_window(O, c), f32=f32)
        rows, counts = RC.drive(m, pcm, c["alpha"])
        t = m.tables()
        out[name + "/rows"], out[name + "/counts"] = rows, counts
        out[name + "/filter_beg"], out[name + "/filters"] = t["filter_beg"], t["filters"]
        if "dct_matrix" in t:
            out[name + "/dct_matrix"] = t["dct_matrix"]
        m.close()
    # filter() and dct() alone on caller-made spectra (incl. an all-zero row: the 1e-30 floor, and a huge one)
    rng = np.random.default_rng(20261004)
    for tag, W, nb, nc, sr, a in (("c2", 400, 40, 13, 16000.0, 1.0), ("c3", 1024, 80, 13, 16000.0, 0.9),
                                  ("c5", 1102, 128, 40, 44100.0, 1.1)):
        cfg = O.make_config(20 * W, window_size=W, shift=W // 2, num_banks=nb, sample_rate=sr, ceps_len=nc, dyn=O.DYN_NONE)
        m = O.RefMfccCpu(cfg, f32=f32)
        W2, rows = m.fft_size, 6
        spec = (rng.standard_normal((rows, W2 // 2 + 1)) + 1j * rng.standard_normal((rows, W2 // 2 + 1))).astype(np.complex64)
        spec *= np.float32(50.0)
        spec[1] = 0
        spec[2] *= np.float32(1e6)
        spec[3] *= np.float32(1e-12)
        m.set_alpha(a)
        m.load_fft(spec)
        m.filter(rows)
        m.dct(rows)
        out["stage_%s/spec" % tag], out["stage_%s/alpha" % tag] = spec, np.float32(a)
        out["stage_%s/mel" % tag], out["stage_%s/mfcc" % tag] = m.tap("mel", rows), m.tap("mfcc", rows)
        m.close()
    fn = "ref_mfcccpu_vectors_f32.npz" if f32 else "ref_mfcccpu_vectors.npz"
    np.savez_compressed(os.path.join(HERE, fn), **out)
    print("wrote", fn, "with", len(out), "arrays")


if __name__ == "__main__":
    if O.refm_available(f32=True):
        ref_mfcccpu_vectors(f32=True)
    else:
        print("oracle/_ref/libref_mfcccpu_f32.so missing: ref_mfcccpu_vectors_f32.npz not regenerated")
    if O.refm_available():
        ref_mfcccpu_vectors()
    else:
        print("oracle/_ref/libref_mfcccpu.so missing: ref_mfcccpu_vectors.npz not regenerated")
    if O.ref_available():
        ref_stage_vectors()
    else:
        print("oracle/_ref missing: ref_stage_vectors.npz not regenerated")
    c1_oracle()
