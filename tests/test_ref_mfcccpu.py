"""The checker (oracle/mfcc_oracle.c) against the REAL reference `mfcccpu.cpp`.

`make -C oracle ref` compiles /root/reference/mfcccpu.cpp in place and links it WITHOUT libfftw3f: the constructor,
destructor and fft() (the only code that references fftwf_*) are never referenced and --gc-sections drops them
(oracle/ref_mfcccpu_shim.cpp explains the mechanism; `nm` finds no fftwf symbol in oracle/_ref/libref_mfcccpu.so).
What runs is the reference's own refresh_filters / filter / dct / do_delta / normalize / apply / get_output_data plus its
segmenter, delta and normaliser members; at the FFTW call site the spectrum is the double-precision DFT rounded to float.

Two comparisons per case of tests/refcases.py (C1 on a0001.wav in multi- and single-block mode, the reference main()'s
defaults on a1.wav, the C2 / C3 / C5 shapes at alpha 0.88 / 1 / 1.12, dyn 0 / 1 / 2 x CMN / CVN / MINMAX x norm-after-dyn,
mel-only output, an odd geometry, silence):

  * libm binding "g++" (the reference's unqualified log/exp/atan/sin/cos/sqrt/abs on floats bound to the C double / int
    functions, which is how g++ compiles mfcccpu.cpp:21-22,37,203,212 and normalizercpu.cpp:66): the oracle must be
    BIT-IDENTICAL to the reference -- rows, frames per call, filter edges, filter weights, DCT matrix;
  * libm binding "MSVC" (float overloads; the reference's own toolchain and the checker's default, which the HIP path is
    compared with): frames per call and every filter edge identical, weights and rows within 5e-6 of the output scale
    (normalised outputs 5e-5: CVN divides by a standard deviation).  MINMAX is left out of this second comparison: the
    g++ build truncates |min - mean| to an integer, which is covered bit for bit by the first one.

Always run against the committed vectors (tests/golden/ref_mfcccpu_vectors.npz, generator tests/golden/make_golden.py);
when the library is present (this container, and the GPU box, where oracle/_ref travels) also live, including a
randomised sweep.
"""
import os

import numpy as np
import pytest

import refcases as RC
from conftest import GOLDEN, assert_close

CASES = RC.cases()


@pytest.fixture(scope="module")
def refvec():
    return np.load(os.path.join(GOLDEN, "ref_mfcccpu_vectors.npz"))


def _run_oracle(orc, case, libm_double):
    o = orc.OracleMfcc(RC.make_cfg(orc, case), RC.case_window(orc, case), libm_double=libm_double)
    rows, counts = RC.drive(o, RC.load_pcm(case["pcm"]), case["alpha"])
    t = o.tables()
    o.close()
    return rows, counts, t


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_gpp_binding_is_bit_identical_to_the_reference(orc, refvec, name):
    rows, counts, t = _run_oracle(orc, CASES[name], True)
    assert np.array_equal(counts, refvec[name + "/counts"])
    assert np.array_equal(t["filter_beg"], refvec[name + "/filter_beg"])
    assert np.array_equal(t["filters"], refvec[name + "/filters"])
    if "dct_matrix" in t:
        assert np.array_equal(t["dct_matrix"], refvec[name + "/dct_matrix"])
    want = refvec[name + "/rows"]
    assert rows.shape == want.shape
    assert np.array_equal(rows, want, equal_nan=True), "max |diff| %.3g" % np.nanmax(np.abs(rows - want))


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if c["cfg"]["norm"] != RC.NORM_MINMAX))
def test_oracle_msvc_binding_stays_within_float_noise_of_the_reference(orc, refvec, name):
    c = CASES[name]
    rows, counts, t = _run_oracle(orc, c, False)
    assert np.array_equal(counts, refvec[name + "/counts"])
    moved = np.nonzero(t["filter_beg"] != refvec[name + "/filter_beg"])[0]
    assert moved.size == 0, "filter edges moved by the libm binding: %s" % moved
    assert np.abs(t["filters"] - refvec[name + "/filters"]).max() <= 2e-5      # 2-bin-wide triangles at 128 mel / 2048 points
    if "dct_matrix" in t:
        assert np.array_equal(t["dct_matrix"], refvec[name + "/dct_matrix"])    # explicit sinf / cosf: no binding question
    want = refvec[name + "/rows"]
    groups = 1 + c["cfg"]["dyn"]
    tol = 5e-6 if c["cfg"]["norm"] == RC.NORM_NONE else 5e-5
    assert_close(rows, want, name, tol_max=tol, tol_l2=tol, groups=groups)


@pytest.mark.parametrize("tag", ["c2", "c3", "c5"])
def test_filter_and_dct_alone_on_synthetic_spectra(orc, refvec, tag):
    """MfccCpu::filter + MfccCpu::dct on caller-made spectra: random, all-zero (the 1e-30 floor), 1e6 x and 1e-12 x."""
    W, nb, nc, sr = {"c2": (400, 40, 13, 16000.0), "c3": (1024, 80, 13, 16000.0), "c5": (1102, 128, 40, 44100.0)}[tag]
    spec, a = refvec["stage_%s/spec" % tag], float(refvec["stage_%s/alpha" % tag])
    cfg = orc.make_config(20 * W, window_size=W, shift=W // 2, num_banks=nb, sample_rate=sr, ceps_len=nc, dyn=orc.DYN_NONE)
    for dbl in (True, False):
        o = orc.OracleMfcc(cfg, libm_double=dbl)
        o.set_alpha(a)
        o.load_fft(spec)
        o.filter(spec.shape[0])
        o.dct(spec.shape[0])
        mel, mfcc = o.tap("mel", spec.shape[0]), o.tap("mfcc", spec.shape[0])
        if dbl:
            assert np.array_equal(mel, refvec["stage_%s/mel" % tag])
            assert np.array_equal(mfcc, refvec["stage_%s/mfcc" % tag])
        else:
            assert_close(mel, refvec["stage_%s/mel" % tag], tag + " mel", tol_max=2e-6, tol_l2=2e-6)
            assert_close(mfcc, refvec["stage_%s/mfcc" % tag], tag + " mfcc", tol_max=5e-6, tol_l2=5e-6)
        assert np.all(mel[1] == np.float32(np.log(np.float32(1e-30))))
        o.close()


def test_msvc_binding_effect_is_reported(orc, refvec):
    """What the libm binding moves on C1 (a0001.wav): nothing discrete, < 1e-6 of the output scale."""
    rows_g, _, tg = _run_oracle(orc, CASES["c1_multi"], True)
    rows_m, _, tm = _run_oracle(orc, CASES["c1_multi"], False)
    assert np.array_equal(tg["filter_beg"], tm["filter_beg"])
    assert list(tg["filter_beg"][:6]) == [2, 4, 7, 10, 13, 16] and tg["filter_beg"][-1] == 256
    d = np.abs(rows_g - rows_m).max() / np.abs(rows_g).max()
    assert 0 < d < 1e-6


# ---------------------------------------------------------------------------------------------
# live against oracle/_ref/libref_mfcccpu.so
# ---------------------------------------------------------------------------------------------
_LIVE = os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libref_mfcccpu.so"))
live = pytest.mark.skipif(not _LIVE, reason="oracle/_ref/libref_mfcccpu.so not built (needs /root/reference)")


@live
def test_reference_library_has_no_fftw_symbols():
    import subprocess
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libref_mfcccpu.so")
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    assert "fftw" not in syms.lower()
    assert "refm_apply" in syms
    # the dropped members are really gone, the pinned ones are really there (hidden visibility: look at all symbols)
    allsyms = subprocess.run(["nm", "-C", so], capture_output=True, text=True).stdout
    if allsyms.strip():   # not stripped
        assert "MfccCpu::filter(int)" in allsyms and "MfccCpu::apply()" in allsyms
        assert "MfccCpu::fft(int)" not in allsyms and "MfccCpu::MfccCpu(" not in allsyms


@live
@pytest.mark.parametrize("name", ["c1_multi", "c1_single", "c2_alpha088", "c5_alpha112", "dyn2_norm2_nad1", "dyn1_norm3_nad0"])
def test_live_reference_reproduces_the_committed_vectors(orc, refvec, name):
    c = CASES[name]
    m = orc.RefMfccCpu(RC.make_cfg(orc, c), RC.case_window(orc, c))
    rows, counts = RC.drive(m, RC.load_pcm(c["pcm"]), c["alpha"])
    assert np.array_equal(counts, refvec[name + "/counts"])
    assert np.array_equal(rows, refvec[name + "/rows"], equal_nan=True)
    m.close()


@live
def test_live_reference_randomised_sweep(orc):
    """40 random configurations and block sizes: the oracle under the g++ binding stays bit-identical to the real MfccCpu
    (rows, counts, tables, normaliser statistics); under the MSVC binding the edges never move."""
    rng = np.random.default_rng(31337)
    for trial in range(40):
        W = int(rng.integers(64, 900))
        S = int(rng.integers(max(W // 5, 8), W))
        sr = float(rng.choice([8000.0, 16000.0, 22050.0, 44100.0]))
        nb = int(rng.integers(6, 60))
        nc = int(rng.integers(0, min(nb, 20)))
        dyn = int(rng.integers(0, 3))
        norm = int(rng.integers(0, 4))
        l1, l2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        low = float(rng.integers(0, 300))
        high = float(sr / 2 - rng.integers(0, 1000))
        alpha = float(rng.choice([1.0, 0.85, 0.93, 1.07, 1.15]))
        n = int(rng.integers(20, 60)) * S + W
        D = (l1 if dyn else 0) + (l2 if dyn == 2 else 0)
        # the reference throws when a first block holds no more than D frames, and overruns its buffers when dyn is off
        # and the carry-over is long: keep the blocks inside what it supports
        blk = int(rng.integers((D + 3) * S + W, n + S))
        if dyn == 0 and W - S > 2 * S:
            blk = n + S
        case = dict(name="rand%d" % trial, pcm=("synth", n, 1000 + trial, sr), ibs=blk, alpha=alpha, window=None,
                    cfg=dict(window_size=W, shift=S, num_banks=nb, sample_rate=sr, low_freq=low, high_freq=high,
                             ceps_len=nc, want_c0=bool(rng.integers(0, 2)) and nc > 0, lift_coef=22.0, norm=norm, dyn=dyn,
                             delta_l1=l1, delta_l2=l2, norm_after_dyn=bool(rng.integers(0, 2))))
        cfg, w, pcm = RC.make_cfg(orc, case), RC.case_window(orc, case), RC.load_pcm(case["pcm"])
        m = orc.RefMfccCpu(cfg, w)
        o = orc.OracleMfcc(cfg, w, libm_double=True)
        o2 = orc.OracleMfcc(cfg, w)
        want, wc = RC.drive(m, pcm, alpha)
        got, gc = RC.drive(o, pcm, alpha)
        RC.drive(o2, pcm, alpha)
        what = "trial %d %s" % (trial, case["cfg"])
        assert np.array_equal(gc, wc), what
        assert np.array_equal(got, want, equal_nan=True), what
        tm, to, to2 = m.tables(), o.tables(), o2.tables()
        for k in tm:
            assert np.array_equal(tm[k], to[k]), what + " table " + k
        assert np.array_equal(tm["filter_beg"], to2["filter_beg"]), what + " (MSVC binding moved an edge)"
        if norm in (1, 2):
            assert np.array_equal(m.norm_stats(), o.norm_stats(), equal_nan=True), what
        for e in (m, o, o2):
            e.close()
