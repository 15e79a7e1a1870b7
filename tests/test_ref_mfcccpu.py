"""The checker (oracle/mfcc_oracle.c) against the REAL reference `mfcccpu.cpp`, built two ways.

`make -C oracle ref` compiles /root/reference/mfcccpu.cpp in place and links it WITHOUT libfftw3f: the constructor,
destructor and fft() (the only code that references fftwf_*) are never referenced and --gc-sections drops them
(oracle/ref_mfcccpu_shim.cpp explains the mechanism; `nm` finds no fftwf symbol in the libraries).
What runs is the reference's own refresh_filters / filter / dct / do_delta / normalize / apply / get_output_data plus its
segmenter, delta and normaliser members; at the FFTW call site the spectrum is the double-precision DFT rounded to float.

The reference calls UNQUALIFIED log / exp / atan / sin / cos / sqrt on floats (mfcccpu.cpp:21-22,37,203,212) and an
unqualified abs (normalizercpu.cpp:66).  Which functions those names select is a property of the toolchain:

  * oracle/_ref/libref_mfcccpu_f32.so  (round 4) -- built with `-include math.h -include stdlib.h`, which puts the std::
    overload sets into the global namespace: the FLOAT overloads are selected, as the reference's own toolchain (MSVC,
    OpenCLProject3.vcxproj) selects them.  Same overload selection, not the same C runtime: logf / expf / atanf / sinf /
    cosf / sqrtf are glibc's.  Committed vectors: tests/golden/ref_mfcccpu_vectors_f32.npz.  The checker's DEFAULT binding
    -- the one every GPU test compares the HIP path with -- must be BIT-IDENTICAL to it: rows, frames per call, filter
    edges, filter weights, DCT matrix, on every case of tests/refcases.py, MINMAX included.
  * oracle/_ref/libref_mfcccpu.so -- plain g++: the C double functions and int abs(int).  Committed vectors:
    tests/golden/ref_mfcccpu_vectors.npz.  The checker under orc_set_libm_binding(1) must be BIT-IDENTICAL to it.

Cases (tests/refcases.py): C1 on a0001.wav in multi- and single-block mode, the reference main()'s defaults on a1.wav,
the C2 / C3 / C5 shapes at alpha 0.88 / 1 / 1.12, dyn 0 / 1 / 2 x CMN / CVN / MINMAX x norm-after-dyn, mel-only output, an
odd geometry, silence.  Always run against the committed vectors (generator tests/golden/make_golden.py); when the
libraries are present (this container, and the GPU box, where oracle/_ref travels) also live, including a randomised
sweep under both bindings.
"""
import os

import numpy as np
import pytest

import refcases as RC
from conftest import GOLDEN, assert_close

CASES = RC.cases()


BINDINGS = {"f32": dict(file="ref_mfcccpu_vectors_f32.npz", libm_double=False, lib="libref_mfcccpu_f32.so"),
            "gpp": dict(file="ref_mfcccpu_vectors.npz", libm_double=True, lib="libref_mfcccpu.so")}


@pytest.fixture(scope="module")
def refvecs():
    return {b: np.load(os.path.join(GOLDEN, v["file"])) for b, v in BINDINGS.items()}


@pytest.fixture(scope="module")
def refvec(refvecs):
    return refvecs["gpp"]


def _run_oracle(orc, case, libm_double):
    o = orc.OracleMfcc(RC.make_cfg(orc, case), RC.case_window(orc, case), libm_double=libm_double)
    rows, counts = RC.drive(o, RC.load_pcm(case["pcm"]), case["alpha"])
    t = o.tables()
    o.close()
    return rows, counts, t


@pytest.mark.parametrize("binding", sorted(BINDINGS))
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_is_bit_identical_to_the_reference_under_either_binding(orc, refvecs, name, binding):
    """binding f32: the checker's DEFAULT arithmetic against the reference built with the float overloads (MINMAX and
    all); binding gpp: orc_set_libm_binding(1) against the plain g++ build."""
    vec = refvecs[binding]
    rows, counts, t = _run_oracle(orc, CASES[name], BINDINGS[binding]["libm_double"])
    assert np.array_equal(counts, vec[name + "/counts"])
    assert np.array_equal(t["filter_beg"], vec[name + "/filter_beg"])
    assert np.array_equal(t["filters"], vec[name + "/filters"])
    if "dct_matrix" in t:
        assert np.array_equal(t["dct_matrix"], vec[name + "/dct_matrix"])
    want = vec[name + "/rows"]
    assert rows.shape == want.shape
    assert np.array_equal(rows, want, equal_nan=True), "max |diff| %.3g" % np.nanmax(np.abs(rows - want))


@pytest.mark.parametrize("name", sorted(CASES))
def test_what_the_binding_moves_between_the_two_reference_builds(refvecs, name):
    """The two committed vector sets against each other (no checker involved): frames per call and every filter edge are
    the same, weights within 2e-5 (the 2-bin-wide triangles at 128 mel / 2048 points), the DCT matrix identical (explicit
    sinf / cosf in the reference), un-normalised rows within 5e-6 of the output scale, CMN / CVN rows within 5e-5 (CVN
    divides by a standard deviation).  MINMAX rows differ by the int truncation of |min - mean|: reported by
    test_minmax_truncation_of_the_gpp_build, not bounded here."""
    c = CASES[name]
    f, g = refvecs["f32"], refvecs["gpp"]
    assert np.array_equal(f[name + "/counts"], g[name + "/counts"])
    assert np.array_equal(f[name + "/filter_beg"], g[name + "/filter_beg"])
    assert np.abs(f[name + "/filters"] - g[name + "/filters"]).max() <= 2e-5
    if name + "/dct_matrix" in f.files:
        assert np.array_equal(f[name + "/dct_matrix"], g[name + "/dct_matrix"])
    if c["cfg"]["norm"] == RC.NORM_MINMAX:
        return
    tol = 5e-6 if c["cfg"]["norm"] == RC.NORM_NONE else 5e-5
    assert_close(f[name + "/rows"], g[name + "/rows"], name, tol_max=tol, tol_l2=tol, groups=1 + c["cfg"]["dyn"])


def test_minmax_truncation_of_the_gpp_build(refvecs):
    """B4 made visible: on the MINMAX cases the float-bound build's rows lie in [-1, 1] with an extreme of exactly +-1 per
    normalised column and block (dyn off; with deltas the extreme may sit in a context row); the plain g++ build divides by int(|extreme - mean|) instead and overshoots."""
    for name, c in CASES.items():
        if c["cfg"]["norm"] != RC.NORM_MINMAX:
            continue
        f, g = refvecs["f32"][name + "/rows"], refvecs["gpp"][name + "/rows"]
        cols = f.shape[1] // (1 + c["cfg"]["dyn"])
        stat = f[:, :cols]                                  # the static group is normalised in every MINMAX case
        n0 = int(refvecs["f32"][name + "/counts"][0])       # first block: statistics of its own rows
        assert np.abs(stat[:n0]).max() <= 1.0 + 1e-6
        if c["cfg"]["dyn"] == 0:    # (with deltas the block's statistics also cover context rows that are not output)
            assert np.allclose(np.abs(stat[:n0]).max(axis=0), 1.0, atol=1e-6)
        assert np.abs(g[:n0, :cols]).max() > 1.0 + 1e-3, name


@pytest.mark.parametrize("tag", ["c2", "c3", "c5"])
def test_filter_and_dct_alone_on_synthetic_spectra(orc, refvecs, tag):
    """MfccCpu::filter + MfccCpu::dct on caller-made spectra: random, all-zero (the 1e-30 floor), 1e6 x and 1e-12 x --
    bit for bit under either binding."""
    W, nb, nc, sr = {"c2": (400, 40, 13, 16000.0), "c3": (1024, 80, 13, 16000.0), "c5": (1102, 128, 40, 44100.0)}[tag]
    for binding, info in BINDINGS.items():
        vec = refvecs[binding]
        spec, a = vec["stage_%s/spec" % tag], float(vec["stage_%s/alpha" % tag])
        cfg = orc.make_config(20 * W, window_size=W, shift=W // 2, num_banks=nb, sample_rate=sr, ceps_len=nc, dyn=orc.DYN_NONE)
        o = orc.OracleMfcc(cfg, libm_double=info["libm_double"])
        o.set_alpha(a)
        o.load_fft(spec)
        o.filter(spec.shape[0])
        o.dct(spec.shape[0])
        mel, mfcc = o.tap("mel", spec.shape[0]), o.tap("mfcc", spec.shape[0])
        assert np.array_equal(mel, vec["stage_%s/mel" % tag]), binding
        assert np.array_equal(mfcc, vec["stage_%s/mfcc" % tag]), binding
        assert np.all(mel[1] == np.float32(np.log(np.float32(1e-30))))
        o.close()


def test_binding_effect_on_c1_is_reported(orc, refvec):
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
_REFDIR = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref")
_LIVE = all(os.path.exists(os.path.join(_REFDIR, v["lib"])) for v in BINDINGS.values())
live = pytest.mark.skipif(not _LIVE, reason="oracle/_ref/libref_mfcccpu{,_f32}.so not built (needs /root/reference)")


@live
@pytest.mark.parametrize("binding", sorted(BINDINGS))
def test_reference_library_has_no_fftw_symbols(binding):
    import subprocess
    so = os.path.join(_REFDIR, BINDINGS[binding]["lib"])
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    assert "fftw" not in syms.lower()
    assert "refm_apply" in syms
    # the binding is what the name says: the float-bound build calls logf / expf / atanf, the plain build log / exp / atan
    und = set(l.split()[-1].split("@")[0] for l in syms.splitlines() if " U " in l)
    if binding == "f32":
        assert {"logf", "expf", "atanf", "sqrtf"} <= und and not ({"log", "exp", "atan"} & und)
    else:
        assert {"log", "exp", "atan"} <= und and not ({"logf", "expf", "atanf"} & und)
    # the dropped members are really gone, the pinned ones are really there (hidden visibility: look at all symbols)
    allsyms = subprocess.run(["nm", "-C", so], capture_output=True, text=True).stdout
    if allsyms.strip():   # not stripped
        assert "MfccCpu::filter(int)" in allsyms and "MfccCpu::apply()" in allsyms
        assert "MfccCpu::fft(int)" not in allsyms and "MfccCpu::MfccCpu(" not in allsyms


@live
@pytest.mark.parametrize("binding", sorted(BINDINGS))
@pytest.mark.parametrize("name", ["c1_multi", "c1_single", "c2_alpha088", "c5_alpha112", "dyn2_norm2_nad1", "dyn1_norm3_nad0"])
def test_live_reference_reproduces_the_committed_vectors(orc, refvecs, name, binding):
    c = CASES[name]
    m = orc.RefMfccCpu(RC.make_cfg(orc, c), RC.case_window(orc, c), f32=binding == "f32")
    rows, counts = RC.drive(m, RC.load_pcm(c["pcm"]), c["alpha"])
    assert np.array_equal(counts, refvecs[binding][name + "/counts"])
    assert np.array_equal(rows, refvecs[binding][name + "/rows"], equal_nan=True)
    m.close()


@live
@pytest.mark.parametrize("binding", sorted(BINDINGS))
def test_live_reference_randomised_sweep(orc, binding):
    """40 random configurations and block sizes per binding: the oracle stays bit-identical to the real MfccCpu built with
    the same overload selection (rows, counts, tables, normaliser statistics -- MINMAX included under f32, where abs is
    the float one); the two oracle bindings never disagree on a filter edge."""
    f32 = binding == "f32"
    rng = np.random.default_rng(31337 if not f32 else 27182)
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
        # the reference throws when a first block holds no more than D frames, reads before its carry buffer when it holds
        # fewer than 2 D, and overruns its buffers when dyn is off and the carry-over is long (B8): keep the
        # blocks inside what it supports
        blk = int(rng.integers((2 * D + 3) * S + W, max(n + S, (2 * D + 4) * S + W)))
        if dyn == 0 and W - S > 2 * S:
            blk = n + S
        case = dict(name="rand%d" % trial, pcm=("synth", n, 1000 + trial, sr), ibs=blk, alpha=alpha, window=None,
                    cfg=dict(window_size=W, shift=S, num_banks=nb, sample_rate=sr, low_freq=low, high_freq=high,
                             ceps_len=nc, want_c0=bool(rng.integers(0, 2)) and nc > 0, lift_coef=22.0, norm=norm, dyn=dyn,
                             delta_l1=l1, delta_l2=l2, norm_after_dyn=bool(rng.integers(0, 2))))
        cfg, w, pcm = RC.make_cfg(orc, case), RC.case_window(orc, case), RC.load_pcm(case["pcm"])
        m = orc.RefMfccCpu(cfg, w, f32=f32)
        o = orc.OracleMfcc(cfg, w, libm_double=not f32)
        o2 = orc.OracleMfcc(cfg, w, libm_double=f32)
        want, wc = RC.drive(m, pcm, alpha)
        got, gc = RC.drive(o, pcm, alpha)
        RC.drive(o2, pcm, alpha)
        what = "trial %d %s" % (trial, case["cfg"])
        assert np.array_equal(gc, wc), what
        assert np.array_equal(got, want, equal_nan=True), what
        tm, to, to2 = m.tables(), o.tables(), o2.tables()
        for k in tm:
            assert np.array_equal(tm[k], to[k]), what + " table " + k
        assert np.array_equal(tm["filter_beg"], to2["filter_beg"]), what + " (the other binding moved an edge)"
        if norm in (1, 2) or (norm == 3 and f32):
            assert np.array_equal(m.norm_stats(), o.norm_stats(), equal_nan=True), what
        for e in (m, o, o2):
            e.close()
