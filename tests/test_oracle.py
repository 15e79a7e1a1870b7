"""CPU tests of the parity checker itself (oracle/): it must agree with
  (1) the REAL reference objects that build here (oracle/_ref: SegmenterCPU, DeltaCPU, NormalizerCPU,
      ParamBase, MfccBase) -- live when the .so is present, and through committed vectors always;
  (2) the DFT definition (numpy float64) where the reference calls FFTW;
  (3) an independent numpy reading of mfcccpu.cpp (oracle/np_restatement.py);
  (4) the observations of the compiled reference recorded in SURVEY.md (frame counts, block
      splits, value ranges, B1 rows);
  (5) its own committed C1 features (regression pin; "parity unpinned" w.r.t. the reference, whose
      mfcccpu.cpp needs libfftw3f and cannot be built in this image).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_close, synth_utterance

fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
sp = lambda a: a.ctypes.data_as(C.POINTER(C.c_short))


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLDEN, "ref_stage_vectors.npz"))


# ---------------------------------------------------------------------------------------------
# (1) against the reference's own objects
# ---------------------------------------------------------------------------------------------

def _oracle_segment_session(orc, W, S, window_limit, D, window, pcm, blocks):
    """Drive the oracle's segmenter (inside an extractor with matching capacity) like the golden session."""
    # capacity: window_limit = input_window_limit + 2 + 3D  ->  choose input_buffer_size accordingly
    iwl = window_limit - 2 - 3 * D
    ibs = iwl * S + W - S
    cfg = orc.make_config(ibs, window_size=W, shift=S, num_banks=26, ceps_len=13, dyn=orc.DYN_ACC, delta_l1=3,
                          delta_l2=D - 3)
    o = orc.OracleMfcc(cfg, window)
    assert o.window_limit == window_limit
    return o


def test_segmenter_matches_reference_vectors(orc, vec):
    W, S, window_limit, D = [int(v) for v in vec["seg_params"]]
    o = _oracle_segment_session(orc, W, S, window_limit, D, vec["seg_window"], vec["seg_pcm"], vec["seg_blocks"])
    pos = 0
    log = vec["seg_log"]
    for i, b in enumerate(vec["seg_blocks"]):
        n = o.set_input(vec["seg_pcm"][pos:pos + b])
        pos += int(b)
        rc, wc, wcnd, remaining, samples, is_fl, was_fl = [int(v) for v in log[i]]
        assert n == wc
        want = vec["seg_frames_%d" % i]
        got = o.tap("frames", want.shape[0])
        assert np.array_equal(got[:, :W], want[:, :W]), "block %d frames differ" % i
        assert np.all(got[:, W:] == 0)
    n = o.flush()
    assert n == int(log[-1][1])
    want = vec["seg_frames_%d" % (len(log) - 1)]
    assert np.array_equal(o.tap("frames", want.shape[0])[:, :W], want[:, :W])


def test_segmenter_short_first_block_error(orc, vec):
    assert int(vec["seg_short_rc"][0]) == -2
    o = orc.OracleMfcc(orc.make_config(8000, num_banks=26, ceps_len=13))
    with pytest.raises(RuntimeError, match="window count is too small"):
        o.set_input(np.zeros(1000, np.int16))


@pytest.mark.parametrize("L", [1, 2, 3])
def test_delta_matches_reference_vectors(orc, vec, L):
    x, want = vec["delta_in_%d" % L], vec["delta_out_%d" % L]
    got = np.zeros_like(want)
    orc.lib().orc_delta_apply(fp(np.ascontiguousarray(x)), x.shape[1], want.shape[0], L, fp(got))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nt", [1, 2, 3])
def test_normalizer_matches_reference_vectors(orc, vec, nt):
    x, want = vec["norm_in_%d" % nt].copy(), vec["norm_out_%d" % nt]
    x2, want2 = vec["norm_in2_%d" % nt].copy(), vec["norm_out2_%d" % nt]
    dim = x.shape[1]
    mean, var, mm = (np.zeros(dim, np.float32) for _ in range(3))
    L = orc.lib()
    L.orc_normalize(nt, fp(x), dim, x.shape[0], 0, fp(mean), fp(var), fp(mm))
    L.orc_normalize(nt, fp(x2), dim, x2.shape[0], 1, fp(mean), fp(var), fp(mm))
    if nt == 3:
        # under g++ the reference's unqualified abs() binds to the int overload, so the
        # g++-built reference object truncates |min-mean|, |max-mean| to integers.  The oracle
        # follows the reference's own toolchain (MSVC: float abs); the two agree once that
        # truncation is applied, which is what this checks.
        src = vec["norm_in_3"]
        m = src.astype(np.float64).sum(0) / src.shape[0]
        m32 = m.astype(np.float32)
        a = np.abs((src.min(0) - m32).astype(np.float32)).astype(np.int32)
        b = np.abs((src.max(0) - m32).astype(np.float32)).astype(np.int32)
        scale_ref = (np.float32(1.0) / np.maximum(a, b).astype(np.float32)).astype(np.float32)
        recon = ((src - m32) * scale_ref).astype(np.float32)
        np.testing.assert_allclose(recon, want, rtol=2e-7, atol=1e-7)
        # and the oracle is the same thing without the truncation
        a_f = np.abs((src.min(0) - m32).astype(np.float32))
        b_f = np.abs((src.max(0) - m32).astype(np.float32))
        np.testing.assert_allclose(x, (src - m32) * (np.float32(1.0) / np.maximum(a_f, b_f)), rtol=2e-7, atol=1e-7)
    else:
        assert np.array_equal(x, want)
        assert np.array_equal(x2, want2)
    # round 4: the same NormalizerCPU built with the float overload of its unqualified abs() (oracle/Makefile ref_f32, the
    # selection the reference's own toolchain makes): the checker's default arithmetic is bit-identical, MINMAX included
    assert np.array_equal(x, vec["norm_out_%d_f32" % nt])
    assert np.array_equal(x2, vec["norm_out2_%d_f32" % nt])


def test_base_arithmetic_matches_reference_vectors(orc, pkg, vec):
    L = orc.lib()
    for row in vec["base_rows"]:
        ibs, W, S, nb, nc, c0, dyn, want_ibs, want_width = [int(v) for v in row[:9]]
        ewcs = [int(v) for v in row[9:]]
        cfg = orc.make_config(ibs, window_size=W, shift=S, num_banks=nb, ceps_len=nc, want_c0=c0, dyn=dyn)
        o = orc.OracleMfcc(cfg, np.ones(W, np.float32))
        assert o.input_buffer_size == want_ibs
        assert o.width == want_width
        for s, e in zip((0, 239, 240, 399, 400, 559, 560, 16000, 114000, 9999999), ewcs):
            assert L.orc_ewc(s, W, S) == e
            # the product's integer frame count agrees wherever the float32 expression is exact
            assert pkg.host_frame_count(s, W, S) == e


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libref_stages.so")),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_objects_random(orc):
    """Same comparisons on fresh random inputs against the loaded reference objects."""
    R = orc.ref()
    L = orc.lib()
    rng = np.random.default_rng(99)
    for trial in range(5):
        dim, wc, Ld = int(rng.integers(1, 50)), int(rng.integers(1, 80)), int(rng.integers(1, 5))
        x = (rng.standard_normal((wc + 2 * Ld, dim)) * 20).astype(np.float32)
        d = R.ref_delta_new(dim, wc + 2, Ld)
        R.ref_delta_apply(d, fp(x), wc)
        want = np.ctypeslib.as_array(R.ref_delta_output(d), shape=(wc * dim,)).reshape(wc, dim).copy()
        R.ref_delta_free(d)
        got = np.zeros_like(want)
        L.orc_delta_apply(fp(x), dim, wc, Ld, fp(got))
        assert np.array_equal(got, want)
        for nt in (1, 2):
            y = (rng.standard_normal((wc + 5, dim)) * 3 - 1).astype(np.float32)
            a, b = y.copy(), y.copy()
            nz = R.ref_norm_new(nt, dim)
            R.ref_norm_normalize(nz, fp(a), wc + 5, 0)
            R.ref_norm_free(nz)
            mean, var, mm = (np.zeros(dim, np.float32) for _ in range(3))
            L.orc_normalize(nt, fp(b), dim, wc + 5, 0, fp(mean), fp(var), fp(mm))
            assert np.array_equal(a, b)
    # segmenter: random block sizes, both objects side by side
    W, S, D, wl = 400, 160, 6, 60
    window = orc.reference_window(W)
    seg = R.ref_seg_new(W, S, wl, D)
    R.ref_seg_set_window(seg, fp(window))
    iwl = wl - 2 - 3 * D
    o = orc.OracleMfcc(orc.make_config(iwl * S + W - S, num_banks=26, ceps_len=13), window)
    data = np.zeros((wl, 512), np.float32)
    for i in range(12):
        b = int(rng.integers(1300 if i == 0 else 1, o.input_buffer_size))
        pcm = rng.integers(-32768, 32767, size=b, dtype=np.int16)
        wc, wcnd = C.c_int(0), C.c_int(0)
        rc = R.ref_seg_set_input(seg, sp(pcm), fp(data), b, C.byref(wc), C.byref(wcnd))
        assert rc == 0
        n = o.set_input(pcm)
        assert n == wc.value
        if n > 0:
            assert np.array_equal(o.tap("frames", wcnd.value)[:, :W], data[:wcnd.value, :W])
    wc, wcnd = C.c_int(0), C.c_int(0)
    R.ref_seg_flush(seg, fp(data), C.byref(wc), C.byref(wcnd))
    assert o.flush() == max(wc.value, 0)
    R.ref_seg_free(seg)


# ---------------------------------------------------------------------------------------------
# (2) FFT stage against the DFT definition
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [64, 256, 512, 1024, 2048])
@pytest.mark.parametrize("mode", [0, 1])
def test_rfft_rows_vs_numpy(orc, n, mode):
    rng = np.random.default_rng(n + mode)
    x = rng.standard_normal((5, n)).astype(np.float32)
    x[1] = 0
    x[1, 3] = 1.0                       # impulse
    x[2] = np.cos(2 * np.pi * 5 * np.arange(n) / n)  # pure tone in bin 5
    out = np.zeros((5, 2 * n), np.float32)
    orc.lib().orc_rfft_rows(fp(x), fp(out), n, 5, mode)
    got = out[:, 0:n + 2:2] + 1j * out[:, 1:n + 2:2]
    want = np.fft.rfft(x.astype(np.float64), axis=1)
    tol = 2e-7 if mode == 0 else 3e-6
    assert np.abs(got - want).max() <= tol * np.abs(want).max() * np.sqrt(np.log2(n))
    assert abs(got[2, 5] - n / 2) < 1e-3 * n


# ---------------------------------------------------------------------------------------------
# (3) independent numpy reading of mfcccpu.cpp, (4) SURVEY-recorded observations, (5) regression pin
# ---------------------------------------------------------------------------------------------

def test_c1_against_numpy_restatement(orc, a0001):
    import np_restatement as NP
    got = orc.run_utterance(orc.make_config(32000, num_banks=26, ceps_len=13), a0001)
    want = NP.mfcc_batch(a0001, orc.reference_window(400), 400, 160, 26, 16000.0, 64.0, 8000.0, 13, False, 22.0, 2, 3, 3)
    assert_close(got, want, "C1 oracle vs numpy float64", groups=3)


def test_c1_survey_recorded_observations(orc, a0001):
    """SURVEY.md Appendix / 8a (probes of the compiled reference): a0001.wav has 114000 samples ->
    711 frames; sample_limit 32000 rounds to 31920 and yields blocks 192,199,200,114 + flush 6;
    sample_limit 1e7 yields 705 + flush 6; both agree except static rows 705-710 (B1, max abs
    12.3); static cepstra span -29.7 .. +34.6."""
    assert a0001.size == 114000
    o = orc.OracleMfcc(orc.make_config(32000, num_banks=26, ceps_len=13))
    assert o.input_buffer_size == 31920 and o.estimated_window_count(114000) == 711
    blocks, pos = [], 0
    while pos < a0001.size:
        n = o.set_input(a0001[pos:pos + 31920])
        o.apply()
        o.get_output_data(n)
        blocks.append(n)
        pos += 31920
    assert blocks == [192, 199, 200, 114] and o.flush() == 6
    o1 = orc.OracleMfcc(orc.make_config(10000000, num_banks=26, ceps_len=13))
    assert o1.set_input(a0001) == 705 and o1.flush() == 6
    multi = orc.run_utterance(orc.make_config(32000, num_banks=26, ceps_len=13), a0001)
    single = orc.run_utterance(orc.make_config(10000000, num_banks=26, ceps_len=13), a0001, bug_compat=True)
    fixed = orc.run_utterance(orc.make_config(10000000, num_banks=26, ceps_len=13), a0001, bug_compat=False)
    assert multi.shape == (711, 39)
    bad = np.unique(np.nonzero(np.abs(multi - single) > 1e-6)[0])
    assert list(bad) == [705, 706, 707, 708, 709, 710]
    assert np.all(np.abs(multi - single)[:, 13:] == 0)           # delta / acc unaffected by B1
    assert abs(np.abs(multi - single).max() - 12.3) < 0.1
    assert np.array_equal(fixed, multi)
    assert abs(multi[:, :13].min() - (-29.7)) < 0.05 and abs(multi[:, :13].max() - 34.6) < 0.05
    # B1 mechanism: rows 705..710 of the single-block run repeat rows 699..704
    assert np.array_equal(single[705:711, :13], multi[699:705, :13])


def test_a1_reference_main_defaults(orc, a1):
    """Reference main() defaults (ASR_OCL.cpp:560: 15 banks, 12 ceps + c0, CVN, no dyn) on a1.wav:
    SURVEY 8c records 504 x 13 with column mean 5e-8 and std 1.000 per block."""
    assert a1.size == 81000
    cfg = orc.make_config(10000000, num_banks=15, ceps_len=12, want_c0=True, norm=orc.NORM_CVN, dyn=orc.DYN_NONE)
    out = orc.run_utterance(cfg, a1)
    assert out.shape == (504, 13)
    assert np.abs(out.mean(0)).max() < 1e-6
    np.testing.assert_allclose(out.std(0, ddof=1), 1.0, atol=1e-5)


def test_c1_regression_pin(orc, a0001, a1):
    z = np.load(os.path.join(GOLDEN, "c1_a0001_oracle.npz"))
    multi = orc.run_utterance(orc.make_config(32000, num_banks=26, ceps_len=13), a0001)
    single = orc.run_utterance(orc.make_config(10000000, num_banks=26, ceps_len=13), a0001, bug_compat=True)
    np.testing.assert_allclose(multi, z["multi_block"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(single, z["single_block_bug"], rtol=0, atol=2e-5)
    dflt = orc.run_utterance(orc.make_config(32000, num_banks=15, ceps_len=12, want_c0=True, norm=orc.NORM_CVN,
                                             dyn=orc.DYN_NONE), a1)
    np.testing.assert_allclose(dflt, z["a1_main_defaults"], rtol=0, atol=2e-5)


def test_streaming_block_size_invariance(orc):
    """multi-block streaming reproduces the single-shot result (with >= 2 blocks)."""
    pcm = synth_utterance(48000, 3)
    cfg = lambda ibs: orc.make_config(ibs, num_banks=40, ceps_len=13)
    ref = orc.run_utterance(cfg(24000), pcm)
    for ibs in (8000, 12345, 30000):
        out = orc.run_utterance(cfg(ibs), pcm)
        assert np.array_equal(out, ref)
    # and the corrected single block equals them too
    assert np.array_equal(orc.run_utterance(cfg(100000), pcm, bug_compat=False), ref)


def test_zero_input_hits_log_floor(orc):
    out = orc.run_utterance(orc.make_config(8000, num_banks=40, ceps_len=0, dyn=orc.DYN_NONE), np.zeros(16000, np.int16))
    assert np.all(out == np.float32(np.log(np.float32(1e-30))))


def test_float_fft_mode_is_within_tolerance(orc):
    pcm = synth_utterance(32000, 5)
    a = orc.run_utterance(orc.make_config(16000, num_banks=40, ceps_len=13, fft_mode=0), pcm)
    b = orc.run_utterance(orc.make_config(16000, num_banks=40, ceps_len=13, fft_mode=1), pcm)
    assert_close(b, a, "oracle f32 FFT vs f64 FFT", groups=3)


def test_run_batch_equals_run_utterance(orc):
    pcm = np.stack([synth_utterance(16000, u) for u in range(4)])
    cfg = orc.make_config(20000, num_banks=40, ceps_len=13)
    got = orc.run_batch(cfg, pcm, n_threads=2)
    for u in range(4):
        assert np.array_equal(got[u], orc.run_utterance(cfg, pcm[u], bug_compat=False))


def test_first_block_shorter_than_two_delta_contexts_is_refused(orc):
    """a first block with D < frames < 2 D makes the reference copy its carry-over from before the start of its
    buffer (segmentercpu.cpp:72-73) unless its `processed_samples <= 0` guard (:70-71) happens to fire.  The checker refuses
    every such block with that guard's error; 2 D frames and more are accepted."""
    W, S = 62, 30
    w = orc.reference_window(W)
    pcm = np.zeros(4000, np.int16)
    for (l1, l2, blk_frames, ok) in ((3, 1, 7, False), (3, 1, 8, True), (3, 3, 11, False), (3, 3, 12, True), (3, 3, 6, False)):
        D = l1 + l2
        blk = blk_frames * S + W - S
        cfg = orc.make_config(blk, window_size=W, shift=S, num_banks=16, sample_rate=8000.0, ceps_len=4, want_c0=True,
                              dyn=orc.DYN_ACC, delta_l1=l1, delta_l2=l2)
        o = orc.OracleMfcc(cfg, w)
        if ok:
            assert o.set_input(pcm[:blk]) == blk_frames - D
        else:
            with pytest.raises(RuntimeError):
                o.set_input(pcm[:blk])
        o.close()
