"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of
include/mfx.h (libmfcchip.so via ctypes) and is compared with the CPU oracle on the same inputs.

Tolerance (north_star: MFCC + d + dd within 1e-4 relative): per column group,
    max |a-b| / max |b| <= 1e-4   and   ||a-b||_2 / ||b||_2 <= 1e-5
(conftest.assert_close).  Integer quantities (frame counts, table edges) are compared exactly.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_close, synth_utterance

pytestmark = pytest.mark.gpu


def make_pair(pkg, orc, ibs, W=400, S=160, nb=40, sr=16000.0, low=64.0, high=None, nc=13, c0=False, lift=22.0,
              norm=0, dyn=2, l1=3, l2=3, nad=True, fft_size=0, channels=1, bug_compat=True, window=None, engine=0):
    """The HIP extractor and the oracle extractor with identical parameters."""
    high = sr / 2 if high is None else high
    if window is None:
        window = pkg.reference_window(W)
    m = pkg.MfccHip(ibs, W, S, nb, sr, low, high, nc, c0, lift, norm, dyn, l1, l2, nad, device=0, fft_size=fft_size,
                    channels=channels, bug_compat=bug_compat, engine=engine)
    m.set_window(window)
    if fft_size:
        # the oracle (like the reference) ties the FFT length to the window: express "W taps zero
        # padded to fft_size" as a window of fft_size taps whose tail is zero
        w_o = np.zeros(fft_size, np.float32)
        w_o[:W] = window
        cfg = orc.make_config(ibs, window_size=fft_size, shift=S, num_banks=nb, sample_rate=sr, low_freq=low,
                              high_freq=high, ceps_len=nc, want_c0=c0, lift_coef=lift, norm=norm, dyn=dyn,
                              delta_l1=l1, delta_l2=l2, norm_after_dyn=nad)
        return m, cfg, w_o
    cfg = orc.make_config(ibs, window_size=W, shift=S, num_banks=nb, sample_rate=sr, low_freq=low, high_freq=high,
                          ceps_len=nc, want_c0=c0, lift_coef=lift, norm=norm, dyn=dyn, delta_l1=l1, delta_l2=l2,
                          norm_after_dyn=nad)
    return m, cfg, window


def groups_of(dyn):
    return {0: 1, 1: 2, 2: 3}[dyn]


def stream_normalised_check(pkg, orc, pcm, ibs, what, norm, dyn, nad=True, alpha=1.0, block=0, **kw):
    """Streams `pcm` block by block through the HIP extractor and the oracle, each with its norm = NONE twin, and
    applies conftest.assert_normalised_close to every block (flush included): statistics from mfx_debug_read(5)
    against the oracle's normaliser instances, un-normalised rows at 1e-4 / 1e-5, normalised rows within the bound
    those imply.  Returns the HIP extractor's normalised rows."""
    from conftest import assert_normalised_close
    m, cfg, w = make_pair(pkg, orc, ibs, norm=norm, dyn=dyn, nad=nad, **kw)
    m0, cfg0, _ = make_pair(pkg, orc, ibs, norm=0, dyn=dyn, nad=nad, **kw)
    o, o0 = orc.OracleMfcc(cfg, w), orc.OracleMfcc(cfg0, w)
    g = groups_of(dyn)
    rows, pos, limit, blk = [], 0, (block or m.get_input_buffer_size()), 0
    while True:
        last = pos >= pcm.size
        if last:
            n = m.flush()
            assert n == o.flush() == m0.flush() == o0.flush()
        else:
            piece = pcm[pos:pos + limit]
            n = m.set_input(piece)
            assert n == o.set_input(piece) == m0.set_input(piece) == o0.set_input(piece)
            pos += limit
        if n > 0:
            for e in (m, o, m0, o0):
                e.set_alpha(alpha)
                e.apply()
            y = m.get_output_data(n)
            cols = y.shape[1] // g
            st = m.debug_read(5).reshape(-1, 2, cols)
            assert_normalised_close(y, o.get_output_data(n), m0.get_output_data(n), o0.get_output_data(n), st,
                                    o.norm_stats(), g, nad, "%s block %d%s" % (what, blk, " (flush)" if last else ""),
                                    norm=norm)
            rows.append(y)
        blk += 1
        if last:
            break
    return np.concatenate(rows)


def batch_normalised_check(pkg, orc, pcm, what, norm, dyn, nad=True, **kw):
    """One utterance through the batch entry (default batch_norm_stats = 0) against the oracle fed the utterance as
    ONE block with the flush rows at their correct place (bug_compat off): the three-part check of conftest.py with
    the statistics of mfx_debug_read(6)."""
    from conftest import assert_normalised_close
    ibs = pcm.size + 1000
    m, cfg, w = make_pair(pkg, orc, ibs, norm=norm, dyn=dyn, nad=nad, **kw)
    m0, cfg0, _ = make_pair(pkg, orc, ibs, norm=0, dyn=dyn, nad=nad, **kw)
    g = groups_of(dyn)
    outs = []
    for e in (m, m0):
        e.batch_plan([0], [pcm.size])
        outs.append(e.batch_run_host(pcm))
    y, x = outs
    cols = y.shape[1] // g
    st = m.debug_read(6).reshape(-1, 1, 2, cols)[:, 0]
    x_want = orc.run_utterance(cfg0, pcm, w, bug_compat=False)
    o = orc.OracleMfcc(cfg, w, bug_compat=False)
    n = o.set_input(pcm)
    o.apply()
    rows, st_want = [o.get_output_data(n)], o.norm_stats()
    nf = o.flush()
    if nf > 0:
        o.apply()
        rows.append(o.get_output_data(nf))
    assert_normalised_close(y, np.concatenate(rows), x, x_want, st, st_want, g, nad, what, norm=norm)
    return y


# ---------------------------------------------------------------------------------------------
# C1: the reference's own audio file
# ---------------------------------------------------------------------------------------------

def test_c1_streaming_multi_block(pkg, orc, a0001):
    m, cfg, w = make_pair(pkg, orc, 32000, nb=26)
    got = m.process_stream(a0001)
    want = orc.run_utterance(cfg, a0001, w)
    assert got.shape == (711, 39)
    assert_close(got, want, "C1 streaming", groups=3)
    gold = np.load(os.path.join(GOLDEN, "c1_a0001_oracle.npz"))["multi_block"]
    assert_close(got, gold, "C1 streaming vs committed fixture", groups=3)


def test_c1_block_structure_and_taps(pkg, orc, a0001):
    """Frame counts per block, table contents and the magnitude spectrum, stage by stage."""
    m, cfg, w = make_pair(pkg, orc, 32000, nb=26)
    o = orc.OracleMfcc(cfg, w)
    t = o.tables()
    assert np.array_equal(m.debug_read(1), t["filter_beg"])
    assert np.array_equal(m.debug_read(0).reshape(2, -1), t["filters"])
    assert np.array_equal(m.debug_read(2).reshape(26, 13), t["dct_matrix"])
    assert m.get_input_buffer_size() == o.input_buffer_size == 31920
    assert m.get_output_data_width() == o.width == 39
    assert m.estimated_window_count(114000) == 711
    pos, blocks = 0, []
    while pos < a0001.size:
        blk = a0001[pos:pos + 31920]
        n_m, n_o = m.set_input(blk), o.set_input(blk)
        assert n_m == n_o
        wcnd = n_o + (6 if pos == 0 else 12)
        fft = o.tap("fft", wcnd).reshape(wcnd, 512, 2)[:, :257].astype(np.float64)
        mag_o = np.sqrt(fft[..., 0] ** 2 + fft[..., 1] ** 2) / 512
        mag_m = m.debug_read(3).reshape(wcnd, -1)[:, :257]
        assert np.abs(mag_m - mag_o).max() <= 2e-6 * mag_o.max()
        m.apply()
        o.apply()
        assert_close(m.get_output_data(n_m), o.get_output_data(n_o), "block at %d" % pos, groups=3)
        blocks.append(n_m)
        pos += 31920
    assert blocks == [192, 199, 200, 114]
    assert m.flush() == o.flush() == 6
    assert m.flush() == 0  # nothing left (mfcccpu.cpp:350-351)
    m.apply()
    o.apply()
    assert_close(m.get_output_data(6), o.get_output_data(6), "flush", groups=3)


@pytest.mark.parametrize("bug_compat", [True, False])
def test_c1_single_block_b1(pkg, orc, a0001, bug_compat):
    """Reference behaviour B1 (static rows of the flush block after exactly one set_input) is
    reproduced with bug_compat and fixed without."""
    m, cfg, w = make_pair(pkg, orc, 10000000, nb=26, bug_compat=bug_compat)
    got = m.process_stream(a0001)
    want = orc.run_utterance(cfg, a0001, w, bug_compat=bug_compat)
    assert_close(got, want, "C1 single block bug_compat=%s" % bug_compat, groups=3)
    multi = np.load(os.path.join(GOLDEN, "c1_a0001_oracle.npz"))["multi_block"]
    diff_rows = np.unique(np.nonzero(np.abs(got - multi) > 1e-2)[0])
    assert list(diff_rows) == ([705, 706, 707, 708, 709, 710] if bug_compat else [])


def test_c1_batch_entry(pkg, orc, a0001):
    m, cfg, w = make_pair(pkg, orc, 32000, nb=26)
    rows, total = m.batch_plan([0], [a0001.size])
    assert total == 711 and m.batch_frames(a0001.size) == 711
    got = m.batch_run_host(a0001)
    assert_close(got, orc.run_utterance(cfg, a0001, w), "C1 batch", groups=3)


def test_a1_reference_main_defaults(pkg, orc, a1):
    """ASR_OCL.cpp:560 defaults: 15 banks, 12 ceps + c0 (last column), CVN, no dyn."""
    total = stream_normalised_check(pkg, orc, a1, 32000, "a1 main defaults (CVN)", nb=15, nc=12, c0=True, norm=2, dyn=0)
    assert total.shape == (504, 13)
    # the committed fixture is the oracle's own output: a regression pin on the same three-part check's result
    gold = np.load(os.path.join(GOLDEN, "c1_a0001_oracle.npz"))["a1_main_defaults"]
    want = orc.run_utterance(orc.make_config(32000, num_banks=15, ceps_len=12, want_c0=True, norm=2, dyn=0), a1)
    assert np.array_equal(want, gold)


# ---------------------------------------------------------------------------------------------
# C2-shaped batches: ragged, empty, silent, odd offsets
# ---------------------------------------------------------------------------------------------

def test_c2_ragged_batch(pkg, orc):
    lens = [160000, 16000, 400, 559, 560, 399, 0, 48000, 2000, 12345]
    utts = [synth_utterance(n, u) for u, n in enumerate(lens)]
    utts[7][:] = 0                      # a silent utterance: exercises the 1e-30 log floor
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos += n + (n & 1)              # keep offsets even (aligned path)
    pcm = np.zeros(pos + 8, np.int16)
    for o_, u in zip(offs, utts):
        pcm[o_:o_ + u.size] = u
    m, cfg, w = make_pair(pkg, orc, 200000)
    rows, total = m.batch_plan(offs, lens)
    got = m.batch_run_host(pcm)
    exp_frames = [max((n - 240) // 160, 0) for n in lens]
    assert total == sum(exp_frames)
    assert list(rows) == list(np.cumsum([0] + exp_frames[:-1]))
    for u, n in enumerate(lens):
        T = exp_frames[u]
        if T == 0:
            continue
        blk = got[rows[u]:rows[u] + T]
        if u == 7:
            # silence: every mel energy is log(1e-30) = -69.08, whose DCT is zero up to rounding
            # (a few 1e-5, summation-order dependent) -- compare on the scale of the inputs instead
            want = orc.run_utterance(cfg, utts[u], w, bug_compat=False)
            assert np.abs(blk - want).max() <= 1e-4 * 69.08
            continue
        if T >= 12:
            want = orc.run_utterance(cfg, utts[u], w, bug_compat=False)
        else:
            # fewer than 2D frames: the streaming reference either refuses the file (T <= D: window
            # count too small) or restarts its flush block off the frame grid (its carry-over starts
            # at sample (T - 2D)*S + ... < 0 frames, segmentercpu.cpp:69-73); the batch entry defines
            # such files by the whole-utterance formulas, checked against the numpy restatement
            import np_restatement as NP
            want = NP.mfcc_batch(utts[u], w, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, 2, 3, 3)
        assert_close(blk, want, "utt %d (%d samples)" % (u, n), groups=3)
    silent = got[rows[7]:rows[7] + exp_frames[7]]
    assert np.all(silent[:, 13:] == 0)


def test_c2_odd_offsets_take_unaligned_path(pkg, orc):
    lens = [16000, 8001, 4000]
    utts = [synth_utterance(n, 20 + u) for u, n in enumerate(lens)]
    offs = [1, 16003, 24005]            # odd sample offsets -> 2-byte aligned frames
    pcm = np.zeros(30000, np.int16)
    for o_, u in zip(offs, utts):
        pcm[o_:o_ + u.size] = u
    m, cfg, w = make_pair(pkg, orc, 20000)
    rows, total = m.batch_plan(offs, lens)
    got = m.batch_run_host(pcm)
    for u in range(3):
        want = orc.run_utterance(cfg, utts[u], w, bug_compat=False)
        assert_close(got[rows[u]:rows[u] + want.shape[0]], want, "odd offset utt %d" % u, groups=3)


def test_batch_equals_streaming_on_device(pkg, orc):
    pcm = synth_utterance(64000, 77)
    m, cfg, w = make_pair(pkg, orc, 16000)
    s = m.process_stream(pcm)
    m.batch_plan([0], [pcm.size])
    b = m.batch_run_host(pcm)
    assert_close(b, s, "batch vs streaming", tol_max=1e-5, tol_l2=5e-6, groups=3)


# ---------------------------------------------------------------------------------------------
# parameter space: dyn / c0 / mel-only / normalisation / VTLN / other FFT sizes / stereo
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dyn,l1,l2", [(0, 1, 1), (1, 2, 2), (2, 3, 3), (2, 1, 2), (2, 2, 1)])
@pytest.mark.parametrize("nc,c0", [(13, False), (12, True), (0, False)])
def test_dyn_and_output_layout(pkg, orc, dyn, l1, l2, nc, c0):
    pcm = synth_utterance(40000, 3)
    m, cfg, w = make_pair(pkg, orc, 12000, nb=26, nc=nc, c0=c0, dyn=dyn, l1=l1, l2=l2)
    got = m.process_stream(pcm)
    want = orc.run_utterance(cfg, pcm, w)
    cols = (nc + (1 if c0 else 0)) if nc > 0 else 26
    assert got.shape[1] == cols * groups_of(dyn) == m.get_output_data_width()
    assert_close(got, want, "stream dyn=%d nc=%d c0=%s" % (dyn, nc, c0), groups=groups_of(dyn))
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), want, "batch dyn=%d" % dyn, groups=groups_of(dyn))


@pytest.mark.parametrize("norm", [1, 2, 3])
@pytest.mark.parametrize("nad", [True, False])
def test_normalisation(pkg, orc, norm, nad):
    """Streaming: per-block statistics as the reference (normalizercpu.cpp:22-27), the flush block re-using the
    previous block's (mfcccpu.cpp:384,388).  Every block passes the three-part check of conftest.py."""
    pcm = synth_utterance(50000, 11)
    got = stream_normalised_check(pkg, orc, pcm, 16000, "stream norm=%d nad=%s" % (norm, nad), norm=norm, dyn=2, nad=nad,
                                  nb=26)
    assert got.shape == (311, 39)


@pytest.mark.parametrize("norm", [1, 2, 3])
@pytest.mark.parametrize("nad", [True, False])
@pytest.mark.parametrize("mode", [0, 1])
def test_batch_normalisation_semantics(pkg, orc, norm, nad, mode):
    """Batch entry.  batch_norm_stats = 0 (default): what the reference delivers for an utterance it consumes as ONE
    block (its default sample_limit holds ten minutes): statistics over the T - D rows of that block, re-used for the
    D flush rows (mfcccpu.cpp:377-388,395-407; normalizercpu.cpp:22-27) -- compared with the oracle run as a single
    block (flush rows at their correct place, i.e. B1 fixed).  batch_norm_stats = 1: statistics over all T rows
    (what no reference run produces; checked against the definition in float64)."""
    from conftest import assert_normalised_close
    pcm = synth_utterance(50000, 11)
    w = pkg.reference_window(400)
    mk = lambda nrm: pkg.MfccHip(100000, 400, 160, 26, 16000.0, 64.0, 8000.0, 13, False, 22.0, nrm, 2, 3, 3, nad,
                                 device=0, batch_norm_stats=mode)
    m, m0 = mk(norm), mk(0)
    outs = []
    for e in (m, m0):
        e.set_window(w)
        e.batch_plan([0], [pcm.size])
        outs.append(e.batch_run_host(pcm))
    y, x = outs
    T = y.shape[0]
    st = m.debug_read(6).reshape(-1, 1, 2, 13)[:, 0]
    cfg = orc.make_config(100000, num_banks=26, ceps_len=13, norm=norm, dyn=2, norm_after_dyn=nad)
    cfg0 = orc.make_config(100000, num_banks=26, ceps_len=13, norm=0, dyn=2, norm_after_dyn=nad)
    x_want = orc.run_utterance(cfg0, pcm, w, bug_compat=False)
    if mode == 0 or not nad:
        # (before the deltas the reference's first block normalises all T rows with context: both modes coincide)
        o = orc.OracleMfcc(cfg, w, bug_compat=False)
        n = o.set_input(pcm)
        assert n == T - 6
        o.apply()
        head, st_want = o.get_output_data(n), o.norm_stats()
        assert o.flush() == 6
        o.apply()
        y_want = np.concatenate([head, o.get_output_data(6)])
    else:
        xs = x_want.astype(np.float64).reshape(T, 3, 13)
        mean = xs.mean(0)
        if norm == 1:
            mult = np.ones_like(mean)
        elif norm == 2:
            mult = 1.0 / xs.std(0, ddof=1)
        else:
            mult = 1.0 / np.maximum(np.abs(xs.min(0) - mean), np.abs(xs.max(0) - mean))
        st_want = np.stack([mean, mult], axis=1)
        y_want = ((xs - mean) * mult).reshape(T, 39)
    assert_normalised_close(y, y_want, x, x_want, st, st_want, 3, nad, "batch norm=%d nad=%s mode=%d" % (norm, nad, mode),
                            norm=norm)


@pytest.mark.parametrize("alpha", [0.88, 1.0, 1.12])
def test_vtln_alpha_sweep_on_one_fft(pkg, orc, alpha):
    """set_input once, then set_alpha + apply repeatedly (ASR_OCL.cpp:236-243)."""
    pcm = synth_utterance(20000, 5)
    m, cfg, w = make_pair(pkg, orc, 30000)
    o = orc.OracleMfcc(cfg, w)
    n = m.set_input(pcm)
    assert n == o.set_input(pcm)
    for a in (1.0, alpha, 1.0):
        m.set_alpha(a)
        o.set_alpha(a)
        m.apply()
        o.apply()
        assert_close(m.get_output_data(n), o.get_output_data(n), "alpha %.2f" % a, groups=3)
    assert np.array_equal(m.debug_read(1), o.tables()["filter_beg"])


@pytest.mark.parametrize("norm,nad", [(0, True), (2, True), (1, False)])
def test_vtln_sweep_in_one_call(pkg, orc, norm, nad):
    """mfx_apply_alphas: the reference's alpha loop (ASR_OCL.cpp:236-243) as one call per block, all
    warped filterbanks over the stored spectrum in one launch per stage.  Streamed in several blocks
    plus flush; every alpha's rows must equal the oracle run with that alpha alone and be bit-identical
    to set_alpha + apply + get_output_data on the same handle."""
    alphas = [0.88, 1.0, 1.12, 0.94]
    pcm = synth_utterance(41000, 17)
    m, cfg, w = make_pair(pkg, orc, 20000, norm=norm, nad=nad)
    got = [[] for _ in alphas]
    ref_same_handle = [[] for _ in alphas]

    def emit(n):
        if n <= 0:
            return
        m.apply_alphas(alphas)
        for i in range(len(alphas)):
            got[i].append(m.get_output_data_alpha(i, n))
        if norm == 0:       # with normalisation the per-alpha loop shares one statistics slot (header note)
            for i, a in enumerate(alphas):
                m.set_alpha(a)
                m.apply()
                ref_same_handle[i].append(m.get_output_data(n))

    for pos in range(0, pcm.size, 16000):
        emit(m.set_input(pcm[pos:pos + 16000]))
    emit(m.flush())
    for i, a in enumerate(alphas):
        o = orc.OracleMfcc(cfg, w)
        rows = []
        for pos in range(0, pcm.size, 16000):
            n = o.set_input(pcm[pos:pos + 16000])
            if n > 0:
                o.set_alpha(a)
                o.apply()
                rows.append(o.get_output_data(n))
        n = o.flush()
        if n > 0:
            o.set_alpha(a)
            o.apply()
            rows.append(o.get_output_data(n))
        want = np.concatenate(rows)
        g = np.concatenate(got[i])
        assert g.shape == want.shape
        if norm == 0:
            assert_close(g, want, "sweep alpha %.2f" % a, groups=3)
            assert np.array_equal(g, np.concatenate(ref_same_handle[i]))
        else:
            # a fresh handle streamed with this alpha alone passes the three-part normalisation check against the
            # oracle block by block; the sweep's rows for that alpha are the same bits
            alone = stream_normalised_check(pkg, orc, pcm, 20000, "alpha %.2f alone, norm %d" % (a, norm), norm=norm,
                                            dyn=2, nad=nad, alpha=a, block=16000)
            assert np.array_equal(g, alone)
    with pytest.raises(Exception):
        m.get_output_data_alpha(len(alphas), 1)


@pytest.mark.parametrize("W,S,off", [(400, 160, 0), (400, 161, 0), (400, 160, 1), (512, 128, 0), (511, 127, 0)])
def test_c3_shape_1024_override(pkg, orc, W, S, off):
    """BASELINE configs[2] in small: 25 ms window zero padded to a 1024-point FFT, 80 mel, 13 MFCC; also with an odd
    shift / odd first sample (the short-window build that loads single samples) and at the build's limit of 512 taps."""
    full = synth_utterance(60000 + off, 9)
    pcm = np.ascontiguousarray(full[off:])
    m, cfg, w_o = make_pair(pkg, orc, 20000, W=W, S=S, nb=80, dyn=0, fft_size=1024)
    assert m.fft_size() == 1024
    m.batch_plan([off], [pcm.size])   # (an odd first sample: single-sample loads in the kernel)
    got = m.batch_run_host(full)
    want = orc.run_utterance(cfg, pcm, w_o, bug_compat=False)
    # the oracle's 1024-tap window loses the last few frames: compare the common prefix
    assert got.shape[0] >= want.shape[0] and got.shape[0] - want.shape[0] <= 5
    assert_close(got[:want.shape[0]], want, "C3 shape")
    s = m.process_stream(pcm)
    assert_close(s, got, "C3 streaming vs batch", tol_max=2e-6, tol_l2=1e-6)


_F1024 = [
    # W,   S,  nb, nc, c0,    dyn, alpha
    (400, 160, 80, 13, False, 0, 1.0),     # BASELINE configs[2]
    (400, 160, 80, 13, False, 2, 1.0),     # with the delta stages (statics through the compact scratch)
    (400, 161, 64, 12, True, 1, 1.0),      # odd shift: single-sample loads
    (512, 256, 23, 15, True, 0, 1.0),      # the build's longest window, c0 -> 16 output columns
    (300, 100, 40, 13, False, 2, 0.9),     # VTLN warp: the lane plan is rebuilt for the warped filterbank
    (416, 208, 8, 4, False, 0, 1.0),       # few, long filters (one round of 136 bins)
    (417, 139, 77, 10, False, 1, 1.1),     # 14 rows of samples (the 16-row build), odd everything
    (800, 320, 64, 13, False, 2, 1.0),     # window longer than 512 samples: the frame's halves are folded (24-row build)
    (1024, 256, 80, 13, False, 0, 1.0),    # full-length window (32-row build)
    (600, 200, 40, 12, True, 1, 0.9),
    (1000, 333, 48, 13, False, 1, 1.0),    # long window on unaligned frames: stays on k_front_reg
    (400, 160, 80, 0, False, 0, 1.0),      # filterbank features: 80 log mel energies per frame, no DCT
    (1024, 256, 80, 0, False, 2, 1.0),     # ... full-length window, with deltas (240 columns)
    (551, 220, 40, 0, False, 1, 1.05),     # ... 25 ms at 22.05 kHz, VTLN
]


@pytest.mark.parametrize("W,S,nb,nc,c0,dyn,alpha", _F1024)
def test_front1024_configurations(pkg, orc, W, S, nb, nc, c0, dyn, alpha):
    """k_front1024 (1024 points, <= 80 filters, <= 16 columns; windows longer than 512 samples on aligned frames only): ragged utterances at odd and even
    offsets through the batch entry against the oracle fed each utterance alone (its 1024-tap window loses the last few
    frames: common prefix), and the same batch through k_front_reg (mfx_config.engine = MFX_ENGINE_NO_FRONT1024: a different factorisation of
    the same transform) within the same tolerance."""
    import os
    rng = np.random.default_rng(W * 7 + S)
    frames = [1, 5, 16, 17, 64, 131]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    if W > 512 and S % 2 == 0:
        lens = [n + (n & 1) for n in lens]     # long windows: every utterance at an even offset (aligned frames)
    offs, pos = [], 0 if (S % 2 == 0) else 1
    for n in lens:
        offs.append(pos)
        pos += n + (int(rng.integers(0, 4)) if S % 2 else 2 * int(rng.integers(0, 3)))
    pcm = np.zeros(pos, np.int16)
    utts = [synth_utterance(n, 900 + 17 * i) for i, n in enumerate(lens)]
    for o_, u in zip(offs, utts):
        pcm[o_:o_ + u.size] = u
    kw = dict(W=W, S=S, nb=nb, nc=nc, c0=c0, dyn=dyn, l1=2, l2=2, fft_size=1024)
    m, cfg, w_o = make_pair(pkg, orc, max(lens) + 2000, **kw)
    assert m.fft_size() == 1024
    if alpha != 1.0:
        m.set_alpha(alpha)
    rows, total = m.batch_plan(offs, lens)
    assert m.dominant_kernel_name() == ("k_front1024" if (W <= 512 or S % 2 == 0) else "k_front_reg")
    got = m.batch_run_host(pcm)
    assert total == sum(frames) and got.shape[0] == total
    m2, _, _ = make_pair(pkg, orc, max(lens) + 2000, engine=pkg.mfcc.ENGINE_NO_FRONT1024, **kw)
    assert m2.dominant_kernel_name() == "k_front_reg"
    if alpha != 1.0:
        m2.set_alpha(alpha)
    m2.batch_plan(offs, lens)
    g = groups_of(dyn)
    assert_close(got, m2.batch_run_host(pcm), "k_front1024 vs k_front_reg", groups=g)
    # the kernel's two builds -- 16 waves per CU (aligned frames, windows up to 512 samples) and 12 (everything else, or on
    # request: MFX_ENGINE_FRONT1024_12_WAVES) -- are the same arithmetic: the same bits
    m3, _, _ = make_pair(pkg, orc, max(lens) + 2000, engine=pkg.mfcc.ENGINE_FRONT1024_12_WAVES, **kw)
    if alpha != 1.0:
        m3.set_alpha(alpha)
    m3.batch_plan(offs, lens)
    assert m3.dominant_kernel_name() == m.dominant_kernel_name()
    assert np.array_equal(got, m3.batch_run_host(pcm))
    D = (2 + (2 if dyn == 2 else 0)) if dyn else 0
    checked = 0
    for i, (T, u) in enumerate(zip(frames, utts)):
        try:
            want = orc.run_utterance(cfg, u, w_o, alpha=alpha, bug_compat=False)
        except RuntimeError:   # fewer frames under the oracle's 1024-tap window than the reference accepts
            continue
        r0 = int(rows[i])
        n_cmp = want.shape[0]
        if n_cmp == 0:
            continue
        assert T >= n_cmp
        if dyn and n_cmp < T:
            n_cmp = max(0, n_cmp - D)   # the oracle's last rows replicate ITS last frame in the delta window
        if n_cmp:
            assert_close(got[r0:r0 + n_cmp], want[:n_cmp], "utterance %d" % i, groups=g)
            checked += 1
    assert checked >= 2


def test_c5_shape_2048_odd_shift_stereo(pkg, orc):
    """BASELINE configs[4] in small: 44.1 kHz, W=1102, S=441 (odd), 2048-pt, 128 mel, 40 MFCC + d + dd,
    stereo input downmixed (L+R)>>1."""
    sr = 44100.0
    n = 90000
    left, right = synth_utterance(n, 31, sr=sr), synth_utterance(n, 32, sr=sr, f=523.0)
    mono = ((left.astype(np.int32) + right.astype(np.int32)) >> 1).astype(np.int16)
    m, cfg, w = make_pair(pkg, orc, 100000, W=1102, S=441, nb=128, sr=sr, nc=40, dyn=2)
    assert m.fft_size() == 2048 and m.get_output_data_width() == 120
    want = orc.run_utterance(cfg, mono, w, bug_compat=False)
    m.batch_plan([0], [n])
    got_mono = m.batch_run_host(mono)
    assert_close(got_mono, want, "C5 mono", groups=3)
    assert_close(m.process_stream(mono, block_samples=30000), orc.run_utterance(cfg, mono, w, block_samples=30000),
                 "C5 streaming", groups=3)
    ms, _, _ = make_pair(pkg, orc, 100000, W=1102, S=441, nb=128, sr=sr, nc=40, dyn=2, channels=2)
    ms.batch_plan([0], [n])
    inter = np.empty(2 * n, np.int16)
    inter[0::2], inter[1::2] = left, right
    assert_close(ms.batch_run_host(inter), want, "C5 stereo downmix", groups=3)


@pytest.mark.parametrize("nb,nc,c0", [(40, 20, False), (64, 24, True), (80, 13, False), (128, 40, False)])
def test_512pt_wide_outputs_and_many_filters(pkg, orc, nb, nc, c0):
    """512-point front end with more than 16 output columns (DCT from the LDS mel scratch instead of
    the DPP-fused path) and with many mel rounds / large tables (falls back to spectrum + melcep when
    the fused kernel's LDS budget is exceeded)."""
    pcm = synth_utterance(30000, 21)
    m, cfg, w = make_pair(pkg, orc, 12000, nb=nb, nc=nc, c0=c0, dyn=1, l1=2)
    want = orc.run_utterance(cfg, pcm, w, bug_compat=False)
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), want, "512-pt nb=%d nc=%d batch" % (nb, nc), groups=2)
    assert_close(m.process_stream(pcm), orc.run_utterance(cfg, pcm, w), "512-pt nb=%d nc=%d stream" % (nb, nc), groups=2)


def test_window_8khz_256(pkg, orc):
    pcm = synth_utterance(24000, 13, sr=8000.0)
    m, cfg, w = make_pair(pkg, orc, 8000, W=200, S=80, nb=23, sr=8000.0, nc=12, c0=True)
    assert m.fft_size() == 256
    assert_close(m.process_stream(pcm), orc.run_utterance(cfg, pcm, w), "8 kHz / 256-pt", groups=3)


def test_window_512_full_taps(pkg, orc):
    """W = 512 exactly (no zero padding): the 16-row variant of the 512-point kernel."""
    pcm = synth_utterance(30000, 17)
    m, cfg, w = make_pair(pkg, orc, 10000, W=512, S=128, nb=40)
    assert_close(m.process_stream(pcm), orc.run_utterance(cfg, pcm, w), "W=512", groups=3)
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), orc.run_utterance(cfg, pcm, w, bug_compat=False), "W=512 batch", groups=3)


# ---------------------------------------------------------------------------------------------
# error behaviour (same messages as the reference's std::runtime_error)
# ---------------------------------------------------------------------------------------------

def test_errors(pkg):
    m = pkg.MfccHip(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, 0, 2, 3, 3, True)
    with pytest.raises(pkg.MfxError, match="set_window"):
        m.set_input(np.zeros(4000, np.int16))
    m.set_window(pkg.reference_window(400))
    with pytest.raises(pkg.MfxError, match="buffer is too small") as e:
        m.set_input(np.zeros(m.get_input_buffer_size() + 1, np.int16))
    assert e.value.status == -1
    with pytest.raises(pkg.MfxError, match="window count is too small") as e:
        m.set_input(np.zeros(1000, np.int16))     # first block shorter than the delta context
    assert e.value.status == -2
    with pytest.raises(pkg.MfxError, match="Window count too high"):
        m.get_output_data(10 ** 6)
    with pytest.raises(pkg.MfxError):
        pkg.MfccHip(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 0.0)   # lift_coef 0 divides by zero
    with pytest.raises(pkg.MfxError):
        pkg.MfccHip(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, fft_size=300)


def test_object_reuse_after_flush(pkg, orc):
    """A second file on the same object starts a new stream."""
    m, cfg, w = make_pair(pkg, orc, 16000)
    a, b = synth_utterance(30000, 40), synth_utterance(25000, 41)
    first = m.process_stream(a)
    second = m.process_stream(b)
    assert_close(first, orc.run_utterance(cfg, a, w), "file 1", groups=3)
    assert_close(second, orc.run_utterance(cfg, b, w), "file 2", groups=3)


# ---------------------------------------------------------------------------------------------
# full-size C2 (BASELINE configs[1]): properties that do not need the oracle at full size
# ---------------------------------------------------------------------------------------------

def test_c2_full_size_properties(pkg, orc):
    import torch
    n_utt, n = 1000, 160000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pcm = (3000.0 * torch.randn((n_utt, n), generator=g, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    pcm[500] = pcm[3]                    # duplicates: position in the batch must not matter
    pcm[999] = pcm[3]
    pcm[17] = 0                          # silence
    m, cfg, w = make_pair(pkg, orc, n + 1000)
    rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
    assert total == 998000
    out = torch.empty((total, 39), dtype=torch.float32, device=dev)
    out.fill_(float("nan"))
    m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
    m.synchronize()
    assert bool(torch.isfinite(out).all())            # every row written
    o3 = out[rows[3]:rows[3] + 998]
    assert torch.equal(o3, out[rows[500]:rows[500] + 998]) and torch.equal(o3, out[rows[999]:rows[999] + 998])
    sil = out[rows[17]:rows[17] + 998]
    assert bool((sil[:, 13:] == 0).all())
    # idempotence: a second pass reproduces the first bit for bit
    out2 = torch.empty_like(out)
    m.batch_run_device(pcm.data_ptr(), pcm.numel(), out2.data_ptr())
    m.synchronize()
    assert torch.equal(out, out2)
    # splitting the batch does not change anything (checksum of checksums)
    m2, _, _ = make_pair(pkg, orc, n + 1000)
    half = n_utt // 2
    r2, t2 = m2.batch_plan(np.arange(half) * n, np.full(half, n))
    outa = torch.empty((t2, 39), dtype=torch.float32, device=dev)
    m2.batch_run_device(pcm.data_ptr(), pcm.numel(), outa.data_ptr())
    m2.synchronize()
    assert torch.equal(outa, out[:t2])
    # sampled parity against the oracle
    assert float(sil[:, :13].abs().max()) <= 1e-4 * 69.08   # DCT of a constant log(1e-30) vector
    for u in (0, 3, 421, 640):
        want = orc.run_utterance(cfg, pcm[u].cpu().numpy(), w, bug_compat=False)
        assert_close(out[rows[u]:rows[u] + 998].cpu().numpy(), want, "C2 utt %d" % u, groups=3)


# ---------------------------------------------------------------------------------------------
# the C++ mirror (class MfccHip : MfccBase) and the afet-style driver built on it
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("W,S,fft", [(2400, 960, 4096), (3000, 1001, 4096), (1024, 256, 1024), (2048, 512, 2048),
                                     (1102, 440, 2048), (1280, 320, 2048), (1281, 320, 2048), (1282, 441, 2048)])
def test_long_transforms_register_kernel(pkg, orc, W, S, fft):
    """k_front_reg at its three sizes, full-length windows (no zero padding inside the transform) and windows either
    side of the short-window build's limit at 2048 points (1280 samples), even and odd shifts (paired
    32-bit loads vs single samples), batch (fused up to 2048 points, spectrum + melcep at 4096) and streaming, against
    the oracle."""
    sr = 96000.0
    n = 40 * S + W + 123
    pcm = synth_utterance(2 * n, 77, sr=sr)
    m, cfg, w = make_pair(pkg, orc, n, W=W, S=S, nb=64, sr=sr, nc=20, dyn=1, l1=2, l2=0)
    assert m.fft_size() == fft
    want = orc.run_utterance(cfg, pcm, w, bug_compat=False)
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), want, "batch %d" % fft, groups=2)
    assert_close(m.process_stream(pcm), orc.run_utterance(cfg, pcm, w), "stream %d" % fft, groups=2)


def test_c4_per_gpu_share_full_size_properties(pkg, orc):
    """BASELINE configs[3] per GPU: 12 500 utterances x 10 s = 2.0e9 samples (4 GB of PCM, byte offsets past
    2^32) -> 12 475 000 frames.  The batch is 25 copies of one 500-utterance block, so every copy of the
    output must equal the first bit for bit, and the first must equal the block run on its own."""
    import torch
    blk_utt, copies, n = 500, 25, 160000
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 12 * 2 ** 30:
        pytest.skip("needs 12 GiB of free HBM")
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    blk = (3000.0 * torch.randn((blk_utt, n), generator=g, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    pcm = blk.repeat(copies, 1)
    n_utt = blk_utt * copies
    m, cfg, w = make_pair(pkg, orc, n + 1000)
    rows, total = m.batch_plan(np.arange(n_utt, dtype=np.int64) * n, np.full(n_utt, n, dtype=np.int64))
    assert total == 998 * n_utt == 12475000
    out = torch.empty((total, 39), dtype=torch.float32, device=dev)
    out.fill_(float("nan"))
    m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
    m.synchronize()
    per = 998 * blk_utt
    first = out[:per]
    assert bool(torch.isfinite(first).all())
    for c in range(1, copies):
        assert torch.equal(out[c * per:(c + 1) * per], first), "copy %d differs" % c
    m2, _, _ = make_pair(pkg, orc, n + 1000)
    m2.batch_plan(np.arange(blk_utt, dtype=np.int64) * n, np.full(blk_utt, n, dtype=np.int64))
    alone = torch.empty((per, 39), dtype=torch.float32, device=dev)
    m2.batch_run_device(blk.data_ptr(), blk.numel(), alone.data_ptr())
    m2.synchronize()
    assert torch.equal(alone, first)
    want = orc.run_utterance(cfg, blk[7].cpu().numpy(), w, bug_compat=False)
    assert_close(out[(24 * blk_utt + 7) * 998:(24 * blk_utt + 8) * 998].cpu().numpy(), want, "C4 share, last copy, utt 7", groups=3)


def test_c3_full_size_stream_properties(pkg, orc):
    """BASELINE configs[2] at full size: one 57 600 000-sample stream (1 hour), 1024-point FFT, 80 mel, 13 MFCC,
    no deltas -> 359 998 frames.  Without deltas a frame depends on its own 400 samples only, so the stream cut
    in two overlapping utterances must reproduce the same rows bit for bit."""
    import torch
    n = 57600000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    pcm = (3000.0 * torch.randn(n, generator=g, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    m, cfg, w_o = make_pair(pkg, orc, 20000, nb=80, dyn=0, fft_size=1024)
    rows, total = m.batch_plan([0], [n])
    assert total == (n - 240) // 160 == 359998
    out = torch.empty((total, 13), dtype=torch.float32, device=dev)
    out.fill_(float("nan"))
    m.batch_run_device(pcm.data_ptr(), pcm.numel(), out.data_ptr())
    m.synchronize()
    assert bool(torch.isfinite(out).all())
    cut = 180000                                   # frames in the first piece
    m2, _, _ = make_pair(pkg, orc, 20000, nb=80, dyn=0, fft_size=1024)
    r2, t2 = m2.batch_plan([0, cut * 160], [cut * 160 + 240, n - cut * 160])
    assert t2 == total and list(r2) == [0, cut]
    out2 = torch.empty_like(out)
    m2.batch_run_device(pcm.data_ptr(), pcm.numel(), out2.data_ptr())
    m2.synchronize()
    assert torch.equal(out, out2)
    # sampled parity: a 2-second excerpt from the middle against the oracle (same frames, same samples)
    f0 = 200000
    seg = pcm[f0 * 160:f0 * 160 + 32000 + 1024].cpu().numpy()
    want = orc.run_utterance(cfg, seg, w_o, bug_compat=False)
    k = min(want.shape[0], 190)
    assert_close(out[f0:f0 + k].cpu().numpy(), want[:k], "C3 excerpt")


def test_cpp_driver_text_output(orc, a0001, tmp_path):
    """asr-featext-opencl_amd/host/afet_hip: the reference's per-file loop (ASR_OCL.cpp:163-321) in
    C++ over MfccHip; its text rows ("| time | v | v | ...", %f) must match the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    out = tmp_path / "a0001.txt"
    subprocess.check_call([exe, "--banks", "26", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2", "--l1", "3",
                           "--l2", "3", "--sample-limit", "32000", os.path.join(GOLDEN, "a0001.wav"), str(out)])
    rows = [[float(v) for v in line.strip().strip("|").split("|")] for line in open(out)]
    got = np.array(rows, dtype=np.float64)
    assert got.shape == (711, 40)
    want = orc.run_utterance(orc.make_config(32000, num_banks=26, ceps_len=13), a0001)
    # column 0 is the frame time as the reference prints it: (0.5f * window_ms + t * shift_ms) / sample_rate, its
    # milliseconds divided by Hz (ASR_OCL.cpp:225-226,254) -- 0.00078125 + 0.000625 t at 25/10 ms, 16 kHz
    np.testing.assert_allclose(got[:, 0], 0.00078125 + 0.000625 * np.arange(711), atol=1e-6)
    assert np.abs(got[:, 1:] - want).max() <= 1e-4 * np.abs(want).max() + 1e-6   # + %f quantisation
    # --bug-compat 0: the same rows with the time column in seconds (0.5 * window + t * shift)
    out0 = tmp_path / "a0001_s.txt"
    subprocess.check_call([exe, "--bug-compat", "0", "--banks", "26", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2",
                           "--l1", "3", "--l2", "3", "--sample-limit", "32000", os.path.join(GOLDEN, "a0001.wav"), str(out0)])
    got0 = np.array([[float(v) for v in line.strip().strip("|").split("|")] for line in open(out0)])
    np.testing.assert_allclose(got0[:, 0], 0.0125 + 0.01 * np.arange(711), atol=1e-6)
    assert np.array_equal(got0[:, 1:], got[:, 1:])


def test_cpp_driver_sphere_input_and_htk_output(orc, tmp_path):
    """The reference's own sample1.wav is a NIST SPHERE file (libsndfile reads it, ASR_OCL.cpp:170-175);
    soundfiles/sample1_1.wav holds the same 54682 samples as RIFF.  The driver must read both to the
    same features, and its HTK binary output (big-endian header + float32 rows; the reference's binary
    branch is a stub, ASR_OCL.cpp:315-319) must hold the floats the text rows print."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    opts = ["--banks", "26", "--ceps", "13", "--c0", "1", "--norm", "0", "--dyn", "2", "--l1", "3", "--l2", "3",
            "--sample-limit", "20000"]
    sph, riff = os.path.join(GOLDEN, "sample1_sphere.wav"), os.path.join(GOLDEN, "sample1_riff.wav")
    t_s, t_r, h_s = tmp_path / "s.txt", tmp_path / "r.txt", tmp_path / "s.htk"
    subprocess.check_call([exe] + opts + [sph, str(t_s), riff, str(t_r)])
    subprocess.check_call([exe] + opts + ["--htk", sph, str(h_s)])
    assert open(t_s).read() == open(t_r).read()
    rows = np.array([[float(v) for v in line.strip().strip("|").split("|")] for line in open(t_s)])
    pcm, sr = orc.read_wav_pcm16(riff)
    pcm = pcm[:, 0].copy()
    assert pcm.size == 54682 and sr == 16000
    want = orc.run_utterance(orc.make_config(20000, num_banks=26, ceps_len=13, want_c0=True), pcm)
    assert rows.shape == (want.shape[0], 1 + 42)
    assert np.abs(rows[:, 1:] - want).max() <= 1e-4 * np.abs(want).max() + 1e-6
    raw = open(h_s, "rb").read()
    n, period, size, kind = struct.unpack(">iihh", raw[:12])
    assert (n, period, size) == (want.shape[0], 100000, 4 * 42)
    assert kind & 0xFFFF == 6 | 0x2000 | 0x0100 | 0x0200          # MFCC_0_D_A
    htk = np.frombuffer(raw[12:], dtype=">f4").reshape(n, 42)
    assert np.abs(htk - rows[:, 1:]).max() <= 5.1e-7 * max(1.0, np.abs(htk).max()) + 5e-7   # %f rounding


def test_cpp_driver_alpha_range_uses_the_sweep(orc, a0001, tmp_path):
    """--alpha-min/--alpha-max/--alpha-step (the reference's option table, ASR_OCL.cpp:569-669): one output
    file per warp factor, written from MfccHip::apply_alphas; each must match the oracle at that alpha."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    out = tmp_path / "v.txt"
    subprocess.check_call([exe, "--banks", "26", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2", "--l1", "3",
                           "--l2", "3", "--sample-limit", "48000", "--alpha-min", "0.9", "--alpha-max", "1.15",
                           "--alpha-step", "0.1", os.path.join(GOLDEN, "a0001.wav"), str(out)])
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["v.txt.0.900000", "v.txt.1.000000", "v.txt.1.100000"]
    cfg = orc.make_config(48000, num_banks=26, ceps_len=13)
    for name, alpha in zip(names, (np.float32(0.9), np.float32(0.9) + np.float32(0.1), np.float32(0.9) + 2 * np.float32(0.1))):
        got = np.array([[float(v) for v in line.strip().strip("|").split("|")] for line in open(tmp_path / name)])
        o = orc.OracleMfcc(cfg, orc.reference_window(400))
        rows = []
        blk = o.input_buffer_size                      # the driver's block size (47920)
        for pos in range(0, a0001.size, blk):
            n = o.set_input(a0001[pos:pos + blk])
            if n > 0:
                o.set_alpha(float(alpha))
                o.apply()
                rows.append(o.get_output_data(n))
        n = o.flush()
        if n > 0:
            o.set_alpha(float(alpha))
            o.apply()
            rows.append(o.get_output_data(n))
        want = np.concatenate(rows)
        assert got.shape == (711, 40)
        assert np.abs(got[:, 1:] - want).max() <= 1e-4 * np.abs(want).max() + 1e-6


def test_cpp_driver_file_queue_workers(tmp_path):
    """--devs: the reference's file queue (ASR_OCL.cpp:340-368) with one worker thread + one extractor per
    listed GPU.  Rehearsed with three workers on this box's single GPU: independent handles on different
    threads must not disturb each other (SURVEY 8b threading), so every output file must be byte-identical
    to the one-worker run; a bad file is reported and does not stop the queue."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    srcs = ["a0001.wav", "a1.wav", "sample1_sphere.wav", "sample1_riff.wav", "a0001.wav", "a1.wav", "a0001.wav"]
    opts = ["--banks", "26", "--ceps", "13", "--norm", "2", "--dyn", "2", "--l1", "3", "--l2", "3",
            "--sample-limit", "30000"]

    def run(tag, extra):
        args = []
        for i, s_ in enumerate(srcs):
            args += [os.path.join(GOLDEN, s_), str(tmp_path / ("%s_%d.txt" % (tag, i)))]
        subprocess.check_call([exe] + opts + extra + args)
        return [open(tmp_path / ("%s_%d.txt" % (tag, i))).read() for i in range(len(srcs))]

    one = run("one", [])
    three = run("three", ["--devs", "0,0,0"])
    assert one == three and all(len(t) > 1000 for t in one)
    assert one[0] == one[4] == one[6] and one[2] == one[3]
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not audio")
    r = subprocess.run([exe] + opts + ["--devs", "0,0", os.path.join(GOLDEN, "a1.wav"), str(tmp_path / "q0.txt"),
                                       str(bad), str(tmp_path / "q1.txt"),
                                       os.path.join(GOLDEN, "a1.wav"), str(tmp_path / "q2.txt")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Exception caught" in r.stderr
    assert open(tmp_path / "q0.txt").read() == open(tmp_path / "q2.txt").read() == one[1]


def test_bench_two_rank_launch_path(tmp_path):
    """bench.py's N>1 path (one process per rank under torch.distributed.run, barrier, max over ranks,
    whole-job aggregate) rehearsed with 2 ranks on this box's single GPU: gloo instead of RCCL, both
    ranks pinned to device 0, tiny workload."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MFX_BENCH_DEVICE="0", MFX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "T", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["frames_rank0_per_step"] == 8 * 98 and d["config"]["frames_per_step"] == 2 * 8 * 98
    # whole-job value = frames of all ranks / max-over-ranks step time
    assert abs(d["value"] - 2 * 8 * 98 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "roofline" in d and "cpu_baseline" not in d
    # the line says what bounds the kernel and carries SURVEY 8(d)'s compute-side model next to the HBM fraction
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "flops_per_frame", "achieved_tflops",
              "frac_fp32_peak", "staged_bytes_per_frame", "staged_pipeline_equivalent", "whole_path"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # the diagnosis is keyed by the kernel that ran and names the committed profile it was read from
    assert rf["kernel"] == "k_front512" and rf["bound_diagnosed"]["label"] == "valu_issue"
    assert rf["bound_diagnosed"]["source"].startswith("profiles/")
    assert 14000 <= rf["flops_per_frame"] <= 16000 and 9000 <= rf["staged_bytes_per_frame"] <= 9300


@pytest.mark.parametrize("norm", [1, 2, 3])
@pytest.mark.parametrize("dyn,nad", [(0, True), (2, True), (2, False)])
def test_one_launch_normaliser_same_bits_as_two_kernels(pkg, orc, norm, dyn, nad):
    """k_norm_seg (statistics + apply of a short segment in one block, rows through LDS) against k_norm_stats +
    k_norm_apply (mfx_config.engine = MFX_ENGINE_NORM_TWO_KERNELS): the same bits, on the batch entry (ragged utterances,
    one segment each) and on the streaming interface (blocks + flush, statistics re-used for the flush rows); the
    two-kernel form is the one every normalisation parity test of rounds 1-2 ran against the oracle
    (normalizercpu.cpp:22-89)."""
    lens = [16000, 4321, 30011, 9000, 16000, 2400]
    pcm = np.concatenate([synth_utterance(n, 40 + i) for i, n in enumerate(lens)])
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    kw = dict(norm=norm, dyn=dyn, nad=nad)
    m1, _, _ = make_pair(pkg, orc, 40000, **kw)
    m2, _, _ = make_pair(pkg, orc, 40000, engine=pkg.mfcc.ENGINE_NORM_TWO_KERNELS, **kw)
    rows1, total1 = m1.batch_plan(offs, lens)
    rows2, total2 = m2.batch_plan(offs, lens)
    assert total1 == total2 and total1 > 0
    a, b = m1.batch_run_host(pcm), m2.batch_run_host(pcm)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    for blk in (16000, 5000):
        got = []
        for m in (m1, m2):
            out = []
            for pos in range(0, 30011, blk):
                n = m.set_input(pcm[pos:pos + blk][:max(0, 30011 - pos)])
                if n > 0:
                    m.apply()
                    out.append(m.get_output_data(n))
            n = m.flush()
            if n > 0:
                m.apply()
                out.append(m.get_output_data(n))
            got.append(np.concatenate(out))
        assert got[0].shape == got[1].shape and np.array_equal(got[0], got[1])


@pytest.mark.parametrize("norm", [1, 2, 3])
@pytest.mark.parametrize("dyn,nad", [(0, True), (2, True), (2, False)])
@pytest.mark.parametrize("nb,nc", [(20, 1), (1, 0)])
def test_one_column_configurations_normalise_correctly(pkg, orc, norm, dyn, nad, nb, nc):
    """k_norm_seg's multiply-high division constant wrapped to 0 for cols == 1 (ceps_len 1 without c0; one filter
    and no DCT) -- every element landed in row 0.  One-column configurations through the streaming interface (three-part
    check against the oracle, normalizercpu.cpp:22-89) and the batch entry, one-launch and two-kernel normaliser: same bits."""
    pcm = synth_utterance(30000, 61)
    got = stream_normalised_check(pkg, orc, pcm, 12000, "one column nb=%d nc=%d norm=%d" % (nb, nc, norm), norm=norm,
                                  dyn=dyn, nad=nad, nb=nb, nc=nc)
    assert got.shape[1] == groups_of(dyn) and np.isfinite(got).all()
    kw = dict(norm=norm, dyn=dyn, nad=nad, nb=nb, nc=nc)
    m1, _, _ = make_pair(pkg, orc, 40000, **kw)
    m2, _, _ = make_pair(pkg, orc, 40000, engine=pkg.mfcc.ENGINE_NORM_TWO_KERNELS, **kw)
    lens = [16000, 4321, 9000]
    offs = [0, 16000, 20322]
    for m in (m1, m2):
        m.batch_plan(offs, lens)
    a, b = m1.batch_run_host(pcm), m2.batch_run_host(pcm)
    assert a.shape[0] > 0 and np.isfinite(a).all() and np.array_equal(a, b)
    batch_normalised_check(pkg, orc, pcm[:16000], "one column batch", norm=norm, dyn=dyn, nad=nad, nb=nb, nc=nc)


@pytest.mark.parametrize("nb", [2, 3, 8, 20, 65])
def test_few_wide_filters_on_a_4096_point_transform(pkg, orc, nb):
    """with all 64 lanes' weight rows staged in LDS, 4096 points / 48 kHz / 20 filters (widest filter 594 bins,
    64 x 600 floats = 153 KB + the magnitude row) was refused with MFX_ERR_CONFIG although the reference accepts any bank
    count (mfcccpu.cpp:24-60).  Only the rows of lanes that carry a filter are staged now (mel64_rows): streaming
    interface, an alpha sweep whose warped banks are wider than the handle's own, and the batch entry against the oracle."""
    sr, W, S = 48000.0, 2400, 480
    nc = min(nb - 1, 12)
    pcm = synth_utterance(40000, 70 + nb, sr=sr)
    m, cfg, w = make_pair(pkg, orc, 30000, W=W, S=S, nb=nb, sr=sr, nc=nc, dyn=2)
    got = m.process_stream(pcm)
    want = orc.run_utterance(cfg, pcm, w)
    assert_close(got, want, "4096 points, %d filters (stream)" % nb, groups=3)
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), orc.run_utterance(cfg, pcm, w, bug_compat=False), "4096 points, %d filters (batch)" % nb,
                 groups=3)
    blk = m.get_input_buffer_size()      # (whole frames: 58 x 480 + 1920 = 29 760 samples)
    n = m.set_input(pcm[:blk])
    alphas = [0.85, 1.0, 1.15]
    m.apply_alphas(alphas)
    o = orc.OracleMfcc(cfg, w)
    assert o.set_input(pcm[:blk]) == n
    for i, a in enumerate(alphas):
        o.set_alpha(a)
        o.apply()
        assert_close(m.get_output_data_alpha(i, n), o.get_output_data(n), "4096 points, %d filters, alpha %.2f" % (nb, a), groups=3)


def test_real_handles_agree_with_the_shape_to_kernel_table(pkg):
    """Every row of mfcc.KERNEL_TABLE on a REAL handle: mfx_dominant_kernel_name after mfx_batch_plan (which derives the
    alignment from the caller's offsets) names the kernel the planning handle named (tests/test_host.py)."""
    for what, kw, want in pkg.KERNEL_TABLE:
        sr = kw["sample_rate"]
        m = pkg.MfccHip(100 * kw["shift"] + kw["window_size"], kw["window_size"], kw["shift"], kw["num_banks"], sr, 64.0, sr / 2,
                        kw["ceps_len"], kw.get("want_c0", False), 22.0, 0, kw.get("dyn", 0), 3, 3, True, device=0,
                        fft_size=kw.get("fft_size", 0), channels=kw.get("channels", 1), engine=kw.get("engine", 0))
        m.set_window(pkg.reference_window(kw["window_size"]))
        n = 40 * kw["shift"] + kw["window_size"]
        m.batch_plan([0 if kw.get("aligned", True) else 1], [n])
        assert m.dominant_kernel_name() == want, what
        m.close()


@pytest.mark.parametrize("W,S,nb,nc", [(91, 19, 32, 6), (122, 102, 15, 7), (62, 30, 16, 4), (44, 40, 12, 5)])
def test_zero_stuffed_forms_read_no_table_memory_they_do_not_own(pkg, orc, W, S, nb, nc):
    """Found by tools/fuzz_all.py (seed 3, case 17): at 128 and 64 points the zero-stuffed form of k_front512 staged 128 split
    twiddles from a table of W2 / 2 + 1 = 65 / 33 entries.  The split's difference term is rounding noise in that form, so
    whatever finite, small values lay behind the table did no harm -- fresh device memory is zero -- but stale memory with
    large values turned the noise into garbage (2.4 x the output scale).  Device memory is poisoned here (a 1 GiB tensor of
    1e30, released to the driver) before the handle allocates its tables; 128- and 64-point transforms, batch entry."""
    import torch
    x = torch.full((1 << 28,), 1e30, dtype=torch.float32, device="cuda:0")
    y = [torch.full((n,), 1e30, dtype=torch.float32, device="cuda:0") for n in (128, 256, 512, 1024, 4096) for _ in range(64)]
    torch.cuda.synchronize()
    del x, y
    torch.cuda.empty_cache()
    lens = [716, 547, 288, 575]
    offs = [2, 721, 1271, 1559]
    pcm = synth_utterance(2135, 17, sr=16000.0)
    m, cfg, w = make_pair(pkg, orc, 4000, W=W, S=S, nb=nb, sr=16000.0, nc=nc, dyn=0, bug_compat=False)
    assert m.dominant_kernel_name() == "k_front512" and m.fft_size() in (64, 128)
    m.set_alpha(0.95)
    rows, total = m.batch_plan(offs, lens)
    got = m.batch_run_host(pcm)
    for u, n in enumerate(lens):
        want = orc.run_utterance(cfg, pcm[offs[u]:offs[u] + n], w, alpha=0.95, bug_compat=False)
        assert_close(got[rows[u]:rows[u] + want.shape[0]], want, "W %d utt %d" % (W, u))


@pytest.mark.parametrize("engine", [0, 32])
def test_first_block_shorter_than_two_delta_contexts_is_refused(pkg, orc, engine):
    """DESIGN.md B13 (found by tools/fuzz_all.py as a host crash): a first block with D < frames < 2 D frames left the carried
    tail starting 30 samples BEFORE the staging buffer (engine 0: copy kernels through pinned staging) / before the device carry
    buffer (engine 32: DMA commands) -- as the reference does (segmentercpu.cpp:72-73, undefined there).  Refused now, with the
    reference's own guard message; the handle stays usable; blocks of 2 D frames and more stream as before."""
    W, S, l1, l2 = 62, 30, 3, 1
    D = l1 + l2
    pcm = synth_utterance(1295, 91, sr=8000.0)
    mk = lambda blk: make_pair(pkg, orc, blk, W=W, S=S, nb=16, sr=8000.0, nc=4, c0=True, dyn=2, l1=l1, l2=l2, engine=engine)
    m, cfg, w = mk(271)                       # 7 frames per block: 3 delivered, carry-over would start at sample -30
    assert m.get_input_buffer_size() == 7 * S + W - S
    with pytest.raises(pkg.MfxError, match="Processed samples"):
        m.set_input(pcm[:m.get_input_buffer_size()])
    with pytest.raises(RuntimeError):
        orc.OracleMfcc(cfg, w).set_input(pcm[:m.get_input_buffer_size()])
    m2, cfg2, _ = mk(2 * D * S + W - S + 5)   # 2 D frames per block: accepted
    got = m2.process_stream(pcm)
    want = orc.run_utterance(cfg2, pcm, w)
    assert_close(got, want, "blocks of exactly 2 D frames", groups=3)


@pytest.mark.parametrize("norm,dyn", [(0, 2), (2, 2), (0, 0)])
def test_small_block_copy_kernels_same_bits_as_dma(pkg, orc, norm, dyn):
    """Streaming interface, blocks under 1 MB: the block goes to the device, the carried tail to the other carry buffer and
    the rows back to the host through a copy KERNEL (pinned staging at the device address's alignment) instead of DMA
    commands (mfx_config.engine = MFX_ENGINE_DMA_SMALL_BLOCKS keeps those).  Odd block lengths put the appended block and
    the tail at every 2-byte alignment; the rows must be the same bits either way, from pageable and from pinned caller
    buffers, and equal to the oracle's."""
    import torch
    pcm = synth_utterance(61003, 77)
    kw = dict(norm=norm, dyn=dyn)
    m1, cfg, w = make_pair(pkg, orc, 20000, **kw)
    m2, _, _ = make_pair(pkg, orc, 20000, engine=pkg.mfcc.ENGINE_DMA_SMALL_BLOCKS, **kw)
    blocks = [7001, 9999, 4443, 12345, 8000, 11111, 8104]
    assert sum(blocks) == pcm.size

    def run(m, pinned):
        out, pos = [], 0
        for b in blocks:
            blk = pcm[pos:pos + b]
            if pinned:
                t = torch.empty(b, dtype=torch.int16).pin_memory()
                t.numpy()[:] = blk
                blk = t.numpy()
            n = m.set_input(blk)
            pos += b
            if n > 0:
                m.apply()
                out.append(m.get_output_data(n))
        n = m.flush()
        if n > 0:
            m.apply()
            out.append(m.get_output_data(n))
        return np.concatenate(out)

    a, b = run(m1, False), run(m2, False)
    assert a.shape == b.shape and np.isfinite(a).all() and np.array_equal(a, b)
    assert np.array_equal(run(m1, True), a)
    # rows straight into a PINNED caller buffer (the copy kernel writes it; deliberately at an odd float offset)
    import ctypes as C
    width = m1.get_output_data_width()
    t_out = torch.empty(200 * width + 3, dtype=torch.float32).pin_memory()
    got, pos = [], 0
    for bl in blocks + [0]:
        n = m1.set_input(pcm[pos:pos + bl]) if bl else m1.flush()
        pos += bl
        if n > 0:
            m1.apply()
            t_out.zero_()
            rc = m1._L.mfx_get_output_data(m1._h, C.cast(t_out.data_ptr() + 12, C.POINTER(C.c_float)), n)
            assert rc == 0
            got.append(t_out.numpy()[3:3 + n * width].reshape(n, width).copy())
    assert np.array_equal(np.concatenate(got), a)
    # ... and into a caller buffer page-locked after the fact (hipHostRegister): the kernel uses its device address
    reg = np.zeros(200 * width + 1024, np.float32)
    rt = torch.cuda.cudart()
    assert int(rt.cudaHostRegister(reg.ctypes.data, reg.nbytes, 0)) == 0
    try:
        m3, _, _ = make_pair(pkg, orc, 20000, **kw)
        got, pos = [], 0
        for bl in blocks + [0]:
            n = m3.set_input(pcm[pos:pos + bl]) if bl else m3.flush()
            pos += bl
            if n > 0:
                m3.apply()
                rc = m3._L.mfx_get_output_data(m3._h, C.cast(reg.ctypes.data + 4, C.POINTER(C.c_float)), n)
                assert rc == 0
                got.append(reg[1:1 + n * width].reshape(n, width).copy())
        assert np.array_equal(np.concatenate(got), a)
    finally:
        rt.cudaHostUnregister(reg.ctypes.data)
    if norm == 0:
        o = orc.OracleMfcc(cfg, w)
        rows, pos = [], 0
        for bl in blocks:
            n = o.set_input(pcm[pos:pos + bl])
            pos += bl
            if n > 0:
                o.apply()
                rows.append(o.get_output_data(n))
        n = o.flush()
        if n > 0:
            o.apply()
            rows.append(o.get_output_data(n))
        assert_close(a, np.concatenate(rows), "small blocks through the copy kernels", groups=1 + dyn)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_streaming_random_block_lengths_vs_oracle(pkg, orc, seed):
    """Random block lengths from 1 sample to the whole input buffer (blocks that add no frame, odd lengths, full blocks),
    delta context on and off, on a small-block handle (carried tail on the host, copy kernels) and on a large-block one
    (device-side tail): block by block the same row counts as the oracle and rows within the parity bar
    (segmentercpu.cpp:56-106 state machine, mfcccpu.cpp:371-444)."""
    rng = np.random.default_rng(seed)
    for ibs, dyn in ((20000, 2), (20000, 0), (700000, 2)):
        pcm = synth_utterance(int(3.2 * ibs) + int(rng.integers(0, 999)), 100 + seed)
        m, cfg, w = make_pair(pkg, orc, ibs, dyn=dyn)
        o = orc.OracleMfcc(cfg, w)
        lim = m.get_input_buffer_size()
        pos, first, k = 0, True, 0
        got, want = [], []
        while pos < pcm.size:
            hi = min(lim, pcm.size - pos)
            b = hi if first else int(rng.choice([1, 7, 159, 160, 161, 401, int(rng.integers(1, hi + 1)), hi]))
            b = max(1, min(b, hi))
            first = False
            n, no = m.set_input(pcm[pos:pos + b]), o.set_input(pcm[pos:pos + b])
            assert n == no, "block %d (%d samples at %d): %d rows, oracle %d" % (k, b, pos, n, no)
            pos += b
            k += 1
            if n > 0:
                m.apply()
                o.apply()
                got.append(m.get_output_data(n))
                want.append(o.get_output_data(n))
        n, no = m.flush(), o.flush()
        assert n == no
        if n > 0:
            m.apply()
            o.apply()
            got.append(m.get_output_data(n))
            want.append(o.get_output_data(n))
        assert_close(np.concatenate(got), np.concatenate(want), "random blocks ibs %d dyn %d" % (ibs, dyn), groups=1 + dyn)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with NO external launcher: the parent starts the two ranks itself (before it touches
    the GPU), relays rank 0's single JSON line and reports n_gpus == 2 (the reference's analogue is
    the sequential file queue, ASR_OCL.cpp:340-368)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MFX_BENCH_DEVICE="0", MFX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--settle-ms", "5", "--workload", "T", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["settle_ms"] >= 5
    assert abs(d["value"] - 2 * 8 * 98 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_bench_two_ranks_default_backend_is_gloo():
    """Round 4: the default carrier of the stopwatch's barrier for `bench.py --gpus N` is gloo (CPU scalars after
    torch.cuda.synchronize()): the data path has no collective (north_star: "no RCCL needed"; ASR_OCL.cpp:340-368), and an
    RCCL bring-up must not eat the driver's time limit.  No MFX_BENCH_BACKEND in the environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MFX_BENCH_BACKEND")}
    env.update(MFX_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--settle-ms", "5", "--workload", "T", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert abs(d["value"] - 2 * 8 * 98 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_bench_two_ranks_rccl_on_request_or_agreed_fallback():
    """`--collective rccl`: device tensors over RCCL, probed with a 60 s limit.  Two ranks pinned to ONE device: RCCL either
    comes up or refuses the duplicate device -- in that case every rank must agree (over gloo) to finish the stopwatch's
    barriers on CPU tensors, and the line must still be produced and say which backend carried them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MFX_BENCH_BACKEND")}
    env.update(MFX_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--settle-ms", "5", "--workload", "T", "--no-cpu-baseline", "--collective", "rccl"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] in ("rccl", "gloo")
    assert abs(d["value"] - 2 * 8 * 98 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_bench_self_launch_two_ranks_strong_scaling():
    """`bench.py --gpus 2 --scaling strong`: ONE job (workload T: 13 utterances) sharded round-robin over the two ranks
    (7 + 6 utterances), total frames / max-over-ranks time, `scaling: "strong"`, and the CPU baseline on rank 0 at N > 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MFX_BENCH_DEVICE="0", MFX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--settle-ms", "5", "--workload", "T", "--scaling", "strong"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    c = d["config"]
    assert c["utterances_total"] == 13 and c["utterances_rank0"] == 7 and c["utterance_ids_rank0_head"] == [0, 2, 4, 6]
    assert c["frames_per_step"] == 13 * 98 and c["frames_rank0_per_step"] == 7 * 98
    assert abs(d["value"] - 13 * 98 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


def test_bench_measures_hbm_traffic_in_the_run():
    """Round 4: the default line's roofline.traffic is MEASURED by bench.py itself -- two child runs of the same command under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` after the timed region -- not copied from a tracked file; the whole step's
    counter bytes stand beside its algorithmic bytes.  (Tiny workload here: the figures are launch overheads, the plumbing is
    what is checked.)"""
    import json
    import shutil
    import subprocess
    import sys
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MFX_BENCH_LIVE_TRAFFIC")}
    env["MFX_CPU_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--settle-ms", "5",
                        "--workload", "T"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rf = d["roofline"]
    assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    assert rf["traffic"] > 0 and set(rf["whole_path"]["traffic_kernels"]) == {"k_front512", "k_delta16"}
    assert rf["whole_path"]["traffic"] >= rf["traffic"] and rf["whole_path"]["traffic_over_algorithmic"] > 0.5
    assert d["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("norm,dyn", [(0, 2), (1, 0), (2, 2)])
def test_more_utterances_than_one_grid_dimension_holds(pkg, orc, norm, dyn):
    """66 000 utterances in one plan: the delta and normaliser launchers walk the segment list in pieces of 65 535 (grid.y);
    utterances on both sides of that seam, and the last one, against the checker (mfcccpu.cpp:234-282 per utterance)."""
    n_utt, n = 66000, 1360                      # 7 frames each
    rng = np.random.default_rng(66)
    pcm = (3000.0 * rng.standard_normal(n_utt * n)).astype(np.int16)
    m, cfg, w = make_pair(pkg, orc, n + 800, nb=26, norm=norm, dyn=dyn, l1=1, l2=1, nad=True, bug_compat=False)
    offs = np.arange(n_utt, dtype=np.int64) * n
    rows, total = m.batch_plan(offs, np.full(n_utt, n, dtype=np.int64))
    assert total == 7 * n_utt
    got = m.batch_run_host(pcm)
    g = groups_of(dyn)
    for u in (0, 1, 65534, 65535, 65536, 65537, n_utt - 1):
        seg = pcm[u * n:(u + 1) * n]
        if norm == 0:
            want = orc.run_utterance(cfg, seg, w, bug_compat=False)
        else:   # the utterance as ONE block, flush rows at their place (batch_norm_stats = 0: DESIGN.md B11)
            o = orc.OracleMfcc(cfg, w, bug_compat=False)
            k = o.set_input(seg)
            o.apply()
            parts = [o.get_output_data(k)]
            kf = o.flush()
            if kf > 0:
                o.apply()
                parts.append(o.get_output_data(kf))
            want = np.concatenate(parts)
            o.close()
        tol = dict() if norm == 0 else dict(tol_max=2e-4, tol_l2=2e-4)   # (7-row statistics: the exact criterion is the three-part check)
        assert_close(got[rows[u]:rows[u] + 7], want, "utterance %d of 66 000" % u, groups=g, **tol)


@pytest.mark.parametrize("tool,args", [("fuzz_all.py", ["1", "60"]), ("fuzz_all.py", ["2", "40", "wide"]), ("fuzz_api.py", ["1", "25"]),
                                       ("fuzz_calls.py", ["1", "25"])])
def test_differential_fuzzers_find_nothing(tool, args):
    """tools/fuzz_all.py: random shapes over every transform size and front-end kernel, batch entry + streaming interface against
    the checker (north-star bar, or 4 x the checker's own float32 noise where that is larger).  tools/fuzz_api.py: the library's
    own equivalence claims (engine bits that promise the same bits, apply_alphas == set_alpha + apply, handle reuse after flush,
    batch on the streaming kernels == streaming rows) and ragged / empty / replaced batch plans.  tools/fuzz_calls.py: random
    call sequences on the streaming interface, legal steps mirrored on the checker, illegal ones (no block, too many rows, NULL
    pointers, oversized blocks ...) thrown in between: status codes, no crash, state intact.  Round 4 found two bugs with
    them within minutes (the zero-stuffed forms' split table); a short run of each stays in the suite."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 failures" in r.stdout.splitlines()[-1]


def test_bench_reference_defaults_workload_runs_the_normaliser():
    """`--workload R` = the reference main()'s defaults (15 banks, 12 + c0, CVN): a bench line whose step contains the
    normaliser kernels (steps kept tiny here; the timed evidence lives in profiles/)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--settle-ms", "5",
                        "--workload", "R", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["frames_per_step"] == 998000 and "CVN" in d["config"]["workload"]
    assert d["roofline"]["algorithmic_bytes_per_frame"] == 320 + 52


# ---------------------------------------------------------------------------------------------
# randomized configuration sweep (seeded): window/shift/filterbank/cepstra/delta/normalisation shapes
# the fixed cases above do not hit -- odd shifts, windows that are not multiples of 32, FFT overrides,
# few or many filters, narrow bands, every delta width
# ---------------------------------------------------------------------------------------------

def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        fft = int(rng.choice([256, 512, 512, 512, 1024, 2048]))
        W = int(rng.integers(fft // 2 + 1, fft + 1))
        S = int(rng.integers(max(W // 8, 8), W // 2 + 1))
        sr = float(rng.choice([8000.0, 16000.0, 22050.0, 44100.0]))
        nb = int(rng.integers(8, 64))
        nc = int(rng.choice([0, 5, 12, 13, 20]))
        nc = min(nc, nb - 1)
        c0 = bool(rng.integers(0, 2)) and nc > 0
        dyn = int(rng.integers(0, 3))
        l1, l2 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        low = float(rng.choice([0.0, 64.0, 300.0]))
        high = float(sr / 2 * rng.choice([1.0, 0.9, 0.5]))
        norm = int(rng.choice([0, 0, 1, 2]))
        cases.append(dict(fft=fft, W=W, S=S, sr=sr, nb=nb, nc=nc, c0=c0, dyn=dyn, l1=l1, l2=l2, low=low, high=high,
                          norm=norm, seed=1000 + i))
    return cases


@pytest.mark.parametrize("case", _random_cases(24, 20260104), ids=lambda c: "fft%d_W%d_S%d_nb%d_nc%d_dyn%d_n%d" % (
    c["fft"], c["W"], c["S"], c["nb"], c["nc"], c["dyn"], c["norm"]))
def test_random_configuration(pkg, orc, case):
    c = case
    n = 40 * c["S"] + c["W"] + int(c["seed"] % 7) * 13
    pcm = synth_utterance(3 * n, c["seed"], sr=c["sr"])
    m, cfg, w = make_pair(pkg, orc, n, W=c["W"], S=c["S"], nb=c["nb"], sr=c["sr"], low=c["low"], high=c["high"],
                          nc=c["nc"], c0=c["c0"], norm=c["norm"], dyn=c["dyn"], l1=c["l1"], l2=c["l2"])
    assert m.fft_size() == c["fft"]
    g = groups_of(c["dyn"])
    if c["norm"]:
        # normalised: statistics, un-normalised twin and derived bound, block by block (conftest.py); the batch entry
        # against the oracle fed the utterance as one block
        kw = dict(W=c["W"], S=c["S"], nb=c["nb"], sr=c["sr"], low=c["low"], high=c["high"], nc=c["nc"], c0=c["c0"],
                  l1=c["l1"], l2=c["l2"])
        stream_normalised_check(pkg, orc, pcm, n, "stream", norm=c["norm"], dyn=c["dyn"], **kw)
        batch_normalised_check(pkg, orc, pcm, "batch", norm=c["norm"], dyn=c["dyn"], **kw)
        return
    assert_close(m.process_stream(pcm), orc.run_utterance(cfg, pcm, w), "stream", groups=g)
    m.batch_plan([0], [pcm.size])
    assert_close(m.batch_run_host(pcm), orc.run_utterance(cfg, pcm, w, bug_compat=False), "batch", groups=g)


@pytest.mark.parametrize("case", _random_cases(40, 77770002)[24:], ids=lambda c: "fft%d_W%d_S%d_nb%d_nc%d_dyn%d" % (
    c["fft"], c["W"], c["S"], c["nb"], c["nc"], c["dyn"]))
def test_random_ragged_batches(pkg, orc, case):
    """Random configurations (second seed) through the BATCH entry with several utterances of ragged lengths packed
    back to back at arbitrary (odd and even) offsets -- utterances that end exactly on the array's last sample, that
    hold exactly one frame, 2D frames, 2D + 1 frames: every utterance against the oracle fed that utterance alone."""
    c = case
    rng = np.random.default_rng(c["seed"])
    W, S = c["W"], c["S"]
    D = (c["l1"] + (c["l2"] if c["dyn"] == 2 else 0)) if c["dyn"] else 0
    frames = [1, 2 * D + 1, 2 * D + 2, 37, 16, 65, int(rng.integers(20, 120))]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    lens[-1] = (frames[-1] - 1) * S + W                    # the last one ends exactly on its last sample
    offs, pos = [], int(rng.integers(0, 3))
    for n in lens:
        offs.append(pos)
        pos += n + int(rng.integers(0, 4))
    pos = offs[-1] + lens[-1]                              # ... which is the array's last element
    pcm = np.zeros(pos, np.int16)
    utts = [synth_utterance(n, c["seed"] + 31 * i, sr=c["sr"]) for i, n in enumerate(lens)]
    for o_, u in zip(offs, utts):
        pcm[o_:o_ + u.size] = u
    m, cfg, w = make_pair(pkg, orc, max(lens) + 1000, W=W, S=S, nb=c["nb"], sr=c["sr"], low=c["low"], high=c["high"],
                          nc=c["nc"], c0=c["c0"], norm=0, dyn=c["dyn"], l1=c["l1"], l2=c["l2"])
    rows, total = m.batch_plan(offs, lens)
    got = m.batch_run_host(pcm)
    assert total == sum(frames) and got.shape[0] == total
    g = groups_of(c["dyn"])
    for i, (T, u) in enumerate(zip(frames, utts)):
        if T < 2 * D + 1 or T <= D:
            continue      # shorter than the reference's streaming protocol can express (test_c2_ragged_batch covers those)
        want = orc.run_utterance(cfg, u, w, bug_compat=False)
        assert_close(got[rows[i]:rows[i] + T], want, "utterance %d (%d frames)" % (i, T), groups=g)


@pytest.mark.parametrize("norm", [0, 2])
def test_batch_run_host_sliced_path_is_bit_identical(pkg, orc, norm):
    """mfx_batch_run_host with PINNED caller buffers runs the batch in slices (upload of slice k + 1 and download of
    slice k - 1 beside the kernels of slice k); with pageable buffers it runs it whole.  Same bits,
    ragged utterances with gaps between them, odd sample counts included."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(9)
    n_utt = 96
    lens = [int(v) for v in rng.integers(150000, 200000, size=n_utt)]
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos += n + int(rng.integers(0, 5))
    pcm = (3000.0 * rng.standard_normal(pos)).astype(np.int16)
    m, cfg, w = make_pair(pkg, orc, 210000, norm=norm)
    rows, total = m.batch_plan(offs, lens)
    plain = m.batch_run_host(pcm)                                 # pageable: one piece
    t_in = torch.from_numpy(pcm).pin_memory()
    t_out = torch.zeros((total, m.get_output_data_width()), dtype=torch.float32).pin_memory()
    rc = m._L.mfx_batch_run_host(m._h, C.cast(t_in.data_ptr(), C.POINTER(C.c_int16)), pos,
                                 C.cast(t_out.data_ptr(), C.POINTER(C.c_float)))
    assert rc == 0
    assert pos * 2 >= 32 << 20                                    # large enough for the sliced path
    assert np.array_equal(t_out.numpy(), plain)
    u = 57
    assert_close(plain[rows[u]:rows[u] + 100], orc.run_utterance(cfg, pcm[offs[u]:offs[u] + lens[u]], w,
                                                                 bug_compat=False)[:100], "utterance 57", groups=3) if norm == 0 else None


def test_last_frame_ends_on_an_odd_last_sample(pkg, orc):
    """An utterance with an ODD number of samples whose last frame ends exactly on the last sample, at the very end of
    the PCM array: the last sample lies in a 32-bit word that is half past the end.  (Found by the widened random
    sweep in round 2: the buffer range check dropped that word and the last frame lost its last sample.)"""
    W, S = 489, 132
    n = 128 * S + W                                   # 129 frames, the last one ends on sample n - 1; n is odd
    assert n % 2 == 1
    pcm = synth_utterance(n, 1003, sr=22050.0)
    m, cfg, w = make_pair(pkg, orc, n + 1000, W=W, S=S, nb=40, sr=22050.0, nc=13, dyn=0)
    m.batch_plan([0], [n])
    got = m.batch_run_host(pcm)
    assert got.shape == (129, 13)
    assert_close(got, orc.run_utterance(cfg, pcm, w, bug_compat=False), "odd tail")
    # odd offset as well (unaligned load path): the same utterance one sample into the array
    pcm1 = np.concatenate([np.zeros(1, np.int16), pcm])
    m.batch_plan([1], [n])
    assert_close(m.batch_run_host(pcm1), got, "odd tail, odd offset", tol_max=1e-6, tol_l2=1e-6)


def test_batch_overlap_mode_is_bit_identical(pkg, orc):
    """mfx_batch_overlap: the delta tail of batch i runs beside the front end of batch i+1 (second stream,
    double-buffered statics).  Back-to-back batches on different inputs must give exactly the results of
    the strictly ordered mode."""
    import torch
    dev = torch.device("cuda", 0)
    n_utt, n = 64, 48000
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    pcms = [(4000.0 * torch.randn((n_utt, n), generator=g, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
            for _ in range(5)]
    m, cfg, w = make_pair(pkg, orc, n + 1000)
    rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
    ref = []
    for x in pcms:
        o = torch.empty((total, 39), dtype=torch.float32, device=dev)
        m.batch_run_device(x.data_ptr(), x.numel(), o.data_ptr())
        m.synchronize()
        ref.append(o)
    m.batch_overlap(True)
    outs = [torch.full((total, 39), float("nan"), dtype=torch.float32, device=dev) for _ in pcms]
    for x, o in zip(pcms, outs):          # no synchronisation between batches
        m.batch_run_device(x.data_ptr(), x.numel(), o.data_ptr())
    m.synchronize()
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)
    m.batch_overlap(False)
    o = torch.empty((total, 39), dtype=torch.float32, device=dev)
    m.batch_run_device(pcms[0].data_ptr(), pcms[0].numel(), o.data_ptr())
    m.synchronize()
    assert torch.equal(o, ref[0])
    want = orc.run_utterance(cfg, pcms[2][5].cpu().numpy(), w, bug_compat=False)
    assert_close(outs[2][rows[5]:rows[5] + want.shape[0]].cpu().numpy(), want, "overlap mode vs oracle", groups=3)


# ---------------------------------------------------------------------------------------------
# fused delta stage of the 512-point kernel (one wave per block consumes the other 15 waves' statics)
# ---------------------------------------------------------------------------------------------

def _fuse_pair(pkg, orc, monkeypatch, **kw):
    """Two extractors with identical parameters: fused delta stage on (opt-in, mfx_config.engine = MFX_ENGINE_FUSE_DELTA)
    and off."""
    m_f, cfg, w = make_pair(pkg, orc, 200000, engine=pkg.mfcc.ENGINE_FUSE_DELTA, **kw)
    m_u, _, _ = make_pair(pkg, orc, 200000, **kw)
    return m_f, m_u, cfg, w


@pytest.mark.parametrize("dyn,l1,l2,norm", [(2, 3, 3, 0), (2, 2, 5, 0), (1, 4, 0, 0), (2, 1, 1, 2), (2, 8, 8, 0)])
def test_fused_delta_bit_identical_to_separate_kernel(pkg, orc, monkeypatch, dyn, l1, l2, norm):
    """The fused path must give the SAME BITS as front end + k_delta (same arithmetic, same order), over
    ragged utterances: block pieces that start/end inside an utterance (halo chunks), utterances shorter than
    the delta context, empty ones, and lengths that leave 1..3 live frames in the last iteration."""
    lens = [160000, 5000, 400, 720, 1040, 1200, 0, 48000, 2000, 12346, 30000, 1360, 100000, 880]
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos += n + (n & 1)
    pcm = np.zeros(pos + 8, np.int16)
    for u, (o_, n) in enumerate(zip(offs, lens)):
        pcm[o_:o_ + n] = synth_utterance(n, 100 + u)
    m_f, m_u, cfg, w = _fuse_pair(pkg, orc, monkeypatch, dyn=dyn, l1=l1, l2=l2, norm=norm)
    rows_f, total_f = m_f.batch_plan(offs, lens)
    rows_u, total_u = m_u.batch_plan(offs, lens)
    assert total_f == total_u and list(rows_f) == list(rows_u)
    got_f = m_f.batch_run_host(pcm)
    got_u = m_u.batch_run_host(pcm)
    # (CVN of a 1-frame utterance is 0/0 in the reference too: NaN on both sides)
    assert got_f.shape == got_u.shape and np.array_equal(got_f, got_u, equal_nan=True)
    # and both agree with the oracle on a long and a short utterance
    # (not with normalisation: the batch normalises per utterance, the streaming oracle per block)
    for u in (0, 9, 12) if norm == 0 else ():
        T = max((lens[u] - 240) // 160, 0)
        want = orc.run_utterance(cfg, pcm[offs[u]:offs[u] + lens[u]], w, bug_compat=False)
        assert_close(got_f[rows_f[u]:rows_f[u] + T], want, "fused delta utt %d" % u, groups=groups_of(dyn))


def test_fused_delta_many_blocks_full_shape(pkg, orc, monkeypatch):
    """C2-shaped batch big enough to use every CU (one block per CU, ~60 tiles per block): fused and
    separate paths bit-identical, repeated launches stable."""
    n_utt, n = 300, 160000
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal(n_utt * n) * 3000).astype(np.int16)
    offs = [u * n for u in range(n_utt)]
    m_f, m_u, cfg, w = _fuse_pair(pkg, orc, monkeypatch)
    m_f.batch_plan(offs, [n] * n_utt)
    m_u.batch_plan(offs, [n] * n_utt)
    a = m_f.batch_run_host(pcm)
    b = m_u.batch_run_host(pcm)
    assert a.shape == (n_utt * 998, 39) and np.array_equal(a, b)
    assert np.array_equal(m_f.batch_run_host(pcm), a)
    want = orc.run_utterance(cfg, pcm[7 * n:8 * n], w, bug_compat=False)
    assert_close(a[7 * 998:8 * 998], want, "utt 7", groups=3)


@pytest.mark.parametrize("fft", [0, 1024])
def test_batch_replan_and_alpha_change_between_runs(pkg, orc, fft):
    """One handle, several batches: a new plan (other lengths, fewer utterances) and a new VTLN factor between
    runs must rebuild whatever depends on them (mel tables, lane / work plans, LDS sizes) -- 512-point register
    kernel and the long-transform kernel."""
    kw = dict(nb=40) if fft == 0 else dict(nb=80, W=1024, S=256)   # (a full-length window: same frame count as the oracle)
    m, cfg, w = make_pair(pkg, orc, 200000, dyn=2, **kw)
    assert m.fft_size() == (fft or 512)

    def check(lens, alpha, tag):
        offs, pos = [], 0
        for n in lens:
            offs.append(pos)
            pos += n + (n & 1)
        pcm = np.zeros(pos + 8, np.int16)
        for u, (o_, n) in enumerate(zip(offs, lens)):
            pcm[o_:o_ + n] = synth_utterance(n, 300 + u)
        m.set_alpha(alpha)
        rows, total = m.batch_plan(offs, lens)
        got = m.batch_run_host(pcm)
        for u, n in enumerate(lens):
            want = orc.run_utterance(cfg, pcm[offs[u]:offs[u] + n], w, alpha=alpha, bug_compat=False)
            T = want.shape[0]
            assert_close(got[rows[u]:rows[u] + T], want, "%s utt %d" % (tag, u), groups=3)

    check([48000, 16000, 30000], 1.0, "first plan")
    check([20000, 64000], 0.9, "second plan, alpha 0.9")
    check([48000, 16000, 30000, 8000], 1.1, "third plan, alpha 1.1")
    check([48000], 1.0, "back to alpha 1")


# ---------------------------------------------------------------------------------------------
# The HIP path against the REAL reference (mfcccpu.cpp compiled in place, see tests/test_ref_mfcccpu.py):
# committed vectors of oracle/_ref/libref_mfcccpu.so for every case of tests/refcases.py; live as well when the
# library travelled with the snapshot.
# ---------------------------------------------------------------------------------------------
import refcases as RC  # noqa: E402

_REFCASES = RC.cases()
_REFDIR = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref")
# Two builds of the reference (tests/test_ref_mfcccpu.py): "f32" = its unqualified libm names bound to the float overloads,
# as its own toolchain binds them -- the checker's DEFAULT arithmetic is bit-identical to it; "gpp" = plain g++ (C double
# functions, int abs), the checker under libm_double.
_REF_BUILDS = {"f32": dict(file="ref_mfcccpu_vectors_f32.npz", libm_double=False, lib="libref_mfcccpu_f32.so"),
               "gpp": dict(file="ref_mfcccpu_vectors.npz", libm_double=True, lib="libref_mfcccpu.so")}
_REF_LIVE = all(os.path.exists(os.path.join(_REFDIR, v["lib"])) for v in _REF_BUILDS.values())


@pytest.fixture(scope="module")
def refvecs():
    return {b: np.load(os.path.join(GOLDEN, v["file"])) for b, v in _REF_BUILDS.items()}


def _scale_floor(name):
    # silence: all 40 log energies are log(1e-30); their DCT cancels to rounding noise (~1e-5), so the yardstick is the
    # size of the DCT's input, |log(1e-30)| = 69.08
    return 69.08 if name == "silence" else 0.0


def _hip_for_case(pkg, c, norm=None, bug_compat=True):
    k = c["cfg"]
    sr = k["sample_rate"]
    high = sr / 2 if k["high_freq"] is None else k["high_freq"]
    m = pkg.MfccHip(c["ibs"], k["window_size"], k["shift"], k["num_banks"], sr, k["low_freq"], high, k["ceps_len"],
                    k["want_c0"], k["lift_coef"], k["norm"] if norm is None else norm, k["dyn"], k["delta_l1"],
                    k["delta_l2"], k["norm_after_dyn"], device=0, bug_compat=bug_compat)
    return m


def _stream_cases():
    """Every case against the float-bound build (MINMAX values included); every case but MINMAX against the plain g++
    build as well -- that build divides by int(|extreme - mean|) (abs binds to int abs(int) under g++), an
    artefact of a toolchain the reference was not written for, reproduced bit for bit by the checker's libm_double mode
    on the CPU (tests/test_ref_mfcccpu.py) and not by the product."""
    out = [(n, "f32") for n in sorted(_REFCASES)]
    out += [(n, "gpp") for n in sorted(_REFCASES) if _REFCASES[n]["cfg"]["norm"] != RC.NORM_MINMAX]
    return out


@pytest.mark.parametrize("name,build", _stream_cases())
def test_hip_streaming_vs_real_reference_vectors(pkg, orc, refvecs, name, build):
    """set_window -> {set_input -> set_alpha -> apply -> get_output_data}* -> flush -> ... (ASR_OCL.cpp:227-301) through the
    C ABI, block by block, against what the reference's own MfccCpu functions returned for the same calls.
    Un-normalised cases: every block at the north-star bar (1e-4 of scale, 1e-5 rel-L2) against the committed rows.
    CMN / CVN / MINMAX: the three-part check of conftest.py against the committed rows, with the checker under the build's
    binding -- which this test first shows to be bit-identical to the committed reference rows, block by block --
    supplying the statistics and the un-normalised twin."""
    from conftest import assert_normalised_close
    c = _REFCASES[name]
    k = c["cfg"]
    dbl = _REF_BUILDS[build]["libm_double"]
    pcm, w = RC.load_pcm(c["pcm"]), RC.case_window(orc, c)
    want_rows, want_counts = refvecs[build][name + "/rows"], refvecs[build][name + "/counts"]
    g = groups_of(k["dyn"])
    m = _hip_for_case(pkg, c)
    m.set_window(w)
    normed = k["norm"] != 0
    if normed:
        m0 = _hip_for_case(pkg, c, norm=0)
        m0.set_window(w)
        case0 = dict(c, cfg=dict(k, norm=0))
        o = orc.OracleMfcc(RC.make_cfg(orc, c), w, libm_double=dbl)
        o0 = orc.OracleMfcc(RC.make_cfg(orc, case0), w, libm_double=dbl)
    engines = [m] + ([m0, o, o0] if normed else [])
    blk, pos, row, counts = m.get_input_buffer_size(), 0, 0, []
    while True:
        last = pos >= pcm.size
        ns = [e.flush() if last else e.set_input(pcm[pos:pos + blk]) for e in engines]
        pos += blk
        n = ns[0]
        assert all(x == n for x in ns)
        counts.append(n)
        if n > 0:
            for e in engines:
                e.set_alpha(c["alpha"])
                e.apply()
            y = m.get_output_data(n)
            want = want_rows[row:row + n]
            what = "%s [%s] block %d" % (name, build, len(counts) - 1)
            if not normed:
                assert_close(y, want, what, groups=g, scale_floor=_scale_floor(name))
            else:
                yo = o.get_output_data(n)
                assert np.array_equal(yo, want, equal_nan=True), what + ": checker != committed reference rows"
                cols = y.shape[1] // g
                st = m.debug_read(5).reshape(-1, 2, cols)
                assert_normalised_close(y, want, m0.get_output_data(n), o0.get_output_data(n), st, o.norm_stats(), g,
                                        k["norm_after_dyn"], what, norm=k["norm"])
            row += n
        if last:
            break
    assert np.array_equal(np.array(counts), want_counts)
    assert row == want_rows.shape[0]


@pytest.mark.parametrize("build", sorted(_REF_BUILDS))
@pytest.mark.parametrize("name", ["c1_multi", "c2_alpha088", "c2_alpha100", "c2_alpha112", "c3_alpha100", "c3_alpha112",
                                  "c3_streamed_dyn", "c5_alpha088", "c5_alpha100", "mel_only", "odd_geometry", "silence"])
def test_hip_batch_entry_vs_real_reference_vectors(pkg, orc, refvecs, name, build):
    """The batch entry (mfx_batch_plan + mfx_batch_run_host: the fused kernels the benchmark times) on the same inputs:
    whole-utterance rows against the reference's multi-block rows (block size does not change an un-normalised result,
    test_streaming_block_size_invariance; c3_alpha* are single blocks without deltas, where B1 cannot occur)."""
    c = _REFCASES[name]
    pcm, w = RC.load_pcm(c["pcm"]), RC.case_window(orc, c)
    m = _hip_for_case(pkg, dict(c, ibs=pcm.size + 1000))
    m.set_window(w)
    m.set_alpha(c["alpha"])
    m.batch_plan([0], [pcm.size])
    got = m.batch_run_host(pcm)
    assert_close(got, refvecs[build][name + "/rows"], "%s [%s] batch entry" % (name, build),
                 groups=groups_of(c["cfg"]["dyn"]), scale_floor=_scale_floor(name))


@pytest.mark.skipif(not _REF_LIVE, reason="oracle/_ref/libref_mfcccpu{,_f32}.so did not travel")
@pytest.mark.parametrize("build", sorted(_REF_BUILDS))
def test_hip_vs_live_real_reference_on_fresh_inputs(pkg, orc, build):
    """The reference's compiled MfccCpu functions, loaded on the GPU box, against the HIP path on inputs no fixture holds:
    C2-, C3- and C5-shaped utterances with new seeds and warps, streamed in uneven blocks; a MINMAX configuration against
    the float-bound build (block-level statistics: compared through the checker-free bound max|dy| <= 2e-4, the extreme's
    own rounding: y = (x - mean) / max|x - mean| lies in [-1, 1])."""
    f32 = build == "f32"
    for (tag, seed, alpha, ibs) in (("c2_alpha100", 501, 0.91, 11111), ("c3_streamed_dyn", 502, 1.09, 9500),
                                    ("c5_alpha100", 503, 1.04, 17001), ("odd_geometry", 504, 0.97, 5003),
                                    ("dyn0_norm3_nad1", 505, 1.0, 5000)):
        base = _REFCASES[tag]
        k = base["cfg"]
        if k["norm"] == RC.NORM_MINMAX and not f32:
            continue
        c = dict(base, ibs=ibs, alpha=alpha, pcm=("synth", base["pcm"][1] + 7001, seed, k["sample_rate"]))
        pcm, w = RC.load_pcm(c["pcm"]), RC.case_window(orc, c)
        r = orc.RefMfccCpu(RC.make_cfg(orc, c), w, f32=f32)
        want, want_counts = RC.drive(r, pcm, alpha)
        r.close()
        m = _hip_for_case(pkg, c)
        m.set_window(w)
        got = m.process_stream(pcm, alpha=alpha)
        assert got.shape == want.shape, tag
        if k["norm"] == RC.NORM_MINMAX:
            assert np.abs(got - want).max() <= 2e-4, "%s live MINMAX: %g" % (tag, np.abs(got - want).max())
        else:
            assert_close(got, want, "%s live reference [%s]" % (tag, build), groups=groups_of(k["dyn"]))


def test_sliced_batch_then_large_streaming_download_on_one_handle(pkg, orc):
    """the staging re-allocation of a >= 4 MiB get_output_data once tore down the sliced batch path's streams and
    events.  One handle: pinned sliced mfx_batch_run_host -> a streaming block whose rows (> 4 MiB) come back into a pageable
    numpy array -> the pinned sliced batch again (same bits) -> a second large download -> close.  Also: a plan whose last
    utterance lies past the array is refused before any copy is queued."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(10)
    n_utt, L = 64, 300000
    pcm = (3000.0 * rng.standard_normal(n_utt * L)).astype(np.int16)
    ibs = 5000000
    m, cfg, w = make_pair(pkg, orc, ibs)
    rows, total = m.batch_plan([u * L for u in range(n_utt)], [L] * n_utt)
    t_in = torch.from_numpy(pcm).pin_memory()
    t_out = torch.zeros((total, m.get_output_data_width()), dtype=torch.float32).pin_memory()
    run = lambda: m._L.mfx_batch_run_host(m._h, C.cast(t_in.data_ptr(), C.POINTER(C.c_int16)), pcm.size,
                                          C.cast(t_out.data_ptr(), C.POINTER(C.c_float)))
    assert pcm.size * 2 >= 32 << 20
    assert run() == 0
    first = t_out.numpy().copy()
    for _ in range(2):
        n = m.set_input(pcm[:ibs - 1000])                     # ~31 k frames x 39 floats = 4.9 MB of rows
        m.apply()
        big = m.get_output_data(n)                            # pageable destination >= 4 MiB: staged, chunked download
        assert big.nbytes >= 4 << 20 and np.isfinite(big).all()
        m.flush()
        t_out.zero_()
        assert run() == 0
        assert np.array_equal(t_out.numpy(), first)
    u = 40
    assert_close(first[rows[u]:rows[u] + 200], orc.run_utterance(cfg, pcm[u * L:(u + 1) * L], w, bug_compat=False)[:200],
                 "utterance 40", groups=3)
    # a plan that points past the array: error, nothing written, handle still usable
    m.batch_plan([u * L for u in range(n_utt)], [L] * (n_utt - 1) + [L + 64])
    t_out.zero_()
    assert run() != 0
    m.batch_plan([u * L for u in range(n_utt)], [L] * n_utt)
    assert run() == 0 and np.array_equal(t_out.numpy(), first)
    m.close()


# ---------------------------------------------------------------------------------------------
# k_front2048: 2048-point transforms of a short window, two frames per wave (BASELINE configs[4])
# ---------------------------------------------------------------------------------------------

_F2048 = [  # W, S, sr, nb, nc, c0, dyn, channels, alpha
    (1102, 441, 44100.0, 128, 40, False, 2, 2, 1.0),     # configs[4] itself: stereo, odd shift
    (1102, 441, 44100.0, 128, 40, False, 2, 1, 1.0),     # mono at an odd shift (10 ms at 44.1 kHz): the any-alignment build
    (1102, 440, 44100.0, 128, 40, False, 2, 1, 1.0),     # mono, aligned pairs
    (1152, 400, 48000.0, 96, 24, True, 1, 2, 0.93),      # the longest window of the 18-row build, c0, VTLN
    (1025, 512, 44100.0, 40, 13, False, 0, 2, 1.0),      # odd window length, few filters, one DCT tile
    (1100, 300, 32000.0, 200, 60, False, 2, 1, 1.07),    # 7 rounds of 32 filters, 4 DCT tiles; fewer waves per block (LDS)
    (1050, 350, 44100.0, 64, 0, False, 1, 2, 1.0),       # log mel energies as the features (no DCT)
    (1102, 441, 44100.0, 100, 70, True, 0, 2, 1.0),      # 71 columns: two 64-column passes of the matrix-pipe DCT
    (1102, 441, 44100.0, 128, 25, True, 2, 2, 1.0),      # split DCT, pass A alone: 26 columns on 2 band halves
    (1102, 440, 44100.0, 64, 39, True, 1, 1, 0.9),       # split DCT A + B at 64 bands (8 / 2 K-groups: the runtime-loop form), mono
    (1100, 320, 32000.0, 256, 33, False, 0, 2, 1.0),     # split DCT A + B at 256 bands, 33 columns: one column in pass B
    (1102, 441, 44100.0, 96, 32, False, 2, 2, 1.1),      # split DCT, pass A alone, exactly 32 columns, 96 bands
    (1200, 480, 48000.0, 128, 40, False, 2, 2, 1.0),     # 25 ms at 48 kHz: the 20-row build (W <= 1280), stereo, split DCT
    (1280, 480, 48000.0, 80, 13, False, 1, 1, 1.05),     # the longest window of the 20-row build, mono, one-tile DCT
    (1153, 577, 48000.0, 40, 20, True, 0, 2, 1.0),       # one tap past the 18-row build, odd shift, stereo
    (2048, 512, 44100.0, 128, 40, False, 2, 1, 1.0),     # n_fft = win_length = 2048, hop 512 (the audio-analysis default): the 32-row build, mono
    (2048, 441, 44100.0, 80, 0, False, 0, 2, 1.0),       # full window, stereo, 80 log mel energies
    (1411, 441, 44100.0, 64, 20, True, 1, 1, 0.95),      # 32 ms at 44.1 kHz, mono at the odd shift (any-alignment build), VTLN
]


@pytest.mark.parametrize("W,S,sr,nb,nc,c0,dyn,ch,alpha", _F2048)
def test_front2048_configurations(pkg, orc, W, S, sr, nb, nc, c0, dyn, ch, alpha):
    """Ragged utterances (1, 2, 3, 15, 16, 17, 33, 70 frames: odd counts leave the wave's upper half idle, 1-frame chunks,
    chunk ends off the 4-frame DCT groups) at arbitrary offsets through the batch entry against the oracle fed each
    utterance alone, and the same batch through k_front_reg (mfx_config.engine = MFX_ENGINE_NO_FRONT2048: the 16.16.4
    factorisation of the same transform)."""
    rng = np.random.default_rng(W + S + nb)
    frames = [1, 2, 3, 15, 16, 17, 33, 70]
    step = 2 if (ch == 1 and S % 2 == 0 and W % 2 == 0 and nb != 80) else 1   # mono pairs: even offsets keep the aligned build (the nb = 80 case: odd offsets at an even shift)
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos += n + step * int(rng.integers(0, 3))
        pos += pos % step
    mono = np.zeros(pos, np.int16)
    if ch == 2:
        left = np.zeros(pos, np.int16)
        right = np.zeros(pos, np.int16)
    for i, (o_, n) in enumerate(zip(offs, lens)):
        if ch == 2:
            left[o_:o_ + n] = synth_utterance(n, 700 + 2 * i, sr=sr)
            right[o_:o_ + n] = synth_utterance(n, 701 + 2 * i, sr=sr, f=311.0 + 40 * i)
        else:
            mono[o_:o_ + n] = synth_utterance(n, 700 + i, sr=sr)
    if ch == 2:
        mono = ((left.astype(np.int32) + right.astype(np.int32)) >> 1).astype(np.int16)
        pcm = np.empty(2 * pos, np.int16)
        pcm[0::2], pcm[1::2] = left, right
    else:
        pcm = mono
    kw = dict(W=W, S=S, nb=nb, sr=sr, nc=nc, c0=c0, dyn=dyn, l1=2, l2=2, channels=ch)
    m, cfg, w = make_pair(pkg, orc, max(lens) + 2000, **kw)
    assert m.fft_size() == 2048
    if alpha != 1.0:
        m.set_alpha(alpha)
    rows, total = m.batch_plan(offs, lens)
    assert m.dominant_kernel_name() == "k_front2048"
    got = m.batch_run_host(pcm)
    assert total == sum(frames) and got.shape[0] == total
    g = groups_of(dyn)
    m2, _, _ = make_pair(pkg, orc, max(lens) + 2000, engine=pkg.mfcc.ENGINE_NO_FRONT2048, **kw)
    assert m2.dominant_kernel_name() == "k_front_reg"
    if alpha != 1.0:
        m2.set_alpha(alpha)
    m2.batch_plan(offs, lens)
    assert_close(got, m2.batch_run_host(pcm), "k_front2048 vs k_front_reg", groups=g)
    for i, (o_, n, T) in enumerate(zip(offs, lens, frames)):
        o = orc.OracleMfcc(cfg, w, bug_compat=False)
        o.set_alpha(alpha)
        D = (2 + (2 if dyn == 2 else 0)) if dyn else 0
        if T <= 2 * D:
            continue   # files of fewer than 2 D frames: the streaming reference refuses them or leaves its frame grid
        want = orc.run_utterance(cfg, mono[o_:o_ + n], w, alpha=alpha, bug_compat=False)
        assert want.shape[0] == T
        assert_close(got[rows[i]:rows[i] + T], want, "utterance %d (%d frames)" % (i, T), groups=g)
    # the same batch with the DCT as one 64-column tile (mfx_config.engine = MFX_ENGINE_NO_DCT_SPLIT): float32 rounding apart
    m3, _, _ = make_pair(pkg, orc, max(lens) + 2000, engine=pkg.mfcc.ENGINE_NO_DCT_SPLIT, **kw)
    if alpha != 1.0:
        m3.set_alpha(alpha)
    m3.batch_plan(offs, lens)
    assert_close(got, m3.batch_run_host(pcm), "k_front2048 split DCT vs one tile", groups=g)


_F256 = [  # W, S, sr, nb, nc, c0, dyn, alpha
    (200, 80, 8000.0, 23, 13, False, 2, 1.0),      # 8 kHz telephony: 25 ms / 10 ms, 13 rows of 16 samples
    (200, 81, 8000.0, 23, 12, True, 2, 1.0),       # odd shift: frames at odd sample offsets (16-bit loads: any alignment)
    (256, 100, 8000.0, 40, 13, False, 1, 0.92),    # the longest window of a 256-point transform: 16 rows; VTLN
    (129, 64, 8000.0, 15, 0, False, 0, 1.0),       # the shortest window that still takes 256 points; log mel energies
    (220, 110, 11025.0, 26, 20, False, 2, 1.1),    # 20 columns: the LDS mat-vec DCT of the 512-point kernel (cols > 16)
    (128, 64, 8000.0, 20, 12, False, 2, 1.0),      # 128 points (16 ms at 8 kHz): a zero after every sample twice over
    (100, 33, 8000.0, 12, 8, True, 1, 0.95),       # 128 points, odd shift, VTLN
    (64, 32, 8000.0, 10, 9, False, 0, 1.0),        # 64 points: three times over (every fourth lane carries a sample)
]


@pytest.mark.parametrize("W,S,sr,nb,nc,c0,dyn,alpha", _F256)
def test_front256_zero_stuffed_on_the_512_point_kernel(pkg, orc, W, S, sr, nb, nc, c0, dyn, alpha):
    """256-point transforms run on k_front512 in its zero-stuffed form (the 512-point real DFT of x[0], 0, x[1], 0, ... is
    X_256[k mod 256]).  Ragged utterances at odd and even offsets through the batch entry against the oracle fed each
    utterance alone (the reference's 256-point path: mfcccpu.cpp:187-220), against the one-wave-per-frame kernel
    (mfx_config.engine = MFX_ENGINE_NO_STUFF256), and the streaming interface (spectrum through HBM) block by block."""
    rng = np.random.default_rng(W + S + nb)
    frames = [1, 2, 3, 4, 5, 17, 64, 131]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    offs, pos = [], int(rng.integers(0, 3))
    for n in lens:
        offs.append(pos)
        pos += n + int(rng.integers(0, 4))
    pcm = np.zeros(pos, np.int16)
    for i, (o_, n) in enumerate(zip(offs, lens)):
        pcm[o_:o_ + n] = synth_utterance(n, 900 + i, sr=sr)
    kw = dict(W=W, S=S, nb=nb, sr=sr, nc=nc, c0=c0, dyn=dyn, l1=2, l2=2)
    m, cfg, w = make_pair(pkg, orc, max(lens) + 2000, **kw)
    assert m.fft_size() in (256, 128, 64) and m.dominant_kernel_name() == "k_front512"
    if alpha != 1.0:
        m.set_alpha(alpha)
    rows, total = m.batch_plan(offs, lens)
    got = m.batch_run_host(pcm)
    assert total == sum(frames) and got.shape[0] == total and np.isfinite(got).all()
    g = groups_of(dyn)
    m2, _, _ = make_pair(pkg, orc, max(lens) + 2000, engine=pkg.mfcc.ENGINE_NO_STUFF256, **kw)
    assert m2.dominant_kernel_name() == "k_front_wave"
    if alpha != 1.0:
        m2.set_alpha(alpha)
    m2.batch_plan(offs, lens)
    assert_close(got, m2.batch_run_host(pcm), "zero-stuffed k_front512 vs k_front_wave", groups=g)
    D = (2 + (2 if dyn == 2 else 0)) if dyn else 0
    for i, (o_, n, T) in enumerate(zip(offs, lens, frames)):
        if T <= 2 * D:
            continue   # files of fewer than 2 D frames: the streaming reference refuses them or leaves its frame grid
        want = orc.run_utterance(cfg, pcm[o_:o_ + n], w, alpha=alpha, bug_compat=False)
        assert want.shape[0] == T
        assert_close(got[rows[i]:rows[i] + T], want, "utterance %d (%d frames)" % (i, T), groups=g)
    # streaming interface on the longest utterance, odd block lengths
    u = int(np.argmax(lens))
    x = pcm[offs[u]:offs[u] + lens[u]]
    ms, cfg_s, w_s = make_pair(pkg, orc, 4000, **kw)
    o = orc.OracleMfcc(cfg_s, w_s)
    if alpha != 1.0:
        ms.set_alpha(alpha)
        o.set_alpha(alpha)
    a_rows, b_rows, p0 = [], [], 0
    lim = ms.get_input_buffer_size()
    while p0 < x.size:
        b = min(lim - (p0 % 7), x.size - p0)
        n, no = ms.set_input(x[p0:p0 + b]), o.set_input(x[p0:p0 + b])
        assert n == no
        p0 += b
        if n > 0:
            ms.apply(); o.apply()
            a_rows.append(ms.get_output_data(n)); b_rows.append(o.get_output_data(n))
    n, no = ms.flush(), o.flush()
    assert n == no
    if n > 0:
        ms.apply(); o.apply()
        a_rows.append(ms.get_output_data(n)); b_rows.append(o.get_output_data(n))
    assert_close(np.concatenate(a_rows), np.concatenate(b_rows), "streaming, 256 points", groups=g)


@pytest.mark.parametrize("W,S,sr,nb,nc,dyn", [
    (400, 160, 16000.0, 40, 13, 2),     # 16 kHz stereo: the 512-point kernel, one 8-byte load per sample pair
    (400, 161, 16000.0, 26, 13, 0),     # odd shift (words are per sample: no alignment cases)
    (512, 200, 16000.0, 40, 20, 1),     # the longest window, 16 rows, 20 columns
    (200, 80, 8000.0, 23, 13, 2),       # 8 kHz stereo (two-channel call recordings): zero-stuffed, one word per lane and row
    (256, 99, 8000.0, 15, 12, 0),
])
def test_front512_stereo_batches(pkg, orc, W, S, sr, nb, nc, dyn):
    """Interleaved stereo through the batch entry on k_front512 (512 points, and 256 points zero-stuffed): mono =
    (L + R) >> 1 in the kernel.  Ragged utterances at arbitrary sample offsets against the oracle fed the host-side
    downmix, and the same extractor fed that downmix as mono input (float32 rounding apart: other load paths)."""
    rng = np.random.default_rng(W + S + nb)
    frames = [1, 2, 5, 9, 33, 100]
    lens = [(T - 1) * S + W + int(rng.integers(0, S)) for T in frames]
    offs, pos = [], int(rng.integers(0, 3))
    for n in lens:
        offs.append(pos)
        pos += n + int(rng.integers(0, 4))
    left = np.zeros(pos, np.int16)
    right = np.zeros(pos, np.int16)
    for i, (o_, n) in enumerate(zip(offs, lens)):
        left[o_:o_ + n] = synth_utterance(n, 500 + 2 * i, sr=sr)
        right[o_:o_ + n] = synth_utterance(n, 501 + 2 * i, sr=sr, f=277.0 + 31 * i)
    mono = ((left.astype(np.int32) + right.astype(np.int32)) >> 1).astype(np.int16)
    inter = np.empty(2 * pos, np.int16)
    inter[0::2], inter[1::2] = left, right
    kw = dict(W=W, S=S, nb=nb, sr=sr, nc=nc, dyn=dyn, l1=2, l2=2)
    ms, cfg, w = make_pair(pkg, orc, max(lens) + 2000, channels=2, **kw)
    assert ms.dominant_kernel_name() == "k_front512"
    rows, total = ms.batch_plan(offs, lens)
    got = ms.batch_run_host(inter)
    assert total == sum(frames) and np.isfinite(got).all()
    g = groups_of(dyn)
    mm, _, _ = make_pair(pkg, orc, max(lens) + 2000, **kw)
    mm.batch_plan(offs, lens)
    assert_close(got, mm.batch_run_host(mono), "stereo in the kernel vs the host-side downmix", groups=g)
    D = (2 + (2 if dyn == 2 else 0)) if dyn else 0
    for i, (o_, n, T) in enumerate(zip(offs, lens, frames)):
        if T <= 2 * D:
            continue
        want = orc.run_utterance(cfg, mono[o_:o_ + n], w, bug_compat=False)
        assert_close(got[rows[i]:rows[i] + T], want, "utterance %d (%d frames)" % (i, T), groups=g)


def test_c5_full_size_properties(pkg, orc):
    """BASELINE configs[4] at full size on one GPU: 200 stereo utterances x 10 s at 44.1 kHz (441 000 samples per channel)
    -> 199 600 frames x 120.  Every row written and finite, duplicate utterances give the same bits wherever they sit,
    a second pass and a split batch reproduce the first bit for bit, silence gives exact-zero deltas, and sampled
    utterances agree with the oracle."""
    import torch
    n_utt, n = 200, 441000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    pcm = (3000.0 * torch.randn((n_utt, n, 2), generator=g, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    pcm[100] = pcm[7]
    pcm[199] = pcm[7]
    pcm[31] = 0
    m, cfg, w = make_pair(pkg, orc, n + 1000, W=1102, S=441, nb=128, sr=44100.0, nc=40, dyn=2, channels=2)
    rows, total = m.batch_plan(np.arange(n_utt) * n, np.full(n_utt, n))
    T = (n - (1102 - 441)) // 441
    assert total == n_utt * T == 199600 and m.dominant_kernel_name() == "k_front2048"
    out = torch.full((total, 120), float("nan"), dtype=torch.float32, device=dev)
    m.batch_run_device(pcm.data_ptr(), n_utt * n, out.data_ptr())
    m.synchronize()
    assert bool(torch.isfinite(out).all())
    o7 = out[rows[7]:rows[7] + T]
    assert torch.equal(o7, out[rows[100]:rows[100] + T]) and torch.equal(o7, out[rows[199]:rows[199] + T])
    sil = out[rows[31]:rows[31] + T]
    assert bool((sil[:, 40:] == 0).all()) and float(sil[:, :40].abs().max()) <= 1e-4 * 69.08
    out2 = torch.empty_like(out)
    m.batch_run_device(pcm.data_ptr(), n_utt * n, out2.data_ptr())
    m.synchronize()
    assert torch.equal(out, out2)
    m2, _, _ = make_pair(pkg, orc, n + 1000, W=1102, S=441, nb=128, sr=44100.0, nc=40, dyn=2, channels=2)
    half = 77
    r2, t2 = m2.batch_plan(np.arange(half) * n, np.full(half, n))
    outa = torch.empty((t2, 120), dtype=torch.float32, device=dev)
    m2.batch_run_device(pcm.data_ptr(), n_utt * n, outa.data_ptr())
    m2.synchronize()
    assert torch.equal(outa, out[:t2])
    for u in (0, 7, 150):
        x = pcm[u].cpu().numpy().astype(np.int32)
        mono = ((x[:, 0] + x[:, 1]) >> 1).astype(np.int16)
        want = orc.run_utterance(cfg, mono, w, bug_compat=False)
        assert_close(out[rows[u]:rows[u] + T].cpu().numpy(), want, "C5 utt %d" % u, groups=3)


@pytest.mark.parametrize("opts", [
    ["--banks", "26", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2", "--l1", "3", "--l2", "3"],
    [],                                                                   # the reference main()'s defaults: 15 / 12 + c0 / CVN
    ["--banks", "40", "--ceps", "13", "--norm", "2", "--dyn", "2", "--bug-compat", "0"],
    ["--banks", "24", "--ceps", "0", "--norm", "1", "--dyn", "1", "--l1", "2", "--norm-after-dyn", "0"],
    ["--banks", "26", "--ceps", "12", "--norm", "3", "--dyn", "2", "--htk"],
])
def test_cpp_driver_batches_are_byte_identical_to_the_per_file_loop(tmp_path, opts):
    """afet_hip drains its file queue into batches (one mfx_batch_plan + mfx_batch_run_host per batch, the extractor on the
    streaming interface's kernels, the reference's single-block flush behaviour B1 applied to the rows): every output file
    must be byte-identical to what the per-file loop (--batch-mb 0: set_input / apply / get_output_data / flush per file,
    ASR_OCL.cpp:227-301) writes -- text and HTK, normalisation on and off, bug-compat on and off, mixed with files that
    stay on the loop (longer than --sample-limit)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    srcs = ["a0001.wav", "a1.wav", "sample1_sphere.wav", "sample1_riff.wav", "a0001.wav", "a1.wav", "a0001.wav"] * 3

    def run(tag, extra):
        args = []
        for i, s_ in enumerate(srcs):
            args += [os.path.join(GOLDEN, s_), str(tmp_path / ("%s_%d.out" % (tag, i)))]
        subprocess.check_call([exe] + opts + extra + args, stdout=subprocess.DEVNULL)
        return [open(tmp_path / ("%s_%d.out" % (tag, i)), "rb").read() for i in range(len(srcs))]

    loop = run("loop", ["--batch-mb", "0"])
    batch = run("batch", [])
    small = run("small", ["--batch-mb", "1", "--io-threads", "3"])          # several batches, double buffering exercised
    mixed = run("mixed", ["--sample-limit", "100000"])                      # a0001 (114 000 samples) stays on the loop
    mixed_loop = run("mixedloop", ["--sample-limit", "100000", "--batch-mb", "0"])
    assert all(len(t) > 1000 for t in loop)
    assert batch == loop and small == loop
    assert mixed == mixed_loop


def test_cpp_driver_8khz_files_batches_identical_to_the_loop(tmp_path):
    """8 kHz files (256-point transforms: the zero-stuffed form of k_front512 in both the batch entry's streaming-kernel
    engine and the per-file loop): batches and loop write the same bytes, and the rows equal the Python streaming path's."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    lens = [8000, 12345, 30001, 4000, 16000]
    for i, n in enumerate(lens):
        data = synth_utterance(n, 300 + i, sr=8000.0).astype("<i2").tobytes()
        hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
            "<IHHIIHH", 16, 1, 1, 8000, 16000, 2, 16) + b"data" + struct.pack("<I", len(data))
        open(tmp_path / ("u%d.wav" % i), "wb").write(hdr + data)
    opts = ["--high-freq", "4000", "--banks", "23", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2", "--htk"]

    def run(tag, extra):
        args = []
        for i in range(len(lens)):
            args += [str(tmp_path / ("u%d.wav" % i)), str(tmp_path / ("%s_%d.htk" % (tag, i)))]
        subprocess.check_call([exe] + opts + extra + args, stdout=subprocess.DEVNULL)
        return [open(tmp_path / ("%s_%d.htk" % (tag, i)), "rb").read() for i in range(len(lens))]

    loop = run("loop", ["--batch-mb", "0"])
    batch = run("batch", [])
    assert all(len(t) > 1000 for t in loop) and batch == loop


def test_cpp_driver_multichannel_downmix_policy(orc, a0001, tmp_path):
    """ONE multi-channel policy across the product, pinned here so it cannot drift: mono = (L + R) >> 1 in integer
    arithmetic over the FIRST TWO channels (the batch kernels' channels = 2 downmix; further channels are ignored).  The
    reference itself has no downmix -- it reads `frames` interleaved shorts into a mono-sized buffer, ASR_OCL.cpp:229-231 --
    and ships no multi-channel fixture: this is a deliberate deviation, parity unpinned."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    n = 48000
    left = a0001[:n].astype(np.int32)
    right = np.roll(a0001[:n], 777).astype(np.int32) // 2
    third = np.full(n, 12345, np.int32)                      # a third channel that must not leak into the result

    def wav(path, chans):
        data = np.stack(chans, axis=1).astype("<i2").tobytes()
        hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
            "<IHHIIHH", 16, 1, len(chans), 16000, 16000 * 2 * len(chans), 2 * len(chans), 16) + b"data" + struct.pack("<I", len(data))
        open(path, "wb").write(hdr + data)

    wav(tmp_path / "mono.wav", [(left + right) >> 1])
    wav(tmp_path / "stereo.wav", [left, right])
    wav(tmp_path / "three.wav", [left, right, third])
    opts = ["--banks", "26", "--ceps", "13", "--c0", "0", "--norm", "0", "--dyn", "2"]
    for mode in ([], ["--batch-mb", "0"]):
        args = []
        for name in ("mono", "stereo", "three"):
            args += [str(tmp_path / (name + ".wav")), str(tmp_path / (name + ".txt"))]
        subprocess.check_call([exe] + opts + mode + args, stdout=subprocess.DEVNULL)
        ref = open(tmp_path / "mono.txt").read()
        assert len(ref) > 10000
        assert open(tmp_path / "stereo.txt").read() == ref and open(tmp_path / "three.txt").read() == ref
