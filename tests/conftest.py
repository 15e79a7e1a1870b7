import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def graft():
    import __graft_entry__ as G
    # make sure the checker and the product library exist (cheap when already built)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")) or not os.path.exists(
            os.path.join(ROOT, "asr-featext-opencl_amd", "libmfcchip.so")):
        G.build()
    return G


@pytest.fixture(scope="session")
def orc(graft):
    return graft.load_oracle()


@pytest.fixture(scope="session")
def pkg(graft):
    return graft.load_package()


@pytest.fixture(scope="session")
def a0001(orc):
    pcm, sr = orc.read_wav_pcm16(os.path.join(GOLDEN, "a0001.wav"))
    assert sr == 16000
    return pcm[:, 0].copy()


@pytest.fixture(scope="session")
def a1(orc):
    pcm, sr = orc.read_wav_pcm16(os.path.join(GOLDEN, "a1.wav"))
    return pcm[:, 0].copy()


def synth_utterance(n, seed, sr=16000.0, f=None):
    """BASELINE.md 3 synthetic PCM: clip16(round(3000*N(0,1) + 6000*sin(2*pi*f*n/sr)))."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    if f is None:
        f = 100.0 + 37.0 * (seed % 64)
    t = np.arange(n)
    x = 3000.0 * rng.standard_normal(n) + 6000.0 * np.sin(2 * np.pi * f * t / sr)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


# Tolerances (BASELINE.json north_star: "MFCC + d + dd matching the CPU reference to <= 1e-4
# relative error"; SURVEY 8c: relative to signal scale, per column group).
TOL_MAX = 1e-4   # max |a-b| / max |b|
TOL_L2 = 1e-5    # ||a-b||_2 / ||b||_2


def assert_close(got, want, what="", tol_max=TOL_MAX, tol_l2=TOL_L2, groups=None, scale_floor=0.0):
    """scale_floor: lower bound of the signal scale the errors are measured against -- for outputs that are themselves
    rounding noise (an all-silent utterance: every log energy is log(1e-30) = -69.08 and the DCT of a constant vector
    cancels to ~1e-5), where the scale of the DCT's INPUT is the meaningful yardstick."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, got.shape, want.shape)
    if got.size == 0:
        return
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    cols = want.shape[1]
    g = groups or 1
    w = cols // g
    for i in range(g):
        a, b = got[:, i * w:(i + 1) * w], want[:, i * w:(i + 1) * w]
        scale = max(np.abs(b).max(), 1e-30, scale_floor)
        emax = np.abs(a - b).max() / scale
        el2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30, scale_floor * np.sqrt(b.size))
        assert emax <= tol_max, "%s group %d: max err / max|ref| = %.3g > %.3g" % (what, i, emax, tol_max)
        assert el2 <= tol_l2, "%s group %d: rel L2 = %.3g > %.3g" % (what, i, el2, tol_l2)


# Normalised outputs (CMN / CVN / MINMAX) are checked in three parts instead of under a looser blanket tolerance:
#   1. the un-normalised rows of the same configuration at the bar above (TOL_MAX / TOL_L2),
#   2. the statistics themselves, each within what part 1 allows it to move:
#        mean:        |d mean| <= rms(dx) <= TOL_L2-sized share of the group's scale             -> TOL_MEAN * scale
#        CVN  1/sigma: |d sigma| / sigma <= rms(dx) / sigma                                      -> 2 TOL_L2 * scale * mult
#        MINMAX 1/ext: |d ext| / ext <= (max|dx| + |d mean|) / ext                               -> (TOL_MAX + TOL_MEAN) * scale * mult
#      (scale = max|x| of the column group; scale * mult is the column's amplification, measured, not assumed),
#   3. every normalised column within the bound that the MEASURED differences of 1 and 2 imply for y = (x - mean) * mult:
#        |dy| <= (|dx| + |d mean|) * mult + |y| * |d mult| / mult  (+ the rounding of the two operations).
# Degenerate statistics (one row: CVN divides 0 by 0 in the reference too, normalizercpu.cpp:48) must be degenerate on
# both sides and are left out of parts 2 and 3.
TOL_MEAN = 1e-5


def assert_normalised_close(y_got, y_want, x_got, x_want, st_got, st_want, groups, nad, what="", norm=2):
    """y_*: normalised rows [n][groups * cols]; x_*: the same rows of the norm = NONE twin; st_*: statistics
    [G][2][cols] (mean, multiplier), G = groups when normalising after the deltas (nad) else 1 (statics only)."""
    y_got, y_want = np.asarray(y_got, np.float64), np.asarray(y_want, np.float64)
    x_got, x_want = np.asarray(x_got, np.float64), np.asarray(x_want, np.float64)
    st_got, st_want = np.asarray(st_got, np.float64), np.asarray(st_want, np.float64)
    assert y_got.shape == y_want.shape == x_got.shape == x_want.shape, what
    if y_got.shape[0] == 0:
        return 0.0
    assert_close(x_got, x_want, what + " (un-normalised twin)", groups=groups)              # part 1
    cols = y_want.shape[1] // groups
    worst = 0.0
    for g in range(groups):
        sl = slice(g * cols, (g + 1) * cols)
        xg, xw, yg, yw = x_got[:, sl], x_want[:, sl], y_got[:, sl], y_want[:, sl]
        sg = g if nad else 0
        mean_g, mean_w = st_got[sg, 0], st_want[sg, 0]
        mult_g, mult_w = st_got[sg, 1], st_want[sg, 1]
        ok = np.isfinite(mult_w) & np.isfinite(mean_w)
        assert np.array_equal(ok, np.isfinite(mult_g) & np.isfinite(mean_g)), "%s group %d: degenerate statistics differ" % (what, g)
        assert np.array_equal(np.isfinite(yg), np.isfinite(yw)), "%s group %d: non-finite rows differ" % (what, g)
        if not ok.any():
            continue
        scale = max(np.abs(x_want[:, (sg * cols):((sg + 1) * cols)]).max(), 1e-30)
        d_mean = np.where(ok, np.abs(mean_g - mean_w), 0.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            # (a one-row block gives sqrt(0 / x) = 0 on both sides: equal, relative difference 0)
            d_mult = np.where(ok, np.abs(mult_g - mult_w) / np.maximum(np.abs(mult_w), 1e-300), 0.0)
        if g == sg:                                                                         # part 2
            assert (d_mean <= TOL_MEAN * scale).all(), "%s group %d: mean differs by %.3g of the scale" % (
                what, g, d_mean.max() / scale)
            amp = np.where(ok, scale * np.abs(mult_w), 0.0)
            lim = {1: 0.0 * amp, 2: 2 * TOL_L2 * amp, 3: (TOL_MAX + TOL_MEAN) * amp}[norm] + 1e-6
            assert (d_mult <= lim).all(), "%s group %d: multiplier differs by %s (allowed %s)" % (
                what, g, d_mult[d_mult > lim], lim[d_mult > lim])
        if g != sg:
            d_mean = 0.0 * d_mean                                   # deltas of normalised statics carry no mean
        dx = np.abs(xg - xw).max(axis=0)
        ymax = np.where(ok, np.abs(np.where(np.isfinite(yw), yw, 0.0)).max(axis=0), 0.0)
        # float32 rounding of the subtraction and the product (and, before the deltas, of the regression on O(1) values)
        rounding = 2.5e-7 * np.maximum(ymax, 1.0) + (0.0 if nad or g == 0 else 1e-6)
        bound = (dx + d_mean) * np.where(ok, np.abs(mult_w), 0.0) * (1.0 + d_mult) + ymax * d_mult + rounding  # part 3
        err = np.where(ok, np.abs(np.where(np.isfinite(yg - yw), yg - yw, 0.0)).max(axis=0), 0.0)
        bad = err > bound * 1.02
        assert not bad.any(), "%s group %d: columns %s exceed the derived bound (err %s, bound %s)" % (
            what, g, np.nonzero(bad)[0], err[bad], bound[bad])
        worst = max(worst, float((err / np.maximum(ymax, 1e-30)).max()))
    return worst
