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


def assert_close(got, want, what="", tol_max=TOL_MAX, tol_l2=TOL_L2, groups=None):
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
        scale = max(np.abs(b).max(), 1e-30)
        emax = np.abs(a - b).max() / scale
        el2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert emax <= tol_max, "%s group %d: max err / max|ref| = %.3g > %.3g" % (what, i, emax, tol_max)
        assert el2 <= tol_l2, "%s group %d: rel L2 = %.3g > %.3g" % (what, i, el2, tol_l2)
