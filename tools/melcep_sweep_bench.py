"""Dev aid / evidence: the VTLN alpha sweep on one FFT result (ASR_OCL.cpp:236-243 as mfx_apply_alphas) -- 8 warped filterbanks
over the stored spectrum of a 10 M-sample block (62 498 frames), i.e. k_melcep with grid.y = 8, timed per call; and the plain
streaming apply() (one filterbank) on the same block.  Prints frames x alphas per second.  MFX_LIB selects a library build."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
n = 10_000_000
rng = np.random.default_rng(5)
pcm = (3000.0 * rng.standard_normal(n)).astype(np.int16)
alphas = [0.88, 0.91, 0.94, 0.97, 1.0, 1.03, 1.06, 1.09]
for name, nb, nc, W, S, sr in (("C2 shape (512 pt, 40 mel, 13 MFCC)", 40, 13, 400, 160, 16000.0),
                               ("C5 shape (2048 pt, 128 mel, 40 MFCC)", 128, 40, 1102, 441, 44100.0)):
    m = pkg.MfccHip(n, W, S, nb, sr, 64.0, sr / 2, nc, False, 22.0, 0, 2, 3, 3, True)
    m.set_window(pkg.reference_window(W))
    frames = m.set_input(pcm[:m.get_input_buffer_size()])
    m.apply_alphas(alphas)
    m.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        m.apply_alphas(alphas)
    m.synchronize()
    dt = (time.perf_counter() - t0) / K
    m.apply()
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        m.apply()
    m.synchronize()
    dt1 = (time.perf_counter() - t0) / K
    print("%-40s %d frames: sweep of %d alphas %.3f ms = %.1f M frame-alphas/s; apply() %.3f ms = %.1f M frames/s" % (
        name, frames, len(alphas), dt * 1e3, frames * len(alphas) / dt / 1e6, dt1 * 1e3, frames / dt1 / 1e6))
    m.close()
