"""Developer aid: stage-by-stage comparison of the HIP path with the oracle on the GPU box.
Run as:  gpurun -- python tools/gpu_debug.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
orc = G.load_oracle()


def synth(n, seed, f=440.0, sr=16000.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return np.clip(np.round(3000 * rng.standard_normal(n) + 6000 * np.sin(2 * np.pi * f * t / sr)),
                   -32768, 32767).astype(np.int16)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    W, S, nb, nc = 400, 160, 40, 13
    pcm = synth(16000, 7)
    window = pkg.reference_window(W)
    cfg = orc.make_config(20000, window_size=W, shift=S, num_banks=nb, ceps_len=nc, dyn=orc.DYN_ACC, delta_l1=3,
                          delta_l2=3)
    o = orc.OracleMfcc(cfg, window)
    m = pkg.MfccHip(20000, W, S, nb, 16000.0, 64.0, 8000.0, nc, False, 22.0, pkg.NORM_NONE, pkg.DYN_ACC, 3, 3, True)
    m.set_window(window)
    # tables
    t = o.tables()
    print("mel table equal:", np.array_equal(m.debug_read(0).reshape(2, -1), t["filters"]),
          "beg equal:", np.array_equal(m.debug_read(1), t["filter_beg"]),
          "dct equal:", np.array_equal(m.debug_read(2).reshape(nb, -1), t["dct_matrix"]))
    n_o = o.set_input(pcm)
    n_m = m.set_input(pcm)
    print("frames", n_o, n_m)
    wcnd = n_o + 6
    fft = o.tap("fft", wcnd).reshape(wcnd, 512, 2)
    mag_o = np.sqrt(fft[:, :257, 0] ** 2 + fft[:, :257, 1] ** 2) / 512
    mag_m = m.debug_read(3).reshape(wcnd, -1)[:, :257]
    print("magnitude rel err", rel(mag_m, mag_o), "worst bin", np.unravel_index(np.abs(mag_m - mag_o).argmax(), mag_o.shape))
    err = np.abs(mag_m - mag_o) / mag_o.max()
    print("err by frame (first 8):", np.round(err.max(1)[:8], 4))
    print("err by k mod 16:", np.round([err[:, r::16].max() for r in range(16)], 4))
    print("err by k // 16:", np.round([err[:, 16 * p:16 * p + 16].max() for p in range(16)], 4), "nyq", err[:, 256].max())
    o.apply()
    m.apply()
    a, b = o.get_output_data(n_o), m.get_output_data(n_m)
    for nm, sl in (("static", slice(0, 13)), ("delta", slice(13, 26)), ("acc", slice(26, 39))):
        print(nm, "rel err", rel(b[:, sl], a[:, sl]))
    f_o = o.flush()
    f_m = m.flush()
    o.apply()
    m.apply()
    print("flush frames", f_o, f_m, "rel err", rel(m.get_output_data(f_m), o.get_output_data(f_o)))
    # batch
    rows, total = m.batch_plan([0], [pcm.size])
    got = m.batch_run_host(pcm)
    want = orc.run_utterance(cfg, pcm, window, bug_compat=False)
    print("batch", got.shape, want.shape, "rel err", rel(got, want))
    for nm, sl in (("static", slice(0, 13)), ("delta", slice(13, 26)), ("acc", slice(26, 39))):
        print("  batch", nm, rel(got[:, sl], want[:, sl]))


if __name__ == "__main__":
    main()
