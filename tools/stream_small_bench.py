"""Dev aid: the streaming (drop-in) interface on SMALL blocks (one 10-s utterance per call sequence), per call of the C ABI."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
ENGINE = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 32 = DMA commands for small blocks (the round-2 path)
rng = np.random.default_rng(0)
for limit, norm in ((160_000, pkg.NORM_NONE), (160_000, 2), (480_000, pkg.NORM_NONE)):
    n = 400 * limit
    pcm = (3000 * rng.standard_normal(n)).astype(np.int16)
    m = pkg.MfccHip(limit, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, norm, pkg.DYN_ACC, 3, 3, True, engine=ENGINE)
    m.set_window(pkg.reference_window(400))
    L, hnd = m._L, m._h
    lim = m.get_input_buffer_size()
    nout = (m.estimated_window_count(lim) + 64) * m.get_output_data_width()
    out = np.zeros(nout, np.float32)
    op = C.cast(out.ctypes.data, C.POINTER(C.c_float))
    nfr = C.c_int32()
    for rep in range(2):
        t = [0.0, 0.0, 0.0]
        frames, pos = 0, 0
        t00 = time.perf_counter()
        while pos + lim <= n:
            ip = C.cast(pcm[pos:pos + lim].ctypes.data, C.POINTER(C.c_short))
            t0 = time.perf_counter()
            L.mfx_set_input(hnd, ip, lim, C.byref(nfr))
            t1 = time.perf_counter()
            L.mfx_apply(hnd)
            t2 = time.perf_counter()
            L.mfx_get_output_data(hnd, op, nfr.value)
            t3 = time.perf_counter()
            t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2
            frames += nfr.value
            pos += lim
        wall = time.perf_counter() - t00
        L.mfx_flush(hnd, C.byref(nfr))
        nb = pos // lim
    print("block %7d samples, norm %d: %5.1f us per block = set_input %5.1f + apply %5.1f + get_output_data %5.1f; %.1f M frames/s "
          "inside the calls, %.1f M with the loop around them" % (lim, norm, 1e6 * sum(t) / nb, 1e6 * t[0] / nb, 1e6 * t[1] / nb,
                                                                    1e6 * t[2] / nb, frames / sum(t) / 1e6, frames / wall / 1e6))
    m.close()
