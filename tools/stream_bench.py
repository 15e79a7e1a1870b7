"""Dev aid: throughput of the streaming (drop-in) interface with the reference's default block size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
rng = np.random.default_rng(0)
n = 160_000_000
pcm = (3000 * rng.standard_normal(n)).astype(np.int16)
for limit in (10_000_000, 1_000_000, 160_000):
    m = pkg.MfccHip(limit, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0, pkg.NORM_NONE, pkg.DYN_ACC, 3, 3, True)
    m.set_window(pkg.reference_window(400))
    m.process_stream(pcm[:limit])            # warm-up
    t0 = time.perf_counter()
    out = m.process_stream(pcm)
    dt = time.perf_counter() - t0
    # stage split on one block
    blk = pcm[:m.get_input_buffer_size()]
    t1 = time.perf_counter(); k = m.set_input(blk); m.synchronize(); t2 = time.perf_counter()
    m.apply(); m.synchronize(); t3 = time.perf_counter()
    o = m.get_output_data(k); t4 = time.perf_counter()
    print("block %9d samples: %.1f M frames/s end to end (%d frames in %.3f s); one block: set_input %.2f ms, apply %.2f ms, "
          "get_output_data %.2f ms for %d frames" % (limit, out.shape[0] / dt / 1e6, out.shape[0], dt, 1e3 * (t2 - t1),
                                                     1e3 * (t3 - t2), 1e3 * (t4 - t3), k))
    def c_abi_stream(in_buf, out_buf, tag):
        import ctypes as C
        L, hnd = m._L, m._h
        limit_s = m.get_input_buffer_size()
        nfr = C.c_int32()
        ip = C.cast(in_buf.ctypes.data, C.POINTER(C.c_short))
        op = C.cast(out_buf.ctypes.data, C.POINTER(C.c_float))
        frames, pos, t_lib = 0, 0, 0.0
        t0 = time.perf_counter()
        while pos < n:
            cnt = min(limit_s, n - pos)
            in_buf[:cnt] = pcm[pos:pos + cnt]              # the application producing its samples
            t1 = time.perf_counter()
            L.mfx_set_input(hnd, ip, cnt, C.byref(nfr))
            if nfr.value > 0:
                L.mfx_apply(hnd)
                L.mfx_get_output_data(hnd, op, nfr.value)
                frames += nfr.value
            t_lib += time.perf_counter() - t1
            pos += cnt
        t1 = time.perf_counter()
        L.mfx_flush(hnd, C.byref(nfr))
        if nfr.value > 0:
            L.mfx_apply(hnd)
            L.mfx_get_output_data(hnd, op, nfr.value)
            frames += nfr.value
        t_lib += time.perf_counter() - t1
        dt = time.perf_counter() - t0
        print("      C ABI, %s: %.1f M frames/s inside the library calls (%.3f s), %.1f M frames/s with the caller filling "
              "its buffer (%.3f s)" % (tag, frames / t_lib / 1e6, t_lib, frames / dt / 1e6, dt))
    lim = m.get_input_buffer_size()
    nout = (m.estimated_window_count(lim) + 64) * m.get_output_data_width()
    c_abi_stream(np.zeros(lim, np.int16), np.zeros(nout, np.float32), "pageable caller buffers (reused)")
    try:
        import torch
        c_abi_stream(torch.empty(lim, dtype=torch.int16).pin_memory().numpy(),
                     torch.empty(nout, dtype=torch.float32).pin_memory().numpy(), "pinned caller buffers")
    except Exception as e:
        print("      pinned caller buffers: skipped (%s)" % e)
    m.close()
