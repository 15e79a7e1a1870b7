"""Dev aid: throughput of the streaming (drop-in) interface with the reference's default block size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
rng = np.random.default_rng(0)
n = 40_000_000
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
    m.close()
