"""Dev aid: exercise the streaming (set_input -> spectrum) kernels for profiling: 512-, 2048- and 4096-point."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
rng = np.random.default_rng(0)
pcm = (3000 * rng.standard_normal(8_000_000)).astype(np.int16)
for W, S, sr, nb, nc in ((400, 160, 16000.0, 40, 13), (1102, 441, 44100.0, 128, 40), (1102, 440, 44100.0, 128, 40),
                         (2400, 960, 96000.0, 64, 20)):
    m = pkg.MfccHip(4_000_000, W, S, nb, sr, 64.0, sr / 2, nc, False, 22.0, 0, 2, 3, 3, True)
    m.set_window(pkg.reference_window(W))
    for _ in range(3):
        out = m.process_stream(pcm)
    print(W, S, out.shape)
    m.close()
