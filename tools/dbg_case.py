"""dev aid (GPU box): one configuration through the batch entry vs the oracle, per column error report.
   python3 tools/dbg_case.py   (MFX_LIB selects the library)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
from conftest import synth_utterance
pkg, orc = G.load_package(), G.load_oracle()
c = dict(fft=512, W=489, S=132, sr=16000.0, nb=63, nc=20, c0=False, dyn=1, l1=3, l2=3, low=64.0, high=8000.0, seed=1000)
for k, v in (a.split("=") for a in sys.argv[1:]):
    c[k] = type(c[k])(v) if not isinstance(c[k], bool) else v == "1"
n = 40 * c["S"] + c["W"] + int(c["seed"] % 7) * 13
pcm = synth_utterance(3 * n, c["seed"], sr=c["sr"])
w = pkg.reference_window(c["W"])
m = pkg.MfccHip(pcm.size + 1000, c["W"], c["S"], c["nb"], c["sr"], c["low"], c["high"], c["nc"], c["c0"], 22.0, 0, c["dyn"], c["l1"], c["l2"], True, device=0)
m.set_window(w)
m.batch_plan([0], [pcm.size])
got = m.batch_run_host(pcm)
cfg = orc.make_config(pcm.size + 1000, window_size=c["W"], shift=c["S"], num_banks=c["nb"], sample_rate=c["sr"], low_freq=c["low"],
                      high_freq=c["high"], ceps_len=c["nc"], want_c0=c["c0"], norm=0, dyn=c["dyn"], delta_l1=c["l1"], delta_l2=c["l2"])
want = orc.run_utterance(cfg, pcm, w, bug_compat=False)
s = m.process_stream(pcm)
e = np.abs(got - want)
print("kernel", m.dominant_kernel_name(), "shape", got.shape, "max|want|", np.abs(want).max())
print("batch  max err %.3g at %s" % (e.max(), np.unravel_index(e.argmax(), e.shape)))
print("stream max err %.3g" % np.abs(s - want).max())
print("rows with err > 1e-3:", np.unique(np.nonzero(e > 1e-3)[0])[:40])
print("per column max:", np.round(e.max(0), 5))
