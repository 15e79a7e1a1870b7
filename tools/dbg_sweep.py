"""dev aid (GPU box): the failing random configuration with one parameter changed at a time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
from conftest import synth_utterance
pkg, orc = G.load_package(), G.load_oracle()
base = dict(W=489, S=132, sr=22050.0, nb=63, nc=20, c0=True, dyn=1, l1=4, l2=3, low=0.0, high=5512.5, seed=1003, mult=3)
def run(**ch):
    c = dict(base); c.update(ch)
    n = 40 * c["S"] + c["W"] + int(c["seed"] % 7) * 13
    pcm = synth_utterance(c["mult"] * n, c["seed"], sr=c["sr"])
    w = pkg.reference_window(c["W"])
    m = pkg.MfccHip(pcm.size + 1000, c["W"], c["S"], c["nb"], c["sr"], c["low"], c["high"], c["nc"], c["c0"], 22.0, 0, c["dyn"], c["l1"], c["l2"], True, device=0, bug_compat=False)
    m.set_window(w)
    m.batch_plan([0], [pcm.size])
    got = m.batch_run_host(pcm)
    cfg = orc.make_config(pcm.size + 1000, window_size=c["W"], shift=c["S"], num_banks=c["nb"], sample_rate=c["sr"], low_freq=c["low"],
                          high_freq=c["high"], ceps_len=c["nc"], want_c0=c["c0"], norm=0, dyn=c["dyn"], delta_l1=c["l1"], delta_l2=c["l2"])
    want = orc.run_utterance(cfg, pcm, w, bug_compat=False)
    s = m.process_stream(pcm)
    e = np.abs(got - want)
    print("%-28s T %4d  batch err %.3g rows %s | stream err %.3g" % (ch, got.shape[0], e.max(), np.unique(np.nonzero(e > 1e-3)[0])[:8], np.abs(s - want).max()))
    m.close()
run()
run(dyn=0)
run(c0=False)
run(nc=13, c0=False)
run(nb=40)
run(nb=40, nc=13, c0=False)
run(high=11025.0)
run(low=64.0)
run(sr=16000.0, high=4000.0)
run(W=400)
run(W=488)
run(S=160)
run(mult=4)
run(mult=5)
run(seed=1004)
run(l1=3)
