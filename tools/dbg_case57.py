import sys, os, faulthandler, numpy as np
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg=G.load_package(); orc=G.load_oracle()
P=dict(W=62,S=30,nb=16,nc=4,c0=True,dyn=2,l1=3,l2=1,sr=8000.0,alpha=0.95,n=1295,blk=271)
rng=np.random.default_rng(5)
pcm=(4000.0*rng.standard_normal(P["n"])).round().clip(-32768,32767).astype(np.int16)
w=pkg.reference_window(P["W"])
def log(*a): print(*a, flush=True)
which=sys.argv[1] if len(sys.argv)>1 else "both"
if which in ("both","batch"):
    m=pkg.MfccHip(P["n"]+4*P["W"],P["W"],P["S"],P["nb"],P["sr"],64.0,P["sr"]/2,P["nc"],P["c0"],22.0,0,P["dyn"],P["l1"],P["l2"],False,bug_compat=False)
    m.set_window(w); m.set_alpha(P["alpha"]); log("batch handle ok", m.dominant_kernel_name())
    rows,total=m.batch_plan([0],[P["n"]]); log("plan", rows,total)
    got=m.batch_run_host(pcm); log("batch ok", got.shape)
    m.close()
if which in ("both","stream"):
    ms=pkg.MfccHip(P["blk"],P["W"],P["S"],P["nb"],P["sr"],64.0,P["sr"]/2,P["nc"],P["c0"],22.0,0,P["dyn"],P["l1"],P["l2"],False,bug_compat=True)
    ms.set_window(w); log("stream handle ok, ibs", ms.get_input_buffer_size(), "max frames", ms.max_frames_out())
    lim=ms.get_input_buffer_size(); pos=0; out=[]
    ms.set_alpha(P["alpha"])
    while pos < pcm.size:
        n=ms.set_input(pcm[pos:pos+lim]); log("set_input", pos, n)
        pos+=lim
        if n>0:
            ms.set_alpha(P["alpha"]); ms.apply(); log(" apply ok")
            out.append(ms.get_output_data(n)); log(" got", out[-1].shape)
    n=ms.flush(); log("flush", n)
    if n>0:
        ms.apply(); out.append(ms.get_output_data(n))
    log("stream ok", sum(o.shape[0] for o in out))
    ms.close()
