import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg=G.load_package(); orc=G.load_oracle()
base=dict(W=91,S=19,nb=32,nc=6,c0=False,sr=16000.0,alpha=0.95,lens=[716,547,288,575],offs=[2,721,1271,1559],engine=0)
rng=np.random.default_rng(7)
pcm=(4000.0*rng.standard_normal(4000)).round().clip(-32768,32767).astype(np.int16)
def run(tag, **kw):
    P=dict(base); P.update(kw)
    w=pkg.reference_window(P["W"])
    try:
        m=pkg.MfccHip(max(P["lens"])+4*P["W"],P["W"],P["S"],P["nb"],P["sr"],64.0,P["sr"]/2,P["nc"],P["c0"],22.0,0,0,3,1,False,bug_compat=False,engine=P["engine"])
    except Exception as e:
        print(tag,"create failed",e); return
    m.set_window(w); m.set_alpha(P["alpha"])
    rows,total=m.batch_plan(P["offs"],P["lens"]); got=m.batch_run_host(pcm); name=m.dominant_kernel_name(); m.close()
    cfg=orc.make_config(max(P["lens"])+5*P["W"],window_size=P["W"],shift=P["S"],num_banks=P["nb"],sample_rate=P["sr"],high_freq=P["sr"]/2,ceps_len=P["nc"],want_c0=P["c0"],dyn=0)
    errs=[]
    for u,n in enumerate(P["lens"]):
        want=orc.run_utterance(cfg,pcm[P["offs"][u]:P["offs"][u]+n],w,alpha=P["alpha"],bug_compat=False)
        g=got[rows[u]:rows[u]+want.shape[0]]
        errs.append(float(np.abs(g-want).max()/max(np.abs(want).max(),1e-30)))
    print("%-34s %-12s errs %s" % (tag,name,["%.1e"%e for e in errs]), flush=True)
    return got
run("case 17 as found")
run("even shift 20", S=20)
run("even offsets", offs=[2,722,1272,1560])
run("even shift + offsets", S=20, offs=[2,722,1272,1560])
run("alpha 1", alpha=1.0)
run("20 filters", nb=20)
run("window 64", W=64)
run("window 88", W=88)
run("one utterance", lens=[716], offs=[2])
run("one utterance at 0", lens=[716], offs=[0])
run("mel only (nc 0)", nc=0)
run("no stuffing (k_front_wave)", engine=128)
run("256 points: W 177 S 19", W=177)
run("64 points: W 44 S 19", W=44, nb=12)
