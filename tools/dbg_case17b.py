import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg=G.load_package(); orc=G.load_oracle()
pcm0=np.load(os.path.join(ROOT,'gpurun_in_case.npy'))
P=dict(W=91,S=19,nb=32,nc=6,sr=16000.0,alpha=0.95,lens=[716,547,288,575],offs=[2,721,1271,1559])
w=pkg.reference_window(P["W"])
def run(tag, pcm, lens=P["lens"], offs=P["offs"], engine=0, alpha=P["alpha"]):
    m=pkg.MfccHip(max(lens)+4*P["W"],P["W"],P["S"],P["nb"],P["sr"],64.0,P["sr"]/2,P["nc"],False,22.0,0,0,3,1,False,bug_compat=False,engine=engine)
    m.set_window(w); m.set_alpha(alpha)
    rows,total=m.batch_plan(offs,lens); got=m.batch_run_host(pcm); m.close()
    cfg=orc.make_config(max(lens)+5*P["W"],window_size=P["W"],shift=P["S"],num_banks=P["nb"],sample_rate=P["sr"],high_freq=P["sr"]/2,ceps_len=P["nc"],dyn=0)
    errs=[]
    for u,n in enumerate(lens):
        want=orc.run_utterance(cfg,pcm[offs[u]:offs[u]+n],w,alpha=alpha,bug_compat=False)
        g=got[rows[u]:rows[u]+want.shape[0]]
        e=np.abs(g-want).max(axis=1)/max(np.abs(want).max(),1e-30)
        bad=np.nonzero(e>1e-4)[0]
        errs.append("%.1e%s"%(e.max(), (" rows "+str(bad[:8].tolist())+"/%d"%want.shape[0]) if bad.size else ""))
    print("%-30s size %d: %s"%(tag,pcm.size," | ".join(errs)), flush=True)
run("as in the fuzz", pcm0)
run("padded with 64 zeros", np.concatenate([pcm0,np.zeros(64,np.int16)]))
run("padded with 1 zero", np.concatenate([pcm0,np.zeros(1,np.int16)]))
run("alpha 1", pcm0, alpha=1.0)
run("engine no-stuff", pcm0, engine=128)
run("only utt 0", pcm0, lens=[716], offs=[2])
run("only utt 3", pcm0, lens=[575], offs=[1559])
run("utts 0,1", pcm0, lens=[716,547], offs=[2,721])
