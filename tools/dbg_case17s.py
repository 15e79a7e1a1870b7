import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg=G.load_package(); orc=G.load_oracle()
rng=np.random.default_rng(7)
pcm=(4000.0*rng.standard_normal(4000)).round().clip(-32768,32767).astype(np.int16)
def run(tag, W=91, S=19, nb=32, nc=6, blk=604, n=716, engine=0, alpha=0.95, dyn=0, sr=16000.0):
    seg=pcm[2:2+n]; w=pkg.reference_window(W)
    cfg=orc.make_config(blk,window_size=W,shift=S,num_banks=nb,sample_rate=sr,high_freq=sr/2,ceps_len=nc,dyn=dyn,delta_l1=3,delta_l2=1)
    o=orc.OracleMfcc(cfg,w)
    m=pkg.MfccHip(blk,W,S,nb,sr,64.0,sr/2,nc,False,22.0,0,dyn,3,1,False,bug_compat=True,engine=engine)
    m.set_window(w)
    lim=m.get_input_buffer_size(); pos=0; res=[]
    while True:
        last = pos>=seg.size
        if last: a=m.flush(); b=o.flush()
        else: a=m.set_input(seg[pos:pos+lim]); b=o.set_input(seg[pos:pos+lim]); pos+=lim
        if a!=b: res.append("COUNT %d/%d"%(a,b)); break
        if a>0:
            m.set_alpha(alpha); o.set_alpha(alpha); m.apply(); o.apply()
            y=m.get_output_data(a); z=o.get_output_data(a)
            res.append("%d:%.1e"%(a, np.abs(y-z).max()/max(np.abs(z).max(),1e-30)))
        if last: break
    m.close(); o.close()
    print("%-40s %s"%(tag," ".join(res)), flush=True)
run("case 17 stream (engine 0)")
run("engine 32 (DMA)", engine=32)
run("S 20", S=20)
run("W 64", W=64)
run("dyn 2", dyn=2)
run("blk 2000 (one block)", blk=2000)
run("blk 300", blk=300)
run("256 pts W 177", W=177, blk=900)
run("512 pts W 400 S 19", W=400, blk=1200, n=3000)
run("512 pts W 400 S 160 dyn0 blk 1200", W=400, S=160, blk=1200, n=3000)
run("64 pts W 44 nb 12", W=44, nb=12)
