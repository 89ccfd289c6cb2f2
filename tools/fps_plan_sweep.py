"""FPS cluster geometry (points per lane -> workgroups per cloud) over batch shapes: the data behind fps_plan()."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch3d_pointops_amd import _C, synth
dev = torch.device("cuda:0")
def ev(fn, n=10):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for (B,N,K) in ((16,131072,1024),(4,32768,512),(1,131072,1024),(8,65536,512),(32,16384,256),(64,8192,256),(2,262144,512),(16,100000,512)):
    pts = torch.from_numpy(synth.uniform_f32(3,(B,N,3))).to(dev)
    L = torch.full((B,),N,dtype=torch.int64,device=dev); Kt=torch.full((B,),K,dtype=torch.int64,device=dev); S=torch.zeros((B,),dtype=torch.int64,device=dev)
    ref=None
    for knob in ("", "fps_small_ppt=0", "fps_ppt=4", "fps_ppt=8"):
        os.environ["POINTOPS_DEBUG"]=knob
        try:
            t=ev(lambda: _C.sample_farthest_points(pts, L, Kt, S))
            out=_C.sample_farthest_points(pts, L, Kt, S)
            if ref is None: ref=out
            print((B,N,K), "%-26s %.3f ms  %.2f us/iter  same=%s" % (knob, t, t/K*1e3, bool(torch.equal(out,ref))), flush=True)
        except Exception as e:
            print((B,N,K), knob, "ERR", e)
