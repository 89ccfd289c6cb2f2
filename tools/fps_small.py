import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pytorch3d_pointops_amd import _C, synth
from pytorch3d_pointops_amd.functions import sample_farthest_points
dev = torch.device("cuda:0")
def ev(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
for (B,N,K) in ((2,1024,128),(2,4096,128),(8,2048,128),(64,8192,256),(2,1024,512)):
    pts = torch.from_numpy(synth.uniform_f32(3,(B,N,3))).to(dev)
    L = torch.full((B,),N,dtype=torch.int64,device=dev); Kt=torch.full((B,),K,dtype=torch.int64,device=dev); S=torch.zeros((B,),dtype=torch.int64,device=dev)
    t_c = ev(lambda: _C.sample_farthest_points(pts, L, Kt, S))
    t_f = ev(lambda: sample_farthest_points(pts, K=K))
    print((B,N,K), "_C %.1f us (%.2f us/iter)  functions %.1f us" % (t_c, t_c/K, t_f), flush=True)
