import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import _C, synth
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for (B,N,D,K) in ((8,4096,64,20),(8,4096,32,20),(8,4096,16,16),(8,4096,9,16),(4,16384,3,64),(8,4096,8,16),(32,1024,64,20),(8,4096,3,40)):
    x=torch.from_numpy(synth.uniform_f32(5,(B,N,D))).to(dev); y=torch.from_numpy(synth.uniform_f32(6,(B,N,D))).to(dev)
    L=torch.full((B,),N,dtype=torch.int64,device=dev)
    ms=timeit(lambda: _C.knn_points_idx(x,y,L,L,2,K,-1))
    print(json.dumps({"B":B,"N":N,"D":D,"K":K,"ms":ms,"Gpair_dims_per_s":B*N*N*D/ms/1e6}))
