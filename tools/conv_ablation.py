"""conv_gemm ablations through bh_debug_force_tile(-1, bits): 2 = skip global reloads, 4 = skip LDS restaging (results are wrong by construction; timing only)."""
import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
shapes=[(128,32,64,64,3,1,1),(128,16,128,128,3,1,1),(128,8,256,256,3,1,1),(128,64,64,64,3,1,1)]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for (N,H,Ci,Co,k,s,p) in shapes:
    d=K.conv_desc(N,H,H,Ci,Co,k,s,p)
    x=torch.randn(N,H,H,Ci,device='cuda'); w=torch.randn(Co,k,k,Ci,device='cuda')*0.05
    fl=K.conv_flops(d); out=[]
    for rep in range(2):
        for pr in (0,2,6):
            lib.bh_debug_force_tile(-1,pr)
            out.append('p%d %.0f'%(pr, fl/bench(lambda: K.conv_fwd(x,w,None,d))/1e9))
    print((N,H,Ci,Co), ' '.join(out), flush=True)
