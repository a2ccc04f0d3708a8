"""Sweep of the wgrad split-K work-item target (bh_debug_force_tile(-3, n))."""
import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
shapes=[(128,32,64,64,3,1,1),(128,16,128,128,3,1,1),(128,8,256,256,3,1,1),(128,64,64,64,3,1,1),(128,16,256,256,3,1,1),(128,32,64,128,3,2,1),(128,32,128,64,1,1,0)]
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
    x=torch.randn(N,H,H,Ci,device='cuda'); gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda'); gw=torch.zeros(Co,k,k,Ci,device='cuda')
    fl=K.conv_flops(d); out=[]
    for tgt in (256,512,1024,2048,4096):
        lib.bh_debug_force_tile(-3,tgt)
        out.append('%d:%.0f'%(tgt, fl/bench(lambda: K.conv_wgrad(x,gy,gw,None,d))/1e9))
    print((N,H,Ci,Co,k,s),' '.join(out),flush=True)
