import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for (N,H,Ci,Co) in [(128,32,64,64),(128,16,128,128),(128,8,256,256)]:
    d=K.conv_desc(N,H,H,Ci,Co,3,1,1)
    x=torch.randn(N,H,H,Ci,device='cuda'); gy=torch.randn(N,H,H,Co,device='cuda'); gw=torch.zeros(Co,3,3,Ci,device='cuda')
    fl=K.conv_flops(d)
    out=[]
    for xm in (0,1):
        lib.bh_debug_force_tile(-10,xm)
        t=bench(lambda: K.conv_wgrad(x,gy,gw,None,d)); out.append('xcdmap%d: %.0fus %.0fTF'%(xm,t*1e3,fl/t/1e9))
    lib.bh_debug_force_tile(-10,0)
    for nf in (0,1):
        lib.bh_debug_force_tile(-7,nf)
        for tgt in (4096,2048,1024):
            lib.bh_debug_force_tile(-3,tgt)
            t=bench(lambda: K.conv_wgrad(x,gy,gw,None,d))
            out.append('nf%d/%d: %.0fus %.0fTF'%(nf,tgt,t*1e3,fl/t/1e9))
    lib.bh_debug_force_tile(-7,0); lib.bh_debug_force_tile(-3,4096)
    print((N,H,Ci,Co),' | '.join(out),flush=True)
