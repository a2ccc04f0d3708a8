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
for prec in (0,1):
  for (N,H,Ci,Co,k,s,p) in [(128,4,512,512,3,1,1),(128,8,256,512,3,2,1),(128,8,256,512,1,2,0)]:
    d=K.conv_desc(N,H,H,Ci,Co,k,s,p,precision=prec)
    x=torch.randn(N,H,H,Ci,device='cuda'); w=torch.randn(Co,k,k,Ci,device='cuda')*0.05
    gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda'); gw=torch.zeros_like(w)
    fl=K.conv_flops(d); out=[]
    for (bm,bn) in [(0,0),(128,128),(128,64),(64,128),(64,64),(128,32)]:
        lib.bh_debug_force_tile(bm,bn)
        try:
            tf=bench(lambda: K.conv_fwd(x,w,None,d)); td=bench(lambda: K.conv_dgrad(gy,w,d))
            out.append('%s f%.0fus d%.0fus'%((bm,bn), tf*1e3, td*1e3))
        except Exception as e:
            out.append('%s n/a'%((bm,bn),))
    lib.bh_debug_force_tile(0,0)
    tw=bench(lambda: K.conv_wgrad(x,gy,gw,None,d))
    print('prec',prec,(N,H,Ci,Co,k,s),' | '.join(out),'| wgrad %.0fus'%(tw*1e3),flush=True)
