"""Implicit-GEMM conv micro-benchmark: TFLOP/s of fwd/dgrad per forced tile (bh_debug_force_tile) and of wgrad, on the network's main 3x3 shapes."""
import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
shapes=[(128,32,64,64,3,1,1),(128,16,128,128,3,1,1),(128,8,256,256,3,1,1),(128,64,64,64,3,1,1),(128,128,32,32,3,1,1),(128,16,256,256,3,1,1)]
tiles=[(0,0),(64,128),(64,64),(6464,0),(64128,0)]
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
    gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda')
    fl=K.conv_flops(d)
    out=[]
    for (bm,bn) in tiles:
        lib.bh_debug_force_tile(bm,bn)
        try:
            tf=bench(lambda: K.conv_fwd(x,w,None,d)); td=bench(lambda: K.conv_dgrad(gy,w,d))
            out.append('%s f%.0f d%.0f'%((bm,bn), fl/tf/1e9, fl/td/1e9))
        except Exception as e:
            out.append('%s n/a'%((bm,bn),))
    lib.bh_debug_force_tile(0,0)
    gw=torch.zeros_like(w)
    tw=bench(lambda: K.conv_wgrad(x,gy,gw,None,d))
    print((N,H,Ci,Co,k), ' | '.join(out), '| wgrad %.0f'%(fl/tw/1e9), flush=True)
