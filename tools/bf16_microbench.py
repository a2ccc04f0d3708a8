"""bf16-operand mode micro-benchmark: error vs the fp32 path and TFLOP/s of fwd/dgrad/wgrad."""
import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
shapes=[(128,32,64,64,3,1,1),(128,16,128,128,3,1,1),(128,8,256,256,3,1,1),(128,64,64,64,3,1,1),(128,128,32,32,3,1,1),(128,16,256,256,3,1,1),(128,32,64,128,3,2,1)]
tiles=[(0,0)]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for (N,H,Ci,Co,k,s,p) in shapes:
    d=K.conv_desc(N,H,H,Ci,Co,k,s,p); d16=K.conv_desc(N,H,H,Ci,Co,k,s,p,precision=1)
    x=torch.randn(N,H,H,Ci,device='cuda'); w=torch.randn(Co,k,k,Ci,device='cuda')*0.05
    gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda')
    y=K.conv_fwd(x,w,None,d); y16=K.conv_fwd(x,w,None,d16)
    g=K.conv_dgrad(gy,w,d); g16=K.conv_dgrad(gy,w,d16)
    ef=((y-y16).abs().max()/y.abs().max()).item(); ed=((g-g16).abs().max()/g.abs().max()).item()
    fl=K.conv_flops(d); byt=4.0*(x.numel()+y.numel())
    out=[]
    for (bm,bn) in tiles:
        lib.bh_debug_force_tile(bm,bn)
        try:
            tf=bench(lambda: K.conv_fwd(x,w,None,d16)); td=bench(lambda: K.conv_dgrad(gy,w,d16))
            out.append('%s f%.0fTF/%.0fGB/s d%.0fTF'%((bm,bn), fl/tf/1e9, byt/tf/1e6, fl/td/1e9))
        except Exception as e:
            out.append('%s n/a'%((bm,bn),))
    lib.bh_debug_force_tile(0,0)
    print((N,H,Ci,Co,k,s), 'relerr f %.1e d %.1e |'%(ef,ed), ' | '.join(out), flush=True)
print('--- wgrad')
for (N,H,Ci,Co,k,s,p) in shapes:
    d=K.conv_desc(N,H,H,Ci,Co,k,s,p); d16=K.conv_desc(N,H,H,Ci,Co,k,s,p,precision=1)
    x=torch.randn(N,H,H,Ci,device='cuda'); gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda')
    g0=torch.zeros(Co,k,k,Ci,device='cuda'); g1=torch.zeros_like(g0)
    K.conv_wgrad(x,gy,g0,None,d); K.conv_wgrad(x,gy,g1,None,d16)
    e=((g0-g1).abs().max()/g0.abs().max()).item()
    fl=K.conv_flops(d)
    t0=bench(lambda: K.conv_wgrad(x,gy,g0,None,d)); t1=bench(lambda: K.conv_wgrad(x,gy,g1,None,d16))
    print((N,H,Ci,Co,k,s),'relerr %.1e f32 %.0fTF bf16 %.0fTF'%(e, fl/t0/1e9, fl/t1/1e9), flush=True)
