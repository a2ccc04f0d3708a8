import sys; sys.path.insert(0, '.')
import numpy as np, torch
from bihome_amd import kernels as K
torch.manual_seed(0)
for (groups,N,H,C) in [(2,2,32,16),(1,2,32,16),(2,2,8,64),(2,2,32,64)]:
    x = (torch.randn(groups*N,H,H,C,device='cuda')*2+0.5)
    res = torch.randn_like(x)
    gy = torch.randn_like(x)
    gm = torch.rand(C,device='cuda')+0.5; bt = torch.randn(C,device='cuda')
    rm = torch.zeros(C,device='cuda'); rv=torch.ones(C,device='cuda')
    y, st = K.bn_fwd(x, gm, bt, rm, rv, res, groups, 1e-5, 0.1, True, True)
    gg, gb = torch.zeros(C,device='cuda'), torch.zeros(C,device='cuda')
    gx, gres = K.bn_bwd(gy, y, x, gm, st, rm, rv, groups, 1e-5, True, True, True, gg, gb)
    xr = x.double().requires_grad_(True); rr = res.double().requires_grad_(True)
    outs=[]
    for g in range(groups):
        xs = xr[g*N:(g+1)*N]
        m = xs.mean((0,1,2)); v = xs.var((0,1,2),unbiased=False)
        outs.append(torch.relu((xs-m)/torch.sqrt(v+1e-5)*gm.double()+bt.double() + rr[g*N:(g+1)*N]))
    ref = torch.cat(outs,0)
    (ref*gy.double()).sum().backward()
    err = (gx.double()-xr.grad).abs()
    mism = ((y>0) != (ref>0)).sum().item()
    e2 = (gres.double()-rr.grad).abs()
    print((groups,N,H,C), 'y err', (y.double()-ref).abs().max().item(), 'gx err', err.max().item(), 'n bad', (err>1e-4).sum().item(), 'mask mismatch', mism, 'gres err', e2.max().item(), (e2>1e-4).sum().item())
    bad = (err>1e-4).nonzero()[:8].cpu().numpy(); print(bad)
