import sys; sys.path.insert(0,'.')
import numpy as np, torch
from bihome_amd import configs, synth
from bihome_amd.weights import load_synthetic
from oracle import bihome_oracle as O
cfg=configs.get('zeng-bihome')
B=int(sys.argv[1]) if len(sys.argv)>1 else 2
d=synth.make_pairs(B,seed=3)
rng=np.random.Generator(np.random.PCG64(0))
g12,g21=rng.standard_normal((B,2,128,128)),rng.standard_normal((B,2,128,128))
grads={}
for dt in (torch.float32, torch.float64):
    bb,_=O.build(cfg); load_synthetic(bb,0); bb.to(dt).train()
    out=bb({k:torch.tensor(d[k],dtype=dt) for k in ('patch_1','patch_2')})
    ((out['pf_hat_12']*torch.tensor(g12,dtype=dt)).sum()+(out['pf_hat_21']*torch.tensor(g21,dtype=dt)).sum()).backward()
    grads[dt]={n:p.grad.double().clone() for n,p in bb.named_parameters()}
    grads[(dt,'out')]=out['pf_hat_12'].detach().double()
print('out relerr', ((grads[(torch.float32,'out')]-grads[(torch.float64,'out')]).abs().max()/grads[(torch.float64,'out')].abs().max()).item())
errs=[]
for n in grads[torch.float32]:
    a,b=grads[torch.float32][n],grads[torch.float64][n]
    errs.append(((a-b).abs().max()/(b.abs().max()+1e-30)).item())
errs=np.array(errs); names=list(grads[torch.float32])
idx=np.argsort(-errs)[:10]
for i in idx: print(names[i], errs[i], grads[torch.float64][names[i]].abs().max().item())
