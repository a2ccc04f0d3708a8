"""Per-launch (per layer) HIP-event timing table of one Zeng+biHomE training step at bs 64: python tools/layer_times.py"""
import sys, json; sys.path.insert(0,'.')
import torch
from bihome_amd import configs, synth, kernels as K
from bihome_amd.step import build_model, build_optimizer, train_step
from bihome_amd.weights import load_synthetic
cfg=configs.get('zeng-bihome'); model=build_model(cfg); load_synthetic(model[0],0); load_synthetic(model[1].auxiliary_resnet,0)
opt,sched=build_optimizer(model,cfg['SOLVER'])
d=synth.make_pairs(64,seed=42); data={k:torch.tensor(d[k]).cuda() for k in ('patch_1','patch_2','delta')}
for _ in range(3): train_step(model,dict(data),opt,sched)
K.TIMING_DETAIL=True; K.TIMING={}
n=2
for _ in range(n): train_step(model,dict(data),opt,sched)
torch.cuda.synchronize()
rows=[]
for name,r in K.TIMING.items():
    ms=sum(a.elapsed_time(b) for a,b in r['events'])/n
    rows.append((ms, r['n']//n, r['flops']/n/(ms*1e-3)/1e12 if ms>0 else 0, r['bytes']/n/(ms*1e-3)/1e9 if ms>0 else 0, name))
rows.sort(reverse=True)
tot=sum(r[0] for r in rows)
print('total timed ms/step', tot)
for r in rows[:70]: print('%8.3f ms x%d %7.1f TF %7.0f GB/s  %s'%r)
