import sys, time, os; sys.path.insert(0,'.')
import torch
from bihome_amd import configs, synth
from bihome_amd.weights import load_synthetic
from oracle import bihome_oracle as O
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
cfg=configs.get('zeng-bihome')
for th in (8,16,32,64):
    torch.set_num_threads(th)
    bb,head=O.build(cfg); load_synthetic(bb,0); load_synthetic(head.auxiliary_resnet,0)
    opt,sched=O.make_optimizer(torch.nn.Sequential(bb,head),cfg['SOLVER'])
    d=synth.make_pairs(8,seed=42)
    ts=[]
    for it in range(3):
        data={k:torch.tensor(d[k]) for k in ('patch_1','patch_2','delta')}
        t0=time.perf_counter(); O.train_step(bb,head,opt,sched,data); ts.append(time.perf_counter()-t0)
        if ts[-1]>60: break
    print(th, ts, flush=True)
