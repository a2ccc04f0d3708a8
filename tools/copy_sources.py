"""Where do the device-to-device copies of a training step come from?  (torch.profiler with stacks, one step of configs[1])"""
import sys; sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
import bench as B
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_optimizer, train_step

cfg = configs.get("zeng-bihome")
model = build_model(cfg, "cuda")
opt, sched = build_optimizer(model, cfg["SOLVER"])
d = synth.make_pairs(64, seed=1)
data = {k: torch.tensor(v).cuda() for k, v in d.items() if k in B.KEYS} if hasattr(B, "KEYS") else {k: torch.tensor(v).cuda() for k, v in d.items()}
for _ in range(3):
    train_step(model, dict(data), opt, sched, loss_fn="biHomE")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(model, dict(data), opt, sched, loss_fn="biHomE")
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::zeros", "aten::fill_", "aten::zero_"):
        st = [s for s in ev.stack if "bihome_amd" in s or "bench" in s][:2]
        c[(ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:60] for s in st))] += 1
for k, v in c.most_common(60):
    print(v, k)
