"""Which Python lines issue the residual torch launches of a training step (aten::copy_ / fill_ / cat / ...)?  One profiled step with
stacks, grouped by op and innermost repository frame.  Usage: python tools/torch_op_sources.py"""
import collections, os, sys; sys.path.insert(0, '.')
import torch
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_optimizer, train_step
from bihome_amd.weights import load_synthetic
cfg = configs.get("zeng-bihome")
model = build_model(cfg)
load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0)
opt, sched = build_optimizer(model, cfg["SOLVER"])
d = synth.make_pairs(64, seed=1)
data = {k: torch.as_tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
for _ in range(3):
    train_step(model, dict(data), opt, sched)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_step(model, dict(data), opt, sched)
    torch.cuda.synchronize()
root = os.path.abspath('.')
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul", "aten::add", "aten::add_", "aten::neg", "aten::rsqrt", "aten::sum",
                  "aten::multinomial", "aten::arange", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros", "aten::zeros_like"):
        fr = "?"
        for s in (e.stack or []):
            if root in s or "bihome_amd" in s or "bench" in s:
                fr = s.replace(root + "/", "")
                break
        cnt[(e.name, fr)] += 1
for (n, fr), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print("%3d  %-18s %s" % (c, n, fr))
