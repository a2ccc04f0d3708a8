"""Per-layer launch table of one training step (HIP-event pairs per launch, keyed by kernel and geometry).
Usage: python tools/step_detail.py [config] [batch]"""
import os, sys; sys.path.insert(0, '.')
os.environ.setdefault('BIHOME_OVERLAP', '0')      # per-launch event timing: every kernel alone on the GPU
import torch
from bihome_amd import configs, kernels as K, synth
from bihome_amd.step import build_model, build_optimizer, train_step
from bihome_amd.weights import load_synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "zeng-bihome"
cfg = configs.get(name)
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["DATA"]["BATCH_SIZE"] if "DATA" in cfg else 64
model = build_model(cfg)
load_synthetic(model[0], 0)
opt, sched = build_optimizer(model, cfg["SOLVER"])
d = synth.make_pairs(B, seed=1)
data = {k: torch.as_tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
for _ in range(3):
    train_step(model, dict(data), opt, sched)
K.TIMING_DETAIL = True
K.TIMING = {}
train_step(model, dict(data), opt, sched)
torch.cuda.synchronize()
K.TIMING = {}
n = 3
for _ in range(n):
    train_step(model, dict(data), opt, sched)
torch.cuda.synchronize()
rows = []
for k, r in K.TIMING.items():
    ms = sum(a.elapsed_time(b) for a, b in r["events"])
    rows.append((ms / n, r["n"] // n, k, r["flops"] / max(ms, 1e-9) / 1e9, r["bytes"] / max(ms, 1e-9) / 1e6))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total timed %.2f ms/step" % tot)
for ms, cnt, k, tf, gbs in rows[:70]:
    print("%7.3f ms %3d x %7.1f us  %6.1f TF %6.0f GB/s  %s" % (ms, cnt, 1e3 * ms / max(cnt, 1), tf, gbs, k))
