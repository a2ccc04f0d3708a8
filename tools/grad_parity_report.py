"""Per-parameter gradient parity table at a bench-size batch: HIP path vs the float64 oracle, next to the oracle's own
float32-vs-float64 differences (dev tool; the assertions live in tests/test_fullsize_gpu.py).
    python tools/grad_parity_report.py [zeng-bihome|detone-bihome] [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bihome_amd import configs, synth                      # noqa: E402
from bihome_amd.step import build_model                    # noqa: E402
from bihome_amd.weights import load_synthetic              # noqa: E402
from oracle import bihome_oracle as O                      # noqa: E402
import test_fullsize_gpu as T                              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "zeng-bihome"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = configs.get(name)
d = synth.make_pairs(B, seed=64)
g = torch.Generator().manual_seed(64)
ch = [O.sample_choice(128 * 128, B * 128, g).reshape(B, 128) for _ in range(2)] if name.startswith("zeng") else None
r64 = T._oracle_step(cfg, d, torch.float64, ch)
r32 = T._oracle_step(cfg, d, torch.float32, ch)
model = build_model(cfg)
load_synthetic(model[0], 0)
load_synthetic(model[1].auxiliary_resnet, 0)
model.train()
data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
if ch is not None:
    data["choice_12"], data["choice_21"] = ch[0].cuda(), ch[1].cuda()
loss, dgt, dh = model(data)
loss.backward()
torch.cuda.synchronize()
print("loss hip %.8f f32 %.8f f64 %.8f" % (loss.item(), r32["loss"], r64["loss"]))
print("%-44s %10s %10s %10s %10s %10s" % ("param", "|g64|", "hip relL2", "f32 relL2", "hip dnorm", "f32 dnorm"))
for n, p in model[0].named_parameters():
    r = r64["grads"][n]
    rn = r.norm().item()
    if rn < 1e-12:
        continue
    got = p.grad.detach().cpu().double()
    f32 = r32["grads"][n]
    print("%-44s %10.3e %10.2e %10.2e %10.2e %10.2e" % (n, rn, (got - r).norm().item() / rn, (f32 - r).norm().item() / rn,
                                                         (got.norm().item() - rn) / rn, (f32.norm().item() - rn) / rn))
