"""Round 4: torch's fused Adam over the backbone's ~190 parameter tensors against the same update over ONE flat tensor (same elements, same math).
python tools/adam_flat_ab.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import configs
from bihome_amd.step import build_model
cfg = configs.get("zeng-bihome")
model = build_model(cfg)
ps = [p for p in model.parameters() if p.requires_grad]
n = sum(p.numel() for p in ps)
print(len(ps), 'tensors', n, 'elements')
for p in ps: p.grad = torch.randn_like(p)
flat = torch.nn.Parameter(torch.randn(n, device='cuda')); flat.grad = torch.randn(n, device='cuda')
oa = torch.optim.Adam(ps, lr=1e-3, fused=True); ob = torch.optim.Adam([flat], lr=1e-3, fused=True)
def t(opt, k=50):
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): opt.step()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
import time
for r in range(3):
    t0 = time.perf_counter(); ta = t(oa); t1 = time.perf_counter(); tb = t(ob); t2 = time.perf_counter()
    print('per-tensor %.1f us (host %.0f us/step)   flat %.1f us (host %.0f us/step)' % (ta, (t1 - t0) / 55 * 1e6, tb, (t2 - t1) / 55 * 1e6))
