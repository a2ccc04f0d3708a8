"""Start-stagger experiment for the halo-tiled 3x3 kernel: every second first-round workgroup starts late."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 64, 64, 64), (128, 8, 256, 256)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    x = torch.randn(N, H, H, Ci, device='cuda'); w = torch.randn(Co, 3, 3, Ci, device='cuda') * 0.05
    gy = torch.randn(N, H, H, Co, device='cuda')
    fl = K.conv_flops(d); out = []
    for mode in (0, 1, 2):
        for sl in ((0,) if mode == 0 else (1, 2, 4)):
            lib.bh_debug_force_tile(-12, mode); lib.bh_debug_force_tile(-13, sl)
            tf = bench(lambda: K.conv_fwd(x, w, None, d)); td = bench(lambda: K.conv_dgrad(gy, w, d))
            out.append('m%d/s%d f%.0f d%.0fus' % (mode, sl, tf, td))
    lib.bh_debug_force_tile(-12, 0)
    print((N, H, Ci, Co), ' | '.join(out), flush=True)
