"""f32x3 wgrad kernel: workgroups per launch (bh_debug_force_tile(-30, n)) and the cost of its atomics flush (-31).  BIHOME_TUNING=1."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
from wx3_check import bench
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64)]:
    x = torch.randn(N, H, H, Ci, device="cuda"); gy = torch.randn(N, H, H, Co, device="cuda")
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2)
    gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    out = []
    for tgt in (128, 256, 512):
        lib.bh_debug_force_tile(-30, tgt)
        t1 = bench(lambda: K.conv_wgrad(x, gy, gw, None, d))
        lib.bh_debug_force_tile(-31, 1)
        t0 = bench(lambda: K.conv_wgrad(x, gy, gw, None, d))
        lib.bh_debug_force_tile(-31, 0)
        out.append("target %d: %.1f (no flush %.1f)" % (tgt, t1, t0))
    lib.bh_debug_force_tile(-30, 256)
    print((N, H, Ci, Co), " | ".join(out), flush=True)
