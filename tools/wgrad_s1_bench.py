"""wgrad_kernel (generic loader) vs wgrad_s1_kernel (stride-1 fast path): python tools/wgrad_s1_bench.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for (N, H, Ci, Co, k) in [(128, 32, 64, 64, 3), (128, 16, 128, 128, 3), (128, 8, 256, 256, 3), (128, 64, 64, 64, 3), (128, 32, 128, 128, 3),
                          (128, 16, 256, 256, 3), (128, 4, 512, 512, 3), (6, 8, 64, 128, 3), (128, 16, 128, 64, 1), (2, 16, 64, 64, 5)]:
    d = K.conv_desc(N, H, H, Ci, Co, k, 1, k // 2)
    x = torch.randn(N, H, H, Ci, device='cuda'); gy = torch.randn(N, H, H, Co, device='cuda')
    fl = K.conv_flops(d)
    lib.bh_debug_force_tile(-16, 0)
    ref = torch.zeros(Co, k, k, Ci, device='cuda'); K.conv_wgrad(x, gy, ref, None, d)
    gw = torch.zeros_like(ref)
    t0 = bench(lambda: K.conv_wgrad(x, gy, gw, None, d))
    out = ['generic %.0f us %.0f TF' % (t0 * 1e3, fl / t0 / 1e9)]
    for mode, hook, tgts in ((1, -17, (2048,)), (3, -19, (512, 768, 1024, 1536))):
        for tg in tgts:
            lib.bh_debug_force_tile(-16, mode); lib.bh_debug_force_tile(hook, tg)
            g2 = torch.zeros_like(ref); K.conv_wgrad(x, gy, g2, None, d)
            err = ((g2 - ref).abs().max() / ref.abs().max()).item()
            t1 = bench(lambda: K.conv_wgrad(x, gy, gw, None, d))
            out.append('nt%d/%d %.0f us %.0f TF (%.0e)' % (mode, tg, t1 * 1e3, fl / t1 / 1e9, err))
    lib.bh_debug_force_tile(-16, 1); lib.bh_debug_force_tile(-17, 2048); lib.bh_debug_force_tile(-19, 768)
    print((N, H, Ci, Co, k), ' | '.join(out), flush=True)
