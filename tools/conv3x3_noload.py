"""3x3 halo kernel with its in-loop DMA removed (timing only): python tools/conv3x3_noload.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
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
    return a.elapsed_time(b) / n
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 32, 128, 128)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    x = torch.randn(N, H, H, Ci, device='cuda'); w = torch.randn(Co, 3, 3, Ci, device='cuda') * 0.05
    y = torch.empty(N, H, H, Co, device='cuda')
    fl = K.conv_flops(d); out = []
    for bits in (0, 1, 2, 3):
        lib.bh_debug_force_tile(-18, bits)
        t = bench(lambda: K.conv_fwd(x, w, None, d))
        out.append('noload=%d: %.1f us %.0f TF' % (bits, t * 1e3, fl / t / 1e9))
    lib.bh_debug_force_tile(-18, 0)
    print((N, H, Ci, Co), ' | '.join(out), flush=True)
