"""ConvTranspose2d(2,2) forward / dgrad on the decoder shapes per forced implicit-GEMM tile (bh_debug_force_tile)."""
import sys; sys.path.insert(0, '.')
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

for (N, H, Ci, Co) in [(128, 64, 32, 32), (128, 64, 32, 16), (128, 32, 64, 64), (128, 32, 64, 32), (128, 16, 128, 128), (128, 8, 256, 256)]:
    d = K.conv_desc(N, H, H, Ci, Co, 2, 2, 0, transposed=True)
    x = torch.randn(N, H, H, Ci, device='cuda'); w = torch.randn(Ci, 2, 2, Co, device='cuda') * 0.1
    gy = torch.randn(N, 2 * H, 2 * H, Co, device='cuda')
    mb = 4.0 * (x.numel() + gy.numel()) / 1e6
    out = []
    for (bm, bn) in [(0, 0), (128, 128), (128, 64), (64, 128), (64, 64), (128, 32)]:
        lib.bh_debug_force_tile(bm, bn)
        try:
            tf = bench(lambda: K.conv_fwd(x, w, None, d)); td = bench(lambda: K.conv_dgrad(gy, w, d))
            out.append('%s f%.0fus(%.1fTB/s) d%.0fus' % ((bm, bn), tf * 1e3, mb / tf / 1e6 * 1e3 / 1e3, td * 1e3))
        except Exception as e:
            out.append('%s n/a' % ((bm, bn),))
    lib.bh_debug_force_tile(0, 0)
    print((N, H, Ci, Co), '%.0f MB' % mb, ' | '.join(out), flush=True)
