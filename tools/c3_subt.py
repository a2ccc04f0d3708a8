"""Round 4: f16x2 3x3 kernels, one vs two 8x8 sub-tiles per workgroup (-9, 1|2) and tile positions per workgroup (-12, 1|2).  BIHOME_TUNING=1"""
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
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4)
    x = torch.randn(N, H, H, Ci, device='cuda')
    gy = torch.randn(N, H, H, Co, device='cuda')
    w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
    for name, fn in (("fwd", lambda: K.conv_fwd(x, wk, None, d, wpacked=pf)), ("dgrad", lambda: K.conv_dgrad(gy, wk, d, wpacked=pd))):
        out = []
        for subt in (2, 1, 3):
            for tpb in (1, 2):
                lib.bh_debug_force_tile(-9, subt); lib.bh_debug_force_tile(-12, tpb)
                K._VARIANT_CACHE.clear()
                out.append('subt%d tpb%d: %.1f' % (subt, tpb, bench(fn)))
        lib.bh_debug_force_tile(-9, 2); lib.bh_debug_force_tile(-12, 2)
        print((N, H, Ci, Co), name, ' | '.join(out), flush=True)
