"""Where the f32x3 3x3 loop's time goes: the kernel without its in-loop weight loads (1), halo staging (2), fragment reads (4)
- wrong results, timing only.  BIHOME_TUNING=1 python tools/x3_ablate.py"""
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
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 64, 64, 64)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2)
    x = torch.randn(N, H, H, Ci, device='cuda')
    w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.WeightPacker(split=True); pf, pd = pk.get(w); pk.refresh()
    out = []
    for bits in (0, 1, 2, 4, 3, 5, 6, 7):
        lib.bh_debug_force_tile(-18, bits)
        out.append('%d: %.1f' % (bits, bench(lambda: K.conv_fwd(x, wk, None, d, wpacked=pf))))
    lib.bh_debug_force_tile(-18, 0)
    print((N, H, Ci, Co), ' | '.join(out), flush=True)
