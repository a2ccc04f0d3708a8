"""Fixed (launch + prologue + epilogue) cost of the packed 3x3 kernels, fp32-MFMA and f32x3 form: time with 0, 1, 2, all
channel chunks (BIHOME_TUNING=1 build).  Usage: BIHOME_TUNING=1 python tools/x3_fixed_cost.py"""
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
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec)
        x = torch.randn(N, H, H, Ci, device='cuda')
        w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
        wk = w.permute(0, 2, 3, 1)
        pk = K.WeightPacker(split=prec == 2); pf, pd = pk.get(w); pk.refresh()
        out = []
        for n in (0, 1, 2, -1):
            lib.bh_debug_force_tile(-8, n)
            out.append('%d chunks: %.1f us' % (n, bench(lambda: K.conv_fwd(x, wk, None, d, wpacked=pf))))
        lib.bh_debug_force_tile(-8, -1)
        print((N, H, Ci, Co), "prec", prec, ' | '.join(out), flush=True)
