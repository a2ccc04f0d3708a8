"""Fixed (launch + prologue + epilogue) cost of the halo-tiled 3x3 kernel: time with 0, 1, 2, ... channel chunks."""
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
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    x = torch.randn(N, H, H, Ci, device='cuda'); w = torch.randn(Co, 3, 3, Ci, device='cuda') * 0.05
    out = []
    for n in (0, 1, 2, -1):
        lib.bh_debug_force_tile(-8, n)
        out.append('%d chunks: %.1f us' % (n, bench(lambda: K.conv_fwd(x, w, None, d))))
    lib.bh_debug_force_tile(-8, -1)
    print((N, H, Ci, Co), ' | '.join(out), flush=True)
