"""Round 5: the 7x7/2 stem forward kernels alone (statistics epilogue on), against torch float64 on a small case."""
import sys; sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from bihome_amd import kernels as K


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for Ci in (1, 2):
    N = 128
    x = torch.randn(N, Ci, 128, 128, device="cuda")
    w = torch.randn(64, 7, 7, Ci, device="cuda") * 0.1
    d = K.conv_desc(N, 128, 128, Ci, 64, 7, 2, 3, in_nchw=True, precision=4)
    s = K.bn_stats_buffer(2, 64, "cuda")
    y = K.conv_fwd(x, w, None, d, bn_sums=s, groups=2)
    ref = F.conv2d(x[:4].double().cpu(), w.double().cpu().permute(0, 3, 1, 2), None, 2, 3).permute(0, 2, 3, 1)
    err = ((y[:4].double().cpu() - ref).norm() / ref.norm()).item()
    print(Ci, K.conv_variant(d, "fwd", bn_groups=2), "rel err %.1e" % err, " %.1f %.1f us" % (bench(lambda: K.conv_fwd(x, w, None, d, bn_sums=s, groups=2)), bench(lambda: K.conv_fwd(x, w, None, d, bn_sums=s, groups=2))), flush=True)
