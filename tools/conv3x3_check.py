"""Halo-tiled 3x3 kernel (csrc/conv3x3.hip) against the generic implicit-GEMM kernel: max abs difference and TFLOP/s."""
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

shapes = [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64), (3, 8, 32, 64), (5, 24, 64, 128),
          (128, 32, 128, 128), (128, 16, 256, 256), (128, 64, 32, 32), (128, 128, 32, 32), (5, 24, 32, 32), (7, 8, 64, 96)]
PREC = 1 if '--bf16' in sys.argv else 0
if '--subt1' in sys.argv:
    lib.bh_debug_force_tile(-9, 1)
lib.bh_debug_force_tile(-5, 1)
for (N, H, Ci, Co) in shapes:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=PREC)
    x = torch.randn(N, H, H, Ci, device='cuda'); w = torch.randn(Co, 3, 3, Ci, device='cuda') * 0.05
    bias = torch.randn(Co, device='cuda')
    gy = torch.randn(N, H, H, Co, device='cuda')
    fl = K.conv_flops(d)
    res = {}
    for mode in (1, 0):            # 1: generic kernel, 0: halo kernel
        lib.bh_debug_force_tile(-4, mode)
        y = K.conv_fwd(x, w, bias, d)
        gx = K.conv_dgrad(gy, w, d)
        acc = torch.ones(N, H, H, Ci, device='cuda')
        K.conv_dgrad(gy, w, d, out=acc)
        tf = bench(lambda: K.conv_fwd(x, w, bias, d)); td = bench(lambda: K.conv_dgrad(gy, w, d))
        res[mode] = (y, gx, acc, fl / tf / 1e9, fl / td / 1e9)
    lib.bh_debug_force_tile(-4, 0)
    e = [float((res[0][i] - res[1][i]).abs().max()) for i in range(3)]
    s = [float(res[1][i].abs().max()) for i in range(3)]
    print((N, H, Ci, Co), "maxdiff fwd %.2e/%.1f dgrad %.2e/%.1f acc %.2e | generic f%.0f d%.0f | halo f%.0f d%.0f TF" % (
        e[0], s[0], e[1], s[1], e[2], res[1][3], res[1][4], res[0][3], res[0][4]), flush=True)
