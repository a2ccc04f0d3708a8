"""Stride-2 3x3 dgrad: the four parity classes as one launch (blockIdx.z = class) against four launches - bitwise equal, and the time of both.
BIHOME_TUNING=1 python tools/s2_multi_ab.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib, TUNING
assert TUNING, "needs the tuning library (make -C bihome_amd/csrc tuning; BIHOME_TUNING=1)"
for (N, Hi, Ci, Co) in [(128, 32, 64, 128), (128, 16, 128, 256), (6, 24, 32, 64)]:
    g = torch.Generator().manual_seed(N + Hi)
    Ho = Hi // 2
    gy = torch.randn(N, Ho, Ho, Co, generator=g).cuda()
    w = (torch.randn(Co, 3, 3, Ci, generator=g) * 0.05).cuda()
    d = K.conv_desc(N, Hi, Hi, Ci, Co, 3, 2, 1)
    old = torch.randn(N, Hi, Hi, Ci, generator=g).cuda()
    res = {}
    for mode in (0, 1):
        lib.bh_debug_force_tile(-37, mode)
        a = K.conv_dgrad(gy, w, d)
        b = old.clone(); K.conv_dgrad(gy, w, d, out=b)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(5): K.conv_dgrad(gy, w, d)
        ev[0].record()
        for _ in range(50): K.conv_dgrad(gy, w, d)
        ev[1].record(); torch.cuda.synchronize()
        res[mode] = (a, b, ev[0].elapsed_time(ev[1]) / 50 * 1e3)
    lib.bh_debug_force_tile(-37, 1)
    ref = torch.nn.functional.conv_transpose2d(gy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1)
    print((N, Hi, Ci, Co), "equal", torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1]),
          "err vs f64 %.2e" % ((res[1][0].double() - ref).abs().max().item() / ref.abs().max().item()),
          "us: four launches %.1f, one %.1f" % (res[0][2], res[1][2]))
