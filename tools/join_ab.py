"""Round 5: adjoint of the two-branch BatchNorm join with the ReLU mask read from y (BIHOME_JOIN_REMASK=0) or recomputed (default)."""
import os, sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, H, C) in [(128, 128, 16), (128, 64, 32), (128, 32, 64), (128, 16, 128)]:
    groups = 2
    xa, xb = torch.randn(N, H, H, C, device="cuda"), torch.randn(N, H, H, C, device="cuda")
    ma, mb = torch.nn.BatchNorm2d(C).cuda(), torch.nn.BatchNorm2d(C).cuda()
    sa, sb = K.bn_stats_buffer(groups, C, "cuda"), K.bn_stats_buffer(groups, C, "cuda")
    K.bn_stats(xa, sa, groups, C); K.bn_stats(xb, sb, groups, C)
    y = K.bn_join_fwd(xa, xb, ma, mb, sa, sb, groups, True, 0.1, 0.1)
    gy = torch.randn_like(xa)
    res = {}
    outs = {}
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ["BIHOME_JOIN_REMASK"] = mode
            outs[mode] = K.bn_join_bwd(gy, y, xa, xb, ma, mb, sa, sb, groups, True, False, False)
            res.setdefault(mode, []).append(bench(lambda: K.bn_join_bwd(gy, y, xa, xb, ma, mb, sa, sb, groups, True, False, False)))
    print((N, H, C), "same:", torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1]),
          " reads y", " ".join("%.1f" % v for v in res["0"]), " recomputes", " ".join("%.1f" % v for v in res["1"]), "us", flush=True)
