"""Sweep of the BatchNorm apply launch width (bh_debug_force_tile(-20, n), -DBH_TUNING build) on the step's dominant BN
shape (128 images x 32 x 32 x 64 channels, 2 groups): forward apply (+ReLU, +residual) and backward apply, producer ->
consumer (the input was just written, as in the step) and cold (rotating over 12 buffer sets).
    BIHOME_TUNING=1 python tools/bn_apply_sweep.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib

dev = "cuda"
shapes = [(128, 32, 64), (128, 16, 128), (128, 64, 32), (128, 128, 16)]
for N, H, C in shapes:
    NSET = 10
    xs = [torch.randn(N, H, H, C, device=dev) for _ in range(NSET)]
    rs = [torch.randn(N, H, H, C, device=dev) for _ in range(NSET)]
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st = K.bn_stats_buffer(2, C, dev)
    K.bn_fwd(xs[0], gm, bt, rm, rv, None, 2, 1e-5, 0.1, True, True, stats=st)          # fills st
    nbytes = xs[0].numel() * 4

    def timeit(fn, n=200):
        for i in range(20): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n): fn(i)
        b.record(); torch.cuda.synchronize()
        return 1e3 * a.elapsed_time(b) / n

    print("shape N%d %dx%d C%d (%.1f MB per tensor)" % (N, H, H, C, nbytes / 1e6))
    for cap in (256, 512, 768, 1024, 1536, 2048, 4096):
        lib.bh_debug_force_tile(-20, cap)
        f_cold = timeit(lambda i: K.bn_fwd(xs[i % NSET], gm, bt, rm, rv, None, 2, 1e-5, 0.1, True, True, stats=st, stats_ready=True))
        f_res = timeit(lambda i: K.bn_fwd(xs[i % NSET], gm, bt, rm, rv, rs[i % NSET], 2, 1e-5, 0.1, True, True, stats=st, stats_ready=True))
        f_hot = timeit(lambda i: K.bn_fwd(xs[0], gm, bt, rm, rv, None, 2, 1e-5, 0.1, True, True, stats=st, stats_ready=True))
        sums = K.bn_stats_buffer(2, C, dev)
        b_cold = timeit(lambda i: K.bn_bwd(rs[i % NSET], None, xs[i % NSET], gm, st, rm, rv, 2, 1e-5, True, True, False, beta=bt, had_res=False, sums_ready=sums))
        print("  cap %4d: fwd cold %6.2f us (%5.0f GB/s)  fwd+res cold %6.2f us (%5.0f GB/s)  fwd hot %6.2f us   bwd-apply cold %6.2f us (%5.0f GB/s)"
              % (cap, f_cold, 2 * nbytes / f_cold / 1e3, f_res, 3 * nbytes / f_res / 1e3, f_hot, b_cold, 3 * nbytes / b_cold / 1e3))
lib.bh_debug_force_tile(-20, 2048)
