"""Round 5: BatchNorm adjoint behind a 1x1 conv - conv_dgrad + bn_bwd against bh_bn_bwd_from_1x1 (the full-resolution decoder unit's shape)."""
import sys; sys.path.insert(0, '.')
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


for (N, H, C, KC) in [(128, 128, 32, 16), (128, 64, 64, 32), (128, 32, 128, 32)]:
    groups = 2
    x = torch.randn(N, H, H, C, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    st = K.bn_stats_buffer(groups, C, "cuda"); K.bn_stats(x, st, groups, C)
    w = torch.randn(KC, 1, 1, C, device="cuda") * 0.2
    gs = torch.randn(N, H, H, KC, device="cuda")
    d = K.conv_desc(N, H, H, C, KC, 1, 1, 0, precision=4)

    def two():
        return K.bn_bwd(K.conv_dgrad(gs, w, d), None, x, gamma, st, rm, rv, groups, 1e-5, True, True, False, beta=beta, had_res=False)[0]

    def one():
        return K.bn_bwd_from_1x1(K.GradFrom1x1(gs, w), x, gamma, beta, st, groups, 1e-5, True)

    e = ((two() - one()).norm() / two().norm()).item()
    print((N, H, C, KC), "rel diff %.1e" % e, " dgrad + bn_bwd %.1f %.1f us   fused %.1f %.1f us" % (bench(two), bench(two), bench(one), bench(one)), flush=True)
