"""Round 5: adjoint of BatchNorm + ReLU + MaxPool - bh_maxpool3s2_bwd + bh_bn_bwd against bh_bn_maxpool_bwd on the extractor stem's shape."""
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


for (groups, N, H, C) in [(2, 128, 64, 64), (2, 256, 64, 64)]:
    x = torch.randn(N, H, H, C, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    st = K.bn_stats_buffer(groups, C, "cuda")
    f = K.bn_maxpool_fwd(x, gamma, beta, rm, rv, groups, 1e-5, 0.1, True, True, st, False)
    gp = torch.randn_like(f.pooled)

    def two():
        gy = K.maxpool_bwd(f.idx, gp, f.shape)
        return K.bn_bwd(gy, None, x, gamma, st, rm, rv, groups, 1e-5, True, True, False, beta=beta, had_res=False)[0]

    def one():
        return K.bn_maxpool_bwd(K.PooledGrad(gp, f.idx), x, gamma, beta, st, rm, rv, groups, 1e-5, True, True)

    print((groups, N, H, C), "same:", torch.equal(two(), one()), " two calls %.1f %.1f us   one call %.1f %.1f us" % (bench(two), bench(two), bench(one), bench(one)), flush=True)
