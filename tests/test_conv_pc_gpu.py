"""conv3x3_pc_kernel (csrc/conv3x3_pc.hip, round 5): the fp16-piece 3x3 forward / dgrad on persistent workgroups with specialised waves,
against conv3x3_halo_kernel (one workgroup per tile, route bit BH_ROUTE_C3_TILE_WG) on the same operands.  Same MFMA order per
accumulator, same epilogue arithmetic per element: the convolution results must be BIT-IDENTICAL; the statistics sums (accumulated in
double per element here, in float per fragment quad there) must agree to rounding.  Every epilogue option of the C ABI is covered:
bias, BatchNorm statistics (1 / 2 groups), BatchNorm-on-load, residual + ReLU, accumulate, BatchNorm-backward sums (mask from z / from y),
column sums; shapes with several tiles per workgroup, several channel tiles, a ragged last tile position, non-power-of-two tile grids.
python tests/test_conv_pc_gpu.py times both kernels on the layer shapes of the training step."""
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(8, 16, 64, 64), (4, 8, 256, 256), (16, 16, 128, 128), (3, 24, 64, 128), (2, 8, 96, 64), (4, 24, 64, 64), (6, 40, 64, 64), (40, 32, 64, 64),
          (128, 32, 64, 64), (24, 16, 32, 64)]


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bihome_amd import kernels
    return kernels


def _descs(K, N, H, Ci, Co):
    from bihome_amd._lib import ROUTE_C3_PC, ROUTE_C3_TILE_WG, ROUTE_HALO_SMALL
    new = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_PC)
    old = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_TILE_WG)
    # (the persistent kernel takes the 64-channel output tile: forward Co % 64 == 0, dgrad Ci % 64 == 0; the other direction of such a
    #  layer runs the halo kernel's 32-channel tile either way)
    if Co % 64 == 0:
        assert K.conv_variant(K._with_layout(new, 4), "fwd").startswith("conv3x3_pc_kernel<false")
    if Ci % 64 == 0:
        assert K.conv_variant(K._with_layout(new, 4), "dgrad").startswith("conv3x3_pc_kernel<true")
    assert K.conv_variant(K._with_layout(old, 4), "fwd").startswith("conv3x3_halo_kernel<false")
    return new, old


def _operands(K, N, H, Ci, Co, seed=0):
    g = torch.Generator().manual_seed(seed + N * 7 + H + Ci)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Co, generator=g).cuda()
    pk = K.packer_for_precision(4)
    pf, pd = pk.get(w)
    pk.refresh()
    return x, gy, w.permute(0, 2, 3, 1), b, pf, pd, g


def _sums_close(a, b, tol=2e-6):
    a, b = a.cpu(), b.cpu()
    scale = b.abs().max().item() + 1e-30
    assert (a - b).abs().max().item() <= tol * scale, ((a - b).abs().max().item(), scale)


@pytest.mark.parametrize("N,H,Ci,Co", SHAPES)
def test_forward_bit_identical(K, N, H, Ci, Co):
    new, old = _descs(K, N, H, Ci, Co)
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co)
    for bias in (None, b):
        y1, y0 = K.conv_fwd(x, wk, bias, new, wpacked=pf), K.conv_fwd(x, wk, bias, old, wpacked=pf)
        assert torch.equal(y1, y0)
    for groups in ((1, 2) if N % 2 == 0 else (1,)):
        s1, s0 = K.bn_stats_buffer(groups, Co, "cuda"), K.bn_stats_buffer(groups, Co, "cuda")
        y1 = K.conv_fwd(x, wk, b, new, bn_sums=s1, groups=groups, wpacked=pf)
        y0 = K.conv_fwd(x, wk, b, old, bn_sums=s0, groups=groups, wpacked=pf)
        assert torch.equal(y1, y0)
        _sums_close(s1, s0)
        # the sums against float64 sums of the tensor the kernel wrote (entry (group, channel, moment) on its own 128-byte line)
        # (where the persistent kernel took the launch: it adds in double per element; the halo kernel in float per fragment quad)
        yd = y1.double().reshape(groups, -1, Co)
        tab = s1.reshape(groups, Co, 2, -1)[..., 0]
        exact = K.conv_variant(K._with_layout(new, 4), "fwd", bn_groups=groups).startswith("conv3x3_pc_kernel")
        _sums_close(tab[..., 0], yd.sum(1), 1e-11 if exact else 1e-6)
        _sums_close(tab[..., 1], (yd * yd).sum(1), 1e-11 if exact else 1e-6)
    res = torch.randn(N, H, H, Co, generator=g).cuda()
    y1 = K.conv_fwd(x, wk, b, new, res=res, relu=True, wpacked=pf)
    y0 = K.conv_fwd(x, wk, b, old, res=res, relu=True, wpacked=pf)
    assert torch.equal(y1, y0)
    # against float64 as well (not only against the sibling kernel)
    ref = torch.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), wk.double().cpu().permute(0, 3, 1, 2), b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
    y = K.conv_fwd(x, wk, b, new, wpacked=pf).cpu().double()
    assert ((y - ref).norm() / ref.norm()).item() < 1e-6


def test_epilogue_store_hazard_stress(K):
    """Round 5: a 128-bit buffer store whose data registers were overwritten by the next VALU instruction corrupted single channels in a few
    of 2048 sub-tiles, and only under some timings (csrc/conv3x3_pc.hip: the `s_nop 3` behind the store).  The forms that hit it - epilogue
    operands on the multi-tile shape - repeated, with a second stream keeping the memory system busy."""
    N, H, Ci, Co = 128, 32, 64, 64
    new, old = _descs(K, N, H, Ci, Co)
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co, seed=5)
    res = torch.randn(N, H, H, Co, generator=g).cuda()
    base = torch.randn(N, H, H, Ci, generator=g).cuda()
    y0 = K.conv_fwd(x, wk, b, old, res=res, relu=True, wpacked=pf)
    o0 = base.clone()
    K.conv_dgrad(gy, wk, old, out=o0, wpacked=pd)
    side = torch.cuda.Stream()
    junk = torch.empty(1 << 26, device="cuda")
    for it in range(12):
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    junk.add_(1.0)
        assert torch.equal(K.conv_fwd(x, wk, b, new, res=res, relu=True, wpacked=pf), y0)
        o1 = base.clone()
        K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd)
        assert torch.equal(o1, o0)
    torch.cuda.synchronize()


# (shapes the BatchNorm kernels take: two groups, C / 4 dividing 256; (4, 16, 32, 64) is a ONE-chunk forward: three barrier phases per tile)
BN_SHAPES = [s for s in SHAPES if s[0] % 2 == 0 and 256 % (s[2] // 4) == 0] + [(4, 16, 32, 64), (24, 16, 32, 64)]


@pytest.mark.parametrize("N,H,Ci,Co", BN_SHAPES)
@pytest.mark.parametrize("relu", [True, False])
def test_forward_batchnorm_on_load_bit_identical(K, N, H, Ci, Co, relu):
    new, old = _descs(K, N, H, Ci, Co)
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co, seed=1)
    groups = 2
    z = x * 1.5 + 0.3
    gamma = (torch.rand(Ci, generator=g) + 0.5).cuda()
    beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
    rm, rv = torch.zeros(Ci).cuda(), torch.ones(Ci).cuda()
    st = K.bn_stats_buffer(groups, Ci, "cuda")
    K.bn_stats(z, st, groups, Ci)
    rec = K.amax_record("cuda")
    table = K.bn_fwd_coeffs(st, gamma, beta, rm, rv, groups, N * H * H // groups, Ci, 1e-5, 0.1, amax=rec)
    lazy = K.BnOnLoad(z, table, groups, relu, amax=rec)
    for sums in (False, True):
        s1 = K.bn_stats_buffer(groups, Co, "cuda") if sums else None
        s0 = K.bn_stats_buffer(groups, Co, "cuda") if sums else None
        y1 = K.conv_fwd(lazy, wk, b, new, bn_sums=s1, groups=groups, wpacked=pf)
        y0 = K.conv_fwd(lazy, wk, b, old, bn_sums=s0, groups=groups, wpacked=pf)
        assert torch.equal(y1, y0)
        if sums:
            _sums_close(s1, s0)


@pytest.mark.parametrize("N,H,Ci,Co", SHAPES)
def test_dgrad_bit_identical(K, N, H, Ci, Co):
    new, old = _descs(K, N, H, Ci, Co)
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co, seed=2)
    g1, g0 = K.conv_dgrad(gy, wk, new, wpacked=pd), K.conv_dgrad(gy, wk, old, wpacked=pd)
    assert torch.equal(g1, g0)
    base = torch.randn(N, H, H, Ci, generator=g).cuda()
    o1, o0 = base.clone(), base.clone()
    K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd); K.conv_dgrad(gy, wk, old, out=o0, wpacked=pd)
    assert torch.equal(o1, o0)
    c1, c0 = K.bn_stats_buffer(1, Ci, "cuda"), K.bn_stats_buffer(1, Ci, "cuda")
    g1, g0 = K.conv_dgrad(gy, wk, new, wpacked=pd, colsum=c1), K.conv_dgrad(gy, wk, old, wpacked=pd, colsum=c0)
    assert torch.equal(g1, g0)
    _sums_close(c1, c0)
    ref = torch.nn.functional.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), wk.double().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    assert ((g1.cpu().double() - ref).norm() / ref.norm()).item() < 1e-6


@pytest.mark.parametrize("N,H,Ci,Co", [s for s in SHAPES if s[0] % 2 == 0 and 256 % (s[2] // 4) == 0])
@pytest.mark.parametrize("relu,with_y,acc", [(True, False, False), (True, True, True), (False, False, True), (True, False, True)])
def test_dgrad_batchnorm_reduce_bit_identical(K, N, H, Ci, Co, relu, with_y, acc):
    new, old = _descs(K, N, H, Ci, Co)
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co, seed=3)
    groups = 2
    z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
    gamma = (torch.rand(Ci, generator=g) + 0.5).cuda()
    beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
    st = K.bn_stats_buffer(groups, Ci, "cuda")
    K.bn_stats(z, st, groups, Ci)
    yb = torch.randn(N, H, H, Ci, generator=g).cuda() if with_y else None
    base = torch.randn(N, H, H, Ci, generator=g).cuda()
    outs = []
    for d in (new, old):
        sums = K.bn_stats_buffer(groups, Ci, "cuda")
        out = base.clone() if acc else None
        r = K.conv_dgrad(gy, wk, d, out=out, wpacked=pd,
                         bn_reduce=dict(z=z, y=yb, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=relu, sums=sums, groups=groups))
        outs.append((r, sums))
    assert torch.equal(outs[0][0], outs[1][0])
    _sums_close(outs[0][1], outs[1][1])


def test_deterministic_call_is_repeatable(K):
    """In a deterministic scope (BH_ROUTE_DETERMINISTIC) the statistics leave through integer limbs: bitwise repeatable, whatever the order
    in which the workgroups arrive."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    N, H, Ci, Co = 16, 16, 128, 128
    x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co, seed=4)
    runs = []
    with K.det_scope(True):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL)
        assert K.conv_variant(K._with_layout(d, 4), "fwd", bn_groups=2).startswith("conv3x3_pc_kernel")
        for _ in range(4):
            s = K.bn_stats_buffer(2, Co, "cuda")
            y = K.conv_fwd(x, wk, b, d, bn_sums=s, groups=2, wpacked=pf)
            runs.append((y, s))
    assert all(torch.equal(runs[0][0], r[0]) for r in runs[1:])
    for r in runs[1:]:
        ds = (runs[0][1].view(torch.int64) != r[1].view(torch.int64)).reshape(-1, 16)       # (bit patterns: a negative limb read as a double is a NaN)
        assert not ds.any(), (int(ds.sum()), ds.any(0).tolist())
    # and the limbs hold the same totals as the default mode's doubles
    d0 = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL)
    s0 = K.bn_stats_buffer(2, Co, "cuda")
    K.conv_fwd(x, wk, b, d0, bn_sums=s0, groups=2, wpacked=pf)
    e = runs[0][1].reshape(-1, 16)
    limbs = e[:, 1:4].contiguous().view(torch.int64).double()
    tot = e[:, 0] + limbs[:, 0] + limbs[:, 1] * 2.0 ** -40 + limbs[:, 2] * 2.0 ** -80
    _sums_close(tot, s0.reshape(-1, 16)[:, 0], 1e-12)


def _bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if __name__ == "__main__":
    sys.path.insert(0, ".")
    from bihome_amd import kernels as K
    from bihome_amd._lib import ROUTE_C3_PC, ROUTE_C3_TILE_WG
    shapes = [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64), (128, 32, 128, 128), (128, 16, 256, 256)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (N, H, Ci, Co) in shapes:
        x, gy, wk, b, pf, pd, g = _operands(K, N, H, Ci, Co)
        z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
        gamma, beta = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.2).cuda()
        st = K.bn_stats_buffer(2, Ci, "cuda"); K.bn_stats(z, st, 2, Ci)
        base = torch.randn(N, H, H, Ci, generator=g).cuda()
        res = {}
        for rnd in range(3):                   # alternate the two kernels (the first variant measured in a process runs slower whatever it is)
            for tag, route in (("pc", ROUTE_C3_PC), ("halo", ROUTE_C3_TILE_WG)):
                d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=route)
                s = K.bn_stats_buffer(2, Co, "cuda"); s2 = K.bn_stats_buffer(2, Ci, "cuda")
                bnr = dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=s2, groups=2)
                for name, fn in (("fwd", lambda: K.conv_fwd(x, wk, None, d, wpacked=pf)),
                                 ("fwd+stats", lambda: K.conv_fwd(x, wk, None, d, bn_sums=s, groups=2, wpacked=pf)),
                                 ("dgrad", lambda: K.conv_dgrad(gy, wk, d, wpacked=pd)),
                                 ("dgrad+bnr+acc", lambda: K.conv_dgrad(gy, wk, d, out=base, wpacked=pd, bn_reduce=bnr))):
                    res.setdefault((name, tag), []).append(_bench(fn))
        fl = 2.0 * N * H * H * Ci * Co * 9
        for name in ("fwd", "fwd+stats", "dgrad", "dgrad+bnr+acc"):
            p, h = min(res[(name, "pc")]), min(res[(name, "halo")])
            print("%-22s %-14s pc %6.1f us (%5.0f TF alg)  halo %6.1f us   %s" % ((N, H, Ci, Co), name, p, fl / p / 1e6, h,
                  " ".join("%.1f/%.1f" % (a, c) for a, c in zip(res[(name, "pc")], res[(name, "halo")]))), flush=True)
