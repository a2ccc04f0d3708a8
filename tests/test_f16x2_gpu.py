"""'f16x2' arithmetic (bh_conv_desc.precision = 4, round 4): two FP16 pieces per operand with a power-of-two scale per tensor, three
products per product on v_mfma_f32_32x32x16_f16 - the 3x3 / stride-1 layers (forward, dgrad, weight gradient) at fp32 accuracy for
half the matrix-pipe work of f32x3.  Every kernel is compared with torch float64 NEXT TO the fp32-input MFMA kernel (precision 0)
and the exact three-piece form (precision 2) on the same data: the claim under test is "within 4x of the fp32-input MFMA kernel's
error" (VERDICT r03 item 1), on every shape of the f32x3 tests including the 24-binade and the cancellation case."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from bihome_amd import kernels
    return kernels


def _packed(K, w, prec):
    pk = K.packer_for_precision(prec) if prec in K.SPLIT_PIECES else K.WeightPacker()
    pf, pd = pk.get(w)
    pk.refresh()
    return pk, pf, pd


def _rel(a, ref):
    return ((a.cpu().double() - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("N,H,Ci,Co,wide", [
    (128, 32, 64, 64, False),    # the bench shape: two tile positions per workgroup, two chunks
    (8, 16, 128, 128, False),    # four chunks, two n tiles
    (8, 8, 256, 256, False),     # one sub-tile per workgroup, eight chunks
    (4, 64, 32, 32, False),      # 32-channel tile, single chunk (one halo stage)
    (3, 24, 96, 160, False),     # odd sub-tile count, three chunks, 64- and 32-wide n tiles
    (5, 16, 32, 64, False),      # single chunk, odd image count
    (8, 16, 64, 64, True),       # operands spread over 24 binades
])
def test_conv3x3_f16x2_forward_and_dgrad(K, N, H, Ci, Co, wide):
    from bihome_amd._lib import ROUTE_HALO_SMALL
    g = torch.Generator().manual_seed(N * 7 + H)
    x = torch.randn(N, H, H, Ci, generator=g)
    gy = torch.randn(N, H, H, Co, generator=g) * 1e-4                    # gradients are small numbers: the scale has work to do
    if wide:
        x = x * torch.exp2(torch.randint(-12, 12, x.shape, generator=g).float())
        gy = gy * torch.exp2(torch.randint(-12, 12, gy.shape, generator=g).float())
    x, gy = x.cuda(), gy.cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
    ref = F.conv2d(xd, wd, b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), wd, None, 1, 1).permute(0, 2, 3, 1)
    err = {}
    for prec in (0, 2, 3, 4):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        pk, pf, pd = _packed(K, w, prec)
        dp = K._with_layout(d, K.packed_layout(prec))
        if prec == 4:
            # (round 5: the 64-channel tile runs on the persistent producer / consumer kernel, the 32-channel tile on the halo kernel)
            assert K.conv_variant(dp, "fwd").endswith((",true,true,false,2,false,true>", "conv3x3_pc_kernel<false,false,0>"))
            assert K.conv_variant(dp, "dgrad").endswith((",2,false,true>", "conv3x3_pc_kernel<true,false,0>"))
        y = K.conv_fwd(x, wk, b, d, wpacked=pf)
        s = K.bn_stats_buffer(1, Co, "cuda")
        assert torch.equal(y, K.conv_fwd(x, wk, b, d, bn_sums=s, groups=1, wpacked=pf))
        gx = K.conv_dgrad(gy, wk, d, wpacked=pd)
        acc = x.clone()
        acc._bh_amax = None
        K.conv_dgrad(gy, wk, d, out=acc, wpacked=pd)
        assert _rel(acc, (x + gx).cpu().double()) < 1e-6
        err[prec] = (_rel(y, ref), _rel(gx, refd))
    print("\nf16x2 fwd/dgrad N%d H%d %d->%d%s: rel-L2 vs f64  fp32-mfma %.2e / %.2e  f32x3 %.2e / %.2e  f32x2 %.2e / %.2e  f16x2 %.2e / %.2e"
          % (N, H, Ci, Co, " wide" if wide else "", *err[0], *err[2], *err[3], *err[4]))
    assert err[4][0] <= 1.5 * err[0][0] and err[4][1] <= 1.5 * err[0][1], err       # (measured 0.6-0.7 x the fp32-input MFMA kernel's error; round-4 VERDICT: was 4 x)
    assert err[4][0] < 0.2 * err[3][0] and err[4][1] < 0.2 * err[3][1], err       # and well below the two-piece bf16 form


def test_f16x2_packed_pieces_reconstruct_the_weights(K):
    Co, Ci = 64, 96
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * torch.exp2(torch.randint(-10, 3, (Co, Ci, 3, 3), generator=g).float())).cuda()
    w = w.contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk, pf, pd = _packed(K, w, 4)
    n = Co * 9 * Ci
    assert pf.numel() == n + 16 and pd.numel() == n + 16
    amax = pf[n:].max().item()
    assert amax == w.abs().max().item() and pd[n:].max().item() == amax
    import math
    k = 14 - math.floor(math.log2(amax))
    pieces = pf[:n].view(torch.float16).view(-1, 2, 2, 64, 8)                     # [chunk*tap*ntile][piece][step][lane][e]
    assert torch.isfinite(pieces.float()).all() and pieces[:, 0].abs().max().item() < 2.0 ** 15 * 1.0001
    total = pieces.double().sum(1) * 2.0 ** -k
    NW = Co // 32
    back = torch.empty(Co, 9, Ci, dtype=torch.float64, device="cuda")
    t = total.view(Ci // 32, 9, NW, 2, 2, 32, 8)                                  # [c][tap][nt][s2][kh2][l31][e]
    back.view(NW, 32, 9, Ci // 32, 2, 2, 8).copy_(t.permute(2, 5, 1, 0, 3, 4, 6))
    wref = wk.reshape(Co, 9, Ci).double()
    # 22 bits and a sign down to 2^-18 of the maximum, an absolute 2^-40 of it (2^-25 in scaled units) below
    tol = torch.maximum(wref.abs() * 2.0 ** -22, torch.full_like(wref, 2.0 ** -25 * 2.0 ** -k))
    assert ((back - wref).abs() <= tol).all()


@pytest.mark.parametrize("N,H,Ci,Co", [
    (128, 32, 64, 64), (2, 8, 64, 64), (3, 24, 64, 128), (8, 8, 256, 128), (16, 16, 128, 128), (5, 40, 64, 64), (4, 64, 32, 32),
    (3, 24, 32, 96), (2, 16, 64, 32),
])
def test_wgrad_f16x2_kernel(K, N, H, Ci, Co):
    g = torch.Generator().manual_seed(N + H + Ci)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = (torch.randn(N, H, H, Co, generator=g) * 3e-5).cuda()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w, None, 1, 1)
    ref = torch.autograd.grad(y, w, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)       # [Co][3][3][Ci]
    err = {}
    for prec in (0, 2, 3, 4):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec)
        need = K.wgrad_det_bytes(d)
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda") if need else None
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
        err[prec] = _rel(gw, ref)
        if prec == 4:
            assert K.conv_variant(d, "wgrad_det").startswith("wgrad_x3_kernel<%d,false,2,true,false,false>" % (64 if (Ci % 64 == 0 and Co % 64 == 0) else 32))
            if Ci % 64 == 0 and Co % 64 == 0:
                # round 5: the eight-wave producer / consumer form (route bit BH_ROUTE_WX3_PC; net.py asks for it in one-stream steps): same tiles,
                # same MFMA order, same partial blocks - bitwise the four-wave gradient
                from bihome_amd._lib import ROUTE_WX3_PC
                dpc = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_WX3_PC)
                assert K.conv_variant(dpc, "wgrad_det").startswith("wgrad_x3_kernel<64,false,2,true,true,false>")
                gpc = torch.zeros(Co, 3, 3, Ci, device="cuda")
                K.conv_wgrad(x, gy, gpc, None, dpc, det_ws=ws)
                assert torch.equal(gpc, gw)
            g1 = torch.ones(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, g1, None, d, det_ws=ws)
            assert _rel(g1 - 1.0, gw.cpu().double()) < 1e-2 * 1.0 and torch.equal(gw, gw)       # lands on what gw holds (1 + 1e-5-sized entries)
            gw2 = torch.zeros(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, gw2, None, d, det_ws=ws)
            assert torch.equal(gw, gw2)                                                          # workspace form: bitwise repeatable
            gw3 = torch.zeros(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, gw3, None, d)                                                    # atomics form
            assert _rel(gw3, gw.cpu().double()) < 2e-6
    print("\nf16x2 wgrad N%d H%d %d->%d: rel-L2 vs f64  fp32-mfma %.2e  f32x3 %.2e  f32x2 %.2e  f16x2 %.2e" % (N, H, Ci, Co, err[0], err[2], err[3], err[4]))
    assert err[4] <= 1.5 * err[0] + 1e-8 and err[4] < 0.25 * err[3], err       # (measured 0.75 x)


@pytest.mark.parametrize("N,Ci,Co", [(128, 512, 512), (8, 128, 64), (12, 64, 128)])
def test_wgrad_f16x2_on_4x4_maps(K, N, Ci, Co):
    """Round 5 (round-4 VERDICT item 9): the fp16-piece weight gradient on 4 x 4 feature maps (layer4 of the ResNet-34 regressor,
    /root/reference/src/backbones/ResNet34.py:6-50) - four images per 8 x 8 tile, each behind its own zero border - instead of the fp32-input
    MFMA kernel: against torch float64 next to that kernel's error, workspace form bitwise repeatable, BatchNorm-on-load variant included."""
    g = torch.Generator().manual_seed(N + Ci)
    x = torch.randn(N, 4, 4, Ci, generator=g).cuda()
    gy = (torch.randn(N, 4, 4, Co, generator=g) * 1e-3).cuda()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w, None, 1, 1)
    ref = torch.autograd.grad(y, w, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    err = {}
    for prec in (0, 4):
        d = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=prec)
        need = K.wgrad_det_bytes(d)
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda") if need else None
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
        err[prec] = _rel(gw, ref)
        if prec == 4:
            assert K.conv_variant(d, "wgrad_det").startswith("wgrad_x3_kernel<64,false,2,true,false,true>"), K.conv_variant(d, "wgrad_det")
            from bihome_amd._lib import ROUTE_WX3_PC
            dpc = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_WX3_PC)
            assert K.conv_variant(dpc, "wgrad_det").startswith("wgrad_x3_kernel<64,false,2,true,true,true>")
            gpc = torch.zeros(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, gpc, None, dpc, det_ws=ws)
            assert torch.equal(gpc, gw)                                                          # the eight-wave form: bitwise the same
            gw2 = torch.zeros(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, gw2, None, d, det_ws=ws)
            assert torch.equal(gw, gw2)
            gw3 = torch.zeros(Co, 3, 3, Ci, device="cuda")
            K.conv_wgrad(x, gy, gw3, None, d)                                                    # atomics form
            assert _rel(gw3, gw.cpu().double()) < 2e-6
    print("\nf16x2 wgrad on 4x4 maps N%d %d->%d: rel-L2 vs f64  fp32-mfma %.2e  f16x2 %.2e" % (N, Ci, Co, err[0], err[4]))
    assert err[4] <= 1.5 * err[0] + 1e-8, err


def test_conv3x3_f16x2_error_bound_under_cancellation(K):
    """Dot products whose terms cancel to ~1e-4 of sum |a||b|, operands with all 24 significand bits set: the ABSOLUTE error in units
    of sum_k |a_k||b_k| 2^-24 stays a small constant (f32x3: <= 4; f16x2 drops 2 bits per operand: measured 0.65, asserted < 2)."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    N, H, Ci, Co = 8, 16, 64, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H, H, Ci, generator=g)
    x = (x.view(torch.int32) | 0x7FF).view(torch.float32)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    w[:, 1::2] = -w[:, 0::2] * (1.0 + 1e-4 * torch.randn(Co, Ci // 2, 3, 3, generator=g))
    x[..., 1::2] = x[..., 0::2] * (1.0 + 1e-4 * torch.randn(N, H, H, Ci // 2, generator=g))
    x, w = x.cuda(), w.cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
    ref = F.conv2d(xd, wd, None, 1, 1).permute(0, 2, 3, 1)
    mag = F.conv2d(xd.abs(), wd.abs(), None, 1, 1).permute(0, 2, 3, 1)
    assert (ref.abs() / mag).median() < 1e-3
    ulps = {}
    for prec in (0, 2, 3, 4):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        pk, pf, _ = _packed(K, w, prec)
        y = K.conv_fwd(x, wk, None, d, wpacked=pf)
        ulps[prec] = ((y.cpu().double() - ref).abs() / (mag * 2.0 ** -24)).max().item()
    print("\ncancellation: max |err| / (sum|a||b| 2^-24): fp32-mfma %.2f  f32x3 %.2f  f32x2 %.2f  f16x2 %.2f" % (ulps[0], ulps[2], ulps[3], ulps[4]))
    assert ulps[4] < 2.0 and ulps[4] < 0.1 * ulps[3], ulps       # (measured 0.65; the fp32-input MFMA kernel 0.17: the one case where the mode is weaker)


@pytest.mark.parametrize("N,H,Ci,Co,relu", [(8, 16, 64, 64, True), (4, 32, 32, 32, False), (16, 8, 256, 256, True), (6, 24, 128, 64, True)])
def test_f16x2_batchnorm_on_load_and_records(K, N, H, Ci, Co, relu):
    """The BatchNorm kernels leave the magnitude records the fp16-piece consumers need: bn_fwd / bn_bwd measure max |y| / max |gx|
    exactly, bn_fwd_coeffs writes the a-priori bound (>= the actual maximum); BatchNorm-on-load forward + weight gradient in f16x2
    against the materialised float64 form."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    groups = 2
    g = torch.Generator().manual_seed(N + Ci)
    z = (torch.randn(N, H, H, Ci, generator=g) * 3.0 + 1.0).cuda()
    gamma = (torch.rand(Ci, generator=g) + 0.5).cuda()
    beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    gy = (torch.randn(N, H, H, Co, generator=g) * 1e-3).cuda()
    rows = N * H * H // groups
    # float64 reference of BatchNorm(+ReLU) per group, conv forward and weight gradient
    zd = z.double().cpu().view(groups, -1, Ci)
    mu, var = zd.mean(1, keepdim=True), zd.var(1, unbiased=False, keepdim=True)
    yd = ((zd - mu) / torch.sqrt(var + 1e-5) * gamma.double().cpu() + beta.double().cpu())
    if relu:
        yd = yd.clamp_min(0)
    yd = yd.view(N, H, H, Ci)
    wt = w.double().cpu().requires_grad_(True)
    out = F.conv2d(yd.permute(0, 3, 1, 2), wt, None, 1, 1)
    gw_ref = torch.autograd.grad(out, wt, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    ref = out.detach().permute(0, 2, 3, 1)
    # measured record of the materialised form
    rm, rv = torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda")
    rec = K.amax_record("cuda")
    y, st = K.bn_fwd(z, gamma, beta, rm, rv, None, groups, 1e-5, 0.1, relu, True, amax=rec)
    assert y._bh_amax is rec and rec.max().item() == y.abs().max().item()
    # a-priori record of the on-load form
    st2 = K.bn_stats_buffer(groups, Ci, "cuda")
    st2.copy_(st)
    rec2 = K.amax_record("cuda")
    table = K.bn_fwd_coeffs(st2, gamma, beta, rm.clone(), rv.clone(), groups, rows, Ci, 1e-5, 0.1, amax=rec2)
    bound = rec2.max().item()
    assert bound >= yd.abs().max().item() and bound <= 4.0 * ((gamma.abs().max().item()) * rows ** 0.5 + beta.abs().max().item())
    err = {}
    for prec in (2, 4):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        d.bh_wx3 = True
        pk, pf, pd = _packed(K, w, prec)
        bol = K.BnOnLoad(z, table, groups, relu, amax=rec2)
        yo = K.conv_fwd(bol, wk, None, d, wpacked=pf)
        need = K.wgrad_det_bytes(d)
        ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(bol, gy, gw, None, d, det_ws=ws)
        err[prec] = (_rel(yo, ref), _rel(gw, gw_ref))
    print("\nBatchNorm-on-load N%d H%d %d->%d: fwd / wgrad rel-L2 vs f64  f32x3 %.2e / %.2e  f16x2 %.2e / %.2e" % (N, H, Ci, Co, *err[2], *err[4]))
    assert err[4][0] <= 4.0 * err[2][0] + 1e-7 and err[4][1] <= 4.0 * err[2][1] + 1e-7, err
    # backward record: max |gx| measured by the apply kernel
    g_in = (torch.randn(N, H, H, Ci, generator=g) * 1e-2).cuda()
    recb = K.amax_record("cuda")
    gx, _ = K.bn_bwd(g_in, y, z, gamma, st, rm, rv, groups, 1e-5, relu, True, False, beta=beta, had_res=False, amax=recb)
    assert gx._bh_amax is recb and recb.max().item() == gx.abs().max().item()


def test_f16x2_a_wrong_record_is_loud_not_silently_wrong(K):
    """A magnitude record that UNDERSTATES the tensor overflows fp16: the result carries inf / NaN, never a plausible wrong number;
    an overstated one (2^10 too large) only costs precision gracefully."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    N, H, Ci, Co = 8, 16, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL)
    pk, pf, _ = _packed(K, w, 4)
    x._bh_amax = torch.full((K.AMAX_FLOATS,), 1e-3, device="cuda")
    y = K.conv_fwd(x, wk, None, d, wpacked=pf)
    assert not torch.isfinite(y).all()
    x._bh_amax = torch.full((K.AMAX_FLOATS,), float(x.abs().max().item()) * 1024.0, device="cuda")
    y = K.conv_fwd(x, wk, None, d, wpacked=pf)
    assert _rel(y, ref) < 2e-6
    x._bh_amax = None
    y = K.conv_fwd(x, wk, None, d, wpacked=pf)           # no record: measured by a streaming pass
    assert x._bh_amax is not None and x._bh_amax.max().item() == x.abs().max().item() and _rel(y, ref) < 1e-6


@pytest.mark.parametrize("N,H,Ci,C,C2,residual,bni", [(8, 16, 64, 64, 64, False, False), (4, 32, 64, 128, 128, True, False), (16, 8, 256, 256, 256, False, True),
                                                      (6, 24, 128, 64, 64, True, True), (128, 32, 64, 64, 64, False, False)])
def test_wgrad_with_the_batchnorm_adjoint_on_load(K, N, H, Ci, C, C2, residual, bni):
    """Round 6 (bh_conv_wgrad_bnadj, include/bihome.h; round-5 VERDICT item 2): conv1 (Ci -> C) -> BatchNorm (+ residual) -> ReLU -> conv2
    (C -> C2).  conv2's dgrad completes d, the gradient of the BatchNorm's output, and leaves the BatchNorm's backward sums and
    max |mask(d)| (bh_bn_reduce.amax_d); conv1's weight gradient is then computed (a) as before - bn_bwd materialises the adjoint g, the
    fp16-piece kernel contracts x with g - and (b) with the adjoint applied ON LOAD from (d, z [, y]) and the sums.  (b) against (a), both
    against a float64 evaluation of the same formulas, the eight-wave form bitwise the four-wave one, the record exact."""
    from bihome_amd._lib import ROUTE_WX3_PC
    groups = 2
    g = torch.Generator().manual_seed(N + H + C)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    z = (torch.randn(N, H, H, C, generator=g) * 2.0 + 0.5).cuda()                   # conv1's output = the BatchNorm's input
    res = torch.randn(N, H, H, C, generator=g).cuda() if residual else None
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y, st = K.bn_fwd(z, gamma, beta, rm, rv, res, groups, 1e-5, 0.1, True, True)
    w2 = (torch.randn(C2, C, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    pk, pf2, pd2 = _packed(K, w2, 4)
    gnext = (torch.randn(N, H, H, C2, generator=g) * 1e-3).cuda()
    from bihome_amd._lib import ROUTE_HALO_SMALL
    d2 = K.conv_desc(N, H, H, C, C2, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL)      # (small grids: keep the launch on the 3x3 kernels)
    sums = K.bn_stats_buffer(groups, C, "cuda")
    rec_d = K.amax_record("cuda")
    K.amax_of(gnext)
    dout = K.conv_dgrad(gnext, w2.permute(0, 2, 3, 1), d2, wpacked=pd2,
                        bn_reduce=dict(z=z, y=y if residual else None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=sums, groups=groups,
                                       amax_d=rec_d))
    mask = (y > 0)
    assert rec_d.max().item() == (dout * mask).abs().max().item()                   # the record: max |mask(d)|, exactly
    # the conv1 input: a tensor, or a BatchNorm(+ReLU) applied on load
    xin = x
    if bni:
        g0, b0 = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.2).cuda()
        st0 = K.bn_stats_buffer(groups, Ci, "cuda"); K.bn_stats(x, st0, groups, Ci)
        rec0 = K.amax_record("cuda")
        table0 = K.bn_fwd_coeffs(st0, g0, b0, torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), groups, N * H * H // groups, Ci, 1e-5, 0.1, amax=rec0)
        xin = K.BnOnLoad(x, table0, groups, True, amax=rec0)
    d1 = K.conv_desc(N, H, H, Ci, C, 3, 1, 1, precision=4)
    d1.bh_wx3 = True
    ws = torch.empty(K.wgrad_det_bytes(d1) // 4, dtype=torch.float32, device="cuda")
    # (a) materialised adjoint
    recg = K.amax_record("cuda")
    gx, _ = K.bn_bwd(dout, y if residual else None, z, gamma, st, rm, rv, groups, 1e-5, True, True, False, beta=beta, had_res=residual,
                     sums_ready=sums, amax=recg)
    gw_a = torch.zeros(C, 3, 3, Ci, device="cuda")
    K.conv_wgrad(xin, gx, gw_a, None, d1, det_ws=ws)
    # (b) on load
    bna = dict(z=z, y=y if residual else None, stats=st, sums=sums, gamma=gamma, beta=beta, eps=1e-5, relu=True, groups=groups)
    gw_b = torch.zeros(C, 3, 3, Ci, device="cuda")
    assert K.conv_wgrad_bnadj(xin, dout, gw_b, d1, ws, bna, rec_d)
    dpc = K.conv_desc(N, H, H, Ci, C, 3, 1, 1, precision=4, route=ROUTE_WX3_PC)
    dpc.bh_wx3 = True
    gw_c = torch.zeros(C, 3, 3, Ci, device="cuda")
    assert K.conv_wgrad_bnadj(xin, dout, gw_c, dpc, ws, bna, rec_d)
    assert torch.equal(gw_c, gw_b)                                                  # eight-wave form: bitwise the four-wave one
    # float64 evaluation of the same formulas from the same d
    dd, zd = (dout * mask).double().cpu().view(groups, -1, C), z.double().cpu().view(groups, -1, C)
    mu, var = zd.mean(1, keepdim=True), zd.var(1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = (zd - mu) * rstd
    gref = (gamma.double().cpu() * rstd) * (dd - dd.mean(1, keepdim=True) - xhat * (dd * xhat).mean(1, keepdim=True))
    xd = x.double().cpu()
    if bni:
        x0 = xd.view(groups, -1, Ci)
        m0, v0 = x0.mean(1, keepdim=True), x0.var(1, unbiased=False, keepdim=True)
        xd = ((x0 - m0) / torch.sqrt(v0 + 1e-5) * g0.double().cpu() + b0.double().cpu()).clamp_min(0).view(N, H, H, Ci)
    wt = torch.zeros(C, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(xd.permute(0, 3, 1, 2), wt, None, 1, 1), wt, gref.view(N, H, H, C).permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    ea, eb = _rel(gw_a, ref), _rel(gw_b, ref)
    print("\nBatchNorm adjoint on load N%d H%d %d->%d%s%s: weight gradient rel-L2 vs f64  materialised %.2e  on load %.2e; on load vs materialised %.2e"
          % (N, H, Ci, C, " +res" if residual else "", " bnin" if bni else "", ea, eb, _rel(gw_b, gw_a.cpu().double())))
    assert eb <= 1.5 * ea + 2e-7, (ea, eb)
    # accumulates into what gw holds
    g1 = torch.ones(C, 3, 3, Ci, device="cuda")
    assert K.conv_wgrad_bnadj(xin, dout, g1, d1, ws, bna, rec_d)
    assert _rel(g1 - 1.0, gw_b.cpu().double()) < 1e-2


@pytest.mark.parametrize("N,H,Ci,Co,n,bni", [(16, 16, 64, 64, 4, False), (128, 32, 64, 64, 2, False), (8, 8, 256, 128, 3, False), (6, 24, 128, 64, 4, True),
                                             (16, 16, 64, 64, 1, False)])
def test_wgrad_f16x2_batched_layers(K, N, H, Ci, Co, n, bni):
    """Round 6 (bh_conv_wgrad_batch, include/bihome.h): the fp16-piece weight gradients of n layers of one geometry in ONE launch, against
    the n single launches (same kernel, fewer workgroups per layer: equal up to the order of the split-K sums) and against float64;
    every layer reads its OWN operands, records and BatchNorm-on-load table, and accumulates into what its gradient buffer holds."""
    from bihome_amd._lib import ROUTE_WX3_SHARED
    groups = 2
    g = torch.Generator().manual_seed(N + H + Ci + n)
    items, singles, refs = [], [], []
    ws = None
    for i in range(n):
        x = (torch.randn(N, H, H, Ci, generator=g) * (1.0 + i)).cuda()
        gy = (torch.randn(N, H, H, Co, generator=g) * 10.0 ** (-3 - i)).cuda()           # (very different magnitudes: one record per layer)
        xin, xd = x, x.double().cpu()
        if bni:
            g0, b0 = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.2).cuda()
            st0 = K.bn_stats_buffer(groups, Ci, "cuda"); K.bn_stats(x, st0, groups, Ci)
            rec0 = K.amax_record("cuda")
            table0 = K.bn_fwd_coeffs(st0, g0, b0, torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), groups, N * H * H // groups, Ci, 1e-5, 0.1, amax=rec0)
            xin = K.BnOnLoad(x, table0, groups, True, amax=rec0)
            x0 = xd.view(groups, -1, Ci)
            xd = ((x0 - x0.mean(1, keepdim=True)) / torch.sqrt(x0.var(1, unbiased=False, keepdim=True) + 1e-5) * g0.double().cpu() + b0.double().cpu()).clamp_min(0).view(N, H, H, Ci)
        wt = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        refs.append(torch.autograd.grad(F.conv2d(xd.permute(0, 3, 1, 2), wt, None, 1, 1), wt, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1))
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_WX3_SHARED); d.bh_wx3 = True
        if ws is None:      # (a batch's partial blocks: <= 256 x 147 KB whatever the layer count - the models' 40 MB workspace)
            ws = torch.empty(max(K.wgrad_det_bytes(d), 40 << 20) // 4, dtype=torch.float32, device="cuda")
        base = torch.randn(Co, 3, 3, Ci, generator=g).cuda() * 1e-6
        gw1 = base.clone()
        K.conv_wgrad(xin, gy, gw1, None, d, det_ws=ws)
        singles.append((gw1, base))
        d2 = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_WX3_SHARED); d2.bh_wx3 = True
        items.append((xin, gy, base.clone(), d2))
    assert K.conv_wgrad_batch(items, ws)
    for i in range(n):
        got, (one, base) = items[i][2], singles[i]
        e_b, e_1 = _rel(got - base, refs[i]), _rel(one - base, refs[i])
        print("batched wgrad N%d H%d %d->%d layer %d of %d%s: rel-L2 vs f64  batched %.2e  single %.2e  batched vs single %.2e"
              % (N, H, Ci, Co, i, n, " bnin" if bni else "", e_b, e_1, _rel(got - base, (one - base).cpu().double())))
        assert e_b <= 1.5 * e_1 + 2e-7, (i, e_b, e_1)
    # repeatable bit for bit (ordered reduction through the workspace)
    again = [(it[0], it[1], singles[i][1].clone(), it[3]) for i, it in enumerate(items)]
    assert K.conv_wgrad_batch(again, ws)
    assert all(torch.equal(a[2], b[2]) for a, b in zip(again, items))
