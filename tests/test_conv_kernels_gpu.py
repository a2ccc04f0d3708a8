"""Conv-stack kernels (through the C ABI) against plain PyTorch CPU ops in float64.

The MFMA path multiplies exact fp32 products and accumulates in fp32, so against a float64 reference
the error is fp32 round-off of a K-term dot product: rtol 2e-5 on outputs normalised by their scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from bihome_amd import kernels
    return kernels


def rnd(shape, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)


def close(a, ref, tol=3e-5):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(a - ref).max() / scale
    assert err < tol, "max normalised error %.3e (tol %.1e)" % (err, tol)


# (N, Hi, Ci, Co, k, stride, pad): the Zeng / ResNet-34 / extractor shape census at small N
CONV_CASES = [
    (2, 32, 64, 64, 3, 1, 1), (2, 32, 64, 128, 3, 2, 1), (3, 16, 128, 128, 3, 1, 1), (2, 16, 128, 256, 3, 2, 1),
    (5, 8, 256, 256, 3, 1, 1), (2, 32, 64, 128, 1, 2, 0), (2, 16, 256, 128, 1, 1, 0), (1, 64, 32, 16, 1, 1, 0),
    (1, 64, 16, 128, 1, 1, 0), (2, 64, 32, 32, 3, 1, 1), (1, 8, 256, 512, 3, 2, 1), (3, 4, 512, 512, 3, 1, 1),
    (1, 20, 24, 40, 3, 1, 1), (1, 1, 512, 8, 1, 1, 0),
]


@pytest.mark.parametrize("N,Hi,Ci,Co,k,s,p", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(K, N, Hi, Ci, Co, k, s, p):
    x = rnd((N, Ci, Hi, Hi), 1)
    w = (rnd((Co, Ci, k, k), 2) / np.sqrt(Ci * k * k)).astype(np.float32)
    b = rnd((Co,), 3)
    d = K.conv_desc(N, Hi, Hi, Ci, Co, k, s, p)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xt, wt, bt, stride=s, padding=p)
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()                 # NHWC
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()                 # [Co][kh][kw][Ci]
    y = K.conv_fwd(xg, wg, torch.tensor(b).cuda(), d)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach())
    gy = rnd(tuple(ref.shape), 4)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gyg = torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda()
    gx = K.conv_dgrad(gyg, wg, d)
    close(gx.cpu().permute(0, 3, 1, 2), xt.grad)
    # accumulate form
    base = torch.ones_like(gx)
    gx2 = K.conv_dgrad(gyg, wg, d, out=base.clone())
    close((gx2 - 1).cpu().permute(0, 3, 1, 2), xt.grad, 1e-4)
    gw = torch.zeros_like(wg)
    gb = torch.zeros(Co, device="cuda")
    K.conv_wgrad(xg, gyg, gw, gb, d)
    close(gw.cpu().permute(0, 3, 1, 2), wt.grad)
    close(gb.cpu(), bt.grad)


@pytest.mark.parametrize("N,Hi,Ci,Co", [(2, 8, 256, 256), (2, 8, 256, 128), (1, 16, 128, 64), (1, 32, 64, 32), (1, 64, 32, 16)])
def test_conv_transpose_2x2(K, N, Hi, Ci, Co):
    x = rnd((N, Ci, Hi, Hi), 5)
    w = (rnd((Ci, Co, 2, 2), 6) / np.sqrt(Ci)).astype(np.float32)
    b = rnd((Co,), 7)
    d = K.conv_desc(N, Hi, Hi, Ci, Co, 2, 2, 0, transposed=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    ref = F.conv_transpose2d(xt, wt, bt, stride=2)
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()                 # [Ci][kh][kw][Co]
    y = K.conv_fwd(xg, wg, torch.tensor(b).cuda(), d)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach())
    gy = rnd(tuple(ref.shape), 8)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gyg = torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda()
    close(K.conv_dgrad(gyg, wg, d).cpu().permute(0, 3, 1, 2), xt.grad)
    gw = torch.zeros_like(wg)
    gb = torch.zeros(Co, device="cuda")
    K.conv_wgrad(xg, gyg, gw, gb, d)
    close(gw.cpu().permute(0, 3, 1, 2), wt.grad)
    close(gb.cpu(), bt.grad)


@pytest.mark.parametrize("N,Hi,Ci,Co", [(3, 32, 2, 64), (2, 48, 3, 64), (8, 128, 2, 64), (5, 160, 2, 64)])
def test_first_conv_nchw_input(K, N, Hi, Ci, Co):
    """7x7/2 on the NCHW network input (Rethinking.py:31, ResNet34.py:17): scalar gather path; two planes and >= 512 8x8 tiles
    (the last two cases): stem7_fwd_kernel<2> and the dedicated weight-gradient kernel (round 4: bh_stem7_wgrad - accumulates into gw,
    bitwise repeatable)."""
    x = rnd((N, Ci, Hi, Hi), 9)
    w = rnd((Co, Ci, 7, 7), 10) / 10
    d = K.conv_desc(N, Hi, Hi, Ci, Co, 7, 2, 3, in_nchw=True)
    xt = torch.tensor(x, dtype=torch.float64)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xt, wt, None, stride=2, padding=3)
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()
    y = K.conv_fwd(torch.tensor(x).cuda(), wg, None, d)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach())
    gy = rnd(tuple(ref.shape), 11)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gw = torch.zeros_like(wg)
    K.conv_wgrad(torch.tensor(x).cuda(), torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda(), gw, None, d)
    close(gw.cpu().permute(0, 3, 1, 2), wt.grad)
    if N * (Hi // 16) ** 2 >= 512 and Ci == 2:
        from bihome_amd._lib import lib
        import ctypes
        assert lib.bh_stem7_wgrad_ws_bytes(ctypes.byref(d)) > 0
        gw2 = gw.clone()
        K.conv_wgrad(torch.tensor(x).cuda(), torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda(), gw2, None, d)      # += : twice the gradient
        close(gw2.cpu().permute(0, 3, 1, 2), 2 * wt.grad)
        gw3 = torch.zeros_like(wg)
        K.conv_wgrad(torch.tensor(x).cuda(), torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda(), gw3, None, d)
        assert torch.equal(gw3, gw)                                                                                         # no atomics


@pytest.mark.parametrize("N,Hi,det", [(2, 32, False), (8, 64, False), (3, 128, False), (8, 64, True)])
def test_gray_stem_conv_and_its_dgrad(K, N, Hi, det):
    """Extractor conv1 on a 1-channel image (PerceptualHead.py:52-55) and the dgrad the warp needs.  Small: the direct gather kernel;
    large: one kernel (tile GEMM + overlap-add in LDS + atomics into the image, round 4: bh_stem7_dgrad_c1); large in a deterministic
    call: the two-pass form (1x1 GEMM into a tap table + col2im)."""
    if det:
        with K.det_scope(True):
            return test_gray_stem_conv_and_its_dgrad(K, N, Hi, False)
    x = rnd((N, 1, Hi, Hi), 12)
    w = rnd((64, 1, 7, 7), 13) / 7
    d = K.conv_desc(N, Hi, Hi, 1, 64, 7, 2, 3)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xt, torch.tensor(w, dtype=torch.float64), None, stride=2, padding=3)
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()
    y = K.conv_fwd(torch.tensor(x).reshape(N, Hi, Hi, 1).cuda(), wg, None, d)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach())
    gy = rnd(tuple(ref.shape), 14)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gx = K.conv_dgrad(torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda(), wg, d)
    close(gx.cpu().reshape(N, 1, Hi, Hi), xt.grad)


@pytest.mark.parametrize("N,Hi,scale", [(8, 64, 1.0), (4, 128, 1e-4), (3, 128, 30.0)])
def test_gray_stem_dgrad_in_fp16_pieces(K, N, Hi, scale):
    """Round 6 (stem7_dgrad_c1_kernel<.., F16>; bh_conv_desc.precision = 4): the one-channel stem's dgrad with the window GEMM in the fp16-piece
    arithmetic - against torch float64 next to the fp32-input MFMA form on the same data (error <= 1.5x), gradients of very different
    magnitudes incl. a tile that is all zeros and one 1e-5 of the rest (the tile's scale comes from its own maximum)."""
    g = torch.Generator().manual_seed(21 + Hi)
    w = torch.randn(64, 1, 7, 7, generator=g, dtype=torch.float64) / 7
    gy = torch.randn(N, 64, Hi // 2, Hi // 2, generator=g, dtype=torch.float64) * scale
    gy[0, :, :8, :8] = 0.0
    gy[-1, :, 8:24, 8:24] *= 1e-5
    xt = torch.zeros(N, 1, Hi, Hi, dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, w, None, stride=2, padding=3).backward(gy)
    ref = xt.grad
    wg = w.float().permute(0, 2, 3, 1).contiguous().cuda()
    gyk = gy.float().permute(0, 2, 3, 1).contiguous().cuda()
    err = {}
    for prec in (0, 4):
        d = K.conv_desc(N, Hi, Hi, 1, 64, 7, 2, 3, precision=prec)
        gx = K.conv_dgrad(gyk, wg, d).cpu().double().reshape(N, 1, Hi, Hi)
        err[prec] = ((gx - ref).norm() / ref.norm()).item()
        err[(prec, "dim")] = ((gx[-1, :, 20:44, 20:44] - ref[-1, :, 20:44, 20:44]).norm() / ref[-1, :, 20:44, 20:44].norm()).item()
    print("\nstem dgrad N%d H%d scale %g: rel-L2 vs f64  fp32-input MFMA %.2e  fp16 pieces %.2e;  dim region %.2e / %.2e"
          % (N, Hi, scale, err[0], err[4], err[(0, "dim")], err[(4, "dim")]))
    assert err[4] <= 1.5 * err[0] + 1e-8 and err[(4, "dim")] <= 1.5 * err[(0, "dim")] + 1e-7, err


def test_last_conv_nchw_output(K):
    """1x1 128->2 with bias writing the NCHW perspective field (Rethinking.py:147) and adjoints fed by an
    NCHW gradient."""
    N, Hi, Ci, Co = 2, 16, 128, 2
    x = rnd((N, Ci, Hi, Hi), 15)
    w = rnd((Co, Ci, 1, 1), 16) / 11
    b = rnd((Co,), 17)
    d = K.conv_desc(N, Hi, Hi, Ci, Co, 1, 1, 0, out_nchw=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xt, wt, bt)
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()
    y = K.conv_fwd(xg, wg, torch.tensor(b).cuda(), d)
    assert tuple(y.shape) == (N, Co, Hi, Hi)
    close(y.cpu(), ref.detach())
    gy = rnd(tuple(ref.shape), 18)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gyg = torch.tensor(gy).cuda()
    close(K.conv_dgrad(gyg, wg, d).cpu().permute(0, 3, 1, 2), xt.grad)
    gw = torch.zeros_like(wg)
    gb = torch.zeros(Co, device="cuda")
    K.conv_wgrad(xg, gyg, gw, gb, d)
    close(gw.cpu().permute(0, 3, 1, 2), wt.grad)
    close(gb.cpu(), bt.grad)


@pytest.mark.parametrize("groups,N,H,C,relu,res,training", [
    (2, 3, 8, 64, True, True, True), (1, 2, 16, 128, True, False, True), (4, 2, 4, 256, False, False, True),
    (2, 2, 32, 16, True, True, True), (1, 4, 8, 32, False, True, True), (2, 2, 8, 64, True, True, False),
    (1, 3, 2, 512, True, False, True)])
def test_batchnorm_fwd_bwd(K, groups, N, H, C, relu, res, training):
    x = (rnd((groups * N, C, H, H), 20) * 2 + 0.5).astype(np.float32)
    r = rnd((groups * N, C, H, H), 21) if res else None
    gamma, beta = (1 + 0.3 * rnd((C,), 22)).astype(np.float32), (0.2 * rnd((C,), 23)).astype(np.float32)
    rm0, rv0 = (0.1 * rnd((C,), 24)).astype(np.float32), (1 + 0.2 * np.abs(rnd((C,), 25))).astype(np.float32)
    bn = torch.nn.BatchNorm2d(C).double()
    bn.weight.data, bn.bias.data = torch.tensor(gamma, dtype=torch.float64), torch.tensor(beta, dtype=torch.float64)
    bn.running_mean.data, bn.running_var.data = torch.tensor(rm0, dtype=torch.float64), torch.tensor(rv0, dtype=torch.float64)
    bn.train(training)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rt = torch.tensor(r, dtype=torch.float64, requires_grad=True) if res else None
    outs = []
    for g in range(groups):                          # one reference call per group, in order
        o = bn(xt[g * N:(g + 1) * N])
        if res:
            o = o + rt[g * N:(g + 1) * N]
        outs.append(F.relu(o) if relu else o)
    ref = torch.cat(outs, 0)
    nhwc = lambda a: torch.tensor(a).permute(0, 2, 3, 1).contiguous().cuda()
    rm, rv = torch.tensor(rm0).cuda(), torch.tensor(rv0).cuda()
    gm, bt = torch.tensor(gamma).cuda(), torch.tensor(beta).cuda()
    y, st = K.bn_fwd(nhwc(x), gm, bt, rm, rv, nhwc(r) if res else None, groups, bn.eps, 0.1, relu, training)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach(), 2e-5)
    close(rm.cpu(), bn.running_mean, 1e-5)
    close(rv.cpu(), bn.running_var, 1e-5)
    gy = rnd(tuple(ref.shape), 26)
    if relu:   # a ReLU input within rounding of 0 may get a different sign in float32: exclude those few elements
        gy[np.abs(y.cpu().permute(0, 3, 1, 2).numpy()) < 1e-5] = 0
        gy[(np.abs(ref.detach().numpy()) < 1e-5) & (ref.detach().numpy() > 0)] = 0
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gg, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gx, gres = K.bn_bwd(nhwc(gy), y, nhwc(x), gm, st, rm, rv, groups, bn.eps, relu, training, res, gg, gb, beta=bt)
    close(gx.cpu().permute(0, 3, 1, 2), xt.grad, 5e-5)
    close(gg.cpu(), bn.weight.grad, 5e-5)
    close(gb.cpu(), bn.bias.grad, 5e-5)
    if res:
        close(gres.cpu().permute(0, 3, 1, 2), rt.grad, 1e-6)


@pytest.mark.parametrize("N,H,C", [(2, 64, 64), (1, 17, 8), (3, 8, 256)])
def test_maxpool_gap_add(K, N, H, C):
    x = rnd((N, C, H, H), 30)
    x[0, :, 2:5, 2:5] = 1.5                      # ties inside windows: first-maximum rule
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = F.max_pool2d(xt, 3, 2, 1)
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()
    y, idx = K.maxpool_fwd(xg)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2).double(), ref.detach())
    gy = rnd(tuple(ref.shape), 31)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gx = K.maxpool_bwd(idx, torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda(), tuple(xg.shape))
    close(gx.cpu().permute(0, 3, 1, 2), xt.grad, 1e-6)
    g = K.gap_fwd(xg)
    close(g.cpu().reshape(N, C), x.mean((2, 3)), 1e-6)
    gg = K.gap_bwd(g, tuple(xg.shape))
    close(gg.cpu(), np.broadcast_to(g.cpu().numpy() / (H * H), (N, H, H, C)), 1e-6)
    a, b = torch.tensor(rnd((1001,), 32)).cuda(), torch.tensor(rnd((1001,), 33)).cuda()
    s = a + b
    K.add_(a, b)
    assert torch.equal(a, s)


def relerr64(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("sparse", [0, 1, 2])
@pytest.mark.parametrize("groups,N,H,Ci,Cm,Co,training", [(2, 2, 16, 16, 128, 2, True), (1, 3, 8, 16, 128, 2, True),
                                                          (2, 1, 32, 8, 64, 3, True), (2, 2, 16, 16, 128, 2, False),
                                                          (2, 3, 32, 16, 128, 2, True), (1, 2, 16, 16, 64, 1, True)])
def test_fused_tail_fwd_bwd(K, groups, N, H, Ci, Cm, Co, training, sparse):
    """conv1x1+BN+ReLU+conv1x1 fused (csrc/tail.hip) against the unfused float64 PyTorch chain, one BN call per group.
    sparse 1 / 2: the output gradient is non-zero on ~1 % / ~15 % of the pixels only (the biHomE field gradient lives on the DSAC
    samples): the zero-skipping reduction and the affine + sparse / + dense-wave forms of the input gradient (round 4)."""
    NN = groups * N
    x = (rnd((NN, Ci, H, H), 40) * 1.5 + 0.3).astype(np.float32)
    w1 = (rnd((Cm, Ci, 1, 1), 41) / np.sqrt(Ci)).astype(np.float32)
    b1 = rnd((Cm,), 42)
    gamma, beta = (1 + 0.3 * rnd((Cm,), 43)).astype(np.float32), (0.3 * rnd((Cm,), 44)).astype(np.float32)
    w2 = (rnd((Co, Cm, 1, 1), 45) / np.sqrt(Cm)).astype(np.float32)
    b2 = rnd((Co,), 46)
    rm0, rv0 = (0.1 * rnd((Cm,), 47)).astype(np.float32), (1 + 0.2 * np.abs(rnd((Cm,), 48))).astype(np.float32)
    D = lambda a: torch.tensor(a, dtype=torch.float64)
    xt, w1t, b1t, w2t, b2t = [D(a).requires_grad_(True) for a in (x, w1, b1, w2, b2)]
    bn = torch.nn.BatchNorm2d(Cm).double()
    bn.weight.data, bn.bias.data, bn.running_mean.data, bn.running_var.data = D(gamma), D(beta), D(rm0), D(rv0)
    bn.train(training)
    pre, outs = [], []
    for g in range(groups):
        p = bn(F.conv2d(xt[g * N:(g + 1) * N], w1t, b1t))
        pre.append(p)
        outs.append(F.conv2d(F.relu(p), w2t, b2t))
    ref = torch.cat(outs, 0)
    C = lambda a: torch.tensor(a).cuda()
    xg = C(x).permute(0, 2, 3, 1).contiguous()
    rm, rv = C(rm0), C(rv0)
    args = (C(w1).reshape(Cm, Ci), C(b1), C(gamma), C(beta), rm, rv, C(w2).reshape(Co, Cm), C(b2))
    out, ws = K.tail_fwd(xg, *args, groups, H * H, bn.eps, 0.1, training)
    close(out.cpu(), ref.detach(), 3e-5)
    close(rm.cpu(), bn.running_mean, 1e-5)
    close(rv.cpu(), bn.running_var, 2e-5)
    if sparse == 0:
        # the matrix-pipe forward (the library's choice where the shape allows) against the per-pixel VALU forward: same fp32-grade result
        rm2, rv2 = C(rm0), C(rv0)
        a2 = args[:4] + (rm2, rv2) + args[6:]
        out_v, _ = K.tail_fwd(xg, *a2, groups, H * H, bn.eps, 0.1, training, route=K.TAIL_ROUTE_VALU_FWD)
        close(out_v.cpu(), ref.detach(), 3e-5)
        close(out.cpu(), out_v.cpu(), 1e-5)
        # the input moments on the matrix pipe (round 4, Ci = 16: X^T X as a GEMM) against the LDS-slab kernel: the BatchNorm statistics
        # derived from them (workspace head: [groups][Cm][2] sums) agree to double-precision rounding of fp32-accumulated blocks
        rm3, rv3 = C(rm0), C(rv0)
        a3 = args[:4] + (rm3, rv3) + args[6:]
        out_l, ws_l = K.tail_fwd(xg, *a3, groups, H * H, bn.eps, 0.1, training, route=K.TAIL_ROUTE_LDS_MOMENTS)
        close(out_l.cpu(), ref.detach(), 3e-5)
        if training:
            n = groups * Cm * 2
            assert relerr64(ws[:n].cpu().numpy(), ws_l[:n].cpu().numpy()) < 1e-6
            close(rv3.cpu(), rv.cpu(), 1e-6)
    gy = rnd(tuple(ref.shape), 49)
    if sparse:
        keep = np.random.RandomState(77).rand(NN, 1, H, H) < (0.01 if sparse == 1 else 0.15)
        gy = np.where(np.broadcast_to(keep, gy.shape), gy, 0).astype(np.float32)
    # ReLU inputs within rounding of zero may take the other branch in float32: keep them out of the gradient check
    near = (torch.cat(pre, 0).detach().abs() < 1e-4).any(1, keepdim=True).numpy()
    gy = np.where(np.broadcast_to(near, gy.shape), 0, gy).astype(np.float32)
    ref.backward(D(gy))
    gw1, gg, gb, gw2, gb2 = [torch.zeros(s, device="cuda") for s in ((Cm, Ci), (Cm,), (Cm,), (Co, Cm), (Co,))]
    gx = K.tail_bwd(C(gy), xg, args[0], args[1], args[2], args[3], args[6], ws, rm, rv, groups, H * H, bn.eps, training, True,
                    gw1, gg, gb, gw2, gb2)
    close(gx.cpu().permute(0, 3, 1, 2), xt.grad, 1e-4)
    close(gw1.cpu().reshape(Cm, Ci, 1, 1), w1t.grad, 1e-4)
    close(gw2.cpu().reshape(Co, Cm, 1, 1), w2t.grad, 1e-4)
    close(gg.cpu(), bn.weight.grad, 1e-4)
    close(gb.cpu(), bn.bias.grad, 1e-4)
    close(gb2.cpu(), b2t.grad, 1e-5)


def _err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)


@pytest.mark.parametrize("prec", [2, 4])
@pytest.mark.parametrize("N,Hi,Ci,Co,k,s,p,T,scale", [
    (4, 32, 64, 128, 3, 2, 1, False, 1.0), (4, 16, 128, 256, 3, 2, 1, False, 1.0), (2, 32, 64, 128, 1, 2, 0, False, 1.0),
    (3, 16, 256, 128, 1, 1, 0, False, 1.0), (4, 16, 128, 128, 2, 2, 0, True, 1.0), (4, 16, 128, 64, 2, 2, 0, True, 1.0),
    (2, 8, 256, 256, 2, 2, 0, True, 1.0), (4, 32, 64, 128, 3, 2, 1, False, 1e-6), (4, 16, 128, 64, 2, 2, 0, True, 3e4)])
def test_generic_kernel_in_three_bf16_pieces(K, N, Hi, Ci, Co, k, s, p, T, scale, prec):
    """Round 6 (BH_ROUTE_GEMM_X3, opt-in): in the fp32-accurate modes (bh_conv_desc.precision 2 / 4) the generic implicit-GEMM kernel -
    strided and 1x1 convs, 128- and 256-channel transposed convs, their dgrads - cuts its operands exactly into three bf16 pieces and makes
    six products per product (conv_gemm_kernel<...,true,true,true>) instead of running the fp32-input MFMA.  Error against float64 no worse than the fp32-input
    MFMA's on the same data (both ~1e-7 of the largest output: the fp32 accumulate), at ordinary, tiny and large magnitudes."""
    x = rnd((N, Ci, Hi, Hi), 150) * scale
    if T:
        w = (rnd((Ci, Co, k, k), 151) / np.sqrt(Ci)).astype(np.float32)
    else:
        w = (rnd((Co, Ci, k, k), 151) / np.sqrt(Ci * k * k)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    ref = F.conv_transpose2d(xt, wt, None, stride=2) if T else F.conv2d(xt, wt, None, stride=s, padding=p)
    gy = rnd(tuple(ref.shape), 152)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()
    gyg = torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda()
    errs = {}
    from bihome_amd._lib import ROUTE_GEMM_X3
    for pr in (0, prec):
        d = K.conv_desc(N, Hi, Hi, Ci, Co, k, s, p, transposed=T, precision=pr, route=ROUTE_GEMM_X3)
        if pr:       # (without the route bit these layers stay on the fp32-input MFMA, bit for bit precision 0)
            assert K.conv_variant(K.conv_desc(N, Hi, Hi, Ci, Co, k, s, p, transposed=T, precision=pr), "fwd") == \
                K.conv_variant(K.conv_desc(N, Hi, Hi, Ci, Co, k, s, p, transposed=T, precision=0), "fwd")
        y = K.conv_fwd(xg, wg, None, d)
        gx = K.conv_dgrad(gyg, wg, d)
        errs[pr] = (_err(y.cpu().permute(0, 3, 1, 2), ref.detach()), _err(gx.cpu().permute(0, 3, 1, 2), xt.grad))
        if pr:
            for which in ("fwd", "dgrad"):
                v = K.conv_variant(d, which)
                # conv_gemm_kernel<BM,BN,BK,VEC,BF16,BUF,X3>
                assert all(part.startswith("conv_gemm_kernel<") and len(part.split(",")) == 7 and part.endswith(",true>")
                           for part in v.split("+") if "pack" not in part), v
    print("MEASURED generic kernel N%d %dx%d C%d->%d k%d s%d%s x%g: fp32-input MFMA fwd %.2e dgrad %.2e | three bf16 pieces (precision %d) fwd %.2e dgrad %.2e"
          % (N, Hi, Hi, Ci, Co, k, s, " T" if T else "", scale, errs[0][0], errs[0][1], prec, errs[prec][0], errs[prec][1]))
    for i in range(2):
        assert errs[prec][i] <= 1.5 * errs[0][i] + 2e-8, errs
        assert errs[prec][i] < 2e-6


@pytest.mark.parametrize("N,Hi,Ci,Co,k,s,p", [(2, 32, 64, 64, 3, 1, 1), (2, 16, 128, 256, 3, 2, 1), (3, 8, 256, 256, 3, 1, 1),
                                              (2, 16, 256, 128, 1, 1, 0)])
def test_bf16_operand_mode(K, N, Hi, Ci, Co, k, s, p):
    """precision=1: operands rounded to bf16 while staging, fp32 accumulate (BASELINE.json configs[3]).  Tolerance =
    bf16 round-off (2^-9 per operand) of a K-term dot product against the float64 result: 1e-2 normalised."""
    x = rnd((N, Ci, Hi, Hi), 50)
    w = (rnd((Co, Ci, k, k), 51) / np.sqrt(Ci * k * k)).astype(np.float32)
    d = K.conv_desc(N, Hi, Hi, Ci, Co, k, s, p, precision=1)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xt, wt, None, stride=s, padding=p)
    xg = torch.tensor(x).permute(0, 2, 3, 1).contiguous().cuda()
    wg = torch.tensor(w).permute(0, 2, 3, 1).contiguous().cuda()
    close(K.conv_fwd(xg, wg, None, d).cpu().permute(0, 3, 1, 2), ref.detach(), 1e-2)
    gy = rnd(tuple(ref.shape), 52)
    ref.backward(torch.tensor(gy, dtype=torch.float64))
    gyg = torch.tensor(gy).permute(0, 2, 3, 1).contiguous().cuda()
    close(K.conv_dgrad(gyg, wg, d).cpu().permute(0, 3, 1, 2), xt.grad, 1e-2)
    gw = torch.zeros_like(wg)
    K.conv_wgrad(xg, gyg, gw, None, d)
    close(gw.cpu().permute(0, 3, 1, 2), wt.grad, 1e-2)


@pytest.mark.parametrize("N,H,Ci,Co,k,groups", [
    (8, 16, 32, 64, 3, 2),      # halo-tiled 3x3 kernel, BN = 64: statistics in the conv epilogue
    (6, 8, 64, 32, 3, 2),       # BN = 32 variant; 8x8 images: the two sub-tiles of a workgroup can sit in different groups
    (2, 8, 32, 64, 3, 2),       # one image per group: every workgroup straddles the group boundary
    (4, 8, 16, 32, 1, 2),       # 1x1 conv: generic kernel, statistics in its epilogue (128 pixels per group)
    (2, 4, 16, 32, 1, 2),       # 16 pixels per group: generic kernel + statistics launch
    (4, 16, 32, 128, 1, 2),     # 128-wide N tile
])
def test_conv_fwd_with_batchnorm_sums(K, N, H, Ci, Co, k, groups):
    """bh_conv_fwd_bnstats: per-group, per-channel (sum y, sum y^2) of the conv output, as BatchNorm consumes them."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    if True:
        x = torch.tensor(rnd((N, H, H, Ci), 70)).cuda()
        w = torch.tensor(rnd((Co, k, k, Ci), 71) * 0.1).cuda()
        b = torch.tensor(rnd((Co,), 72)).cuda()
        d = K.conv_desc(N, H, H, Ci, Co, k, 1, k // 2, route=ROUTE_HALO_SMALL)   # let the halo kernel take these small grids
        sums = K.bn_stats_buffer(groups, Co, "cuda")
        y = K.conv_fwd(x, w, b, d, bn_sums=sums, groups=groups)
        y0 = K.conv_fwd(x, w, b, d)
        assert torch.equal(y, y0)
        yd = y0.double().reshape(groups, -1, Co)
        ref = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)          # [groups, Co, 2]
        close(sums.reshape(K.BN_SUM_SLOTS, groups, Co, 2, K.BN_SUM_STRIDE)[..., 0].sum(0).cpu(), ref.cpu(), 1e-6)
        # and through BatchNorm: identical output whether the sums come from the conv or from the statistics kernel
        gm, bt = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        o1, _ = K.bn_fwd(y, gm, bt, rm.clone(), rv.clone(), None, groups, 1e-5, 0.1, True, True, stats=sums, stats_ready=True)
        o2, _ = K.bn_fwd(y, gm, bt, rm.clone(), rv.clone(), None, groups, 1e-5, 0.1, True, True)
        close(o1.cpu(), o2.cpu(), 1e-5)


@pytest.mark.parametrize("N,H,C,Co,groups,relu,res,acc", [
    (8, 16, 64, 64, 2, True, False, False),     # inner BatchNorm of a residual unit: mask recomputed from z
    (8, 16, 64, 32, 2, True, True, True),       # block-output BatchNorm: residual added, gradient joined (accumulate)
    (4, 8, 32, 64, 2, False, False, False),     # no ReLU, 32-channel variant of the dgrad kernel
])
def test_dgrad_epilogue_accumulates_batchnorm_backward_sums(K, N, H, C, Co, groups, relu, res, acc):
    """bh_conv_dgrad_bnreduce + bh_bn_bwd(flags bit4) against the three-launch BatchNorm adjoint on the same gradient."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    if True:
        z = torch.tensor(rnd((N, H, H, C), 80) * 1.5 + 0.3).cuda()
        r = torch.tensor(rnd((N, H, H, C), 81)).cuda() if res else None
        gm, bt = torch.tensor(1 + 0.3 * rnd((C,), 82)).cuda(), torch.tensor(0.2 * rnd((C,), 83)).cuda()
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y, st = K.bn_fwd(z, gm, bt, rm, rv, r, groups, 1e-5, 0.1, relu, True)
        # the conv that consumes y: 3x3, C -> Co
        d = K.conv_desc(N, H, H, C, Co, 3, 1, 1, route=ROUTE_HALO_SMALL)
        w = torch.tensor(rnd((Co, 3, 3, C), 84) * 0.1).cuda()
        gy = torch.tensor(rnd((N, H, H, Co), 85)).cuda()
        part = torch.tensor(rnd((N, H, H, C), 86)).cuda() if acc else None      # gradient already joined from another branch
        # reference path: plain dgrad, then the three-launch adjoint
        g_ref = K.conv_dgrad(gy, w, d, out=part.clone()) if acc else K.conv_dgrad(gy, w, d)
        gg0, gb0 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        gx0, gr0 = K.bn_bwd(g_ref, y, z, gm, st, rm, rv, groups, 1e-5, relu, True, res, gg0, gb0, beta=bt)
        # fused path
        sums = K.bn_stats_buffer(groups, C, "cuda")
        red = dict(z=z, y=y if (relu and res) else None, stats=st, gamma=gm, beta=bt, eps=1e-5, relu=relu, sums=sums, groups=groups)
        assert K.dgrad_bn_reduce_ok(d)
        g_f = K.conv_dgrad(gy, w, d, out=part.clone(), bn_reduce=red) if acc else K.conv_dgrad(gy, w, d, bn_reduce=red)
        assert torch.equal(g_f, g_ref)
        gg1, gb1 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        gx1, gr1 = K.bn_bwd(g_f, y, z, gm, st, rm, rv, groups, 1e-5, relu, True, res, gg1, gb1, beta=bt, sums_ready=sums)
        close(gx1.cpu(), gx0.cpu(), 2e-5)
        close(gg1.cpu(), gg0.cpu(), 2e-5)
        close(gb1.cpu(), gb0.cpu(), 2e-5)
        if res:
            assert torch.equal(gr1, gr0)


@pytest.mark.parametrize("N,H,Ci,Co,k", [(2, 8, 64, 64, 3), (1, 32, 64, 128, 3), (3, 16, 128, 64, 3), (2, 4, 64, 64, 3),
                                         (2, 16, 64, 128, 1), (1, 16, 64, 64, 5), (8, 2, 64, 64, 7)])
def test_wgrad_stride1_fast_path_matches_generic_kernel(K, N, H, Ci, Co, k):
    """wgrad_s1_kernel<1> / <3> (tap shift folded into the buffer base, window tests for the borders) against the generic
    split-K kernel and torch float64, including maps narrower than the tile step and a 5x5 kernel."""
    from bihome_amd._lib import ROUTE_WGRAD_3TAP, ROUTE_WGRAD_GENERIC
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    d = K.conv_desc(N, H, H, Ci, Co, k, 1, k // 2)
    ref64 = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (Co, Ci, k, k), gy.double().cpu().permute(0, 3, 1, 2),
                                        stride=1, padding=k // 2).permute(0, 2, 3, 1)
    out = {}
    for mode, route in ((0, ROUTE_WGRAD_GENERIC), (1, 0), (3, ROUTE_WGRAD_3TAP)):
        dm = K.conv_desc(N, H, H, Ci, Co, k, 1, k // 2, route=route)
        gw = torch.zeros(Co, k, k, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, dm)
        out[mode] = gw.cpu()
    assert K.conv_variant(d, "wgrad") == "wgrad_s1_kernel<1,false>"
    assert K.conv_variant(K.conv_desc(N, H, H, Ci, Co, k, 1, k // 2, route=ROUTE_WGRAD_GENERIC), "wgrad") == "wgrad_kernel<true,false>"
    scale = float(ref64.abs().max())
    for mode in (0, 1, 3):
        assert float((out[mode].double() - ref64).abs().max()) < 2e-5 * scale + 1e-5, mode


def test_conv3x3_two_tile_positions_per_workgroup(K):
    """Launches of 513-1024 workgroups run as one round of workgroups that each walk two tile positions (conv3x3.hip, a.tpb):
    forward with BatchNorm sums and dgrad against the generic kernel, odd position count."""
    from bihome_amd._lib import ROUTE_GENERIC_CONV
    N, H, C = 65, 32, 64                                  # 65 * 16 sub-tiles / 2 = 520 workgroups -> 260 x 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, H, C, generator=g).cuda()
    w = (torch.randn(C, 3, 3, C, generator=g) * 0.05).cuda()
    gy = torch.randn(N, H, H, C, generator=g).cuda()
    res = {}
    for mode in (1, 0):                                   # 1: generic kernel, 0: halo kernel
        d = K.conv_desc(N, H, H, C, C, 3, 1, 1, route=ROUTE_GENERIC_CONV if mode else 0)
        assert K.conv_variant(d, "fwd").startswith("conv_gemm_kernel" if mode else "conv3x3_halo_kernel<false,64,false,2,")
        sums = K.bn_stats_buffer(1, C, "cuda")
        res[mode] = (K.conv_fwd(x, w, None, d, bn_sums=sums, groups=1), K.conv_dgrad(gy, w, d), sums)
    assert float((res[0][0] - res[1][0]).abs().max()) < 5e-5
    assert float((res[0][1] - res[1][1]).abs().max()) < 5e-5
    assert float((res[0][2] - res[1][2]).abs().max() / res[1][2].abs().max()) < 1e-6


@pytest.mark.parametrize("N,H,Ci,relu", [(4, 128, 1, False), (4, 128, 2, False), (2, 256, 3, False), (1, 256, 6, True)])
def test_stem7_forward_kernel(K, N, H, Ci, relu):
    """csrc/stem7.hip (7x7 / stride 2 / pad 3 stems, NCHW planes -> NHWC) against torch float64 and the generic kernel."""
    from bihome_amd._lib import ROUTE_NO_STEM7
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Ci, H, H, generator=g, dtype=torch.float64)
    w = torch.randn(64, Ci, 7, 7, generator=g, dtype=torch.float64) * 0.1
    b = torch.randn(64, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, b, 2, 3)
    if relu:
        ref = F.relu(ref)
    d = K.conv_desc(N, H, H, Ci, 64, 7, 2, 3, in_nchw=Ci != 1)
    assert K.conv_variant(d, "fwd") == "stem7_fwd_kernel<%d>" % Ci
    xk = x.float().cuda().contiguous()
    if Ci == 1:
        xk = xk.view(N, H, H, 1)
    wk = w.float().cuda().permute(0, 2, 3, 1).contiguous()
    y = K.conv_fwd(xk, wk, b.float().cuda(), d, relu=relu)
    close(y.permute(0, 3, 1, 2).cpu(), ref, 2e-5)
    d0 = K.conv_desc(N, H, H, Ci, 64, 7, 2, 3, in_nchw=Ci != 1, route=ROUTE_NO_STEM7)
    assert K.conv_variant(d0, "fwd").startswith("conv_gemm_kernel")
    y0 = K.conv_fwd(xk, wk, b.float().cuda(), d0, relu=relu)
    close(y.cpu(), y0.cpu(), 2e-5)


@pytest.mark.parametrize("N,H,Ci,relu,scale", [(4, 128, 1, False, 1.0), (4, 128, 2, True, 1.0), (16, 64, 1, False, 300.0), (8, 128, 2, False, 1e-3)])
def test_stem7_forward_in_fp16_pieces(K, N, H, Ci, relu, scale):
    """Round 6 (stem7_fwd_f16_kernel, csrc/stem7.hip; bh_conv_desc.precision = 4): the one- and two-plane stems in the fp16-piece
    arithmetic of the 3x3 layers - against torch float64, next to the fp32-input MFMA kernel on the same data (error <= 1.5x its error),
    with inputs of very different magnitudes (the patch's scale is taken per tile, the filter bank's per launch: no magnitude record), dark
    tiles in a bright image, and the BatchNorm sums of the epilogue."""
    g = torch.Generator().manual_seed(7 + Ci)
    x = torch.randn(N, Ci, H, H, generator=g, dtype=torch.float64) * scale
    x[0, :, : H // 2] *= 1e-4                                 # a dark half image: its tiles get their own scale
    x[-1, :, 16:32, 16:32] = 0.0                              # an all-zero tile
    w = torch.randn(64, Ci, 7, 7, generator=g, dtype=torch.float64) * 0.1
    b = torch.randn(64, generator=g, dtype=torch.float64) * scale
    ref = F.conv2d(x, w, b, 2, 3)
    if relu:
        ref = F.relu(ref)
    xk = x.float().cuda().contiguous()
    if Ci == 1:
        xk = xk.view(N, H, H, 1)
    wk = w.float().cuda().permute(0, 2, 3, 1).contiguous()
    err = {}
    for prec in (0, 4):
        d = K.conv_desc(N, H, H, Ci, 64, 7, 2, 3, in_nchw=Ci != 1, precision=prec)
        assert K.conv_variant(d, "fwd") == ("stem7_fwd_f16_kernel<%d>" if prec == 4 else "stem7_fwd_kernel<%d>") % Ci
        y = K.conv_fwd(xk, wk, b.float().cuda(), d, relu=relu)
        yd = y.permute(0, 3, 1, 2).cpu().double()
        err[prec] = ((yd - ref).norm() / ref.norm()).item()
        # per dark region as well: the dark half image must keep its relative accuracy
        dark = ((yd[0, :, : H // 4 - 2] - ref[0, :, : H // 4 - 2]).norm() / ref[0, :, : H // 4 - 2].norm()).item()
        err[(prec, "dark")] = dark
    print("\nstem forward N%d H%d Ci%d scale %g: rel-L2 vs f64  fp32-input MFMA %.2e  fp16 pieces %.2e;  dark half image %.2e / %.2e"
          % (N, H, Ci, scale, err[0], err[4], err[(0, "dark")], err[(4, "dark")]))
    assert err[4] <= 1.5 * err[0] + 1e-8 and err[(4, "dark")] <= 1.5 * err[(0, "dark")] + 1e-7, err
    # BatchNorm statistics of the output from the epilogue (what the model asks for), in two groups
    if not relu:
        d = K.conv_desc(N, H, H, Ci, 64, 7, 2, 3, in_nchw=Ci != 1, precision=4)
        sums = K.bn_stats_buffer(2, 64, "cuda")
        y2 = K.conv_fwd(xk, wk, None, d, bn_sums=sums, groups=2)
        yd = y2.double().reshape(2, -1, 64)
        tab = sums.reshape(K.BN_SUM_SLOTS, 2, 64, 2, K.BN_SUM_STRIDE)[..., 0].sum(0).cpu()
        refs = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1).cpu()
        assert ((tab - refs).abs() <= 1e-6 * refs.abs().max()).all()


@pytest.mark.parametrize("N,H,Ci,Co,groups", [(4, 8, 32, 32, 2), (8, 8, 64, 32, 2)])
def test_conv_transpose_fwd_with_batchnorm_sums(K, N, H, Ci, Co, groups):
    """ConvTranspose2d(2, 2) -> BatchNorm (the decoder's lower branch): statistics accumulated in the scatter epilogue."""
    x = torch.tensor(rnd((N, H, H, Ci), 90)).cuda()
    w = torch.tensor(rnd((Ci, 2, 2, Co), 91) * 0.1).cuda()
    d = K.conv_desc(N, H, H, Ci, Co, 2, 2, 0, transposed=True)
    sums = K.bn_stats_buffer(groups, Co, "cuda")
    y = K.conv_fwd(x, w, None, d, bn_sums=sums, groups=groups)
    y0 = K.conv_fwd(x, w, None, d)
    assert torch.equal(y, y0)
    yd = y0.double().reshape(groups, -1, Co)
    ref = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)
    close(sums.reshape(K.BN_SUM_SLOTS, groups, Co, 2, K.BN_SUM_STRIDE)[..., 0].sum(0).cpu(), ref.cpu(), 1e-6)


def test_c_abi_error_codes(K):
    """The C ABI reports misuse instead of computing something else: -1 = bad argument, -2 = unsupported geometry
    (include/bihome.h); the Python binding turns both into BihomeLibError."""
    import ctypes
    from bihome_amd._lib import BihomeLibError, lib
    x = torch.zeros(2, 8, 8, 32, device="cuda")
    w = torch.zeros(32, 3, 3, 32, device="cuda")
    d = K.conv_desc(2, 8, 8, 32, 32, 3, 1, 1)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = ctypes.c_void_p
    assert lib.bh_conv_fwd(None, P(w.data_ptr()), None, P(x.data_ptr()), ctypes.byref(d), s) == -1
    assert lib.bh_conv_dgrad_s2(P(x.data_ptr()), P(w.data_ptr()), P(x.data_ptr()), ctypes.byref(d), 0, P(w.data_ptr()), s) == -2
    bad = K.conv_desc(2, 8, 8, 32, 32, 3, 2, 1, transposed=True)          # ConvTranspose2d needs kernel == stride
    y = torch.zeros(2, 16, 16, 32, device="cuda")
    assert lib.bh_conv_wgrad(P(x.data_ptr()), P(y.data_ptr()), P(w.data_ptr()), None, ctypes.byref(bad), s) == -2
    st = K.bn_stats_buffer(1, 6, "cuda")
    z = torch.zeros(4, 6, device="cuda")
    assert lib.bh_bn_fwd(P(z.data_ptr()), None, None, None, None, None, P(z.data_ptr()), P(st.data_ptr()), 1, 4, 6, 1e-5, 0.1, 0, 0, s) == -2
    img = torch.zeros(1, 1, 24, 24, device="cuda")
    H = torch.eye(3, dtype=torch.float64, device="cuda").reshape(1, 9)
    assert lib.bh_warp_fwd(P(img.data_ptr()), P(H.data_ptr()), 1, 1, 24, 24, 4, P(img.data_ptr()), None, s) == -2
    with pytest.raises(BihomeLibError, match="BH_E_UNSUPPORTED"):
        K.warp_fwd(img, H)
    with pytest.raises(RuntimeError, match="no CPU"):
        K.conv_fwd(x.cpu(), w, None, d)


@pytest.mark.parametrize("N,H,Ci,Co,prec", [
    (128, 32, 64, 64, 0),       # the bench shape: two tile positions per workgroup, two chunks
    (8, 16, 128, 128, 0),       # four chunks, two n tiles
    (8, 8, 256, 256, 0),        # one sub-tile per workgroup
    (4, 64, 32, 32, 0),         # 32-channel tile, single chunk (one halo stage)
    (3, 24, 64, 32, 0),         # odd sub-tile count, 64 -> 32
    (5, 16, 32, 64, 0),         # single chunk, 64-wide tile, odd image count
    (8, 16, 64, 128, 1),        # bf16 operand mode
])
def test_conv3x3_packed_weights_match_lds_slab_path(K, N, H, Ci, Co, prec):
    """bh_conv3x3_pack + w_layout = 1 (B fragments streamed from the fragment-ordered copy into registers, one barrier per
    chunk) against the LDS-slab form of the same kernel: same MFMA order, so forward, forward + BatchNorm sums, dgrad and
    accumulating dgrad must be BIT-identical; and against torch float64."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    g = torch.Generator().manual_seed(N * 7 + H)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
    assert K.packs_3x3(d)
    pk = K.WeightPacker()
    pf, pd = pk.get(w)
    pk.refresh()
    dp = K._with_layout(d, 1)
    assert K.conv_variant(dp, "fwd").endswith(",true,false,false,3,false,false>") and K.conv_variant(dp, "dgrad").endswith(",true,false,false,3,false,false>")
    y0 = K.conv_fwd(x, wk, b, d)
    y1 = K.conv_fwd(x, wk, b, d, wpacked=pf)
    assert torch.equal(y0, y1)
    s0, s1 = K.bn_stats_buffer(1, Co, "cuda"), K.bn_stats_buffer(1, Co, "cuda")
    assert torch.equal(K.conv_fwd(x, wk, b, d, bn_sums=s0, groups=1), K.conv_fwd(x, wk, b, d, bn_sums=s1, groups=1, wpacked=pf))
    close(s1.cpu(), s0.cpu(), 1e-12)
    g0 = K.conv_dgrad(gy, wk, d)
    g1 = K.conv_dgrad(gy, wk, d, wpacked=pd)
    assert torch.equal(g0, g1)
    acc0, acc1 = x.clone(), x.clone()
    K.conv_dgrad(gy, wk, d, out=acc0)
    K.conv_dgrad(gy, wk, d, out=acc1, wpacked=pd)
    assert torch.equal(acc0, acc1)
    if prec == 0:
        ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
        close(y1.cpu(), ref, 3e-5)
    # a parameter update is picked up by the next refresh (version counter), without reallocating the buffers
    p0 = pf.data_ptr()
    with torch.no_grad():
        w.mul_(0.5)
    pk.refresh()
    assert pf.data_ptr() == p0
    y2 = K.conv_fwd(x, wk, None, d, wpacked=pf)
    assert torch.equal(y2, K.conv_fwd(x, wk, None, d))


@pytest.mark.parametrize("N,H,Ci,Co,wide", [
    (128, 32, 64, 64, False),    # the bench shape: two tile positions per workgroup, two chunks
    (8, 16, 128, 128, False),    # four chunks, two n tiles
    (8, 8, 256, 256, False),     # one sub-tile per workgroup, eight chunks
    (4, 64, 32, 32, False),      # 32-channel tile, single chunk (one halo stage)
    (3, 24, 96, 160, False),     # odd sub-tile count, three chunks, 64- and 32-wide n tiles
    (5, 16, 32, 64, False),      # single chunk, odd image count
    (8, 16, 64, 64, True),       # operands spread over 24 binades
])
def test_conv3x3_f32x3_is_fp32_accurate(K, N, H, Ci, Co, wide):
    """precision 2 / w_layout 2 of the packed 3x3 kernel: every fp32 operand is cut exactly into three bf16 pieces and six
    partial products per product run on the bf16 MFMA with fp32 accumulate.  Claim under test: this is fp32 arithmetic - the
    error against torch float64 is no larger than that of the fp32-input MFMA kernel on the same data (measured 0.85x), for
    forward, forward + BatchNorm sums, dgrad and accumulating dgrad; the piece split itself is exact (pack -> sum of pieces
    = weight, bit for bit)."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    g = torch.Generator().manual_seed(N * 7 + H)
    x = torch.randn(N, H, H, Ci, generator=g)
    gy = torch.randn(N, H, H, Co, generator=g)
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
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        pk = K.WeightPacker(split=prec == 2)
        pf, pd = pk.get(w)
        pk.refresh()
        dp = K._with_layout(d, 2 if prec == 2 else 1)
        assert K.conv_variant(dp, "fwd").endswith(",true,true,false,3,false,false>" if prec == 2 else ",true,false,false,3,false,false>")
        assert K.conv_variant(dp, "dgrad").endswith(",true,true,false,3,false,false>" if prec == 2 else ",true,false,false,3,false,false>")
        y = K.conv_fwd(x, wk, b, d, wpacked=pf)
        s = K.bn_stats_buffer(1, Co, "cuda")
        assert torch.equal(y, K.conv_fwd(x, wk, b, d, bn_sums=s, groups=1, wpacked=pf))
        yd = y.double()
        tot = torch.stack([yd.sum((0, 1, 2)), (yd * yd).sum((0, 1, 2))], 1).cpu()
        got = s.view(K.BN_SUM_SLOTS, 1, Co, 2, K.BN_SUM_STRIDE)[..., 0].sum(0).reshape(Co, 2).cpu()
        close(got, tot, 1e-6)                      # (the kernel sums float partials of 4 elements in double)
        gx = K.conv_dgrad(gy, wk, d, wpacked=pd)
        acc = x.clone()
        K.conv_dgrad(gy, wk, d, out=acc, wpacked=pd)
        close(acc.cpu(), (x + gx).cpu(), 1e-6)
        err[prec] = (((y.cpu().double() - ref).norm() / ref.norm()).item(), ((gx.cpu().double() - refd).norm() / refd.norm()).item())
        if prec == 2:
            # the three pieces of every packed weight add up to the weight exactly
            pieces = pf.view(torch.int16).view(-1, 3, 2, 64, 8).to(torch.int32) << 16        # [chunk*tap*ntile][piece][step][lane][e]
            total = pieces.view(torch.float32).double().sum(1)                               # exact in float64
            NW = Co // 32
            back = torch.empty(Co, 9, Ci, dtype=torch.float64, device="cuda")
            t = total.view(Ci // 32, 9, NW, 2, 2, 32, 8)                                     # [c][tap][nt][s2][kh2][l31][e]
            back.view(NW, 32, 9, Ci // 32, 2, 2, 8).copy_(t.permute(2, 5, 1, 0, 3, 4, 6))    # n = nt*32 + l31, k = c*32 + (2*s2 + kh2)*8 + e
            assert torch.equal(back.float(), wk.reshape(Co, 9, Ci)) and torch.equal(back, wk.reshape(Co, 9, Ci).double())
    assert err[2][0] <= 1.1 * err[0][0] + 1e-9 and err[2][1] <= 1.1 * err[0][1] + 1e-9, err
    assert err[2][0] < 2e-6 and err[2][1] < 2e-6


@pytest.mark.parametrize("N,H,Ci,Co", [
    (128, 32, 64, 64),      # the bench shape: 8 tiles per workgroup, 256 partial blocks
    (2, 8, 64, 64),         # one tile per image (image = tile: the whole halo ring is padding), fewer tiles than CUs
    (3, 24, 64, 128),       # non-power-of-two tile grid, two output-channel blocks
    (8, 8, 256, 128),       # 4 x 2 channel blocks
    (16, 16, 128, 128),
    (5, 40, 64, 64),        # odd image count, 5 x 5 tiles
    (4, 64, 32, 32),        # 32-channel block: the four waves split the pixels of a tile
    (3, 24, 32, 96),        # 32-channel blocks, 1 x 3 of them, non-power-of-two tile grid
    (2, 16, 64, 32),        # 64 -> 32: 32-channel blocks on both sides
])
def test_wgrad_f32x3_kernel(K, N, H, Ci, Co):
    """csrc/wgrad_x3.hip (precision 2: halo-tiled, all nine taps per workgroup, three exact bf16 pieces per operand, transposing
    LDS reads): against torch float64 with an error no larger than the fp32-input MFMA kernel's on the same data; accumulates
    onto gw; with a workspace the split-K reduction runs in fixed order - bitwise repeatable - and agrees with the atomics
    form to fp32 rounding."""
    g = torch.Generator().manual_seed(N + H + Ci)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w, None, 1, 1)
    ref = torch.autograd.grad(y, w, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)       # [Co][3][3][Ci]
    err = {}
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec)
        assert K.conv_variant(d, "wgrad").startswith("wgrad_x3_kernel") == (prec == 2)
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d)
        err[prec] = ((gw.cpu().double() - ref).norm() / ref.norm()).item()
        g1 = torch.ones(Co, 3, 3, Ci, device="cuda")                     # gw += : the result lands on what gw holds
        K.conv_wgrad(x, gy, g1, None, d)
        close((g1 - 1.0).cpu(), gw.cpu(), 2e-5)
    # (both errors are fp32 ACCUMULATION rounding over N*H*W terms - the products are exact in either form.  The f32x3 kernel adds
    #  up to 512 pixels per accumulator before the split-K reduction, the fp32-MFMA one 64, hence up to ~1.2x at the bench shape;
    #  without a workspace the 32-channel form adds 4 x 256 partial blocks with atomics one after the other - a longer chain of
    #  fp32 roundings, 6e-7 on 4x64x64 - while the workspace form the models use sums them in four groups, checked below)
    assert err[2] <= (1.5 if min(Ci, Co) % 64 == 0 else 3.0) * err[0] + 1e-8 and err[2] < 2e-6, err
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2)
    need = K.wgrad_det_bytes(d)
    assert 0 < need <= 40 << 20
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    runs = []
    for _ in range(3):
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        gb = torch.zeros(Co, device="cuda")
        K.conv_wgrad(x, gy, gw, gb, d, det_ws=ws)
        runs.append(gw)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    assert ((runs[0].cpu().double() - ref).norm() / ref.norm()).item() <= 1.5 * err[0] + 1e-8
    close(gb.cpu(), gy.double().sum((0, 1, 2)).cpu(), 1e-5)


def test_conv3x3_f32x3_error_bound_under_cancellation(K):
    """Worst case for a split-operand scheme: dot products whose terms cancel (the result is ~1e-4 of sum |a||b|), operands with
    all 24 significand bits set, and weights whose three pieces have mixed signs after the cut.  The ABSOLUTE error of every
    output against float64, in units of sum_k |a_k||b_k| * 2^-24 (one fp32 ulp of the magnitude being accumulated), must stay
    within a small constant for f32x3 - and does not exceed what the fp32-input MFMA kernel shows on the same data."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    N, H, Ci, Co = 8, 16, 64, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H, H, Ci, generator=g)
    x = (x.view(torch.int32) | 0x7FF).view(torch.float32)                   # low mantissa bits all set
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    w[:, 1::2] = -w[:, 0::2] * (1.0 + 1e-4 * torch.randn(Co, Ci // 2, 3, 3, generator=g))     # channel pairs nearly cancel when x is smooth
    x[..., 1::2] = x[..., 0::2] * (1.0 + 1e-4 * torch.randn(N, H, H, Ci // 2, generator=g))
    x, w = x.cuda(), w.cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
    ref = F.conv2d(xd, wd, None, 1, 1).permute(0, 2, 3, 1)
    mag = F.conv2d(xd.abs(), wd.abs(), None, 1, 1).permute(0, 2, 3, 1)      # sum |a||b| per output
    assert (ref.abs() / mag).median() < 1e-3                                  # the case really cancels
    ulps = {}
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        pk = K.WeightPacker(split=prec == 2)
        pf, _ = pk.get(w)
        pk.refresh()
        y = K.conv_fwd(x, wk, None, d, wpacked=pf)
        ulps[prec] = ((y.cpu().double() - ref).abs() / (mag * 2.0 ** -24)).max().item()
    assert ulps[2] <= max(1.25 * ulps[0], 4.0) and ulps[2] < 16.0, ulps      # (K = 576 terms: a few ulps of the accumulated magnitude)


@pytest.mark.parametrize("N,H,Ci,Co,relu", [
    (128, 32, 64, 64, True),     # bench shape, two statistics groups
    (8, 16, 128, 64, True),      # four chunks
    (8, 8, 256, 256, False),     # one sub-tile per workgroup, 4 KB table, no ReLU
    (4, 64, 32, 32, True),       # 32-channel tile, single chunk
    (6, 24, 64, 96, True),       # image borders inside every tile row, three images per group
])
def test_batchnorm_on_load_matches_materialised_batchnorm(K, N, H, Ci, Co, relu):
    """bh_bn_fwd_coeffs + bh_conv_fwd_bnin / bh_conv_wgrad_bnin: the consumer conv applies the BatchNorm(+ReLU) in front of it
    while staging its operand, so that BatchNorm's output is never stored.  Against the unfused sequence (bh_bn_fwd writes the
    activation, the same f32x3 kernels read it): forward, forward + BatchNorm sums of the conv's own output, weight gradient
    and the running-statistics update - equal up to the rounding of scale / shift arithmetic (fma vs separate ops), with the
    zero padding ring intact (the transform of 0 is `shift`, not 0: a wrong border shows at 1e-1)."""
    groups = 2
    g = torch.Generator().manual_seed(N + H + Ci)
    z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
    gamma = (torch.rand(Ci, generator=g) + 0.5).cuda()
    beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    rm0, rv0 = torch.randn(Ci, generator=g).cuda(), (torch.rand(Ci, generator=g) + 0.5).cuda()
    rm1, rv1 = rm0.clone(), rv0.clone()
    a, st = K.bn_fwd(z, gamma, beta, rm0, rv0, None, groups, 1e-5, 0.1, relu, True)
    table = K.bn_fwd_coeffs(st, gamma, beta, rm1, rv1, groups, N * H * H // groups, Ci, 1e-5, 0.1)
    assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2, route=0)
    if not K.packs_3x3(d):
        from bihome_amd._lib import ROUTE_HALO_SMALL
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2, route=ROUTE_HALO_SMALL)
    pk = K.WeightPacker(split=True)
    pf, _ = pk.get(w)
    pk.refresh()
    lazy = K.BnOnLoad(z, table, groups, relu)
    y_ref = K.conv_fwd(a, wk, b, d, wpacked=pf)
    y = K.conv_fwd(lazy, wk, b, d, wpacked=pf)
    close(y.cpu(), y_ref.cpu(), 2e-6)
    s0, s1 = K.bn_stats_buffer(groups, Co, "cuda"), K.bn_stats_buffer(groups, Co, "cuda")
    K.conv_fwd(a, wk, b, d, bn_sums=s0, groups=groups, wpacked=pf)
    y2 = K.conv_fwd(lazy, wk, b, d, bn_sums=s1, groups=groups, wpacked=pf)
    assert torch.equal(y2, y)
    close(s1.cpu(), s0.cpu(), 1e-5)
    need = K.wgrad_det_bytes(d)
    assert need > 0
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    g0, g1 = torch.zeros(Co, 3, 3, Ci, device="cuda"), torch.zeros(Co, 3, 3, Ci, device="cuda")
    gb0, gb1 = torch.zeros(Co, device="cuda"), torch.zeros(Co, device="cuda")
    K.conv_wgrad(a, gy, g0, gb0, d, det_ws=ws)
    K.conv_wgrad(lazy, gy, g1, gb1, d, det_ws=ws)
    close(g1.cpu(), g0.cpu(), 2e-6)
    close(gb1.cpu(), gb0.cpu(), 1e-5)


def test_conv3x3_f32x3_layout_and_precision_must_agree(K):
    """w_layout 2 (split weights) <-> precision 2: a mismatch is a caller error, not a silently wrong operand format; precision 2
    without packed weights computes as precision 0 (bit-identical to the fp32-input MFMA kernel)."""
    x = torch.randn(8, 16, 16, 64, device="cuda")
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    d0, d2 = K.conv_desc(8, 16, 16, 64, 64, 3, 1, 1, precision=0), K.conv_desc(8, 16, 16, 64, 64, 3, 1, 1, precision=2)
    assert torch.equal(K.conv_fwd(x, wk, None, d0), K.conv_fwd(x, wk, None, d2))
    pk = K.WeightPacker(split=True)
    pf, _ = pk.get(w)
    pk.refresh()
    from bihome_amd._lib import lib, ROUTE_HALO_SMALL
    import ctypes
    y = torch.empty(8, 16, 16, 64, device="cuda")
    for prec, lay in ((0, 2), (2, 1), (1, 2)):
        d = K._with_layout(K.conv_desc(8, 16, 16, 64, 64, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL), lay)
        rc = lib.bh_conv_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(pf.data_ptr()), None, ctypes.c_void_p(y.data_ptr()),
                             ctypes.byref(d), None)
        assert rc == -1, (prec, lay, rc)            # BH_E_BADARG


@pytest.mark.parametrize("N,H,Ci,Co", [(128, 32, 64, 64), (16, 16, 128, 128), (8, 8, 256, 256)])
def test_wgrad_deterministic_mode_is_bitwise_repeatable(K, N, H, Ci, Co):
    """bh_conv_wgrad_det: split-K partial tiles + a fixed-order second pass instead of fp32 atomics.  Two runs give
    bit-identical weight gradients (the default atomics path does not), the values match torch float64, and the result
    accumulates onto what gw already holds."""
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    need = K.wgrad_det_bytes(d)
    assert 0 < need <= 40 << 20
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    runs = []
    for _ in range(3):
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
        runs.append(gw.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    if N * H * H <= 8192:
        ref = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (Co, Ci, 3, 3), gy.double().cpu().permute(0, 3, 1, 2),
                                          stride=1, padding=1).permute(0, 2, 3, 1)
        close(runs[0].cpu(), ref, 2e-5)
    a = torch.zeros(Co, 3, 3, Ci, device="cuda")
    K.conv_wgrad(x, gy, a, None, d)                          # default (atomics) path: same values up to summation order
    close(runs[0].cpu(), a.cpu(), 2e-5)
    gw2 = runs[0].clone()
    K.conv_wgrad(x, gy, gw2, None, d, det_ws=ws)             # accumulates
    close(gw2.cpu(), 2 * runs[0].cpu(), 1e-6)
    assert K.wgrad_det_bytes(K.conv_desc(8, 64, 64, 32, 32, 3, 1, 1)) == 0      # small-channel layers: no deterministic form


@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_on_load_propagates_nan(K, relu):
    """Round-2 ADVICE: the BatchNorm-on-load staging clamps with a NaN-PROPAGATING maximum (v_maximum3_f32), as the materialised
    bn_fwd path and torch do - a diverging run must not be masked in exactly the layers that take the fused path (fmaxf would
    have turned the NaN into 0)."""
    groups, N, H, Ci, Co = 2, 4, 16, 64, 64
    g = torch.Generator().manual_seed(9)
    z = torch.randn(N, H, H, Ci, generator=g).cuda()
    gamma, beta = torch.ones(Ci).cuda(), torch.zeros(Ci).cuda()
    rm, rv = torch.zeros(Ci).cuda(), torch.ones(Ci).cuda()
    _, st = K.bn_fwd(z, gamma, beta, rm.clone(), rv.clone(), None, groups, 1e-5, 0.1, relu, True)       # statistics of the clean tensor
    table = K.bn_fwd_coeffs(st, gamma, beta, rm, rv, groups, N * H * H // groups, Ci, 1e-5, 0.1)
    z[1, 5, 7, 13] = float("nan")
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    from bihome_amd._lib import ROUTE_HALO_SMALL
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2, route=ROUTE_HALO_SMALL)
    pk = K.WeightPacker(split=True)
    pf, _ = pk.get(w)
    pk.refresh()
    y = K.conv_fwd(K.BnOnLoad(z, table, groups, relu), w.permute(0, 2, 3, 1), None, d, wpacked=pf)
    bad = torch.isnan(y).any(dim=-1).cpu()                 # [N,H,H]: exactly the 3x3 neighbourhood of the poisoned pixel
    want = torch.zeros(N, H, H, dtype=torch.bool)
    want[1, 4:7, 6:9] = True
    assert torch.equal(bad, want)
    need = K.wgrad_det_bytes(d)
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    K.conv_wgrad(K.BnOnLoad(z, table, groups, relu), torch.randn(N, H, H, Co, generator=g).cuda(), gw, None, d, det_ws=ws)
    nanc = torch.isnan(gw).any(dim=0).any(dim=0).any(dim=0).cpu()      # per input channel
    assert bool(nanc[13]) and int(nanc.sum()) == 1


@pytest.mark.parametrize("N,H,Ci,Co", [(128, 32, 64, 64), (8, 16, 128, 128), (8, 8, 256, 256), (4, 64, 32, 32), (3, 24, 96, 160)])
def test_conv3x3_f32x2_two_piece_mode(K, N, H, Ci, Co):
    """precision 3 / w_layout 3 ("f32x2"): two bf16 pieces per operand, both rounded to nearest, three MFMA products.  Stated
    accuracy: relative L2 error against torch float64 below 8e-6 for forward, dgrad and weight gradient (operand representation
    2^-18 = 3.8e-6 worst case, zero mean), i.e. >= 100x below the bf16-operand mode (2.4e-3) and ~10x above fp32; the two
    pieces of a packed weight reproduce it to 2^-17 relative."""
    from bihome_amd._lib import ROUTE_HALO_SMALL
    g = torch.Generator().manual_seed(N * 5 + H)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
    ref = F.conv2d(xd, wd, b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), wd, None, 1, 1).permute(0, 2, 3, 1)
    refw = torch.einsum("nyxo,nyxtc->otc", gy.double().cpu(),
                        F.unfold(xd, 3, padding=1).view(N, Ci, 9, H, H).permute(0, 3, 4, 2, 1)).reshape(Co, 3, 3, Ci)
    errs = {}
    for prec in (3, 2, 1):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        if prec == 1:
            y, gx = K.conv_fwd(x, wk, b, d), K.conv_dgrad(gy, wk, d)
        else:
            pk = K.WeightPacker(split=K.SPLIT_PIECES[prec])
            pf, pd = pk.get(w)
            pk.refresh()
            dp = K._with_layout(d, K.packed_layout(prec))
            assert K.conv_variant(dp, "fwd").endswith(",true,true,false,%d,false,false>" % K.SPLIT_PIECES[prec])
            y = K.conv_fwd(x, wk, b, d, wpacked=pf)
            s = K.bn_stats_buffer(1, Co, "cuda")
            assert torch.equal(y, K.conv_fwd(x, wk, b, d, bn_sums=s, groups=1, wpacked=pf))
            gx = K.conv_dgrad(gy, wk, d, wpacked=pd)
            acc = x.clone()
            K.conv_dgrad(gy, wk, d, out=acc, wpacked=pd)
            close(acc.cpu(), (x + gx).cpu(), 2e-5)
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        need = K.wgrad_det_bytes(d)
        if prec != 1 and need > 0:
            assert K.conv_variant(d, "wgrad_det").startswith("wgrad_x3_kernel<%d,false,%d,false,false,false>" % (64 if (Ci % 64 == 0 and Co % 64 == 0) else 32, K.SPLIT_PIECES[prec]))
            K.conv_wgrad(x, gy, gw, None, d, det_ws=torch.empty(need // 4, dtype=torch.float32, device="cuda"))
        else:
            K.conv_wgrad(x, gy, gw, None, d)
        errs[prec] = tuple(((a.cpu().double() - r).norm() / r.norm()).item() for a, r in ((y, ref), (gx, refd), (gw, refw)))
        if prec == 3:
            pieces = pf.view(torch.int16).view(-1, 2, 2, 64, 8).to(torch.int32) << 16        # [chunk*tap*ntile][piece][step][lane][e]
            total = pieces.view(torch.float32).double().sum(1)
            NW = Co // 32
            back = torch.empty(Co, 9, Ci, dtype=torch.float64, device="cuda")
            t = total.view(Ci // 32, 9, NW, 2, 2, 32, 8)
            back.view(NW, 32, 9, Ci // 32, 2, 2, 8).copy_(t.permute(2, 5, 1, 0, 3, 4, 6))
            wref = wk.reshape(Co, 9, Ci).double()
            assert ((back - wref).abs() <= wref.abs() * 2.0 ** -17).all()
            assert not torch.equal(back, wref)              # (it IS a reduced representation)
    print("f32x2 / f32x3 / bf16 relative L2 error (fwd, dgrad, wgrad):", errs[3], errs[2], errs[1])
    assert max(errs[3]) < 8e-6, errs
    # (forward and dgrad; the bf16-operand weight gradient of some shapes runs the fp32 kernel)
    assert all(e3 < e1 / 100 for e3, e1 in zip(errs[3][:2], errs[1][:2])), errs


@pytest.mark.parametrize("N,Ci,Co,prec,groups", [(128, 512, 512, 2, 2), (128, 512, 512, 3, 2), (16, 64, 128, 2, 1), (8, 192, 64, 2, 2)])
def test_conv3x3_halo_kernel_on_4x4_maps(K, N, Ci, Co, prec, groups):
    """The MAP4 form of the split-operand halo kernel (round 3): 4 x 4 feature maps (layer4 of the ResNet-34 regressor), four images
    with their own 6 x 6 zero-padded halos per 64-row sub-tile.  Forward (+ bias, + BatchNorm sums), dgrad, accumulating dgrad and
    dgrad with the BatchNorm backward sums in the epilogue against torch float64 and against the generic kernel."""
    from bihome_amd._lib import ROUTE_GENERIC_CONV, ROUTE_HALO_SMALL
    g = torch.Generator().manual_seed(N + Ci)
    x = torch.randn(N, 4, 4, Ci, generator=g).cuda()
    gy = torch.randn(N, 4, 4, Co, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.03).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    d = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
    dp = K._with_layout(d, K.packed_layout(prec))
    assert K.conv_variant(dp, "fwd").endswith(",%d,true,false>" % K.SPLIT_PIECES[prec]) and K.conv_variant(dp, "dgrad").endswith(",%d,true,false>" % K.SPLIT_PIECES[prec])
    assert K.packs_3x3(d)
    pk = K.WeightPacker(split=K.SPLIT_PIECES[prec])
    pf, pd = pk.get(w)
    pk.refresh()
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    tol = 2e-6 if prec == 2 else 1e-5
    y = K.conv_fwd(x, wk, b, d, wpacked=pf)
    assert ((y.cpu().double() - ref).norm() / ref.norm()).item() < tol
    s = K.bn_stats_buffer(groups, Co, "cuda")
    assert torch.equal(y, K.conv_fwd(x, wk, b, d, bn_sums=s, groups=groups, wpacked=pf))
    yd = y.double().view(groups, -1, Co)
    tot = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1).cpu()
    got = s.view(groups, Co, 2, K.BN_SUM_STRIDE)[..., 0].cpu()
    close(got, tot, 1e-6)
    gx = K.conv_dgrad(gy, wk, d, wpacked=pd)
    assert ((gx.cpu().double() - refd).norm() / refd.norm()).item() < tol
    acc = x.clone()
    K.conv_dgrad(gy, wk, d, out=acc, wpacked=pd)
    close(acc.cpu(), (x + gx).cpu(), 2e-5)
    # the generic kernel on the same data (what these layers ran on before)
    dg = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=0, route=ROUTE_GENERIC_CONV)
    close(y.cpu(), K.conv_fwd(x, wk, b, dg).cpu(), 2e-5)
    close(gx.cpu(), K.conv_dgrad(gy, wk, dg).cpu(), 2e-5)


@pytest.mark.parametrize("groups,N,H,C,relu", [(2, 2, 16, 16, True), (1, 3, 8, 64, True), (2, 2, 8, 128, False), (2, 1, 32, 32, True)])
def test_two_branch_batchnorm_join(K, groups, N, H, C, relu):
    """y = act(bn_a(xa) + bn_b(xb)) (round 4: the end of a decoder / strided residual unit in one pass, src/backbones/utils.py:60-82) and
    its adjoint against the unfused float64 PyTorch chain, one BatchNorm call per group; running statistics of both BatchNorms."""
    NN = groups * N
    xa = (rnd((NN, C, H, H), 60) * 1.3 + 0.2).astype(np.float32)
    xb = (rnd((NN, C, H, H), 61) * 0.7 - 0.1).astype(np.float32)
    D = lambda a: torch.tensor(a, dtype=torch.float64)
    bna, bnb = torch.nn.BatchNorm2d(C).double(), torch.nn.BatchNorm2d(C).double()
    for i, bn in enumerate((bna, bnb)):
        bn.weight.data, bn.bias.data = D(1 + 0.3 * rnd((C,), 62 + i)), D(0.3 * rnd((C,), 64 + i))
    xat, xbt = D(xa).requires_grad_(True), D(xb).requires_grad_(True)
    outs = []
    for g in range(groups):
        o = bna(xat[g * N:(g + 1) * N]) + bnb(xbt[g * N:(g + 1) * N])
        outs.append(F.relu(o) if relu else o)
    ref = torch.cat(outs, 0)
    C_ = lambda a: torch.tensor(a).cuda()
    ma, mb = torch.nn.BatchNorm2d(C).cuda(), torch.nn.BatchNorm2d(C).cuda()
    for m, bn in ((ma, bna), (mb, bnb)):
        m.weight.data, m.bias.data = bn.weight.data.float().cuda(), bn.bias.data.float().cuda()
        m.weight.grad, m.bias.grad = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    xag, xbg = C_(xa).permute(0, 2, 3, 1).contiguous(), C_(xb).permute(0, 2, 3, 1).contiguous()
    sa, sb = K.bn_stats_buffer(groups, C, "cuda"), K.bn_stats_buffer(groups, C, "cuda")
    K.bn_stats(xag, sa, groups, C); K.bn_stats(xbg, sb, groups, C)
    y = K.bn_join_fwd(xag, xbg, ma, mb, sa, sb, groups, relu, 0.1, 0.1)
    close(y.cpu().permute(0, 3, 1, 2), ref.detach(), 2e-6)
    close(ma.running_mean.cpu(), bna.running_mean, 1e-5); close(mb.running_var.cpu(), bnb.running_var, 1e-5)
    gy = rnd(tuple(ref.shape), 66)
    near = (torch.cat(outs, 0).detach().abs() < 1e-5).numpy() if relu else np.zeros(gy.shape, bool)
    gy = np.where(near, 0, gy).astype(np.float32)
    ref.backward(D(gy))
    gxa, gxb = K.bn_join_bwd(C_(gy).permute(0, 2, 3, 1).contiguous(), y, xag, xbg, ma, mb, sa, sb, groups, relu, True, True)
    close(gxa.cpu().permute(0, 3, 1, 2), xat.grad, 3e-5)
    close(gxb.cpu().permute(0, 3, 1, 2), xbt.grad, 3e-5)
    close(ma.weight.grad.cpu(), bna.weight.grad, 3e-5); close(ma.bias.grad.cpu(), bna.bias.grad, 3e-5)
    close(mb.weight.grad.cpu(), bnb.weight.grad, 3e-5); close(mb.bias.grad.cpu(), bnb.bias.grad, 3e-5)
    # round 5: K.bn_join_bwd recomputes the ReLU mask from xa, xb (bh_bn_join_bwd_remask); the form that reads y gives bitwise the same
    import os
    os.environ["BIHOME_JOIN_REMASK"] = "0"
    try:
        gxa0, gxb0 = K.bn_join_bwd(C_(gy).permute(0, 2, 3, 1).contiguous(), y, xag, xbg, ma, mb, sa, sb, groups, relu, False, False)
    finally:
        del os.environ["BIHOME_JOIN_REMASK"]
    assert torch.equal(gxa0, gxa) and torch.equal(gxb0, gxb)


@pytest.mark.parametrize("N,H,C,relu", [(4, 32, 32, True), (2, 64, 32, False), (6, 24, 64, True), (16, 64, 32, True)])
def test_batchnorm_adjoint_rebuilds_the_1x1_dgrad(K, N, H, C, relu):
    """bh_bn_bwd_from_1x1 (round 5: BatchNorm + ReLU in front of the decoder units' 1x1 conv with 16 output channels): the BatchNorm's
    output gradient g = gs w is rebuilt per element and never stored - against conv_dgrad (1x1) followed by bn_bwd with the mask recomputed
    from x: input gradient, parameter gradients, magnitude record; deterministic call repeatable."""
    groups, KC = 2, 16
    g = torch.Generator().manual_seed(N + H + C)
    x = (torch.randn(N, H, H, C, generator=g) * 1.2 + 0.2).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    st = K.bn_stats_buffer(groups, C, "cuda")
    K.bn_stats(x, st, groups, C)
    w = (torch.randn(KC, 1, 1, C, generator=g) * 0.2).cuda()
    gs = torch.randn(N, H, H, KC, generator=g).cuda()
    d = K.conv_desc(N, H, H, C, KC, 1, 1, 0, precision=0)
    gfull = K.conv_dgrad(gs, w, d)
    gg0, gb0, gg1, gb1 = (torch.zeros(C, device="cuda") for _ in range(4))
    rec0, rec1 = K.amax_record("cuda"), K.amax_record("cuda")
    ref, _ = K.bn_bwd(gfull, None, x, gamma, st, rm, rv, groups, 1e-5, relu, True, False, gg0, gb0, beta=beta, had_res=False, amax=rec0)
    out = K.bn_bwd_from_1x1(K.GradFrom1x1(gs, w), x, gamma, beta, st, groups, 1e-5, relu, gg1, gb1, amax=rec1)
    close(out.cpu(), ref.cpu(), 5e-6)
    close(gg1.cpu(), gg0.cpu(), 5e-6); close(gb1.cpu(), gb0.cpu(), 5e-6)
    assert abs(float(rec1.max()) - float(out.abs().max())) == 0.0
    # float64 reference of the whole chain
    xd = x.double().cpu().reshape(groups, -1, C)
    gd = (gs.double().cpu().reshape(-1, KC) @ w.double().cpu().reshape(KC, C)).reshape(groups, -1, C)
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    xh = (xd - mu) / torch.sqrt(var + 1e-5)
    y = xh * gamma.double().cpu() + beta.double().cpu()
    dd = gd * (y > 0) if relu else gd
    gx64 = gamma.double().cpu() / torch.sqrt(var + 1e-5) * (dd - dd.mean(1, keepdim=True) - xh * (dd * xh).mean(1, keepdim=True))
    close(out.cpu().reshape(groups, -1, C), gx64, 2e-5)
    with K.det_scope(True):
        a1 = K.bn_bwd_from_1x1(K.GradFrom1x1(gs, w), x, gamma, beta, st, groups, 1e-5, relu)
        a2 = K.bn_bwd_from_1x1(K.GradFrom1x1(gs, w), x, gamma, beta, st, groups, 1e-5, relu)
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("N,H,Ci,Co,relu", [(4, 32, 32, 16, True), (2, 64, 64, 32, True), (6, 16, 32, 64, False), (128, 32, 32, 16, True)])
def test_batchnorm_on_load_1x1_matches_materialised_batchnorm(K, N, H, Ci, Co, relu):
    """Round 4: the 1x1 conv of a decoder unit behind BatchNorm + ReLU (src/backbones/utils.py:60-82) - generic forward kernel and
    small-channel weight-gradient kernel apply the BatchNorm while staging their operand.  Against the unfused sequence: forward,
    forward + BatchNorm sums of the conv's own output, weight gradient, in the default and in a deterministic call."""
    groups = 2
    g = torch.Generator().manual_seed(N + H + Ci)
    z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
    gamma = (torch.rand(Ci, generator=g) + 0.5).cuda()
    beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
    w = (torch.randn(Co, Ci, 1, 1, generator=g) * 0.2).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    rm, rv = torch.zeros(Ci).cuda(), torch.ones(Ci).cuda()
    a, st = K.bn_fwd(z, gamma, beta, rm, rv, None, groups, 1e-5, 0.1, relu, True)
    table = K.bn_fwd_coeffs(st, gamma, beta, rm.clone(), rv.clone(), groups, N * H * H // groups, Ci, 1e-5, 0.1)
    lazy = K.BnOnLoad(z, table, groups, relu)
    for det_on in (False, True):
        with K.det_scope(det_on):
            d = K.conv_desc(N, H, H, Ci, Co, 1, 1, 0)
            y_ref = K.conv_fwd(a, wk, None, d)
            y = K.conv_fwd(lazy, wk, None, d)
            close(y.cpu(), y_ref.cpu(), 2e-6)
            s0, s1 = K.bn_stats_buffer(groups, Co, "cuda"), K.bn_stats_buffer(groups, Co, "cuda")
            K.conv_fwd(a, wk, None, d, bn_sums=s0, groups=groups)
            y2 = K.conv_fwd(lazy, wk, None, d, bn_sums=s1, groups=groups)
            assert torch.equal(y2, y)
            o0, _ = K.bn_fwd(y_ref, None, None, torch.zeros(Co).cuda(), torch.ones(Co).cuda(), None, groups, 1e-5, 0.1, False, True, stats=s0, stats_ready=True)
            o1, _ = K.bn_fwd(y2, None, None, torch.zeros(Co).cuda(), torch.ones(Co).cuda(), None, groups, 1e-5, 0.1, False, True, stats=s1, stats_ready=True)
            close(o1.cpu(), o0.cpu(), 1e-5)
            assert K.conv_variant(d, "wgrad").startswith("wgrad_small_kernel<true>")
            ws = torch.empty(max(K.wgrad_det_bytes(d), 4) // 4 + 64, dtype=torch.float32, device="cuda") if det_on else None
            g0, g1 = torch.zeros(Co, 1, 1, Ci, device="cuda"), torch.zeros(Co, 1, 1, Ci, device="cuda")
            K.conv_wgrad(a, gy, g0, None, d, det_ws=ws)
            K.conv_wgrad(lazy, gy, g1, None, d, det_ws=ws)
            close(g1.cpu(), g0.cpu(), 3e-6)


@pytest.mark.parametrize("groups,N,H,C,training", [(2, 4, 16, 64, True), (1, 3, 10, 32, True), (2, 2, 64, 64, False)])
def test_batchnorm_relu_maxpool_in_one_pass(K, groups, N, H, C, training):
    """bh_bn_maxpool_fwd (round 4: conv1-bn1-relu-maxpool of the stems) against bh_bn_fwd followed by bh_maxpool3s2_fwd: pooled values, arg-max
    positions, running statistics; and the adjoint through the saved positions + the BatchNorm backward with the mask recomputed from x
    equals the unfused adjoint."""
    g = torch.Generator().manual_seed(H + C)
    x = (torch.randn(N, H, H, C, generator=g) * 1.2 + 0.1).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    rm0, rv0 = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    rm1, rv1 = rm0.clone(), rv0.clone()
    y, st = K.bn_fwd(x, gamma, beta, rm0, rv0, None, groups, 1e-5, 0.1, True, training)
    p_ref, i_ref = K.maxpool_fwd(y)
    st1 = K.bn_stats_buffer(groups, C, "cuda")
    fused = K.bn_maxpool_fwd(x, gamma, beta, rm1, rv1, groups, 1e-5, 0.1, True, training, st1, False)
    close(fused.pooled.cpu(), p_ref.cpu(), 2e-6)
    assert (fused.idx != i_ref).float().mean().item() < 1e-3          # (ties / values within a rounding of each other may pick the other tap)
    close(rm1.cpu(), rm0.cpu(), 1e-6); close(rv1.cpu(), rv0.cpu(), 1e-6)
    gp = torch.randn(p_ref.shape, generator=g).cuda()
    gy_ref = K.maxpool_bwd(i_ref, gp, tuple(y.shape))
    gy = K.maxpool_bwd(fused.idx, gp, fused.shape)
    gx_ref, _ = K.bn_bwd(gy_ref, y, x, gamma, st, rm0, rv0, groups, 1e-5, True, training, False, beta=beta, had_res=False)
    gx, _ = K.bn_bwd(gy, None, x, gamma, st1, rm1, rv1, groups, 1e-5, True, training, False, beta=beta, had_res=False)
    assert ((gx - gx_ref).norm() / gx_ref.norm()).item() < 2e-3       # (a different tap of a tie moves single elements)
    # round 5: the adjoint in one call (bh_bn_maxpool_bwd - the full-resolution gradient is never stored): bitwise the two-call result, also
    # the parameter gradients and the magnitude record
    gg0, gb0, gg1, gb1 = (torch.zeros(C, device="cuda") for _ in range(4))
    rec0, rec1 = K.amax_record("cuda"), K.amax_record("cuda")
    gx0, _ = K.bn_bwd(gy, None, x, gamma, st1, rm1, rv1, groups, 1e-5, True, training, False, gg0, gb0, beta=beta, had_res=False, amax=rec0)
    gx1 = K.bn_maxpool_bwd(K.PooledGrad(gp, fused.idx), x, gamma, beta, st1, rm1, rv1, groups, 1e-5, True, training, gg1, gb1, amax=rec1)
    assert torch.equal(gx1, gx0) and torch.equal(gg1, gg0) and torch.equal(gb1, gb0)
    assert float(rec1.max()) == float(rec0.max()) == float(gx0.abs().max())
