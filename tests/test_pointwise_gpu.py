"""pw_kernel (csrc/pointwise.hip, round 5): the persistent streaming kernel that takes the decoder's 1x1 convolutions and 2x2 / stride-2 transposed
convolutions with 32 or 64 input channels on maps of >= 65536 pixels - against torch float64 (values, BatchNorm sums of the output, magnitude
record, BatchNorm-on-load), on every (K, columns) instantiation, one and two statistics groups, a map whose sides are not powers of two."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bihome_amd import kernels
    return kernels


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# (kind, N, H, W, Ci, Co): M = N H W >= 65536
CASES = [("convT", 16, 64, 64, 32, 32), ("convT", 16, 64, 64, 32, 16), ("convT", 16, 64, 64, 64, 32), ("convT", 16, 64, 64, 64, 64),
         ("convT", 18, 48, 80, 32, 32), ("1x1", 16, 64, 64, 32, 16), ("1x1", 16, 64, 64, 64, 32), ("1x1", 4, 128, 128, 32, 16)]


def _ref(kind, x, w, b):
    xd = x.double().cpu().permute(0, 3, 1, 2)
    if kind == "convT":
        y = F.conv_transpose2d(xd, w.double().cpu().permute(0, 3, 1, 2), b.double().cpu(), stride=2)     # w [Ci][2][2][Co] -> [Ci][Co][2][2]
    else:
        y = F.conv2d(xd, w.double().cpu().permute(0, 3, 1, 2), b.double().cpu())                         # w [Co][1][1][Ci]
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("kind,N,H,W,Ci,Co", CASES)
@pytest.mark.parametrize("groups", [1, 2])
def test_pointwise_forward_against_float64(K, kind, N, H, W, Ci, Co, groups):
    g = torch.Generator().manual_seed(N + H + Ci + Co)
    x = torch.randn(N, H, W, Ci, generator=g).cuda()
    b = torch.randn(Co, generator=g).cuda()
    if kind == "convT":
        d = K.conv_desc(N, H, W, Ci, Co, 2, 2, 0, transposed=True, precision=4)
        w = (torch.randn(Ci, 2, 2, Co, generator=g) * 0.1).cuda()
    else:
        d = K.conv_desc(N, H, W, Ci, Co, 1, 1, 0, precision=4)
        w = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.1).cuda()
    assert K.conv_variant(d, "fwd").startswith("pw_kernel<%d," % Ci), K.conv_variant(d, "fwd")
    ref = _ref(kind, x, w, b)
    y = K.conv_fwd(x, w, b, d)
    assert _rel(y, ref) < 2e-6
    # with the BatchNorm sums of the output (what the decoder's lower branch asks for) and the magnitude record
    s = K.bn_stats_buffer(groups, Co, "cuda")
    y2 = K.conv_fwd(x, w, b, d, bn_sums=s, groups=groups)
    assert torch.equal(y2, y)
    yd = ref.reshape(groups, -1, Co)
    tab = s.reshape(groups, Co, 2, -1)[..., 0].cpu()
    assert (tab[..., 0] - yd.sum(1)).abs().max().item() <= 2e-6 * yd.abs().sum(1).max().item()
    assert (tab[..., 1] - (yd * yd).sum(1)).abs().max().item() <= 2e-6 * (yd * yd).sum(1).max().item()
    rec = K.amax_record("cuda")
    y3 = K.conv_fwd(x, w, b, d, amax=rec)
    assert torch.equal(y3, y) and float(rec.max()) == float(y.abs().max())


@pytest.mark.parametrize("kind,N,H,W,Ci,Co", [CASES[0], CASES[3], CASES[5]])
def test_pointwise_deterministic_call_repeats_bitwise(K, kind, N, H, W, Ci, Co):
    """BH_ROUTE_DETERMINISTIC: the statistics leave through the integer limbs - two calls give bitwise the same sums buffer (and values)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, W, Ci, generator=g).cuda()
    b = torch.randn(Co, generator=g).cuda()
    with K.det_scope(True):
        if kind == "convT":
            d = K.conv_desc(N, H, W, Ci, Co, 2, 2, 0, transposed=True, precision=4)
            w = (torch.randn(Ci, 2, 2, Co, generator=g) * 0.1).cuda()
        else:
            d = K.conv_desc(N, H, W, Ci, Co, 1, 1, 0, precision=4)
            w = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.1).cuda()
        outs = []
        for _ in range(2):
            s = K.bn_stats_buffer(2, Co, "cuda")
            y = K.conv_fwd(x, w, b, d, bn_sums=s, groups=2)
            outs.append((y, s))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1].view(torch.int64), outs[1][1].view(torch.int64))
    # and the limbs hold the same totals as the default call's doubles
    s0 = K.bn_stats_buffer(2, Co, "cuda")
    K.conv_fwd(x, w, b, K.conv_desc(N, H, W, Ci, Co, 2 if kind == "convT" else 1, 2 if kind == "convT" else 1, 0, transposed=kind == "convT", precision=4),
               bn_sums=s0, groups=2)
    st = K.bn_stats_buffer(2, Co, "cuda")
    yb = outs[0][0]
    K.bn_stats(yb, st, 2, Co)
    tab0 = s0.reshape(2, Co, 2, -1)[..., 0].cpu()
    tab1 = st.reshape(2, Co, 2, -1)[..., 0].cpu()
    assert (tab0 - tab1).abs().max().item() <= 1e-9 * tab1.abs().max().item() + 1e-6


@pytest.mark.parametrize("N,H,Ci,Co", [(16, 64, 32, 16), (16, 64, 64, 32), (4, 128, 32, 16)])
@pytest.mark.parametrize("relu", [True, False])
def test_pointwise_batchnorm_on_load(K, N, H, Ci, Co, relu):
    g = torch.Generator().manual_seed(H + Ci)
    groups = 2
    z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.2).cuda()
    rm, rv = torch.zeros(Ci).cuda(), torch.ones(Ci).cuda()
    st = K.bn_stats_buffer(groups, Ci, "cuda")
    K.bn_stats(z, st, groups, Ci)
    rec = K.amax_record("cuda")
    table = K.bn_fwd_coeffs(st, gamma, beta, rm, rv, groups, N * H * H // groups, Ci, 1e-5, 0.1, amax=rec)
    lazy = K.BnOnLoad(z, table, groups, relu, amax=rec)
    d = K.conv_desc(N, H, H, Ci, Co, 1, 1, 0, precision=4)
    w = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.1).cuda()
    b = torch.randn(Co, generator=g).cuda()
    s = K.bn_stats_buffer(groups, Co, "cuda")
    y = K.conv_fwd(lazy, w, b, d, bn_sums=s, groups=groups)
    # reference: training-mode BatchNorm per group in float64, then the 1x1 conv
    zd = z.double().cpu().reshape(groups, -1, Ci)
    mu, var = zd.mean(1, keepdim=True), zd.var(1, unbiased=False, keepdim=True)
    a = (zd - mu) / torch.sqrt(var + 1e-5) * gamma.double().cpu() + beta.double().cpu()
    if relu:
        a = a.clamp_min(0)
    ref = a.reshape(N, H, H, Ci) @ w.double().cpu().reshape(Co, Ci).t() + b.double().cpu()
    assert _rel(y, ref) < 3e-6
    yd = ref.reshape(groups, -1, Co)
    tab = s.reshape(groups, Co, 2, -1)[..., 0].cpu()
    assert (tab[..., 0] - yd.sum(1)).abs().max().item() <= 3e-6 * yd.abs().sum(1).max().item()


def test_pointwise_statistics_and_table_groups_may_differ(K):
    """bh_conv_fwd_bnin takes the layout of `sums` (groups) and of the BatchNorm-on-load table (bni groups) as independent arguments
    (round-5 ADVICE): with sums in ONE group and a table of TWO the streaming kernel declines (it walks one partition of the images) and the
    generic kernel - which keeps the two counts apart - makes the launch; the sums land in the caller's [1][Co][2] entries."""
    N, H, Ci, Co = 16, 64, 32, 16
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).cuda(), (torch.randn(Ci, generator=g) * 0.2).cuda()
    rm, rv = torch.zeros(Ci).cuda(), torch.ones(Ci).cuda()
    st = K.bn_stats_buffer(2, Ci, "cuda")
    K.bn_stats(z, st, 2, Ci)
    rec = K.amax_record("cuda")
    table = K.bn_fwd_coeffs(st, gamma, beta, rm, rv, 2, N * H * H // 2, Ci, 1e-5, 0.1, amax=rec)
    lazy = K.BnOnLoad(z, table, 2, True, amax=rec)
    d = K.conv_desc(N, H, H, Ci, Co, 1, 1, 0, precision=4)
    w = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.1).cuda()
    b = torch.randn(Co, generator=g).cuda()
    s = K.bn_stats_buffer(1, Co, "cuda")
    guard = s.clone()
    y = K.conv_fwd(lazy, w, b, d, bn_sums=s, groups=1)
    zd = z.double().cpu().reshape(2, -1, Ci)
    mu, var = zd.mean(1, keepdim=True), zd.var(1, unbiased=False, keepdim=True)
    a = ((zd - mu) / torch.sqrt(var + 1e-5) * gamma.double().cpu() + beta.double().cpu()).clamp_min(0)
    ref = a.reshape(N, H, H, Ci) @ w.double().cpu().reshape(Co, Ci).t() + b.double().cpu()
    assert _rel(y, ref) < 3e-6
    yd = ref.reshape(1, -1, Co)
    tab = s.reshape(1, Co, 2, -1)[..., 0].cpu()
    assert (tab[..., 0] - yd.sum(1)).abs().max().item() <= 3e-6 * yd.abs().sum(1).max().item()
    assert (tab[..., 1] - (yd * yd).sum(1)).abs().max().item() <= 3e-6 * (yd * yd).sum(1).max().item()
    assert s.shape == guard.shape
