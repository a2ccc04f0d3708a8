"""Round 3: the Zhang "Content-Aware" baseline on the HIP path - ContentAware backbone (trainable 1 -> 4 -> 8 -> 1-channel feature
extractor with a ONE-channel BatchNorm, all-ones mask, resnet34) under the TripletHead - against the fixture produced by the
reference's own src/backbones/ContentAware.py + src/heads/TripletHead.py (tests/golden/zhang_orig_b4_*.npz, oracle/make_golden.py
--round3), and its two new kernels against torch float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bihome_amd import configs, synth
from bihome_amd.weights import load_synthetic

pytestmark = pytest.mark.gpu


def cuda(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dtype).cuda()


def relerr(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)


@pytest.mark.parametrize("groups,N,H,relu,res,training", [(2, 4, 32, True, False, True), (1, 3, 17, False, True, True),
                                                          (2, 2, 16, True, True, True), (1, 2, 16, True, False, False)])
def test_one_channel_batchnorm(groups, N, H, relu, res, training):
    """bh_bn_fwd / bh_bn_bwd with C = 1 (csrc/bn1.hip) against torch float64 autograd: output, running statistics, input / residual
    gradients, dgamma / dbeta; the mask recomputed from x where there is no residual."""
    from bihome_amd import kernels as K
    g = torch.Generator().manual_seed(N + H)
    x = (torch.randn(N, 1, H, H, generator=g) * 1.7 + 0.4)
    r = torch.randn(N, 1, H, H, generator=g) if res else None
    gy = torch.randn(N, 1, H, H, generator=g)
    gamma, beta = torch.tensor([1.3]), torch.tensor([-0.2])
    rm, rv = torch.tensor([0.1]), torch.tensor([0.9])
    # reference: one F.batch_norm call per statistics group, in order
    xd = x.double().requires_grad_(True)
    rd = r.double().requires_grad_(True) if res else None
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    outs = []
    for q in range(groups):
        sl = slice(q * N // groups, (q + 1) * N // groups)
        o = F.batch_norm(xd[sl], rmd, rvd, gd, bd, training, 0.1, 1e-5)
        if res:
            o = o + rd[sl]
        outs.append(torch.relu(o) if relu else o)
    yref = torch.cat(outs, 0)
    yref.backward(gy.double())
    xc, gc, bc = x.cuda(), gamma.cuda(), beta.cuda()
    rmc, rvc = rm.cuda(), rv.cuda()
    y, st = K.bn_fwd(xc, gc, bc, rmc, rvc, r.cuda() if res else None, groups, 1e-5, 0.1, relu, training)
    assert relerr(y.cpu(), yref.detach()) < 2e-6
    if training:
        assert relerr(rmc.cpu(), rmd) < 1e-6 and relerr(rvc.cpu(), rvd) < 1e-6
    gg, gb = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    gx, gres = K.bn_bwd(gy.cuda(), y, xc, gc, st, rmc, rvc, groups, 1e-5, relu, training, res, gg, gb, beta=bc)
    assert relerr(gx.cpu(), xd.grad) < 2e-5
    assert relerr(gg.cpu(), gd.grad) < 2e-5 and relerr(gb.cpu(), bd.grad) < 2e-5
    if res:
        assert relerr(gres.cpu(), rd.grad) < 1e-6


@pytest.mark.parametrize("B,h,margin,double", [(3, 32, 1.0, True), (2, 16, "inf", True), (2, 32, 0.5, False)])
def test_zhang_triplet_kernels(B, h, margin, double):
    """bh_zhang_triplet_fwd / bwd + bh_bihome_loss_fwd against torch float64 autograd of TripletHead.py:75-152 written per sample
    (without the reference's B-fold broadcast, which the head applies as a factor)."""
    from bihome_amd import kernels as K
    g = torch.Generator().manual_seed(B * h)
    f = [torch.randn(B, 1, h, h, generator=g) for _ in range(4)]                 # f1, f2, f1w, f2w
    m = [torch.rand(B, h, h, generator=g) for _ in range(2)]                     # m1w, m2w
    hinge = not isinstance(margin, str)
    fd = [t.double().requires_grad_(True) for t in f]
    md = [t.double().requires_grad_(True) for t in m]

    def line(fw, fo, fs, mw):
        t = (fw - fo).abs().sum(1) - (fs - fo).abs().sum(1)
        if hinge:
            t = torch.clamp(t + margin, min=0)
        den = mw.sum((-1, -2))
        return ((mw * t).sum((-1, -2)) / torch.max(den, torch.ones_like(den))).sum()
    ref = line(fd[2], fd[1], fd[0], md[0])
    if double:
        ref = ref + line(fd[3], fd[0], fd[1], md[1])
    ref.backward()
    fc, mc = [t.cuda() for t in f], [t.cuda() for t in m]
    _check_unwarped_mask_gradients(K, f, m, margin, hinge, double, g)
    T1, T2, numden = K.zhang_triplet_fwd(fc[0], fc[1], fc[2], fc[3] if double else None, mc[0], mc[1] if double else None,
                                         margin if hinge else 0.0, hinge)
    eye = torch.eye(3, dtype=torch.float64, device="cuda").reshape(1, 9).expand(B, 9).contiguous()
    loss4 = K.bihome_loss_fwd(numden, eye, eye, 0.0)
    assert abs(loss4[0].item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-6
    one = torch.ones(1, device="cuda")
    g_f1, g_f2, g_f1w, g_f2w, g_m1w, g_m2w = K.zhang_triplet_bwd(one, fc[0], fc[1], fc[2], fc[3] if double else None, mc[0],
                                                                mc[1] if double else None, T1, T2, numden, hinge)
    assert relerr(g_f1.cpu(), fd[0].grad) < 1e-5 and relerr(g_f2.cpu(), fd[1].grad) < 1e-5
    assert relerr(g_f1w.cpu(), fd[2].grad) < 1e-5 and relerr(g_m1w.cpu(), md[0].grad) < 1e-5
    if double:
        assert relerr(g_f2w.cpu(), fd[3].grad) < 1e-5 and relerr(g_m2w.cpu(), md[1].grad) < 1e-5


def _check_unwarped_mask_gradients(K, f, m, margin, hinge, double, gen):
    """Trained masks (round 4): the unwarped masks m2 (line 1) / m1 (line 2) enter bh_zhang_triplet_fwd and bh_zhang_triplet_bwd_m returns
    their gradients as well - against torch float64 autograd."""
    B, h = f[0].shape[0], f[0].shape[-1]
    u = [torch.rand(B, h, h, generator=gen) for _ in range(2)]                   # m1, m2
    fd = [t.double().requires_grad_(True) for t in f]
    md = [t.double().requires_grad_(True) for t in m]
    ud = [t.double().requires_grad_(True) for t in u]

    def line(fw, fo, fs, mw, mo):
        t = (fw - fo).abs().sum(1) - (fs - fo).abs().sum(1)
        if hinge:
            t = torch.clamp(t + margin, min=0)
        den = (mw * mo).sum((-1, -2))
        return ((mw * mo * t).sum((-1, -2)) / torch.max(den, torch.ones_like(den))).sum()
    ref = line(fd[2], fd[1], fd[0], md[0], ud[1])
    if double:
        ref = ref + line(fd[3], fd[0], fd[1], md[1], ud[0])
    ref.backward()
    fc, mc, uc = [t.cuda() for t in f], [t.cuda() for t in m], [t.cuda() for t in u]
    T1, T2, numden = K.zhang_triplet_fwd(fc[0], fc[1], fc[2], fc[3] if double else None, mc[0], mc[1] if double else None,
                                         margin if hinge else 0.0, hinge, m1=uc[0] if double else None, m2=uc[1])
    eye = torch.eye(3, dtype=torch.float64, device="cuda").reshape(1, 9).expand(B, 9).contiguous()
    loss4 = K.bihome_loss_fwd(numden, eye, eye, 0.0)
    assert abs(loss4[0].item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-6
    out = K.zhang_triplet_bwd(torch.ones(1, device="cuda"), fc[0], fc[1], fc[2], fc[3] if double else None, mc[0], mc[1] if double else None,
                              T1, T2, numden, hinge, m1=uc[0] if double else None, m2=uc[1], mask_grads=True)
    g_f1, g_f2, g_f1w, g_f2w, g_m1w, g_m2w, g_m1, g_m2 = out
    assert relerr(g_f1.cpu(), fd[0].grad) < 1e-5 and relerr(g_f2.cpu(), fd[1].grad) < 1e-5
    assert relerr(g_f1w.cpu(), fd[2].grad) < 1e-5 and relerr(g_m1w.cpu(), md[0].grad) < 1e-5
    assert relerr(g_m2.cpu(), ud[1].grad) < 1e-5
    if double:
        assert relerr(g_f2w.cpu(), fd[3].grad) < 1e-5 and relerr(g_m2w.cpu(), md[1].grad) < 1e-5
        assert relerr(g_m1.cpu(), ud[0].grad) < 1e-5
    else:
        assert float(g_m1.abs().max()) == 0.0


@pytest.mark.parametrize("strength", [-1.0, 0.5, 0.9, 1.5])
def test_mask_gate_kernels(strength):
    """bh_mask_fwd / bh_mask_bwd (Sigmoid + per-sample max normalisation + G = mask * features, ContentAware.py:24-35,128-134) against
    torch float64 autograd of the reference's expressions, with a gradient on the mask (from the head) and on G (from the resnet)."""
    from bihome_amd import kernels as K
    N, h = 5, 32
    g = torch.Generator().manual_seed(int(strength * 10) + 77)
    y, f = torch.randn(N, 1, h, h, generator=g) * 2.0, torch.randn(N, 1, h, h, generator=g)
    gm, gg = torch.randn(N, 1, h, h, generator=g), torch.randn(N, 1, h, h, generator=g)
    yd, fd = y.double().requires_grad_(True), f.double().requires_grad_(True)
    m = torch.sigmoid(yd)
    if strength > 0:
        mx = m.reshape(N, -1).max(1)[0].reshape(N, 1, 1, 1)
        m = torch.clamp(m / (mx * strength), 0, 1)
    G = m * fd
    ((m * gm.double()).sum() + (G * gg.double()).sum()).backward()
    mk, Gk, smax, imax = K.mask_fwd(y.cuda(), f.cuda(), strength)
    assert relerr(mk.cpu(), m.detach()) < 1e-6 and relerr(Gk.cpu(), G.detach()) < 1e-6
    assert torch.equal(imax.cpu().long(), torch.sigmoid(y).reshape(N, -1).argmax(1)) or strength <= 0
    g_y, g_f = K.mask_bwd(y.cuda(), f.cuda(), mk, smax, imax, gm.cuda(), gg.cuda(), strength)
    assert relerr(g_f.cpu(), fd.grad) < 1e-6
    assert relerr(g_y.cpu(), yd.grad) < 2e-5, relerr(g_y.cpu(), yd.grad)
    # the mask alone (MaskPredictor.forward), and a consumer that sends no mask gradient (the biHomE head on this backbone)
    m_only, none, _, _ = K.mask_fwd(y.cuda(), None, strength)
    assert none is None and torch.equal(m_only, mk)
    yd2, fd2 = y.double().requires_grad_(True), f.double().requires_grad_(True)
    m2 = torch.sigmoid(yd2)
    if strength > 0:
        m2 = torch.clamp(m2 / (m2.reshape(N, -1).max(1)[0].reshape(N, 1, 1, 1) * strength), 0, 1)
    ((m2 * fd2) * gg.double()).sum().backward()
    g_y2, g_f2 = K.mask_bwd(y.cuda(), f.cuda(), mk, smax, imax, None, gg.cuda(), strength)
    assert relerr(g_y2.cpu(), yd2.grad) < 2e-5 and relerr(g_f2.cpu(), fd2.grad) < 1e-6


@pytest.mark.parametrize("det", [False, True])
def test_warp_adjoint_with_respect_to_the_image(det):
    """bh_warp_bwd_img_f is the transpose of bh_warp_fwd's bilinear gather: <warp(img), g> == <img, warp^T(g)> for random images and
    gradients under homographies that push part of the patch out of the image (zero padding), and equal to torch float64 autograd of
    grid_sample-style sampling through the forward kernel's linearity (finite differences are exact for a linear map)."""
    from bihome_amd import kernels as K
    B, C, h = 3, 2, 64
    g = torch.Generator().manual_seed(5)
    delta = (torch.rand(B, 4, 2, generator=g) - 0.5) * 40.0
    H64, _ = K.h4pt_fwd(delta.cuda().contiguous(), h)
    img = torch.randn(B, C, h, h, generator=g).cuda()
    gout = torch.randn(B, C, h, h, generator=g).cuda()
    with K.det_scope(det):
        warped, _ = K.warp_fwd(img, H64, 1, want_cov=False)
        gimg = K.warp_bwd_img(H64, gout)
        gimg2 = K.warp_bwd_img(H64, gout)
    lhs, rhs = (warped.double() * gout.double()).sum().item(), (img.double() * gimg.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)
    # column by column: the adjoint applied to a one-hot gradient is the row of weights the forward used for that output pixel
    e = torch.zeros_like(gout); e[1, 0, 20, 33] = 1.0
    row = K.warp_bwd_img(H64, e)
    probe = torch.zeros_like(img)
    nz = row[1, 0].nonzero()
    assert 1 <= len(nz) <= 4 and abs(row.sum().item() - 1.0) < 1e-5            # (inside the image: four bilinear weights that sum to 1)
    for (yy, xx) in nz.tolist():
        probe.zero_(); probe[1, 0, yy, xx] = 1.0
        w_fwd, _ = K.warp_fwd(probe, H64, 1, want_cov=False)
        assert abs(w_fwd[1, 0, 20, 33].item() - row[1, 0, yy, xx].item()) < 1e-6
    if det:
        assert torch.equal(gimg, gimg2)                                        # integer-limb accumulation: order-independent


class _Rec:
    def __init__(self):
        self.scalars = {}

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.scalars["tb/%s/%s" % (tag, k)] = float(v)


def _zhang_cfg(fixture):
    import copy
    cfg = copy.deepcopy(configs.get("zhang-orig"))
    if fixture != "zhang_orig":
        cfg["MODEL"]["BACKBONE"]["FIX_MASK"] = False                           # round 4: the mask predictor runs and is trained
    if fixture == "zhang_masknorm":
        cfg["MODEL"]["BACKBONE"]["MASK_NORMALIZATION_STRENGTH"] = 0.5
    return cfg


@pytest.mark.parametrize("fixture", ["zhang_orig", "zhang_mask", "zhang_masknorm"])
def test_zhang_orig_two_steps_vs_reference_fixture(golden, fixture):
    """config/s-coco/zhang-orig (ContentAware + TripletHead, DoubleLine, margin 1.0, channel-agnostic) at B = 4 through two Adam steps
    (lr 1e-2) and an eval forward against the reference fixture: first step tight (north_star tolerances: loss 1e-4, MACE 1e-3),
    every TensorBoard scalar the head writes, the feature maps, the feature extractor's gradients (four calls per step) and its
    running statistics; the second step within a multiple of the reference's own float32-vs-float64 spread.
    zhang_mask / zhang_masknorm (round 4): the same config with FIX_MASK False (plain, and with MASK_NORMALIZATION_STRENGTH 0.5) - fixtures
    made by the reference's own ContentAware.py / TripletHead.py (oracle/make_golden.py --round4): the predicted masks, the mask
    predictor's gradients (they arrive through the loss weights, through the warp of the masks and through G = mask * features) and its
    running statistics are pinned as well."""
    from bihome_amd.step import build_model, build_optimizer, mace, predict, train_step
    g32, g64 = golden(fixture + "_b4_f32"), golden(fixture + "_b4_f64")
    cfg = _zhang_cfg(fixture)
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    d = synth.make_pairs(4, seed=41)
    losses, maces = [], []
    for it in range(2):
        data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
        if it == 0:
            rec = _Rec()
            data["summary_writer"], data["summary_writer_step"] = rec, 0
            model.train()
            opt.zero_grad()
            loss, dgt, dh = model(data)
            loss.backward()
            assert relerr(data["delta_hat_12"].detach().cpu(), g64["delta_hat_12"]) < 1e-4
            assert relerr(data["delta_hat_21"].detach().cpu(), g64["delta_hat_21"]) < 1e-4
            assert relerr(data["feature_1"].detach().cpu()[..., ::8, ::8], g64["feature_1_sub"]) < 1e-4
            if "mask_1_sub" in g64:
                assert relerr(data["mask_1"].detach().cpu()[..., ::8, ::8], g64["mask_1_sub"]) < 1e-4
                m2 = data["mask_2"].detach().double().cpu()
                assert relerr(torch.stack([m2.sum(), m2.abs().sum(), (m2 * m2).sum()]), g64["mask_2_csum"]) < 1e-5
            for k in (k for k in g64 if k.startswith("tb/")):
                assert abs(rec.scalars[k] - g64[k]) <= max(3 * abs(g32[k] - g64[k]), 2e-4 * abs(g64[k])), (k, rec.scalars[k], g64[k])
            params = dict(model[0].named_parameters())
            for k in (k for k in g64 if k.startswith("gradnorm/")):
                got = params[k[9:]].grad.double().norm().item()
                assert abs(got - g64[k]) <= max(5 * abs(g32[k] - g64[k]), 2e-3 * g64[k]), (k, got, g64[k], g32[k])
            for k in (k for k in g64 if k.startswith("grad/")):
                assert relerr(params[k[5:]].grad.cpu(), g64[k]) < max(5 * relerr(g32[k], g64[k]), 2e-3), k
            opt.step(); sched.step()
            loss, dh = loss.detach(), dh.detach()
        else:
            loss, dgt, dh = train_step(model, data, opt, sched, loss_fn="TripletLoss")
        losses.append(loss.item()); maces.append(mace(dgt, dh))
    print(fixture, "loss", losses, "ref", g64["loss"], g32["loss"], "mace", maces, g64["mace"])
    assert abs(losses[0] - g64["loss"][0]) <= 1e-4 * abs(g64["loss"][0])
    assert abs(maces[0] - g64["mace"][0]) < 1e-3
    sp_l, sp_m = abs(g32["loss"][1] - g64["loss"][1]), abs(g32["mace"][1] - g64["mace"][1])
    assert abs(losses[1] - g64["loss"][1]) <= max(5 * sp_l, 2e-2 * abs(g64["loss"][1])), (losses, g64["loss"], g32["loss"])
    assert abs(maces[1] - g64["mace"][1]) <= max(5 * sp_m, 5e-2), (maces, g64["mace"], g32["mace"])
    sd = model[0].state_dict()
    for k in (k for k in g64 if k.startswith("state/")):
        # (after the second step - lr 1e-2 from random weights: the step-1 loss already differs by 0.4 % between this path, the
        #  reference's float32 and its float64 run - the statistics are held to 3 %; the oracle test pins them to 1e-7)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k[6:]]) == int(g64[k]), k        # (the mask predictor: two calls per step -> 4; the extractor: four -> 8)
            continue
        assert relerr(sd[k[6:]].double().cpu(), g64[k]) < max(10 * relerr(g32[k], g64[k]), 3e-2), k
    assert int(sd["feature_extractor.layer3.1.num_batches_tracked"]) == int(g64["state/feature_extractor.layer3.1.num_batches_tracked"]) == 8
    ev = predict(model, {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")})
    assert relerr(ev.cpu(), g64["eval_delta_hat"]) < max(10 * relerr(g32["eval_delta_hat"], g64["eval_delta_hat"]), 2e-2)


def test_zhang_bihome_config_runs_and_matches_oracle():
    """config/s-coco/zhang-bihome (the ContentAware backbone under the biHomE PerceptualHead with directly regressed offsets): one
    training step at B = 4 against the float64 oracle (oracle pinned for both parts separately)."""
    from bihome_amd.step import build_model, mace
    from oracle import bihome_oracle as O
    cfg = configs.get("zhang-bihome")
    d = synth.make_pairs(4, seed=43)
    bb, head = O.build(cfg)
    load_synthetic(bb, 0); load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double(); bb.train(); head.train()
    od = {k: torch.tensor(d[k], dtype=torch.float64) for k in ("patch_1", "patch_2", "delta")}
    oloss, odgt, odh = head(bb(od))
    model = build_model(cfg)
    load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0)
    model.train()
    loss, dgt, dh = model({k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")})
    loss.backward()
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item()), (loss.item(), oloss.item())
    assert abs(mace(dgt, dh) - O.mace(odgt, odh)) < 1e-3


@pytest.mark.parametrize("trained_masks", [False, True])
def test_contentaware_two_rank_data_parallel(tmp_path, trained_masks):
    """Round-3 VERDICT missing #3: data-parallel training of the ContentAware backbone (train.py:513-518 wraps any Model).  Two ranks
    share the one MI355X over gloo; attach_reducer gives the resnet's AND the feature extractor's flat gradient buffer a reducer; after
    the step both ranks hold the SUM of the two shards' gradients in both buffers (= the single-process gradients of the two shards added),
    and rank 1 - initialised with another seed - computed its shard with rank 0's weights."""
    import os, socket, subprocess, sys
    from bihome_amd.ddp import shard_range
    from bihome_amd.step import build_model
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out, B, world = str(tmp_path / "z"), 8, 2
    env = dict(os.environ, BIHOME_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "zhang_ddp_worker.py"), out, str(B)] + (["trained-masks"] if trained_masks else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = [dict(np.load(out + ".rank%d.npz" % k)) for k in range(world)]
    names = ("resnet", "extractor") + (("predictor",) if trained_masks else ())     # (FIX_MASK False: the mask predictor's buffer as well)
    assert int(got[0]["n_reducers"]) == len(names)
    for k in names:
        assert np.array_equal(got[0][k], got[1][k])                     # both replicas hold the same reduced gradient
    cfg = _zhang_cfg("zhang_mask" if trained_masks else "zhang_orig")
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    model.train()
    d = synth.make_pairs(B, seed=78)
    tot = {k: None for k in names}
    for rank in range(world):
        lo, hi = shard_range(B, rank, world)
        data = {k: torch.tensor(d[k][lo:hi]).cuda() for k in ("patch_1", "patch_2", "delta")}
        for p in model.parameters():
            p.grad = None
        loss = model(data)[0]
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - float(got[rank]["loss"])) <= 1e-4 * abs(loss.item()) + 1e-5, (rank, loss.item(), got[rank]["loss"])
        for k, fl in (("resnet", model[0]._runner.flat.flat), ("extractor", model[0].feature_extractor._runner.flat.flat)) + \
                ((("predictor", model[0].mask_predictor._runner.flat.flat),) if trained_masks else ()):
            v = fl.detach().cpu().numpy().astype(np.float64)
            tot[k] = v if tot[k] is None else tot[k] + v
    for k in tot:
        rel = np.sqrt(((got[0][k] - tot[k]) ** 2).sum()) / np.sqrt((tot[k] ** 2).sum())
        assert rel < 1e-4, (k, rel)
