"""The CPU oracle (oracle/bihome_oracle.py) against the golden vectors produced by the reference's
own files (oracle/make_golden.py).  Same torch ops on the same box class => tight tolerances."""
import numpy as np
import pytest
import torch

from bihome_amd import configs, synth
from bihome_amd.weights import load_synthetic
from oracle import bihome_oracle as O


def _t(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


@pytest.mark.parametrize("tag,dtype,rtol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-9)])
def test_head_scenario(golden, tag, dtype, rtol):
    g = golden("head_b8_" + tag)
    cfg = configs.get("zeng-bihome")
    head = O.BiHomEHead(torch.nn.Identity(), **cfg["MODEL"]["HEAD"])
    load_synthetic(head.auxiliary_resnet, 0)
    head.to(dtype).train()
    d = synth.make_head_inputs(8, 7)
    data = {k: _t(d[k], dtype) for k in ("patch_1", "patch_2", "delta", "pf_hat_12", "pf_hat_21")}
    data["pf_hat_12"].requires_grad_(True)
    data["pf_hat_21"].requires_grad_(True)
    loss, dgt, dh = head(data, _t(g["choice_12"], torch.int64), _t(g["choice_21"], torch.int64))
    loss.backward()
    L = head.last
    assert abs(loss.item() - g["loss"]) <= rtol * abs(g["loss"])
    np.testing.assert_allclose(L["H_dlt_12"].detach().squeeze(1).numpy(), g["H_dlt_12"], rtol=50 * rtol, atol=50 * rtol)
    np.testing.assert_allclose(L["delta_hat_12"].detach().numpy(), g["delta_hat_12"], atol=2000 * rtol)
    np.testing.assert_allclose(L["delta_hat_21"].detach().numpy(), g["delta_hat_21"], atol=2000 * rtol)
    np.testing.assert_allclose(L["H_4pt_12"].detach().numpy(), g["H_4pt_12"], rtol=50 * rtol, atol=50 * rtol)
    np.testing.assert_allclose(L["warp_12"].detach().numpy()[..., ::4, ::4], g["warp_sub_12"], atol=500 * rtol)
    np.testing.assert_allclose(L["mask_pooled_21"].detach().numpy(), g["mask_pooled_21"], atol=100 * rtol)
    np.testing.assert_allclose(data["pf_hat_12"].grad.numpy()[:, :, ::4, ::4], g["grad_pf_hat_12_sub"],
                               rtol=2000 * rtol, atol=2000 * rtol * np.abs(g["grad_pf_hat_12_sub"]).max())
    np.testing.assert_allclose(O.mace(dgt, dh), g["mace"], rtol=100 * rtol)
    np.testing.assert_allclose(head.auxiliary_resnet.resnet.bn1.running_mean.numpy(), g["aux_bn1_running_mean"],
                               rtol=100 * rtol, atol=100 * rtol)


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_dsac_n4_argmin_bit_exact(golden, tag, dtype):
    g = golden("dsac_n4_" + tag)
    cfg = configs.get("zeng-bihome")
    cfg["MODEL"]["HEAD"].update(RANSAC_HYPOTHESIS_NO=4, POINTS_PER_HYPOTHESIS=16)
    head = O.BiHomEHead(torch.nn.Identity(), **cfg["MODEL"]["HEAD"]).to(dtype).eval()
    d = synth.make_head_inputs(8, 11, noise=2.0)
    with torch.no_grad():
        dh, _ = head.predict_homography({"pf_hat_12": _t(d["pf_hat_12"], dtype)}, _t(g["choice"], torch.int64))
    assert np.array_equal(head.last["best"].numpy(), g["best"])          # integer output: bit-exact
    np.testing.assert_allclose(dh.numpy(), g["delta_hat"], atol=5e-2 if tag == "f32" else 1e-7)


@pytest.mark.parametrize("tag,dtype,rtol", [("f32", torch.float32, 1e-4), ("f64", torch.float64, 1e-8)])
def test_zeng_first_step(golden, tag, dtype, rtol):
    g = golden("zeng_b8_" + tag)
    cfg = configs.get("zeng-bihome")
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.to(dtype); head.to(dtype)
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(8, seed=42)
    torch.set_num_threads(8)
    data = {k: _t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
    loss, dgt, dh = O.train_step(bb, head, opt, sched, data, _t(g["choice_12"][0], torch.int64),
                                 _t(g["choice_21"][0], torch.int64))
    assert abs(loss.item() - g["loss"][0]) <= rtol * abs(g["loss"][0])
    np.testing.assert_allclose(data["pf_hat_12"].detach().numpy()[..., ::8, ::8], g["pf_hat_12_sub"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][0], rtol=rtol)
    gn = dict(bb.named_parameters())["layer1.0.weight"].grad.double().norm().item()
    np.testing.assert_allclose(gn, g["gradnorm/layer1.0.weight"], rtol=20 * rtol)
    np.testing.assert_allclose(bb.layer1[1].running_mean.numpy(), g["bn_layer1_running_mean"], rtol=10 * rtol, atol=10 * rtol)


def test_detone_first_step(golden):
    g = golden("detone_b4_f32")
    cfg = configs.get("detone-bihome")
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.train(); head.train()
    d = synth.make_pairs(4, seed=5)
    data = {k: _t(d[k], torch.float32) for k in ("patch_1", "patch_2", "delta")}
    loss, dgt, dh = head(bb(data))
    loss.backward()
    assert abs(loss.item() - g["loss"]) <= 1e-4 * abs(g["loss"])
    np.testing.assert_allclose(dh.detach().numpy(), g["delta_hat_12"], rtol=1e-4, atol=1e-5)
    gn = dict(bb.named_parameters())["resnet34.fc.weight"].grad.double().norm().item()
    np.testing.assert_allclose(gn, g["gradnorm/resnet34.fc.weight"], rtol=1e-3)


@pytest.mark.parametrize("name,batch,seed", [("zeng-orig", 4, 21), ("detone-orig", 4, 22)])
def test_supervised_orig_configs(golden, name, batch, seed):
    """The supervised "-orig" experiments (same backbones, OneLine, NoOpHead + torch loss, train.py:318-322): the oracle's
    first two Adam steps and an eval-mode forward against the reference's own modules."""
    g = golden(name.replace("-", "_") + "_b4_f64")
    cfg = configs.get(name)
    bb, head = O.build(cfg)
    assert isinstance(head, O.NoOpHead)
    load_synthetic(bb, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    loss_fn = getattr(torch.nn, cfg["SOLVER"]["LOSS"])()
    d = synth.make_pairs(batch, seed=seed, target=True)
    key0 = cfg["MODEL"]["BACKBONE"]["TARGET_KEYS"][0]
    for it in range(2):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta", "target")}
        loss, dgt, dh = O.train_step(bb, head, opt, sched, data, loss_fn=loss_fn)
        assert abs(loss.item() - g["loss"][it]) <= 1e-8 * abs(g["loss"][it])
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
        if it == 0:
            out = data[key0].detach()
            np.testing.assert_allclose(out[..., ::8, ::8].numpy() if out.dim() == 4 else out.numpy(), g["output0"], atol=1e-8)
            np.testing.assert_allclose(dh.numpy(), g["delta_hat0"], atol=1e-8)
    bb.eval()
    with torch.no_grad():
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta", "target")}
        bb(data)
    out = data[key0]
    np.testing.assert_allclose(out[..., ::8, ::8].numpy() if out.dim() == 4 else out.numpy(), g["eval_output"], atol=1e-7)


def test_ihome_one_line(golden):
    """iHomE (TRIPLET_LOSS 'one-line', numeric margin; PerceptualHead.py:465-538): the oracle's two Adam steps against the
    reference's own PerceptualHead + Rethinking (OneLine) on recorded DSAC draws."""
    g = golden("zeng_ihome_b4_f64")
    cfg = configs.get("zeng-ihome")
    bb, head = O.build(cfg)
    assert head.one_line
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(4, seed=31)
    for it in range(2):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        loss, dgt, dh = O.train_step(bb, head, opt, sched, data, choice_12=_t(g["choice_12"][it], torch.int64))
        assert abs(loss.item() - g["loss"][it]) <= 1e-8 * abs(g["loss"][it])
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
        if it == 0:
            np.testing.assert_allclose(dh.numpy(), g["delta_hat_12"], atol=1e-7)


class _Rec:
    def __init__(self):
        self.scalars = {}

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.scalars["tb/%s/%s" % (tag, k)] = float(v)


def test_detone_three_steps(golden):
    """configs[3]'s model, three Adam steps at B=8 (fixture: reference ResNet34.py + PerceptualHead.py), float64."""
    g = golden("detone_b8_f64")
    cfg = configs.get("detone-bihome")
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(8, seed=5)
    for it in range(3):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        loss, dgt, dh = O.train_step(bb, head, opt, sched, data)
        assert abs(loss.item() - g["loss"][it]) <= 1e-7 * abs(g["loss"][it]), it
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
        np.testing.assert_allclose(dh.numpy(), g["delta_hat_12"][it], atol=1e-6)


@pytest.mark.parametrize("layer", [2, 3, 4])
def test_extractor_output_layers(golden, layer):
    """AUXILIARY_RESNET_OUTPUT_LAYER 2/3/4 (PerceptualHead.py:24-33,62-67; mask downsample factor 8/16/32 :450): first
    step loss, delta_hat, gradient norms and the head's TensorBoard side channel (:678-697) against the reference."""
    g = golden("zeng_aux%d_b4_f64" % layer)
    cfg = configs.get("zeng-bihome")
    cfg["MODEL"]["HEAD"]["AUXILIARY_RESNET_OUTPUT_LAYER"] = layer
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(4, seed=17)
    data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
    rec = _Rec()
    data["summary_writer"], data["summary_writer_step"] = rec, 1
    loss, dgt, dh = O.train_step(bb, head, opt, sched, data, _t(g["choice_12"][0], torch.int64), _t(g["choice_21"][0], torch.int64))
    assert abs(loss.item() - g["loss"][0]) <= 1e-8 * abs(g["loss"][0])
    np.testing.assert_allclose(dh.numpy(), g["delta_hat_12"], atol=1e-7)
    for k in g:
        if k.startswith("tb/"):
            np.testing.assert_allclose(rec.scalars[k], g[k], rtol=1e-7, err_msg=k)
    assert set(rec.scalars) == {k for k in g if k.startswith("tb/")}


def test_multihead_feature_loss(golden):
    """TRIPLET_LOSS '' -> multihead_resnet_loss (PerceptualHead.py:245-315) + the driver's torch loss (train.py:318-322)."""
    g = golden("zeng_multihead_b4_f64")
    cfg = configs.get("zeng-multihead")
    bb, head = O.build(cfg)
    assert head.multihead
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    loss_fn = getattr(torch.nn, cfg["SOLVER"]["LOSS"])()
    d = synth.make_pairs(4, seed=17)
    for it in range(2):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        rec = _Rec()
        if it == 0:
            data["summary_writer"], data["summary_writer_step"] = rec, 1
        loss, dgt, dh = O.train_step(bb, head, opt, sched, data, _t(g["choice_12"][it], torch.int64), loss_fn=loss_fn)
        assert abs(loss.item() - g["loss"][it]) <= 1e-8 * abs(g["loss"][it])
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
        if it == 0:
            for k in g:
                if k.startswith("tb/"):
                    np.testing.assert_allclose(rec.scalars[k], g[k], rtol=1e-7, err_msg=k)


def test_bihome_tensorboard_keys(golden):
    """The shipped biHomE config with the driver's log-step side channel on: every key / value the reference head writes."""
    g = golden("zeng_tb_b4_f64")
    cfg = configs.get("zeng-bihome")
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(4, seed=17)
    data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
    rec = _Rec()
    data["summary_writer"], data["summary_writer_step"] = rec, 1
    loss, _, _ = O.train_step(bb, head, opt, sched, data, _t(g["choice_12"][0], torch.int64), _t(g["choice_21"][0], torch.int64))
    assert abs(loss.item() - g["loss"][0]) <= 1e-8 * abs(g["loss"][0])
    tb = {k for k in g if k.startswith("tb/")}
    assert tb == set(rec.scalars) and len(tb) == 8
    for k in tb:
        np.testing.assert_allclose(rec.scalars[k], g[k], rtol=1e-7, err_msg=k)


@pytest.mark.parametrize("base,loss_name", [("zeng-ihome", None), ("zeng-multihead", "L1Loss")])
def test_score_weighted_multi_hypothesis_training(golden, base, loss_name):
    """RANSAC_HYPOTHESIS_NO = 4, 16 points each: the hinge loss (one-line) / the feature maps (multihead) are weighted by
    softmax(-reprojection error) and delta_hat is the score-weighted mean (PerceptualHead.py:276-280,505-511,708-710)."""
    g = golden(base.replace("-", "_") + "_n4_b4_f64")
    cfg = configs.get(base)
    cfg["MODEL"]["HEAD"].update(RANSAC_HYPOTHESIS_NO=4, POINTS_PER_HYPOTHESIS=16)
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    load_synthetic(head.auxiliary_resnet, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    loss_fn = getattr(torch.nn, loss_name)() if loss_name else None
    d = synth.make_pairs(4, seed=19)
    for it in range(2):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        loss, dgt, dh = O.train_step(bb, head, opt, sched, data, _t(g["choice_12"][it], torch.int64), loss_fn=loss_fn)
        assert abs(loss.item() - g["loss"][it]) <= 1e-7 * abs(g["loss"][it]), (it, loss.item(), g["loss"])
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
        if it == 0:
            np.testing.assert_allclose(dh.numpy(), g["delta_hat_12"], atol=1e-6)


def _zhang_cfg(fixture):
    """zhang-orig, and its two trained-mask variants (round 4: FIX_MASK False, plain and with MASK_NORMALIZATION_STRENGTH 0.5)."""
    import copy
    cfg = copy.deepcopy(configs.get("zhang-orig"))
    if fixture != "zhang_orig":
        cfg["MODEL"]["BACKBONE"]["FIX_MASK"] = False
    if fixture == "zhang_masknorm":
        cfg["MODEL"]["BACKBONE"]["MASK_NORMALIZATION_STRENGTH"] = 0.5
    return cfg


@pytest.mark.parametrize("fixture", ["zhang_orig", "zhang_mask", "zhang_masknorm"])
def test_zhang_content_aware_triplet_head(golden, fixture):
    """Round 3: the Zhang baseline - oracle ContentAwareBackbone + ZhangTripletHead against the fixture made by the reference's own
    src/backbones/ContentAware.py + src/heads/TripletHead.py (oracle/make_golden.py --round3): two Adam steps at B = 4 in float64 -
    losses, MACE, both regressed offsets, the feature maps, gradients of the feature extractor (which runs four times per step: two
    patches in the backbone, two warped patches in the head), the running statistics after the steps (they pin the order of those
    four calls) and the eval-mode prediction."""
    g = golden(fixture + "_b4_f64")
    cfg = _zhang_cfg(fixture)
    bb, head = O.build(cfg)
    load_synthetic(bb, 0)
    bb.double(); head.double()
    opt, sched = O.make_optimizer(torch.nn.Sequential(bb, head), cfg["SOLVER"])
    d = synth.make_pairs(4, seed=41)
    for it in range(2):
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        bb.train(); head.train()
        opt.zero_grad()
        loss, dgt, dh = head(bb(data))
        loss.backward()
        if it == 0:
            if "mask_1_sub" in g:          # trained masks: the predictor's output (Sigmoid, per-sample max normalisation) and its gradients
                np.testing.assert_allclose(data["mask_1"].detach().numpy()[..., ::8, ::8], g["mask_1_sub"], atol=1e-9)
                m2 = data["mask_2"].detach().numpy().astype(np.float64)
                np.testing.assert_allclose([m2.sum(), np.abs(m2).sum(), (m2 * m2).sum()], g["mask_2_csum"], rtol=1e-9)
            np.testing.assert_allclose(data["delta_hat_12"].detach().numpy(), g["delta_hat_12"], atol=1e-8)
            np.testing.assert_allclose(data["delta_hat_21"].detach().numpy(), g["delta_hat_21"], atol=1e-8)
            np.testing.assert_allclose(data["feature_1"].detach().numpy()[..., ::8, ::8], g["feature_1_sub"], atol=1e-9)
            params = dict(bb.named_parameters())
            for k in (k for k in g if k.startswith("gradnorm/")):
                np.testing.assert_allclose(params[k[9:]].grad.norm().item(), g[k], rtol=1e-6, err_msg=k)
            for k in (k for k in g if k.startswith("grad/")):
                np.testing.assert_allclose(params[k[5:]].grad.numpy(), g[k], rtol=1e-6, atol=1e-10, err_msg=k)
            np.testing.assert_allclose(head.last["ln1"].item(), g["tb/loss_comp/ln1"], rtol=1e-8)
            np.testing.assert_allclose(head.last["ln2"].item(), g["tb/loss_comp/ln2"], rtol=1e-8)
            np.testing.assert_allclose(cfg["MODEL"]["HEAD"]["MU"] * head.last["ln3"].item(), g["tb/loss_comp/ln3"], rtol=1e-8)
        opt.step(); sched.step()
        assert abs(loss.item() - g["loss"][it]) <= 1e-7 * abs(g["loss"][it]), (it, loss.item(), g["loss"][it])
        np.testing.assert_allclose(O.mace(dgt, dh), g["mace"][it], rtol=1e-7)
    sd = bb.state_dict()
    for k in (k for k in g if k.startswith("state/")):
        np.testing.assert_allclose(sd[k[6:]].double().numpy(), g[k], rtol=1e-7, atol=1e-10, err_msg=k)
    bb.eval(); head.eval()
    with torch.no_grad():
        data = {k: _t(d[k], torch.float64) for k in ("patch_1", "patch_2", "delta")}
        dh, H = head.predict_homography(bb.predict_homography(data))
    np.testing.assert_allclose(dh.numpy(), g["eval_delta_hat"], atol=1e-7)
    np.testing.assert_allclose(H.numpy(), g["eval_H"], atol=1e-8)
