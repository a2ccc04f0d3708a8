"""SURVEY.md 8(f4) branches of the head on the HIP path against fixtures from the reference's own PerceptualHead.py:
deeper extractor outputs (AUXILIARY_RESNET_OUTPUT_LAYER 2-4), the multihead feature loss (TRIPLET_LOSS ''), the
TensorBoard side channel, and configs[3] (ResNet-34 regressor) through three Adam steps in float32 and in the bf16 mode."""
import numpy as np
import pytest
import torch

from bihome_amd import configs, synth
from bihome_amd.weights import load_synthetic

pytestmark = pytest.mark.gpu


def cuda(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dtype).cuda()


def relerr(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)


class _Rec:
    def __init__(self):
        self.scalars = {}

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.scalars["tb/%s/%s" % (tag, k)] = float(v)


def _model(cfg):
    from bihome_amd.step import build_model
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    load_synthetic(model[1].auxiliary_resnet, 0)
    return model


@pytest.mark.parametrize("layer", [1, 2, 3, 4])
def test_extractor_output_layers_vs_golden(golden, layer):
    """Loss, delta_hat, MACE, gradient norms and every TensorBoard scalar of the reference head for extractor outputs
    layer1..layer4 (feature strides 4 / 8 / 16 / 32 = mask pooling factors; 64 / 128 / 256 / 512 channels)."""
    from bihome_amd.step import mace
    name = "zeng_tb_b4" if layer == 1 else "zeng_aux%d_b4" % layer
    g32, g64 = golden(name + "_f32"), golden(name + "_f64")
    cfg = configs.get("zeng-bihome")
    cfg["MODEL"]["HEAD"]["AUXILIARY_RESNET_OUTPUT_LAYER"] = layer
    model = _model(cfg)
    assert set(k for k in model[1].auxiliary_resnet.state_dict() if "layer" in k) == \
        set(k for k in model[1].auxiliary_resnet.state_dict() if any("layer%d." % i in k for i in range(1, layer + 1)))
    model.train()
    d = synth.make_pairs(4, seed=17)
    data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
    data["choice_12"], data["choice_21"] = cuda(g64["choice_12"][0], torch.int64), cuda(g64["choice_21"][0], torch.int64)
    rec = _Rec()
    data["summary_writer"], data["summary_writer_step"] = rec, 1
    loss, dgt, dh = model(data)
    loss.backward()
    spread = abs(g32["loss"][0] - g64["loss"][0])
    assert abs(loss.item() - g64["loss"][0]) <= max(3 * spread, 2e-4 * abs(g64["loss"][0])), (loss.item(), g64["loss"][0], g32["loss"][0])
    assert relerr(dh.detach().cpu(), g64["delta_hat_12"]) < 1e-3
    assert abs(mace(dgt, dh) - g64["mace"][0]) < 1e-3
    params = dict(model[0].named_parameters())
    for n in ("layer1.0.weight", "layer4.6.upper_branch.0.weight", "layer8.3.weight", "layer8.3.bias"):
        gn, ref, sp = params[n].grad.double().norm().item(), g64["gradnorm/" + n], abs(g64["gradnorm/" + n] - g32["gradnorm/" + n])
        assert abs(gn - ref) <= max(5 * sp, 5e-3 * ref), (n, gn, ref, sp)
    tb = {k for k in g64 if k.startswith("tb/")}
    assert set(rec.scalars) == tb and len(tb) == 8
    for k in tb:
        tol = max(3 * abs(g32[k] - g64[k]), 2e-4 * abs(g64[k]))
        assert abs(rec.scalars[k] - g64[k]) <= tol, (k, rec.scalars[k], g64[k], g32[k])
    np.testing.assert_allclose(model[1].auxiliary_resnet.resnet.bn1.running_mean.cpu().numpy(), g64["aux_bn1_running_mean"],
                               rtol=1e-4, atol=1e-5)


def test_multihead_feature_loss_vs_golden(golden):
    """TRIPLET_LOSS '' (multihead_resnet_loss, PerceptualHead.py:245-315) with the driver's L1Loss (train.py:318-322): the
    returned feature tensors (NCHW as upstream), two Adam steps, TensorBoard scalars."""
    from bihome_amd.step import build_loss, build_optimizer, mace, train_step
    g32, g64 = golden("zeng_multihead_b4_f32"), golden("zeng_multihead_b4_f64")
    cfg = configs.get("zeng-multihead")
    model = _model(cfg)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    loss_fn = build_loss(cfg["SOLVER"])
    assert isinstance(loss_fn, torch.nn.L1Loss)
    d = synth.make_pairs(4, seed=17)
    losses, maces = [], []
    for it in range(2):
        data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"] = cuda(g64["choice_12"][it], torch.int64)
        rec = _Rec()
        if it == 0:
            data["summary_writer"], data["summary_writer_step"] = rec, 1
            model.train()
            gt, out, dgt, dh = model(data)
            assert gt.shape == out.shape == (4, 64, 32, 32)
            o = out.detach().double().cpu()
            cs = np.array([o.sum().item(), o.abs().sum().item(), (o * o).sum().item()])
            np.testing.assert_allclose(cs, g64["network_output_csum"], rtol=2e-4)
            assert relerr(o[:, ::8, ::2, ::2], g64["network_output_sub"]) < 1e-3
            for k in (k for k in g64 if k.startswith("tb/")):
                assert abs(rec.scalars[k] - g64[k]) <= max(3 * abs(g32[k] - g64[k]), 2e-4 * abs(g64[k])), k
        loss, dgt, dh = train_step(model, data, opt, sched, loss_fn=loss_fn)
        losses.append(loss.item()); maces.append(mace(dgt, dh))
    assert abs(losses[0] - g64["loss"][0]) <= 1e-4 * abs(g64["loss"][0]), (losses, g64["loss"])
    assert abs(maces[0] - g64["mace"][0]) < 1e-3
    assert abs(losses[1] - g64["loss"][1]) <= max(20 * abs(g32["loss"][1] - g64["loss"][1]), 2e-3 * abs(g64["loss"][1]))


@pytest.mark.parametrize("precision,head_precision", [("f32", "f32"), ("f32x2", "f32x2"), ("bf16", "bf16"), ("bf16", "f32")])
def test_detone_three_steps_vs_golden(golden, precision, head_precision):
    """configs[3] (ResNet-34 regressor + biHomE, B = 8, three Adam steps, lr 5e-3) against the reference fixture.
    float32: first step tight, later steps within a multiple of the reference's own float32-vs-float64 spread (loss 86.0
    vs 81.8 at step 2: training from random weights at this learning rate amplifies rounding).
    bf16 (bf16 MFMA operands, float32 accumulate; backbone only, or backbone and extractor): delta_hat is a direct
    network output, so 2^-9 operand rounding through 36 conv layers shows up directly in MACE (measured 0.0105 px at
    step 0), and the loss is a difference of two nearly equal feature distances (|f1w-f2| - |f1-f2|), which amplifies
    the 0.01 px error of delta_hat (measured 3.4 % at step 0 with a bf16 extractor, 3.1 % with a float32 extractor: the
    backbone's operand rounding dominates).  Stated tolerances: step 0 loss within 5 %, MACE within 0.02 px; later steps loss within 25 %, MACE within the larger of
    15x the float32 reference's own f32-vs-f64 spread and 0.3 px.  north_star's "MACE within 1e-3" is NOT met in bf16 - DESIGN.md 8."""
    from bihome_amd.step import build_optimizer, mace, train_step
    g32, g64 = golden("detone_b8_f32"), golden("detone_b8_f64")
    cfg = configs.get("detone-bihome")
    cfg["MODEL"]["BACKBONE"]["PRECISION"] = precision
    cfg["MODEL"]["HEAD"]["PRECISION"] = head_precision
    model = _model(cfg)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    d = synth.make_pairs(8, seed=5)
    losses, maces = [], []
    for it in range(3):
        data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
        loss, dgt, dh = train_step(model, data, opt, sched)
        losses.append(loss.item()); maces.append(mace(dgt, dh))
        if it == 0 and precision == "f32":
            assert relerr(dh.cpu(), g64["delta_hat_12"][0]) < 1e-3
    print(precision, head_precision, "loss", losses, "mace", maces, "ref", g64["loss"], g64["mace"])
    print("  step-0 margins: loss rel %.2e (f32 reference: %.2e), MACE diff %.2e"
          % (abs(losses[0] - g64["loss"][0]) / abs(g64["loss"][0]), abs(g32["loss"][0] - g64["loss"][0]) / abs(g64["loss"][0]),
             abs(maces[0] - g64["mace"][0])))
    if precision in ("f32", "f32x2"):      # 'f32x2' (two rounded bf16 pieces, three products) is held to the float32 tolerances
        assert abs(losses[0] - g64["loss"][0]) <= max(3 * abs(g32["loss"][0] - g64["loss"][0]), 1e-4 * abs(g64["loss"][0]))
        assert abs(maces[0] - g64["mace"][0]) < 1e-3
        # (mfloor, round 6: profiles/r06p_detone_perturb.txt - inputs x (1 + k 2^-22), k = 0..8, move the SECOND step's MACE of one and the same
        #  arithmetic over -0.020 ... +0.065 px (fp32-input MFMA) / -0.009 ... +0.068 px (fp16 pieces), the third step's by +-0.5 px: which side
        #  of the earlier 0.05 a build lands on is rounding (this build +0.067 with the stems in fp16 pieces, +0.019 ... +0.065 without);
        #  DESIGN.md 0.1 finding 3 has the mechanism.  0.15 px = ~5 sigma of that spread.)
        mult, lrel, mfloor = 5.0, 0.05, 0.15
    else:
        assert abs(losses[0] - g64["loss"][0]) <= 5e-2 * abs(g64["loss"][0])
        assert abs(maces[0] - g64["mace"][0]) < 2e-2
        mult, lrel, mfloor = 15.0, 0.25, 0.3       # (run-to-run: 26.60-26.63 / 22.9-23.5 at steps 1 / 2: fp32 atomics order + bf16)
    for it in (1, 2):
        sp_l, sp_m = abs(g32["loss"][it] - g64["loss"][it]), abs(g32["mace"][it] - g64["mace"][it])
        assert abs(losses[it] - g64["loss"][it]) <= max(mult * sp_l, lrel * abs(g64["loss"][it])), (it, losses, g64["loss"])
        assert abs(maces[it] - g64["mace"][it]) <= max(mult * sp_m, mfloor), (it, maces, g64["mace"])


@pytest.mark.parametrize("precision", ["f32", "f32x3"])
def test_pds_coco_three_steps_vs_golden(golden, precision):
    """(precision 'f32x3': the bit-exact arithmetic of rounds 2-3 runs the same trajectory - round-4 ADVICE asked for the earlier band on it;
    see the note at the last assertion for why both variants share one band.)
    BASELINE.json configs[2] (pds-coco: both images of a pair independently photometrically distorted,
    config/pds-coco/zeng-bihome-lr-1e-3.yaml:62-67) at B = 8 through three Adam steps against the reference's own modules on
    the same distorted inputs and DSAC draws: first step tight (north_star tolerances), later steps within a multiple of
    the reference's own float32-vs-float64 spread."""
    from bihome_amd.step import build_optimizer, mace, train_step
    g32, g64 = golden("zeng_pds_b8_f32"), golden("zeng_pds_b8_f64")
    cfg = configs.get("zeng-bihome-pds")
    assert cfg["DATA"]["PHOTOMETRIC_MAX_DELTA"] == 32
    cfg["MODEL"]["BACKBONE"]["PRECISION"] = precision
    cfg["MODEL"]["HEAD"]["PRECISION"] = precision
    model = _model(cfg)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    d = synth.make_pairs(8, seed=8, photometric_max_delta=32)
    losses, maces = [], []
    for it in range(3):
        data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"], data["choice_21"] = cuda(g64["choice_12"][it], torch.int64), cuda(g64["choice_21"][it], torch.int64)
        loss, dgt, dh = train_step(model, data, opt, sched)
        losses.append(loss.item()); maces.append(mace(dgt, dh))
        if it == 0:
            assert relerr(data["pf_hat_12"].detach().cpu()[..., ::8, ::8], g64["pf_hat_12_sub"]) < 2e-4
            assert relerr(dh.cpu(), g64["delta_hat_12"]) < 1e-3
    print("pds", precision, "loss", losses, "mace", maces, "ref", g64["loss"], g64["mace"], g32["loss"],
          "mace diffs", [abs(maces[i] - g64["mace"][i]) for i in range(3)])
    # the loss is a difference of feature distances and sits near zero here (0.354 against ~55 per term): absolute floor
    assert abs(losses[0] - g64["loss"][0]) <= max(3 * abs(g32["loss"][0] - g64["loss"][0]), 1e-4 * abs(g64["loss"][0]))
    assert abs(maces[0] - g64["mace"][0]) < 1e-3
    for it in (1, 2):
        sp_l, sp_m = abs(g32["loss"][it] - g64["loss"][it]), abs(g32["mace"][it] - g64["mace"][it])
        assert abs(losses[it] - g64["loss"][it]) <= max(5 * sp_l, 0.05 * abs(g64["loss"][it])), (it, losses, g64["loss"])
        # Training from random weights at B = 8 amplifies rounding-level differences ~20x per Adam step, and round 6 measured how far (round-5
        # ADVICE asked for the cause of f32x3's 0.086 px): (1) tools/grad_arith_diff.py - the first step's gradient of ANY two arithmetics
        # differs by 1-2 % in EVERY tensor incl. the last layer's bias, for forward passes that agree to 1e-6: the bilinear warp's derivative
        # jumps where a sample coordinate crosses an integer, and dL/dH is a near-cancelling sum over 16 k pixels - one pixel on the other
        # side of a kink moves it by ~1 % (the float32 reference sits the same 0.5-1 % from its float64 self, tests/test_fullsize_gpu.py);
        # (2) profiles/r06f_pds_perturb.txt - multiplying the INPUTS by (1 + k 2^-22), k = 0..8, moves the third step's MACE of one and the
        # same arithmetic over -0.062 ... +0.027 px (fp32-input MFMA) and -0.069 ... +0.037 px (fp16 pieces): standard deviation 0.03 px for
        # both, no arithmetic stands out; (3) profiles/r06e_pds_variants.txt - the kernel-fusion switches (which do not touch the forward
        # pass) move it by < 0.012 px.  The third step is therefore held at 0.15 px (5 sigma of that spread; f32x3 lands at -0.104 in this
        # build, -0.049 in round 3's), the second at 0.03; the first step - before any update - stays at the north_star tolerances above.
        assert abs(maces[it] - g64["mace"][it]) <= max(10 * sp_m, 0.03 if it == 1 else 0.15), (it, maces, g64["mace"])


@pytest.mark.parametrize("base,loss_name", [("zeng-ihome", None), ("zeng-multihead", "L1Loss")])
def test_score_weighted_multi_hypothesis_training_vs_golden(golden, base, loss_name):
    """Training with RANSAC_HYPOTHESIS_NO = 4 (16 points per hypothesis): softmax(-reprojection error) scores weight the
    hinge loss (one-line, PerceptualHead.py:505-511) or both feature maps (multihead, :276-280), delta_hat is the
    score-weighted mean (:309-312,:708-710), and the gradient reaches the perspective field through the DLT of every
    hypothesis AND through the scores (bh_dsac_scores_bwd: every point of the field).  Against the reference's own
    modules on the recorded draws: step 0 tight, step 1 (after one Adam update - it sees the gradients) within the
    reference's float32-vs-float64 spread."""
    from bihome_amd.step import build_loss, build_optimizer, mace, train_step
    name = base.replace("-", "_") + "_n4_b4"
    g32, g64 = golden(name + "_f32"), golden(name + "_f64")
    cfg = configs.get(base)
    cfg["MODEL"]["HEAD"].update(RANSAC_HYPOTHESIS_NO=4, POINTS_PER_HYPOTHESIS=16)
    model = _model(cfg)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    loss_fn = build_loss(cfg["SOLVER"])
    assert isinstance(loss_fn, torch.nn.Module) == (loss_name is not None)
    d = synth.make_pairs(4, seed=19)
    losses, maces = [], []
    for it in range(2):
        data = {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"] = cuda(g64["choice_12"][it], torch.int64)
        if it == 0:                                              # step 0 by hand: the gradients are checked before Adam consumes them
            model.train()
            opt.zero_grad()
            out = model(data)
            loss = loss_fn(out[0], out[1]) if loss_name else out[0]
            dgt, dh = out[-2], out[-1]
            loss.backward()
            params = dict(model[0].named_parameters())
            for k in ("layer1.0.weight", "layer4.6.upper_branch.0.weight", "layer8.3.weight", "layer8.3.bias"):
                gn, ref, sp = params[k].grad.double().norm().item(), g64["gradnorm/" + k], abs(g64["gradnorm/" + k] - g32["gradnorm/" + k])
                assert abs(gn - ref) <= max(5 * sp, 5e-3 * ref), (k, gn, ref, sp)
            opt.step(); sched.step()
            loss, dh = loss.detach(), dh.detach()
            assert dh.shape == (4, 4, 2)
            assert relerr(dh.cpu(), g64["delta_hat_12"]) < 2e-3
        else:
            loss, dgt, dh = train_step(model, data, opt, sched, loss_fn=loss_fn)
        losses.append(loss.item()); maces.append(mace(dgt, dh))
    print(base, "loss", losses, "mace", maces, "ref", g64["loss"], g64["mace"], g32["loss"])
    assert abs(losses[0] - g64["loss"][0]) <= max(3 * abs(g32["loss"][0] - g64["loss"][0]), 2e-4 * abs(g64["loss"][0]))
    assert abs(maces[0] - g64["mace"][0]) < 2e-3
    # step 1: the scores are softmax(-error) of errors ~1e5 apart - effectively an arg-max over the hypotheses - so a
    # rounding-level difference in the updated weights can hand a sample to another hypothesis (measured: MACE 24.48 against
    # 24.55 with gradient norms equal to 3 digits at step 0); the band is the size of one such flip.  (Round 4: with the two-branch
    # BatchNorm join - one rounding less in four layers, kernel-level equal to float64 to 2e-6, the B = 64 oracle test unchanged - the
    # one-line config flips a different sample: 6.96 against 7.91, 12 %; with BIHOME_BN_JOIN=0 it is 7.9 again.  Band 15 %; the step-0
    # assertions above are the tight ones.)
    assert abs(losses[1] - g64["loss"][1]) <= 0.15 * abs(g64["loss"][1]), (losses, g64["loss"], g32["loss"])
    assert abs(maces[1] - g64["mace"][1]) <= 0.15, (maces, g64["mace"], g32["mace"])
