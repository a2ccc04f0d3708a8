"""Generate tests/golden/*.npz by running the REFERENCE's own model files in the build container.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (never on the GPU box).  The
reference's src/backbones/{Rethinking,ResNet34,utils}.py, src/heads/{PerceptualHead,ransac_utils}.py
and src/data/utils.py are imported verbatim from /root/reference after registering three stand-ins
for third-party packages that are absent offline (oracle/refshim/: kornia 0.5.0 - four functions
restated from its published algorithm; torchvision.models - resnet18/34 definitions; cv2 - empty,
the torch path never calls it).  All heavy arithmetic runs in real torch CPU ops.  Weights and
inputs are pure functions of seeds (bihome_amd/weights.py, bihome_amd/synth.py) so the tests
regenerate them instead of storing them; only small outputs are stored.

    python oracle/make_golden.py            # writes tests/golden/{zeng_b8,head_b8,dsac_n4,detone_b4}.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BIHOME_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def install_standins():
    from oracle.refshim import kornia_standin, torchvision_standin
    sys.modules["kornia"] = kornia_standin
    tv = types.ModuleType("torchvision")
    tv.models = torchvision_standin
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = torchvision_standin
    from oracle.refshim import cv2_standin
    sys.modules["cv2"] = cv2_standin      # (the torch hot path never calls it; the numpy data-generation path does)
    # the reference package is `src`; the build's drop-in shims are ALSO `src.*`, so make sure the
    # reference's is the one imported in this process
    if REF in sys.path:
        sys.path.remove(REF)
    sys.path.insert(0, REF)


class RecordMultinomial:
    """Wrap torch.multinomial to record the sampled indices (ransac_utils.py:54-57)."""

    def __init__(self):
        self.calls = []
        self._orig = torch.multinomial

    def __enter__(self):
        def wrapped(*a, **k):
            out = self._orig(*a, **k)
            self.calls.append(out.clone())
            return out
        torch.multinomial = wrapped
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._orig


def t(x, dtype=torch.float32):
    return torch.from_numpy(np.asarray(x)).to(dtype)


def sub(a, step=8):
    return a[..., ::step, ::step].detach().double().numpy().copy()


def csum(a):
    a = a.detach().double()
    return np.array([a.sum().item(), a.abs().sum().item(), (a * a).sum().item()])


def run_head_scenario(ref_head_cls, cfg, dtype, batch=8, seed=7):
    """Head only (biHomE double-line) on given pf fields. Returns dict of outputs."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    head = ref_head_cls(torch.nn.Identity(), **cfg["MODEL"]["HEAD"]).to(dtype)
    load_synthetic(head.auxiliary_resnet, seed=0)
    head.to(dtype).train()
    d = synth.make_head_inputs(batch, seed)
    data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta", "pf_hat_12", "pf_hat_21")}
    data["pf_hat_12"].requires_grad_(True)
    data["pf_hat_21"].requires_grad_(True)
    torch.manual_seed(1234)
    with RecordMultinomial() as rec:
        loss, delta_gt, delta_hat = head(data)
    loss.backward()
    out = {"choice_12": rec.calls[0].reshape(batch, -1).numpy(), "choice_21": rec.calls[1].reshape(batch, -1).numpy(),
           "loss": np.float64(loss.item()), "delta_hat_12": delta_hat.detach().double().numpy().copy(),
           "grad_pf12_csum": csum(data["pf_hat_12"].grad), "grad_pf21_csum": csum(data["pf_hat_21"].grad)}
    # sparse gradient: store nonzero entries' positions & values compactly (<= 128 per sample per dir)
    for name in ("pf_hat_12", "pf_hat_21"):
        g = data[name].grad.detach().double().numpy().copy()
        out["grad_" + name + "_sub"] = g[:, :, ::4, ::4]
    # intermediates, recomputed through the reference's own helper methods with the recorded choice
    import kornia
    with torch.no_grad():
        for tag, pfk, ch in (("12", "pf_hat_12", rec.calls[0]), ("21", "pf_hat_21", rec.calls[1])):
            mf, cf, fp = head.forward_map_field(data[pfk], None, None)
            choice = ch.reshape(batch, -1, 1).repeat(1, 1, 2)
            H = kornia.find_homography_dlt(torch.gather(cf, 1, choice), torch.gather(mf, 1, choice))
            dh = kornia.transform_points(H, fp) - fp
            out["H_dlt_" + tag] = H.double().numpy().copy()
            out["delta_hat_" + tag] = dh.double().numpy().copy()
            src = "patch_1" if tag == "12" else "patch_2"
            pw, h4 = head._warp(data[src], delta_hat=dh)
            mw, _ = head._warp(torch.ones_like(data[src]), delta_hat=dh)
            out["H_4pt_" + tag] = h4.double().numpy().copy()
            out["warp_sub_" + tag] = sub(pw, 4)
            out["warp_csum_" + tag] = csum(pw)
            out["mask_pooled_" + tag] = torch.nn.AvgPool2d(4, 4)(mw).squeeze(1).double().numpy().copy()
        # features: extractor is in train mode (batch statistics) exactly as in the training step;
        # calling it here would also move running stats, so use a deep copy
        import copy
        aux = copy.deepcopy(head.auxiliary_resnet)
        out["feat_p1_csum"] = csum(aux(data["patch_1"]))
    mace = np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) - delta_hat.detach().numpy().reshape(-1, 2), axis=-1))
    out["mace"] = np.float64(mace)
    out["aux_bn1_running_mean"] = head.auxiliary_resnet.resnet.bn1.running_mean.double().numpy().copy()
    out["aux_bn1_running_var"] = head.auxiliary_resnet.resnet.bn1.running_var.double().numpy().copy()
    return out


def run_zeng_scenario(ref_bb_cls, ref_head_cls, cfg, dtype, batch=8, seed=42, steps=3):
    """End to end: Rethinking backbone + biHomE head, `steps` Adam steps on one fixed batch."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    load_synthetic(head.auxiliary_resnet, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    s = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=s["LR"], betas=(s["MOMENTUM_1"], s["MOMENTUM_2"]), weight_decay=0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=s["MILESTONES"], gamma=s["LR_DECAY"])
    d = synth.make_pairs(batch, seed=seed)
    out = {"loss": [], "mace": [], "gnorm": [], "choice_12": [], "choice_21": []}
    model.train()                                                       # train.py:296
    for it in range(steps):
        opt.zero_grad()                                                 # train.py:305
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        torch.manual_seed(1000 + it)
        with RecordMultinomial() as rec:
            loss, delta_gt, delta_hat = model(data)                     # train.py:357
        if it == 0:
            data["pf_hat_12"].retain_grad()
        loss.backward()                                                 # train.py:379
        if it == 0:
            out["pf_hat_12_sub"] = sub(data["pf_hat_12"], 8)
            out["pf_hat_21_sub"] = sub(data["pf_hat_21"], 8)
            out["pf_hat_12_csum"] = csum(data["pf_hat_12"])
            out["delta_hat_12"] = delta_hat.detach().double().numpy().copy()
            out["grad_pf12_csum"] = csum(data["pf_hat_12"].grad)
            for name in ("layer1.0.weight", "layer2.0.upper_branch.0.weight", "layer4.6.upper_branch.0.weight",
                         "layer8.0.weight", "layer8.3.weight", "layer8.3.bias", "layer1.1.weight", "layer1.1.bias"):
                p = dict(bb.named_parameters())[name]
                out["gradnorm/" + name] = np.float64(p.grad.double().norm().item())
            out["bn_layer1_running_mean"] = bb.layer1[1].running_mean.double().numpy().copy()
            out["bn_layer1_running_var"] = bb.layer1[1].running_var.double().numpy().copy()
        gn = sum(p.grad.double().norm().item() ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
        opt.step()
        sched.step()                                                    # train.py:386-387
        mace = np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) -
                                      delta_hat.detach().numpy().reshape(-1, 2), axis=-1))   # train.py:402-403
        out["loss"].append(loss.item()); out["mace"].append(mace); out["gnorm"].append(gn)
        out["choice_12"].append(rec.calls[0].reshape(batch, -1).numpy())
        out["choice_21"].append(rec.calls[1].reshape(batch, -1).numpy())
    # eval-mode prediction after the steps (eval.py:109 path: backbone.predict_homography -> head.predict_homography)
    model.eval()
    with torch.no_grad():
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        torch.manual_seed(2000)
        with RecordMultinomial() as rec:
            dh, _ = head.predict_homography(bb.predict_homography(data))
        out["eval_choice"] = rec.calls[0].reshape(batch, -1).numpy()
        out["eval_delta_hat"] = dh.double().numpy().copy()
        out["eval_mace"] = np.float64(np.mean(np.linalg.norm(d["delta"].reshape(-1, 2) - dh.numpy().reshape(-1, 2), axis=-1)))
    for k in ("loss", "mace", "gnorm", "choice_12", "choice_21"):
        out[k] = np.asarray(out[k])
    return out


def run_dsac_n4(ref_head_cls, cfg, dtype, batch=8, seed=11):
    """predict_homography with 4 hypotheses: known-answer for the argmin index (PerceptualHead.py:755-757)."""
    import copy
    c = copy.deepcopy(cfg)
    c["MODEL"]["HEAD"]["RANSAC_HYPOTHESIS_NO"] = 4
    c["MODEL"]["HEAD"]["POINTS_PER_HYPOTHESIS"] = 16
    head = ref_head_cls(torch.nn.Identity(), **c["MODEL"]["HEAD"]).to(dtype).eval()
    from bihome_amd import synth
    d = synth.make_head_inputs(batch, seed, noise=2.0)
    data = {"pf_hat_12": t(d["pf_hat_12"], dtype)}
    torch.manual_seed(99)
    with RecordMultinomial() as rec, torch.no_grad():
        dh, _ = head.predict_homography(data)
        mf, cf, fp = head.forward_map_field(data["pf_hat_12"], None, None)
        torch.manual_seed(99)
        Hs, scores = head.dsac(cf, mf, hypothesis_no=4, points_per_hypothesis=16)
    err = -torch.log(scores)   # monotone in reprojection error; store raw errors recomputed below
    import kornia
    with torch.no_grad():
        e = []
        for j in range(4):
            pt = kornia.transform_points(Hs[:, j], cf)
            e.append((pt - mf).abs().sum(-1).sum(-1))
        e = torch.stack(e, 1)
    return {"choice": rec.calls[0].reshape(batch, -1).numpy(), "H": Hs.double().numpy().copy(),
            "repr_error": e.double().numpy().copy(), "scores": scores.double().numpy().copy(),
            "best": torch.argmax(scores, -1).numpy(), "delta_hat": dh.double().numpy().copy()}


def run_detone_scenario(ref_bb_cls, ref_head_cls, cfg, dtype, batch=4, seed=5):
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    load_synthetic(head.auxiliary_resnet, seed=0)
    # the fc layer's random init gives |delta| ~ 1; scale is irrelevant for parity
    model = torch.nn.Sequential(bb, head).to(dtype).train()
    d = synth.make_pairs(batch, seed=seed)
    data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
    loss, delta_gt, delta_hat = model(data)
    loss.backward()
    out = {"loss": np.float64(loss.item()), "delta_hat_12": delta_hat.detach().double().numpy().copy(),
           "delta_hat_21": data["delta_hat_21"].detach().double().numpy().copy()}
    for name in ("resnet34.conv1.weight", "resnet34.layer2.0.downsample.0.weight", "resnet34.fc.weight", "resnet34.fc.bias"):
        out["gradnorm/" + name] = np.float64(dict(bb.named_parameters())[name].grad.double().norm().item())
    return out


def run_orig_scenario(ref_bb_cls, ref_head_cls, cfg, dtype, batch, seed, steps=2):
    """The supervised "-orig" experiments (reference backbone, OneLine + reference NoOpHead + torch loss,
    train.py:318-322,379-387): `steps` Adam steps on one batch, then an eval-mode forward."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    sol = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=sol["LR"], betas=(sol["MOMENTUM_1"], sol["MOMENTUM_2"]), weight_decay=0)
    loss_fn = getattr(torch.nn, sol["LOSS"])()
    d = synth.make_pairs(batch, seed=seed, target=True)
    out = {"loss": [], "mace": []}
    key0 = cfg["MODEL"]["BACKBONE"]["TARGET_KEYS"][0]
    for it in range(steps):
        model.train()
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta", "target")}
        opt.zero_grad()
        ground_truth, network_output, delta_gt, delta_hat = model(data)
        loss = loss_fn(ground_truth, network_output)
        loss.backward()
        if it == 0:
            out["output0"] = sub(network_output) if network_output.dim() == 4 else network_output.detach().double().numpy().copy()
            out["output0_csum"] = csum(network_output)
            out["delta_hat0"] = delta_hat.detach().double().numpy().copy()
            names = dict(bb.named_parameters())
            for name in list(names)[:2] + list(names)[-2:]:
                out["gradnorm/" + name] = np.float64(names[name].grad.double().norm().item())
        opt.step()
        out["loss"].append(loss.item())
        out["mace"].append(float(np.mean(np.linalg.norm(
            (delta_gt - delta_hat).detach().double().numpy().reshape(-1, 2), axis=-1))))
    model.eval()
    with torch.no_grad():
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta", "target")}
        bb(data)
        out["eval_output"] = sub(data[key0]) if data[key0].dim() == 4 else data[key0].detach().double().numpy().copy()
        out["eval_output_csum"] = csum(data[key0])
    out["loss"], out["mace"] = np.array(out["loss"]), np.array(out["mace"])
    return out


def run_ihome_scenario(ref_bb_cls, ref_head_cls, cfg, dtype, batch=4, seed=31, steps=2):
    """iHomE (one-line): reference Rethinking (OneLine) + reference PerceptualHead with TRIPLET_LOSS='one-line' and a
    numeric margin; `steps` Adam steps on one batch with the multinomial draws recorded."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    load_synthetic(head.auxiliary_resnet, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    s = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=s["LR"], betas=(s["MOMENTUM_1"], s["MOMENTUM_2"]), weight_decay=0)
    d = synth.make_pairs(batch, seed=seed)
    out = {"loss": [], "mace": [], "choice_12": []}
    model.train()
    for it in range(steps):
        opt.zero_grad()
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        torch.manual_seed(3000 + it)
        with RecordMultinomial() as rec:
            loss, delta_gt, delta_hat = model(data)
        loss.backward()
        if it == 0:
            out["pf_hat_12_sub"] = sub(data["pf_hat_12"], 8)
            out["delta_hat_12"] = delta_hat.detach().double().numpy().copy()
            for name in ("layer1.0.weight", "layer4.6.upper_branch.0.weight", "layer8.3.weight", "layer8.3.bias"):
                out["gradnorm/" + name] = np.float64(dict(bb.named_parameters())[name].grad.double().norm().item())
        opt.step()
        out["loss"].append(loss.item())
        out["mace"].append(float(np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) -
                                                        delta_hat.detach().numpy().reshape(-1, 2), axis=-1))))
        out["choice_12"].append(rec.calls[0].reshape(batch, -1).numpy())
    for k in ("loss", "mace", "choice_12"):
        out[k] = np.asarray(out[k])
    return out


class RecordingWriter:
    """Stands in for the driver's SummaryWriter (train.py:312-314, 507): records what the head writes."""

    def __init__(self):
        self.scalars = {}

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.scalars["tb/%s/%s" % (tag, k)] = np.float64(v)


def run_detone_steps(ref_bb_cls, ref_head_cls, cfg, dtype, batch=8, seed=5, steps=3):
    """configs[3]'s model (ResNet-34 regressor + biHomE head): `steps` Adam steps on one batch (train.py:296-387)."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    load_synthetic(head.auxiliary_resnet, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    s = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=s["LR"], betas=(s["MOMENTUM_1"], s["MOMENTUM_2"]), weight_decay=0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=s["MILESTONES"], gamma=s["LR_DECAY"])
    d = synth.make_pairs(batch, seed=seed)
    out = {"loss": [], "mace": [], "delta_hat_12": []}
    model.train()
    for it in range(steps):
        opt.zero_grad()
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        loss, delta_gt, delta_hat = model(data)
        loss.backward()
        if it == 0:
            for name in ("resnet34.conv1.weight", "resnet34.layer2.0.downsample.0.weight", "resnet34.fc.weight", "resnet34.fc.bias"):
                out["gradnorm/" + name] = np.float64(dict(bb.named_parameters())[name].grad.double().norm().item())
        opt.step()
        sched.step()
        out["loss"].append(loss.item())
        out["mace"].append(float(np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) -
                                                        delta_hat.detach().numpy().reshape(-1, 2), axis=-1))))
        out["delta_hat_12"].append(delta_hat.detach().double().numpy().copy())
    for k in ("loss", "mace", "delta_hat_12"):
        out[k] = np.asarray(out[k])
    return out


def run_bihome_variant(ref_bb_cls, ref_head_cls, cfg, dtype, batch=4, seed=17, steps=1, photometric=0, loss_name=None):
    """Reference Rethinking + PerceptualHead with modified HEAD kwargs (extractor output layer, TRIPLET_LOSS '' ...) or
    data (photometric distortion): `steps` Adam steps on one batch with the driver's log-step side channel on
    (train.py:312-314), multinomial draws recorded.  loss_name: torch.nn loss applied by the driver (train.py:318-322)
    for the multihead branch, None for the head's own loss (train.py:330)."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    load_synthetic(head.auxiliary_resnet, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    s = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=s["LR"], betas=(s["MOMENTUM_1"], s["MOMENTUM_2"]), weight_decay=0)
    loss_fn = getattr(torch.nn, loss_name)() if loss_name else None
    d = synth.make_pairs(batch, seed=seed, photometric_max_delta=photometric)
    double = "double-line" in cfg["MODEL"]["HEAD"]["TRIPLET_LOSS"]
    out = {"loss": [], "mace": [], "choice_12": [], "choice_21": []}
    model.train()
    for it in range(steps):
        opt.zero_grad()
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        rw = RecordingWriter()
        if it == 0:
            data["summary_writer"], data["summary_writer_step"] = rw, 1
        torch.manual_seed(4000 + it)
        with RecordMultinomial() as rec:
            if loss_fn is not None:
                ground_truth, network_output, delta_gt, delta_hat = model(data)
                loss = loss_fn(ground_truth, network_output)
            else:
                loss, delta_gt, delta_hat = model(data)
        loss.backward()
        if it == 0:
            out.update(rw.scalars)
            out["pf_hat_12_sub"] = sub(data["pf_hat_12"], 8)
            out["delta_hat_12"] = delta_hat.detach().double().numpy().copy()
            if loss_fn is not None:
                out["ground_truth_csum"] = csum(ground_truth)
                out["network_output_csum"] = csum(network_output)
                out["network_output_sub"] = network_output[:, ::8, ::2, ::2].detach().double().numpy().copy()
            for name in ("layer1.0.weight", "layer4.6.upper_branch.0.weight", "layer8.3.weight", "layer8.3.bias"):
                out["gradnorm/" + name] = np.float64(dict(bb.named_parameters())[name].grad.double().norm().item())
            out["aux_bn1_running_mean"] = head.auxiliary_resnet.resnet.bn1.running_mean.double().numpy().copy()
        opt.step()
        out["loss"].append(loss.item())
        out["mace"].append(float(np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) -
                                                        delta_hat.detach().numpy().reshape(-1, 2), axis=-1))))
        out["choice_12"].append(rec.calls[0].reshape(batch, -1).numpy())
        if double:
            out["choice_21"].append(rec.calls[1].reshape(batch, -1).numpy())
    for k in ("loss", "mace", "choice_12", "choice_21"):
        out[k] = np.asarray(out[k])
    return out


class _ScalarRecorder:
    """Stand-in for the TensorBoard SummaryWriter the reference head writes to (TripletHead.py:158-186)."""

    def __init__(self):
        self.scalars = {}

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.scalars["tb/%s/%s" % (tag, k)] = float(v)


def run_zhang_scenario(ref_bb_cls, ref_head_cls, cfg, dtype, batch=4, seed=41, steps=2):
    """Round 3: the Zhang "Content-Aware" baseline - the reference's ContentAware.Model (feature extractor + fixed all-ones mask +
    resnet34) under its TripletHead.Model (config/s-coco/zhang-orig-lr-1e-2.yaml), `steps` Adam steps on one batch, then an eval
    forward.  The feature extractor runs four times per step (two patches in the backbone, two warped patches in the head), each call
    with its own BatchNorm batch statistics: the running statistics after the steps pin that order."""
    from bihome_amd import synth
    from bihome_amd.weights import load_synthetic
    bb = ref_bb_cls(**cfg["MODEL"]["BACKBONE"])
    head = ref_head_cls(bb, **cfg["MODEL"]["HEAD"])
    load_synthetic(bb, seed=0)
    model = torch.nn.Sequential(bb, head).to(dtype)
    s = cfg["SOLVER"]
    opt = torch.optim.Adam(model.parameters(), lr=s["LR"], betas=(s["MOMENTUM_1"], s["MOMENTUM_2"]), weight_decay=0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=s["MILESTONES"], gamma=s["LR_DECAY"])
    d = synth.make_pairs(batch, seed=seed)
    out = {"loss": [], "mace": []}
    grads_of = ("feature_extractor.layer1.0.weight", "feature_extractor.layer2.1.weight", "feature_extractor.layer3.0.weight",
                "feature_extractor.layer3.1.weight", "feature_extractor.layer3.1.bias", "resnet34.conv1.weight",
                "resnet34.layer4.2.conv2.weight", "resnet34.fc.bias")
    trained_mask = not cfg["MODEL"]["BACKBONE"]["FIX_MASK"]             # round 4: the mask predictor runs and is trained (ContentAware.py:36-50)
    if trained_mask:
        grads_of += ("mask_predictor.layer1.0.weight", "mask_predictor.layer3.1.weight", "mask_predictor.layer5.0.weight",
                     "mask_predictor.layer5.1.weight", "mask_predictor.layer5.1.bias")
    model.train()
    for it in range(steps):
        opt.zero_grad()
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        if it == 0:
            rec = _ScalarRecorder()
            data["summary_writer"], data["summary_writer_step"] = rec, 0
        loss, delta_gt, delta_hat = model(data)
        loss.backward()
        if it == 0:
            out.update({k: np.float64(v) for k, v in rec.scalars.items()})
            out["delta_hat_12"] = data["delta_hat_12"].detach().double().numpy().copy()
            out["delta_hat_21"] = data["delta_hat_21"].detach().double().numpy().copy()
            out["feature_1_sub"] = sub(data["feature_1"], 8)
            out["feature_2_csum"] = csum(data["feature_2"])
            params = dict(bb.named_parameters())
            for name in grads_of:
                g = params[name].grad.double()
                out["gradnorm/" + name] = np.float64(g.norm().item())
            out["grad/feature_extractor.layer3.0.weight"] = params["feature_extractor.layer3.0.weight"].grad.double().numpy().copy()
            out["grad/feature_extractor.layer1.0.weight"] = params["feature_extractor.layer1.0.weight"].grad.double().numpy().copy()
            if trained_mask:
                out["mask_1_sub"], out["mask_2_csum"] = sub(data["mask_1"], 8), csum(data["mask_2"])
                out["grad/mask_predictor.layer5.0.weight"] = params["mask_predictor.layer5.0.weight"].grad.double().numpy().copy()
                out["grad/mask_predictor.layer1.0.weight"] = params["mask_predictor.layer1.0.weight"].grad.double().numpy().copy()
        opt.step()
        sched.step()
        out["loss"].append(loss.item())
        out["mace"].append(float(np.mean(np.linalg.norm(delta_gt.numpy().reshape(-1, 2) -
                                                        delta_hat.detach().numpy().reshape(-1, 2), axis=-1))))
    sd = bb.state_dict()
    for k in ("feature_extractor.layer1.1.running_mean", "feature_extractor.layer1.1.running_var", "feature_extractor.layer3.1.running_mean",
              "feature_extractor.layer3.1.running_var", "feature_extractor.layer3.1.num_batches_tracked", "resnet34.bn1.running_mean") + \
            (("mask_predictor.layer1.1.running_mean", "mask_predictor.layer5.1.running_mean", "mask_predictor.layer5.1.running_var",
              "mask_predictor.layer5.1.num_batches_tracked") if trained_mask else ()):
        out["state/" + k] = sd[k].double().numpy().copy()
    model.eval()
    with torch.no_grad():
        data = {k: t(d[k], dtype) for k in ("patch_1", "patch_2", "delta")}
        delta_hat, H = head.predict_homography(bb.predict_homography(data))
    out["eval_delta_hat"] = delta_hat.double().numpy().copy()
    out["eval_H"] = H.double().numpy().copy()
    out["loss"], out["mace"] = np.asarray(out["loss"]), np.asarray(out["mace"])
    return out


def run_datagen(outdir):
    """The reference's own data-generation classes (src/data/transforms.py PhotometricDistortSimple :296-330,
    HomographyNetPrep :421-725, DictToGrayscale :344-354, DictStandardize :369-378) on seeded inputs, with the OpenCV
    calls served by oracle/refshim/cv2_standin.py.  Pins bihome_amd/synth.py (host generator) draw for draw."""
    import importlib
    from bihome_amd import synth
    T = importlib.import_module("src.data.transforms")
    assert os.path.realpath(T.__file__).startswith(os.path.realpath(REF)), T.__file__
    out = {}
    # (1) photometric distortion alone: 16 seeds x {32, 0} on a small uint8 image (+ one float image with out-of-range values)
    rng = np.random.Generator(np.random.PCG64(5))
    small = np.clip(synth.texture_image(rng, 16, 24), 0, 255).astype(np.uint8)
    out["photo_input"] = small
    for md in (32, 0):
        res = []
        for seed in range(16):
            rs = np.random.RandomState(seed)
            res.append(T.PhotometricDistortSimple(keys=["im"], max_delta=md, random_state=rs)({"im": small})["im"])
        out["photo_out_md%d" % md] = np.stack(res).astype(np.float32)
    # (2) whole samples: HomographyNetPrep -> DictToGrayscale -> DictStandardize on 240x320 uint8 images
    rng = np.random.Generator(np.random.PCG64(6))
    image = np.clip(synth.texture_image(rng, 240, 320), 0, 255).astype(np.uint8)
    out["prep_image_seed"] = np.int64(6)
    for md in (0, 32):
        recs = {"corners": [], "delta": [], "homography": [], "patch_1_sub": [], "patch_2_sub": [], "patch_1_csum": [],
                "patch_2_csum": [], "p1_std": [], "p2_std": []}
        for seed in range(4):
            prep = T.HomographyNetPrep(32, 128, ["image_1", "image_2"], md, "4_points", seed)
            d = prep(([image], None))
            recs["corners"].append(d["corners"]); recs["delta"].append(d["delta"]); recs["homography"].append(d["homography"])
            recs["patch_1_sub"].append(d["patch_1"][::4, ::4].astype(np.float32))
            recs["patch_2_sub"].append(d["patch_2"][::4, ::4].astype(np.float32))
            for k in ("patch_1", "patch_2"):
                a = d[k].astype(np.float64)
                recs[k + "_csum"].append([a.sum(), np.abs(a).sum(), (a * a).sum()])
            d = T.DictToGrayscale(["patch_1", "patch_2"])(d)
            d = T.DictStandardize([0.443], [0.129], ["patch_1", "patch_2"])(d)
            recs["p1_std"].append(d["patch_1"][::4, ::4, 0].astype(np.float32))
            recs["p2_std"].append(d["patch_2"][::4, ::4, 0].astype(np.float32))
        for k, v in recs.items():
            out["prep_md%d_%s" % (md, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(outdir, "datagen_ref.npz"), **out)
    print("datagen: photo", out["photo_out_md32"].shape, "prep corners", out["prep_md32_corners"][0].tolist())


def _round2(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir, which):
    """Fixtures added in round 2 (each file independent of the round-1 ones above)."""
    import copy
    PerceptualHead = importlib.import_module("src.heads.PerceptualHead")
    assert os.path.realpath(PerceptualHead.__file__).startswith(os.path.realpath(REF)), PerceptualHead.__file__

    def want(name):
        return not which or name in which

    if want("detone_b8"):
        r = run_detone_steps(ResNet34.Model, PerceptualHead.Model, configs.get("detone-bihome"), dtype)
        np.savez_compressed(os.path.join(outdir, "detone_b8_%s.npz" % tag), **r)
        print("detone_b8", tag, "loss", r["loss"], "mace", r["mace"])
    for layer in (2, 3, 4):
        if want("zeng_aux%d_b4" % layer):
            cfg = configs.get("zeng-bihome")
            cfg["MODEL"]["HEAD"]["AUXILIARY_RESNET_OUTPUT_LAYER"] = layer
            r = run_bihome_variant(Rethinking.Model, PerceptualHead.Model, cfg, dtype)
            np.savez_compressed(os.path.join(outdir, "zeng_aux%d_b4_%s.npz" % (layer, tag)), **r)
            print("zeng_aux%d" % layer, tag, "loss", r["loss"], "mace", r["mace"])
    if want("zeng_tb_b4"):
        r = run_bihome_variant(Rethinking.Model, PerceptualHead.Model, configs.get("zeng-bihome"), dtype, steps=2)
        np.savez_compressed(os.path.join(outdir, "zeng_tb_b4_%s.npz" % tag), **r)
        print("zeng_tb", tag, "loss", r["loss"], {k: float(v) for k, v in r.items() if k.startswith("tb/")})
    if want("zeng_multihead_b4"):
        cfg = configs.get("zeng-multihead")
        r = run_bihome_variant(Rethinking.Model, PerceptualHead.Model, cfg, dtype, steps=2, loss_name=cfg["SOLVER"]["LOSS"])
        np.savez_compressed(os.path.join(outdir, "zeng_multihead_b4_%s.npz" % tag), **r)
        print("zeng_multihead", tag, "loss", r["loss"], "mace", r["mace"])
    for base, loss_name in (("zeng-ihome", None), ("zeng-multihead", "L1Loss")):
        name = base.replace("-", "_") + "_n4_b4"
        if want(name):              # score-weighted multi-hypothesis training (PerceptualHead.py:276-280,505-511,708-710)
            cfg = configs.get(base)
            cfg["MODEL"]["HEAD"].update(RANSAC_HYPOTHESIS_NO=4, POINTS_PER_HYPOTHESIS=16)
            r = run_bihome_variant(Rethinking.Model, PerceptualHead.Model, cfg, dtype, batch=4, seed=19, steps=2, loss_name=loss_name)
            np.savez_compressed(os.path.join(outdir, "%s_%s.npz" % (name, tag)), **r)
            print(name, tag, "loss", r["loss"], "mace", r["mace"])
    if want("zeng_pds_b8"):
        r = run_bihome_variant(Rethinking.Model, PerceptualHead.Model, configs.get("zeng-bihome-pds"), dtype, batch=8, seed=8,
                               steps=3, photometric=32)
        np.savez_compressed(os.path.join(outdir, "zeng_pds_b8_%s.npz" % tag), **r)
        print("zeng_pds", tag, "loss", r["loss"], "mace", r["mace"])


def _orig(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir):
    NoOpHead = importlib.import_module("src.heads.NoOpHead")
    assert os.path.realpath(NoOpHead.__file__).startswith(os.path.realpath(REF)), NoOpHead.__file__
    r = run_orig_scenario(Rethinking.Model, NoOpHead.Model, configs.get("zeng-orig"), dtype, batch=4, seed=21)
    np.savez_compressed(os.path.join(outdir, "zeng_orig_b4_%s.npz" % tag), **r)
    print("zeng-orig", tag, "loss", r["loss"], "mace", r["mace"])
    r = run_orig_scenario(ResNet34.Model, NoOpHead.Model, configs.get("detone-orig"), dtype, batch=4, seed=22)
    np.savez_compressed(os.path.join(outdir, "detone_orig_b4_%s.npz" % tag), **r)
    print("detone-orig", tag, "loss", r["loss"], "mace", r["mace"])
    PerceptualHead = importlib.import_module("src.heads.PerceptualHead")
    r = run_ihome_scenario(Rethinking.Model, PerceptualHead.Model, configs.get("zeng-ihome"), dtype)
    np.savez_compressed(os.path.join(outdir, "zeng_ihome_b4_%s.npz" % tag), **r)
    print("zeng-ihome", tag, "loss", r["loss"], "mace", r["mace"])


def main():
    install_standins()
    import importlib
    Rethinking = importlib.import_module("src.backbones.Rethinking")
    ResNet34 = importlib.import_module("src.backbones.ResNet34")
    PerceptualHead = importlib.import_module("src.heads.PerceptualHead")
    assert os.path.realpath(Rethinking.__file__).startswith(os.path.realpath(REF)), Rethinking.__file__
    from bihome_amd import configs
    zeng, detone = configs.get("zeng-bihome"), configs.get("detone-bihome")
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    torch.set_num_threads(8)
    import warnings
    warnings.filterwarnings("ignore")
    orig_only = "--orig-only" in sys.argv          # regenerate just the supervised "-orig" fixtures
    round2 = [a for a in sys.argv[1:] if a.startswith("--round2")]      # --round2 or --round2=name1,name2
    if "--round4" in sys.argv:                     # trained masks (FIX_MASK False; plain and with the per-sample max normalisation): zhang_mask*_b4_*.npz
        import copy
        ContentAware = importlib.import_module("src.backbones.ContentAware")
        TripletHead = importlib.import_module("src.heads.TripletHead")
        for m in (ContentAware, TripletHead):
            assert os.path.realpath(m.__file__).startswith(os.path.realpath(REF)), m.__file__
        for name, strength in (("zhang_mask", None), ("zhang_masknorm", 0.5)):
            cfg = copy.deepcopy(configs.get("zhang-orig"))
            cfg["MODEL"]["BACKBONE"]["FIX_MASK"] = False
            if strength is not None:
                cfg["MODEL"]["BACKBONE"]["MASK_NORMALIZATION_STRENGTH"] = strength
            for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
                r = run_zhang_scenario(ContentAware.Model, TripletHead.Model, cfg, dtype)
                np.savez_compressed(os.path.join(outdir, "%s_b4_%s.npz" % (name, tag)), **r)
                print(name, tag, "loss", r["loss"], "mace", r["mace"])
        return
    if "--round3" in sys.argv:                     # the Zhang baseline (ContentAware + TripletHead): tests/golden/zhang_orig_b4_{f32,f64}.npz
        ContentAware = importlib.import_module("src.backbones.ContentAware")
        TripletHead = importlib.import_module("src.heads.TripletHead")
        for m in (ContentAware, TripletHead):
            assert os.path.realpath(m.__file__).startswith(os.path.realpath(REF)), m.__file__
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            r = run_zhang_scenario(ContentAware.Model, TripletHead.Model, configs.get("zhang-orig"), dtype)
            np.savez_compressed(os.path.join(outdir, "zhang_orig_b4_%s.npz" % tag), **r)
            print("zhang-orig", tag, "loss", r["loss"], "mace", r["mace"])
        return
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        if round2:
            names = round2[0].split("=", 1)[1].split(",") if "=" in round2[0] else []
            if tag == "f32" and (not names or "datagen" in names):
                run_datagen(outdir)
            _round2(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir, names)
            continue
        if orig_only:
            _orig(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir)
            continue
        r = run_head_scenario(PerceptualHead.Model, zeng, dtype)
        np.savez_compressed(os.path.join(outdir, "head_b8_%s.npz" % tag), **r)
        print("head", tag, "loss", r["loss"], "mace", r["mace"])
        r = run_dsac_n4(PerceptualHead.Model, zeng, dtype)
        np.savez_compressed(os.path.join(outdir, "dsac_n4_%s.npz" % tag), **r)
        print("dsac", tag, "best", r["best"])
        r = run_zeng_scenario(Rethinking.Model, PerceptualHead.Model, zeng, dtype)
        np.savez_compressed(os.path.join(outdir, "zeng_b8_%s.npz" % tag), **r)
        print("zeng", tag, "loss", r["loss"], "mace", r["mace"], "eval_mace", r["eval_mace"])
        r = run_detone_scenario(ResNet34.Model, PerceptualHead.Model, detone, dtype)
        np.savez_compressed(os.path.join(outdir, "detone_b4_%s.npz" % tag), **r)
        print("detone", tag, "loss", r["loss"])
        _orig(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir)
        _round2(importlib, Rethinking, ResNet34, configs, dtype, tag, outdir, [])
        if tag == "f32":
            run_datagen(outdir)


if __name__ == "__main__":
    main()
