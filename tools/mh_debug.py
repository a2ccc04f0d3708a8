import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_loss
from bihome_amd.weights import load_synthetic
for base in ("zeng-ihome", "zeng-multihead"):
    name = base.replace("-", "_") + "_n4_b4"
    g64 = dict(np.load(os.path.join(ROOT, "tests/golden", name + "_f64.npz")))
    g32 = dict(np.load(os.path.join(ROOT, "tests/golden", name + "_f32.npz")))
    cfg = configs.get(base)
    cfg["MODEL"]["HEAD"].update(RANSAC_HYPOTHESIS_NO=4, POINTS_PER_HYPOTHESIS=16)
    for variant in ("full", "no-score-grad"):
        model = build_model(cfg)
        load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0)
        loss_fn = build_loss(cfg["SOLVER"])
        d = synth.make_pairs(4, seed=19)
        data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"] = torch.tensor(g64["choice_12"][0]).cuda()
        model.train()
        if variant == "no-score-grad":
            import bihome_amd.heads.PerceptualHead as PH
            orig = PH._DsacScores.apply
            PH._DsacScores.apply = staticmethod(lambda pf, Hd: orig(pf.detach(), Hd.detach()))
        out = model(data)
        loss = loss_fn(out[0], out[1]) if isinstance(loss_fn, torch.nn.Module) else out[0]
        loss.backward()
        if variant == "no-score-grad":
            PH._DsacScores.apply = orig
        p = dict(model[0].named_parameters())
        print(base, variant, "loss", loss.item(), g64["loss"][0])
        for n_ in ("layer1.0.weight", "layer4.6.upper_branch.0.weight", "layer8.3.weight", "layer8.3.bias"):
            print("   %-34s hip %.6e  f64 %.6e  f32 %.6e" % (n_, p[n_].grad.double().norm().item(), g64["gradnorm/" + n_], g32["gradnorm/" + n_]))
