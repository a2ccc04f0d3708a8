"""Run-to-run and arithmetic-to-arithmetic spread of tests/test_branches_gpu.py::test_pds_coco_three_steps_vs_golden (B = 8, three Adam steps
from random weights: chaotic after the first step).  Prints loss / MACE of the three steps for each arithmetic, several runs each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_optimizer, mace, train_step
from bihome_amd.weights import load_synthetic
g64 = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "zeng_pds_b8_f64.npz"))
g32 = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "zeng_pds_b8_f32.npz"))
print("reference f64: loss", g64["loss"], "mace", g64["mace"]); print("reference f32: loss", g32["loss"], "mace", g32["mace"])
d = synth.make_pairs(8, seed=8, photometric_max_delta=32)
for prec in ("f32-mfma", "f32x3", "f16x2", "f32x2"):
    for run in range(4):
        cfg = configs.get("zeng-bihome-pds")
        cfg["MODEL"]["BACKBONE"]["PRECISION"] = cfg["MODEL"]["HEAD"]["PRECISION"] = prec
        model = build_model(cfg); load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0); model.train()
        opt, sched = build_optimizer(model, cfg["SOLVER"])
        L, M = [], []
        for it in range(3):
            data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
            data["choice_12"] = torch.tensor(g64["choice_12"][it]).long().cuda(); data["choice_21"] = torch.tensor(g64["choice_21"][it]).long().cuda()
            loss, dgt, dh = train_step(model, data, opt, sched)
            L.append(round(loss.item(), 4)); M.append(round(mace(dgt, dh), 4))
        print("%-9s run %d: loss %s mace %s" % (prec, run, L, M))
