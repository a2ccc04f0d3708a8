"""(PDS_ONLY_PERTURB=1: instead of the switches, nine rounding-level perturbations of the INPUTS per arithmetic.)
Round-5 ADVICE (tests/test_branches_gpu.py f32x3 band): where does the third step of the pds three-step test land, per build switch?
One subprocess per (arithmetic, switch): prints step-2 / step-3 MACE minus the float64 reference's.  Every variant computes the SAME
mathematics (the switches only choose fused / unfused kernels and summation orders), so the scatter over the variants is the sensitivity of
this trajectory to rounding-level changes.  python tools/pds_variants.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_optimizer, mace, train_step
from bihome_amd.weights import load_synthetic
CASE = os.environ.get("PDS_CASE", "zeng-pds")      # "detone": configs[3]'s three-step test (tests/test_branches_gpu.py::test_detone_three_steps_vs_golden)
g64 = np.load(os.path.join(%r, "tests", "golden", "detone_b8_f64.npz" if CASE == "detone" else "zeng_pds_b8_f64.npz"))
d = synth.make_pairs(8, seed=5) if CASE == "detone" else synth.make_pairs(8, seed=8, photometric_max_delta=32)
_k = int(os.environ.get("PDS_PERTURB", "0"))
if _k:        # rounding-level perturbation of the inputs: every pixel times (1 + k 2^-22) - what a different summation order does to a forward pass
    for key in ("patch_1", "patch_2"):
        d[key] = (d[key].astype(np.float64) * (1.0 + _k * 2.0 ** -22)).astype(np.float32)
cfg = configs.get("detone-bihome" if CASE == "detone" else "zeng-bihome-pds")
cfg["MODEL"]["BACKBONE"]["PRECISION"] = cfg["MODEL"]["HEAD"]["PRECISION"] = sys.argv[1]
model = build_model(cfg); load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0); model.train()
opt, sched = build_optimizer(model, cfg["SOLVER"])
M = []
for it in range(3):
    data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
    if CASE != "detone":
        data["choice_12"] = torch.tensor(g64["choice_12"][it]).long().cuda(); data["choice_21"] = torch.tensor(g64["choice_21"][it]).long().cuda()
    loss, dgt, dh = train_step(model, data, opt, sched)
    M.append(mace(dgt, dh) - float(g64["mace"][it]))
print("RESULT %%+.4f %%+.4f %%+.4f" %% tuple(M))
''' % (ROOT, ROOT)
VARIANTS = [("default", {}), ("warp adjoint as its own launch", {"BIHOME_WARP_IN_STEM_DGRAD": "0"}), ("BatchNorm adjoint not rebuilding the 1x1 dgrad", {"BIHOME_BN_FROM_1X1": "0"}),
            ("join adjoint reading y", {"BIHOME_JOIN_REMASK": "0"}), ("BatchNorm sums not in the dgrad epilogue", {"BIHOME_FUSE_BN_REDUCE": "0"}),
            ("no BatchNorm-on-load", {"BIHOME_BN_ON_LOAD": "0", "BIHOME_BN_ON_LOAD_1X1": "0"}), ("one stream", {"BIHOME_OVERLAP": "0"}),
            ("deterministic calls", {"BIHOME_DETERMINISTIC": "1"}), ("no weight packs (LDS-slab 3x3 path)", {"BIHOME_PACK_WEIGHTS": "0"})]
if os.environ.get("PDS_ONLY_PERTURB") == "1":
    VARIANTS = [("inputs x (1 + %d 2^-22)" % k, {"PDS_PERTURB": str(k)}) for k in range(0, 9)]
for prec in (("f32-mfma", "f16x2") if os.environ.get("PDS_ONLY_PERTURB") == "1" else ("f32x3", "f16x2", "f32-mfma")):
    for name, env in VARIANTS:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD, prec], env=e, capture_output=True, text=True)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-9s %-48s MACE - float64 reference at steps 1, 2, 3: %s" % (prec, name, res[0][7:] if res else "FAILED " + r.stderr[-300:]), flush=True)
