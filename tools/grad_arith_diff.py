"""Round 6: where do the split-operand arithmetics' gradients differ from the fp32-input MFMA path's?  Same weights, same pds batch (B = 8),
one forward + backward each (no optimizer step); per parameter: relative L2 difference to the 'f32-mfma' model's gradient, and the same
for a second 'f32-mfma' run (the noise floor: order of the fp32 atomics).  python tools/grad_arith_diff.py [precisions...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bihome_amd import configs, synth
from bihome_amd.step import build_model, build_optimizer
from bihome_amd.weights import load_synthetic
g64 = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "zeng_pds_b8_f64.npz"))
d = synth.make_pairs(8, seed=8, photometric_max_delta=32)
def grads(prec):
    # "precision[@ENV=VALUE]": an environment switch for this run only (read at call time by the kernels' Python layer)
    prec, _, envkv = prec.partition("@")
    saved = None
    if envkv:
        ek, _, ev = envkv.partition("=")
        saved = (ek, os.environ.get(ek)); os.environ[ek] = ev
    try:
        return _grads(prec)
    finally:
        if saved is not None:
            if saved[1] is None: os.environ.pop(saved[0], None)
            else: os.environ[saved[0]] = saved[1]


def _grads(prec):
    cfg = configs.get("zeng-bihome-pds")
    bbp, hp = (prec.split("/") + [prec])[:2]          # "backbone/head" or one name for both
    cfg["MODEL"]["BACKBONE"]["PRECISION"], cfg["MODEL"]["HEAD"]["PRECISION"] = bbp, hp
    model = build_model(cfg); load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0); model.train()
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
    data["choice_12"] = torch.tensor(g64["choice_12"][0]).long().cuda(); data["choice_21"] = torch.tensor(g64["choice_21"][0]).long().cuda()
    opt.zero_grad()
    loss, _, _ = model(data)
    loss.backward()
    torch.cuda.synchronize()
    gpf = {"pf_hat_12.grad": None}
    return loss.item(), {n: p.grad.detach().double().cpu().clone() for n, p in model[0].named_parameters() if p.grad is not None}
l0, ref = grads(os.environ.get("GRAD_DIFF_REF", "f32-mfma"))
for prec in (sys.argv[1:] or ["f32-mfma", "f32x3", "f16x2"]):
    l, g = grads(prec)
    rows = []
    num = den = 0.0
    for n in ref:
        a, b = g[n], ref[n]
        dn, bn = (a - b).norm().item(), b.norm().item()
        num += dn * dn; den += bn * bn
        flips = int(((a * b) < 0).sum()); 
        rows.append((dn / (bn + 1e-300), n, tuple(b.shape), bn, flips, b.numel()))
    rows.sort(reverse=True)
    print("== %s vs f32-mfma: loss %.8f vs %.8f; whole-gradient relative L2 difference %.3e; sign flips %d of %d" %
          (prec, l, l0, (num / den) ** 0.5, sum(r[4] for r in rows), sum(r[5] for r in rows)))
    for r in rows[:12]:
        print("   %.3e  %-44s %-18s |g| %.3e  sign flips %d / %d" % r)
    if os.environ.get("GRAD_DIFF_ALL") == "1":            # every tensor, in the network's (forward) order
        byname = {r[1]: r for r in rows}
        for n in ref:
            print("   all  %.3e  %-44s %-18s |g| %.3e" % byname[n][:4])
