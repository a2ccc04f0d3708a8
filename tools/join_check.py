"""Two-branch BatchNorm join (BIHOME_BN_JOIN) against the unfused BatchNorms on the full Zeng model: loss and every parameter gradient of one
step.  Runs itself twice (child processes: the switch is read when the step runs).  python tools/join_check.py"""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np, torch
    from bihome_amd import configs, synth
    from bihome_amd.step import build_model
    from bihome_amd.weights import load_synthetic
    cfg = configs.get("zeng-bihome")
    model = build_model(cfg); load_synthetic(model[0], 0); load_synthetic(model[1].auxiliary_resnet, 0); model.train()
    B = 8
    d = synth.make_pairs(B, seed=5)
    g = torch.Generator().manual_seed(5)
    data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
    data["choice_12"], data["choice_21"] = [torch.randint(1, 128 * 128, (B, 128), generator=g).cuda() for _ in range(2)]
    loss, _, _ = model(data); loss.backward(); torch.cuda.synchronize()
    out = {k: p.grad.detach().float().cpu().numpy() for k, p in model[0].named_parameters()}
    out["__loss"] = np.array(loss.item())
    np.savez(sys.argv[1], **out)
    sys.exit(0)
import numpy as np
for v in ("1", "0"):
    subprocess.run([sys.executable, __file__, "/tmp/join_%s.npz" % v], env=dict(os.environ, BIHOME_BN_JOIN=v, BIHOME_DETERMINISTIC="1"), check=True)
a, b = np.load("/tmp/join_1.npz"), np.load("/tmp/join_0.npz")
print("loss join %.8f unfused %.8f" % (a["__loss"], b["__loss"]))
worst = sorted(((np.linalg.norm(a[k].astype(np.float64) - b[k]) / (np.linalg.norm(b[k]) + 1e-30), k) for k in a.files if k != "__loss"), reverse=True)
print("worst relative L2 differences of parameter gradients:", [(round(e, 9), k) for e, k in worst[:6]])
num = sum(np.sum((a[k].astype(np.float64) - b[k]) ** 2) for k in a.files if k != "__loss"); den = sum(np.sum(b[k].astype(np.float64) ** 2) for k in a.files if k != "__loss")
print("whole gradient relative L2 difference %.3e" % (num / den) ** 0.5)
