"""The same short training run (fresh device-generated pairs every step, same seeds) in the two fp32-accurate arithmetics -
'f32' (round 4: fp16 pieces in the 3x3 layers), 'f32x3' (the exact three-piece cut) and 'f32-mfma' (fp32-input MFMA everywhere) - and, for scale, twice in 'f32-mfma' with
different atomics orders: loss / MACE averaged over windows of steps.  Training from random weights is chaotic, so the
trajectories differ step by step; what must agree is their statistics.   python tools/train_compare.py [steps] [batch]"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import configs
from bihome_amd.step import build_model, build_optimizer, mace, train_step
from bihome_amd.synth_gpu import GpuPairGenerator
from bihome_amd.weights import load_synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
win = max(steps // 6, 1)


def run(precision, tag):
    cfg = configs.get("zeng-bihome")
    cfg["MODEL"]["BACKBONE"]["PRECISION"] = cfg["MODEL"]["HEAD"]["PRECISION"] = precision
    torch.manual_seed(0)
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    gen = GpuPairGenerator(n_images=16, seed=42)
    L, M = [], []
    for it in range(steps):
        data = gen.next(B)
        loss, dgt, dh = train_step(model, data, opt, sched)
        L.append(loss.item() / B); M.append(mace(dgt, dh))
    out = []
    for w in range(0, steps, win):
        out.append("%8.3f/%6.2f" % (sum(L[w:w + win]) / len(L[w:w + win]), sum(M[w:w + win]) / len(M[w:w + win])))
    print("%-22s loss per pair / MACE per window of %d steps: %s" % (tag, win, "  ".join(out)), flush=True)


run("f32", "f32 (f16x2, round 4)")
run("f32x3", "f32x3 (exact cut)")
run("f32-mfma", "f32-mfma run 1")
run("f32-mfma", "f32-mfma run 2")
