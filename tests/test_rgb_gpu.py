"""BASELINE.json configs[4]: 256x256 RGB patch pairs (6-channel stem, 3-channel warp / extractor).

Upstream has no RGB path (Rethinking.py:31 hard-wires 2 input channels, PerceptualHead.py:352,361 reshape to one
channel), so parity here is (i) the individual kernels against plain torch float64 on the CPU and (ii) the
self-consistency SURVEY.md 0 prescribes: an RGB batch with three equal channels must reproduce the grayscale
path (whose parity with the reference the other GPU tests pin).
"""
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


@pytest.mark.parametrize("N,H", [(2, 32), (4, 128)])      # general NCHW-epilogue GEMM path / two-step (tap table + col2im) path
def test_rgb_stem_dgrad_vs_torch(N, H):
    from bihome_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 3, 7, 7, generator=g, dtype=torch.float64) * 0.05
    x = torch.randn(N, 3, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 3)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (gx_ref,) = torch.autograd.grad(y, x, gy)
    d = K.conv_desc(N, H, H, 3, 64, 7, 2, 3, in_nchw=True)
    wk = w.float().cuda().permute(0, 2, 3, 1).contiguous()
    gyk = gy.float().cuda().permute(0, 2, 3, 1).contiguous()
    y_gpu = K.conv_fwd(x.detach().float().cuda().contiguous(), wk, None, d)
    assert relerr(y_gpu.permute(0, 3, 1, 2).cpu(), y.detach()) < 2e-5
    gx = K.conv_dgrad(gyk, wk, d)
    assert gx.shape == (N, 3, H, H)
    assert relerr(gx.cpu(), gx_ref) < 2e-5


def test_rgb_equal_channels_reproduce_grayscale_step():
    """One training step at the config-5 shapes (256x256, 6-channel stem) with R=G=B against the grayscale model whose
    stem weights are the channel sums: same loss, same delta_hat, same gradients."""
    from bihome_amd.step import build_model
    P, B = 256, 2
    cfg3 = configs.get("zeng-bihome-rgb256")
    cfg1 = configs.get("zeng-bihome")
    cfg1["MODEL"]["BACKBONE"]["IMAGE_SIZE"] = P
    cfg1["MODEL"]["HEAD"]["PATCH_SIZE"] = P
    m1, m3 = build_model(cfg1), build_model(cfg3)
    load_synthetic(m3[0], 0)
    load_synthetic(m3[1].auxiliary_resnet, 0)
    sd = {k: v.clone() for k, v in m3.state_dict().items()}
    w6 = sd["0.layer1.0.weight"]                                   # [64,6,7,7]: channels (p1 r,g,b, p2 r,g,b)
    w2 = torch.stack([w6[:, 0:3].sum(1), w6[:, 3:6].sum(1)], 1)
    for k in sd:
        if k.endswith("layer1.0.weight") and sd[k].shape[1] == 6:   # "0.layer1.0.weight" and the head's "1.backbone." alias
            sd[k] = w2
    m1.load_state_dict(sd)
    d = synth.make_pairs(B, patch=P, rho=64, seed=5)
    g = torch.Generator().manual_seed(11)
    choice = {k: torch.randint(0, P * P, (B, 128), generator=g).cuda() for k in ("choice_12", "choice_21")}
    out = {}
    for name, model, rep in (("gray", m1, 1), ("rgb", m3, 3)):
        model.train()
        data = {k: cuda(d[k]).repeat(1, rep, 1, 1).contiguous() for k in ("patch_1", "patch_2")}
        data["delta"] = cuda(d["delta"])
        data.update(choice)
        loss, _, dh = model(data)
        loss.backward()
        torch.cuda.synchronize()
        out[name] = (loss.item(), dh.detach().cpu().numpy(),
                     {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None})
    (l1, dh1, g1), (l3, dh3, g3) = out["gray"], out["rgb"]
    assert np.isfinite(l3)
    assert abs(l3 - l1) <= 2e-3 * abs(l1)
    assert relerr(dh3, dh1) < 2e-3
    # d loss / d w6[:, j] = d loss / d w2[:, j // 3] when the three planes are equal
    gw6, gw2 = g3["0.layer1.0.weight"], g1["0.layer1.0.weight"]
    for j in range(6):
        assert relerr(gw6[:, j], gw2[:, j // 3]) < 5e-2
    num = sum(float(((g3[k] - g1[k]) ** 2).sum()) for k in g1 if k != "0.layer1.0.weight")
    den = sum(float((g1[k] ** 2).sum()) for k in g1 if k != "0.layer1.0.weight")
    assert (num / den) ** 0.5 < 5e-2


def test_rgb_pairs_train_step_runs():
    """Distinct RGB channels: a full optimiser step at 256x256x3 gives finite loss / MACE and changes the weights."""
    from bihome_amd.step import build_model, build_optimizer, mace, train_step
    cfg = configs.get("zeng-bihome-rgb256")
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    load_synthetic(model[1].auxiliary_resnet, 0)
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    d = synth.make_pairs(2, patch=256, rho=64, seed=6, channels=3)
    assert d["patch_1"].shape == (2, 3, 256, 256)
    w0 = model[0].layer1[0].weight.detach().clone()
    loss, dgt, dh = train_step(model, {k: cuda(d[k]) for k in ("patch_1", "patch_2", "delta")}, opt, sched)
    assert np.isfinite(loss.item()) and np.isfinite(mace(dgt, dh))
    assert model[0].layer1[0].weight.shape == (64, 6, 7, 7)
    assert (model[0].layer1[0].weight.detach() - w0).abs().max().item() > 0
