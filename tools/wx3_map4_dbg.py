import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from bihome_amd import kernels as K
for (N, Ci, Co) in [(8, 64, 64), (8, 128, 64), (128, 512, 512)]:
    g = torch.Generator().manual_seed(N + Ci)
    x = torch.randn(N, 4, 4, Ci, generator=g).cuda()
    gy = (torch.randn(N, 4, 4, Co, generator=g)).cuda()
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w, None, 1, 1)
    ref = torch.autograd.grad(y, w, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    d = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=4)
    need = K.wgrad_det_bytes(d)
    outs = []
    for fill in (0.0, 7.0, float("nan")):
        ws = torch.full((need // 4,), fill, dtype=torch.float32, device="cuda")
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
        outs.append(gw)
    rel = lambda a: ((a.double().cpu() - ref).norm() / ref.norm()).item()
    print((N, Ci, Co), K.conv_variant(d, "wgrad_det"), "need", need, "rel", [("%.2e" % rel(o)) for o in outs],
          "equal01", torch.equal(outs[0], outs[1]), "nan count", int(torch.isnan(outs[2]).sum()), flush=True)
    gwa = torch.zeros(Co, 3, 3, Ci, device="cuda"); K.conv_wgrad(x, gy, gwa, None, d)
    print("   atomics form rel %.2e" % rel(gwa))
    dd = (outs[0] - outs[1]).abs()
    if dd.max() > 0:
        idx = (dd > 0).nonzero()
        print("   differing entries", idx.shape[0], "co", idx[:, 0].unique().tolist()[:20], "taps", (idx[:, 1] * 3 + idx[:, 2]).unique().tolist(), "ci", idx[:, 3].unique().tolist()[:20])
