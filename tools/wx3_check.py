"""f32x3 weight-gradient kernel (wgrad_x3.hip) against torch float64, next to the fp32-MFMA kernel's error and time."""
import sys; sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
from bihome_amd import kernels as K


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(N, H, Ci, Co, time_it=True):
    g = torch.Generator().manual_seed(N + H + Ci)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    ref = None
    if N * H * H * max(Ci, Co) <= 1 << 23:
        xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
        w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xd, w, None, 1, 1)
        ref = torch.autograd.grad(y, w, gy.double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)      # [Co][3][3][Ci]
    out = {}
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec)
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d)
        out[prec] = gw
        msg = "%-28s" % K.conv_variant(d, "wgrad")
        if ref is not None:
            msg += " rel L2 %.3e max %.3e" % (((gw.cpu().double() - ref).norm() / ref.norm()).item(),
                                            ((gw.cpu().double() - ref).abs().max() / ref.abs().max()).item())
        if time_it:
            us = bench(lambda: K.conv_wgrad(x, gy, gw, None, d))
            msg += "  %.1f us (%.0f TF)" % (us, K.conv_flops(d) / us / 1e6)
        print(msg, flush=True)
        need = K.wgrad_det_bytes(d)
        if need:
            ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
            runs = []
            for _ in range(2):
                g2 = torch.zeros(Co, 3, 3, Ci, device="cuda")
                K.conv_wgrad(x, gy, g2, None, d, det_ws=ws)
                runs.append(g2)
            msg = "%-28s" % ("  det " + K.conv_variant(d, "wgrad")) + " ws %.1f MB repeatable %s" % (need / 1e6, torch.equal(runs[0], runs[1]))
            if ref is not None:
                msg += " rel L2 %.3e" % ((runs[0].cpu().double() - ref).norm() / ref.norm()).item()
            if time_it:
                us = bench(lambda: K.conv_wgrad(x, gy, g2, None, d, det_ws=ws))
                msg += "  %.1f us" % us
            print(msg, flush=True)
    dd = ((out[0] - out[2]).norm() / out[0].norm()).item()
    print("   N%d %dx%d %d->%d  |f32 - f32x3| / |f32| = %.3e" % (N, H, H, Ci, Co, dd), flush=True)


if __name__ == "__main__":
    for shp in ((2, 8, 64, 64), (3, 24, 64, 128), (8, 8, 256, 128), (4, 64, 32, 32), (3, 24, 32, 96)):
        run(*shp, time_it=False)
    for shp in ((128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64), (128, 64, 32, 32), (128, 128, 32, 32)):
        run(*shp)
