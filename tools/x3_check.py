"""precision 2 ("f32x3": three exact bf16 pieces per operand, six bf16 MFMAs per product) of the packed 3x3 kernel against
torch float64, next to the exact-fp32 MFMA kernel's error on the same data, and back-to-back timings."""
import sys
import torch
import torch.nn.functional as F
from bihome_amd import kernels as K
from bihome_amd._lib import ROUTE_HALO_SMALL


def run(N, H, Ci, Co, scale=0.05, wide=False, time_it=True):
    g = torch.Generator().manual_seed(N * 7 + H)
    x = torch.randn(N, H, H, Ci, generator=g)
    if wide:                                   # operands spread over many binades
        x = x * torch.exp2(torch.randint(-12, 12, x.shape, generator=g).float())
    x = x.cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * scale).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    b = torch.randn(Co, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    out = {}
    ref = refd = None
    if N * H * H * Ci <= 1 << 24:
        xd, wd = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu()
        ref = F.conv2d(xd, wd, b.double().cpu(), 1, 1).permute(0, 2, 3, 1)
        refd = F.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), wd, None, 1, 1).permute(0, 2, 3, 1)
    for prec in (0, 2):
        d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=prec, route=ROUTE_HALO_SMALL)
        pk = K.WeightPacker(split=prec == 2)
        pf, pd = pk.get(w)
        pk.refresh()
        y = K.conv_fwd(x, wk, b, d, wpacked=pf)
        gx = K.conv_dgrad(gy, wk, d, wpacked=pd)
        s = K.bn_stats_buffer(1, Co, "cuda")
        y2 = K.conv_fwd(x, wk, b, d, bn_sums=s, groups=1, wpacked=pf)
        assert torch.equal(y, y2)
        out[prec] = (y, gx)
        name = K.conv_variant(K._with_layout(d, 2 if prec == 2 else 1), "fwd")
        msg = "%-52s" % name
        if ref is not None:
            e = ((y.cpu().double() - ref).norm() / ref.norm()).item()
            em = ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item()
            ed = ((gx.cpu().double() - refd).norm() / refd.norm()).item()
            msg += " fwd rel L2 %.3e max %.3e  dgrad rel L2 %.3e" % (e, em, ed)
        if time_it:
            for fn, tag in ((lambda: K.conv_fwd(x, wk, b, d, wpacked=pf), "fwd"), (lambda: K.conv_dgrad(gy, wk, d, wpacked=pd), "dgrad")):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / 20
                msg += "  %s %.1f us (%.0f TF)" % (tag, us, K.conv_flops(d) / us / 1e6)
        print(msg, flush=True)
    dd = (out[0][0] - out[2][0]).abs().max().item()
    print("   N%d %dx%d %d->%d%s  max |f32 - f32x3| = %.3e" % (N, H, H, Ci, Co, " wide" if wide else "", dd), flush=True)


if __name__ == "__main__":
    for shp in ((8, 16, 64, 64), (4, 8, 256, 256), (16, 16, 32, 32), (16, 16, 128, 128), (3, 24, 96, 160)):
        run(*shp, time_it=False)
    run(8, 16, 64, 64, wide=True, time_it=False)
    for shp in ((128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 32, 32), (128, 64, 64, 64)):
        run(*shp)
