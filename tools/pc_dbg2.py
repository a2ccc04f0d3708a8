import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import ROUTE_C3_PC, ROUTE_C3_TILE_WG, ROUTE_HALO_SMALL
N, H, Ci, Co = 40, 32, 64, 64
g = torch.Generator().manual_seed(3)
gy = torch.randn(N, H, H, Co, generator=g).cuda()
w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
wk = w.permute(0, 2, 3, 1)
pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
z = (torch.randn(N, H, H, Ci, generator=g) * 1.5 + 0.3).cuda()
gamma = (torch.rand(Ci, generator=g) + 0.5).cuda(); beta = (torch.randn(Ci, generator=g) * 0.2).cuda()
st = K.bn_stats_buffer(2, Ci, "cuda"); K.bn_stats(z, st, 2, Ci)
new = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_PC)
old = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_TILE_WG)
for rep in range(3):
    outs = []
    for d in (new, old):
        sums = K.bn_stats_buffer(2, Ci, "cuda")
        r = K.conv_dgrad(gy, wk, d, wpacked=pd, bn_reduce=dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=sums, groups=2))
        outs.append(r)
    bad = (outs[0] != outs[1])
    print("rep", rep, "mismatches", int(bad.sum()), "of", bad.numel(), " max|diff| %.3e" % (outs[0] - outs[1]).abs().max().item())
    if bad.any():
        idx = bad.nonzero()
        imgs = sorted(set(idx[:, 0].tolist()))
        print("  images", imgs[:20], " rows", sorted(set(idx[:, 1].tolist()))[:16], " cols", sorted(set(idx[:, 2].tolist()))[:16], " ch", sorted(set(idx[:, 3].tolist()))[:16])
        b = idx[0].tolist(); print("  first", b, outs[0][tuple(b)].item(), outs[1][tuple(b)].item())
        # per (image, 8x8 tile) counts
        t = bad.reshape(N, 4, 8, 4, 8, Co).sum((2, 4, 5))
        nz = t.nonzero()
        print("  tiles (img, ty, tx): count", [(tuple(v.tolist()), int(t[tuple(v.tolist())])) for v in nz[:12]])
ref = torch.nn.functional.conv_transpose2d(gy.double().cpu().permute(0, 3, 1, 2), wk.double().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
for name, d in (("pc", new), ("halo", old)):
    errs = []
    for rep in range(4):
        sums = K.bn_stats_buffer(2, Ci, "cuda")
        r = K.conv_dgrad(gy, wk, d, wpacked=pd, bn_reduce=dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=sums, groups=2))
        e = (r.cpu().double() - ref).abs()
        errs.append((int((e > 1e-3).sum()), e.max().item()))
    print(name, "elements off by > 1e-3 from float64, max error:", errs)
