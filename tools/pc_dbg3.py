"""Where do conv3x3_pc_kernel and conv3x3_halo_kernel differ?  (debug aid: prints the mismatch pattern per case on the multi-tile shape)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from bihome_amd import kernels as K
import test_conv_pc_gpu as T

N, H, Ci, Co = 128, 32, 64, 64
new, old = T._descs(K, N, H, Ci, Co)
x, gy, wk, b, pf, pd, g = T._operands(K, N, H, Ci, Co)


def report(name, y1, y0):
    d = (y1 != y0)
    n = int(d.sum())
    print(name, "mismatches", n, "of", d.numel())
    if n:
        idx = d.nonzero()
        imgs = idx[:, 0].unique().tolist()
        print("   images", imgs[:20], "... n=", len(imgs))
        print("   rows", idx[:, 1].unique().tolist()[:40])
        print("   cols", idx[:, 2].unique().tolist()[:40])
        print("   chans", idx[:, 3].unique().tolist()[:70])
        i0 = idx[0].tolist()
        print("   first", i0, float(y1[tuple(i0)]), float(y0[tuple(i0)]))
        # per (image, 8x8 tile) counts
        t = (idx[:, 0] * 16 + (idx[:, 1] // 8) * 4 + idx[:, 2] // 8)
        u, c = t.unique(return_counts=True)
        print("   sub-tiles hit", len(u), "first", u[:16].tolist(), c[:16].tolist())


for rep in range(2):
    report("fwd", K.conv_fwd(x, wk, b, new, wpacked=pf), K.conv_fwd(x, wk, b, old, wpacked=pf))
    res = torch.randn(N, H, H, Co, generator=g).cuda()
    report("fwd res relu", K.conv_fwd(x, wk, b, new, res=res, relu=True, wpacked=pf), K.conv_fwd(x, wk, b, old, res=res, relu=True, wpacked=pf))
    report("fwd res", K.conv_fwd(x, wk, b, new, res=res, wpacked=pf), K.conv_fwd(x, wk, b, old, res=res, wpacked=pf))
    base = torch.randn(N, H, H, Ci, generator=g).cuda()
    o1, o0 = base.clone(), base.clone()
    K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd); K.conv_dgrad(gy, wk, old, out=o0, wpacked=pd)
    report("dgrad acc", o1, o0)
    d = o1 != o0
    print("   of these equal to the old gradient (nothing added):", int((o1[d] == base[d]).sum()), " plain dgrad value there == 0:", int(((o0 - base)[d] == 0).sum()))
    report("dgrad", K.conv_dgrad(gy, wk, new, wpacked=pd), K.conv_dgrad(gy, wk, old, wpacked=pd))
print("DONE")
# which term is wrong?  accumulate onto a constant 1000: mismatching values near dgrad (old term lost), near 1000 + other (accumulator wrong) or == 1000
g0 = K.conv_dgrad(gy, wk, old, wpacked=pd)
for rep in range(3):
    o1 = torch.full((N, H, H, Ci), 1000.0, device="cuda")
    K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd)
    d = o1 != (g0 + 1000.0)
    idx = d.nonzero()
    print("const-base: mismatches", int(d.sum()))
    for i in idx[:12].tolist():
        print("   ", i, "pc", float(o1[tuple(i)]), "expected", float(g0[tuple(i)]) + 1000.0, "dgrad", float(g0[tuple(i)]))
print("DONE2")
# do the wrong values belong to another tile of the same workgroup (T = 4 tiles: 8 tiles per image, tile = (ty, tx pair))?
o1 = torch.full((N, H, H, Ci), 1000.0, device="cuda")
K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd)
d = o1 != (g0 + 1000.0)
hits = {}
for i in d.nonzero()[:400].tolist():
    img, y, x, c = i
    v = float(o1[img, y, x, c]) - 1000.0
    wt = (y // 8) * 2 + (x // 16)
    for dt in range(-3, 4):
        w2 = wt + dt
        if dt == 0 or not (0 <= w2 < 8):
            continue
        y2, x2 = (w2 // 2) * 8 + y % 8, (w2 % 2) * 16 + x % 16
        if abs(float(g0[img, y2, x2, c]) - v) < 2e-3:
            hits[dt] = hits.get(dt, 0) + 1
    for dimg in (-1, 1):
        if 0 <= img + dimg < N:
            for w2 in range(8):
                y2, x2 = (w2 // 2) * 8 + y % 8, (w2 % 2) * 16 + x % 16
                if abs(float(g0[img + dimg, y2, x2, c]) - v) < 2e-3:
                    hits[(dimg, w2)] = hits.get((dimg, w2), 0) + 1
print("value belongs to tile offset:", hits, "of", min(400, int(d.sum())))
print("DONE3")
# exact search: accumulate onto zeros, then look for each wrong value anywhere in the reference result
o1 = torch.zeros((N, H, H, Ci), device="cuda")
K.conv_dgrad(gy, wk, new, out=o1, wpacked=pd)
d = o1 != g0
print("zero-base mismatches", int(d.sum()))
for i in d.nonzero()[:24].tolist():
    v = o1[tuple(i)]
    where = (g0 == v).nonzero()
    print("   at", i, "value", float(v), "expected", float(g0[tuple(i)]), "found at", where[:4].tolist())
print("DONE4")
