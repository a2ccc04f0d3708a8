"""Fused-tail timings at the bench shape (2 x 64 images of 128 x 128 x 16): forward on the matrix pipe vs the per-pixel VALU kernel, backward with a
DSAC-sparse (128 points per image) and a dense output gradient.  Usage: python tools/tail_bench.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
torch.manual_seed(0)
G, N, H, Ci, Cm, Co = 2, 64, 128, 16, 128, 2
x = torch.randn(G * N, H, H, Ci, device='cuda') * 1.5 + 0.3
w1 = torch.randn(Cm, Ci, device='cuda') / 4; b1 = torch.randn(Cm, device='cuda')
gamma = 1 + 0.3 * torch.randn(Cm, device='cuda'); beta = 0.3 * torch.randn(Cm, device='cuda')
w2 = torch.randn(Co, Cm, device='cuda') / 11; b2 = torch.randn(Co, device='cuda')
rm, rv = torch.zeros(Cm, device='cuda'), torch.ones(Cm, device='cuda')
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for route in (0, K.TAIL_ROUTE_VALU_FWD):
    t = bench(lambda: K.tail_fwd(x, w1, b1, gamma, beta, rm, rv, w2, b2, G, H * H, 1e-5, 0.1, True, route=route))
    print("tail_fwd route %d: %.1f us" % (route, t))
o0, ws = K.tail_fwd(x, w1, b1, gamma, beta, rm, rv, w2, b2, G, H * H, 1e-5, 0.1, True)
o1, _ = K.tail_fwd(x, w1, b1, gamma, beta, rm, rv, w2, b2, G, H * H, 1e-5, 0.1, True, route=K.TAIL_ROUTE_VALU_FWD)
print("fwd mfma vs valu: max abs diff %.3e (max |out| %.3f)" % ((o0 - o1).abs().max().item(), o1.abs().max().item()))
gw1, gg, gb, gw2, gb2 = [torch.zeros(s, device='cuda') for s in ((Cm, Ci), (Cm,), (Cm,), (Co, Cm), (Co,))]
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, dens in (("sparse(128/img)", 128.0 / (H * H)), ("dense", 1.0)):
    if which != "both" and not name.startswith(which):
        continue
    g = torch.randn(G * N, Co, H, H, device='cuda')
    if dens < 1:
        g = g * (torch.rand(G * N, 1, H, H, device='cuda') < dens)
    t = bench(lambda: K.tail_bwd(g, x, w1, b1, gamma, beta, w2, ws, rm, rv, G, H * H, 1e-5, True, True, gw1, gg, gb, gw2, gb2))
    print("tail_bwd %s: %.1f us" % (name, t))
