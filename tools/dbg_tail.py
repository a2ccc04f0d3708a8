import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
torch.manual_seed(0)
G, N, H, Ci, Cm, Co = 2, 2, 16, 16, 128, 2
x = torch.randn(G * N, H, H, Ci, device='cuda') * 1.5 + 0.3
w1 = torch.randn(Cm, Ci, device='cuda') / 4; b1 = torch.randn(Cm, device='cuda')
gamma = 1 + 0.3 * torch.randn(Cm, device='cuda'); beta = 0.3 * torch.randn(Cm, device='cuda')
w2 = torch.randn(Co, Cm, device='cuda') / 11; b2 = torch.randn(Co, device='cuda')
def run(route):
    rm, rv = torch.zeros(Cm, device='cuda'), torch.ones(Cm, device='cuda')
    o, ws = K.tail_fwd(x, w1, b1, gamma, beta, rm, rv, w2, b2, G, H * H, 1e-5, 0.1, True, route=route)
    return o, ws, rm, rv
o0, ws0, rm0, rv0 = run(0)
o1, ws1, rm1, rv1 = run(1)
# float64 reference
xd = x.double().reshape(G, N * H * H, Ci)
outs = []
for g in range(G):
    h = xd[g] @ w1.double().t() + b1.double()
    mu, var = h.mean(0), h.var(0, unbiased=False)
    t = torch.relu((h - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
    outs.append((t @ w2.double().t() + b2.double()).reshape(N, H * H, Co).permute(0, 2, 1))
ref = torch.cat(outs, 0).reshape(G * N, Co, H, H)
print("ws equal:", torch.equal(ws0[:G*Cm*2], ws1[:G*Cm*2]), "rm equal", torch.equal(rm0, rm1))
for name, o in (("mfma", o0), ("valu", o1)):
    e = (o.double() - ref).abs()
    print(name, "max err %.3e" % e.max().item(), "bad count", int((e > 1e-4).sum()))
    bad = (e > 1e-4).nonzero()
    print(bad[:40].tolist())
