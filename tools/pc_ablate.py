"""Round 5: where the time of conv3x3_pc_kernel goes.  Ablation bits (BH_TUNING build, bh_debug_force_tile(-18, bits); wrong results, timing
only): 1 consumers issue no MFMA, 2 consumers issue no fragment reads, 4 H waves stage nothing, 8 E waves DMA no weights, 16 E waves run no
epilogue.  31 = barriers only.  BIHOME_TUNING=1 python tools/pc_ablate.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib, ROUTE_C3_PC, ROUTE_C3_TILE_WG
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (N, H, Ci, Co) in shapes:
    x = torch.randn(N, H, H, Ci, device='cuda')
    gy = torch.randn(N, H, H, Co, device='cuda')
    w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
    z = (torch.randn(N, H, H, Ci, device='cuda') * 1.5 + 0.3)
    gamma, beta = torch.rand(Ci, device='cuda') + 0.5, torch.randn(Ci, device='cuda') * 0.2
    st = K.bn_stats_buffer(2, Ci, "cuda"); K.bn_stats(z, st, 2, Ci)
    base = torch.randn(N, H, H, Ci, device='cuda')
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_C3_PC)
    dh = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_C3_TILE_WG)
    s = K.bn_stats_buffer(2, Co, "cuda"); s2 = K.bn_stats_buffer(2, Ci, "cuda")
    bnr = dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=s2, groups=2)
    for name, fn, fh in (("fwd", lambda: K.conv_fwd(x, wk, None, d, wpacked=pf), lambda: K.conv_fwd(x, wk, None, dh, wpacked=pf)),
                         ("fwd+stats", lambda: K.conv_fwd(x, wk, None, d, bn_sums=s, groups=2, wpacked=pf), None),
                         ("dgrad+bnr+acc", lambda: K.conv_dgrad(gy, wk, d, out=base, wpacked=pd, bn_reduce=bnr), None)):
        out = []
        for bits in (0, 31, 28, 3, 1, 4, 8, 16, 24, 12, 20):
            lib.bh_debug_force_tile(-18, bits)
            out.append('%d: %.1f' % (bits, bench(fn)))
        lib.bh_debug_force_tile(-18, 0)
        print((N, H, Ci, Co), name, ' | '.join(out), (' || halo %.1f' % bench(fh)) if fh else '', flush=True)
