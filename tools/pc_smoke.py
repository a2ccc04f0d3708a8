"""First contact with conv3x3_pc_kernel: one small and one multi-tile launch against conv3x3_halo_kernel, max |difference| printed.
Run under `timeout -s KILL`: a barrier-count mistake in a persistent kernel hangs the GPU."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import ROUTE_C3_PC, ROUTE_C3_TILE_WG, ROUTE_HALO_SMALL
for (N, H, Ci, Co) in [(2, 8, 64, 64), (8, 16, 64, 64), (4, 8, 256, 256), (40, 32, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, H, Ci, generator=g).cuda()
    gy = torch.randn(N, H, H, Co, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
    new = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_PC)
    old = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_HALO_SMALL | ROUTE_C3_TILE_WG)
    print((N, H, Ci, Co), K.conv_variant(K._with_layout(new, 4), "fwd"), flush=True)
    y0 = K.conv_fwd(x, wk, None, old, wpacked=pf); torch.cuda.synchronize()
    print("  halo done", flush=True)
    y1 = K.conv_fwd(x, wk, None, new, wpacked=pf); torch.cuda.synchronize()
    print("  fwd  max|diff| %.3e  equal %s  (|y| max %.3f)" % ((y1 - y0).abs().max().item(), torch.equal(y1, y0), y0.abs().max().item()), flush=True)
    if not torch.equal(y1, y0):
        bad = (y1 != y0).nonzero()
        print("  mismatches %d of %d; first %s; per-image %s" % (bad.shape[0], y0.numel(), bad[:5].tolist(), (y1 != y0).flatten(1).sum(1).tolist()[:8]))
    g0 = K.conv_dgrad(gy, wk, old, wpacked=pd); g1 = K.conv_dgrad(gy, wk, new, wpacked=pd); torch.cuda.synchronize()
    print("  dgrad max|diff| %.3e  equal %s" % ((g1 - g0).abs().max().item(), torch.equal(g1, g0)), flush=True)
print("SMOKE_DONE")
