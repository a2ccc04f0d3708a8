"""Round 6: wall time per launch of conv3x3_pc_kernel in the library variant BIHOME_LIB_VARIANT names (tools/pc_emu_build.sh: PC_EMU timing
variants, wrong results) - forward, forward + statistics, dgrad with BatchNorm sums, on the three step shapes."""
import sys; sys.path.insert(0, '.')
import os, torch
from bihome_amd import kernels as K
from bihome_amd._lib import ROUTE_C3_PC
def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
DATA = os.environ.get("PC_DATA", "randn")      # randn | zeros | ones: what the operands toggle (the matrix pipe's clock depends on it)
def mk(*shape):
    return torch.randn(*shape, device='cuda') if DATA == "randn" else (torch.zeros(*shape, device='cuda') if DATA == "zeros" else torch.ones(*shape, device='cuda'))
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64)]:
    x = mk(N, H, H, Ci); gy = mk(N, H, H, Co)
    w = (mk(Co, Ci, 3, 3) * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
    z = torch.randn(N, H, H, Ci, device='cuda') * 1.5 + 0.3
    gamma, beta = torch.rand(Ci, device='cuda') + 0.5, torch.randn(Ci, device='cuda') * 0.2
    st = K.bn_stats_buffer(2, Ci, "cuda"); K.bn_stats(z, st, 2, Ci)
    base = torch.randn(N, H, H, Ci, device='cuda')
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_C3_PC)
    s = K.bn_stats_buffer(2, Co, "cuda"); s2 = K.bn_stats_buffer(2, Ci, "cuda")
    bnr = dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=s2, groups=2)
    r = [bench(lambda: K.conv_fwd(x, wk, None, d, wpacked=pf)), bench(lambda: K.conv_fwd(x, wk, None, d, bn_sums=s, groups=2, wpacked=pf)),
         bench(lambda: K.conv_dgrad(gy, wk, d, out=base, wpacked=pd, bn_reduce=bnr))]
    print(os.environ.get("BIHOME_LIB_VARIANT", "product"), DATA, (N, H, Ci, Co), "fwd %.1f  fwd+stats %.1f  dgrad+bnr+acc %.1f us" % tuple(r), flush=True)
