"""(product library: no tuning hooks; BIHOME_LIB_VARIANT=ab for an A/B)  Micro-benchmark of the HBM-bound launches of the head (warp fwd / adjoint, triplet fwd / bwd) at the bs-64 shapes,
rotating over buffer sets larger than the Infinity Cache: python tools/hbm_path_bench.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib

B, h, w, Cf, hf = 128, 128, 128, 64, 32
NSET = 12
torch.manual_seed(0)
dev = 'cuda'
delta = (torch.rand(B, 4, 2, device=dev) - 0.5) * 32
H64, _ = K.h4pt_fwd(delta, w)
imgs = [torch.randn(B, 1, h, w, device=dev) for _ in range(NSET)]
gos = [torch.randn(B, 1, h, w, device=dev) for _ in range(NSET)]
gcs = [torch.randn(B, h // 4, w // 4, device=dev) for _ in range(NSET)]
feats = [[torch.randn(B // 2, hf, hf, Cf, device=dev) for _ in range(4)] for _ in range(NSET)]
masks = [[torch.rand(B // 2, hf, hf, device=dev) for _ in range(2)] for _ in range(NSET)]


def timeit(fn, n=240):
    for i in range(24): fn(i % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i % NSET)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def timeit_pairs(fn, n=96):
    """per-launch event pairs, like bench.py's roofline leg"""
    ev = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i % NSET); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * sum(a.elapsed_time(b) for a, b in ev[16:]) / (n - 16)


def report(name, fn, nbytes):
    t = timeit(fn); tp = timeit_pairs(fn)
    print('%-34s %7.2f us back-to-back (%6.0f GB/s)   %7.2f us per event pair' % (name, t, nbytes / t / 1e3, tp))


print('empty event pair: %.2f us' % timeit_pairs(lambda i: None))
wb = 4.0 * (2 * B * h * w + B * (h // 4) * (w // 4))
report('warp_fwd', lambda i: K.warp_fwd(imgs[i], H64, 4), wb)
gH = torch.zeros(B, 9, dtype=torch.float64, device=dev)
report('warp_bwd', lambda i: K.warp_bwd(imgs[i], H64, gos[i], gcs[i], 4, gH=gH), wb)
fb = 4.0 * (4 * feats[0][0].numel() + 4 * masks[0][0].numel())
saved = {}
def tf(i):
    f = feats[i]; m = masks[i]
    saved[i] = K.triplet_l1_fwd(f[0], f[1], f[2], f[3], m[0], m[1])
report('triplet_fwd', tf, fb)
g = torch.ones(1, device=dev)
H1 = H64[:B // 2].contiguous(); H2 = H64[B // 2:].contiguous()
def tb(i):
    f = feats[i]; m = masks[i]; M1, M2, nd = saved[i]
    K.bihome_loss_bwd(g, f[0], f[1], f[2], f[3], m[0], m[1], None, None, M1, M2, nd, H1, H2, 0.01)
report('triplet_bwd', tb, 4.0 * (6 * feats[0][0].numel() + 6 * masks[0][0].numel()))
