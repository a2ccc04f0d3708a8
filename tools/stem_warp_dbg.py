"""debug: fused stem dgrad + warp adjoint against the two calls, term by term"""
import ctypes, os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import torch.nn.functional as F
from bihome_amd import kernels as K
from bihome_amd._lib import lib, check
def rand_delta(B, seed, amp=32):
    g = np.random.Generator(np.random.PCG64(seed))
    return g.uniform(-amp, amp, (B, 4, 2)).astype(np.float32)
B, size, pool = 6, 128, 4
rng = np.random.Generator(np.random.PCG64(B * 7 + size))
src = F.avg_pool2d(torch.tensor(rng.standard_normal((B, 1, size, size)).astype(np.float32)), 3, 1, 1).cuda().contiguous()
gy = torch.tensor(rng.standard_normal((B, size // 2, size // 2, 64)).astype(np.float32)).cuda()
w = torch.tensor((rng.standard_normal((64, 7, 7, 1)) * 0.05).astype(np.float32)).cuda()
gcov = torch.tensor(rng.standard_normal((B, size // pool, size // pool)).astype(np.float32)).cuda()
H64, _ = K.h4pt_fwd(torch.tensor(rand_delta(B, size + 1, amp=size / 4.0)).cuda(), size)
d = K.conv_desc(B, size, size, 1, 64, 7, 2, 3)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, gy_, gc_ in (("both", gy, gcov), ("image term only", gy, None), ("coverage term only", torch.zeros_like(gy), gcov)):
    gx0 = torch.empty(B, size, size, 1, device="cuda")
    check(lib.bh_stem7_dgrad_c1(p(gy_), p(w), p(gx0), ctypes.byref(d), st), "plain")
    g0 = K.warp_bwd(src, H64, gx0.view(B, 1, size, size), gc_, pool)
    g1 = torch.zeros(B, 9, dtype=torch.float64, device="cuda")
    check(lib.bh_stem7_dgrad_c1_warp(p(gy_), p(w), None, ctypes.byref(d), p(src), p(H64), p(gc_), pool, p(g1), st), "fused")
    torch.cuda.synchronize()
    print(name)
    for b in range(B):
        print("  img %d two calls %s" % (b, " ".join("%11.4e" % v for v in g0[b].tolist())))
        print("        fused     %s" % " ".join("%11.4e" % v for v in g1[b].tolist()))
