"""Round 6: what the warp's adjoint costs inside the extractor stem's dgrad (bh_stem7_dgrad_c1_warp) - alternating launches of the plain stem
dgrad, the fused form, and the plain dgrad + warp_bwd it replaces, at the step's shape (128 images of 128 x 128, pool 4), rotating over
buffer sets larger than the Infinity Cache.  python tools/stem_warp_ab.py"""
import ctypes, os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib, check
B, size, pool, NSET = 128, 128, 4, 6
torch.manual_seed(0)
H64, _ = K.h4pt_fwd((torch.rand(B, 4, 2, device='cuda') - 0.5) * 32, size)
src = [torch.randn(B, 1, size, size, device='cuda') for _ in range(NSET)]
gy = [torch.randn(B, size // 2, size // 2, 64, device='cuda') for _ in range(NSET)]
gcov = [torch.randn(B, size // pool, size // pool, device='cuda') for _ in range(NSET)]
w = torch.randn(64, 7, 7, 1, device='cuda') * 0.05
gx = torch.empty(B, size, size, 1, device='cuda')
gH = torch.zeros(B, 9, dtype=torch.float64, device='cuda')
d = K.conv_desc(B, size, size, 1, 64, 7, 2, 3, precision=int(os.environ.get("STEM_PREC", "4")))
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def plain(i): check(lib.bh_stem7_dgrad_c1(p(gy[i]), p(w), p(gx), ctypes.byref(d), st()), "plain")
def fused(i): check(lib.bh_stem7_dgrad_c1_warp(p(gy[i]), p(w), None, ctypes.byref(d), p(src[i]), p(H64), p(gcov[i]), pool, p(gH), st()), "fused")
def two(i): plain(i); K.warp_bwd(src[i], H64, gx.view(B, 1, size, size), gcov[i], pool, gH=gH)
def bench(fn, n=60):
    for i in range(6): fn(i % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i % NSET)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for rnd in range(3):
    print("round %d: plain stem dgrad %.1f us | with the warp adjoint folded in %.1f us | plain + warp_bwd %.1f us" % (rnd, bench(plain), bench(fused), bench(two)), flush=True)
# the forward side (bh_stem7_fwd_warp; fp16-piece stem only): plain stem forward on a resident image | the stem making the warped pixels |
# warp_fwd + plain stem forward
if d.precision == 4:
    y = [torch.empty(B, size // 2, size // 2, 64, device='cuda') for _ in range(NSET)]
    warped = torch.empty(B, 1, size, size, device='cuda'); cov = torch.empty(B, size // 4, size // 4, device='cuda')
    sums = K.bn_stats_buffer(2, 64, 'cuda')
    def fplain(i): check(lib.bh_conv_fwd_bnstats(p(src[i]), p(w), None, p(y[i]), ctypes.byref(d), p(sums), 2, st()), "fwd plain")
    def ffused(i): check(lib.bh_stem7_fwd_warp(p(src[i]), p(H64), 4, p(w), None, p(y[i]), ctypes.byref(d), p(warped), p(cov), p(sums), 2, st()), "fwd fused")
    def ffused_noimg(i): check(lib.bh_stem7_fwd_warp(p(src[i]), p(H64), 4, p(w), None, p(y[i]), ctypes.byref(d), None, p(cov), p(sums), 2, st()), "fwd fused")
    def ftwo(i):
        check(lib.bh_warp_fwd_f(p(src[i]), p(H64), B, 1, size, size, 4, p(warped), p(cov), 0, st()), "warp_fwd")
        check(lib.bh_conv_fwd_bnstats(p(warped), p(w), None, p(y[i]), ctypes.byref(d), p(sums), 2, st()), "fwd plain")
    for rnd in range(3):
        print("round %d: plain stem fwd %.1f us | with the warp folded in %.1f us (without the image write %.1f) | warp_fwd + plain %.1f us"
              % (rnd, bench(fplain), bench(ffused), bench(ffused_noimg), bench(ftwo)), flush=True)
