"""Round 5: tile sweep of conv_gemm_kernel on the transposed 2x2/2 and 1x1 layers of the decoder (bh_debug_force_tile(bm, bn), tuning build).
BIHOME_TUNING=1 python tools/gemm_tile_sweep.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


cases = [("convT", 128, 32, 64, 32), ("convT", 128, 64, 32, 32), ("convT", 128, 32, 64, 64), ("convT", 128, 64, 32, 16), ("convT", 128, 16, 128, 128),
         ("convT", 128, 8, 256, 128), ("1x1", 128, 128, 32, 16), ("1x1", 128, 64, 64, 32), ("1x1", 128, 32, 128, 64), ("1x1", 128, 16, 256, 128)]
tiles = [(0, 0), (-33, 0)]       # default (pw_kernel where it applies), then the pointwise kernel off
for kind, N, H, Ci, Co in cases:
    x = torch.randn(N, H, H, Ci, device="cuda")
    if kind == "convT":
        d = K.conv_desc(N, H, H, Ci, Co, 2, 2, 0, transposed=True, precision=4)
        w = torch.randn(Ci, 2, 2, Co, device="cuda") * 0.05
    else:
        d = K.conv_desc(N, H, H, Ci, Co, 1, 1, 0, precision=4)
        w = torch.randn(Co, 1, 1, Ci, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    row = []
    ref = None
    for bm, bn in tiles:
        lib.bh_debug_force_tile(bm, bn)
        try:
            y = K.conv_fwd(x, w, b, d)
            if ref is None:
                ref = y
            err = (y - ref).abs().max().item()
            row.append("%dx%d %.1f%s" % (bm, bn, bench(lambda: K.conv_fwd(x, w, b, d)), "" if err < 1e-4 else " ERR"))
            if bm == -33:
                lib.bh_debug_force_tile(-33, 1)
        except Exception as e:
            row.append("%dx%d n/a" % (bm, bn))
    lib.bh_debug_force_tile(0, 0)
    print(kind, (N, H, Ci, Co), K.conv_variant(d, "fwd"), " | ".join(row), flush=True)
