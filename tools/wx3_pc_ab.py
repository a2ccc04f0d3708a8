"""Round 5: fp16-piece weight gradient, four-wave form against the eight-wave producer / consumer form (bh_debug_force_tile(-32, 0 / 1)),
alternating; checks that both give bitwise the same gradient.  BIHOME_TUNING=1 python tools/wx3_pc_ab.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K, net
from bihome_amd._lib import lib


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ws = torch.empty(net.X3_WS_BYTES // 4, dtype=torch.float32, device='cuda')
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64)]:
    x = torch.relu(torch.randn(N, H, H, Ci, device="cuda")); gy = torch.randn(N, H, H, Co, device="cuda") * (torch.rand(N, H, H, Co, device="cuda") > 0.5)
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4)
    K.amax_of(x); K.amax_of(gy)
    out = {}
    for pc in (0, 1):
        lib.bh_debug_force_tile(-32, pc)
        gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
        out[pc] = gw
    same = torch.equal(out[0], out[1])
    res = {0: [], 1: []}
    gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    for rnd in range(3):
        for pc in (0, 1):
            lib.bh_debug_force_tile(-32, pc)
            res[pc].append(bench(lambda: K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)))
    lib.bh_debug_force_tile(-32, 1)
    print((N, H, Ci, Co), "bitwise same:", same, " four-wave", " ".join("%.1f" % v for v in res[0]), " eight-wave", " ".join("%.1f" % v for v in res[1]), "us (kernel + reduce)", flush=True)
