"""Round 4: fp16-piece weight-gradient kernel (wgrad_x3_kernel<..,2,true> + reduce) - workgroups per launch (bh_debug_force_tile(-30, n)),
alternating.  BIHOME_TUNING=1 python tools/wx3_f16_sweep.py"""
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
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256), (128, 64, 64, 64), (128, 128, 32, 32), (128, 64, 32, 32)]:
    x = torch.relu(torch.randn(N, H, H, Ci, device="cuda")); gy = torch.randn(N, H, H, Co, device="cuda") * (torch.rand(N, H, H, Co, device="cuda") > 0.5)
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4)
    gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    K.amax_of(x); K.amax_of(gy)
    res = {}
    for rnd in range(2):
        for tgt in (128, 256, 384, 512, 768):
            lib.bh_debug_force_tile(-30, tgt)
            res.setdefault(tgt, []).append(bench(lambda: K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)))
    lib.bh_debug_force_tile(-30, 256)
    print((N, H, Ci, Co), K.conv_variant(d, "wgrad_det"), " | ".join("%d: %s" % (t, " ".join("%.1f" % v for v in vs)) for t, vs in res.items()), flush=True)
