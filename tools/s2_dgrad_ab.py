"""Round 6: the stride-2 3x3 dgrad by parity class (bh_conv_dgrad_s2) as ONE launch against one launch per class, and the generic kernel's
three-bf16-piece form against the fp32-input MFMA, on the step's two downsampling layers (tuning build: hooks -49 / -48), rotating buffer
sets.  BIHOME_TUNING=1 python tools/s2_dgrad_ab.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
NSET = 4
def bench(fn, n=40):
    for i in range(4): fn(i % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i % NSET)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for (N, Hi, Ci, Co) in ((128, 32, 64, 128), (128, 16, 128, 256)):
    d = K.conv_desc(N, Hi, Hi, Ci, Co, 3, 2, 1, precision=4)
    gy = [torch.randn(N, Hi // 2, Hi // 2, Co, device='cuda') for _ in range(NSET)]
    x = [torch.randn(N, Hi, Hi, Ci, device='cuda') for _ in range(NSET)]
    w = torch.randn(Co, 3, 3, Ci, device='cuda') * 0.05
    for rnd in range(2):
        out = []
        for merge, x3 in ((1, 1), (0, 1), (1, 0), (0, 0)):
            lib.bh_debug_force_tile(-49, merge); lib.bh_debug_force_tile(-48, x3)
            out.append("dgrad %s, %s: %.1f us" % ("one launch" if merge else "four launches", "3 bf16 pieces" if x3 else "fp32 MFMA", bench(lambda i: K.conv_dgrad(gy[i], w, d))))
        for x3 in (1, 0):
            lib.bh_debug_force_tile(-48, x3)
            out.append("fwd %s: %.1f us" % ("3 bf16 pieces" if x3 else "fp32 MFMA", bench(lambda i: K.conv_fwd(x[i], w, None, d))))
        lib.bh_debug_force_tile(-49, 1); lib.bh_debug_force_tile(-48, 1)
        print("N%d %dx%d C%d->%d k3 s2 | " % (N, Hi, Hi, Ci, Co) + " | ".join(out), flush=True)
