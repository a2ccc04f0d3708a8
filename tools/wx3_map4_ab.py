"""Round 5: weight gradient of a 3x3 layer on 4 x 4 maps (ResNet-34 layer4) - fp32-input MFMA kernel (precision 0) against the fp16-piece
MAP4 form of wgrad_x3_kernel (precision 4)."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K, net


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ws = torch.empty(net.X3_WS_BYTES // 4, dtype=torch.float32, device='cuda')
N, Ci, Co = 128, 512, 512
x = torch.relu(torch.randn(N, 4, 4, Ci, device="cuda")); gy = torch.randn(N, 4, 4, Co, device="cuda")
K.amax_of(x); K.amax_of(gy)
gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
for prec in (0, 4, 0, 4):
    d = K.conv_desc(N, 4, 4, Ci, Co, 3, 1, 1, precision=prec)
    w_ = ws if K.wgrad_det_bytes(d) else None
    print(prec, K.conv_variant(d, "wgrad_det" if w_ is not None else "wgrad"), "%.1f us" % bench(lambda: K.conv_wgrad(x, gy, gw, None, d, det_ws=w_)), flush=True)
