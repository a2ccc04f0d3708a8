"""Five launches of the f32x3 wgrad kernel on the bench shapes, for `rocprofv3 --pmc ... -- python3 tools/wx3_pmc_run.py`."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 64, 64, 64)]:
    x = torch.randn(N, H, H, Ci, device="cuda"); gy = torch.randn(N, H, H, Co, device="cuda")
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=2)
    gw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    ws = torch.empty(K.wgrad_det_bytes(d) // 4, dtype=torch.float32, device="cuda")
    for _ in range(5):
        K.conv_wgrad(x, gy, gw, None, d, det_ws=ws)
    torch.cuda.synchronize()
