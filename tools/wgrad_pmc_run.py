"""Workload for PMC passes over the wgrad kernels: python tools/wgrad_pmc_run.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
for (N, H, Ci, Co) in [(128, 32, 64, 64), (128, 16, 128, 128), (128, 8, 256, 256)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    x = torch.randn(N, H, H, Ci, device='cuda'); gy = torch.randn(N, H, H, Co, device='cuda'); gw = torch.zeros(Co, 3, 3, Ci, device='cuda')
    for _ in range(10): K.conv_wgrad(x, gy, gw, None, d)
    # the forward kernel on the same shape, for comparison
    w = torch.randn(Co, 3, 3, Ci, device='cuda')
    for _ in range(10): K.conv_fwd(x, w, None, d)
torch.cuda.synchronize()
