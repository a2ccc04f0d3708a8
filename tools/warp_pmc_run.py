"""The two homography-warp launches alone (forward + pooled coverage, adjoint) at BASELINE configs[1] (128 x 1 x 128 x 128) and
configs[4] (64 x 3 x 256 x 256), rotating over buffer sets larger than the Infinity Cache, for
`rocprofv3 --pmc ... -- python3 tools/warp_pmc_run.py` (counter passes) and `--kernel-trace --stats`."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
torch.manual_seed(0)
for (B, C, h, nset) in [(128, 1, 128, 40), (64, 3, 256, 8)]:
    delta = (torch.rand(B, 4, 2, device="cuda") - 0.5) * (h // 4)
    H64, _ = K.h4pt_fwd(delta, h)
    imgs = [torch.randn(B, C, h, h, device="cuda") for _ in range(nset)]
    gos = [torch.randn(B, C, h, h, device="cuda") for _ in range(nset)]
    gcs = [torch.randn(B, h // 4, h // 4, device="cuda") for _ in range(nset)]
    gH = torch.zeros(B, 9, dtype=torch.float64, device="cuda")
    for i in range(3 * nset):
        K.warp_fwd(imgs[i % nset], H64, 4)
        K.warp_bwd(imgs[i % nset], H64, gos[i % nset], gcs[i % nset], 4, gH=gH)
    torch.cuda.synchronize()
