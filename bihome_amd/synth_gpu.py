"""Device-side synthetic COCO-style pair generator.

Same sample distribution as the host generator `bihome_amd.synth.make_pairs` (which mirrors
`HomographyNetPrep`, src/data/transforms.py:441-725), but the crops/warps run in one HIP kernel
(`bh_synth_pairs`) over base images that stay resident in HBM, so a training loop never waits for CPU workers."""
import ctypes

import numpy as np
import torch

from . import kernels as K
from . import synth
from ._lib import check, lib


class GpuPairGenerator:

    def __init__(self, n_images=16, patch=128, rho=32, seed=42, photometric_max_delta=0, device="cuda"):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.h = max(240, patch + 2 * rho + 48)
        self.w = max(320, patch + 2 * rho + 128)
        imgs = np.stack([synth.texture_image(rng, self.h, self.w).transpose(2, 0, 1) for _ in range(n_images)])
        self.images = torch.tensor(imgs, dtype=torch.float32, device=device).contiguous()      # [NI,3,H,W] 0..255
        self.patch, self.rho, self.pmd = patch, rho, photometric_max_delta
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self._rs = np.random.RandomState(seed)            # photometric decisions (host draws, reference order)
        self.device = device

    def draw(self, B):
        """Random sample parameters exactly as transforms.py:505-506,538 draw them (uniform integer position with a
        rho margin, integer corner offsets in [-rho, rho-1])."""
        g, dev, half = self.gen, self.device, self.patch // 2
        idx = torch.randint(0, self.images.shape[0], (B,), generator=g, device=dev, dtype=torch.int32)
        px = torch.randint(self.rho + half, self.w - self.rho - half + 1, (B,), generator=g, device=dev)
        py = torch.randint(self.rho + half, self.h - self.rho - half + 1, (B,), generator=g, device=dev)
        origin = torch.stack([px - half, py - half], 1).to(torch.float32).contiguous()
        delta = torch.randint(-self.rho, self.rho, (B, 4, 2), generator=g, device=dev).to(torch.float32).contiguous()
        photo = None
        if self.pmd > 0:
            # PhotometricDistortSimple's decisions (transforms.py:296-330) for both images of every pair: the same record
            # layout and draw order as the host generator (synth.draw_photometric), drawn on the host - 12 floats per pair
            recs = np.stack([np.concatenate([synth.draw_photometric(self._rs, self.pmd), synth.draw_photometric(self._rs, self.pmd)])
                             for _ in range(B)])
            photo = torch.tensor(recs, dtype=torch.float32, device=dev).contiguous()
        return idx, origin, delta, photo

    def make(self, idx, origin, delta, photo=None):
        B, P = delta.shape[0], self.patch
        H64, _ = K.h4pt_fwd(delta, P)
        p1 = torch.empty(B, 1, P, P, dtype=torch.float32, device=self.device)
        p2 = torch.empty_like(p1)
        pv = ctypes.c_void_p
        check(lib.bh_synth_pairs(pv(self.images.data_ptr()), pv(idx.data_ptr()), pv(origin.data_ptr()), pv(H64.data_ptr()),
                                 pv(photo.data_ptr()) if photo is not None else None, B, self.images.shape[0], self.h,
                                 self.w, P, 0.443, 0.129, pv(p1.data_ptr()), pv(p2.data_ptr()),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "bh_synth_pairs")
        return {"patch_1": p1, "patch_2": p2, "delta": delta}

    def next(self, B):
        return self.make(*self.draw(B))
