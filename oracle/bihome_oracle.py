"""CPU oracle: a plain-PyTorch restatement of the reference's biHomE training hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg as the checker / CPU baseline; the product (bihome_amd/, src/) never imports it.

Parity status: PINNED against tests/golden/*.npz, which were produced by the reference's own files
run in the build container (oracle/make_golden.py).  The third-party arithmetic (kornia 0.5.0,
torchvision) is not vendored in the reference and not installable offline, so those four kornia
functions are restated from the published algorithm here and in the stand-in used to make the
golden vectors: that boundary is pinned only by oracle-free properties (tests/test_oracle_props.py).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Works in float32 and float64 (`.double()`), the latter for finite-difference gradient checks.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# kornia 0.5.0 arithmetic, restated
# --------------------------------------------------------------------------------------------

def transform_points(T, p):
    """q = (T [x y 1]^T)[:2] / z, z guarded at 1e-8.  Call sites: ransac_utils.py:90,
    PerceptualHead.py:175,201,765."""
    ph = torch.cat([p, torch.ones_like(p[..., :1])], -1).to(T.dtype)
    q = torch.einsum("...ij,...nj->...ni", T, ph)
    z = q[..., 2:3]
    scale = torch.where(z.abs() > 1e-8, 1.0 / torch.where(z.abs() > 1e-8, z, torch.ones_like(z)), torch.ones_like(z))
    return q[..., :2] * scale


def four_point_to_homography(corners, deltas):
    """src/data/utils.py:7-33 torch branch (crop=False) -> kornia.get_perspective_transform:
    8x8 system, rows [x y 1 0 0 0 -xu -yu] / [0 0 0 x y 1 -xv -yv], LU solve, H22 = 1."""
    dst = corners + deltas
    x, y, u, v = corners[..., 0], corners[..., 1], dst[..., 0], dst[..., 1]
    o, z = torch.ones_like(x), torch.zeros_like(x)
    rx = torch.stack([x, y, o, z, z, z, -x * u, -y * u], -1)
    ry = torch.stack([z, z, z, x, y, o, -x * v, -y * v], -1)
    A = torch.stack([rx, ry], 2).reshape(-1, 8, 8)
    b = torch.stack([u, v], 2).reshape(-1, 8, 1)
    h = torch.linalg.solve(A, b).squeeze(-1)
    return torch.cat([h, torch.ones_like(h[:, :1])], 1).reshape(-1, 3, 3)


def image_shape_to_corners(patch):
    """src/data/utils.py:36-51: [[0,0],[W,0],[W,H],[0,H]] with W,H the patch size (not size-1)."""
    hh, ww = patch.shape[-2], patch.shape[-1]
    c = torch.tensor([[0, 0], [hh, 0], [hh, ww], [0, ww]], dtype=patch.dtype, device=patch.device)
    return c.repeat(patch.shape[0], 1, 1)


def _pixel_normaliser(h, w, like):
    return torch.tensor([[2.0 / (w - 1), 0, -1], [0, 2.0 / (h - 1), -1], [0, 0, 1]], dtype=like.dtype)


def warp_image(image, H):
    """src/data/utils.py:54-59: torch.inverse(H) then kornia.warp_perspective(img, H^-1, (h, w)):
    normalise to [-1,1] with (W-1,H-1), invert again, transform the base grid, grid_sample
    (bilinear, zeros, align_corners=True).  Net effect out(x) = bilinear img(H x)."""
    B, C, h, w = image.shape
    M = torch.inverse(H)
    N = _pixel_normaliser(h, w, image)
    G = torch.inverse(N @ (M @ torch.inverse(N)))
    ys = torch.linspace(-1, 1, h, dtype=image.dtype)
    xs = torch.linspace(-1, 1, w, dtype=image.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    base = torch.stack([gx, gy], -1).reshape(1, h * w, 2).expand(B, -1, -1)
    grid = transform_points(G, base).reshape(B, h, w, 2)
    return F.grid_sample(image, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def find_homography_dlt(p1, p2):
    """kornia.find_homography_dlt as called at ransac_utils.py:72 (no weights): Hartley
    normalisation, A[2N,9], A^T A, svd, last right-singular vector, T2^-1 H T1, / (H22 + 1e-8)."""
    def hartley(p):
        m = p.mean(1, keepdim=True)
        s = np.sqrt(2.0) / ((p - m).norm(dim=-1).mean(-1) + 1e-8)
        o, z = torch.ones_like(s), torch.zeros_like(s)
        T = torch.stack([s, z, -s * m[:, 0, 0], z, s, -s * m[:, 0, 1], z, z, o], -1).reshape(-1, 3, 3)
        return transform_points(T, p), T
    p1 = p1.to(p2.dtype)
    q1, T1 = hartley(p1)
    q2, T2 = hartley(p2)
    x1, y1, x2, y2 = q1[..., 0:1], q1[..., 1:2], q2[..., 0:1], q2[..., 1:2]
    o, z = torch.ones_like(x1), torch.zeros_like(x1)
    ax = torch.cat([z, z, z, -x1, -y1, -o, y2 * x1, y2 * y1, y2], -1)
    ay = torch.cat([x1, y1, o, z, z, z, -x2 * x1, -x2 * y1, -x2], -1)
    A = torch.stack([ax, ay], 2).reshape(p1.shape[0], -1, 9)
    AtA = A.transpose(1, 2) @ A
    V = torch.linalg.svd(AtA)[2].transpose(1, 2)
    H = torch.inverse(T2) @ (V[..., -1].reshape(-1, 3, 3) @ T1)
    return H / (H[:, 2:3, 2:3] + 1e-8)


# --------------------------------------------------------------------------------------------
# Backbones
# --------------------------------------------------------------------------------------------

def _cbr(cin, cout, k, s, p, bias=False):
    return [nn.Conv2d(cin, cout, k, s, p, bias=bias), nn.BatchNorm2d(cout), nn.ReLU()]


class _Res(nn.Module):
    """upper_branch/lower_branch residual unit, ReLU(upper + lower); lower is identity when absent
    (src/backbones/utils.py:85-131 ResNet34ConvBlock / ResNet34IdentityBlock, :60-82 ResNet50DeconvBlock)."""

    def __init__(self, upper, lower=None):
        super().__init__()
        self.upper_branch = nn.Sequential(*upper)
        if lower is not None:
            self.lower_branch = nn.Sequential(*lower)
        self._has_lower = lower is not None

    def forward(self, x):
        return F.relu(self.upper_branch(x) + (self.lower_branch(x) if self._has_lower else x))


def res34(cin, cout, stride=1):
    upper = _cbr(cin, cout, 3, stride, 1) + [nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)]
    lower = None if cin == cout else [nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout)]
    return _Res(upper, lower)


def deconv50(c):
    upper = [nn.ConvTranspose2d(c, c, 2, 2, 0)] + _cbr(c, c, 3, 1, 1) + \
            [nn.Conv2d(c, c // 2, 1, 1, 0, bias=False), nn.BatchNorm2d(c // 2)]
    lower = [nn.ConvTranspose2d(c, c // 2, 2, 2, 0, bias=False), nn.BatchNorm2d(c // 2)]
    return _Res(upper, lower)


class ZengBackbone(nn.Module):
    """src/backbones/Rethinking.py:13-156 (RESNET_BLOCK='ResNet34') and :284-316."""

    def __init__(self, **kw):
        super().__init__()
        self.patch_keys, self.target_keys = kw["PATCH_KEYS"], kw["TARGET_KEYS"]
        self.variant = str.lower(kw.get("VARIANT", "oneline"))
        assert kw["RESNET_BLOCK"] == "ResNet34"
        S = nn.Sequential
        self.layer1 = S(*_cbr(2, 64, 7, 2, 3), nn.MaxPool2d(3, 2, 1))                      # :31-35
        self.layer2 = S(res34(64, 64), res34(64, 64), res34(64, 64))                       # :46-48
        self.layer3 = S(res34(64, 128, 2), *[res34(128, 128) for _ in range(3)])           # :62-65
        self.layer4 = S(res34(128, 256, 2), *[res34(256, 256) for _ in range(5)], deconv50(256))   # :82-88
        self.layer5 = S(*[res34(128, 128) for _ in range(3)], deconv50(128))               # :102-105
        self.layer6 = S(res34(64, 64), res34(64, 64), deconv50(64))                        # :118-120
        self.layer7 = S(res34(32, 32), deconv50(32))                                       # :132-133
        self.layer8 = S(nn.Conv2d(16, 128, 1), nn.BatchNorm2d(128), nn.ReLU(), nn.Conv2d(128, 2, 1))  # :145-147

    def _forward(self, x):                                                                 # :284-294
        for i in range(1, 9):
            x = getattr(self, "layer%d" % i)(x)
        return x

    def forward(self, data):                                                               # :296-313
        p1, p2 = data[self.patch_keys[0]], data[self.patch_keys[1]]
        data[self.target_keys[0]] = self._forward(torch.cat([p1, p2], 1))
        if self.variant == "doubleline":
            data[self.target_keys[1]] = self._forward(torch.cat([p2, p1], 1))
        return data

    predict_homography = forward                                                           # :315-316


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout)
        self.conv2, self.bn2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _TVResNet34(nn.Module):
    """torchvision resnet34 layout (module names conv1,bn1,layer1-4,fc) - third-party, restated."""

    def __init__(self, in_ch=3, num_out=1000, depth=(3, 4, 6, 3)):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(in_ch, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        cin = 64
        for i, (n, c) in enumerate(zip(depth, (64, 128, 256, 512))):
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(cin, c, 2 if (j == 0 and i > 0) else 1))
                cin = c
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        self.fc = nn.Linear(512, num_out)

    def stem(self, x):
        return F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.stem(x)))))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class ResNet34Backbone(nn.Module):
    """src/backbones/ResNet34.py:6-50: resnet34 with 2-channel conv1 (:17) and fc 512->8 (:19)."""

    def __init__(self, **kw):
        super().__init__()
        self.patch_keys, self.target_keys = kw["PATCH_KEYS"], kw["TARGET_KEYS"]
        self.variant = str.lower(kw.get("VARIANT", "oneline"))
        self.resnet34 = _TVResNet34(in_ch=2, num_out=8)

    def forward(self, data):
        p1, p2 = data[self.patch_keys[0]], data[self.patch_keys[1]]
        data[self.target_keys[0]] = self.resnet34(torch.cat([p1, p2], 1)).reshape(-1, 4, 2)    # :28,:41
        if self.variant == "doubleline":
            data[self.target_keys[1]] = self.resnet34(torch.cat([p2, p1], 1)).reshape(-1, 4, 2)  # :45
        return data

    predict_homography = forward


# --------------------------------------------------------------------------------------------
# Head
# --------------------------------------------------------------------------------------------

class AuxiliaryResnet(nn.Module):
    """PerceptualHead.py:15-76 with AUXILIARY_RESNET='resnet34', frozen: gray->3ch repeat (:52-53), conv1, bn1, relu,
    maxpool, layer1 and - for AUXILIARY_RESNET_OUTPUT_LAYER 2/3/4 - layer2/3/4 (:62-67; the unused layers are Identity
    upstream, so they own no state-dict keys).  Weights frozen (:36-39) but the BatchNorms still follow
    module.train()/eval() (batch statistics in training)."""

    def __init__(self, output_layer=1):
        super().__init__()
        full = _TVResNet34()
        self.output_layer = int(output_layer)
        self.resnet = nn.Module()
        self.resnet.conv1, self.resnet.bn1, self.resnet.layer1 = full.conv1, full.bn1, full.layer1
        for i in range(2, self.output_layer + 1):
            setattr(self.resnet, "layer%d" % i, getattr(full, "layer%d" % i))
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        if x.shape[1] == 1:
            x = x.repeat(1, 3, 1, 1)
        r = self.resnet
        x = r.layer1(F.max_pool2d(F.relu(r.bn1(r.conv1(x))), 3, 2, 1))
        for i in range(2, self.output_layer + 1):
            x = getattr(r, "layer%d" % i)(x)
        return x


def sample_choice(n_points, count, generator=None):
    """ransac_utils.py:54-56: torch.multinomial with weights arange(N) (p(i) ~ i; index 0 never drawn)."""
    w = torch.arange(0, n_points, dtype=torch.float32)
    return torch.multinomial(w, count, replacement=True, generator=generator)


class BiHomEHead(nn.Module):
    """src/heads/PerceptualHead.py:79-767, restricted to the branch the shipped biHomE configs select:
    TRIPLET_LOSS='double-line', TRIPLET_DISTANCE='l1', TRIPLET_AGGREGATION='channel-agnostic',
    TRIPLET_MARGIN=str ('inf'), SAMPLING_STRATEGY='downsample-mask', MASK_KEYS=[]."""

    def __init__(self, backbone, **kw):
        super().__init__()
        self.backbone = backbone                                                          # :83
        self.patch_size, self.patch_keys = kw["PATCH_SIZE"], kw["PATCH_KEYS"]
        self.delta_hat_keys = kw["DELTA_HAT_KEYS"]
        if len(self.delta_hat_keys):
            self.hypothesis_no = 1                                                        # :92-93
        else:
            self.pf_keys = kw["PF_KEYS"]
            self.hypothesis_no = kw["RANSAC_HYPOTHESIS_NO"]
            self.points_per_hypothesis = kw["POINTS_PER_HYPOTHESIS"]
        self.one_line = "one-line" in kw["TRIPLET_LOSS"]                                 # iHomE (:465-538)
        self.multihead = kw["TRIPLET_LOSS"] == ""                                        # :108,:230-235 -> multihead_resnet_loss
        if not self.multihead:
            assert kw["TRIPLET_DISTANCE"] == "l1" and not len(kw["MASK_KEYS"]) and not kw.get("MASK_CRD", False)
            if self.one_line:
                assert isinstance(kw["TRIPLET_MARGIN"], (int, float))
            else:
                assert "double-line" in kw["TRIPLET_LOSS"]
                assert kw["TRIPLET_AGGREGATION"] == "channel-agnostic" and isinstance(kw["TRIPLET_MARGIN"], str)
        self.triplet_margin = kw.get("TRIPLET_MARGIN")
        self.triplet_mu = kw.get("TRIPLET_MU", 0.0)
        self.auxiliary_resnet = AuxiliaryResnet(kw.get("AUXILIARY_RESNET_OUTPUT_LAYER", 1))
        self.last = {}

    # -- DSAC ---------------------------------------------------------------------------------
    def _fields(self, pf):
        """forward_map_field :125-146: coord grid (x,y) row-major idx = y*W + x; map = coord + pf."""
        B, _, h, w = pf.shape
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        coord = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).to(torch.float32).unsqueeze(0).expand(B, -1, -1)
        four = torch.tensor([[0, 0], [w, 0], [w, h], [0, h]], dtype=torch.float32)
        return coord, coord + pf.reshape(B, 2, -1).permute(0, 2, 1), four

    def dsac(self, pf, choice=None):
        """ransac_utils.py:47-74 (sample + DLT) and :76-128 (repr_error scoring, softmax(-err)).
        `choice` int64 [B, n*P]: sampled indices; drawn like the reference when None."""
        B = pf.shape[0]
        n, P = self.hypothesis_no, self.points_per_hypothesis
        coord, mapf, four = self._fields(pf)
        if choice is None:
            choice = sample_choice(coord.shape[1], B * P * n).reshape(B, -1)
        idx = choice.reshape(B, -1, 1).repeat(1, 1, 2)
        p1 = torch.gather(coord, 1, idx).reshape(B * n, P, 2)
        p2 = torch.gather(mapf, 1, idx).reshape(B * n, P, 2)
        H = find_homography_dlt(p1, p2).reshape(B, n, 3, 3)
        err = torch.stack([(transform_points(H[:, j], coord) - mapf).abs().sum(-1).sum(-1) for j in range(n)], 1)
        scores = torch.softmax(-err, -1)
        return H, scores, err, four

    def _delta_from_pf(self, pf, choice):
        H, scores, err, four = self.dsac(pf, choice)
        B, n = H.shape[:2]
        fp = four.unsqueeze(0).expand(B * n, -1, -1)
        dh = (transform_points(H.reshape(-1, 3, 3), fp) - fp).reshape(B, n, 4, 2)         # :175-178
        return dh, H, scores

    # -- loss -----------------------------------------------------------------------------------
    @staticmethod
    def _warp(image, delta_hat):                                                          # :237-243
        H = four_point_to_homography(image_shape_to_corners(image), delta_hat)
        return warp_image(image, H), H

    def forward(self, data, choice_12=None, choice_21=None):                              # :148-235
        if self.one_line or self.multihead:
            if not len(self.delta_hat_keys):
                d12, H12, scores = self._delta_from_pf(data[self.pf_keys[0]], choice_12)
                self.last.update(H_dlt_12=H12)
            else:
                d12, scores = data[self.delta_hat_keys[0]], None
            return self.multihead_loss(data, d12, scores) if self.multihead else self.one_line_loss(data, d12, scores)
        if not len(self.delta_hat_keys):
            d12, H12, _ = self._delta_from_pf(data[self.pf_keys[0]], choice_12)
            d21, H21, _ = self._delta_from_pf(data[self.pf_keys[1]], choice_21)
            self.last.update(H_dlt_12=H12, H_dlt_21=H21)
        else:
            d12, d21 = data[self.delta_hat_keys[0]], data[self.delta_hat_keys[1]]        # :211-218
        return self.triplet_loss(data, d12, d21)

    def triplet_loss(self, data, d12, d21):                                               # :320-714
        p1, p2 = data[self.patch_keys[0]], data[self.patch_keys[1]]
        B, i = d12.shape[0], self.patch_size
        aux = self.auxiliary_resnet
        f1 = aux(p1)                                                                      # :358
        f2 = aux(p2)                                                                      # :367
        d12 = d12.reshape(B, 4, 2)
        p1w, h1 = self._warp(p1, d12)                                                     # :371
        f1w = aux(p1w)                                                                    # :377
        m1w, _ = self._warp(torch.ones_like(p1), d12)                                     # :382
        d21 = d21.reshape(B, 4, 2)
        p2w, h2 = self._warp(p2, d21)                                                     # :392
        f2w = aux(p2w)                                                                    # :398
        m2w, _ = self._warp(torch.ones_like(p2), d21)                                     # :401
        k = i // f1w.shape[-1]                                                            # :450
        m1w, m2w = F.avg_pool2d(m1w, k).squeeze(1), F.avg_pool2d(m2w, k).squeeze(1)       # :451-459
        m1 = F.avg_pool2d(torch.ones_like(p1), k).squeeze(1)
        m2 = F.avg_pool2d(torch.ones_like(p2), k).squeeze(1)
        l1, l2, l3 = (f1w - f2).abs(), (f2w - f1).abs(), (f1 - f2).abs()                  # :559-561
        den1 = (m1w * m2).sum((-1, -2))                                                   # :616
        M1 = l1.sum(1) - l3.sum(1)                                                        # :621
        ln1 = (m1w * m2 * M1).sum((-1, -2)) / torch.max(den1, torch.ones_like(den1))      # :631-632
        den2 = (m2w * m1).sum((-1, -2))                                                   # :635
        M2 = l2.sum(1) - l3.sum(1)                                                        # :640
        ln2 = (m2w * m1 * M2).sum((-1, -2)) / torch.max(den2, torch.ones_like(den2))      # :652-653
        eye = torch.eye(3, dtype=h1.dtype).unsqueeze(0)
        ln3 = ((h1 @ h2 - eye) ** 2).sum()                                                # :660-662
        loss = ln1.sum() + ln2.sum() + self.triplet_mu * ln3                              # :656-665
        if "summary_writer" in data:                                                      # :678-697 (log steps)
            sw, step = data["summary_writer"], data["summary_writer_step"]
            sw.add_scalars("feature_space", {"patch_1_f": f1.mean().item()}, step)
            sw.add_scalars("feature_space", {"patch_2_f": f2.mean().item()}, step)
            sw.add_scalars("feature_space", {"patch_1_f_prime": f1w.mean().item()}, step)
            sw.add_scalars("loss_comp", {"l1": (f2 - f1w).abs().mean().item()}, step)
            sw.add_scalars("loss_comp", {"l3": (f2 - f1).abs().mean().item()}, step)
            sw.add_scalars("h", {"h1": ((h1 - eye) ** 2).sum().item()}, step)
            sw.add_scalars("loss_den", {"l1_den": den1.min().item()}, step)
            sw.add_scalars("loss_den", {"l2_den": den2.min().item()}, step)
        self.last.update(ln1=ln1.sum(), ln2=ln2.sum(), ln3=ln3, H_4pt_12=h1, H_4pt_21=h2, warp_12=p1w, warp_21=p2w,
                         mask_pooled_12=m1w, mask_pooled_21=m2w, f1=f1, f2=f2, f1w=f1w, f2w=f2w,
                         delta_hat_12=d12, delta_hat_21=d21)
        return loss, data.get("delta"), d12                                               # :703-714

    def one_line_loss(self, data, d12, scores=None):                                      # :320-538 ('one-line', 'l1')
        p1, p2 = data[self.patch_keys[0]], data[self.patch_keys[1]]
        B, n, i = d12.shape[0], self.hypothesis_no, self.patch_size
        if n > 1:                                                                         # :352,:361 (one copy per hypothesis)
            p1 = p1.reshape(B, 1, i, i).repeat(1, n, 1, 1).reshape(B * n, 1, i, i)
            p2 = p2.reshape(B, 1, i, i).repeat(1, n, 1, 1).reshape(B * n, 1, i, i)
        aux = self.auxiliary_resnet
        f1 = aux(p1)                                                                      # :358
        f2 = aux(p2)                                                                      # :367
        d12 = d12.reshape(B * n, 4, 2)
        p1w, h1 = self._warp(p1, d12)                                                     # :371
        f1w = aux(p1w)                                                                    # :377
        m1w, _ = self._warp(torch.ones_like(p1), d12)                                     # :382
        k = i // f1w.shape[-1]                                                            # :450
        m1w = F.avg_pool2d(m1w, k).squeeze(1)                                             # :451-452
        m2 = F.avg_pool2d(torch.ones_like(p2), k).squeeze(1)                              # :453
        l1 = (f1w - f2).abs().sum(1)                                                      # :481
        l3 = (f1 - f2).abs().sum(1)                                                       # :482
        loss_mat = torch.max(l1 - l3 + torch.ones_like(l1) * self.triplet_margin, torch.zeros_like(l1))   # :505
        if scores is not None:
            loss_mat = loss_mat * scores.reshape(B * n, 1, 1)                             # :508-511
        den = (m1w * m2).sum((-1, -2))                                                    # :523
        loss = ((m1w * m2 * loss_mat).sum((-1, -2)) / torch.max(den, torch.ones_like(den))).sum()   # :524-538
        if "summary_writer" in data:                                                      # :678-692
            sw, step = data["summary_writer"], data["summary_writer_step"]
            eye = torch.eye(3, dtype=h1.dtype).unsqueeze(0)
            sw.add_scalars("feature_space", {"patch_1_f": f1.mean().item()}, step)
            sw.add_scalars("feature_space", {"patch_2_f": f2.mean().item()}, step)
            sw.add_scalars("feature_space", {"patch_1_f_prime": f1w.mean().item()}, step)
            sw.add_scalars("loss_comp", {"l1": (f2 - f1w).abs().mean().item()}, step)
            sw.add_scalars("loss_comp", {"l3": (f2 - f1).abs().mean().item()}, step)
            sw.add_scalars("h", {"h1": ((h1 - eye) ** 2).sum().item()}, step)
        self.last.update(H_4pt_12=h1, warp_12=p1w, mask_pooled_12=m1w, f1=f1, f2=f2, f1w=f1w, delta_hat_12=d12)
        if scores is not None:
            d12 = (d12 * scores.reshape(B * n, 1, 1)).reshape(B, n, 4, 2).sum(1)          # :708-710
        return loss, data.get("delta"), d12

    def multihead_loss(self, data, d12, scores=None):                                     # :245-315 (TRIPLET_LOSS == '')
        """Returns (ground_truth, network_output, delta_gt, delta_hat) = (features of patch_2, features of the warped
        patch_1, ...): the driver applies a torch loss to the first two (train.py:318-322)."""
        p1, p2 = data[self.patch_keys[0]], data[self.patch_keys[1]]
        B, n, i = d12.shape[0], self.hypothesis_no, self.patch_size
        p1 = p1.reshape(B, 1, i, i).repeat(1, n, 1, 1).reshape(B * n, 1, i, i)           # :265
        p2 = p2.reshape(B, 1, i, i).repeat(1, n, 1, 1).reshape(B * n, 1, i, i)           # :268
        aux = self.auxiliary_resnet
        f2 = aux(p2)                                                                      # :269
        d12 = d12.reshape(B * n, 4, 2)
        p1w, h1 = self._warp(p1, d12)                                                     # :272
        f1w = aux(p1w)                                                                    # :273
        if scores is not None:                                                            # :276-280
            sc = scores.reshape(B * n, 1, 1, 1)
            f1w, f2 = f1w * sc, f2 * sc
        if "summary_writer" in data:                                                      # :286-298
            sw, step = data["summary_writer"], data["summary_writer_step"]
            eye = torch.eye(3, dtype=h1.dtype).unsqueeze(0)
            sw.add_scalars("feature_space", {"patch_2_f": f2.mean().item()}, step)
            sw.add_scalars("feature_space", {"patch_1_f_prime": f1w.mean().item()}, step)
            sw.add_scalars("loss_comp", {"l1": (f2 - f1w).abs().mean().item()}, step)
            sw.add_scalars("h", {"h1": ((h1 - eye) ** 2).sum().item()}, step)
        self.last.update(H_4pt_12=h1, warp_12=p1w, f2=f2, f1w=f1w)
        if scores is not None:                                                            # :309-312
            d12 = (d12 * scores.reshape(B * n, 1, 1)).reshape(B, n, 4, 2).sum(1)
        return f2, f1w, data.get("delta"), d12                                            # :315

    def predict_homography(self, data, choice=None):                                      # :716-767
        if len(self.delta_hat_keys):
            return data[self.delta_hat_keys[0]], None
        dh, H, scores = self._delta_from_pf(data[self.pf_keys[0]], choice)
        best = torch.argmax(scores, -1)                                                   # :755
        self.last.update(best=best, H=H, scores=scores)
        return dh[torch.arange(dh.shape[0]), best], None


# --------------------------------------------------------------------------------------------
# Step harness (train.py:296-387, 402-403)
# --------------------------------------------------------------------------------------------

class NoOpHead(nn.Module):
    """src/heads/NoOpHead.py:10-53: the head of the supervised "-orig" experiments; routes (ground_truth,
    network_output, delta_gt, delta_hat) to the torch loss of train.py:318-322.  'all_points': delta_hat is the
    predicted field at the four patch corners (:33-50)."""

    def __init__(self, backbone, **kw):
        super().__init__()
        self.target_gen, self.learning_keys = kw["TARGET_GEN"], kw["LEARNING_KEYS"]

    def forward(self, data, *unused):
        ret = [data[k] for k in self.learning_keys[:-1]]
        last = data[self.learning_keys[-1]]
        if self.target_gen == "4_points":
            ret.append(last)
        else:
            h, w = last.shape[-2:]
            d = torch.zeros((last.shape[0], 4, 2), dtype=last.dtype)
            for i, (yy, xx) in enumerate(((0, 0), (0, w - 1), (h - 1, w - 1), (h - 1, 0))):
                d[:, i, 0], d[:, i, 1] = last[:, 0, yy, xx], last[:, 1, yy, xx]
            ret.append(d)
        return ret


# --------------------------------------------------------------------------------------------
# Zhang et al. "Content-Aware" baseline (round 3): backbone src/backbones/ContentAware.py, head src/heads/TripletHead.py
# --------------------------------------------------------------------------------------------

def _cbr1(cin, cout, act=True):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU() if act else nn.Sigmoid())


class ZhangMaskPredictor(nn.Module):
    """ContentAware.py:6-52.  fix_mask (every shipped config): all ones."""

    def __init__(self, fix_mask=False, normalization_strength=-1):
        super().__init__()
        self.fix_mask, self.normalization_strength = fix_mask, normalization_strength
        self.layer1, self.layer2, self.layer3, self.layer4 = _cbr1(1, 4), _cbr1(4, 8), _cbr1(8, 16), _cbr1(16, 32)
        self.layer5 = _cbr1(32, 1, act=False)                                             # :24-26 (Sigmoid)

    def forward(self, x):
        if self.fix_mask:
            return torch.ones_like(x)                                                     # :38-39
        out = self.layer5(self.layer4(self.layer3(self.layer2(self.layer1(x)))))          # :41-45
        if self.normalization_strength > 0:                                               # :28-35,49-50
            mx = out.reshape(out.shape[0], -1).max(1)[0].reshape(-1, 1, 1, 1)
            out = torch.clamp(out / (mx * self.normalization_strength), 0, 1)
        return out


class ZhangFeatureExtractor(nn.Module):
    """ContentAware.py:55-74: 1 -> 4 -> 8 -> 1 channels, 3x3 conv + BatchNorm + ReLU each."""

    def __init__(self):
        super().__init__()
        self.layer1, self.layer2, self.layer3 = _cbr1(1, 4), _cbr1(4, 8), _cbr1(8, 1)

    def forward(self, x):
        return self.layer3(self.layer2(self.layer1(x)))


class ContentAwareBackbone(nn.Module):
    """ContentAware.py:84-179: G = mask * features of each patch, resnet34 (2-channel stem, 8-way fc) on cat(G1, G2)."""

    def __init__(self, **kw):
        super().__init__()
        self.patch_keys, self.mask_keys = kw["PATCH_KEYS"], kw["MASK_KEYS"]
        self.feature_keys, self.target_keys = kw["FEATURE_KEYS"], kw["TARGET_KEYS"]
        self.mask_predictor = ZhangMaskPredictor(kw["FIX_MASK"], kw.get("MASK_NORMALIZATION_STRENGTH", -1))   # :93-94
        self.feature_extractor = ZhangFeatureExtractor()
        self.variant = str.lower(kw["VARIANT"])
        self.resnet34 = _TVResNet34(in_ch=2, num_out=8)                                   # :106-110

    def _forward(self, x1, x2):                                                           # :124-144
        m1, f1 = self.mask_predictor(x1), self.feature_extractor(x1)
        m2, f2 = self.mask_predictor(x2), self.feature_extractor(x2)
        g1, g2 = m1 * f1, m2 * f2
        return m1, f1, m2, f2, g1, g2, self.resnet34(torch.cat([g1, g2], 1)).reshape(-1, 4, 2)

    def forward(self, data):                                                              # :146-173
        e1, e2 = self.patch_keys
        (data[self.mask_keys[0]], data[self.feature_keys[0]], data[self.mask_keys[1]], data[self.feature_keys[1]], g1, g2,
         data[self.target_keys[0]]) = self._forward(data[e1], data[e2])
        if self.variant == "doubleline":
            data[self.target_keys[1]] = self.resnet34(torch.cat([g2, g1], 1)).reshape(-1, 4, 2)
        return data

    def predict_homography(self, data):                                                   # :175-187
        e1, e2 = self.patch_keys
        data[self.mask_keys[0]], _, data[self.mask_keys[1]], _, _, _, data[self.target_keys[0]] = self._forward(data[e1], data[e2])
        return data


class ZhangTripletHead(nn.Module):
    """src/heads/TripletHead.py:9-212."""

    def __init__(self, backbone, **kw):
        super().__init__()
        self.backbone = backbone
        self.patch_keys, self.mask_keys = kw["PATCH_KEYS"], kw["MASK_KEYS"]
        self.feature_keys, self.target_keys = kw["FEATURE_KEYS"], kw["TARGET_KEYS"]
        self.mu = kw["MU"]
        self.variant = str.lower(kw["VARIANT"])
        self.triplet_margin, self.aggregation = kw["TRIPLET_MARGIN"], kw["TRIPLET_AGGREGATION"]
        self.last = {}

    @staticmethod
    def _warp(image, delta_hat):                                                          # :30-35
        H = four_point_to_homography(image_shape_to_corners(image), delta_hat)
        return warp_image(image, H), H

    def _line(self, fw, f_other, f_same, mw, m_other):                                    # :75-114 (and :124-147 mirrored)
        l1, l3 = torch.abs(fw - f_other), torch.abs(f_same - f_other)
        mo, mw = m_other.squeeze(1), mw.squeeze(1)
        den = (mw * mo).sum((-1, -2))
        if isinstance(self.triplet_margin, str):
            mat = (l1 - l3).sum(1) if self.aggregation == "channel-aware" else l1.sum(1) - l3.sum(1)
        elif self.aggregation == "channel-aware":
            mat = torch.clamp(l1 - l3 + self.triplet_margin, min=0).sum(1)
        else:
            # :104-105 (and :141-142) as written upstream: torch.max of the [B,h,w] channel sums with zeros_like(l1) = [B,1,h,w]
            # BROADCASTS to [B,B,h,w] (entry [i,j] = hinge of sample j), and so do the mask product, the spatial sums ([B,B]) and
            # the division by the [B] denominators: the final sum counts every sample B times.  A quirk of the numeric-margin /
            # channel-agnostic branch (the shipped zhang-orig config); kept, because results must equal the reference's.
            mat = torch.max(l1.sum(1) - l3.sum(1) + self.triplet_margin, torch.zeros_like(l1))
        return ((mw * mo * mat).sum((-1, -2)) / torch.max(den, torch.ones_like(den))).sum()

    def forward(self, data, *unused):                                                     # :37-199
        e1, e2 = self.patch_keys
        m1k, m2k = self.mask_keys
        f1k, f2k = self.feature_keys
        o1, o2 = (self.target_keys + [None])[:2]
        p1, p2, m1, m2, f1, f2 = data[e1], data[e2], data[m1k], data[m2k], data[f1k], data[f2k]
        p1w, _ = self._warp(p1, data[o1])
        f1w = self.backbone.feature_extractor(p1w)                                        # :59
        m1w, h1 = self._warp(m1, data[o1])
        ln1 = self._line(f1w, f2, f1, m1w, m2)
        loss = ln1
        self.last = {"ln1": ln1.detach(), "h1": h1.detach(), "f1w": f1w.detach()}
        if self.variant == "doubleline":
            p2w, _ = self._warp(p2, data[o2])
            f2w = self.backbone.feature_extractor(p2w)                                    # :68
            m2w, h2 = self._warp(m2, data[o2])
            ln2 = self._line(f2w, f1, f2, m2w, m1)
            eye = torch.eye(3, dtype=h1.dtype).unsqueeze(0)
            ln3 = ((torch.matmul(h1, h2) - eye) ** 2).sum()                               # :150-152
            loss = ln1 + ln2 + self.mu * ln3
            self.last.update(ln2=ln2.detach(), ln3=ln3.detach(), h2=h2.detach(), f2w=f2w.detach())
        return loss, data.get("delta"), data[o1]

    def predict_homography(self, data, *unused):                                          # :201-212
        delta_hat = data[self.target_keys[0]]
        _, H = self._warp(data[self.patch_keys[0]], delta_hat)
        return delta_hat, H


def mace(delta_gt, delta_hat):
    """train.py:402-403 / eval.py:133-134."""
    a = delta_gt.detach().cpu().numpy().reshape(-1, 2)
    b = delta_hat.detach().cpu().numpy().reshape(-1, 2)
    return float(np.mean(np.linalg.norm(a - b, axis=-1)))


def build(cfg, dtype=torch.float32, seed=0):
    """Backbone + head + Adam + MultiStepLR exactly as train.py:675-711 wires them, with the
    deterministic synthetic weights (weights are produced by the caller-supplied loader to keep
    this file free of product imports)."""
    bcfg, hcfg = cfg["MODEL"]["BACKBONE"], cfg["MODEL"]["HEAD"]
    bb = {"Rethinking": ZengBackbone, "ResNet34": ResNet34Backbone, "ContentAware": ContentAwareBackbone}[bcfg["NAME"]](**bcfg)
    head = {"NoOpHead": NoOpHead, "PerceptualHead": BiHomEHead, "TripletHead": ZhangTripletHead}[hcfg["NAME"]](bb, **hcfg)
    return bb, head


def make_optimizer(model, solver):
    opt = torch.optim.Adam(model.parameters(), lr=solver["LR"], betas=(solver["MOMENTUM_1"], solver["MOMENTUM_2"]),
                           weight_decay=0)                                                # train.py:703-707
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=solver["MILESTONES"], gamma=solver["LR_DECAY"])
    return opt, sched


def train_step(bb, head, opt, sched, data, choice_12=None, choice_21=None, clip=-1.0, loss_fn=None):
    """One iteration of train.py:296-387 (model.train(), zero_grad, forward, backward, clip, step).
    loss_fn: a torch.nn loss module for the supervised branch (train.py:318-322), None for the biHomE head loss."""
    bb.train(); head.train()
    opt.zero_grad()
    if loss_fn is not None:
        ground_truth, network_output, delta_gt, delta_hat = head(bb(data), choice_12, choice_21)
        loss = loss_fn(ground_truth, network_output)
    else:
        loss, delta_gt, delta_hat = head(bb(data), choice_12, choice_21)
    loss.backward()
    if clip > 0:
        torch.nn.utils.clip_grad_norm_(list(bb.parameters()), clip)
    opt.step()
    sched.step()
    return loss.detach(), delta_gt, delta_hat.detach()
