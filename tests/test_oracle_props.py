"""Oracle-free properties that pin the restated kornia 0.5.0 arithmetic (SURVEY.md 4 i-iv): the
reference holds no golden vectors of its own for that third-party boundary, so conventions
(corner placement at 0/W, align_corners=True pixel-centre sampling, DLT normalisation) are pinned by
identities that need no reference implementation."""
import numpy as np
import torch

from bihome_amd import synth
from oracle import bihome_oracle as O

C = torch.tensor([[0, 0], [128, 0], [128, 128], [0, 128]], dtype=torch.float64)


def test_four_point_homography_maps_corners():
    g = torch.Generator().manual_seed(0)
    delta = (torch.rand(16, 4, 2, generator=g, dtype=torch.float64) - 0.5) * 64
    H = O.four_point_to_homography(C.repeat(16, 1, 1), delta)
    assert torch.allclose(H[:, 2, 2], torch.ones(16, dtype=torch.float64))
    mapped = O.transform_points(H, C.repeat(16, 1, 1))
    assert (mapped - (C + delta)).abs().max() < 1e-9


def test_dlt_recovers_exact_field():
    d = synth.make_pairs(4, seed=2)
    pf = np.stack([synth.perspective_field(synth.four_point_homography(C.numpy(), C.numpy() + d["delta"][b]))
                   for b in range(4)])
    head = O.BiHomEHead(torch.nn.Identity(), PATCH_SIZE=128, PATCH_KEYS=["patch_1", "patch_2"], DELTA_HAT_KEYS=[],
                        PF_KEYS=["a", "b"], RANSAC_HYPOTHESIS_NO=1, POINTS_PER_HYPOTHESIS=128, TRIPLET_LOSS="double-line",
                        TRIPLET_DISTANCE="l1", TRIPLET_AGGREGATION="channel-agnostic", TRIPLET_MARGIN="inf",
                        MASK_KEYS=[], TRIPLET_MU=0.01).double()
    choice = O.sample_choice(128 * 128, 4 * 128, torch.Generator().manual_seed(1)).reshape(4, 128)
    assert choice.min() >= 1                       # weights arange(N): index 0 has probability 0
    dh, H, _ = head._delta_from_pf(torch.tensor(pf, dtype=torch.float64), choice)
    assert (dh.reshape(4, 4, 2) - torch.tensor(d["delta"], dtype=torch.float64)).abs().max() < 1e-7


def test_warp_identity_and_alignment_convention():
    d = synth.make_pairs(3, seed=4)
    p1 = torch.tensor(d["patch_1"], dtype=torch.float64)
    p2 = torch.tensor(d["patch_2"], dtype=torch.float64)
    eye = torch.eye(3, dtype=torch.float64).repeat(3, 1, 1)
    assert (O.warp_image(p1, eye) - p1).abs().max() < 1e-12
    # delta_hat == delta_gt must map patch_1 onto patch_2 (same pixel-centre convention as the data
    # generator); a half-pixel or align_corners=False mismatch would leave a systematic residual
    H = O.four_point_to_homography(C.repeat(3, 1, 1), torch.tensor(d["delta"], dtype=torch.float64))
    w = O.warp_image(p1, H)
    m = O.warp_image(torch.ones_like(p1), H) == 1
    assert ((w - p2).abs() * m).sum() / m.sum() < 0.02
    # and the wrong convention is measurably worse (shift the sampling grid by half a pixel)
    T = torch.tensor([[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]], dtype=torch.float64)
    w_bad = O.warp_image(p1, H @ T)
    assert ((w_bad - p2).abs() * m).sum() / m.sum() > 2 * ((w - p2).abs() * m).sum() / m.sum()


def test_mace_definition():
    a = torch.zeros(2, 4, 2)
    b = torch.zeros(2, 4, 2)
    b[..., 0] = 3
    b[..., 1] = 4
    assert abs(O.mace(a, b) - 5.0) < 1e-6
