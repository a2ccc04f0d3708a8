"""Stand-in for the four kornia==0.5.0 functions the reference's hot path calls.

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py in the build container to run the
reference's own files; never imported by the product).  kornia is pinned by the reference at
requirements.txt:1 but is not installable offline, so these are restatements of its published
algorithm.  `torch.solve` (removed from torch>=2) is replaced by
`torch.linalg.solve`.  Call sites in the reference: src/heads/ransac_utils.py:72,90,143,
src/heads/PerceptualHead.py:175,201,765, src/data/utils.py:24,59.
"""
import torch
import torch.nn.functional as F


def convert_points_to_homogeneous(p):
    return F.pad(p, [0, 1], "constant", 1.0)


def convert_points_from_homogeneous(p, eps=1e-8):
    z = p[..., -1:]
    mask = torch.abs(z) > eps
    scale = torch.ones_like(z).masked_scatter_(mask, 1.0 / z[mask])
    return scale * p[..., :-1]


def transform_points(trans_01, points_1):
    # [B,3,3] x [B,N,2]  (also [B,1,1,3,3] x [B,H,W,2] inside warp_perspective)
    # (fp64 golden runs only: the reference builds its grids with .float(); promote so that the
    #  whole chain is double there.  No effect in the fp32 runs.)
    points_1_h = convert_points_to_homogeneous(points_1.to(trans_01.dtype))
    points_0_h = torch.matmul(trans_01.unsqueeze(-3) if trans_01.dim() == points_1.dim() else trans_01,
                              points_1_h.unsqueeze(-1)).squeeze(-1)
    return convert_points_from_homogeneous(points_0_h)


def get_perspective_transform(src, dst):
    # 8x8 system, two rows per correspondence, solve with LU (partial pivoting); H22 == 1
    B = src.shape[0]
    x, y = src[..., 0], src[..., 1]
    u, v = dst[..., 0], dst[..., 1]
    zeros, ones = torch.zeros_like(x), torch.ones_like(x)
    ax = torch.stack([x, y, ones, zeros, zeros, zeros, -x * u, -y * u], dim=-1)
    ay = torch.stack([zeros, zeros, zeros, x, y, ones, -x * v, -y * v], dim=-1)
    A = torch.stack([ax, ay], dim=2).reshape(B, 8, 8)       # rows interleaved per point
    b = torch.stack([u, v], dim=2).reshape(B, 8, 1)
    X = torch.linalg.solve(A, b)
    M = torch.ones(B, 9, device=src.device, dtype=src.dtype)
    M[..., :8] = X.squeeze(-1)
    return M.view(-1, 3, 3)


def _normal_transform_pixel(height, width, device, dtype):
    tr = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], device=device, dtype=dtype)
    tr[0, 0] = tr[0, 0] * 2.0 / (width - 1.0)
    tr[1, 1] = tr[1, 1] * 2.0 / (height - 1.0)
    return tr.unsqueeze(0)


def warp_perspective(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True):
    B, C, H, W = src.shape
    h_out, w_out = dsize
    src_norm = _normal_transform_pixel(H, W, src.device, src.dtype)
    dst_norm = _normal_transform_pixel(h_out, w_out, src.device, src.dtype)
    dst_norm_trans_src_norm = dst_norm @ (M @ torch.inverse(src_norm))
    src_norm_trans_dst_norm = torch.inverse(dst_norm_trans_src_norm)
    xs = torch.linspace(-1, 1, w_out, device=src.device, dtype=src.dtype)
    ys = torch.linspace(-1, 1, h_out, device=src.device, dtype=src.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    base = torch.stack([gx, gy], dim=-1).unsqueeze(0).repeat(B, 1, 1, 1)          # [B,h,w,2] (x,y)
    grid = transform_points(src_norm_trans_dst_norm[:, None, None], base)
    return F.grid_sample(src, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def _normalize_points(points, eps=1e-8):
    x_mean = torch.mean(points, dim=1, keepdim=True)
    scale = (points - x_mean).norm(dim=-1).mean(dim=-1)
    scale = torch.sqrt(torch.tensor(2.0, device=points.device, dtype=points.dtype)) / (scale + eps)
    ones, zeros = torch.ones_like(scale), torch.zeros_like(scale)
    transform = torch.stack([scale, zeros, -scale * x_mean[..., 0, 0],
                             zeros, scale, -scale * x_mean[..., 0, 1],
                             zeros, zeros, ones], dim=-1).view(-1, 3, 3)
    return transform_points(transform, points), transform


def find_homography_dlt(points1, points2, weights=None):
    eps = 1e-8
    points1 = points1.to(points2.dtype)
    p1n, t1 = _normalize_points(points1)
    p2n, t2 = _normalize_points(points2)
    x1, y1 = torch.chunk(p1n, 2, dim=-1)
    x2, y2 = torch.chunk(p2n, 2, dim=-1)
    ones, zeros = torch.ones_like(x1), torch.zeros_like(x1)
    ax = torch.cat([zeros, zeros, zeros, -x1, -y1, -ones, y2 * x1, y2 * y1, y2], dim=-1)
    ay = torch.cat([x1, y1, ones, zeros, zeros, zeros, -x2 * x1, -x2 * y1, -x2], dim=-1)
    A = torch.cat((ax, ay), dim=-1).reshape(ax.shape[0], -1, ax.shape[-1])
    if weights is None:
        A = A.transpose(-2, -1) @ A
    else:
        w_diag = torch.diag_embed(weights.unsqueeze(dim=-1).repeat(1, 1, 2).reshape(weights.shape[0], -1))
        A = A.transpose(-2, -1) @ w_diag @ A
    _, _, Vh = torch.linalg.svd(A)
    V = Vh.transpose(-2, -1)
    H = V[..., -1].view(-1, 3, 3)
    H = torch.inverse(t2) @ (H @ t1)
    return H / (H[..., -1:, -1:] + eps)
