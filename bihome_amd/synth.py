"""Synthetic COCO-style 128x128 patch pairs (host / numpy version).

Mirrors the distribution of the reference's data pipeline without COCO, cv2 or a network
(SURVEY.md §8(d)): `HomographyNetPrep` (src/data/transforms.py:441-725), `DictToGrayscale`
(:344-354), `DictStandardize` (:369-378) and, for pds-coco, the brightness/contrast part of
`PhotometricDistortSimple` (:296-330).  Base images stand in for the offline-preprocessed
240x320 COCO crops (src/data/coco/preprocess_offline.py:22).  Deterministic: numpy PCG64.
"""
import numpy as np


def _bilinear_resize(a, h, w):
    gh, gw = a.shape[:2]
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a00, a01 = a[y0][:, x0], a[y0][:, x0 + 1]
    a10, a11 = a[y0 + 1][:, x0], a[y0 + 1][:, x0 + 1]
    return (a00 * (1 - fy) * (1 - fx) + a01 * (1 - fy) * fx + a10 * fy * (1 - fx) + a11 * fy * fx)


def texture_image(rng, h=240, w=320, channels=3):
    """Smooth random RGB texture in [0,255]: octaves of low-pass noise (natural-image-like spectrum)."""
    img = np.zeros((h, w, channels), np.float64)
    amp = 1.0
    for cells in (3, 6, 12, 24, 48, 96):
        g = rng.standard_normal((cells * h // w + 2, cells + 2, channels))
        img += amp * _bilinear_resize(g, h, w)
        amp *= 0.6
    img = (img - img.mean()) / (img.std() + 1e-9)
    return np.clip(128.0 + 55.0 * img, 0, 255)


def four_point_homography(src, dst):
    """8x8 DLT for 4 correspondences, H22 == 1 (what src/data/utils.py:7-33 computes)."""
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i, ((x, y), (u, v)) in enumerate(zip(src, dst)):
        A[2 * i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[2 * i + 1] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[2 * i], b[2 * i + 1] = u, v
    return np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)


def warp_bilinear(img, H, out_h, out_w):
    """out(x, y) = bilinear img(H . (x, y, 1)), zeros outside (pixel-centre convention of
    cv2.warpPerspective(img, inv(H)), src/data/utils.py:61-64)."""
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.float64)
    den = H[2, 0] * xs + H[2, 1] * ys + H[2, 2]
    u = (H[0, 0] * xs + H[0, 1] * ys + H[0, 2]) / den
    v = (H[1, 0] * xs + H[1, 1] * ys + H[1, 2]) / den
    x0 = np.floor(u).astype(int)
    y0 = np.floor(v).astype(int)
    fx, fy = (u - x0)[..., None], (v - y0)[..., None]
    h, w = img.shape[:2]
    out = np.zeros((out_h, out_w, img.shape[2]))
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            out += np.where(ok[..., None], img[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)], 0.0) * wy * wx
    return out


def _photometric(rng, img, max_delta):
    """Brightness +-max_delta and contrast x[1-d/64, 1+d/64], each with p=0.5 (transforms.py:296-330,
    RGB part; the HSV hue/saturation jitter does not survive the grayscale conversion materially)."""
    img = img.copy()
    if rng.integers(2):
        img += rng.uniform(-max_delta, max_delta)
    if rng.integers(2):
        img *= rng.uniform(1.0 - max_delta / 64.0, 1.0 + max_delta / 64.0)
    return img


def make_pairs(batch, patch=128, rho=32, seed=42, photometric_max_delta=0, channels=1, pool=4, target=False):
    """Return dict of float32 arrays: patch_1, patch_2 [B,C,P,P] (standardised), delta [B,4,2]
    (ground-truth 4-point offsets, integers in [-rho, rho-1]), corners [B,4,2], homography [B,3,3]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    h = max(240, patch + 2 * rho + 48)
    w = max(320, patch + 2 * rho + 128)
    images = [texture_image(rng, h, w) for _ in range(min(pool, batch))]
    p1 = np.zeros((batch, channels, patch, patch), np.float32)
    p2 = np.zeros_like(p1)
    deltas = np.zeros((batch, 4, 2), np.float32)
    corners_all = np.zeros((batch, 4, 2), np.float32)
    Hs = np.zeros((batch, 3, 3), np.float32)
    half = patch // 2
    for b in range(batch):
        image = images[b % len(images)]
        im1, im2 = image, image
        if photometric_max_delta > 0:
            im1 = _photometric(rng, image, photometric_max_delta)
            im2 = _photometric(rng, image, photometric_max_delta)
        # transforms.py:505-506 (randint upper bound exclusive)
        px = int(rng.integers(rho + half, w - rho - half + 1))
        py = int(rng.integers(rho + half, h - rho - half + 1))
        corners = np.array([(px - half, py - half), (px + half, py - half),
                            (px + half, py + half), (px - half, py + half)], np.float64)
        delta = rng.integers(-rho, rho, 8).reshape(4, 2).astype(np.float64)        # transforms.py:538
        H = four_point_homography(corners, corners + delta)
        # image_2(x) = image(H x); crop both at `corners` (transforms.py:571-576)
        x0, y0 = int(corners[0, 0]), int(corners[0, 1])
        T = np.array([[1, 0, x0], [0, 1, y0], [0, 0, 1.0]])
        crop2 = warp_bilinear(im2, H @ T, patch, patch)
        crop1 = im1[y0:y0 + patch, x0:x0 + patch]
        for dst, crop in ((p1, crop1), (p2, crop2)):
            if channels == 1:
                g = crop[..., 0] * 0.299 + crop[..., 1] * 0.587 + crop[..., 2] * 0.114   # :351-353
                dst[b, 0] = ((g.astype(np.float32) / 255) - 0.443) / 0.129              # :377
            else:
                dst[b] = ((crop.astype(np.float32) / 255) - 0.443).transpose(2, 0, 1) / 0.129
        deltas[b], corners_all[b], Hs[b] = delta, corners, H
    out = {"patch_1": p1, "patch_2": p2, "delta": deltas, "corners": corners_all, "homography": Hs}
    if target:      # HomographyNetPrep 'all_points' (transforms.py:635-685): pf(x) = H x - x over the patch of image 1
        c = np.array([[0, 0], [patch, 0], [patch, patch], [0, patch]], np.float64)
        out["target"] = np.stack([perspective_field(four_point_homography(c, c + deltas[b].astype(np.float64)), patch)
                                  for b in range(batch)]).astype(np.float32)
    return out


def perspective_field(H, patch=128):
    """pf(x) = H x - x on the patch grid, [2,P,P] (x- then y-displacement): the quantity the Zeng
    backbone regresses (what `forward_map_field`, PerceptualHead.py:125-146, adds the grid back to)."""
    ys, xs = np.mgrid[0:patch, 0:patch].astype(np.float64)
    den = H[2, 0] * xs + H[2, 1] * ys + H[2, 2]
    u = (H[0, 0] * xs + H[0, 1] * ys + H[0, 2]) / den
    v = (H[1, 0] * xs + H[1, 1] * ys + H[1, 2]) / den
    return np.stack([u - xs, v - ys])


def make_head_inputs(batch, seed, noise=0.5, patch=128):
    """Head-only scenario: patch pairs plus a noisy ground-truth perspective field in both
    directions (stands in for the backbone output so head kernels can be checked in isolation)."""
    d = make_pairs(batch, patch=patch, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    c = np.array([[0, 0], [patch, 0], [patch, patch], [0, patch]], np.float64)
    pf12 = np.zeros((batch, 2, patch, patch), np.float32)
    pf21 = np.zeros_like(pf12)
    for b in range(batch):
        H12 = four_point_homography(c, c + d["delta"][b])
        pf12[b] = perspective_field(H12, patch) + noise * rng.standard_normal((2, patch, patch))
        pf21[b] = perspective_field(np.linalg.inv(H12), patch) + noise * rng.standard_normal((2, patch, patch))
    d["pf_hat_12"], d["pf_hat_21"] = pf12, pf21
    return d
