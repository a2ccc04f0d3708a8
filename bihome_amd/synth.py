"""Synthetic COCO-style 128x128 patch pairs (host / numpy version).

Mirrors the distribution of the reference's data pipeline without COCO, cv2 or a network
(SURVEY.md §8(d)): `HomographyNetPrep` (src/data/transforms.py:441-725), `DictToGrayscale`
(:344-354), `DictStandardize` (:369-378) and `PhotometricDistortSimple` (:296-330: brightness, contrast,
HSV saturation / hue, channel permutation).  `homography_net_prep` reproduces one reference sample draw for draw
(checked against the reference's own classes through tests/golden/datagen_*.npz).  Base images stand in for the offline-preprocessed
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


_EPS32 = np.float32(1.1920929e-07)
_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))        # transforms.py:235-237
_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])   # HSV->RGB: tab index of (b, g, r)


def rgb_to_hsv(img):
    """float32 RGB -> HSV the way the reference's cv2.cvtColor(COLOR_RGB2HSV) call does for CV_32F (transforms.py:167-168):
    V = max, S = (V - min) / (|V| + eps), H in degrees from the channel holding the max; inputs are not clipped."""
    img = np.asarray(img, np.float32)
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    diff = (v - np.minimum(np.minimum(r, g), b)).astype(np.float32)
    sat = diff / (np.abs(v) + _EPS32)
    d = (np.float32(60.0) / (diff + _EPS32)).astype(np.float32)
    h = np.where(v == r, (g - b) * d, np.where(v == g, (b - r) * d + np.float32(120.0), (r - g) * d + np.float32(240.0)))
    h = np.where(h < 0, h + np.float32(360.0), h)
    return np.stack([h, sat, v], -1).astype(np.float32)


def hsv_to_rgb(img):
    """float32 HSV -> RGB, sector-table form (the reference's cv2.cvtColor(COLOR_HSV2RGB), transforms.py:173-174)."""
    img = np.asarray(img, np.float32)
    h, sat, v = img[..., 0], img[..., 1], img[..., 2]
    hh = (h * np.float32(6.0 / 360.0)).astype(np.float32)
    hh = (hh - np.floor(hh / 6.0) * 6.0 * ((hh < 0) | (hh >= 6))).astype(np.float32)
    sector = np.floor(hh).astype(np.int64)
    frac = (hh - sector).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector, frac = np.where(bad, 0, sector), np.where(bad, np.float32(0), frac)
    one = np.float32(1.0)
    tab = np.stack([v, v * (one - sat), v * (one - sat * frac), v * (one - sat * (one - frac))], -1).astype(np.float32)
    idx = _SECTOR[sector]
    b, g, r = (np.take_along_axis(tab, idx[..., i:i + 1], -1)[..., 0] for i in range(3))
    grey = sat == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.float32)


def draw_photometric(rs, max_delta):
    """The random decisions of one PhotometricDistortSimple call (transforms.py:296-330, classes :141-245) in the
    reference's draw order, as a parameter record the host and device generators both apply:
    (brightness delta, contrast-first alpha, saturation alpha, hue delta, contrast-last alpha, channel permutation index).
    `rs`: numpy RandomState-like (randint / uniform)."""
    lower, upper = 1.0 - max_delta / 32 * 0.5, 1.0 + max_delta / 32 * 0.5            # :301-302
    br = rs.uniform(-max_delta, max_delta) if rs.randint(2) else 0.0                  # :152-155
    first = bool(rs.randint(2))                                                       # :322 (pd[:-1] or pd[1:])
    c1 = c2 = 1.0
    if first and rs.randint(2):                                                       # contrast before HSV (:167-169)
        c1 = rs.uniform(lower, upper)
    sat = rs.uniform(lower, upper) if rs.randint(2) else 1.0                          # :194-196
    hue = rs.uniform(-max_delta / 2, max_delta / 2) if rs.randint(2) else 0.0         # :206-210
    if not first and rs.randint(2):                                                   # contrast after HSV
        c2 = rs.uniform(lower, upper)
    perm = 0
    if max_delta > 0 and rs.randint(2):                                               # :327-328, :241-245
        perm = int(rs.randint(len(_PERMS)))
    return np.array([br, c1, sat, hue, c2, perm], np.float64)


def apply_photometric(img, p):
    """Apply one parameter record of `draw_photometric` to an HxWx3 image (float32 arithmetic, as upstream)."""
    br, c1, sat, hue, c2, perm = p
    im = np.asarray(img).astype(np.float32)                                           # ImageConvertFromInts :125-127
    im = im + np.float32(br)
    im = im * np.float32(c1)
    hsv = rgb_to_hsv(im)
    hsv[..., 1] *= np.float32(sat)
    if hue != 0.0:
        hch = hsv[..., 0] + np.float32(hue)
        hch = np.where(hch > 360.0, hch - np.float32(360.0), hch)
        hsv[..., 0] = np.where(hch < 0.0, hch + np.float32(360.0), hch)
    im = hsv_to_rgb(hsv) * np.float32(c2)
    return im[..., list(_PERMS[int(perm)])]


class _RandomStateAdapter:
    """numpy Generator (PCG64) behind the two RandomState calls the reference's transforms use."""

    def __init__(self, gen):
        self.gen = gen

    def randint(self, low, high=None, size=None):
        if high is None:
            low, high = 0, low
        return self.gen.integers(low, high, size)

    def uniform(self, low, high):
        return self.gen.uniform(low, high)


def _photometric(rng, img, max_delta):
    """PhotometricDistortSimple (transforms.py:296-330): brightness, contrast, HSV saturation / hue, channel permutation,
    each with p = 0.5, contrast either before or after the HSV part."""
    return apply_photometric(img, draw_photometric(_RandomStateAdapter(rng), max_delta)).astype(np.float64)


def homography_net_prep(rs, image, rho=32, patch=128, max_delta=0, photometric_keys=("image_1", "image_2")):
    """One sample exactly in the reference's order of operations and random draws (HomographyNetPrep.__call__,
    transforms.py:458-725, '4_points'): photometric distortion of both copies (applied even for max_delta 0, where it only
    consumes draws and a float32 HSV round trip), patch centre, corner offsets, homography, warp of image_2, crops.
    `rs`: numpy RandomState (the reference seeds one per transform, :451-454).  Returns the reference's dict (numpy)."""
    h, w = image.shape[:2]
    im1 = apply_photometric(image, draw_photometric(rs, max_delta)) if "image_1" in photometric_keys else np.copy(image)
    im2 = apply_photometric(image, draw_photometric(rs, max_delta)) if "image_2" in photometric_keys else np.copy(image)
    half = patch // 2
    if patch != w:                                                                    # :504-509
        px = int(rs.randint(rho + half, w - rho - half + 1))
        py = int(rs.randint(rho + half, h - rho - half + 1))
    else:
        px, py = w // 2, h // 2
    corners = np.array([(px - half, py - half), (px + half, py - half), (px + half, py + half), (px - half, py + half)])
    p1 = im1[corners[0, 1]:corners[3, 1], corners[0, 0]:corners[1, 0]]               # :521
    delta = rs.randint(-rho, rho, 8).reshape(4, 2)                                    # :538
    H = four_point_homography(corners.astype(np.float64), (corners + delta).astype(np.float64))       # :569-570
    # warp_image(image_2, H) = cv2.warpPerspective(image_2, inv(H)): image_2'(x) = image_2(H x)    (:571, utils.py:61-64)
    im2w = warp_bilinear(im2.astype(np.float64), H, h, w).astype(im2.dtype if im2.dtype.kind == "f" else np.float64)
    p2 = im2w[corners[0, 1]:corners[3, 1], corners[0, 0]:corners[1, 0]]               # :576
    return {"image_1": im1, "image_2": im2w, "patch_1": p1, "patch_2": p2, "corners": corners, "target": delta,
            "delta": delta, "homography": H}


def gray_standardize(patch, mean=0.443, std=0.129):
    """DictToGrayscale (:344-354) + DictStandardize (:369-378) + DictToTensor's HWC->CHW (:728-743) for one patch."""
    g = patch[:, :, 0] * 0.299 + patch[:, :, 1] * 0.587 + patch[:, :, 2] * 0.114
    return ((g[None].astype(np.float32) / 255) - mean) / std


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
