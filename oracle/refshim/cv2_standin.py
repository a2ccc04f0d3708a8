"""Stand-in for the five OpenCV calls on the reference's DATA-GENERATION path (src/data/transforms.py, src/data/utils.py
numpy branches), so that the reference's own `PhotometricDistortSimple` / `HomographyNetPrep` classes run in the build
container (opencv is not installable offline).  TEST INFRASTRUCTURE ONLY: used by oracle/make_golden.py to produce
tests/golden/datagen_*.npz; never imported by the product.

Restated from OpenCV's published algorithms (modules/imgproc/src/color_hsv.simd.hpp RGB2HSV_f / HSV2RGB_f for CV_32F,
imgwarp.cpp getPerspectiveTransform / perspectiveTransform / warpPerspective):
  * cvtColor(float32 RGB -> HSV): V = max, S = (V - min) / (|V| + FLT_EPSILON), H in degrees [0, 360) from the
    channel that holds the max (R first, then G, then B), 60 / (diff + FLT_EPSILON) scale; no clipping of inputs.
  * cvtColor(float32 HSV -> RGB): sector table form, h wrapped into [0, 6), S == 0 -> grey.
  * getPerspectiveTransform: the 8x8 system in double, H22 = 1.
  * warpPerspective(img, M, dsize): dst(x) = bilinear src(M^-1 x), constant-zero border.  **Idealised**: OpenCV rounds
    the sampling coordinates to 1/32 px (INTER_BITS = 5) and, for the border, blends with zeros the same way; this
    stand-in samples at the exact coordinate.  Fixtures made through it therefore pin the reference's control flow,
    random-draw order, corner / homography conventions and crop arithmetic, not OpenCV's fixed-point rounding.
"""
import numpy as np

COLOR_BGR2HSV, COLOR_RGB2HSV, COLOR_BGR2RGB, COLOR_HSV2BGR, COLOR_HSV2RGB, COLOR_RGB2GRAY = 40, 41, 4, 54, 55, 7
FLT_EPSILON = np.float32(1.1920929e-07)


def _rgb2hsv_f(img):
    img = np.asarray(img, np.float32)
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = (v - vmin).astype(np.float32)
    s = diff / (np.abs(v) + FLT_EPSILON)
    d = (np.float32(60.0) / (diff + FLT_EPSILON)).astype(np.float32)
    h = np.where(v == r, (g - b) * d, np.where(v == g, (b - r) * d + np.float32(120.0), (r - g) * d + np.float32(240.0)))
    h = np.where(h < 0, h + np.float32(360.0), h).astype(np.float32)
    return np.stack([h, s.astype(np.float32), v.astype(np.float32)], -1)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # (b, g, r) <- tab index


def _hsv2rgb_f(img):
    img = np.asarray(img, np.float32)
    h, s, v = img[..., 0], img[..., 1], img[..., 2]
    hh = (h * np.float32(6.0 / 360.0)).astype(np.float32)
    hh = np.where(hh < 0, hh - np.floor(hh / 6.0) * 6.0, hh)
    hh = np.where(hh >= 6, hh - np.floor(hh / 6.0) * 6.0, hh).astype(np.float32)
    sector = np.floor(hh).astype(np.int64)
    frac = (hh - sector).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, np.float32(0), frac)
    one = np.float32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * frac), v * (one - s * (one - frac))], -1).astype(np.float32)
    idx = _SECTOR[sector]                                           # [..., 3] tab indices of (b, g, r)
    b = np.take_along_axis(tab, idx[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, idx[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, idx[..., 2:3], -1)[..., 0]
    grey = s == 0
    r, g, b = np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)
    return np.stack([r, g, b], -1).astype(np.float32)


def cvtColor(image, code):
    if code == COLOR_RGB2HSV:
        return _rgb2hsv_f(image)
    if code == COLOR_HSV2RGB:
        return _hsv2rgb_f(image)
    if code == COLOR_BGR2HSV:
        return _rgb2hsv_f(np.asarray(image)[..., ::-1])
    if code == COLOR_HSV2BGR:
        return _hsv2rgb_f(image)[..., ::-1]
    if code == COLOR_BGR2RGB:
        return np.asarray(image)[..., ::-1].copy()
    raise NotImplementedError(code)


def getPerspectiveTransform(src, dst):
    src, dst = np.asarray(src, np.float64).reshape(4, 2), np.asarray(dst, np.float64).reshape(4, 2)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)


def perspectiveTransform(points, M):
    p = np.asarray(points, np.float64)
    M = np.asarray(M, np.float64)
    w = p[..., 0] * M[2, 0] + p[..., 1] * M[2, 1] + M[2, 2]
    x = (p[..., 0] * M[0, 0] + p[..., 1] * M[0, 1] + M[0, 2]) / w
    y = (p[..., 0] * M[1, 0] + p[..., 1] * M[1, 1] + M[1, 2]) / w
    return np.stack([x, y], -1).astype(np.asarray(points).dtype)


def warpPerspective(image, M, dsize):
    """dst(x, y) = bilinear image(M^-1 (x, y, 1)), zeros outside (flags INTER_LINEAR, BORDER_CONSTANT 0)."""
    img = np.asarray(image)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    w, h = dsize
    Mi = np.linalg.inv(np.asarray(M, np.float64))
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    u = (Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / den
    v = (Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / den
    x0, y0 = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
    fx, fy = (u - x0)[..., None], (v - y0)[..., None]
    H, W = img.shape[:2]
    out = np.zeros((h, w, img.shape[2]), np.float64)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            out += np.where(ok[..., None], img[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)], 0.0) * wy * wx
    out = out.astype(img.dtype if img.dtype.kind == "f" else np.float64)
    return out[..., 0] if squeeze else out
