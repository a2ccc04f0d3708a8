// GPU-side synthetic pair generator: the reference's CPU data pipeline for one sample (src/data/transforms.py:
// HomographyNetPrep :441-725 -> crop patch_1, warp the image with the 4-point homography and crop patch_2;
// DictToGrayscale :344-354; DictStandardize :369-378; PhotometricDistortSimple :296-330 - brightness, contrast, HSV
// saturation / hue, channel permutation) as ONE kernel over resident base images.  At >= 2k pairs/s per GPU the 8 cv2 DataLoader workers of the
// reference cannot keep up; this reuses the homography-warp arithmetic of csrc/warp.hip.
// HBM-bound: reads <= 4 taps x 3 channels of the base image per output pixel (cache-resident), writes 8 B/pixel.
#include "common.h"

// One PhotometricDistortSimple record (bihome_amd/synth.py draw_photometric; transforms.py:296-330) applied to an RGB
// triple: brightness, contrast (before the HSV part), saturation / hue in OpenCV's float HSV (cvtColor CV_32F: V = max,
// S = (V - min) / (|V| + eps), H in degrees), contrast (after), channel permutation.  Returns the grayscale value
// 0.299 R' + 0.587 G' + 0.114 B' of the permuted triple (transforms.py:351-353).
struct PhotoRec { float br, c1, sat, hue, c2; int perm; };

__device__ __forceinline__ float photo_gray(float r, float g, float b, const PhotoRec& p) {
    constexpr float EPS = 1.1920929e-07f;
    r = (r + p.br) * p.c1; g = (g + p.br) * p.c1; b = (b + p.br) * p.c1;
    // RGB -> HSV
    const float v = fmaxf(fmaxf(r, g), b), vmin = fminf(fminf(r, g), b);
    const float diff = v - vmin;
    float s = diff / (fabsf(v) + EPS);
    const float d = 60.0f / (diff + EPS);
    float h = (v == r) ? (g - b) * d : ((v == g) ? (b - r) * d + 120.0f : (r - g) * d + 240.0f);
    if (h < 0.0f) h += 360.0f;
    s *= p.sat;
    if (p.hue != 0.0f) {
        h += p.hue;
        if (h > 360.0f) h -= 360.0f;
        if (h < 0.0f) h += 360.0f;
    }
    // HSV -> RGB (sector table)
    float R = v, G = v, B = v;
    if (s != 0.0f) {
        float hh = h * (6.0f / 360.0f);
        if (hh < 0.0f || hh >= 6.0f) hh -= floorf(hh / 6.0f) * 6.0f;
        int sector = (int)floorf(hh);
        float fr = hh - (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; fr = 0.0f; }
        const float t1 = v * (1.0f - s), t2 = v * (1.0f - s * fr), t3 = v * (1.0f - s * (1.0f - fr));
        switch (sector) {
            case 0: R = v;  G = t3; B = t1; break;
            case 1: R = t2; G = v;  B = t1; break;
            case 2: R = t1; G = v;  B = t3; break;
            case 3: R = t1; G = t2; B = v;  break;
            case 4: R = t3; G = t1; B = v;  break;
            default: R = v; G = t1; B = t2; break;
        }
    }
    R *= p.c2; G *= p.c2; B *= p.c2;
    // out[c] = in[perm[c]] for perm in ((0,1,2),(0,2,1),(1,0,2),(1,2,0),(2,0,1),(2,1,0))
    float o0 = R, o1 = G, o2 = B;
    switch (p.perm) {
        case 1: o1 = B; o2 = G; break;
        case 2: o0 = G; o1 = R; break;
        case 3: o0 = G; o1 = B; o2 = R; break;
        case 4: o0 = B; o1 = R; o2 = G; break;
        case 5: o0 = B; o2 = R; break;
        default: break;
    }
    return o0 * 0.299f + o1 * 0.587f + o2 * 0.114f;
}

// grid (P/16, P/16, B), block 256 = 16x16.  photo: [B][2 images][6] records or NULL (no distortion: plain grayscale).
__global__ void __launch_bounds__(256) synth_pairs_kernel(const float* __restrict__ images, const int* __restrict__ img_idx,
                                                          const float* __restrict__ origin, const double* __restrict__ Hp,
                                                          const float* __restrict__ photo, int Hs, int Ws, int P, float mean,
                                                          float inv_std, float* __restrict__ p1, float* __restrict__ p2) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    const float* img = images + (size_t)img_idx[b] * 3 * Hs * Ws;
    const size_t plane = (size_t)Hs * Ws;
    const int x0 = (int)origin[b * 2], y0 = (int)origin[b * 2 + 1];
    PhotoRec r1 = {0.f, 1.f, 1.f, 0.f, 1.f, 0}, r2 = r1;
    if (photo) {
        const float* q = photo + (size_t)b * 12;
        r1 = {q[0], q[1], q[2], q[3], q[4], (int)q[5]};
        r2 = {q[6], q[7], q[8], q[9], q[10], (int)q[11]};
    }
    // the distortion is applied to the IMAGE before it is warped (transforms.py:474-481 precede :571): per tap
    auto gray = [&](int yy, int xx, const PhotoRec& r) {
        const float* q = img + (size_t)yy * Ws + xx;
        if (!photo) return q[0] * 0.299f + q[plane] * 0.587f + q[2 * plane] * 0.114f;      // transforms.py:351-353
        return photo_gray(q[0], q[plane], q[2 * plane], r);
    };
    // patch_1: plain crop at `origin`
    {
        const int yy = y0 + y, xx = x0 + x;
        const float g = (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) ? gray(yy, xx, r1) : 0.f;
        p1[((size_t)b * P + y) * P + x] = (g * (1.0f / 255.0f) - mean) * inv_std;                 // transforms.py:377
    }
    // patch_2(x) = image(origin + Hpatch.x), bilinear, zeros outside (cv2.warpPerspective(img, inv(H)), utils.py:61-64)
    {
        const double* H = Hp + (size_t)b * 9;
        const double fx = x, fy = y;
        const double qz = H[6] * fx + H[7] * fy + H[8];
        const double u = (H[0] * fx + H[1] * fy + H[2]) / qz + x0, v = (H[3] * fx + H[4] * fy + H[5]) / qz + y0;
        const float uf = (float)u, vf = (float)v;
        const float xf = floorf(uf), yf = floorf(vf);
        const float ax = uf - xf, ay = vf - yf;
        float g = 0.f;
        if (xf >= -1.f && xf <= (float)Ws && yf >= -1.f && yf <= (float)Hs) {
            const int xi = (int)xf, yi = (int)yf;
            const bool vx0 = xi >= 0 && xi < Ws, vx1 = xi + 1 >= 0 && xi + 1 < Ws;
            const bool vy0 = yi >= 0 && yi < Hs, vy1 = yi + 1 >= 0 && yi + 1 < Hs;
            if (vx0 && vy0) g += gray(yi, xi, r2) * (1 - ax) * (1 - ay);
            if (vx1 && vy0) g += gray(yi, xi + 1, r2) * ax * (1 - ay);
            if (vx0 && vy1) g += gray(yi + 1, xi, r2) * (1 - ax) * ay;
            if (vx1 && vy1) g += gray(yi + 1, xi + 1, r2) * ax * ay;
        }
        p2[((size_t)b * P + y) * P + x] = (g * (1.0f / 255.0f) - mean) * inv_std;
    }
}

extern "C" {

int bh_synth_pairs(const float* images, const int* img_idx, const float* origin, const double* Hpatch, const float* photo,
                   int B, int n_images, int Hs, int Ws, int P, float mean, float std, float* patch1, float* patch2,
                   void* stream) {
    if (!images || !img_idx || !origin || !Hpatch || !patch1 || !patch2 || B < 0 || n_images < 1 || std == 0.f)
        return BH_E_BADARG;
    if (P % 16) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    hipLaunchKernelGGL(synth_pairs_kernel, dim3(P / 16, P / 16, B), dim3(256), 0, bh_stream(stream), images, img_idx, origin,
                       Hpatch, photo, Hs, Ws, P, mean, 1.0f / std, patch1, patch2);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
