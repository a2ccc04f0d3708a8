// GPU-side synthetic pair generator: the reference's CPU data pipeline for one sample (src/data/transforms.py:
// HomographyNetPrep :441-725 -> crop patch_1, warp the image with the 4-point homography and crop patch_2;
// DictToGrayscale :344-354; DictStandardize :369-378; the brightness/contrast part of PhotometricDistortSimple
// :296-330) as ONE kernel over resident base images.  At >= 2k pairs/s per GPU the 8 cv2 DataLoader workers of the
// reference cannot keep up; this reuses the homography-warp arithmetic of csrc/warp.hip.
// HBM-bound: reads <= 4 taps x 3 channels of the base image per output pixel (cache-resident), writes 8 B/pixel.
#include "common.h"

// grid (P/16, P/16, B), block 256 = 16x16
__global__ void __launch_bounds__(256) synth_pairs_kernel(const float* __restrict__ images, const int* __restrict__ img_idx,
                                                          const float* __restrict__ origin, const double* __restrict__ Hp,
                                                          const float* __restrict__ photo, int Hs, int Ws, int P, float mean,
                                                          float inv_std, float* __restrict__ p1, float* __restrict__ p2) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    const float* img = images + (size_t)img_idx[b] * 3 * Hs * Ws;
    const size_t plane = (size_t)Hs * Ws;
    const int x0 = (int)origin[b * 2], y0 = (int)origin[b * 2 + 1];
    // photometric: out = (in + brightness) * contrast, per image of the pair
    const float br1 = photo ? photo[b * 4 + 0] : 0.f, ct1 = photo ? photo[b * 4 + 1] : 1.f;
    const float br2 = photo ? photo[b * 4 + 2] : 0.f, ct2 = photo ? photo[b * 4 + 3] : 1.f;
    auto gray = [&](int yy, int xx) {
        const float* q = img + (size_t)yy * Ws + xx;
        return q[0] * 0.299f + q[plane] * 0.587f + q[2 * plane] * 0.114f;      // transforms.py:351-353
    };
    // patch_1: plain crop at `origin`
    {
        const int yy = y0 + y, xx = x0 + x;
        float g = (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) ? gray(yy, xx) : 0.f;
        g = (g + br1 * 1.0f) * ct1;
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
            if (vx0 && vy0) g += gray(yi, xi) * (1 - ax) * (1 - ay);
            if (vx1 && vy0) g += gray(yi, xi + 1) * ax * (1 - ay);
            if (vx0 && vy1) g += gray(yi + 1, xi) * (1 - ax) * ay;
            if (vx1 && vy1) g += gray(yi + 1, xi + 1) * ax * ay;
        }
        g = (g + br2) * ct2;
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
