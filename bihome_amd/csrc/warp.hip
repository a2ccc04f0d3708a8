// Fused homography warp: out(x,y) = bilinear img(H.(x,y,1)) with zero padding, plus the pooled
// in-bounds coverage of the same map (= AvgPool(warp(ones))), and the adjoint w.r.t. H.
// HBM-bound: 8 B per pixel per channel (one gathered read, one coalesced write).
// Tile = 16x16 output pixels per 256-thread block; a wavefront owns a 16x4 strip, i.e. exactly one
// row of pooling windows for pool=4, so the pooled coverage is a pure in-wave shuffle reduction.
#include "common.h"

struct Tap {
    float x0f, y0f, fx, fy;
    bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ void project(const double* __restrict__ Hm, int x, int y, float& u, float& v, float& iz,
                                        bool& guard) {
    // same arithmetic order as transform_points on the pixel grid, in float like grid_sample's input
    float fx = (float)x, fy = (float)y;
    float h0 = (float)Hm[0], h1 = (float)Hm[1], h2 = (float)Hm[2], h3 = (float)Hm[3], h4 = (float)Hm[4],
          h5 = (float)Hm[5], h6 = (float)Hm[6], h7 = (float)Hm[7], h8 = (float)Hm[8];
    float qx = h0 * fx + h1 * fy + h2, qy = h3 * fx + h4 * fy + h5, qz = h6 * fx + h7 * fy + h8;
    guard = !(fabsf(qz) > 1e-8f);
    iz = guard ? 1.0f : 1.0f / qz;
    u = qx * iz;
    v = qy * iz;
}

__device__ __forceinline__ Tap make_tap(float u, float v, int w, int h) {
    Tap t;
    t.x0f = floorf(u);
    t.y0f = floorf(v);
    t.fx = u - t.x0f;
    t.fy = v - t.y0f;
    // comparisons in float so that wild coordinates (inf/nan/huge) are simply out of bounds
    t.vx0 = (t.x0f >= 0.0f) && (t.x0f <= (float)(w - 1));
    t.vx1 = (t.x0f >= -1.0f) && (t.x0f <= (float)(w - 2));
    t.vy0 = (t.y0f >= 0.0f) && (t.y0f <= (float)(h - 1));
    t.vy1 = (t.y0f >= -1.0f) && (t.y0f <= (float)(h - 2));
    return t;
}

// sum over a pool x pool window held by lanes of one wave laid out 16 wide x 4 tall (lane = ty*16+tx)
__device__ __forceinline__ float window_sum_16x4(float v, int pool) {
    // x direction: lanes differ in bits 0..3 ; y direction: bits 4..5
    for (int off = 1; off < pool && off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    for (int off = 1; off < pool && off < 4; off <<= 1) v += __shfl_xor(v, off * 16, 64);
    return v;
}

// grid: (w/16, h/16, B) ; block 256 = 16x16
__global__ void __launch_bounds__(256) warp_fwd_kernel(const float* __restrict__ img, const double* __restrict__ H64,
                                                       int C, int h, int w, int pool, float* __restrict__ out,
                                                       float* __restrict__ cov) {
    __shared__ float covrows[4][16];   // for pool > 4: per-wave partial rows
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    float u, v, iz;
    bool guard;
    project(H64 + (size_t)b * 9, x, y, u, v, iz, guard);
    Tap t = make_tap(u, v, w, h);
    const int x0 = (int)fminf(fmaxf(t.x0f, -2.0f), (float)w), y0 = (int)fminf(fmaxf(t.y0f, -2.0f), (float)h);
    const float w00 = (1 - t.fx) * (1 - t.fy), w01 = t.fx * (1 - t.fy), w10 = (1 - t.fx) * t.fy, w11 = t.fx * t.fy;
    const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
    if (img) {
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((size_t)b * C + c) * h * w;
            float acc = 0.0f;
            if (v00) acc += p[y0 * w + x0] * w00;
            if (v01) acc += p[y0 * w + x0 + 1] * w01;
            if (v10) acc += p[(y0 + 1) * w + x0] * w10;
            if (v11) acc += p[(y0 + 1) * w + x0 + 1] * w11;
            out[((size_t)b * C + c) * h * w + (size_t)y * w + x] = acc;
        }
    }
    if (cov) {
        float cv = (v00 ? w00 : 0.0f) + (v01 ? w01 : 0.0f) + (v10 ? w10 : 0.0f) + (v11 ? w11 : 0.0f);
        const int pw = w / pool;
        if (pool <= 4) {
            float s = window_sum_16x4(cv, pool);
            if ((tx % pool) == 0 && (ty % pool) == 0)
                cov[(size_t)b * (h / pool) * pw + (size_t)(y / pool) * pw + x / pool] = s / (float)(pool * pool);
        } else {
            // pool 8 or 16: reduce 16x4 strip per wave to per-window-column sums, combine via LDS
            float s = window_sum_16x4(cv, pool);           // x: full pool (<=16) ; y: 4 rows
            const int wave = threadIdx.x >> 6;
            if ((ty & 3) == 0 && (tx % pool) == 0) covrows[wave][tx] = s;
            __syncthreads();
            if (ty % pool == 0 && (tx % pool) == 0) {
                float tot = 0.0f;
                for (int k = 0; k < pool / 4; ++k) tot += covrows[ty / 4 + k][tx];
                cov[(size_t)b * (h / pool) * pw + (size_t)(y / pool) * pw + x / pool] = tot / (float)(pool * pool);
            }
        }
    }
}

// adjoint: per pixel dL/du, dL/dv -> 9 sums per sample
__global__ void __launch_bounds__(256) warp_bwd_kernel(const float* __restrict__ img, const double* __restrict__ H64,
                                                       const float* __restrict__ g_out, const float* __restrict__ g_cov,
                                                       int C, int h, int w, int pool, double* __restrict__ gH) {
    __shared__ double part[4][9];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    float u, v, iz;
    bool guard;
    project(H64 + (size_t)b * 9, x, y, u, v, iz, guard);
    Tap t = make_tap(u, v, w, h);
    const int x0 = (int)fminf(fmaxf(t.x0f, -2.0f), (float)w), y0 = (int)fminf(fmaxf(t.y0f, -2.0f), (float)h);
    const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
    // d(bilinear)/du = sum_taps val * d(weight)/du, out-of-bounds taps contribute nothing (grid_sampler_2d_backward)
    float gu = 0.0f, gv = 0.0f;
    if (img && g_out) {
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((size_t)b * C + c) * h * w;
            float go = g_out[((size_t)b * C + c) * h * w + (size_t)y * w + x];
            float p00 = v00 ? p[y0 * w + x0] : 0.0f, p01 = v01 ? p[y0 * w + x0 + 1] : 0.0f;
            float p10 = v10 ? p[(y0 + 1) * w + x0] : 0.0f, p11 = v11 ? p[(y0 + 1) * w + x0 + 1] : 0.0f;
            gu += go * ((p01 - p00) * (1 - t.fy) + (p11 - p10) * t.fy);
            gv += go * ((p10 - p00) * (1 - t.fx) + (p11 - p01) * t.fx);
        }
    }
    if (g_cov) {
        float gc = g_cov[(size_t)b * (h / pool) * (w / pool) + (size_t)(y / pool) * (w / pool) + x / pool] /
                   (float)(pool * pool);
        float o00 = v00 ? 1.0f : 0.0f, o01 = v01 ? 1.0f : 0.0f, o10 = v10 ? 1.0f : 0.0f, o11 = v11 ? 1.0f : 0.0f;
        gu += gc * ((o01 - o00) * (1 - t.fy) + (o11 - o10) * t.fy);
        gv += gc * ((o10 - o00) * (1 - t.fx) + (o11 - o01) * t.fx);
    }
    // u = qx*iz, v = qy*iz, iz = 1/qz (or 1 under the guard)
    double s[9];
    const double fx = (double)x, fy = (double)y, dgu = (double)gu, dgv = (double)gv, diz = (double)iz;
    s[0] = dgu * diz * fx; s[1] = dgu * diz * fy; s[2] = dgu * diz;
    s[3] = dgv * diz * fx; s[4] = dgv * diz * fy; s[5] = dgv * diz;
    double gz = guard ? 0.0 : -(dgu * (double)u + dgv * (double)v) * diz;
    s[6] = gz * fx; s[7] = gz * fy; s[8] = gz;
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = wave_sum(s[i]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int i = 0; i < 9; ++i) part[wave][i] = s[i];
    __syncthreads();
    if (threadIdx.x < 9) {
        double tot = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        atomicAdd(gH + (size_t)b * 9 + threadIdx.x, tot);
    }
}

// ---------------------------------------------------------------------------------------------
// pool = 4 fast path (the shipped configuration: 128x128 patches, 32x32 features): four consecutive pixels of a row per
// thread, so the output is one 16-byte store per lane, a thread's four pixels are exactly one pooling-window row (the
// pooled coverage needs two shuffles over the wave's four rows), and the adjoint's nine double sums are reduced over a
// quarter of the wavefronts.  Per-pixel arithmetic is the same as in the generic kernels (same results).
// grid (w/64, h/16, B), block 256 = 16 (x quads) x 16 (rows); a wave = 16 quads x 4 rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) warp_fwd4_kernel(const float* __restrict__ img, const double* __restrict__ H64, int C,
                                                        int h, int w, float* __restrict__ out, float* __restrict__ cov) {
    const int b = blockIdx.z;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int xq = blockIdx.x * 64 + tq * 4, y = blockIdx.y * 16 + ty;
    // branch-free taps: every tap index is clamped into the image and an out-of-bounds tap gets weight 0, so the sixteen
    // gathers of a thread issue back to back instead of sitting behind per-tap branches
    int i00[4], i01[4], i10[4], i11[4];
    float w00[4], w01[4], w10[4], w11[4];
    float cv = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float u, v, iz;
        bool guard;
        project(H64 + (size_t)b * 9, xq + i, y, u, v, iz, guard);
        const Tap t = make_tap(u, v, w, h);
        const int x0 = (int)fminf(fmaxf(t.x0f, -2.0f), (float)w), y0 = (int)fminf(fmaxf(t.y0f, -2.0f), (float)h);
        const int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
        const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
        i00[i] = ya * w + xa; i01[i] = ya * w + xb; i10[i] = yb * w + xa; i11[i] = yb * w + xb;
        w00[i] = (t.vx0 && t.vy0) ? (1 - t.fx) * (1 - t.fy) : 0.0f;
        w01[i] = (t.vx1 && t.vy0) ? t.fx * (1 - t.fy) : 0.0f;
        w10[i] = (t.vx0 && t.vy1) ? (1 - t.fx) * t.fy : 0.0f;
        w11[i] = (t.vx1 && t.vy1) ? t.fx * t.fy : 0.0f;
        cv += w00[i] + w01[i] + w10[i] + w11[i];
    }
    if (img) {
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((size_t)b * C + c) * h * w;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p00 = p[i00[i]], p01 = p[i01[i]], p10 = p[i10[i]], p11 = p[i11[i]];
                float acc = 0.0f;
                acc += p00 * w00[i];
                acc += p01 * w01[i];
                acc += p10 * w10[i];
                acc += p11 * w11[i];
                o[i] = acc;
            }
            *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * h * w + (size_t)y * w + xq) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (cov) {
        cv += __shfl_xor(cv, 16, 64);                 // the four rows of the wave (lane = (ty & 3) * 16 + tq)
        cv += __shfl_xor(cv, 32, 64);
        if ((ty & 3) == 0) cov[(size_t)b * (h / 4) * (w / 4) + (size_t)(y / 4) * (w / 4) + xq / 4] = cv * (1.0f / 16.0f);
    }
}

__global__ void __launch_bounds__(256) warp_bwd4_kernel(const float* __restrict__ img, const double* __restrict__ H64,
                                                        const float* __restrict__ g_out, const float* __restrict__ g_cov, int C,
                                                        int h, int w, double* __restrict__ gH) {
    __shared__ double part[4][9];
    const int b = blockIdx.z;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int xq = blockIdx.x * 64 + tq * 4, y = blockIdx.y * 16 + ty;
    const float gc = g_cov ? g_cov[(size_t)b * (h / 4) * (w / 4) + (size_t)(y / 4) * (w / 4) + xq / 4] * (1.0f / 16.0f) : 0.0f;
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    float4 go4[4];                                     // up to 4 channels cached (C = 1 or 3 here); more are re-read
#pragma unroll
    for (int c = 0; c < 4; ++c)
        go4[c] = (img && g_out && c < C) ? *reinterpret_cast<const float4*>(g_out + ((size_t)b * C + c) * h * w + (size_t)y * w + xq)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = xq + i;
        float u, v, iz;
        bool guard;
        project(H64 + (size_t)b * 9, x, y, u, v, iz, guard);
        const Tap t = make_tap(u, v, w, h);
        const int x0 = (int)fminf(fmaxf(t.x0f, -2.0f), (float)w), y0 = (int)fminf(fmaxf(t.y0f, -2.0f), (float)h);
        const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
        const int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);       // clamped: loads are unconditional
        const int ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
        float gu = 0.0f, gv = 0.0f;
        if (img && g_out) {
            for (int c = 0; c < C; ++c) {
                const float* p = img + ((size_t)b * C + c) * h * w;
                float go;
                if (c < 4) { const float4 q = go4[c]; go = i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w)); }
                else go = g_out[((size_t)b * C + c) * h * w + (size_t)y * w + x];
                const float q00 = p[ya * w + xa], q01 = p[ya * w + xb], q10 = p[yb * w + xa], q11 = p[yb * w + xb];
                const float p00 = v00 ? q00 : 0.0f, p01 = v01 ? q01 : 0.0f, p10 = v10 ? q10 : 0.0f, p11 = v11 ? q11 : 0.0f;
                gu += go * ((p01 - p00) * (1 - t.fy) + (p11 - p10) * t.fy);
                gv += go * ((p10 - p00) * (1 - t.fx) + (p11 - p01) * t.fx);
            }
        }
        if (g_cov) {
            const float o00 = v00 ? 1.0f : 0.0f, o01 = v01 ? 1.0f : 0.0f, o10 = v10 ? 1.0f : 0.0f, o11 = v11 ? 1.0f : 0.0f;
            gu += gc * ((o01 - o00) * (1 - t.fy) + (o11 - o10) * t.fy);
            gv += gc * ((o10 - o00) * (1 - t.fx) + (o11 - o01) * t.fx);
        }
        const double fx = (double)x, fy = (double)y, dgu = (double)gu, dgv = (double)gv, diz = (double)iz;
        s[0] += dgu * diz * fx; s[1] += dgu * diz * fy; s[2] += dgu * diz;
        s[3] += dgv * diz * fx; s[4] += dgv * diz * fy; s[5] += dgv * diz;
        const double gz = guard ? 0.0 : -(dgu * (double)u + dgv * (double)v) * diz;
        s[6] += gz * fx; s[7] += gz * fy; s[8] += gz;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 9; ++k) part[wave][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 9)
        atomicAdd(gH + (size_t)b * 9 + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" {

int bh_warp_fwd(const float* img, const double* H64, int B, int C, int h, int w, int pool, float* out, float* cov,
                void* stream) {
    if (!H64 || B < 0 || (img && !out) || (!img && !cov)) return BH_E_BADARG;
    if ((h % 16) || (w % 16) || (pool != 1 && pool != 2 && pool != 4 && pool != 8 && pool != 16)) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    if (pool == 4 && (w % 64) == 0) {
        hipLaunchKernelGGL(warp_fwd4_kernel, dim3(w / 64, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, C, h, w, out, cov);
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(w / 16, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, C, h, w, pool,
                       out, cov);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_warp_bwd(const float* img, const double* H64, const float* g_out, const float* g_cov, int B, int C, int h, int w,
                int pool, double* gH, void* stream) {
    if (!H64 || !gH || B < 0 || (g_out && !img)) return BH_E_BADARG;
    if ((h % 16) || (w % 16) || (pool != 1 && pool != 2 && pool != 4 && pool != 8 && pool != 16)) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    if (pool == 4 && (w % 64) == 0) {
        hipLaunchKernelGGL(warp_bwd4_kernel, dim3(w / 64, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, g_out, g_cov, C,
                           h, w, gH);
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(w / 16, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, g_out, g_cov,
                       C, h, w, pool, gH);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
