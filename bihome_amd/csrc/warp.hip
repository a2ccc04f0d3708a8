// Fused homography warp: out(x,y) = bilinear img(H.(x,y,1)) with zero padding, plus the pooled
// in-bounds coverage of the same map (= AvgPool(warp(ones))), and the adjoint w.r.t. H.
// HBM-bound: 8 B per pixel per channel (one gathered read, one coalesced write).
// Tile = 16x16 output pixels per 256-thread block; a wavefront owns a 16x4 strip, i.e. exactly one
// row of pooling windows for pool=4, so the pooled coverage is a pure in-wave shuffle reduction.
#include "common.h"
#include "warp_tap.h"

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
    // (the coordinate arithmetic of warp_tap.h make_tap4, spelled the same way: the kernels of every pooling size agree bitwise on where a
    //  pixel lands - and so on which side of an integer coordinate, where the bilinear derivative jumps)
    float qx = __builtin_fmaf(h0, fx, __builtin_fmaf(h1, fy, h2)), qy = __builtin_fmaf(h3, fx, __builtin_fmaf(h4, fy, h5)),
          qz = __builtin_fmaf(h6, fx, __builtin_fmaf(h7, fy, h8));
    guard = !(fabsf(qz) > 1e-8f);
    float r = __builtin_amdgcn_rcpf(qz);
    r = __builtin_fmaf(__builtin_fmaf(-qz, r, 1.0f), r, r);
    iz = guard ? 1.0f : r;
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
        } else if (pool > 16) {
            // pool 32 (extractor output layer 4): the 16x16 tile is a quarter of one window - tile sum, one atomic add
            // into the coverage the host zeroed before the launch
            float s = window_sum_16x4(cv, 16);             // 16 x 4 strip of this wave
            const int wave = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) covrows[wave][0] = s;
            __syncthreads();
            if (threadIdx.x == 0)
                atomicAdd(&cov[(size_t)b * (h / pool) * pw + (size_t)(y / pool) * pw + x / pool],
                          (covrows[0][0] + covrows[1][0] + covrows[2][0] + covrows[3][0]) / (float)(pool * pool));
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

// pooled coverage for pool = 32, deterministic form (round 4): one workgroup per 32 x 32 window walks its four 16 x 16 quarters in a
// fixed order and is the only writer of the window (the default form adds the quarter sums with float atomics).  grid (w/32, h/32, B)
__global__ void __launch_bounds__(256) warp_cov32_kernel(const double* __restrict__ H64, int h, int w, float* __restrict__ cov) {
    __shared__ float part[4];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float tot = 0.0f;
    for (int q = 0; q < 4; ++q) {
        const int x = blockIdx.x * 32 + (q & 1) * 16 + tx, y = blockIdx.y * 32 + (q >> 1) * 16 + ty;
        float u, v, iz;
        bool guard;
        project(H64 + (size_t)b * 9, x, y, u, v, iz, guard);
        const Tap t = make_tap(u, v, w, h);
        const float w00 = (1 - t.fx) * (1 - t.fy), w01 = t.fx * (1 - t.fy), w10 = (1 - t.fx) * t.fy, w11 = t.fx * t.fy;
        const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
        const float cv = (v00 ? w00 : 0.0f) + (v01 ? w01 : 0.0f) + (v10 ? w10 : 0.0f) + (v11 ? w11 : 0.0f);
        const float s = window_sum_16x4(cv, 16);           // the 16 x 4 strip of this wave
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        tot += (part[0] + part[1] + part[2] + part[3]) / 1024.0f;      // (the same quarter value the atomic form adds)
    }
    if (threadIdx.x == 0) cov[((size_t)b * (h / 32) + blockIdx.y) * (w / 32) + blockIdx.x] = tot;
}

// adjoint: per pixel dL/du, dL/dv -> 9 sums per sample
__global__ void __launch_bounds__(256) warp_bwd_kernel(const float* __restrict__ img, const double* __restrict__ H64,
                                                       const float* __restrict__ g_out, const float* __restrict__ g_cov,
                                                       int C, int h, int w, int pool, double* __restrict__ gH) {
    __shared__ double part[4][9];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (deterministic mode: grid (1, 1, B) - one workgroup walks every tile of its sample and is the only writer of gH[b])
    for (int by = blockIdx.y; by < h / 16; by += gridDim.y)
    for (int bx = blockIdx.x; bx < w / 16; bx += gridDim.x) {
    const int x = bx * 16 + tx, y = by * 16 + ty;
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
    {
    const double fx = (double)x, fy = (double)y, dgu = (double)gu, dgv = (double)gv, diz = (double)iz;
    s[0] = dgu * diz * fx; s[1] = dgu * diz * fy; s[2] = dgu * diz;
    s[3] = dgv * diz * fx; s[4] = dgv * diz * fy; s[5] = dgv * diz;
    double gz = guard ? 0.0 : -(dgu * (double)u + dgv * (double)v) * diz;
    s[6] = gz * fx; s[7] = gz * fy; s[8] = gz;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) acc9[i] += s[i];
    }
    double s[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = wave_sum(acc9[i]);
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
// pool = 4 fast path (the shipped configuration: 128x128 patches, 32x32 features).  At 64-128 images of 128x128 these
// launches move ~17-25 MB and are bound by VALU issue, not by HBM (wave64 on 16-lane SIMDs: the generic kernels spend
// ~95 / ~215 vector instructions per pixel), so this path is written for instruction count:
//   * four consecutive pixels of a row per thread and RPT rows per thread: the output is one 16-byte store per lane,
//     a thread's four pixels are one pooling-window row (the pooled coverage needs two shuffles over the wave's four
//     rows), and the adjoint's nine double sums are reduced across the wave once per 4*RPT pixels;
//   * taps clamped into the image with the validity folded into per-axis weights (w00 = wx0 * wy0, products with an
//     exact 0 or 1 - same values as masking), unsigned 32-bit indexing, 1/qz as v_rcp_f32 + one Newton step;
//   * adjoint: per row sum(a) and sum(a * x) are accumulated and multiplied by y once per row.
// grid (w/64, h/(16*RPT), B), block 256 = 16 (x quads) x 16 (rows); a wave = 16 quads x 4 rows; row r of a thread is
// y = (blockIdx.y * RPT + r) * 16 + ty.
// ---------------------------------------------------------------------------------------------
template <int RPT>
__global__ void __launch_bounds__(256) warp_fwd4_kernel(const float* __restrict__ img, const double* __restrict__ H64, int C,
                                                        int h, int w, float* __restrict__ out, float* __restrict__ cov) {
    const int b = blockIdx.z;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int xq = blockIdx.x * 64 + tq * 4;
    const Hf H = load_h(H64 + (size_t)b * 9);
    const unsigned plane = (unsigned)h * (unsigned)w;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = (blockIdx.y * RPT + r) * 16 + ty;
        unsigned o00[4], o01[4], o10[4], o11[4];
        float w00[4], w01[4], w10[4], w11[4];
        float cv = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const Tap4 t = make_tap4(H, xq + i, y, w, h);
            o00[i] = t.o00; o01[i] = t.o01; o10[i] = t.o10; o11[i] = t.o11;
            float ws_;
            tap_weights(t, w00[i], w01[i], w10[i], w11[i], ws_);
            cv += ws_;
        }
        if (img) {
            for (int c = 0; c < C; ++c) {
                const __amdgpu_buffer_rsrc_t rs = plane_rsrc(img + ((size_t)b * C + c) * plane, plane * 4u);
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p00 = ldtap(rs, o00[i]), p01 = ldtap(rs, o01[i]), p10 = ldtap(rs, o10[i]), p11 = ldtap(rs, o11[i]);
                    o[i] = tap_blend(p00, p01, p10, p11, w00[i], w01[i], w10[i], w11[i]);
                }
                *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * plane + (unsigned)y * (unsigned)w + xq) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        if (cov) {
            cv += __shfl_xor(cv, 16, 64);                 // the four rows of the wave (lane = (ty & 3) * 16 + tq)
            cv += __shfl_xor(cv, 32, 64);
            if ((ty & 3) == 0) cov[(size_t)b * (h / 4) * (w / 4) + (size_t)(y / 4) * (w / 4) + xq / 4] = cv * (1.0f / 16.0f);
        }
    }
}

template <int RPT>
__global__ void __launch_bounds__(256) warp_bwd4_kernel(const float* __restrict__ img, const double* __restrict__ H64,
                                                        const float* __restrict__ g_out, const float* __restrict__ g_cov, int C,
                                                        int h, int w, double* __restrict__ gH) {
    __shared__ double part[4][9];
    const int b = blockIdx.z;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const Hf H = load_h(H64 + (size_t)b * 9);
    const unsigned plane = (unsigned)h * (unsigned)w;
    const bool have_img = img && g_out;
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    // (deterministic mode: grid (1, 1, B) - one workgroup walks every tile position of its sample and is the only writer of gH[b])
    for (int by = blockIdx.y; by < h / (16 * RPT); by += gridDim.y)
    for (int bx = blockIdx.x; bx < w / 64; bx += gridDim.x) {
    const int xq = bx * 64 + tq * 4;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = (by * RPT + r) * 16 + ty;
        const unsigned pix = (unsigned)y * (unsigned)w + xq;
        const float gc = g_cov ? g_cov[(size_t)b * (h / 4) * (w / 4) + (size_t)(y / 4) * (w / 4) + xq / 4] * (1.0f / 16.0f) : 0.0f;
        float gu[4], gv[4];
        Tap4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i] = make_tap4(H, xq + i, y, w, h);
            // coverage part: d/du of sum(valid taps' weights) = (vx1 - vx0) * (wy0 + wy1), same for v
            const float sy = t[i].wy0 + t[i].wy1, sx = t[i].wx0 + t[i].wx1;
            gu[i] = gc * ((t[i].vx1 ? sy : 0.0f) - (t[i].vx0 ? sy : 0.0f));
            gv[i] = gc * ((t[i].vy1 ? sx : 0.0f) - (t[i].vy0 ? sx : 0.0f));
        }
        if (have_img) {
            for (int c = 0; c < C; ++c) {
                const __amdgpu_buffer_rsrc_t rs = plane_rsrc(img + ((size_t)b * C + c) * plane, plane * 4u);
                const float4 go4 = *reinterpret_cast<const float4*>(g_out + ((size_t)b * C + c) * plane + pix);
                const float go[4] = {go4.x, go4.y, go4.z, go4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // taps outside the image load 0: they contribute nothing (grid_sampler_2d_backward)
                    const float p00 = ldtap(rs, t[i].o00), p01 = ldtap(rs, t[i].o01), p10 = ldtap(rs, t[i].o10), p11 = ldtap(rs, t[i].o11);
                    // d(bilinear)/du = sum_taps val * d(weight)/du; the per-axis weights are 0 on rows / columns outside
                    gu[i] += go[i] * ((p01 - p00) * t[i].wy0 + (p11 - p10) * t[i].wy1);
                    gv[i] += go[i] * ((p10 - p00) * t[i].wx0 + (p11 - p01) * t[i].wx1);
                }
            }
        }
        // u = qx*iz, v = qy*iz, iz = 1/qz (or 1 under the guard): per row  A = sum a, Ax = sum a*x  for a in (gu*iz, gv*iz, gz);
        // the per-pixel products are float (like u, v and the weights), the sums double
        double au = 0, aux = 0, av = 0, avx = 0, az = 0, azx = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double fx = (double)(xq + i);
            const float a = gu[i] * t[i].iz, bq = gv[i] * t[i].iz;
            const float gz = t[i].guard ? 0.0f : -(gu[i] * t[i].u + gv[i] * t[i].v) * t[i].iz;
            au += (double)a; aux += (double)a * fx; av += (double)bq; avx += (double)bq * fx; az += (double)gz; azx += (double)gz * fx;
        }
        const double fy = (double)y;
        s[0] += aux; s[1] += au * fy; s[2] += au;
        s[3] += avx; s[4] += av * fy; s[5] += av;
        s[6] += azx; s[7] += az * fy; s[8] += az;
    }
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

// ---------------------------------------------------------------------------------------------
// adjoint w.r.t. the IMAGE (round 4: the trained masks of the Zhang baseline are warped - src/heads/TripletHead.py:60,69 - and their
// gradient has to reach the mask predictor): g_img[b,c,tap] += g_out[b,c,y,x] * bilinear weight, the transpose of warp_fwd_kernel's
// gather with the same taps, weights and validity.  A scatter with float atomics into the caller-zeroed g_img; a deterministic call adds
// into integer-limb entries (common.h) and a second kernel rounds them to float.  grid (w/16, h/16, B)
// ---------------------------------------------------------------------------------------------
template <bool DET>
__global__ void __launch_bounds__(256) warp_bwd_img_kernel(const double* __restrict__ H64, const float* __restrict__ g_out, int C, int h, int w,
                                                           float* __restrict__ g_img, double* __restrict__ entries) {
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
    for (int c = 0; c < C; ++c) {
        const size_t base = ((size_t)b * C + c) * h * w;
        const float g = g_out[base + (size_t)y * w + x];
        if (g == 0.0f) continue;
        auto add = [&](size_t idx, float val) {
            if constexpr (DET) bh_det_add(entries + idx * BH_ACC_WORDS, (double)val);
            else atomicAdd(g_img + idx, val);
        };
        if (v00) add(base + (size_t)y0 * w + x0, g * w00);
        if (v01) add(base + (size_t)y0 * w + x0 + 1, g * w01);
        if (v10) add(base + (size_t)(y0 + 1) * w + x0, g * w10);
        if (v11) add(base + (size_t)(y0 + 1) * w + x0 + 1, g * w11);
    }
}

__global__ void __launch_bounds__(256) warp_entries_to_float_kernel(const double* __restrict__ entries, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)bh_acc_read(entries + i * BH_ACC_WORDS, 1);
}

// rows per thread of the pool = 4 kernels (tuning hook: bh_debug_force_tile(-14 / -15, n))
BH_KNOB(g_warp_rpt_fwd, 1); BH_KNOB(g_warp_rpt_bwd, 2);       // measured (tools/hbm_path_bench.py): fwd 11.6 / 12.0 us, adjoint 14.7 / 13.4 / 15.4 us
#ifdef BH_TUNING
void bh_warp_tune(int which, int n) {
    if (which == 0) g_warp_rpt_fwd = n;
    else g_warp_rpt_bwd = n;
}
#endif

extern "C" {

int bh_warp_fwd(const float* img, const double* H64, int B, int C, int h, int w, int pool, float* out, float* cov,
                void* stream) {
    return bh_warp_fwd_f(img, H64, B, C, h, w, pool, out, cov, 0, stream);
}

int bh_warp_fwd_f(const float* img, const double* H64, int B, int C, int h, int w, int pool, float* out, float* cov,
                  int flags, void* stream) {
    if (!H64 || B < 0 || (img && !out) || (!img && !cov)) return BH_E_BADARG;
    if ((h % 16) || (w % 16) || (pool != 1 && pool != 2 && pool != 4 && pool != 8 && pool != 16 && pool != 32) || (h % pool) || (w % pool)) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    if (pool == 4 && (w % 64) == 0) {
        const int rpt = (g_warp_rpt_fwd == 2 && h % 32 == 0) ? 2 : 1;
        if (rpt == 2) hipLaunchKernelGGL(warp_fwd4_kernel<2>, dim3(w / 64, h / 32, B), dim3(256), 0, bh_stream(stream), img, H64, C, h, w, out, cov);
        else hipLaunchKernelGGL(warp_fwd4_kernel<1>, dim3(w / 64, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, C, h, w, out, cov);
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    if (pool > 16 && cov && (flags & BH_F_DETERMINISTIC)) {
        // deterministic call: the coverage by the one-writer kernel, the image (if any) by the generic kernel without a coverage output
        hipLaunchKernelGGL(warp_cov32_kernel, dim3(w / 32, h / 32, B), dim3(256), 0, bh_stream(stream), H64, h, w, cov);
        BH_LAUNCH_CHECK();
        if (!img) return BH_OK;
        cov = nullptr;
    }
    if (pool > 16 && cov) {       // quarter-window partial sums are added with atomics
        hipError_t e = hipMemsetAsync(cov, 0, (size_t)B * (h / pool) * (w / pool) * sizeof(float), bh_stream(stream));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(w / 16, h / 16, B), dim3(256), 0, bh_stream(stream), img, H64, C, h, w, pool,
                       out, cov);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_warp_bwd(const float* img, const double* H64, const float* g_out, const float* g_cov, int B, int C, int h, int w,
                int pool, double* gH, void* stream) {
    return bh_warp_bwd_f(img, H64, g_out, g_cov, B, C, h, w, pool, gH, 0, stream);
}

int bh_warp_bwd_f(const float* img, const double* H64, const float* g_out, const float* g_cov, int B, int C, int h, int w,
                  int pool, double* gH, int flags, void* stream) {
    if (!H64 || !gH || B < 0 || (g_out && !img)) return BH_E_BADARG;
    if ((h % 16) || (w % 16) || (pool != 1 && pool != 2 && pool != 4 && pool != 8 && pool != 16 && pool != 32) || (h % pool) || (w % pool)) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    if (pool == 4 && (w % 64) == 0) {
        int rpt = g_warp_rpt_bwd;
        while (rpt > 1 && h % (16 * rpt)) rpt >>= 1;
        const dim3 grid = (flags & BH_F_DETERMINISTIC) ? dim3(1, 1, B) : dim3(w / 64, h / (16 * rpt), B);
        hipStream_t s = bh_stream(stream);
        if (rpt >= 4) hipLaunchKernelGGL(warp_bwd4_kernel<4>, grid, dim3(256), 0, s, img, H64, g_out, g_cov, C, h, w, gH);
        else if (rpt == 2) hipLaunchKernelGGL(warp_bwd4_kernel<2>, grid, dim3(256), 0, s, img, H64, g_out, g_cov, C, h, w, gH);
        else hipLaunchKernelGGL(warp_bwd4_kernel<1>, grid, dim3(256), 0, s, img, H64, g_out, g_cov, C, h, w, gH);
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    hipLaunchKernelGGL(warp_bwd_kernel, (flags & BH_F_DETERMINISTIC) ? dim3(1, 1, B) : dim3(w / 16, h / 16, B), dim3(256), 0, bh_stream(stream), img,
                       H64, g_out, g_cov, C, h, w, pool, gH);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

size_t bh_warp_bwd_img_scratch_doubles(int B, int C, int h, int w, int flags) {
    return (flags & BH_F_DETERMINISTIC) ? (size_t)B * C * h * w * BH_ACC_WORDS : 0;
}

int bh_warp_bwd_img_f(const double* H64, const float* g_out, int B, int C, int h, int w, float* g_img, double* scratch, int flags,
                      void* stream) {
    if (!H64 || !g_out || !g_img || B < 0 || C < 1 || ((flags & BH_F_DETERMINISTIC) && !scratch)) return BH_E_BADARG;
    if ((h % 16) || (w % 16)) return BH_E_UNSUPPORTED;
    if (B == 0) return BH_OK;
    hipStream_t s = bh_stream(stream);
    const size_t n = (size_t)B * C * h * w;
    if (flags & BH_F_DETERMINISTIC) {
        hipError_t e = hipMemsetAsync(scratch, 0, n * BH_ACC_WORDS * sizeof(double), s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(warp_bwd_img_kernel<true>, dim3(w / 16, h / 16, B), dim3(256), 0, s, H64, g_out, C, h, w, g_img, scratch);
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(warp_entries_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scratch, n, g_img);
    } else {
        hipError_t e = hipMemsetAsync(g_img, 0, n * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(warp_bwd_img_kernel<false>, dim3(w / 16, h / 16, B), dim3(256), 0, s, H64, g_out, C, h, w, g_img, scratch);
    }
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
