// The pool = 4 warp kernels' per-pixel homography tap (csrc/warp.hip) - shared with the extractor stem's dgrad, whose fused form applies the
// warp's adjoint to the gradient it has just made (csrc/stem7.hip stem7_dgrad_c1_kernel<true>; round 6).
#pragma once
#include "common.h"

struct Hf { float h0, h1, h2, h3, h4, h5, h6, h7, h8; };
__device__ __forceinline__ Hf load_h(const double* __restrict__ Hm) {
    Hf f;
    f.h0 = (float)Hm[0]; f.h1 = (float)Hm[1]; f.h2 = (float)Hm[2]; f.h3 = (float)Hm[3]; f.h4 = (float)Hm[4];
    f.h5 = (float)Hm[5]; f.h6 = (float)Hm[6]; f.h7 = (float)Hm[7]; f.h8 = (float)Hm[8];
    return f;
}

struct Tap4 {
    float u, v, iz, fx, fy;
    float wx0, wx1, wy0, wy1;      // per-axis bilinear weights, 0 where that tap column / row is outside the image
    bool vx0, vx1, vy0, vy1, guard;
    unsigned o00, o01, o10, o11;   // byte offsets of the taps inside one image plane; 0xFFFFFFFF (= out of range for the
                                   // buffer load, which then returns 0) for a tap outside the image
};

__device__ __forceinline__ Tap4 make_tap4(const Hf& H, int x, int y, int w, int h) {
#pragma clang fp contract(off)      // (only the fused multiply-adds that are written out: u - floor(u) after u = qx * iz is one the compiler would make)
    Tap4 t;
    const float fxp = (float)x, fyp = (float)y;
    // (spelled as fused multiply-adds in ONE order: every kernel that includes this header - forward, adjoint, the stem dgrad's folded
    //  adjoint - gets bitwise the same coordinates, so they agree on which side of an integer coordinate a pixel falls: the bilinear
    //  derivative jumps there, and with a coordinate left to the compiler's contraction one pixel in ~10^5 took the other branch)
    const float qx = __builtin_fmaf(H.h0, fxp, __builtin_fmaf(H.h1, fyp, H.h2)), qy = __builtin_fmaf(H.h3, fxp, __builtin_fmaf(H.h4, fyp, H.h5)),
                qz = __builtin_fmaf(H.h6, fxp, __builtin_fmaf(H.h7, fyp, H.h8));
    t.guard = !(fabsf(qz) > 1e-8f);
    float r = __builtin_amdgcn_rcpf(qz);
    r = __builtin_fmaf(__builtin_fmaf(-qz, r, 1.0f), r, r);     // one Newton step: within an ulp of 1/qz, exact for qz = 1
    t.iz = t.guard ? 1.0f : r;
    t.u = qx * t.iz;
    t.v = qy * t.iz;
    const float x0f = floorf(t.u), y0f = floorf(t.v);
    t.fx = t.u - x0f;
    t.fy = t.v - y0f;
    // clamp in float first so that wild coordinates (inf / nan / huge) become plain out-of-bounds integers
    const int x0 = (int)fminf(fmaxf(x0f, -2.0f), (float)w), y0 = (int)fminf(fmaxf(y0f, -2.0f), (float)h);
    t.vx0 = (unsigned)x0 < (unsigned)w; t.vx1 = (unsigned)(x0 + 1) < (unsigned)w;
    t.vy0 = (unsigned)y0 < (unsigned)h; t.vy1 = (unsigned)(y0 + 1) < (unsigned)h;
    t.wx0 = t.vx0 ? 1.0f - t.fx : 0.0f; t.wx1 = t.vx1 ? t.fx : 0.0f;
    t.wy0 = t.vy0 ? 1.0f - t.fy : 0.0f; t.wy1 = t.vy1 ? t.fy : 0.0f;
    const int w4 = 4 * w;
    const int o = y0 * w4 + 4 * x0;
    t.o00 = (t.vx0 && t.vy0) ? (unsigned)o : 0xFFFFFFFFu;
    t.o01 = (t.vx1 && t.vy0) ? (unsigned)(o + 4) : 0xFFFFFFFFu;
    t.o10 = (t.vx0 && t.vy1) ? (unsigned)(o + w4) : 0xFFFFFFFFu;
    t.o11 = (t.vx1 && t.vy1) ? (unsigned)(o + w4 + 4) : 0xFFFFFFFFu;
    return t;
}

// the blend of the four taps, spelled out for the same reason: warp_fwd4_kernel and the stem forward that makes its own warped pixels
// (stem7_fwd_f16_kernel<1, true>) produce bitwise the same image
__device__ __forceinline__ void tap_weights(const Tap4& t, float& w00, float& w01, float& w10, float& w11, float& wsum) {
#pragma clang fp contract(off)
    w00 = t.wx0 * t.wy0; w01 = t.wx1 * t.wy0; w10 = t.wx0 * t.wy1; w11 = t.wx1 * t.wy1;
    wsum = ((w00 + w01) + w10) + w11;      // the warped all-ones mask at this pixel
}
__device__ __forceinline__ float tap_wsum(float w00, float w01, float w10, float w11) {
#pragma clang fp contract(off)
    return ((w00 + w01) + w10) + w11;
}
__device__ __forceinline__ float tap_blend(float p00, float p01, float p10, float p11, float w00, float w01, float w10, float w11) {
#pragma clang fp contract(off)
    float acc = p00 * w00;
    acc = __builtin_fmaf(p01, w01, acc);
    acc = __builtin_fmaf(p10, w10, acc);
    acc = __builtin_fmaf(p11, w11, acc);
    return acc;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ldtap(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

