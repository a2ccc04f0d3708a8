// Shared declarations of the halo-tiled 3x3 kernels (conv3x3.hip: one workgroup per tile position; conv3x3_pc.hip: persistent
// workgroups with specialised waves).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// one split-operand MFMA step (16 channels): bf16 pieces or fp16 pieces (F16X2, common.h)
template <bool F16>
__device__ __forceinline__ f32x16 c3_mfma16(const uint4& av, const uint4& bv, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
}

// two k-planes (2 x float4) of a fragment -> the 8 bf16 operands of one v_mfma_f32_32x32x16_bf16 lane (round to nearest even)
__device__ __forceinline__ bf16x8 c3_pack_bf16(const float4& lo, const float4& hi) {
    f32x8v v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_convertvector(v, bf16x8);
}

// pixel row (0..7) of the 8x4 strip held by lane quad q = l31 >> 2: 0 1 3 2 5 4 6 7 (see a_lane in the kernel)
__device__ __forceinline__ int c3_strip_row(int q) { return q ^ (((q >> 1) ^ (q >> 2)) & 1); }

struct C3Args {
    const float* Src;
    const float* Wt;
    const float* bias;
    float* Out;
    int N, H, W;
    int Kc, Nn;            // contraction channels (= source channels), output channels
    int Cw;                // innermost dim of the weight tensor [Co][9][Cw]
    int accumulate;
    unsigned src_bytes, w_bytes, out_bytes;
    int tx_shift, tpi_shift;   // log2(tiles_x), log2(tiles_per_img) when both are powers of two, else -1
    int tiles_x, tiles_per_img, subtiles;
    const float* res;      // optional residual (same layout as Out) and ReLU applied in the epilogue (inference path)
    int relu;
    // optional (dgrad): the tensor written here is the gradient w.r.t. the OUTPUT of a training-mode BatchNorm (+ReLU);
    // the epilogue also accumulates that BatchNorm's backward sums (sum g*mask, sum g*mask*xhat) into bn_sums, so its
    // adjoint needs no separate reduce pass.  bnr_z: the BatchNorm input, bnr_y: its output (ReLU mask when a residual
    // was added; NULL -> the mask is recomputed from z), bnr_stats: forward sums (mean / variance), rows per group
    const float* bnr_z;
    const float* bnr_y;
    const double* bnr_stats;
    const float* bnr_gamma;
    const float* bnr_beta;
    float bnr_eps;
    int bnr_relu, bnr_rows;
    double* bn_sums;       // optional [groups][Nn][2]: += per-channel (sum, sum of squares) of the output (forward only)
    int imgs_per_group, groups;
    int dbg_nch;           // ablation: number of channel chunks to run (-1 = all)
    int tpb, gx_total;     // tile positions per workgroup (see the loop in the kernel), total positions along x
    int desync;            // > 0: the workgroups of every second dispatch round of 256 sleep desync x 8128 cycles before their first load, so that
                           // the two workgroups that share a CU run out of phase (one in its MFMA loop while the other loads / stores)
    int stat_acc;          // the tpb positions of a workgroup lie in ONE statistics group: their column sums are added in registers and leave
                           // with one atomic per (channel, moment) and workgroup (round 4: the full-resolution 32-channel layers launched
                           // 16384 workgroups = 8192 same-address f64 atomics of ~30 ns each per entry - 245 us of a 275 us kernel)
    int NW;                // PACKED: number of 32-wide output-channel tiles (Nn / 32)
    // X3 forward with BNI: Src is the INPUT of a training-mode BatchNorm (+ReLU) whose output this convolution consumes; the halo
    // staging applies y = max(x * scale + shift, lo) per channel on the way to LDS (padding stays zero), so that BatchNorm's
    // output tensor never exists.  bni: table[groups][Kc] x (scale, shift) of bh_bn_fwd_coeffs, copied to LDS at bni_lds
    const float* bni;
    int bni_relu, bni_ipg, bni_groups, bni_lds;
    int dbg_noload;        // ablation bits: 1 no weight-slab DMA in the loop, 2 no halo DMA in the loop (wrong results, timing only)
    int dbg_ts;            // BH_TUNING: record phase time stamps into g_c3_ts
    int det;               // deterministic mode: the statistics / backward sums go through integer limbs (common.h bh_det_add)
    // F16 (two fp16 pieces, common.h F16X2): magnitude record of Src (BH_AMAX_WORDS words); the weights' sixteen partial maxima sit
    // behind their pieces (word w_bytes / 4 of Wt, written by pack_weights_amax_kernel)
    const unsigned* amax_src;
    // round 6 (bh_bn_reduce.amax_d): the BatchNorm-backward forms also leave max |mask(d)| of the gradient they write in this magnitude record
    unsigned* amax_out;
};


// conv3x3_pc.hip: the fp16-piece launches of the 64-channel tile on persistent producer / consumer workgroups (round 5).
// `a` arrives filled by bh_conv3x3_try (tensor pointers, geometry, epilogue options); returns BH_OK after the launch, or
// BH_E_UNSUPPORTED (nothing launched) when the launch does not fit the kernel.
// force: take every launch the kernel supports (else only those where it is the faster kernel today)
int bh_conv3x3_pc_launch(C3Args& a, int dgrad, const float* bni_table, int bni_groups, int bni_relu, bool force, hipStream_t stream);
