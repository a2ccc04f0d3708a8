// Weight gradient as a split-K GEMM on the f32 MFMA:
//   Out[p][t][q] += sum_m  P[m][p] * Q[pix(m, t)][q]
// P is the plain operand indexed by the pixel m ([M][Np], channels contiguous), Q the operand gathered
// with the forward rule of the convolution.  Conv2d: P = gy, Q = x.  ConvTranspose2d(k == s): P = x,
// Q = gy (gathered at 2*iy+a, 2*ix+b).  The reduction runs over pixels, so both LDS tiles are
// written row = pixel (k), float4 along channels - no transpose needed - and every block owns a
// pixel range (split-K) whose partial result is added with hardware fp32 atomics.
// Tile 64 (p) x 64 (q) x 32 pixels, 4 waves as 2x2, one 32x32 accumulator each.
#include "common.h"

int bh_wgrad_x3_try(const float* x, const float* gy, float* gw, const bh_conv_desc* d, hipStream_t stream, int* taken, float* ws,
                    long long ws_bytes, long long* ws_need, const bh_bn_in* bni = nullptr, const bh_bn_adj* bna = nullptr,
                    const struct WX3Batch* extra = nullptr);    // wgrad_x3.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const float* P;
    const float* Q;
    float* Out;
    int M, Np, Nq, T;
    int Ho, Wo;              // pixel grid of m
    int Hs, Ws, Cq;          // gathered source grid / channels
    int kw, stride, pad;
    int q_nchw, p_nchw;
    long long sOp, sOt;
    int wshift, hwshift;     // log2(Wo), log2(Ho*Wo) when both are powers of two, else -1
    int use_buf;             // buffer-descriptor loads (vectorised, NHWC operands): no branches / 64-bit address math
    unsigned p_bytes, q_bytes;
    int joint;               // scalar path: GEMM columns run over (t, c) jointly, T' = 1
    int rows_per_block;
    int xcd_map;             // XCD-aware (tile, tap, split) order, see wgrad_kernel
    int noflush;             // ablation (bh_debug_force_tile(-7, 1)): skip the atomic flush
    float* partials;         // deterministic mode (wgrad_s1): per-workgroup partial tiles go here instead of into atomics
    // round 4 (wgrad_small_kernel<true>, 1x1 convs): Q is the INPUT of a training-mode BatchNorm (+ReLU) whose output the conv consumed -
    // transformed per channel while staging; bni[groups <= 2][Cq] x (scale, shift), images per group bni_ipg
    const float* bni;
    int bni_relu, bni_ipg, bni_groups;
    double* shadow;          // deterministic mode (every other kernel): integer-limb entries (common.h bh_det_add), BH_ACC_WORDS words per
                             // element of Out at the element's offset; wgrad_shadow_finalize_kernel adds them to Out
};

// one split-K contribution to Out[off]: fp32 atomic (order-dependent rounding) or, in deterministic mode, the integer limbs
__device__ __forceinline__ void wg_acc(const WgradArgs& a, long long off, float v) {
    if (a.shadow) bh_det_add(a.shadow + off * BH_ACC_WORDS, (double)v);
    else atomicAdd(a.Out + off, v);
}

#define WBK 32
#define WLD 68

__device__ __forceinline__ bool q_coord(const WgradArgs& a, int oy, int ox, int t, int& iy, int& ix) {
    const int ky = t / a.kw, kx = t - ky * a.kw;
    iy = oy * a.stride - a.pad + ky;
    ix = ox * a.stride - a.pad + kx;
    return iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws;
}

// grid: (ptiles*qtiles, T', splitK)
// BF16 = operands rounded to bf16 while staging (fp32 accumulate, v_mfma_f32_32x32x16_bf16): tiles are stored
// [channel][pixel + pad] so that the 8 consecutive k (= pixels) of a fragment are one ds_read_b128; lanes run along
// pixels in the loader so the transposed 2-byte LDS writes of a wave are contiguous.
template <bool VEC, bool BF16 = false>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs a) {
    constexpr int WLH = WBK + 8;
    __shared__ __attribute__((aligned(16))) float As[BF16 ? 64 * WLH / 2 : WBK * WLD];
    __shared__ __attribute__((aligned(16))) float Bs[BF16 ? 64 * WLH / 2 : WBK * WLD];
    __bf16* Ah = reinterpret_cast<__bf16*>(As);
    __bf16* Bh = reinterpret_cast<__bf16*>(Bs);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncols = a.joint ? a.T * a.Nq : a.Nq;
    const int qtiles = (ncols + 63) / 64;
    // XCD-aware work order (a.xcd_map: the split count is a multiple of 8).  Workgroups are dealt to the 8 XCDs round
    // robin in launch order (x fastest, then y, z) and every XCD has its own L2: in the natural order the nine taps of
    // one pixel range land on different XCDs and each of them fetches the same gy / x rows (6x the operand bytes from
    // HBM / Infinity Cache).  Remapped, all (tile, tap) workgroups of a pixel range run on ONE XCD, back to back.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_map) {
        const int G = gridDim.x * gridDim.y;
        const int L = bx + gridDim.x * (by + gridDim.y * bz);
        const int xcd = L & 7, idx = L >> 3;
        const int zhi = idx / G, rem = idx - zhi * G;
        by = rem / gridDim.x; bx = rem - by * gridDim.x; bz = zhi * 8 + xcd;
    }
    const int p0 = (bx / qtiles) * 64, q0 = (bx % qtiles) * 64;
    const int t = a.joint ? 0 : by;
    const int mbeg = bz * a.rows_per_block;
    const int mend = min(a.M, mbeg + a.rows_per_block);
    if (mbeg >= mend) return;

    // fp32: 16 float4 chunks x 16 pixel rows per pass, 2 passes; bf16: 32 pixel rows x 8 chunks per pass, 2 passes
    const int chunk0 = BF16 ? (tid >> 5) : (tid & 15), prow0 = BF16 ? (tid & 31) : (tid >> 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    float4 rp[2], rq[2];
    const int hw = a.Ho * a.Wo;

    __amdgpu_buffer_rsrc_t rsP, rsQ;
    if (VEC && a.use_buf) {
        rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, a.p_bytes, 0x00020000);
        rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q), 0, a.q_bytes, 0x00020000);
    }
    const int tky = t / a.kw, tkx = t - tky * a.kw;
    auto load_tile = [&](int mk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = BF16 ? chunk0 + 8 * i : chunk0;
            const int m = mk + (BF16 ? prow0 : prow0 + i * 16);
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f), vq = vp;
            if (VEC && a.use_buf) {
                const int pc = p0 + chunk * 4, qc = q0 + chunk * 4;
                int nb, rr, oy, ox;
                if (a.hwshift >= 0) { nb = m >> a.hwshift; rr = m & (hw - 1); oy = rr >> a.wshift; ox = rr & (a.Wo - 1); }
                else { nb = m / hw; rr = m - nb * hw; oy = rr / a.Wo; ox = rr - oy * a.Wo; }
                const int iy = oy * a.stride - a.pad + tky, ix = ox * a.stride - a.pad + tkx;
                const bool okp = m < mend && pc < a.Np;
                const bool okq = m < mend && qc < a.Nq && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
                const unsigned vop = ((unsigned)m * (unsigned)a.Np + (unsigned)pc) * 4u;
                const unsigned voq = ((((unsigned)nb * (unsigned)a.Hs + (unsigned)iy) * (unsigned)a.Ws + (unsigned)ix) * (unsigned)a.Cq + (unsigned)qc) * 4u;
                rp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsP, okp ? vop : 0xFFFFFFF0u, 0, 0));
                rq[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsQ, okq ? voq : 0xFFFFFFF0u, 0, 0));
                continue;
            }
            if (m < mend) {
                const int pc = p0 + chunk * 4;
                int nb, rr, oy, ox;
                if (a.hwshift >= 0) { nb = m >> a.hwshift; rr = m & (hw - 1); oy = rr >> a.wshift; ox = rr & (a.Wo - 1); }
                else { nb = m / hw; rr = m - nb * hw; oy = rr / a.Wo; ox = rr - oy * a.Wo; }
                if (a.p_nchw) {
                    const float* pp = a.P + ((size_t)nb * a.Np + pc) * hw + rr;
                    if (pc < a.Np) vp.x = pp[0];
                    if (pc + 1 < a.Np) vp.y = pp[(size_t)hw];
                    if (pc + 2 < a.Np) vp.z = pp[2 * (size_t)hw];
                    if (pc + 3 < a.Np) vp.w = pp[3 * (size_t)hw];
                } else {
                    const float* pp = a.P + (size_t)m * a.Np + pc;
                    if ((a.Np & 3) == 0 && pc + 3 < a.Np) vp = *reinterpret_cast<const float4*>(pp);
                    else {
                        if (pc < a.Np) vp.x = pp[0];
                        if (pc + 1 < a.Np) vp.y = pp[1];
                        if (pc + 2 < a.Np) vp.z = pp[2];
                        if (pc + 3 < a.Np) vp.w = pp[3];
                    }
                }
                const int qc = q0 + chunk * 4;
                if (VEC) {
                    int iy, ix;
                    if (qc < a.Nq && q_coord(a, oy, ox, t, iy, ix))
                        vq = *reinterpret_cast<const float4*>(a.Q + (((size_t)nb * a.Hs + iy) * a.Ws + ix) * a.Cq + qc);
                } else {
                    float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int col = qc + j;
                        if (col < ncols) {
                            int tt = a.joint ? col / a.Nq : t;
                            int c = a.joint ? col - tt * a.Nq : col;
                            int iy, ix;
                            if (q_coord(a, oy, ox, tt, iy, ix)) {
                                size_t off = a.q_nchw ? ((((size_t)nb * a.Cq + c) * a.Hs + iy) * a.Ws + ix)
                                                      : ((((size_t)nb * a.Hs + iy) * a.Ws + ix) * a.Cq + c);
                                e[j] = a.Q[off];
                            }
                        }
                    }
                    vq = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            rp[i] = vp; rq[i] = vq;
        }
    };

    const int kh2 = lane >> 5, l31 = lane & 31;
    load_tile(mbeg);
    for (int mk = mbeg; mk < mend; mk += WBK) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (BF16) {
                const int ch = (chunk0 + 8 * i) * 4;
                Ah[(ch + 0) * WLH + prow0] = (__bf16)rp[i].x; Ah[(ch + 1) * WLH + prow0] = (__bf16)rp[i].y;
                Ah[(ch + 2) * WLH + prow0] = (__bf16)rp[i].z; Ah[(ch + 3) * WLH + prow0] = (__bf16)rp[i].w;
                Bh[(ch + 0) * WLH + prow0] = (__bf16)rq[i].x; Bh[(ch + 1) * WLH + prow0] = (__bf16)rq[i].y;
                Bh[(ch + 2) * WLH + prow0] = (__bf16)rq[i].z; Bh[(ch + 3) * WLH + prow0] = (__bf16)rq[i].w;
            } else {
                const int row = prow0 + i * 16;
                *reinterpret_cast<float4*>(&As[row * WLD + chunk0 * 4]) = rp[i];
                *reinterpret_cast<float4*>(&Bs[row * WLD + chunk0 * 4]) = rq[i];
            }
        }
        __syncthreads();
        if (mk + WBK < mend) load_tile(mk + WBK);
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < WBK / 16; ++ks) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Ah[(wm * 32 + l31) * WLH + ks * 16 + kh2 * 8]);
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&Bh[(wn * 32 + l31) * WLH + ks * 16 + kh2 * 8]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < WBK / 2; ++kk) {
                const int k = 2 * kk + kh2;
                float av = As[k * WLD + wm * 32 + l31];
                float bv = Bs[k * WLD + wn * 32 + l31];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int q = q0 + wn * 32 + l31;
    if (q < ncols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            if (p < a.Np && !a.noflush) wg_acc(a, (long long)p * a.sOp + (long long)t * a.sOt + q, acc[r]);
        }
    }
}

// Stride-1 "same" convolutions on NHWC fp32 operands with power-of-two Wo and Ho*Wo, M % 32 == 0, Np % 64 == 0,
// Nq % 64 == 0 (every 3x3 layer of the backbones from 64 channels up): same tiling and split-K as wgrad_kernel, written
// for VALU instruction count.  On gfx950 a VALU instruction does NOT issue under a running MFMA of another wave - each one
// costs ~3.3 cycles of a 64-cycle v_mfma_f32_32x32x2_f32 slot (tools/mfma_coexec_bench.hip; SQ_VALU_MFMA_COEXEC_CYCLES is 0)
// - and the generic loader spends ~100 of them per 16 MFMAs on pixel decoding and bounds.  Here the gathered operand of
// tap (dy, dx) is the plain operand shifted by dy*Wo + dx pixels (the shift is folded into the buffer base), the
// advancing pixel index lives in the scalar offset of the buffer load, and a tap is valid iff t = m mod Ho*Wo and
// x = t mod Wo lie in workgroup-uniform windows: 8 VALU per gathered row, none for the plain rows.
// NT = taps per workgroup: 1, or 3 (kw = 3: the three taps of a kernel row share the plain operand tile - one gy load and
// one A fragment read feed three MFMAs, a barrier every 48 MFMAs per wave instead of every 16, a third less L2 traffic).
// BF16: operands rounded to bf16 while staging, fp32 accumulate on v_mfma_f32_32x32x16_bf16; tiles are stored
// [channel][pixel + pad] like wgrad_kernel's bf16 path (lanes run along pixels in the loader: one pixel row and two
// float4 channel chunks per lane, so the tap test is evaluated once per lane and step).
template <int NT, bool BF16 = false>
__global__ void __launch_bounds__(256) wgrad_s1_kernel(WgradArgs a) {
    constexpr int WLH = WBK + 8;
    __shared__ __attribute__((aligned(16))) float As[BF16 ? 64 * WLH / 2 : WBK * WLD];
    __shared__ __attribute__((aligned(16))) float Bs[NT][BF16 ? 64 * WLH / 2 : WBK * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int qtiles = a.Nq >> 6;
    const int p0 = (blockIdx.x / qtiles) * 64, q0 = (blockIdx.x % qtiles) * 64;
    const int t0 = blockIdx.y * NT;                              // first tap of this workgroup
    const int mbeg = blockIdx.z * a.rows_per_block;
    const int mend = min(a.M, mbeg + a.rows_per_block);          // a multiple of 32, like mbeg
    if (mbeg >= mend) return;
    // fp32: 16 float4 chunks x 16 pixel rows per pass, 2 passes (rows prow0, prow0 + 16, one chunk);
    // bf16: 32 pixel rows x 8 chunks per pass, 2 passes (one row, chunks chunk0 and chunk0 + 8)
    const int chunk0 = BF16 ? (tid >> 5) : (tid & 15), prow0 = BF16 ? (tid & 31) : (tid >> 4);
    const int tky = t0 / a.kw, dy = tky - a.pad, dx0 = t0 - tky * a.kw - a.pad;       // taps (dy, dx0 + j), j < NT
    const int hw = a.Ho * a.Wo;
    // validity windows: (unsigned)(t - lo_t) < n_t  and  (unsigned)(x - lo_x[j]) < n_x[j]
    const unsigned lo_t = dy < 0 ? (unsigned)(-dy * a.Wo) : 0u, n_t = (unsigned)max(0, hw - (dy < 0 ? -dy : dy) * a.Wo);   // (0: kernel taller than the map)
    unsigned lo_x[NT], n_x[NT];
    __amdgpu_buffer_rsrc_t rsQ[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int dx = dx0 + j;
        lo_x[j] = dx < 0 ? (unsigned)(-dx) : 0u;
        n_x[j] = (unsigned)max(0, a.Wo - (dx < 0 ? -dx : dx));
        const long long shift = (long long)(dy * a.Wo + dx) * a.Cq;      // elements: the tap shift is folded into the base
        rsQ[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q + shift), 0, 0x80000000u, 0x00020000);
    }
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, a.p_bytes, 0x00020000);
    unsigned vP[2], vQ[2];
    int rowi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rowi[i] = BF16 ? prow0 : prow0 + 16 * i;
        const int ch = BF16 ? chunk0 + 8 * i : chunk0;
        vP[i] = ((unsigned)rowi[i] * (unsigned)a.Np + (unsigned)(p0 + ch * 4)) * 4u;
        vQ[i] = ((unsigned)rowi[i] * (unsigned)a.Cq + (unsigned)(q0 + ch * 4)) * 4u;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float4 rp[2], rq[NT][2];
    auto load_tile = [&](int mk) {
        const unsigned sP = (unsigned)mk * (unsigned)a.Np * 4u, sQ = (unsigned)mk * (unsigned)a.Cq * 4u;     // scalar
        bool okj[2][NT];
#pragma unroll
        for (int i = 0; i < (BF16 ? 1 : 2); ++i) {
            const unsigned tt = (unsigned)(mk + rowi[i]) & (unsigned)(hw - 1);
            const unsigned xx = tt & (unsigned)(a.Wo - 1);
            const bool okt = (tt - lo_t) < n_t;
#pragma unroll
            for (int j = 0; j < NT; ++j) okj[i][j] = okt && (xx - lo_x[j]) < n_x[j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsP, vP[i], sP, 0));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                rq[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsQ[j], okj[BF16 ? 0 : i][j] ? vQ[i] : 0xFFFFFFF0u, sQ, 0));
        }
    };
    const int kh2 = lane >> 5, l31 = lane & 31;
    const int woff = prow0 * WLD + chunk0 * 4;
    const float* const ra = &As[kh2 * WLD + wm * 32 + l31];
    const int roff = kh2 * WLD + wn * 32 + l31;
    __bf16* const Ah = reinterpret_cast<__bf16*>(As);
    load_tile(mbeg);
    for (int mk = mbeg; mk < mend; mk += WBK) {
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ch = (chunk0 + 8 * i) * 4;
                Ah[(ch + 0) * WLH + prow0] = (__bf16)rp[i].x; Ah[(ch + 1) * WLH + prow0] = (__bf16)rp[i].y;
                Ah[(ch + 2) * WLH + prow0] = (__bf16)rp[i].z; Ah[(ch + 3) * WLH + prow0] = (__bf16)rp[i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    __bf16* const Bj = reinterpret_cast<__bf16*>(&Bs[j][0]);
                    Bj[(ch + 0) * WLH + prow0] = (__bf16)rq[j][i].x; Bj[(ch + 1) * WLH + prow0] = (__bf16)rq[j][i].y;
                    Bj[(ch + 2) * WLH + prow0] = (__bf16)rq[j][i].z; Bj[(ch + 3) * WLH + prow0] = (__bf16)rq[j][i].w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<float4*>(&As[woff + i * 16 * WLD]) = rp[i];
#pragma unroll
                for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(&Bs[j][woff + i * 16 * WLD]) = rq[j][i];
            }
        }
        __syncthreads();
        if (mk + WBK < mend) load_tile(mk + WBK);
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < WBK / 16; ++ks) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Ah[(wm * 32 + l31) * WLH + ks * 16 + kh2 * 8]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&reinterpret_cast<const __bf16*>(&Bs[j][0])[(wn * 32 + l31) * WLH + ks * 16 + kh2 * 8]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < WBK / 2; ++kk) {
                const float av = ra[2 * kk * WLD];
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bs[j][roff + 2 * kk * WLD], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (a.noflush) return;
    if (a.partials) {
        // deterministic mode: partial[split][tap][tile][wave][r][lane] (1 KB per wave and accumulator register: coalesced);
        // wgrad_s1_reduce_kernel adds the splits in index order
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float* const o = a.partials + ((((size_t)blockIdx.z * a.T + (t0 + j)) * gridDim.x + blockIdx.x) * 4 + wave) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r * 64] = acc[j][r];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const long long o = (long long)(p0 + wm * 32 + 4 * kh2) * a.sOp + (long long)(t0 + j) * a.sOt + q0 + wn * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) wg_acc(a, o + (long long)((r & 3) + 8 * (r >> 2)) * a.sOp, acc[j][r]);
    }
}

// Small-channel variant (min(Np, Nq) <= 32: the 16/32-channel decoder layers at 64x64 / 128x128, where K = pixels
// is huge and a 64x64 tile would be 3/4 padding): 32x32 tile per WAVE, the four waves of a workgroup split the
// workgroup's pixel range four ways and never synchronise (wave-private LDS), each adds its partial with atomics.
#define SLD 36
template <bool VEC>
__global__ void __launch_bounds__(256) wgrad_small_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float As[4][WBK * SLD];
    __shared__ __attribute__((aligned(16))) float Bs[4][WBK * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncols = a.joint ? a.T * a.Nq : a.Nq;
    const int qtiles = (ncols + 31) / 32;
    const int p0 = (blockIdx.x / qtiles) * 32, q0 = (blockIdx.x % qtiles) * 32;
    const int t = a.joint ? 0 : blockIdx.y;
    const int bbeg = blockIdx.z * a.rows_per_block;
    const int bend = min(a.M, bbeg + a.rows_per_block);
    const int rpw = (((bend - bbeg) + 3) / 4 + WBK - 1) / WBK * WBK;
    const int mbeg = bbeg + wave * rpw, mend = min(bend, mbeg + rpw);      // may be empty: the wave then only joins the reduction
    float* as = As[wave];
    float* bs = Bs[wave];
    const int chunk = lane & 7, prow0 = lane >> 3;      // 8 float4 chunks x 8 rows per pass, 4 passes
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float4 rp[4], rq[4];
    const int hw = a.Ho * a.Wo;
    float4 tq[2][2];                                    // a.bni: (scale, shift) of this thread's four Q channels, per group
    if (VEC && a.bni) {
        const int qc_ = min(q0 + chunk * 4, a.Cq - 4);
#pragma unroll
        for (int g_ = 0; g_ < 2; ++g_) {
            const float4* tb = reinterpret_cast<const float4*>(a.bni + ((size_t)(g_ < a.bni_groups ? g_ : 0) * a.Cq + qc_) * 2);
            tq[g_][0] = tb[0]; tq[g_][1] = tb[1];
        }
    }
    const float bni_lo = a.bni_relu ? 0.0f : -__builtin_inff();

    auto load_tile = [&](int mk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mk + prow0 + i * 8;
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f), vq = vp;
            if (m < mend) {
                const int pc = p0 + chunk * 4;
                int nb, rr, oy, ox;
                if (a.hwshift >= 0) { nb = m >> a.hwshift; rr = m & (hw - 1); oy = rr >> a.wshift; ox = rr & (a.Wo - 1); }
                else { nb = m / hw; rr = m - nb * hw; oy = rr / a.Wo; ox = rr - oy * a.Wo; }
                if (a.p_nchw) {
                    const float* pp = a.P + ((size_t)nb * a.Np + pc) * hw + rr;
                    if (pc < a.Np) vp.x = pp[0];
                    if (pc + 1 < a.Np) vp.y = pp[(size_t)hw];
                    if (pc + 2 < a.Np) vp.z = pp[2 * (size_t)hw];
                    if (pc + 3 < a.Np) vp.w = pp[3 * (size_t)hw];
                } else {
                    const float* pp = a.P + (size_t)m * a.Np + pc;
                    if ((a.Np & 3) == 0 && pc + 3 < a.Np) vp = *reinterpret_cast<const float4*>(pp);
                    else {
                        if (pc < a.Np) vp.x = pp[0];
                        if (pc + 1 < a.Np) vp.y = pp[1];
                        if (pc + 2 < a.Np) vp.z = pp[2];
                        if (pc + 3 < a.Np) vp.w = pp[3];
                    }
                }
                const int qc = q0 + chunk * 4;
                if (VEC) {
                    int iy, ix;
                    if (qc < a.Nq && q_coord(a, oy, ox, t, iy, ix)) {
                        vq = *reinterpret_cast<const float4*>(a.Q + (((size_t)nb * a.Hs + iy) * a.Ws + ix) * a.Cq + qc);
                        if (a.bni) {
                            const int g_ = (nb >= a.bni_ipg) ? 1 : 0;
                            const float4 t0 = g_ ? tq[1][0] : tq[0][0], t1 = g_ ? tq[1][1] : tq[0][1];
                            vq.x = __builtin_elementwise_maximum(__builtin_fmaf(vq.x, t0.x, t0.y), bni_lo);
                            vq.y = __builtin_elementwise_maximum(__builtin_fmaf(vq.y, t0.z, t0.w), bni_lo);
                            vq.z = __builtin_elementwise_maximum(__builtin_fmaf(vq.z, t1.x, t1.y), bni_lo);
                            vq.w = __builtin_elementwise_maximum(__builtin_fmaf(vq.w, t1.z, t1.w), bni_lo);
                        }
                    }
                } else {
                    float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int col = qc + j;
                        if (col < ncols) {
                            int tt = a.joint ? col / a.Nq : t;
                            int c = a.joint ? col - tt * a.Nq : col;
                            int iy, ix;
                            if (q_coord(a, oy, ox, tt, iy, ix)) {
                                size_t off = a.q_nchw ? ((((size_t)nb * a.Cq + c) * a.Hs + iy) * a.Ws + ix)
                                                      : ((((size_t)nb * a.Hs + iy) * a.Ws + ix) * a.Cq + c);
                                e[j] = a.Q[off];
                            }
                        }
                    }
                    vq = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            rp[i] = vp; rq[i] = vq;
        }
    };

    const int kh2 = lane >> 5, l31 = lane & 31;
    if (mbeg < mend) load_tile(mbeg);
    for (int mk = mbeg; mk < mend; mk += WBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = prow0 + i * 8;
            *reinterpret_cast<float4*>(&as[row * SLD + chunk * 4]) = rp[i];
            *reinterpret_cast<float4*>(&bs[row * SLD + chunk * 4]) = rq[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (mk + WBK < mend) load_tile(mk + WBK);
#pragma unroll
        for (int kk = 0; kk < WBK / 2; ++kk) {
            const int k = 2 * kk + kh2;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k * SLD + l31], bs[k * SLD + l31], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // add the four waves' partials of the same tile in LDS, then one atomic per element per workgroup
    float* red = &As[0][0];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j, r = idx >> 6, ln = idx & 63;
        const float v = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
        const int p = p0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), q = q0 + (ln & 31);
        if (p < a.Np && q < ncols) wg_acc(a, (long long)p * a.sOp + (long long)t * a.sOt + q, v);
    }
}

// Small-channel, taps-fused variant (VEC path, Np <= 32 or Nq <= 32, T = NT taps known at compile time): a wave
// owns a 32x32 (p, q) tile for ALL taps - NT accumulators - so the plain operand P (gy) is read from memory once
// per pixel instead of once per tap, and the NT gathered Q tiles of one 32-pixel step hit the same cache lines.
// grid: (ptiles*qtiles, T/NT, splitK); the four waves of a workgroup split its pixel range and never synchronise.
template <int NT>
__global__ void __launch_bounds__(256) wgrad_small_taps_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float As[4][WBK * SLD];
    __shared__ __attribute__((aligned(16))) float Bs[4][WBK * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qtiles = (a.Nq + 31) / 32;
    const int p0 = (blockIdx.x / qtiles) * 32, q0 = (blockIdx.x % qtiles) * 32;
    const int bbeg = blockIdx.z * a.rows_per_block;
    const int bend = min(a.M, bbeg + a.rows_per_block);
    const int rpw = (((bend - bbeg) + 3) / 4 + WBK - 1) / WBK * WBK;
    const int mbeg = bbeg + wave * rpw, mend = min(bend, mbeg + rpw);      // may be empty: the wave then only joins the reduction
    const int t0 = blockIdx.y * NT;                     // this workgroup's group of NT consecutive taps
    float* as = As[wave];
    float* bs = Bs[wave];
    const int chunk = lane & 7, prow0 = lane >> 3;      // 8 float4 chunks x 8 rows per pass, 4 passes
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int hw = a.Ho * a.Wo;
    const int kh2 = lane >> 5, l31 = lane & 31;
    const int pc = p0 + chunk * 4, qc = q0 + chunk * 4;
    const bool pvec = (a.Np & 3) == 0;
    __amdgpu_buffer_rsrc_t rsP, rsQ;
    if (a.use_buf) {
        rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, a.p_bytes, 0x00020000);
        rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q), 0, a.q_bytes, 0x00020000);
    }

    for (int mk = mbeg; mk < mend; mk += WBK) {
        // rows of this step handled by this lane: decode once, reuse for every tap
        int r_nb[4], r_oy[4], r_ox[4];
        bool r_ok[4];
        float4 rp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mk + prow0 + i * 8;
            r_ok[i] = m < mend;
            const int mm = r_ok[i] ? m : mbeg;
            int rr;
            if (a.hwshift >= 0) { r_nb[i] = mm >> a.hwshift; rr = mm & (hw - 1); r_oy[i] = rr >> a.wshift; r_ox[i] = rr & (a.Wo - 1); }
            else { r_nb[i] = mm / hw; rr = mm - r_nb[i] * hw; r_oy[i] = rr / a.Wo; r_ox[i] = rr - r_oy[i] * a.Wo; }
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.use_buf) {
                const unsigned vop = ((unsigned)mm * (unsigned)a.Np + (unsigned)pc) * 4u;
                vp = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsP, (r_ok[i] && pc < a.Np) ? vop : 0xFFFFFFF0u, 0, 0));
            } else if (r_ok[i]) {
                const float* pp = a.P + (size_t)mm * a.Np + pc;
                if (pvec && pc + 3 < a.Np) vp = *reinterpret_cast<const float4*>(pp);
                else {
                    if (pc < a.Np) vp.x = pp[0];
                    if (pc + 1 < a.Np) vp.y = pp[1];
                    if (pc + 2 < a.Np) vp.z = pp[2];
                    if (pc + 3 < a.Np) vp.w = pp[3];
                }
            }
            rp[i] = vp;
        }
        auto load_q = [&](int t, float4 (&rq)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iy, ix;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool ok = q_coord(a, r_oy[i], r_ox[i], t, iy, ix) && r_ok[i] && qc < a.Nq;
                if (a.use_buf) {
                    const unsigned voq = ((((unsigned)r_nb[i] * (unsigned)a.Hs + (unsigned)iy) * (unsigned)a.Ws + (unsigned)ix) * (unsigned)a.Cq + (unsigned)qc) * 4u;
                    v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsQ, ok ? voq : 0xFFFFFFF0u, 0, 0));
                } else if (ok)
                    v = *reinterpret_cast<const float4*>(a.Q + (((size_t)r_nb[i] * a.Hs + iy) * a.Ws + ix) * a.Cq + qc);
                rq[i] = v;
            }
        };
        float4 rq[4];
        load_q(t0, rq);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&as[(prow0 + i * 8) * SLD + chunk * 4]) = rp[i];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&bs[(prow0 + i * 8) * SLD + chunk * 4]) = rq[i];
            __builtin_amdgcn_wave_barrier();
            if (t + 1 < NT) load_q(t0 + t + 1, rq);
#pragma unroll
            for (int kk = 0; kk < WBK / 2; ++kk) {
                const int k = 2 * kk + kh2;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k * SLD + l31], bs[k * SLD + l31], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the four waves hold partial sums of the SAME 32x32 tile: add them in LDS first, then one atomic per element
    // per workgroup (the output is tiny - e.g. 9x32x32 - so every split-K partial lands on the same few addresses)
    float* red = &As[0][0];                      // 4 x 1024 floats
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, r = idx >> 6, ln = idx & 63;
            const float v = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
            const int p = p0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), q = q0 + (ln & 31);
            if (p < a.Np && q < a.Nq) wg_acc(a, (long long)p * a.sOp + (long long)(t0 + t) * a.sOt + q, v);
        }
    }
}

// column sums of a [M][C] matrix accumulated into out[C] (bias gradients)
// (shadow != NULL, here and in the two kernels below: deterministic mode - integer-limb entries instead of fp32 atomics)
__device__ __forceinline__ void colsum_acc(float* out, double* shadow, int c, float v) {
    if (shadow) bh_det_add(shadow + (size_t)c * BH_ACC_WORDS, (double)v);
    else atomicAdd(out + c, v);
}
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, int M, int C, int rows_per_block,
                                                     float* __restrict__ out, double* __restrict__ shadow) {
    __shared__ float sm[256];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;
    const int mbeg = blockIdx.x * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float acc = 0.f;
    if (c < C)
        for (int m = mbeg + rsub; m < mend; m += 4) acc += x[(size_t)m * C + c];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) colsum_acc(out, shadow, c, sm[threadIdx.x] + sm[threadIdx.x + 64] + sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
}

// same, float4 per lane (C % 4 == 0, C/4 divides 256): a lane keeps its four channels, 256/(C/4) rows per pass, four
// independent loads in flight per lane; streams at HBM rate where the scalar version managed ~1.5 TB/s
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ out,
                                                      double* __restrict__ shadow) {
    __shared__ float4 sm[256];
    const int LPR = C >> 2, RPP = 256 / LPR;
    const int cq = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    const int stride = gridDim.x * RPP;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = blockIdx.x * RPP + r0; m < M; m += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int mm = m + u * stride;
            v[u] = mm < M ? *reinterpret_cast<const float4*>(x + (size_t)mm * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < LPR) {
        for (int r = 1; r < RPP; ++r) {
            const float4 o = sm[threadIdx.x + r * LPR];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        colsum_acc(out, shadow, cq * 4 + 0, acc.x);
        colsum_acc(out, shadow, cq * 4 + 1, acc.y);
        colsum_acc(out, shadow, cq * 4 + 2, acc.z);
        colsum_acc(out, shadow, cq * 4 + 3, acc.w);
    }
}

// sum of channel plane c of an NCHW tensor, accumulated into out[0]
__global__ void __launch_bounds__(256) plane_sum_kernel(const float* __restrict__ x, int N, int C, int c, int hw,
                                                        float* __restrict__ out, double* __restrict__ shadow) {
    __shared__ float sm[4];
    float acc = 0.f;
    const size_t total = (size_t)N * hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t n = i / hw, r = i - n * hw;
        acc += x[(n * C + c) * hw + r];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) colsum_acc(out, shadow, 0, sm[0] + sm[1] + sm[2] + sm[3]);
}

// deterministic mode, second pass: out[i] += (float)(value of shadow entry i)
__global__ void __launch_bounds__(256) wgrad_shadow_finalize_kernel(const double* __restrict__ shadow, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] += (float)bh_acc_read(shadow + i * BH_ACC_WORDS);
}

// Second pass of the deterministic mode: Out[p][t][q] += sum over the pixel splits, in split order (fixed), of the partial
// tiles written by wgrad_s1_kernel.  One thread per output element in partial-tile order (coalesced reads and, per
// accumulator register, 128-byte coalesced writes).
__global__ void __launch_bounds__(256) wgrad_s1_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int nsplit,
                                                              int T, int ntiles, int qtiles, long long sOp, long long sOt) {
    const size_t per_split = (size_t)T * ntiles * 4096;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_split) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += partials[(size_t)z * per_split + i];
    const int lane = (int)(i & 63), r = (int)((i >> 6) & 15), wave = (int)((i >> 10) & 3);
    const size_t tt = i >> 12;
    const int tile = (int)(tt % ntiles), t = (int)(tt / ntiles);
    const int p0 = (tile / qtiles) * 64, q0 = (tile % qtiles) * 64;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh2 = lane >> 5;
    const int p = p0 + wm * 32 + 4 * kh2 + (r & 3) + 8 * (r >> 2), q = q0 + wn * 32 + l31;
    out[(long long)p * sOp + (long long)t * sOt + q] += s;
}

BH_KNOB(g_wgrad_noflush, 0); BH_KNOB(g_wgrad_xcd_map, 0);          // stride-1 fast path: 0 off, 1 one tap per workgroup, 3 a kernel row of taps (bh_debug_force_tile(-16, n); 3 is faster
                                              // back to back on the 64-channel layers - 97 vs 105 us - and 3 % slower in the training step)
BH_KNOB(g_wgrad_s3_target, 768);  // workgroups per launch of the three-tap variant (bh_debug_force_tile(-19, n))   // (XCD-aware order: measured 7-18 % slower)
BH_KNOB(g_wgrad_s1_target, 2048);  // split-K work items per launch of the stride-1 kernel (bh_debug_force_tile(-17, n))
BH_KNOB(g_wgrad_target, 4096);     // split-K work items per launch (tuning hook: bh_debug_force_tile(-3, n))

extern "C" {

// shadow (deterministic mode): Co entries of BH_ACC_WORDS doubles, any content (zeroed here), or NULL (fp32 atomics)
static int bias_grad_impl(const float* gy, float* gbias, const bh_conv_desc* d, void* stream, double* shadow) {
    if (!gy || !gbias || !d) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    // bias gradient = column sums of gy over all output pixels
    const int M = d->N * d->Ho * d->Wo, C = d->Co;
    int blocks = (M + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    int rpb = (M + blocks - 1) / blocks;
    if (shadow) {
        const hipError_t e = hipMemsetAsync(shadow, 0, (size_t)C * BH_ACC_WORDS * sizeof(double), s);
        if (e != hipSuccess) return (int)e;
    }
    if (d->out_nchw) {
        // planes: treat each (n, c) plane as a [hw][1] matrix
        const int hwp = d->Ho * d->Wo;
        for (int c = 0; c < C; ++c) {
            hipLaunchKernelGGL(plane_sum_kernel, dim3(64), dim3(256), 0, s, gy, d->N, C, c, hwp, gbias + c, shadow ? shadow + (size_t)c * BH_ACC_WORDS : nullptr);
            BH_LAUNCH_CHECK();
        }
    } else if (C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0) {
        const int rpp = 256 / (C / 4);
        int nb = (M + rpp * 8 - 1) / (rpp * 8);
        if (nb > 512) nb = 512;          // every workgroup ends with C same-address atomics (~28 ns each, serialised)
        hipLaunchKernelGGL(colsum4_kernel, dim3(nb), dim3(256), 0, s, gy, M, C, gbias, shadow);
        BH_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3(blocks, (C + 63) / 64), dim3(256), 0, s, gy, M, C, rpb, gbias, shadow);
        BH_LAUNCH_CHECK();
    }
    if (shadow) {
        hipLaunchKernelGGL(wgrad_shadow_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, shadow, gbias, (long long)C);
        BH_LAUNCH_CHECK();
    }
    return BH_OK;
}

int bh_conv_bias_grad(const float* gy, float* gbias, const bh_conv_desc* d, void* stream) {
    // (no workspace on this entry point: in deterministic mode callers take bh_conv_wgrad_det(..., gbias, ...), which has one)
    return bias_grad_impl(gy, gbias, d, stream, nullptr);
}

static int conv_wgrad_impl(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, void* stream,
                           float* ws, long long ws_bytes, long long* ws_need, const bh_bn_in* bni = nullptr);

int bh_conv_wgrad(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, void* stream) {
    return conv_wgrad_impl(x, gy, gw, gbias, d, stream, nullptr, 0, nullptr);
}

long long bh_conv_wgrad_det_bytes(const bh_conv_desc* d) {
    long long need = 0;
    float* p = reinterpret_cast<float*>(static_cast<uintptr_t>(256));
    BhQuery q; q.name[0] = 0; q.len = 0;
    BhQuery* saved = bh_query_ctx;
    bh_query_ctx = &q;                      // dry run: nothing is launched
    const int rc = conv_wgrad_impl(p, p, p, nullptr, d, nullptr, p, 0, &need);
    bh_query_ctx = saved;
    return rc == BH_OK ? need : 0;
}

// Round 6: the weight gradient whose gradient operand is a BatchNorm's adjoint applied on load (wgrad_x3_kernel<..., BNA>): `d` is the gradient
// of the BatchNorm's OUTPUT, completed - with the sums of bna->sums - by bh_conv_dgrad_bnreduce; desc->b_bound is the magnitude record of d
int bh_conv_wgrad_bnadj(const float* x, const float* d_out, float* gw, const bh_conv_desc* d, float* ws, long long ws_bytes, const bh_bn_in* bni,
                        const bh_bn_adj* bna, void* stream) {
    if (!d || !x || !d_out || !gw || !bna || !ws) return BH_E_BADARG;
    if (d->precision != 4) return BH_E_UNSUPPORTED;
    int taken = 0;
    const int rc = bh_wgrad_x3_try(x, d_out, gw, d, bh_stream(stream), &taken, ws, ws_bytes, nullptr, bni, bna);
    if (rc != BH_OK) return rc;
    return taken ? BH_OK : BH_E_UNSUPPORTED;
}

int bh_conv_wgrad_bnin(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, float* ws, long long ws_bytes,
                       const bh_bn_in* bni, void* stream) {
    if (!d || !x || !gy || !gw || !bni) return BH_E_BADARG;
    if (d->kh == 1 && d->kw == 1)        // round 4: 1x1 convs behind BatchNorm + ReLU - the small-channel kernel transforms x while staging
        return conv_wgrad_impl(x, gy, gw, gbias, d, stream, ws, ws_bytes, nullptr, bni);
    if (!ws) return BH_E_BADARG;
    if (d->precision < 2 || d->precision > 4) return BH_E_UNSUPPORTED;
    int taken = 0;
    const int rc = bh_wgrad_x3_try(x, gy, gw, d, bh_stream(stream), &taken, ws, ws_bytes, nullptr, bni);
    if (rc != BH_OK) return rc;
    if (!taken) return BH_E_UNSUPPORTED;
    return gbias ? bh_conv_bias_grad(gy, gbias, d, stream) : BH_OK;
}

int bh_conv_wgrad_det(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, float* ws, long long ws_bytes,
                      void* stream) {
    if (!ws) return BH_E_BADARG;
    return conv_wgrad_impl(x, gy, gw, gbias, d, stream, ws, ws_bytes, nullptr);
}

}  // extern "C"

static int conv_wgrad_impl(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, void* stream,
                           float* ws, long long ws_bytes, long long* ws_need, const bh_bn_in* bni) {
    if (!d || !x || !gy || !gw) return BH_E_BADARG;
    if (d->out_nchw && (d->Co > 4 || d->transposed)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    // deterministic call (BH_ROUTE_DETERMINISTIC): with a workspace EVERY shape has an order-independent form - the f32x3 and
    // stride-1 kernels store partial tiles, all others accumulate through integer-limb shadow entries; the last Co entries of
    // the workspace serve the bias gradient
    const bool det = ws && (d->route & BH_ROUTE_DETERMINISTIC);
    if (det) ws_bytes &= ~7ll;                                          // (the bias entries at the end of the workspace are doubles)
    const long long bias_bytes = det ? (long long)(d->transposed ? d->Co : d->Co) * BH_ACC_WORDS * 8 : 0;
    if (bni && (d->transposed || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->in_nchw || d->out_nchw || !bni->table ||
                bni->groups < 1 || bni->groups > 2 || d->N % bni->groups || d->Ci % 4))
        return BH_E_UNSUPPORTED;
    if (!bni && d->precision >= 2 && d->precision <= 4 && !(d->route & BH_ROUTE_WGRAD_GENERIC)) {
        // f32x3 / f32x2 arithmetic: the halo-tiled split-operand kernel (wgrad_x3.hip) takes the 3x3 layers with channels % 64 == 0
        int taken = 0;
        const int rc = bh_wgrad_x3_try(x, gy, gw, d, s, &taken, ws, ws_bytes > bias_bytes ? ws_bytes - bias_bytes : 0, ws_need);
        if (rc != BH_OK) return rc;
        if (taken) {
            if (ws_need) { *ws_need += bias_bytes; return BH_OK; }
            if (!gbias || bh_query_ctx) return BH_OK;
            return bias_grad_impl(gy, gbias, d, stream, det ? reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + ws_bytes - bias_bytes) : nullptr);
        }
    }
    WgradArgs a = {};
    a.Out = gw;
    a.T = d->kh * d->kw; a.kw = d->kw;
    bool vec;
    if (!d->transposed) {
        a.P = gy; a.Q = x;
        a.M = d->N * d->Ho * d->Wo; a.Np = d->Co; a.Nq = d->Ci;
        a.Ho = d->Ho; a.Wo = d->Wo; a.Hs = d->Hi; a.Ws = d->Wi; a.Cq = d->Ci;
        a.stride = d->stride; a.pad = d->pad; a.q_nchw = d->in_nchw;
        a.sOp = (long long)a.T * d->Ci; a.sOt = d->Ci;
        vec = !d->in_nchw && (d->Ci % 4 == 0);
        a.p_nchw = d->out_nchw;
    } else {
        if (d->kh != d->stride || d->kw != d->stride || d->pad != 0) return BH_E_UNSUPPORTED;
        a.P = x; a.Q = gy;
        a.M = d->N * d->Hi * d->Wi; a.Np = d->Ci; a.Nq = d->Co;
        a.Ho = d->Hi; a.Wo = d->Wi; a.Hs = d->Ho; a.Ws = d->Wo; a.Cq = d->Co;
        a.stride = d->stride; a.pad = 0; a.q_nchw = 0;
        a.sOp = (long long)a.T * d->Co; a.sOt = d->Co;
        vec = (d->Co % 4 == 0);
    }
    a.joint = vec ? 0 : 1;
    {
        const long long pe = (long long)a.M * a.Np, qe = (long long)d->N * a.Hs * a.Ws * a.Cq;
        a.use_buf = vec && !a.p_nchw && !a.q_nchw && (a.Np % 4 == 0) && pe < (1ll << 29) && qe < (1ll << 29);
        a.p_bytes = (unsigned)(pe * 4); a.q_bytes = (unsigned)(qe * 4);
    }
    a.wshift = a.hwshift = -1;
    for (int b = 0; b < 24; ++b) {
        if (a.Wo == (1 << b)) a.wshift = b;
        if (a.Ho * a.Wo == (1 << b)) a.hwshift = b;
    }
    if (a.wshift < 0 || a.hwshift < 0) a.wshift = a.hwshift = -1;
    const int ncols = a.joint ? a.T * a.Nq : a.Nq;
    const bool small = (a.Np <= 32 || ncols <= 32);
    const int tsz = small ? 32 : 64;
    const int tiles = ((a.Np + tsz - 1) / tsz) * ((ncols + tsz - 1) / tsz);
    const int ty = a.joint ? 1 : a.T;
    int split = (g_wgrad_target + tiles * ty - 1) / (tiles * ty);
    int maxsplit = (a.M + (small ? 1023 : 255)) / (small ? 1024 : 256);
    if (split > maxsplit) split = maxsplit;
    if (split < 1) split = 1;
    const int gran = small ? 4 * WBK : WBK;
    a.rows_per_block = (((a.M + split - 1) / split) + gran - 1) / gran * gran;
    a.noflush = g_wgrad_noflush;
    split = (a.M + a.rows_per_block - 1) / a.rows_per_block;
    if (!small && g_wgrad_xcd_map && split >= 16) {        // main kernel: a multiple of 8 splits (the surplus ones exit at once)
        split = (split + 7) / 8 * 8;
        a.xcd_map = 1;
    }
    dim3 grid(tiles, ty, split);
    if (bni) {
        if (!small || !vec) return BH_E_UNSUPPORTED;     // (the small-channel kernel only: the decoder units' 1x1 convs up to 64 -> 32)
        a.bni = bni->table; a.bni_relu = bni->relu; a.bni_groups = bni->groups; a.bni_ipg = d->N / bni->groups;
    }
    if (ws && small && !det) { if (bni) ws = nullptr; else return BH_E_UNSUPPORTED; }
    const long long numel = (long long)a.Np * a.sOp;                    // elements of gw
    auto shadow_begin = [&]() -> int {                                  // the shapes without a partial-tile form
        const long long need = numel * BH_ACC_WORDS * 8 + bias_bytes;
        if (ws_need) { *ws_need = need; return 1; }
        if (ws_bytes < need) return BH_E_BADARG;
        if (!bh_query_ctx) {
            const hipError_t e = hipMemsetAsync(ws, 0, (size_t)(numel * BH_ACC_WORDS * 8), s);
            if (e != hipSuccess) return (int)e;
        }
        a.shadow = reinterpret_cast<double*>(ws);
        return 0;
    };
    bool shadowed = false;
    if (det && small) { const int r_ = shadow_begin(); if (r_ == 1) return BH_OK; if (r_) return r_; shadowed = true; }
    if (small && vec && !a.p_nchw && (a.T == 9 || a.T == 4) && d->precision != 1) {
        // taps-fused: one launch dimension less, pixel ranges sized for ~2048 wave-level work items
        const int groups_y = (a.T == 9) ? 3 : 2;        // 3 taps (T = 9) or 2 taps (T = 4) per wave: 48 / 32 accumulator regs
        int sp = (2048 + tiles * groups_y - 1) / (tiles * groups_y);
        const int mx = (a.M + 1023) / 1024;
        if (sp > mx) sp = mx;
        if (sp < 1) sp = 1;
        a.rows_per_block = (((a.M + sp - 1) / sp) + 4 * WBK - 1) / (4 * WBK) * (4 * WBK);
        sp = (a.M + a.rows_per_block - 1) / a.rows_per_block;
        dim3 g2(tiles, groups_y, sp);
        if (bh_query("wgrad_small_taps_kernel<%d>", a.T == 9 ? 3 : 2)) return BH_OK;
        if (a.T == 9) hipLaunchKernelGGL((wgrad_small_taps_kernel<3>), g2, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_small_taps_kernel<2>), g2, dim3(256), 0, s, a);
    } else if (small) {
        if (bh_query("wgrad_small_kernel<%s>", vec ? "true" : "false")) return BH_OK;
        if (vec) hipLaunchKernelGGL((wgrad_small_kernel<true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_small_kernel<false>), grid, dim3(256), 0, s, a);
    } else if (!(d->route & BH_ROUTE_WGRAD_GENERIC) && vec && a.use_buf && d->precision >= 0 && d->precision <= 4 && !d->transposed && d->stride == 1 && d->Ho == d->Hi &&
               d->Wo == d->Wi && a.hwshift >= 0 && a.M % WBK == 0 && a.Np % 64 == 0 && a.Nq % 64 == 0 && !a.xcd_map) {
        // taps per workgroup: in bf16 mode the loop is so short that the launch is bound by its operand traffic (each tap
        // re-reads x and gy: 320 MB per launch) - three taps per workgroup share gy and a third of it goes away
        const int nt = (a.kw == 3 && ((d->route & BH_ROUTE_WGRAD_3TAP) || d->precision == 1)) ? 3 : 1;
        const int gy = ty / nt;
        int sp = (nt == 3 ? g_wgrad_s3_target : g_wgrad_s1_target) / (tiles * gy);        // round DOWN: at most `target` workgroups
        if (sp > maxsplit) sp = maxsplit;
        if (sp < 1) sp = 1;
        a.rows_per_block = (((a.M + sp - 1) / sp) + WBK - 1) / WBK * WBK;
        sp = (a.M + a.rows_per_block - 1) / a.rows_per_block;
        if (ws) {                           // deterministic mode: partial tiles + fixed-order second pass
            const long long need = (long long)sp * a.T * tiles * 4096 * 4;
            if (ws_need) *ws_need = need + bias_bytes;
            if (!ws_need && ws_bytes < need + bias_bytes) return BH_E_BADARG;
            a.partials = ws;
        }
        if (bh_query(ws ? "wgrad_s1_kernel<%d,%s>+wgrad_s1_reduce_kernel" : "wgrad_s1_kernel<%d,%s>", nt, d->precision == 1 ? "true" : "false"))
            return BH_OK;
        if (d->precision == 1 && nt == 3) hipLaunchKernelGGL((wgrad_s1_kernel<3, true>), dim3(tiles, gy, sp), dim3(256), 0, s, a);
        else if (d->precision == 1) hipLaunchKernelGGL((wgrad_s1_kernel<1, true>), dim3(tiles, gy, sp), dim3(256), 0, s, a);
        else if (nt == 3) hipLaunchKernelGGL((wgrad_s1_kernel<3, false>), dim3(tiles, gy, sp), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_s1_kernel<1, false>), dim3(tiles, gy, sp), dim3(256), 0, s, a);
        if (ws) {
            BH_LAUNCH_CHECK();
            const long long per_split = (long long)a.T * tiles * 4096;
            hipLaunchKernelGGL(wgrad_s1_reduce_kernel, dim3((unsigned)((per_split + 255) / 256)), dim3(256), 0, s, ws, gw, sp, a.T, tiles,
                               a.Nq >> 6, a.sOp, a.sOt);
        }
    } else if (ws && !det) {
        return BH_E_UNSUPPORTED;            // (outside deterministic mode the workspace form exists for the stride-1 fast path only)
    } else {
        if (det) { const int r_ = shadow_begin(); if (r_ == 1) return BH_OK; if (r_) return r_; shadowed = true; }
        if (bh_query("wgrad_kernel<%s,%s>", vec ? "true" : "false", (vec && d->precision == 1) ? "true" : "false")) return BH_OK;
        if (vec && d->precision == 1) hipLaunchKernelGGL((wgrad_kernel<true, true>), grid, dim3(256), 0, s, a);
        else if (vec) hipLaunchKernelGGL((wgrad_kernel<true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<false>), grid, dim3(256), 0, s, a);
    }
    BH_LAUNCH_CHECK();
    if (shadowed) {
        long long nb = (numel + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(wgrad_shadow_finalize_kernel, dim3((unsigned)nb), dim3(256), 0, s, a.shadow, gw, numel);
        BH_LAUNCH_CHECK();
    }
    if (gbias) return bias_grad_impl(gy, gbias, d, stream, det ? reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + ws_bytes - bias_bytes) : nullptr);
    return BH_OK;
}
