// Implicit-GEMM convolution on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulate, 64 FLOP/clk/SIMD).
//
//   Out[m][n] = sum_{tap t} sum_{c < Kc}  Src[pix(m, t)][c] * Bw[t][c][n]      (+ bias[n])
//
// m runs over the output-side pixel grid [N, Ho, Wo]; pix() gathers from an NHWC source tensor with
// either the forward rule  (iy = oy*s - p + ky)  or the adjoint rule  (iy = (oy + p - ky)/s when
// divisible).  With the right (Src, Bw strides, epilogue) this one kernel is Conv2d forward, Conv2d
// dgrad, ConvTranspose2d(k == s) forward (M = input pixels, N = taps*Co, scatter epilogue) and its
// dgrad.  Feature maps are NHWC so a tap's Kc channels are one contiguous 16 B-vectorisable run.
//
// Block = 256 threads (4 waves), tile BM=128 pixels x BN channels x BK=32 (or 16) reduction channels.
// A and B tiles are staged k-major in LDS ([BK][BM+1]: the transposed ds_write_b32 pattern and the
// ds_read_b32 fragment reads are both bank-conflict free), global loads for tile i+1 are issued
// before the MFMAs of tile i (register prefetch), one barrier pair per tile.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* Src;
    const float* Bw;
    const float* bias;
    float* Out;
    int M, Nn, Kc, T;
    int Ho, Wo;            // output-side pixel grid
    int Hs, Ws, Cs;        // source grid and its channel count
    int kw, stride, pad, sshift;   // sshift = log2(stride) when it is a power of two, else -1
    int adjoint;           // 0: forward gather, 1: adjoint (dgrad of a strided conv) gather
    int src_nchw;          // scalar path, channels are planes
    long long sBt, sBc, sBn;
    int b_kcontig;         // 1: sBc == 1, 0: sBn == 1
    int epi;               // 0: NHWC [M][Nn]; 1: ConvTranspose scatter; 2: NCHW
    int accumulate;        // Out += result
    int use_buf;           // buffer-descriptor loads with per-row tap masks (vectorised path; see load_tile)
    unsigned src_bytes, bw_bytes;
    unsigned out_bytes;    // bytes of the output tensor when below 2^31 (32-bit epilogue addressing through a buffer descriptor), else 0
    long long src_elems, bw_elems;   // tensor sizes (host side: buffer descriptors)
    int prio;              // experiment: s_setprio(1) around the MFMA block
    int bf16;              // operands rounded to bf16 (fp32 accumulate) where the vectorised path applies
    int x3;                // operands cut exactly into three bf16 pieces, six products (fp32 accuracy; X3 template form) where that path applies
    int ek, eC;            // scatter: kernel (== stride) and real channel count (Nn = ek*ek*eC)
    int ewshift, ehwshift; // log2(Wo), log2(Ho*Wo) when both are powers of two (epilogue pixel decode by shifts), else -1
    int etap0;             // scatter: tap index of column 0 (a parity class of a strided dgrad writes ONE tap position)
    const float* res;      // optional residual added in the epilogue (same layout as Out)
    int relu;              // epilogue ReLU (inference: BatchNorm folded into the weights, activation fused)
    // optional: per-channel (sum, sum of squares) of the output accumulated into the padded BatchNorm sums table
    // (forward statistics of the BatchNorm that follows).  bn_rpg = GEMM rows per statistics group, a multiple of
    // every BM (host check), so a tile never straddles groups; bn_C = channels of the table (eC for the scatter epilogue)
    double* bn_sums;
    int bn_det;              // deterministic mode: integer-limb accumulation (common.h bh_det_add)
    int bn_rpg, bn_groups, bn_C;
    // round 4: BatchNorm-on-load for 1x1 / stride-1 convs (the 1x1 conv of a decoder unit behind BatchNorm + ReLU): Src is the INPUT of that
    // BatchNorm, the A operand is transformed y = max(x * scale + shift, lo) per channel between the global load and the LDS store
    // (buffer-loader, fp32 C4 layout only).  bni: table[groups][Kc] x (scale, shift); rows of an M tile lie in one group (bni_rpg % BM == 0)
    const float* bni;
    int bni_relu, bni_rpg;
    unsigned* amax_out;      // optional magnitude record of the output (common.h F16X2): max |value stored|, one atomic max per workgroup
    // round 6: nz > 0 - ONE launch for nz problems that differ only in their B operand and tap set (the parity classes of a stride-2
    // dgrad, bh_conv_dgrad_s2): workgroup z = blockIdx.z takes (Bw, T, kw, etap0, bw_bytes) from these tables
    int nz;
    const float* Bw_z[4];
    long long sBn_z[4];
    int T_z[4], kw_z[4], etap0_z[4];
    unsigned bw_bytes_z[4];
};


__device__ __forceinline__ bool src_coord(const GemmArgs& a, int kw, int oy, int ox, int t, int& iy, int& ix) {
    const int ky = t / kw, kx = t - ky * kw;
    if (!a.adjoint) {
        iy = oy * a.stride - a.pad + ky;
        ix = ox * a.stride - a.pad + kx;
    } else {
        int ty = oy + a.pad - ky, tx = ox + a.pad - kx;
        if (ty < 0 || tx < 0) return false;
        if (a.stride > 1) {
            if ((ty % a.stride) || (tx % a.stride)) return false;
            ty /= a.stride; tx /= a.stride;
        }
        iy = ty; ix = tx;
    }
    return iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws;
}

// BUF: the buffer-descriptor loader (GemmArgs::use_buf) as a compile-time switch, so that the instantiation that runs
// does not also carry the pointer-based loader's prologue (VALU work is paid at MFMA price on this chip)
template <int BM, int BN, int BK, bool VEC, bool BF16 = false, bool BUF = false, bool X3 = false>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const GemmArgs a) {
    // the B operand and tap set of THIS workgroup's problem (a.nz > 0: one of nz problems in the launch, chosen by blockIdx.z).  Scalars
    // picked with constant indices: writing to the by-value argument struct, or indexing its tables with blockIdx.z, puts the whole
    // struct into scratch memory - every later a.field a scratch load (measured: every instantiation of this kernel 1.5-3x slower)
    const float* zBw = a.Bw;
    int zT = a.T, zkw = a.kw, zetap0 = a.etap0;
    unsigned zbw_bytes = a.bw_bytes;
    long long zsBn = a.sBn;
    if (a.nz) {
        const int z = blockIdx.z;
#define BH_ZSEL(f) (z == 0 ? a.f[0] : z == 1 ? a.f[1] : z == 2 ? a.f[2] : a.f[3])
        zBw = BH_ZSEL(Bw_z); zT = BH_ZSEL(T_z); zkw = BH_ZSEL(kw_z); zetap0 = BH_ZSEL(etap0_z); zbw_bytes = BH_ZSEL(bw_bytes_z); zsBn = BH_ZSEL(sBn_z);
#undef BH_ZSEL
    }
    constexpr int WN = (BN >= 64) ? 2 : 1;         // waves along N
    constexpr int WM = 4 / WN;                     // waves along M
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");
    constexpr int TM = BM / (WM * 32);             // 32x32 MFMA tiles per wave along M
    constexpr int TN = BN / (WN * 32);
    constexpr int LDA = BM + 1;
    constexpr int LDB_K = BN + 1, LDB_N = BN + 4;
    constexpr int CH = BK / 4;                     // float4 chunks per tile row
    constexpr int AROWS = 256 / CH;                // rows covered per pass
    constexpr int AIT = BM / AROWS;
    constexpr int BIT_K = (BN + AROWS - 1) / AROWS;   // K-contiguous B: rows = n
    // BF16 operand mode (mixed precision: fp32 tensors in HBM, operands rounded to bf16 while staging, fp32
    // accumulate on v_mfma_f32_32x32x16_bf16): tiles are row-major [row][BK + 8] bf16 - the 80-byte row stride makes
    // the ds_read_b128 fragment reads bank-conflict free and the float4 -> bf16x4 staging writes need no transpose.
    static_assert(!BF16 || (VEC && BK == 32), "bf16 operand mode needs the vectorised loader and BK = 32");
    // X3 (round 6; with BF16): fp32 accuracy on that pipe - every operand is cut EXACTLY into three bf16 pieces while staging (common.h X3,
    // the arithmetic of the 3x3 layers' f32x3 mode), three LDS planes per operand, six products per product (total order <= 2, small
    // ones first): 2.67x less matrix-pipe time than v_mfma_f32_32x32x2_f32.  The layers that run here in the fp32-accurate modes - strided
    // 3x3 convs and their parity-class dgrads, 128-channel transposed convs, 1x1 convs off the streaming kernel - were bound by the fp32 pipe.
    static_assert(!X3 || BF16, "the three-piece mode is a form of the bf16 operand mode");
    constexpr int NPL = X3 ? 3 : 1;
    constexpr int LDH = BK + 8;
    constexpr int PA = BM * LDH, PB = BN * LDH;    // bf16 elements per plane
    // C4 layout (vectorised fp32 path): tiles are stored as float4 k-chunks, [BK/4][rows + 1] x float4.  One
    // ds_write_b128 stages a loaded float4 (the +1 row of padding spreads the 8 chunk-lanes of a row over all 32
    // banks), one ds_read_b128 feeds FOUR MFMAs (the two half-waves take chunks 2c and 2c+1, so MFMA step j
    // multiplies k = 8c + j and 8c + 4 + j; any pairing of k is valid for a sum over k).
    constexpr bool C4 = VEC && !BF16;
    constexpr bool B4 = C4 || X3;                  // N-contiguous B: the lanes-along-k thread mapping (conflict-free transposing LDS writes)
    constexpr int A_FLOATS = BF16 ? NPL * (BM * LDH / 2) : (C4 ? CH * (BM + 1) * 4 : BK * LDA);
    constexpr int B_FLOATS = BF16 ? NPL * (BN * LDH / 2) : (C4 ? CH * (BN + 1) * 4 : BK * LDB_N);
    __shared__ __attribute__((aligned(16))) float As[A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[B_FLOATS];
    __bf16* Ah = reinterpret_cast<__bf16*>(As);
    __bf16* Bh = reinterpret_cast<__bf16*>(Bs);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // ---- per-thread A rows (fixed for the whole K loop) ----
    const int a_chunk = tid % CH, a_row0 = tid / CH;
    int a_n[AIT], a_oy[AIT], a_ox[AIT];
    bool a_ok[AIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        int m = m0 + a_row0 + i * AROWS;
        a_ok[i] = m < a.M;
        int mm = a_ok[i] ? m : 0;
        if (a.ehwshift >= 0) {                     // power-of-two output grid: shifts (an integer division is ~35 VALU
            a_n[i] = mm >> a.ehwshift;             //  instructions, and VALU work does not overlap other waves' MFMAs)
            const int r = mm & ((1 << a.ehwshift) - 1);
            a_oy[i] = r >> a.ewshift;
            a_ox[i] = r & ((1 << a.ewshift) - 1);
        } else {
            int hw = a.Ho * a.Wo;
            a_n[i] = mm / hw;
            int r = mm - a_n[i] * hw;
            a_oy[i] = r / a.Wo;
            a_ox[i] = r - a_oy[i] * a.Wo;
        }
    }
    // ---- B rows ----
    const int bk_chunk = tid % CH, bk_row0 = tid / CH;                    // K-contiguous orientation
    constexpr int NCH = BN / 4;                                           // N-contiguous orientation
    const int bn_chunk = tid % NCH, bn_row0 = tid / NCH;
    constexpr int BROWS_N = 256 / NCH;
    constexpr int BIT_N = (BK + BROWS_N - 1) / BROWS_N;
    // C4 layout, N-contiguous B: lanes run along k (conflict-free scalar LDS writes), 256/BK n-chunks per pass
    constexpr int C4_CPP = 256 / BK;
    constexpr int BIT_N4 = (NCH + C4_CPP - 1) / C4_CPP;
    const int b4_k = tid % BK, b4_c0 = tid / BK;
    constexpr int BIT01 = (BIT_K > BIT_N) ? BIT_K : BIT_N;
    constexpr int BIT = (BIT01 > BIT_N4) ? BIT01 : BIT_N4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int kchunks = (a.Kc + BK - 1) / BK;
    const int ktiles = VEC ? zT * kchunks : (zT * a.Kc + BK - 1) / BK;   // scalar path: K linear over (t, c)

    float4 ra[AIT], rb[BIT];
    float4 bt0 = make_float4(1.f, 0.f, 1.f, 0.f), bt1 = bt0;      // a.bni: (scale, shift) of the four channels of this thread's A chunk

    // division-free tile walk for the vectorised path: (tap, ky, kx, channel chunk) of the NEXT tile to load
    int nx_t = 0, nx_ky = 0, nx_kx = 0, nx_c0 = 0;
    // per-row anchors: forward iy = ay + ky ; adjoint ty = ay - ky (then >> sshift when strided)
    int a_ay[AIT], a_ax[AIT];
    unsigned a_base[AIT];                              // pixel index of (n, 0, 0) in the source grid
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        a_ay[i] = a.adjoint ? a_oy[i] + a.pad : a_oy[i] * a.stride - a.pad;
        a_ax[i] = a.adjoint ? a_ox[i] + a.pad : a_ox[i] * a.stride - a.pad;
        a_base[i] = (unsigned)a_n[i] * (unsigned)(a.Hs * a.Ws);
    }
    const float* b_rowptr[BIT_K];
#pragma unroll
    for (int i = 0; i < BIT_K; ++i) {
        int n = n0 + bk_row0 + i * AROWS;
        b_rowptr[i] = ((bk_row0 + i * AROWS) < BN && n < a.Nn) ? zBw + (long long)n * zsBn + bk_chunk * 4 : nullptr;
    }

    // ---- buffer-load fast path (use_buf): every A row keeps ONE 32-bit byte offset (its anchor pixel) and a bit
    // mask of the taps that fall inside the source image; a tile's address is that offset + a wave-uniform tap/
    // chunk offset, invalid taps get an out-of-range offset and the buffer descriptor returns zeros.  No per-load
    // bounds arithmetic, no 64-bit address math, no divergent branches around the loads.
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_voff[AIT];
    unsigned long long a_mask[AIT];
    unsigned bK_voff[BIT_K], bN_voff[BIT];
    __amdgpu_buffer_rsrc_t rsA, rsB;
    if constexpr (VEC && BUF) {
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Src), 0, a.src_bytes, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zBw), 0, zbw_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int pix = (int)a_base[i] + a_ay[i] * a.Ws + a_ax[i];
            a_voff[i] = ((unsigned)pix * (unsigned)a.Cs + (unsigned)(a_chunk * 4)) * 4u;
            unsigned long long mk = 0;
            int t = 0;
            for (int ky = 0; ky * zkw < zT; ++ky)
                for (int kx = 0; kx < zkw; ++kx, ++t) {
                    const int iy = a.adjoint ? a_ay[i] - ky : a_ay[i] + ky, ix = a.adjoint ? a_ax[i] - kx : a_ax[i] + kx;
                    if (a_ok[i] && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws) mk |= 1ull << t;
                }
            a_mask[i] = mk;
        }
#pragma unroll
        for (int i = 0; i < BIT_K; ++i) {
            const int n = n0 + bk_row0 + i * AROWS;
            bK_voff[i] = ((bk_row0 + i * AROWS) < BN && n < a.Nn) ? (unsigned)((long long)n * zsBn + bk_chunk * 4) * 4u : OOB;
        }
        if constexpr (C4 || BF16) {
#pragma unroll
            for (int i = 0; i < BIT; ++i) {
                if (B4) {
                    const int chn = b4_c0 + i * C4_CPP, n = n0 + chn * 4;
                    bN_voff[i] = (i < BIT_N4 && chn < NCH && n < a.Nn) ? (unsigned)((long long)b4_k * a.sBc + n) * 4u : OOB;
                } else {
                    const int kr = bn_row0 + i * BROWS_N, n = n0 + bn_chunk * 4;
                    bN_voff[i] = (i < BIT_N && kr < BK && n < a.Nn) ? (unsigned)((long long)kr * a.sBc + n) * 4u : OOB;
                }
            }
        }
    }

    auto load_tile = [&](int kt) {
        if constexpr (VEC && BUF) {
            if constexpr (C4 || BF16) {
            const int t = nx_t, c0 = nx_c0;
            const int tap_pix = nx_ky * a.Ws + nx_kx;
            const unsigned toffA = (unsigned)((a.adjoint ? -tap_pix : tap_pix) * a.Cs + c0) * 4u;
            const bool cokA = c0 + a_chunk * 4 < a.Kc;
            if (a.bni) {
                const float4* tb = reinterpret_cast<const float4*>(a.bni + ((size_t)(m0 / a.bni_rpg) * a.Kc + (cokA ? c0 + a_chunk * 4 : 0)) * 2);
                bt0 = tb[0]; bt1 = tb[1];
            }
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                const bool ok = ((a_mask[i] >> t) & 1ull) && cokA;
                const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? a_voff[i] + toffA : OOB, 0, 0);
                ra[i] = __builtin_bit_cast(float4, v);
            }
            if (a.b_kcontig) {
                const unsigned toffB = (unsigned)((long long)t * a.sBt + c0) * 4u;
                const bool cokB = c0 + bk_chunk * 4 < a.Kc;
#pragma unroll
                for (int i = 0; i < BIT_K; ++i) {
                    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, (cokB && bK_voff[i] != OOB) ? bK_voff[i] + toffB : OOB, 0, 0);
                    rb[i] = __builtin_bit_cast(float4, v);
                }
            } else {
                const unsigned toffB = (unsigned)((long long)t * a.sBt + (long long)c0 * a.sBc) * 4u;
                constexpr int NB = B4 ? BIT_N4 : BIT_N;
                const bool cokB = c0 + (B4 ? b4_k : 0) < a.Kc;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    bool ok = cokB && bN_voff[i] != OOB;
                    if (!B4) ok = ok && (c0 + bn_row0 + i * BROWS_N < a.Kc);
                    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? bN_voff[i] + toffB : OOB, 0, 0);
                    rb[i] = __builtin_bit_cast(float4, v);
                }
            }
            nx_c0 += BK;
            if (nx_c0 >= a.Kc) {
                nx_c0 = 0; ++nx_t;
                if (++nx_kx == zkw) { nx_kx = 0; ++nx_ky; }
            }
            }
        } else if (VEC) {
            const int t = nx_t, ky = nx_ky, kx = nx_kx, c0 = nx_c0;
            const int c = c0 + a_chunk * 4;
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                int iy, ix;
                bool ok = a_ok[i] && c < a.Kc;
                if (!a.adjoint) {
                    iy = a_ay[i] + ky; ix = a_ax[i] + kx;
                } else {
                    iy = a_ay[i] - ky; ix = a_ax[i] - kx;
                    if (a.stride > 1) {
                        if (a.sshift >= 0) {
                            const int msk = a.stride - 1;
                            ok = ok && iy >= 0 && ix >= 0 && !((iy | ix) & msk);
                            iy >>= a.sshift; ix >>= a.sshift;
                        } else {
                            ok = ok && iy >= 0 && ix >= 0 && !(iy % a.stride) && !(ix % a.stride);
                            iy /= a.stride; ix /= a.stride;
                        }
                    }
                }
                ok = ok && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const unsigned pix = a_base[i] + (unsigned)(iy * a.Ws + ix);
                    v = *reinterpret_cast<const float4*>(a.Src + (size_t)pix * a.Cs + c);
                }
                ra[i] = v;
            }
            if (a.b_kcontig) {
                const long long toff = (long long)t * a.sBt + c0;
                const bool cok = c0 + bk_chunk * 4 < a.Kc;
#pragma unroll
                for (int i = 0; i < BIT_K; ++i) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (b_rowptr[i] && cok) v = *reinterpret_cast<const float4*>(b_rowptr[i] + toff);
                    rb[i] = v;
                }
            } else if constexpr (B4) {
                const int c2 = c0 + b4_k;
#pragma unroll
                for (int i = 0; i < BIT_N4; ++i) {
                    const int chn = b4_c0 + i * C4_CPP, n = n0 + chn * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (chn < NCH && c2 < a.Kc) {
                        const float* p = zBw + t * a.sBt + (long long)c2 * a.sBc + n;
                        if (n + 3 < a.Nn) v = *reinterpret_cast<const float4*>(p);
                        else {
                            if (n < a.Nn) v.x = p[0];
                            if (n + 1 < a.Nn) v.y = p[1];
                            if (n + 2 < a.Nn) v.z = p[2];
                        }
                    }
                    rb[i] = v;
                }
            } else {
                const int n = n0 + bn_chunk * 4;
#pragma unroll
                for (int i = 0; i < BIT_N; ++i) {
                    int kr = bn_row0 + i * BROWS_N;
                    int c2 = c0 + kr;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kr < BK && c2 < a.Kc) {
                        const float* p = zBw + t * a.sBt + (long long)c2 * a.sBc + n;
                        if (n + 3 < a.Nn) v = *reinterpret_cast<const float4*>(p);
                        else {
                            if (n < a.Nn) v.x = p[0];
                            if (n + 1 < a.Nn) v.y = p[1];
                            if (n + 2 < a.Nn) v.z = p[2];
                        }
                    }
                    rb[i] = v;
                }
            }
            // advance the walk
            nx_c0 += BK;
            if (nx_c0 >= a.Kc) {
                nx_c0 = 0; ++nx_t;
                if (++nx_kx == zkw) { nx_kx = 0; ++nx_ky; }
            }
        } else {
            // scalar path (tiny channel counts / NCHW network input): k = t*Kc + c decoded per element
            const int Ktot = zT * a.Kc;
            const int k0 = kt * BK + a_chunk * 4;
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = k0 + j;
                    if (a_ok[i] && k < Ktot) {
                        int t = k / a.Kc, c = k - t * a.Kc, iy, ix;
                        if (src_coord(a, zkw, a_oy[i], a_ox[i], t, iy, ix)) {
                            size_t off = a.src_nchw ? ((((size_t)a_n[i] * a.Cs + c) * a.Hs + iy) * a.Ws + ix)
                                                    : ((((size_t)a_n[i] * a.Hs + iy) * a.Ws + ix) * a.Cs + c);
                            e[j] = a.Src[off];
                        }
                    }
                }
                ra[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
            const int kb0 = kt * BK + bk_chunk * 4;
#pragma unroll
            for (int i = 0; i < BIT_K; ++i) {
                int n = n0 + bk_row0 + i * AROWS;
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                if ((bk_row0 + i * AROWS) < BN && n < a.Nn) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int k = kb0 + j;
                        if (k < Ktot) {
                            int t = k / a.Kc, c = k - t * a.Kc;
                            e[j] = zBw[t * a.sBt + (long long)c * a.sBc + (long long)n * zsBn];
                        }
                    }
                }
                rb[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    auto store_tile = [&]() {
        if constexpr (BF16 && X3) {
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                const int row = a_row0 + i * AROWS;
                uint2 h, m, l;
                bh_split4(ra[i], h, m, l);
                *reinterpret_cast<uint2*>(&Ah[row * LDH + a_chunk * 4]) = h;
                *reinterpret_cast<uint2*>(&Ah[PA + row * LDH + a_chunk * 4]) = m;
                *reinterpret_cast<uint2*>(&Ah[2 * PA + row * LDH + a_chunk * 4]) = l;
            }
            if (a.b_kcontig) {
#pragma unroll
                for (int i = 0; i < BIT_K; ++i) {
                    const int row = bk_row0 + i * AROWS;
                    if (row < BN) {
                        uint2 h, m, l;
                        bh_split4(rb[i], h, m, l);
                        *reinterpret_cast<uint2*>(&Bh[row * LDH + bk_chunk * 4]) = h;
                        *reinterpret_cast<uint2*>(&Bh[PB + row * LDH + bk_chunk * 4]) = m;
                        *reinterpret_cast<uint2*>(&Bh[2 * PB + row * LDH + bk_chunk * 4]) = l;
                    }
                }
            } else {
                // (lanes along k, as the fp32 C4 layout stages this orientation: consecutive lanes write consecutive halfs of one row)
                unsigned short* const Bu = reinterpret_cast<unsigned short*>(Bh);
#pragma unroll
                for (int i = 0; i < BIT_N4; ++i) {
                    const int chn = b4_c0 + i * C4_CPP;
                    if (chn < NCH) {
                        const float e[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            unsigned short h, m, l;
                            bh_split1(e[q], h, m, l);
                            Bu[(chn * 4 + q) * LDH + b4_k] = h;
                            Bu[PB + (chn * 4 + q) * LDH + b4_k] = m;
                            Bu[2 * PB + (chn * 4 + q) * LDH + b4_k] = l;
                        }
                    }
                }
            }
            return;
        } else if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                const int row = a_row0 + i * AROWS;
                f32x4v v = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                *reinterpret_cast<bf16x4*>(&Ah[row * LDH + a_chunk * 4]) = __builtin_convertvector(v, bf16x4);
            }
            if (a.b_kcontig) {
#pragma unroll
                for (int i = 0; i < BIT_K; ++i) {
                    const int row = bk_row0 + i * AROWS;
                    if (row < BN) {
                        f32x4v v = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
                        *reinterpret_cast<bf16x4*>(&Bh[row * LDH + bk_chunk * 4]) = __builtin_convertvector(v, bf16x4);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < BIT_N; ++i) {
                    const int kr = bn_row0 + i * BROWS_N;
                    if (kr < BK) {
                        Bh[(bn_chunk * 4 + 0) * LDH + kr] = (__bf16)rb[i].x;
                        Bh[(bn_chunk * 4 + 1) * LDH + kr] = (__bf16)rb[i].y;
                        Bh[(bn_chunk * 4 + 2) * LDH + kr] = (__bf16)rb[i].z;
                        Bh[(bn_chunk * 4 + 3) * LDH + kr] = (__bf16)rb[i].w;
                    }
                }
            }
            return;
        }
        if constexpr (C4) {
            if (a.bni) {
                const float lo = a.bni_relu ? 0.0f : -__builtin_inff();
#pragma unroll
                for (int i = 0; i < AIT; ++i) {
                    if (m0 + a_row0 + i * AROWS < a.M) {          // (rows past the end stay zero)
                        ra[i].x = __builtin_elementwise_maximum(__builtin_fmaf(ra[i].x, bt0.x, bt0.y), lo);
                        ra[i].y = __builtin_elementwise_maximum(__builtin_fmaf(ra[i].y, bt0.z, bt0.w), lo);
                        ra[i].z = __builtin_elementwise_maximum(__builtin_fmaf(ra[i].z, bt1.x, bt1.y), lo);
                        ra[i].w = __builtin_elementwise_maximum(__builtin_fmaf(ra[i].w, bt1.z, bt1.w), lo);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                const int row = a_row0 + i * AROWS;
                *reinterpret_cast<float4*>(&As[(a_chunk * (BM + 1) + row) * 4]) = ra[i];
            }
            if (a.b_kcontig) {
#pragma unroll
                for (int i = 0; i < BIT_K; ++i) {
                    const int row = bk_row0 + i * AROWS;
                    if (row < BN) *reinterpret_cast<float4*>(&Bs[(bk_chunk * (BN + 1) + row) * 4]) = rb[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < BIT_N4; ++i) {
                    const int chn = b4_c0 + i * C4_CPP;
                    if (chn < NCH) {
                        float* q = &Bs[((b4_k >> 2) * (BN + 1) + chn * 4) * 4 + (b4_k & 3)];
                        q[0] = rb[i].x; q[4] = rb[i].y; q[8] = rb[i].z; q[12] = rb[i].w;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int row = a_row0 + i * AROWS;
            As[(a_chunk * 4 + 0) * LDA + row] = ra[i].x;
            As[(a_chunk * 4 + 1) * LDA + row] = ra[i].y;
            As[(a_chunk * 4 + 2) * LDA + row] = ra[i].z;
            As[(a_chunk * 4 + 3) * LDA + row] = ra[i].w;
        }
        if (!VEC || a.b_kcontig) {
#pragma unroll
            for (int i = 0; i < BIT_K; ++i) {
                const int row = bk_row0 + i * AROWS;
                if (row < BN) {
                    Bs[(bk_chunk * 4 + 0) * LDB_K + row] = rb[i].x;
                    Bs[(bk_chunk * 4 + 1) * LDB_K + row] = rb[i].y;
                    Bs[(bk_chunk * 4 + 2) * LDB_K + row] = rb[i].z;
                    Bs[(bk_chunk * 4 + 3) * LDB_K + row] = rb[i].w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BIT_N; ++i) {
                const int kr = bn_row0 + i * BROWS_N;
                if (kr < BK) *reinterpret_cast<float4*>(&Bs[kr * LDB_N + bn_chunk * 4]) = rb[i];
            }
        }
    };

    const int ldb = (!VEC || a.b_kcontig) ? LDB_K : LDB_N;
    const int kh2 = lane >> 5, l31 = lane & 31;

    load_tile(0);
    for (int kt = 0; kt < ktiles; ++kt) {
        if (!(a.prio & 4) || kt == 0) store_tile();           // (ablation bits, bh_debug_force_tile(-1, x): 2 = no reloads, 4 = no restaging)
        __syncthreads();
        if (kt + 1 < ktiles && !(a.prio & 2)) load_tile(kt + 1);
        if constexpr (BF16 && X3) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        af[i][pc] = *reinterpret_cast<const bf16x8*>(&Ah[pc * PA + ((wm * TM + i) * 32 + l31) * LDH + ks * 16 + kh2 * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        bf[j][pc] = *reinterpret_cast<const bf16x8*>(&Bh[pc * PB + ((wn * TN + j) * 32 + l31) * LDH + ks * 16 + kh2 * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        // the six products of total order <= 2, small ones first (0 = hi, 1 = mid, 2 = lo)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&Ah[((wm * TM + i) * 32 + l31) * LDH + ks * 16 + kh2 * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(&Bh[((wn * TN + j) * 32 + l31) * LDH + ks * 16 + kh2 * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (C4) {
#pragma unroll
            for (int c2 = 0; c2 < CH / 2; ++c2) {
                const int ch = 2 * c2 + kh2;
                float4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const float4*>(&As[(ch * (BM + 1) + (wm * TM + i) * 32 + l31) * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const float4*>(&Bs[(ch * (BN + 1) + (wn * TN + j) * 32 + l31) * 4]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        } else {
        if (a.prio & 1) __builtin_amdgcn_s_setprio(1);
        // fragments of KG k-pairs are read from LDS up front, then their MFMAs issue back to back (the compiler
        // otherwise interleaves each ds_read with a full lgkmcnt(0) wait right in front of its MFMA)
        constexpr int KG = (TM * TN >= 4) ? 4 : 8;
#pragma unroll
        for (int kg = 0; kg < BK / 2; kg += KG) {
            float av[KG][TM], bv[KG][TN];
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const int k = 2 * (kg + q) + kh2;
#pragma unroll
                for (int i = 0; i < TM; ++i) av[q][i] = As[k * LDA + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[q][j] = Bs[k * ldb + (wn * TN + j) * 32 + l31];
            }
#pragma unroll
            for (int q = 0; q < KG; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], bv[q][j], acc[i][j], 0, 0, 0);
        }
        if (a.prio & 1) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) ----
    // element offset = row part (pixel of m) + column part (channel / tap of n): the 64-bit pixel arithmetic is done once
    // per row and once per column, one add per element; columns are the inner loop so that the stores of adjacent taps /
    // channels of one pixel leave back to back
    double st1[TN], st2[TN];
    float bvj[TN];
    bool nok[TN];
    int cco[TN], ctap[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        st1[j] = 0.0; st2[j] = 0.0;
        const int n = n0 + (wn * TN + j) * 32 + l31;
        nok[j] = n < a.Nn;
        int co = n, tap = 0;
        if (a.epi == 1) { tap = n / a.eC; co = n - tap * a.eC; tap += zetap0; }
        cco[j] = co; ctap[j] = tap;
        bvj[j] = (a.bias && nok[j]) ? a.bias[co] : 0.0f;
    }
    const bool want_stats = a.bn_sums != nullptr;
    float vmax = 0.f;                                   // a.amax_out: max |v| of this thread's stores
    if (a.out_bytes) {
        // 32-bit addressing through buffer descriptors (out-of-range rows / columns get an out-of-range offset: the
        // store is dropped, a load returns 0); statistics as float partial sums per fragment quad, totals in double,
        // and only when a BatchNorm table was passed
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.Out), 0, a.out_bytes, 0x00020000);
        unsigned cp[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
            if (a.epi == 0) cp[j] = (unsigned)n * 4u;
            else if (a.epi == 1) {
                const int ay = ctap[j] / a.ek, ax = ctap[j] - ay * a.ek;
                cp[j] = ((unsigned)(ay * (a.Wo * a.ek) + ax) * (unsigned)a.eC + (unsigned)cco[j]) * 4u;
            } else cp[j] = (unsigned)n * (unsigned)(a.Ho * a.Wo) * 4u;
            if (!nok[j]) cp[j] = 0xFFFFFFF0u;
        }
        const bool extra = a.accumulate || a.res;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float q1[TN], q2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) { q1[j] = 0.f; q2[j] = 0.f; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = rq * 4 + c;
                    const int m = m0 + (wm * TM + i) * 32 + c + 8 * rq + 4 * kh2;
                    unsigned rp;
                    if (a.epi == 0) {
                        rp = (unsigned)m * (unsigned)a.Nn * 4u;
                    } else {
                        int nb, oy, ox;
                        if (a.ehwshift >= 0) {
                            nb = m >> a.ehwshift;
                            const int rr = m & ((1 << a.ehwshift) - 1);
                            oy = rr >> a.ewshift; ox = rr & ((1 << a.ewshift) - 1);
                        } else {
                            const int hw = a.Ho * a.Wo;
                            nb = m / hw;
                            const int rr = m - nb * hw;
                            oy = rr / a.Wo; ox = rr - oy * a.Wo;
                        }
                        if (a.epi == 1) rp = (((unsigned)(nb * (a.Ho * a.ek) + oy * a.ek) * (unsigned)(a.Wo * a.ek) + (unsigned)(ox * a.ek)) * (unsigned)a.eC) * 4u;
                        else rp = ((unsigned)nb * (unsigned)a.Nn * (unsigned)(a.Ho * a.Wo) + (unsigned)(oy * a.Wo + ox)) * 4u;
                    }
                    const bool mok = m < a.M;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned off = (mok && nok[j]) ? rp + cp[j] : 0xFFFFFFF0u;
                        float v = acc[i][j][r] + bvj[j];
                        if (extra) {
                            if (a.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, off, 0, 0));
                            if (a.res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off, 0, 0));
                        }
                        if (a.relu) v = fmaxf(v, 0.0f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, off, 0, 0);
                        if (mok && nok[j]) vmax = fmaxf(vmax, fabsf(v));
                        if (want_stats) {
                            const float vs = (mok && nok[j]) ? v : 0.f;
                            q1[j] += vs; q2[j] = __builtin_fmaf(vs, vs, q2[j]);
                        }
                    }
                }
                if (want_stats) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) { st1[j] += (double)q1[j]; st2[j] += (double)q2[j]; }
                }
            }
        }
    } else {
    size_t cpart[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        if (a.epi == 0) cpart[j] = (size_t)n;
        else if (a.epi == 1) {
            const int ay = ctap[j] / a.ek, ax = ctap[j] - ay * a.ek;
            cpart[j] = ((size_t)ay * (a.Wo * a.ek) + ax) * a.eC + cco[j];
        } else cpart[j] = (size_t)n * a.Ho * a.Wo;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            if (m >= a.M) continue;
            size_t rpart;
            if (a.epi == 0) {
                rpart = (size_t)m * a.Nn;
            } else {
                int nb, oy, ox;
                if (a.ehwshift >= 0) {
                    nb = m >> a.ehwshift;
                    const int rr = m & ((1 << a.ehwshift) - 1);
                    oy = rr >> a.ewshift; ox = rr & ((1 << a.ewshift) - 1);
                } else {
                    const int hw = a.Ho * a.Wo;
                    nb = m / hw;
                    const int rr = m - nb * hw;
                    oy = rr / a.Wo; ox = rr - oy * a.Wo;
                }
                if (a.epi == 1) rpart = (((size_t)nb * (a.Ho * a.ek) + oy * a.ek) * (a.Wo * a.ek) + ox * a.ek) * a.eC;
                else rpart = (size_t)nb * a.Nn * a.Ho * a.Wo + (size_t)oy * a.Wo + ox;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!nok[j]) continue;
                const size_t off = rpart + cpart[j];
                float v = acc[i][j][r] + bvj[j];
                if (a.accumulate) v += a.Out[off];
                if (a.res) v += a.res[off];
                if (a.relu) v = fmaxf(v, 0.0f);
                a.Out[off] = v;
                vmax = fmaxf(vmax, fabsf(v));
                st1[j] += (double)v;
                st2[j] += (double)v * (double)v;
            }
        }
    }
    }
    if (a.amax_out) {
        __shared__ float sm_amax[4];
        bh_amax_commit(a.amax_out, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
    }
    if (a.bn_sums) {
        // column sums of the tile: half-waves merged by a shuffle, the WM waves of a column range through LDS (the
        // operand tiles are dead after the last barrier of the k loop), one f64 atomic per (column, moment)
        double* red = reinterpret_cast<double*>(As);            // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const double s1 = st1[j] + __shfl_xor(st1[j], 32, 64), s2 = st2[j] + __shfl_xor(st2[j], 32, 64);
            if (kh2 == 0) {
                const int col = (wn * TN + j) * 32 + l31;
                red[(wm * BN + col) * 2] = s1;
                red[(wm * BN + col) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int col = tid >> 1, mom = tid & 1, n = n0 + col;
            if (n < a.Nn) {
                double tot = 0.0;
#pragma unroll
                for (int w = 0; w < WM; ++w) tot += red[(w * BN + col) * 2 + mom];
                const int co = a.epi == 1 ? n % a.eC : n;
                bh_acc_add(&a.bn_sums[bn_sum_index(0, a.bn_groups, m0 / a.bn_rpg, a.bn_C, co, mom)], tot, a.bn_det);
            }
        }
    }
}

template <int BM, int BN, int BK, bool VEC, bool BF16 = false, bool X3 = false>
static int launch(const GemmArgs& a, hipStream_t s) {
    if (bh_query(X3 ? "conv_gemm_kernel<%d,%d,%d,%s,%s,%s,true>" : "conv_gemm_kernel<%d,%d,%d,%s,%s,%s>", BM, BN, BK, VEC ? "true" : "false",
                 BF16 ? "true" : "false", (VEC && a.use_buf) ? "true" : "false"))
        return BH_OK;
    dim3 grid((a.M + BM - 1) / BM, (a.Nn + BN - 1) / BN, a.nz ? a.nz : 1);
    if constexpr (VEC) {
        if (a.use_buf) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, VEC, BF16, true, X3>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, VEC, BF16, false, X3>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, VEC, BF16, false, X3>), grid, dim3(256), 0, s, a);
    }
    BH_LAUNCH_CHECK();
    return BH_OK;
}

// Tile choice: the largest tile that still gives the 256 CUs about two workgroups each; the small-spatial
// layers (8x8x256ch, 16x16x128ch at 2B = 128) otherwise launch only 128-256 workgroups.
BH_KNOB(g_s2_merge, 1);            // (tuning build, hook -49: the parity classes of a stride-2 dgrad in one launch / one launch each)
BH_KNOB(g_gemm_x3, 1);             // (tuning build, hook -48: the three-piece form of this kernel in the fp32-accurate modes off / on)
#ifdef BH_TUNING
extern int g_wgrad_target, g_wgrad_noflush, g_wgrad_xcd_map, g_wgrad_s1, g_wgrad_s1_target, g_wgrad_s3_target;
static int g_force_bm = 0, g_force_bn = 0, g_prio = 0, g_no_buf = 0;     // tuning hook (bh_debug_force_tile), 0 = automatic
#else
static constexpr int g_force_bm = 0, g_force_bn = 0, g_prio = 0, g_no_buf = 0;
#endif

static int dispatch(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    a.prio = g_prio;
    a.sshift = -1;
    for (int b = 0; b < 8; ++b)
        if (a.stride == (1 << b)) a.sshift = b;
    a.ewshift = a.ehwshift = -1;
    for (int b = 0; b < 16; ++b)
        if (a.Wo == (1 << b)) a.ewshift = b;
    for (int b = 0; b < 31; ++b)
        if ((long long)a.Ho * a.Wo == (1ll << b)) a.ehwshift = b;
    if (a.ewshift < 0) a.ehwshift = -1;
    if (a.M <= 0 || a.Nn <= 0) return BH_OK;
    const bool vec = !a.src_nchw && (a.Kc % 4 == 0) && (a.Cs % 4 == 0);
    if (a.bni && (!vec || a.bf16)) return BH_E_UNSUPPORTED;
    a.use_buf = vec && !(a.adjoint && a.stride > 1) && (a.Nn % 4 == 0) && a.T <= 64 && a.src_elems > 0 &&
                a.src_elems < (1ll << 29) && a.bw_elems > 0 && a.bw_elems < (1ll << 29) && (a.Kc % 32 == 0) && !g_no_buf;
    if (a.bni && !a.use_buf) return BH_E_UNSUPPORTED;
    a.src_bytes = (unsigned)(a.src_elems * 4);
    a.bw_bytes = (unsigned)(a.bw_elems * 4);
    for (int z = 0; z < a.nz; ++z) a.bw_bytes_z[z] = (unsigned)((long long)a.Nn * a.T_z[z] * a.Kc * 4);      // (class-packed [Nn][T][Kc])
    {
        const long long oe = a.epi == 1 ? (long long)a.M * a.ek * a.ek * a.eC : (long long)a.M * a.Nn;
        a.out_bytes = (oe > 0 && oe < (1ll << 29)) ? (unsigned)(oe * 4) : 0u;
    }
    if (vec && a.bf16 && (a.Kc % 32) == 0) {
        // bf16 operands: the MFMA is 16x faster, the kernel is bound by staging traffic -> widest N tile that fits
        int bm = g_force_bm, bn = g_force_bn;
        if (!bm) {      // measured (tools/bf16_microbench.py): 64-wide N tiles; 64-row tiles while the grid is small
            if (a.Nn <= 32) { bm = 128; bn = 32; }
            else { bn = 64; bm = (((a.M + 127) / 128) * ((a.Nn + 63) / 64) < 1024) ? 64 : 128; }
        }
        if (bm == 128 && bn == 128) return launch<128, 128, 32, true, true>(a, s);
        if (bm == 64 && bn == 128) return launch<64, 128, 32, true, true>(a, s);
        if (bm == 64 && bn == 64) return launch<64, 64, 32, true, true>(a, s);
        if (bm == 128 && bn == 64) return launch<128, 64, 32, true, true>(a, s);
        if (bm == 128 && bn == 32) return launch<128, 32, 32, true, true>(a, s);
        return BH_E_UNSUPPORTED;
    }
    if (vec && a.x3 && !a.bf16 && !a.bni && (a.Kc % 32) == 0 && g_gemm_x3) {
        // fp32-accurate modes (bh_conv_desc.precision 2 / 4), round 6: three exact bf16 pieces per operand, six products - the tiles of the
        // fp32 path (64-row tiles: latency hiding over reuse)
        if (a.Nn > 64) {
            const long long nt = (a.Nn + 127) / 128, mt64 = (a.M + 63) / 64;
            if (a.Nn >= 256 && mt64 * nt >= 512) return launch<64, 128, 32, true, true, true>(a, s);
            return launch<64, 64, 32, true, true, true>(a, s);
        }
        if (a.Nn > 32) return launch<64, 64, 32, true, true, true>(a, s);
        return launch<128, 32, 32, true, true, true>(a, s);
    }
    if (vec && g_force_bm && (a.Kc % 32) == 0) {
        const int bm = g_force_bm, bn = g_force_bn;
        if (bm == 128 && bn == 128) return launch<128, 128, 32, true>(a, s);
        if (bm == 64 && bn == 128) return launch<64, 128, 32, true>(a, s);
        if (bm == 64 && bn == 64) return launch<64, 64, 32, true>(a, s);
        if (bm == 128 && bn == 64) return launch<128, 64, 32, true>(a, s);
        if (bm == 128 && bn == 32) return launch<128, 32, 32, true>(a, s);
        if (bm == 6464 && (a.Kc % 64) == 0) return launch<64, 64, 64, true>(a, s);        // BK = 64 experiments
        if (bm == 64128 && (a.Kc % 64) == 0) return launch<64, 128, 64, true>(a, s);
        return BH_E_UNSUPPORTED;
    }
    if (!vec) {
        if (a.Nn > 64) return launch<128, 128, 32, false>(a, s);
        if (a.Nn > 32) return launch<128, 64, 32, false>(a, s);
        return launch<128, 32, 32, false>(a, s);
    }
    // measured on MI355X (tools/conv_microbench.py): the 64-row tiles (4-5 waves/SIMD resident) beat the 128-row
    // ones (2-3 waves/SIMD) on every 3x3 layer of the network - latency hiding matters more than tile reuse
    // at the fp32 MFMA rate
    const bool k16 = (a.Kc % 32) != 0 && a.Kc <= 16;
    const long long mt64 = (a.M + 63) / 64;
    const bool k64 = (a.Kc % 64) == 0 && a.Kc >= 256;      // BK = 64 only pays on the long-K (256-channel) layers
    if (a.Nn > 64) {
        if (k16) return launch<128, 128, 16, true>(a, s);
        const long long nt = (a.Nn + 127) / 128;
        if (a.Nn >= 256 && mt64 * nt >= 512) return launch<64, 128, 32, true>(a, s);
        return k64 ? launch<64, 64, 64, true>(a, s) : launch<64, 64, 32, true>(a, s);
    }
    if (a.Nn > 32) {
        if (k16) return launch<128, 64, 16, true>(a, s);
        return launch<64, 64, 32, true>(a, s);
    }
    return k16 ? launch<128, 32, 16, true>(a, s) : launch<128, 32, 32, true>(a, s);
}

// ---------------------------------------------------------------------------------------------
// dgrad of a strided stem conv with ONE input channel (the extractor's 7x7/2 on a grayscale patch,
// PerceptualHead.py:52-55): gx[n][iy][ix] = sum_{ky,kx valid} sum_c gy[n][(iy+p-ky)/s][(ix+p-kx)/s][c] w[c][ky][kx].
// As a GEMM this has N = 1 (31/32 of an MFMA tile wasted, 49 mostly-empty taps); here a workgroup owns one
// output parity class (iy%s, ix%s) so its valid taps are wave-uniform, 16 lanes x float4 read one source
// pixel's channels coalesced, weights sit transposed in LDS, and the channel sum is a 16-lane xor-shuffle.
// ---------------------------------------------------------------------------------------------
template <int KK, int ST>     // KK = kernel size, ST = stride (compile-time: the tap loops unroll and their loads batch)
__global__ void __launch_bounds__(256) stem_dgrad_c1_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                            float* __restrict__ gx, int N, int Hi, int Wi, int Ho, int Wo,
                                                            int Co, int pad) {
    extern __shared__ __attribute__((aligned(16))) float wT[];     // [k*k][Co]
    for (int i = threadIdx.x; i < KK * KK * Co; i += 256) {
        int t = i / Co, c = i - t * Co;
        wT[i] = w[c * KK * KK + t];
    }
    __syncthreads();
    constexpr int NT = (KK + ST - 1) / ST;          // taps per axis for one parity class
    const int cls = blockIdx.y, py = cls / ST, px = cls % ST;
    const int Ha = (Hi - py + ST - 1) / ST, Wa = (Wi - px + ST - 1) / ST;   // pixels of this class
    const int LP = Co / 4;                        // lanes per pixel (Co = 64 -> 16)
    const int PPW = 64 / LP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LP, cl = lane % LP;
    const int ky0 = (py + pad) % ST, kx0 = (px + pad) % ST;
    const long long total = (long long)N * Ha * Wa;
    const long long wave_global = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
    for (long long q0 = wave_global * PPW; q0 < total; q0 += nwaves * PPW) {
        const long long q = q0 + sub;
        const bool ok = q < total;
        const long long qq = ok ? q : 0;
        const int n = (int)(qq / ((long long)Ha * Wa));
        const int r = (int)(qq - (long long)n * Ha * Wa);
        const int a = r / Wa, b = r - a * Wa;
        const int iy = a * ST + py, ix = b * ST + px;
        // source rows/cols for tap index (ty, tx): oy = (iy + pad - ky0)/ST - ty
        const int oyb = (iy + pad - ky0) / ST, oxb = (ix + pad - kx0) / ST;
        float4 gv[NT][NT];
#pragma unroll
        for (int ty = 0; ty < NT; ++ty)
#pragma unroll
            for (int tx = 0; tx < NT; ++tx) {
                const int oy = oyb - ty, ox = oxb - tx;
                const bool v = (ky0 + ty * ST < KK) && (kx0 + tx * ST < KK) && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
                gv[ty][tx] = v ? *reinterpret_cast<const float4*>(gy + (((size_t)n * Ho + oy) * Wo + ox) * Co + cl * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < NT; ++ty)
#pragma unroll
            for (int tx = 0; tx < NT; ++tx) {
                const int ky = ky0 + ty * ST, kx = kx0 + tx * ST;
                if (ky < KK && kx < KK) {
                    const float4 ww = *reinterpret_cast<const float4*>(&wT[(ky * KK + kx) * Co + cl * 4]);
                    const float4 g = gv[ty][tx];
                    acc += g.x * ww.x + g.y * ww.y + g.z * ww.z + g.w * ww.w;
                }
            }
        for (int off = 1; off < LP; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (ok && cl == 0) gx[((size_t)n * Hi + iy) * Wi + ix] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// col2im for the same 1-input-channel stem dgrad, second half of the two-step form:
//   step 1 (a plain 1x1 implicit GEMM on the MFMA): Tm[m][t] = sum_c gy[m][c] * w[c][t]   for every SOURCE pixel m, tap t
//   step 2 (this kernel): gx[n][iy][ix] = sum over the valid taps of Tm[(n, (iy+p-ky)/s, (ix+p-kx)/s)][ky*k+kx]
// Every element of Tm is used exactly once, so the total traffic is |gy| + 2|Tm| instead of 49/4 cached re-reads of
// gy per output pixel.  A workgroup owns one output parity class, lanes run along the class row (consecutive
// source pixels), ldT = padded row length of Tm.
// ---------------------------------------------------------------------------------------------
template <int KK, int ST, int CI>
__global__ void __launch_bounds__(256) col2im_c1_kernel(const float* __restrict__ Tm, float* __restrict__ gx, int N, int Hi,
                                                        int Wi, int Ho, int Wo, int pad, int ldT) {
    // CI = stem input channels (1: grayscale patch, 3: RGB patch); Tm column = tap*CI + c; gx is NCHW [N][CI][Hi][Wi]
    constexpr int NT = (KK + ST - 1) / ST;
    const int cls = blockIdx.y, py = cls / ST, px = cls % ST;
    const int Ha = (Hi - py + ST - 1) / ST, Wa = (Wi - px + ST - 1) / ST;
    const int ky0 = (py + pad) % ST, kx0 = (px + pad) % ST;
    const long long total = (long long)N * Ha * Wa;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int n = (int)(q / ((long long)Ha * Wa));
        const int r = (int)(q - (long long)n * Ha * Wa);
        const int a = r / Wa, b = r - a * Wa;
        const int iy = a * ST + py, ix = b * ST + px;
        const int oyb = (iy + pad - ky0) / ST, oxb = (ix + pad - kx0) / ST;
        float v[NT][NT][CI];
#pragma unroll
        for (int ty = 0; ty < NT; ++ty)
#pragma unroll
            for (int tx = 0; tx < NT; ++tx) {
                const int oy = oyb - ty, ox = oxb - tx, ky = ky0 + ty * ST, kx = kx0 + tx * ST;
                const bool ok = ky < KK && kx < KK && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
                const float* tp = Tm + (((size_t)n * Ho + oy) * Wo + ox) * ldT + (ky * KK + kx) * CI;
#pragma unroll
                for (int c = 0; c < CI; ++c) v[ty][tx][c] = ok ? tp[c] : 0.f;
            }
#pragma unroll
        for (int c = 0; c < CI; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int ty = 0; ty < NT; ++ty)
#pragma unroll
                for (int tx = 0; tx < NT; ++tx) acc += v[ty][tx][c];
            gx[(((size_t)n * CI + c) * Hi + iy) * Wi + ix] = acc;
        }
    }
}

#ifdef BH_TUNING
void bh_conv3x3_tune(int disable, int min_blocks);
void bh_stem7_tune(int disable);
void bh_warp_tune(int which, int n);
void bh_bn_tune(int cap);
void bh_wgrad_x3_tune(int what, int v);
#endif
int bh_conv3x3_try(const float* src, const float* w, const float* bias, float* out, const bh_conv_desc* d, int dgrad,
                   int accumulate, hipStream_t stream, int* taken, double* bn_sums, int groups, const float* res = nullptr,
                   int relu = 0, const bh_bn_reduce* bnr = nullptr, const bh_bn_in* bni = nullptr);
int bn_launch_stats(const float* x, int groups, int rows, int C, double* sums, hipStream_t s, int det);
int bh_stem7_try(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, int relu,
                 hipStream_t stream, int* taken, double* bn_sums = nullptr, int groups = 1);
int bh_pointwise_try(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, hipStream_t stream, int* taken,
                     double* bn_sums, int groups, float* amax_y, const bh_bn_in* bni);
#ifdef BH_TUNING
void bh_pointwise_tune(int what, int v);
#endif

static int check_desc(const bh_conv_desc* d) {
    if (!d) return BH_E_BADARG;
    if (d->transposed) {
        if (d->kh != d->stride || d->kw != d->stride || d->pad != 0) return BH_E_UNSUPPORTED;
        if (d->Ho != d->Hi * d->stride || d->Wo != d->Wi * d->stride) return BH_E_BADARG;
        if (d->in_nchw || d->out_nchw) return BH_E_UNSUPPORTED;
    } else {
        if (d->Ho != (d->Hi + 2 * d->pad - d->kh) / d->stride + 1) return BH_E_BADARG;
        if (d->Wo != (d->Wi + 2 * d->pad - d->kw) / d->stride + 1) return BH_E_BADARG;
    }
    return BH_OK;
}

extern "C" {

int bh_col2im_c1(const float* Tm, float* gx, const bh_conv_desc* d, int ldT, void* stream) {
    if (!Tm || !gx || !d) return BH_E_BADARG;
    if (d->transposed || (d->Ci != 1 && d->Ci != 3) || d->kh != 7 || d->kw != 7 || d->stride != 2 || ldT < 49 * d->Ci)
        return BH_E_UNSUPPORTED;
    if (d->Ci > 1 && !d->in_nchw) return BH_E_UNSUPPORTED;      // multi-channel stems read/write the NCHW network input
    if (d->Ci == 1)
        hipLaunchKernelGGL((col2im_c1_kernel<7, 2, 1>), dim3(1024, 4), dim3(256), 0, bh_stream(stream), Tm, gx, d->N, d->Hi,
                           d->Wi, d->Ho, d->Wo, d->pad, ldT);
    else
        hipLaunchKernelGGL((col2im_c1_kernel<7, 2, 3>), dim3(1024, 4), dim3(256), 0, bh_stream(stream), Tm, gx, d->N, d->Hi,
                           d->Wi, d->Ho, d->Wo, d->pad, ldT);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

#ifdef BH_TUNING
int bh_debug_force_tile(int bm, int bn) {
    if (bm == -1) { g_prio = bn; return BH_OK; }            // (-1, bits): 1 s_setprio, 2 no global reloads, 4 no LDS restaging (ablations)
    if (bm == -2) { g_no_buf = bn; return BH_OK; }          // (-2, 1): disable the buffer-load fast path
    if (bm == -3) { g_wgrad_target = bn; return BH_OK; }    // (-3, n): wgrad split-K work items per launch
    // (-4 / -5 / -6 / -16 of round 1 - kernel routing - are now per-call bits of bh_conv_desc.route)
    if (bm == -7) { g_wgrad_noflush = bn; return BH_OK; }
    if (bm == -10) { g_wgrad_xcd_map = bn; return BH_OK; }              // (-10, 0|1): XCD-aware wgrad work order off / on
    if (bm == -19) { g_wgrad_s3_target = bn; return BH_OK; }             // (-19, n): workgroups per launch of the three-tap wgrad variant
    if (bm == -17) { g_wgrad_s1_target = bn; return BH_OK; }             // (-17, n): its split-K work items per launch
    if (bm == -40) { bh_conv3x3_tune(500 + (bn > 0 ? (bn < 400 ? bn : 400) : 0), 0); return BH_OK; }   // (-40, 0|1): 3x3 kernel phase time stamps (bh_debug_read_c3_stamps)
    if (bm == -43) { bh_conv3x3_tune(300 + (bn ? 1 : 0), 0); return BH_OK; }   // (-43, 0|1): 32-channel 3x3 launches without statistics walk several positions per workgroup off / on
    if (bm == -45) { bh_conv3x3_tune(2000 + (bn < 4000 ? bn : 3999), 0); return BH_OK; }   // (-45, n): halo kernel, workgroups per statistics / BatchNorm-sums launch (64-channel tile; default 512)
    if (bm == -46) { bh_conv3x3_tune(6000 + (bn < 4000 ? bn : 3999), 0); return BH_OK; }   // (-46, n): the same for the 32-channel tile (default 512)
    if (bm == -44) { bh_conv3x3_tune(310 + (bn ? 1 : 0), 0); return BH_OK; }   // (-44, 0|1): fp16-piece 3x3 launches on the persistent producer / consumer kernel off / on
    if (bm == -41) { bh_conv3x3_tune(200 + bn, 0); return BH_OK; }      // (-41, n): 3x3 kernel - second-round workgroups sleep n x 8128 cycles first
    if (bm == -18) { bh_conv3x3_tune(400 + bn, 0); return BH_OK; }       // (-18, bits): 3x3 kernel ablation - 1 no weight DMA, 2 no halo DMA in the loop
    if (bm == -30) { bh_wgrad_x3_tune(0, bn); return BH_OK; }            // (-30, n): workgroups per launch of the f32x3 wgrad kernel
    if (bm == -31) { bh_wgrad_x3_tune(1, bn); return BH_OK; }            // (-31, 1): ablation - that kernel without its atomic flush
    if (bm == -33) { bh_pointwise_tune(0, bn); return BH_OK; }           // (-33, 0 / 1): the pointwise streaming kernel off / on
    if (bm == -34) { bh_pointwise_tune(1, bn); return BH_OK; }           // (-34, n): its workgroups per launch
    if (bm == -36) { bh_wgrad_x3_tune(3, bn); return BH_OK; }            // (-36, 0 / 1): the 4 x 4-map form of the fp16-piece weight gradient off / on
    if (bm == -32) { bh_wgrad_x3_tune(2, bn); return BH_OK; }            // (-32, 0 / 1): the fp16-piece kernel's four-wave / eight-wave (producer + consumer) form
    if (bm == -20) { bh_bn_tune(bn); return BH_OK; }                     // (-20, n): workgroups per BatchNorm apply launch
    if (bm == -49) { g_s2_merge = bn; return BH_OK; }                    // (-49, 0 / 1): stride-2 dgrad - one launch per parity class / one launch
    if (bm == -48) { g_gemm_x3 = bn; return BH_OK; }                     // (-48, 0 / 1): the generic kernel's three-bf16-piece form in the fp32-accurate modes off / on
    if (bm == -47) { bh_stem7_tune(bn); return BH_OK; }                  // (-47, 0 / 1): the stems' fp16-piece forms (forward, one-channel dgrad) off / on
    if (bm == -14) { bh_warp_tune(0, bn); return BH_OK; }               // (-14, 1|2): warp forward rows per thread
    if (bm == -15) { bh_warp_tune(1, bn); return BH_OK; }               // (-15, 1|2|4): warp adjoint rows per thread
    if (bm == -12) { bh_conv3x3_tune(20 + bn, 0); return BH_OK; }      // (-12, 1|2): 3x3 kernel tile positions per workgroup on two-round launches
    if (bm == -9) { bh_conv3x3_tune(10 + bn, 0); return BH_OK; }     // (-9, 1|2): 3x3 kernel sub-tiles per workgroup (64-channel tile)
    if (bm == -8) { bh_conv3x3_tune(-101 - bn, 0); return BH_OK; }  // (-8, n): ablation - 3x3 kernel runs n channel chunks only (-1: all)   // (-7, 1): ablation - wgrad without its atomic flush      // (-6, 1): disable the dedicated 7x7 stem forward kernel
    g_force_bm = bm; g_force_bn = bn; return BH_OK;
}
#endif

static int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* res, float* y, const bh_conv_desc* d,
                         int relu, void* stream, double* bn_sums = nullptr, int groups = 1, float* amax_y = nullptr,
                         const bh_bn_in* bni = nullptr) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return BH_E_BADARG;
    if (res && d->out_nchw) return BH_E_UNSUPPORTED;
    if (bni) {
        // 1x1 / stride 1 / pad 0 NHWC conv through the buffer-loader fp32 kernel only; the table's groups are equal stacks of images
        if (d->transposed || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->in_nchw || d->out_nchw || d->precision == 1 ||
            !bni->table || bni->groups < 1 || d->N % bni->groups || d->Ci % 32 || d->Co % 4 || res || relu)
            return BH_E_UNSUPPORTED;
        const long long rpg_ = (long long)(d->N / bni->groups) * d->Ho * d->Wo;
        if (rpg_ % 128 || (long long)d->N * d->Hi * d->Wi * d->Ci >= (1ll << 29)) return BH_E_UNSUPPORTED;
    }
    if (!res && !bn_sums && !amax_y && !bni) {
        int taken = 0;
        rc = bh_stem7_try(x, w, bias, y, d, relu, bh_stream(stream), &taken);
        if (rc || taken) return rc;
    }
    if (!bn_sums && !amax_y && !bni) {
        int taken = 0;
        rc = bh_conv3x3_try(x, w, bias, y, d, 0, 0, bh_stream(stream), &taken, nullptr, 1, res, relu);
        if (rc || taken) return rc;
    }
    if (!res && !relu && (!bn_sums || (groups >= 1 && d->N % groups == 0))) {
        // the decoder's pointwise layers on the large maps: the persistent streaming kernel (csrc/pointwise.hip)
        int taken = 0;
        rc = bh_pointwise_try(x, w, bias, y, d, bh_stream(stream), &taken, bn_sums, groups, amax_y, bni);
        if (rc || taken) return rc;
    }
    GemmArgs a = {};
    a.amax_out = reinterpret_cast<unsigned*>(amax_y);
    if (bni) { a.bni = bni->table; a.bni_relu = bni->relu; a.bni_rpg = (int)((long long)(d->N / bni->groups) * d->Ho * d->Wo); }
    a.res = res; a.relu = relu;
    if (bn_sums) {
        // rows of the GEMM per statistics group (input pixels for the transposed conv, whose taps scatter inside the image)
        const long long rpg = (long long)(d->N / groups) * (d->transposed ? d->Hi * d->Wi : d->Ho * d->Wo);
        // every workgroup ends with one atomic per (channel, moment): beyond ~2k workgroups per address the atomic unit
        // (one same-address f64 atomic per ~30 ns) is slower than a separate statistics pass
        if (d->out_nchw || rpg % 128 || rpg * groups > 2048ll * 64) return BH_E_UNSUPPORTED;
        a.bn_sums = bn_sums; a.bn_rpg = (int)rpg; a.bn_groups = groups; a.bn_C = d->Co; a.bn_det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0;
    }
    a.Src = x; a.Bw = w; a.bias = bias; a.Out = y; a.bf16 = d->precision == 1; a.x3 = (d->route & BH_ROUTE_GEMM_X3) && (d->precision == 2 || d->precision == 4);
    a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Ci; a.Kc = d->Ci;
    a.src_elems = (long long)d->N * d->Hi * d->Wi * d->Ci;
    a.bw_elems = (long long)d->Co * d->kh * d->kw * d->Ci;
    if (!d->transposed) {
        a.M = d->N * d->Ho * d->Wo; a.Nn = d->Co; a.T = d->kh * d->kw;
        a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
        a.src_nchw = d->in_nchw;
        a.sBt = d->Ci; a.sBc = 1; a.sBn = (long long)a.T * d->Ci; a.b_kcontig = 1;   // W[n][t][c]
        a.epi = d->out_nchw ? 2 : 0;
    } else {
        // M = input pixels, N = taps*Co, scatter epilogue. Wt[ci][tap][co]
        const int taps = d->kh * d->kw;
        a.M = d->N * d->Hi * d->Wi; a.Nn = taps * d->Co; a.T = 1;
        a.Ho = d->Hi; a.Wo = d->Wi; a.kw = 1; a.stride = 1; a.pad = 0;
        a.sBt = 0; a.sBc = (long long)taps * d->Co; a.sBn = 1; a.b_kcontig = 0;
        a.epi = 1; a.ek = d->stride; a.eC = d->Co;
    }
    return dispatch(a, bh_stream(stream));
}

int bh_conv_fwd(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, void* stream) {
    return conv_fwd_impl(x, w, bias, nullptr, y, d, 0, stream);
}

int bh_conv_fwd_amax(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, float* amax_y, void* stream) {
    if (!amax_y) return BH_E_BADARG;
    return conv_fwd_impl(x, w, bias, nullptr, y, d, 0, stream, nullptr, 1, amax_y);
}

int bh_conv_fwd_act(const float* x, const float* w, const float* bias, const float* res, float* y, const bh_conv_desc* d,
                    int relu, void* stream) {
    return conv_fwd_impl(x, w, bias, res, y, d, relu, stream);
}

int bh_conv_fwd_bnin(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, double* sums, int groups,
                     const bh_bn_in* bni, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !bni || d->out_nchw || (sums && (groups < 1 || d->N % groups))) return BH_E_BADARG;
    if (d->kh == 1 && d->kw == 1) {
        // round 4: the 1x1 conv behind BatchNorm + ReLU (decoder units): the generic kernel transforms its A operand while staging;
        // statistics of the output in its epilogue where the grid allows, else one statistics pass over y
        rc = conv_fwd_impl(x, w, bias, nullptr, y, d, 0, stream, sums, groups, nullptr, bni);
        if (rc == BH_E_UNSUPPORTED && sums) {
            rc = conv_fwd_impl(x, w, bias, nullptr, y, d, 0, stream, nullptr, 1, nullptr, bni);
            if (rc) return rc;
            if (bh_query("bn_stats_kernel")) return BH_OK;
            return bn_launch_stats(y, groups, (d->N / groups) * d->Ho * d->Wo, d->Co, sums, bh_stream(stream), (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0);
        }
        return rc;
    }
    int taken = 0;
    rc = bh_conv3x3_try(x, w, bias, y, d, 0, 0, bh_stream(stream), &taken, sums, sums ? groups : 1, nullptr, 0, nullptr, bni);
    if (rc) return rc;
    return taken ? BH_OK : BH_E_UNSUPPORTED;
}

int bh_conv_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, double* sums,
                        int groups, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !sums || groups < 1 || d->N % groups || d->out_nchw) return BH_E_BADARG;
    int taken = 0;
    rc = bh_conv3x3_try(x, w, bias, y, d, 0, 0, bh_stream(stream), &taken, sums, groups);
    if (rc || taken) return rc;
    rc = bh_stem7_try(x, w, bias, y, d, 0, bh_stream(stream), &taken, sums, groups);      // (statistics in its epilogue)
    if (rc || taken) return rc;
    if (!taken) {
        rc = conv_fwd_impl(x, w, bias, nullptr, y, d, 0, stream, sums, groups);   // generic kernel, statistics in its epilogue
        if (rc == BH_OK) return rc;
        if (rc != BH_E_UNSUPPORTED) return rc;
        rc = bh_conv_fwd(x, w, bias, y, d, stream);
        if (rc) return rc;
    }
    if (bh_query("bn_stats_kernel")) return BH_OK;
    return bn_launch_stats(y, groups, (d->N / groups) * d->Ho * d->Wo, d->Co, sums, bh_stream(stream), (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// dgrad of a stride-2 conv (3x3 pad 1, or 1x1 pad 0) by output parity class.  The adjoint gather of the generic kernel
// visits all kh*kw taps for every input pixel although only those with (iy + pad - ky) even exist: 3/4 of the MFMA work
// is on zeros.  Per class (iy%2, ix%2) the valid taps form a dense 1x1 / 1x2 / 2x1 / 2x2 stride-1 convolution over gy
// (pad 0), so each class is one forward-gather launch of the same kernel on class-packed, transposed weights
// Wc[ci][ty][tx][co] whose scatter epilogue writes pixel (2a + py, 2b + px).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_s2_dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co,
                                                                    int Ci, int k, int pad) {
    // wp = four class blocks; class (py, px): taps ky = ky0 + 2*ty (ty < nty), source row a + cy - ty; packed tap index
    // ky' = nty - 1 - ty so that source row = a + ky' (forward rule, pad 0).  Layout per class [Ci][nty][ntx][Co].
    const int total = Co * k * k * Ci;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % Ci, t = (i / Ci) % (k * k), co = i / (Ci * k * k);
        const int ky = t / k, kx = t - ky * k;
        // class of this tap: (iy + pad - ky) even  <=>  iy = (ky - pad) mod 2
        const int py = ((ky - pad) % 2 + 2) % 2, px = ((kx - pad) % 2 + 2) % 2;
        const int ky0 = (py + pad) % 2, kx0 = (px + pad) % 2;
        const int nty = (k - ky0 + 1) / 2, ntx = (k - kx0 + 1) / 2;
        const int ty = (ky - ky0) / 2, tx = (kx - kx0) / 2;
        // class block offsets in order (0,0), (0,1), (1,0), (1,1)
        int base = 0;
        for (int c = 0; c < py * 2 + px; ++c) {
            const int qy = c >> 1, qx = c & 1;
            const int a0 = (qy + pad) % 2, b0 = (qx + pad) % 2;
            base += Ci * ((k - a0 + 1) / 2) * ((k - b0 + 1) / 2) * Co;
        }
        wp[base + ((ci * nty + (nty - 1 - ty)) * ntx + (ntx - 1 - tx)) * Co + co] = w[i];
    }
}

int bh_conv_dgrad_s2(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate, float* wpack,
                     void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!gy || !w || !gx || !wpack) return BH_E_BADARG;
    const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1, k1 = d->kh == 1 && d->kw == 1 && d->pad == 0;
    if (d->transposed || d->stride != 2 || !(k3 || k1) || d->in_nchw || d->out_nchw || (d->Hi & 1) || (d->Wi & 1) ||
        d->Ho * 2 != d->Hi || d->Wo * 2 != d->Wi || (d->Co % 4) || (d->Ci % 4))
        return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    const int k = d->kh, pad = d->pad;
    const int total = d->Co * k * k * d->Ci;
    if (!bh_query("pack_s2_dgrad_weights_kernel")) {
        hipLaunchKernelGGL(pack_s2_dgrad_weights_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0, s,
                           w, wpack, d->Co, d->Ci, k, pad);
        BH_LAUNCH_CHECK();
    }
    if (k1 && !accumulate && !bh_query_ctx) {                       // three of the four classes have no tap: their gradient is zero
        hipError_t e = hipMemsetAsync(gx, 0, sizeof(float) * (size_t)d->N * d->Hi * d->Wi * d->Ci, s);
        if (e != hipSuccess) return (int)e;
    }
    // the classes are ONE launch (round 6; they were four - each a few k-tiles on 512 workgroups, latency-bound): gridDim.z = classes with
    // at least one tap, the workgroup takes its class's packed weights and tap set from the argument tables
    GemmArgs a = {};
    a.Src = gy; a.bias = nullptr; a.Out = gx; a.accumulate = accumulate; a.bf16 = d->precision == 1; a.x3 = (d->route & BH_ROUTE_GEMM_X3) && (d->precision == 2 || d->precision == 4);
    a.M = d->N * d->Ho * d->Wo; a.Nn = d->Ci; a.Kc = d->Co;
    a.Ho = d->Ho; a.Wo = d->Wo;                 // class grid (Hi/2 x Wi/2) == gy grid
    a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Co;
    a.src_elems = (long long)d->N * d->Ho * d->Wo * d->Co;
    a.adjoint = 0; a.stride = 1; a.pad = 0;
    a.epi = 1; a.ek = 2; a.eC = d->Ci;
    long long base = 0;
    int tmax = 0;
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int ky0 = (py + pad) % 2, kx0 = (px + pad) % 2;
        const int nty = (k - ky0 + 1) / 2, ntx = (k - kx0 + 1) / 2;
        if (nty > 0 && ntx > 0) {
            // source rows: a + cy - ty with cy = (py + pad - ky0) / 2; with the flipped packed order the first packed tap
            // reads row a + cy - (nty - 1): the forward rule iy = a - padv + ky' needs padv = (nty - 1) - cy
            const int cy = (py + pad - ky0) / 2, cx = (px + pad - kx0) / 2;
            const int padv_y = (nty - 1) - cy, padv_x = (ntx - 1) - cx;
            if (padv_y != 0 || padv_x != 0) return BH_E_UNSUPPORTED;      // (0 for the two supported geometries)
            const int z = a.nz++;
            a.Bw_z[z] = wpack + base; a.T_z[z] = nty * ntx; a.kw_z[z] = ntx; a.etap0_z[z] = py * 2 + px;
            a.sBn_z[z] = (long long)nty * ntx * d->Co;      // B[t][k = co][n = ci] = Wc[ci][t][co]
            if (nty * ntx > tmax) tmax = nty * ntx;
        }
        base += (long long)d->Ci * nty * ntx * d->Co;
    }
    if (!a.nz) return BH_OK;
    // (the fields the dispatcher looks at: the largest class; the kernel replaces them per workgroup)
    a.Bw = a.Bw_z[0]; a.T = tmax; a.kw = a.kw_z[0]; a.etap0 = a.etap0_z[0];
    a.bw_elems = (long long)d->Ci * tmax * d->Co;
    a.sBt = d->Co; a.sBc = 1; a.sBn = a.sBn_z[0]; a.b_kcontig = 1;
    if (!g_s2_merge) {                           // (tuning build: one launch per class, as before)
        const int nz = a.nz;
        GemmArgs b = a;
        b.nz = 0;
        for (int z = 0; z < nz; ++z) {
            b.Bw = a.Bw_z[z]; b.T = a.T_z[z]; b.kw = a.kw_z[z]; b.etap0 = a.etap0_z[z]; b.sBn = a.sBn_z[z];
            b.bw_elems = (long long)d->Ci * b.T * d->Co;
            rc = dispatch(b, s);
            if (rc) return rc;
        }
        return BH_OK;
    }
    return dispatch(a, s);
}

int bh_conv_dgrad_bnreduce(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate,
                           const bh_bn_reduce* bnr, double* sums, int groups, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!gy || !w || !gx || !bnr || !sums) return BH_E_BADARG;
    int taken = 0;
    rc = bh_conv3x3_try(gy, w, nullptr, gx, d, 1, accumulate, bh_stream(stream), &taken, sums, groups, nullptr, 0, bnr);
    if (rc) return rc;
    return taken ? BH_OK : BH_E_UNSUPPORTED;       // only where the halo-tiled 3x3 kernel applies: the caller checks
}

int bh_conv_dgrad_colsum(const float* gy, const float* w, float* gx, const bh_conv_desc* d, double* sums, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!gy || !w || !gx || !sums) return BH_E_BADARG;
    int taken = 0;
    rc = bh_conv3x3_try(gy, w, nullptr, gx, d, 1, 0, bh_stream(stream), &taken, sums, 1, nullptr, 0, nullptr);
    if (rc) return rc;
    return taken ? BH_OK : BH_E_UNSUPPORTED;
}

__global__ void bias_grad_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ gbias, int groups, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t = 0;
    for (int g = 0; g < groups; ++g) t += bn_sum_total(sums, groups, g, C, c, 0);
    gbias[c] += (float)t;
}

int bh_bias_grad_from_sums(const double* sums, float* gbias, int groups, int C, void* stream) {
    if (!sums || !gbias || groups < 1 || C < 1) return BH_E_BADARG;
    hipLaunchKernelGGL(bias_grad_from_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, bh_stream(stream), sums, gbias, groups, C);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_conv_dgrad(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!gy || !w || !gx) return BH_E_BADARG;
    if (d->in_nchw && (d->transposed || accumulate)) return BH_E_UNSUPPORTED;
    {
        int taken = 0;
        rc = bh_conv3x3_try(gy, w, nullptr, gx, d, 1, accumulate, bh_stream(stream), &taken, nullptr, 1);
        if (rc || taken) return rc;
    }
    if (!d->transposed && d->Ci == 1 && !d->out_nchw && !accumulate && d->Co % 4 == 0 && d->Co <= 256 &&
        (64 % (d->Co / 4)) == 0 && d->kh == 7 && d->kw == 7 && d->stride == 2) {
        const size_t lds = sizeof(float) * d->kh * d->kw * d->Co;
        dim3 grid(1024, d->stride * d->stride);
        if (bh_query("stem_dgrad_c1_kernel<7,2>")) return BH_OK;
        hipLaunchKernelGGL((stem_dgrad_c1_kernel<7, 2>), grid, dim3(256), lds, bh_stream(stream), gy, w, gx, d->N, d->Hi,
                           d->Wi, d->Ho, d->Wo, d->Co, d->pad);
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    GemmArgs a = {};
    a.src_nchw = d->out_nchw;                         // gradient of the NCHW network output
    a.Src = gy; a.Bw = w; a.bias = nullptr; a.Out = gx; a.accumulate = accumulate; a.bf16 = d->precision == 1; a.x3 = (d->route & BH_ROUTE_GEMM_X3) && (d->precision == 2 || d->precision == 4);
    a.M = d->N * d->Hi * d->Wi; a.Nn = d->Ci; a.Kc = d->Co; a.T = d->kh * d->kw;
    a.Ho = d->Hi; a.Wo = d->Wi;               // output-side grid of this GEMM = conv input grid
    a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Co;  // gathered source = gy grid
    a.src_elems = (long long)d->N * d->Ho * d->Wo * d->Co;
    a.bw_elems = (long long)d->Co * d->kh * d->kw * d->Ci;
    a.kw = d->kw;
    if (!d->transposed) {
        a.adjoint = 1; a.stride = d->stride; a.pad = d->pad;
        // B[t][k = co][j = ci] = W[co][t][ci]
        a.sBt = d->Ci; a.sBc = (long long)a.T * d->Ci; a.sBn = 1; a.b_kcontig = 0;
    } else {
        // gx[iy][ix][ci] = sum_{a,b,co} gy[2iy+a][2ix+b][co] * Wt[ci][tap][co]: forward gather, stride k, pad 0
        a.adjoint = 0; a.stride = d->stride; a.pad = 0;
        a.sBt = d->Co; a.sBc = 1; a.sBn = (long long)a.T * d->Co; a.b_kcontig = 1;
    }
    a.epi = d->in_nchw ? 2 : 0;                       // gradient w.r.t. an NCHW network input (RGB patch into the extractor)
    return dispatch(a, bh_stream(stream));
}

}  // extern "C"
