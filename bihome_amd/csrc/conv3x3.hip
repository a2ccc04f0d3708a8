// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) as a halo-tiled implicit GEMM on the gfx950 f32-input MFMA.
//
// The generic implicit-GEMM kernel (conv_gemm.hip) re-gathers the A operand from L2 once per tap: for a 3x3 conv
// every input pixel crosses the L2 -> LDS path nine times and a 64x64 tile moves 16 KB per 64x64x32 MFMA step.
// Here a workgroup owns two 8x8-pixel sub-tiles (128 GEMM rows) x 64 output channels.  For each 32-channel chunk
// the two 10x10 input halos are brought into LDS ONCE and all nine taps read them with a shifted address; only
// the 32x64 weight slab of a tap is streamed per step.  Per MFMA the L2 -> LDS traffic is ~3x lower, the LDS read
// traffic ~25% lower (each wave owns 64x32 of the tile: two A fragments share one B fragment).
//
// All staging is direct-to-LDS (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass; out-of-image halo
// pixels and out-of-range channels use an out-of-range buffer offset, which the hardware turns into zeros in LDS.
// LDS image (lane-linear per wave instruction, 1 KB = 64 x float4):
//   halo stage : [8 k-planes][2 sub-tiles][100 halo pixels] x float4     (a k-plane = 4 consecutive channels)
//   B, forward : [8 k-planes][64 n] x float4          (weights [Co][tap][Ci] are k-contiguous: same C4 scheme)
//   B, dgrad   : [32 k][64 n] floats                  (weights are n-contiguous: fragments by 4 x ds_read_b32)
// One ds_read_b128 feeds four v_mfma_f32_32x32x2_f32 (half-wave 0 takes plane 2q, half-wave 1 plane 2q+1; MFMA
// step j contracts channel 8q+j with 8q+4+j - any pairing is valid inside a sum over k).
// Double-buffered: the weight slab of step t+1 and (one wave instruction per tap step) the halo of the next
// chunk are in flight while the 32 MFMAs per wave of step t run; one barrier per step.
//
// PACKED variant (training path, round 2): the weights arrive pre-packed in MFMA-fragment order (bh_conv3x3_pack, one
// batched launch per optimizer step for all layers: [chunk][tap][32-wide n tile][q][lane] x float4), so a wave fetches the
// B fragments of a tap with four fully coalesced 1 KB buffer_load_dwordx4 straight into registers, one tap ahead.  No
// weight slab in LDS, no slab DMA, no B fragment ds_reads, and - because nothing in LDS changes inside a chunk - ONE
// workgroup barrier per 32-channel chunk instead of nine: the four waves of a workgroup (and the two workgroups of a
// CU) drift freely through the taps.  Forward and dgrad are the same loop (the flip / transpose lives in the pack).
#include "common.h"
#include "conv3x3_args.h"
#include <type_traits>

#ifndef C3_X3_PIPE
// 1: the software-pipelined tap loop of round 3 (fragments half a tap, weights a full tap ahead; A/B builds: make ab
// ABFLAGS=-DC3_X3_PIPE=1).  Measured: 70.5 vs 70.7 us (dgrad), 71.5 vs 72.5 us (forward) per launch, 18.07 vs 18.19 ms per step,
// and the BatchNorm-on-load variant spills (58.8 vs 55.7 us) - the tap loop is NOT latency-bound: tools/x3_timeline.py shows the
// CU's matrix pipe busy from the first stage-in to the last tile at the rate ONE wave per SIMD reaches alone, and
// tools/mfma_clock_probe.hip shows why that rate is what it is (the chip sustains 1.6-1.9 PFLOP/s of bf16 MFMA on random-bit
// operands whatever the pipe occupancy: a power limit, not an issue limit).  Default: the plain loop.
#define C3_X3_PIPE 0
#endif

#ifndef C3_BN32_WAVES
// waves per SIMD the two-piece 32-channel-tile kernels (the single-chunk full-resolution decoder layers) are compiled for
#define C3_BN32_WAVES 2
#endif

#ifdef BH_TUNING
// phase time stamps (100 MHz wall clock) of the halo kernels: [workgroup tile][8] = start, first barrier, loop end, epilogue end, XCC / CU id
__device__ unsigned long long g_c3_ts[8 * 16384];
#define C3_STAMP(tile, k) do { if (a.dbg_ts && threadIdx.x == 0 && (tile) < 16384) { g_c3_ts[(tile) * 8 + (k)] = wall_clock64();          \
        if ((k) == 1 || (k) == 2) g_c3_ts[(tile) * 8 + 4 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define C3_STAMP(tile, k) do { } while (0)
#endif

constexpr int C3_HALO_BYTES = 8 * 200 * 16;          // 25600
constexpr int C3_B_BYTES = 8192;
constexpr int C3_LDS_BYTES = 2 * C3_HALO_BYTES + 2 * C3_B_BYTES;      // 67584

// ---- epilogue of one wave's accumulators: C/D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) ----
// (shared by the halo kernels: the wave that holds - or was handed - the accumulators of tile position bx calls it with that
//  wave's (wm, wn, wh) roles.  s1 / s2 return the tile's column sums for c3_stats_merge.)
template <int BN, int SUBT, int TM, bool MAP4 = false>
__device__ __forceinline__ void c3_epilogue(const C3Args& a, f32x16 (&acc)[TM], int bx, int n0, int wm, int wn, int wh, int l31, int kh2,
                                            double& s1, double& s2, int& img, int& g, float& amx) {
    // ---- epilogue: C/D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) ----
    // Element (i, r) of a lane is pixel (c3_strip_row(2*(r>>2) + kh2), 4*(i + wh) + (r&3)) of the sub-tile: the row part of
    // its address is one of four per-lane VGPRs, the column part (4*(i + wh) + (r&3)) * Nn is workgroup-uniform and rides in
    // the scalar offset of the buffer instruction - no per-element address arithmetic.
    g = bx * SUBT + wm;
    const int n = n0 + wn * 32 + l31;
    const bool valid = g < a.subtiles && n < a.Nn;
    const int gg = g < a.subtiles ? g : 0;
    int ty, tx;
    if constexpr (MAP4) { img = gg * 4; ty = 0; tx = 0; }       // a sub-tile = four 4x4 images (first image: the statistics group)
    else if (a.tpi_shift >= 0) { img = gg >> a.tpi_shift; const int t = gg & (a.tiles_per_img - 1); ty = t >> a.tx_shift; tx = t & (a.tiles_x - 1); }
    else { img = gg / a.tiles_per_img; const int t = gg - img * a.tiles_per_img; ty = t / a.tiles_x; tx = t - ty * a.tiles_x; }
    const float bv = (a.bias && n < a.Nn) ? a.bias[n] : 0.0f;
    s1 = 0.0; s2 = 0.0;                            // BatchNorm statistics of the tile (a.bn_sums): sum y, sum y^2
    float r_mean = 0.f, r_invstd = 0.f, r_sc = 0.f, r_sh = 0.f;
    if (a.bnr_z && valid) {                        // coefficients of the BatchNorm whose output gradient this tile is
        const int grp = img / a.imgs_per_group;
        const double rows = (double)a.bnr_rows;
        const double mu = bn_sum_total(a.bnr_stats, a.groups, grp, a.Nn, n, 0, a.det) / rows;
        double var = bn_sum_total(a.bnr_stats, a.groups, grp, a.Nn, n, 1, a.det) / rows - mu * mu;
        if (var < 0) var = 0;
        r_mean = (float)mu;
        r_invstd = 1.0f / sqrtf((float)var + a.bnr_eps);
        r_sc = (a.bnr_gamma ? a.bnr_gamma[n] : 1.f) * r_invstd;
        r_sh = (a.bnr_beta ? a.bnr_beta[n] : 0.f) - r_mean * r_sc;
    }
    if (valid) {
        unsigned rowoff[4];                          // byte offsets (< 2^31: the host checks the tensor sizes)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int sr = c3_strip_row(2 * rq + kh2);
            if constexpr (MAP4)      // strip row sr of the wave's 8x4 strip = row sr & 3 of image 2 wh + (sr >> 2) of the sub-tile
                rowoff[rq] = ((unsigned)((img + 2 * wh + (sr >> 2)) * 16 + (sr & 3) * 4) * (unsigned)a.Nn + (unsigned)n) * 4u;
            else
                rowoff[rq] = ((unsigned)((img * a.H + ty * 8 + sr) * a.W + tx * 8) * (unsigned)a.Nn + (unsigned)n) * 4u;
        }
        const unsigned colstep = (unsigned)a.Nn * 4u;                                   // one pixel to the right
        const unsigned col0 = MAP4 ? 0u : (unsigned)(wh * 4) * colstep;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, a.out_bytes, 0x00020000);
#define C3_SOFF(i, r) (col0 + (unsigned)((i) * 4 + ((r) & 3)) * colstep)
        if (!a.accumulate && !a.res && !a.bnr_z) {
            // plain store (+ bias, ReLU) and the forward statistics: partial sums of a fragment quad in float, totals in double
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float q1 = 0.f, q2 = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = acc[i][rq * 4 + c] + bv;
                        if (a.relu) v = fmaxf(v, 0.0f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, rowoff[rq], C3_SOFF(i, c), 0);
                        q1 += v; q2 = __builtin_fmaf(v, v, q2);
                    }
                    s1 += (double)q1; s2 += (double)q2;
                }
        } else {
            // Generic epilogue, instantiated per combination of (old gradient / residual present, BatchNorm reduce with or
            // without a saved output) so that no variant carries the loads, zero fills and selects of the others.
            // Phase 1 issues every global read (old gradient, residual, BatchNorm input / output) before the first store -
            // a load behind a store would otherwise wait for that store (one vmcnt queue, in order).
            const bool rd_old = a.accumulate, rd_res = a.res != nullptr, rd_z = a.bnr_z != nullptr;
            const bool rd_y = rd_z && a.bnr_relu && a.bnr_y != nullptr;
            const float c_x0 = -r_mean * r_invstd;                     // xhat = z * invstd + c_x0
            auto body = [&](auto EXTRA, auto ZMODE) {                  // ZMODE 0: forward statistics, 1: reduce (mask from z), 2: reduce (mask from y)
                constexpr bool HAS_EXTRA = decltype(EXTRA)::value;
                constexpr int ZM = decltype(ZMODE)::value;
                float ext[HAS_EXTRA ? TM : 1][16], zin[ZM > 0 ? TM : 1][16], yin[ZM == 2 ? TM : 1][16];
                if constexpr (HAS_EXTRA) {
                    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rd_res ? a.res : a.Out), 0, a.out_bytes, 0x00020000);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float o = 0.f;
                            if (rd_old) o = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, rowoff[r >> 2], C3_SOFF(i, r), 0));
                            if (rd_res) o += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, rowoff[r >> 2], C3_SOFF(i, r), 0));
                            ext[i][r] = o;
                        }
                }
                if constexpr (ZM > 0) {
                    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bnr_z), 0, a.out_bytes, 0x00020000);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            zin[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsZ, rowoff[r >> 2], C3_SOFF(i, r), 0));
                }
                if constexpr (ZM == 2) {
                    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bnr_y), 0, a.out_bytes, 0x00020000);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            yin[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, rowoff[r >> 2], C3_SOFF(i, r), 0));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        float q1 = 0.f, q2 = 0.f;                          // partial sums of a fragment quad in float, totals in double
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int r = rq * 4 + c;
                            float v = acc[i][r] + bv;
                            if constexpr (HAS_EXTRA) v += ext[i][r];
                            if (a.relu) v = fmaxf(v, 0.0f);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, rowoff[rq], C3_SOFF(i, r), 0);
                            if constexpr (ZM > 0) {
                                const float zv = zin[i][r];
                                if (a.bnr_relu) {
                                    const float yv = ZM == 2 ? yin[i][r] : __builtin_fmaf(zv, r_sc, r_sh);
                                    if (!(yv > 0.f)) v = 0.f;
                                }
                                amx = fmaxf(amx, fabsf(v));            // (bh_bn_reduce.amax_d: max |mask(d)|)
                                q1 += v; q2 = __builtin_fmaf(v, __builtin_fmaf(zv, r_invstd, c_x0), q2);
                            } else {
                                q1 += v; q2 = __builtin_fmaf(v, v, q2);
                            }
                        }
                        s1 += (double)q1; s2 += (double)q2;
                    }
            };
            using T_ = std::true_type; using F_ = std::false_type;
            using Z0 = std::integral_constant<int, 0>; using Z1 = std::integral_constant<int, 1>; using Z2 = std::integral_constant<int, 2>;
            if (rd_old || rd_res) {
                if (!rd_z) body(T_{}, Z0{}); else if (!rd_y) body(T_{}, Z1{}); else body(T_{}, Z2{});
            } else {
                if (!rd_z) body(F_{}, Z0{}); else if (!rd_y) body(F_{}, Z1{}); else body(F_{}, Z2{});
            }
        }
#undef C3_SOFF
    }
}

// column sums of a tile -> the [groups][C][2] totals.  PHASE 0: both halves with a workgroup barrier in between (the halo
// kernels); 1: the LDS writes only; 2: the merge + atomics only (the caller puts a barrier between the two).
template <int BN, int PHASE>
__device__ __forceinline__ void c3_stats_merge(const C3Args& a, char* redb, double s1, double s2, int img, int g, int bx, int n0, int wave,
                                               int lane, int tid) {
    const int l31 = lane & 31, kh2 = lane >> 5;
    double* red = reinterpret_cast<double*>(redb);          // [4 waves][32 columns][2]
    int* grp_of = reinterpret_cast<int*>(redb + 4 * 32 * 2 * 8);
    if constexpr (PHASE != 2) {
        // the two half-waves hold the two row halves of a column; the waves that share the channel range are merged through
        // LDS when they belong to the same statistics group, then one f64 atomic per (channel, moment) goes to the totals
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const int grp = img / a.imgs_per_group;
        if (kh2 == 0) { red[(wave * 32 + l31) * 2] = s1; red[(wave * 32 + l31) * 2 + 1] = s2; }
        if (lane == 0) grp_of[wave] = g < a.subtiles ? grp : -1;
    }
    if constexpr (PHASE == 0) __syncthreads();
    if constexpr (PHASE != 1) {
        constexpr int SHARE = BN == 64 ? 2 : 4;                  // waves per channel range
        // thread -> (channel range cr, column, moment); waves of range cr: BN=64: wave = wm + 2*cr; BN=32: all four
        if (tid < (4 / SHARE) * 64) {
            const int cr = tid >> 6, col = (tid >> 1) & 31, mom = tid & 1;
            const int nn = n0 + cr * 32 + col;
            if (nn < a.Nn) {
                int done = 0;                                    // bit w: wave already merged
#pragma unroll
                for (int w0 = 0; w0 < SHARE; ++w0) {
                    const int wv0 = BN == 64 ? (w0 + 2 * cr) : w0;
                    const int g0 = grp_of[wv0];
                    if (g0 < 0 || (done >> w0) & 1) continue;
                    double tot = red[(wv0 * 32 + col) * 2 + mom];
#pragma unroll
                    for (int w1 = w0 + 1; w1 < SHARE; ++w1) {
                        const int wv1 = BN == 64 ? (w1 + 2 * cr) : w1;
                        if (grp_of[wv1] == g0) { tot += red[(wv1 * 32 + col) * 2 + mom]; done |= 1 << w1; }
                    }
                    bh_acc_add(&a.bn_sums[bn_sum_index(bx % BH_BN_SUM_SLOTS, a.groups, g0, a.Nn, nn, mom)], tot, a.det);
                }
            }
        }
    }
}

// BN = 64: wave (wm, wn) owns sub-tile wm x channels [32 wn, 32 wn + 32) (two A fragments per B fragment);
// BN = 32: wave w owns rows [32 w, 32 w + 32) of the 128-row tile x all 32 channels (the 32-channel decoder layers).
// BF16 (bh_conv_desc.precision = 1): same fp32 LDS image, the fragments are rounded to bf16 in registers and fed to
// v_mfma_f32_32x32x16_bf16 (16 channels per instruction: half-wave 0 holds planes 4s, 4s+2, half-wave 1 planes 4s+1,
// 4s+3 of the A and of the B fragment alike); fp32 accumulate.  The MFMA time drops 16x, the kernel becomes LDS-bound.
// SUBT = 8x8 sub-tiles per workgroup.  2 (default): 128 GEMM rows, 67.5 KB of LDS, two workgroups per CU.  1 (BN = 64
// only): 64 rows, a wave owns 32 rows x 32 channels, 41.6 KB of LDS, three workgroups per CU - more weight-slab traffic
// and LDS reads per MFMA, but three workgroups drift out of lockstep and cover each other's prologue / epilogue.
// X3 (PACKED only, bh_conv_desc.precision = 2): the LDS image of a halo stage is [3 pieces][4 k-planes of 8 bf16 channels]
// [SUBT][100 halo pixels] x 16 B - the same slot geometry with 12 instead of 8 plane images (38.4 KB per stage, two stages,
// two workgroups per CU).  The halo goes through registers (two dwordx4 per slot, cut into the three pieces, three
// ds_write_b128) two taps after it was requested: no LDS-DMA, one bare s_barrier per chunk.  Per tap and 32x32 accumulator:
// 6 ds_read_b128 and 12 MFMAs of 32 cycles (fp32 form: 4 reads, 16 MFMAs of 64 cycles).
// NP (X3 only): bf16 pieces per operand.  3: the exact cut, six products ("f32x3", precision 2).  2: both pieces rounded to nearest,
// three products ("f32x2", precision 3: common.h bh_split8_2) - 8 instead of 12 plane images, 4 instead of 6 weight loads per tap.
// MAP4 (X3, SUBT = 1, BN = 64 only; round 3): 4 x 4 feature maps (the 512-channel layer4 of the ResNet-34 regressor) - a "sub-tile" of 64
// GEMM rows is FOUR IMAGES, each with its own 6 x 6 zero-padded halo (144 halo slots instead of 100); a wave's 8 x 4 pixel strip is the
// two images 2 wh, 2 wh + 1, a tap is the shift dy * 6 + dx.  Same loop, same fragment reads.
// F16 (X3, NP = 2; round 4): the two pieces are FP16 numbers of the operand times a power-of-two scale per tensor ("f16x2", precision 4,
// w_layout 4; common.h F16X2) - same loop, same LDS image as NP = 2, ~2^-22 per product instead of 2^-18; the accumulators are
// rescaled by 2^-(k_src + k_w) in front of the epilogue.
template <bool FLIP, int BN, bool BF16 = false, int SUBT = 2, bool PACKED = false, bool X3 = false, bool BNI = false, int NP = 3, bool MAP4 = false, bool F16 = false>
__global__ void __launch_bounds__(256, X3 ? ((BN == 32 && NP == 2 && !BNI) ? C3_BN32_WAVES : 2) : 1) conv3x3_halo_kernel(C3Args a) {
    static_assert(NP == 3 || (NP == 2 && X3), "two pieces: split form only");
    static_assert(!F16 || (X3 && NP == 2), "fp16 pieces: the two-piece split form");
    static_assert(!MAP4 || (X3 && SUBT == 1 && BN == 64 && !BNI), "the 4 x 4 map form: split operands, one sub-tile, 64-channel tile");
    constexpr int SLOTS = MAP4 ? 144 : 100;                // halo slots of a sub-tile per k-plane
    constexpr int ROWP = MAP4 ? 6 : 10;                    // halo row pitch
    static_assert(SUBT == 2 || BN == 64, "one sub-tile per workgroup is built for the 64-channel tile only");
    static_assert(!X3 || (PACKED && !BF16), "the split form exists for packed weights only");
    static_assert(!BNI || (X3 && !FLIP), "the BatchNorm-on-load form is a forward f32x3 kernel");
    constexpr int TM = (BN == 64 && SUBT == 2) ? 2 : 1;
    constexpr int HPL = SLOTS * SUBT;                      // halo slots per k-plane
    constexpr int HALO_B = (X3 ? 4 * NP : 8) * HPL * 16;   // bytes of one halo stage
    constexpr int XJ = (4 * HPL + 255) / 256;              // X3: halo slots (of 8 channels) per thread and chunk (4 / 2)
    constexpr int HINS = (8 * HPL + 63) / 64;              // wave instructions per halo stage (25 / 13)
    constexpr int HJ = (HINS + 3) / 4;                     // ... per wave (7 / 4)
    constexpr int BINS = BN == 64 ? 2 : 1;                 // weight-slab wave instructions per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: LDS-DMA bases stay in SGPRs)
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int wm = SUBT == 1 ? 0 : (BN == 64 ? (wave & 1) : (wave >> 1));    // sub-tile of this wave
    const int wn = BN == 64 ? (wave >> 1) : 0;
    const int wh = (BN == 64 && SUBT == 2) ? 0 : (wave & 1);   // 32-row waves: left / right 8x4 strip of the sub-tile
    const int n0 = blockIdx.y * BN;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Src), 0, a.src_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wt), 0, a.w_bytes, 0x00020000);
    float f16_s = 1.0f;                                  // F16: 2^k_src, and the exponent that undoes both scales
    int f16_kout = 0;
    if constexpr (F16) {
        const unsigned* const wrec = reinterpret_cast<const unsigned*>(a.Wt) + (a.w_bytes >> 2);
        unsigned wv = lane < 16 ? wrec[lane] : 0u;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)wv, off, 64); wv = o > wv ? o : wv; }
        const int kw = bh_f16_scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)wv));
        const int ka = bh_f16_scale_exp(bh_amax_read(a.amax_src, lane));
        f16_s = __builtin_bit_cast(float, (unsigned)(127 + ka) << 23);
        f16_kout = -(ka + kw);
    }

    // ---- weight slab slots ----
    unsigned boff[2] = {0u, 0u};    // (fixed size: a template-dependent array type as a builtin operand silently drops
                                    //  the host-side kernel stub with this compiler)
#pragma unroll
    for (int i = 0; i < BINS; ++i) {
        const int idx = i * 4 + wave;                      // 1 KB wave instruction of the slab image
        if (!FLIP) {            // B[k][n] = W[n][tap][c0 + k]: image [8 planes][BN n] x float4
            const int plane = BN == 64 ? idx : idx * 2 + (lane >> 5), n = n0 + (BN == 64 ? lane : (lane & 31));
            boff[i] = n < a.Nn ? ((unsigned)(n * 9) * (unsigned)a.Cw + (unsigned)(plane * 4)) * 4u : OOB;
        } else {                // B[k][n] = W[c0 + k][8 - tap][n]: image [32 k][BN n] floats, BN/4 lanes x float4 per k row
            constexpr int LPR = BN / 4;
            const int k = idx * (64 / LPR) + lane / LPR, n = n0 + (lane % LPR) * 4;
            boff[i] = n < a.Nn ? ((unsigned)(k * 9) * (unsigned)a.Cw + (unsigned)n) * 4u : OOB;
        }
    }
#ifdef BH_TUNING
    const int nch = a.dbg_nch >= 0 ? a.dbg_nch : a.Kc / 32;     // (ablation hook: bh_debug_force_tile(-8, n) caps the chunk loop)
    const int dbg_noload = a.dbg_noload;
#else
    const int nch = a.Kc / 32;
    constexpr int dbg_noload = 0;
#endif
    // PACKED: B fragments straight from the fragment-ordered weight buffer; soffset = ((chunk*9 + tap)*NW + n tile) * 4 KB
    const int wnG = blockIdx.y * (BN / 32) + (BN == 64 ? (wave >> 1) : 0);
    const unsigned pb_voff = (unsigned)lane * 16u;
#define C3_LOAD_B(dst, c, tap)                                                                                          \
    do {                                                                                                                \
        const unsigned so_ = (unsigned)(((c) * 9 + (tap)) * a.NW + wnG) * 4096u;                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                              \
            const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(rsB, pb_voff + (unsigned)q_ * 1024u, so_, 0);         \
            dst[q_] = __builtin_bit_cast(float4, v_);                                                                   \
        }                                                                                                               \
    } while (0)
    // X3: [chunk][tap][n tile][piece][16-channel step][lane] x (8 bf16): six 1 KB loads per tap, dst index = piece * 2 + step
#define C3_LOAD_BX(dst, c, tap)                                                                                         \
    do {                                                                                                                \
        const unsigned so_ = (unsigned)(((c) * 9 + (tap)) * a.NW + wnG) * (unsigned)(NP * 2048);                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 2 * NP; ++q_) {                                                              \
            const auto v_ = __builtin_amdgcn_raw_buffer_load_b128(rsB, pb_voff + (unsigned)q_ * 1024u, so_, 0);         \
            dst[q_] = __builtin_bit_cast(uint4, v_);                                                                    \
        }                                                                                                               \
    } while (0)
    // a single chunk (32 source channels) never touches the second halo stage: the host then launches with one stage less
    // of LDS (41.6 instead of 67.2 KB: three workgroups per CU for the 32-channel layers) and the slabs move down
    const int slab0 = (a.Kc / 32 > 1 ? 2 : 1) * HALO_B;
#define issue_B(c, tap, bs)                                                                                             \
    do {                                                                                                                \
        const unsigned soff_ = FLIP ? (unsigned)(((c) * 32 * 9 + (8 - (tap))) * a.Cw) * 4u                             \
                                    : (unsigned)((tap) * a.Cw + (c) * 32) * 4u;                                        \
        char* base_ = smem + slab0 + (bs) * C3_B_BYTES + wave * 1024;                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(base_), 16, boff[0], soff_, 0, 0);                 \
        if (BINS == 2)                                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(base_ + 4096), 16, boff[1], soff_, 0, 0); \
    } while (0)
    // (the last wave instruction of a stage is partial - 64 resp. 32 of its lanes: the others are masked off, an
    //  out-of-range lane would still write zeros past the stage)
#define C3_ISSUE_HALO(j, c, hs)                                                                                         \
    do {                                                                                                                \
        if ((j) < HJ && (j) * 4 + wave < HINS && (((j) * 4 + wave) * 64 + lane) < 8 * HPL)                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem + (hs) * HALO_B + ((j) * 4 + wave) * 1024), \
                                                     16, hoff[(j) < HJ ? (j) : 0], (unsigned)((c) * 32) * 4u, 0, 0);    \
    } while (0)

    // A workgroup walks a.tpb consecutive tile positions (two on the launches that would otherwise be exactly two rounds of
    // resident workgroups: the second tile starts whenever the first is done instead of waiting for a dispatch slot, and
    // its prologue loads queue behind the first tile's stores without a launch-wide phase change)
    if constexpr (BNI) {
        // the coefficient table of the fused BatchNorm: [groups][Kc] x (scale, shift), 8 bytes per channel
        const int n4 = a.bni_groups * a.Kc / 2;
        for (int i = tid; i < n4; i += 256)
            reinterpret_cast<float4*>(smem + a.bni_lds)[i] = reinterpret_cast<const float4*>(a.bni)[i];
        __syncthreads();
    }
    double S1 = 0.0, S2 = 0.0;                          // stat_acc: column sums over this workgroup's tile positions
    float AMX = 0.0f;                                   // (a.amax_out: max |mask(d)| over this workgroup's tiles)
    int Simg = 0, Sg = 0x7fffffff, Sbx = 0;
    if (a.desync > 0 && (((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 1)) {
        for (int i = 0; i < a.desync; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int it = 0; it < a.tpb; ++it) {
    const int bx = blockIdx.x * a.tpb + it;
    if (bx >= a.gx_total) break;
    C3_STAMP(bx * gridDim.y + blockIdx.y, 0);
#ifdef BH_TUNING
    if (a.dbg_ts && threadIdx.x == 0 && bx * gridDim.y + blockIdx.y < 16384) {
        unsigned hw_, xcc_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        g_c3_ts[(bx * gridDim.y + blockIdx.y) * 8 + 4] = ((unsigned long long)xcc_ << 32) | hw_;
    }
#endif
    // ---- halo slots of this lane: slot q = (j*4 + wave)*64 + lane of the [8][SUBT][100] image ----
    // (written for instruction count - VALU work does not overlap the other workgroup's MFMAs: the per-sub-tile origin
    //  is computed once, with shifts when the tile grid is a power of two, and the slot -> (plane, sub-tile, hy, hx)
    //  decoding uses small-range multiply-shift divisions)
    int org[SUBT];                                   // pixel index of halo position (0, 0) of each sub-tile, or INT_MIN
    int oy0[SUBT], ox0[SUBT];
    int bni_grp[SUBT];                               // BNI: statistics group of each sub-tile's image
#pragma unroll
    for (int s = 0; s < SUBT; ++s) {
        const int g = bx * SUBT + s;
        int img, ty, tx;
        if constexpr (MAP4) { img = g * 4; ty = 0; tx = 0; }
        else if (a.tpi_shift >= 0) { img = g >> a.tpi_shift; const int t = g & (a.tiles_per_img - 1); ty = t >> a.tx_shift; tx = t & (a.tiles_x - 1); }
        else { img = g / a.tiles_per_img; const int t = g - img * a.tiles_per_img; ty = t / a.tiles_x; tx = t - ty * a.tiles_x; }
        oy0[s] = ty * 8 - 1; ox0[s] = tx * 8 - 1;
        org[s] = g < a.subtiles ? (MAP4 ? img * 16 : (img * a.H + oy0[s]) * a.W + ox0[s]) : (int)0x80000000;
        if constexpr (BNI) bni_grp[s] = g < a.subtiles ? img / a.bni_ipg : 0;
    }
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    if constexpr (X3) {
        // ---- halo slots of this thread: slot q = j*256 + tid of the [4 k-planes][SUBT][100] image (8 channels = 32 B each) ----
        constexpr unsigned XOOB = 0x80000000u;           // (tensor sizes are below 2^31: stays out of range with the +16 / chunk offsets added)
        uint4 bcur[2 * NP];
#if !C3_X3_PIPE
        uint4 bnext[2 * NP];
#endif
        C3_LOAD_BX(bcur, 0, 0);
        unsigned xoff[XJ];
        int xtb[XJ];                                     // BNI: byte offset of the slot's 8 coefficients pairs in the LDS table (chunk 0)
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int q = j * 256 + tid;
            unsigned off = XOOB;
            xtb[j] = 0;
            if constexpr (MAP4) {
                if (q < 4 * HPL) {
                    const int plane = q / 144, hp = q - plane * 144;                        // slot = (plane, image of the sub-tile, 6 x 6 halo position)
                    const int im = hp / 36, r = hp - im * 36, hy = r / 6, hx = r - hy * 6;
                    if (org[0] != (int)0x80000000 && (unsigned)(hy - 1) < 4u && (unsigned)(hx - 1) < 4u)
                        off = ((unsigned)(org[0] + im * 16 + (hy - 1) * 4 + (hx - 1)) * (unsigned)a.Kc + (unsigned)(plane * 8)) * 4u;
                }
            } else
            if (q < 4 * HPL) {
                const int plane = SUBT == 2 ? (q * 5243) >> 20 : (q * 10486) >> 20;      // q / 200, q / 100
                const int rem = q - plane * HPL;
                const int s = SUBT == 2 ? (rem >= 100 ? 1 : 0) : 0, hp = rem - s * 100;
                const int hy = (hp * 205) >> 11, hx = hp - hy * 10;
                const int y = (s ? oy0[SUBT - 1] : oy0[0]) + hy, x = (s ? ox0[SUBT - 1] : ox0[0]) + hx;
                const int o = s ? org[SUBT - 1] : org[0];
                if (o != (int)0x80000000 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                    off = ((unsigned)(o + hy * a.W + hx) * (unsigned)a.Kc + (unsigned)(plane * 8)) * 4u;
                if constexpr (BNI) xtb[j] = a.bni_lds + ((s ? bni_grp[SUBT - 1] : bni_grp[0]) * a.Kc + plane * 8) * 8;
            }
            xoff[j] = off;
        }
        const float bni_lo = a.bni_relu ? 0.0f : -__builtin_inff();   // v_maximum3_f32: NaN propagates (fmaxf would drop it), -inf passes everything
        // y = max(x * scale + shift, lo) on the 8 channels of a slot; out-of-image slots stay zero
#define X3_BNI(j, c, h)                                                                                                 \
    do {                                                                                                                \
        if constexpr (BNI) {                                                                                            \
            const float4* tb_ = reinterpret_cast<const float4*>(smem + xtb[j] + (c) * 256);                             \
            const float4 t0_ = tb_[0], t1_ = tb_[1], t2_ = tb_[2], t3_ = tb_[3];                                       \
            const bool ok_ = xoff[j] != XOOB;                                                                           \
            h[0].x = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[0].x, t0_.x, t0_.y), bni_lo) : 0.f;                                   \
            h[0].y = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[0].y, t0_.z, t0_.w), bni_lo) : 0.f;                                   \
            h[0].z = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[0].z, t1_.x, t1_.y), bni_lo) : 0.f;                                   \
            h[0].w = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[0].w, t1_.z, t1_.w), bni_lo) : 0.f;                                   \
            h[1].x = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[1].x, t2_.x, t2_.y), bni_lo) : 0.f;                                   \
            h[1].y = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[1].y, t2_.z, t2_.w), bni_lo) : 0.f;                                   \
            h[1].z = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[1].z, t3_.x, t3_.y), bni_lo) : 0.f;                                   \
            h[1].w = ok_ ? __builtin_elementwise_maximum(__builtin_fmaf(h[1].w, t3_.z, t3_.w), bni_lo) : 0.f;                                   \
        }                                                                                                               \
    } while (0)
        // (rounds whose first slot of this wave is past the image are skipped wave-uniformly: 800 = 3 x 256 + 32 slots)
#define X3_ISSUE(j, c, h)                                                                                               \
    do {                                                                                                                \
        if ((j) * 256 + wave * 64 < 4 * HPL) {                                                                          \
            h[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xoff[j], (unsigned)((c) * 128), 0));       \
            h[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xoff[j] + 16u, (unsigned)((c) * 128), 0)); \
        }                                                                                                               \
    } while (0)
#define X3_STORE(j, hs, c, h)                                                                                           \
    do {                                                                                                                \
        if ((j) * 256 + wave * 64 < 4 * HPL && (j) * 256 + tid < 4 * HPL) {                                             \
            uint4 p_[3];                                                                                                \
            X3_BNI(j, c, h);                                                                                            \
            bh_split8_any<NP, F16>(h[0], h[1], f16_s, p_);                                                              \
            char* d_ = smem + (hs) * HALO_B + ((j) * 256 + tid) * 16;                                                   \
            _Pragma("unroll") for (int pc_ = 0; pc_ < NP; ++pc_)                                                        \
                *reinterpret_cast<uint4*>(d_ + pc_ * 4 * HPL * 16) = p_[pc_];                                           \
        }                                                                                                               \
    } while (0)
        {
            float4 hp[XJ][2];
#pragma unroll
            for (int j = 0; j < XJ; ++j) X3_ISSUE(j, 0, hp[j]);
#pragma unroll
            for (int j = 0; j < XJ; ++j) X3_STORE(j, 0, 0, hp[j]);
        }
        const int sr_ = c3_strip_row(l31 >> 2);
        const int a_lane = MAP4 ? (kh2 * HPL + (2 * wh + (sr_ >> 2)) * 36 + (sr_ & 3) * 6 + (l31 & 3)) * 16
                                : (kh2 * HPL + wm * 100 + sr_ * 10 + wh * 4 + (l31 & 3)) * 16;
        __syncthreads();
        C3_STAMP(bx * gridDim.y + blockIdx.y, 1);
        float4 hb[2][2];
#define X3_MFMA(afc, bc, s2, PA, PB)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                      \
        acc[i] = c3_mfma16<F16>(afc[i][s2][PA], bc[(PB) * 2 + (s2)], acc[i])
        // A fragments of one tap: [fragment][16-channel step][piece], 4 * NP ds_read_b128 per wave (TM = 2)
#define X3_LOAD_A(dst, ap)                                                                                              \
    _Pragma("unroll") for (int s2_ = 0; s2_ < 2; ++s2_)                                                                 \
        _Pragma("unroll") for (int pc_ = 0; pc_ < NP; ++pc_)                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                           \
                dst[i_][s2_][pc_] = *reinterpret_cast<const uint4*>((ap) + (pc_ * 4 + 2 * s2_) * (HPL * 16) + i_ * 64)
#if C3_X3_PIPE
        // Software-pipelined tap loop (round 3).  The compiler's schedule of the plain loop below requested the fragments and the
        // weights of tap t + 1 at the END of tap t and waited for them (lgkmcnt / vmcnt) one or two MFMAs into tap t + 1: LDS and
        // L2 latency exposed once per tap and wave.  Here a tap is two halves of 4 * NP MFMAs (16-channel step s2 = 0 / 1):
        //   weights of tap t + 1        -> requested at the start of tap t into the OTHER weight register set (a full tap ahead;
        //                                  the sets alternate per tap, nine taps: the chunk loop is unrolled by two, no copies);
        //   fragments, step 1 of tap t  -> requested at the start of tap t, into the registers its predecessor's second half freed;
        //   fragments, step 0 of t + 1  -> requested between the halves of tap t, into the registers the first half just freed
        // - every request has >= 4 * NP MFMAs (384 cycles) to arrive and no register is added.  sched_barriers pin the order.
        // The halo of the next chunk is requested at taps 0 .. XJ-1 and cut / written at taps 2 .. XJ+1; ONE barrier per chunk
        // between the halves of tap 8 (every wave has then read its last fragments of this stage and written its part of the
        // next one), after which step 0 of the next chunk's tap 0 comes from the other stage.
#define X3_LOAD_A_HALF(dst, ap, s2_)                                                                                    \
    _Pragma("unroll") for (int pc_ = 0; pc_ < NP; ++pc_)                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                               \
            dst[i_][s2_][pc_] = *reinterpret_cast<const uint4*>((ap) + (pc_ * 4 + 2 * (s2_)) * (HPL * 16) + i_ * 64)
#define X3_MFMA_HALF(afc, bc, s2)                                                                                       \
    do {                                                                                                                \
        if constexpr (NP == 3) { X3_MFMA(afc, bc, s2, 2, 0); X3_MFMA(afc, bc, s2, 0, 2); X3_MFMA(afc, bc, s2, 1, 1); }  \
        X3_MFMA(afc, bc, s2, 1, 0); X3_MFMA(afc, bc, s2, 0, 1); X3_MFMA(afc, bc, s2, 0, 0);                             \
    } while (0)
        uint4 af[TM][2][NP], bX[2 * NP], bY[2 * NP];
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) bX[q] = bcur[q];
        X3_LOAD_A_HALF(af, smem + a_lane, 0);
        auto tap_step = [&](auto TAP, const int c, const bool more, auto& bc, auto& bn) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value;
            constexpr int dy = tap / 3, dx = tap - dy * 3, dy1 = (tap + 1) / 3, dx1 = (tap + 1) - dy1 * 3;
            const int hs = c & 1;
            const char* const hbase = smem + hs * HALO_B + a_lane;
            if constexpr (tap < 8) C3_LOAD_BX(bn, c, tap + 1);
            else if (more) C3_LOAD_BX(bn, c + 1, 0);
            X3_LOAD_A_HALF(af, hbase + (dy * ROWP + dx) * 16, 1);
            if (more) {
                if constexpr (tap >= 2 && tap - 2 < XJ) X3_STORE(tap - 2, hs ^ 1, c + 1, hb[(tap - 2) & 1]);
                if constexpr (tap < XJ) X3_ISSUE(tap, c + 1, hb[tap & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);         // requests first: nothing of the above sinks into the MFMA block
            X3_MFMA_HALF(af, bc, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tap < 8) X3_LOAD_A_HALF(af, hbase + (dy1 * ROWP + dx1) * 16, 0);
            else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (more) X3_LOAD_A_HALF(af, smem + (hs ^ 1) * HALO_B + a_lane, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            X3_MFMA_HALF(af, bc, 1);
            __builtin_amdgcn_sched_barrier(0);
        };
#define X3_TAP(t, c, more, CUR, NXT) tap_step(std::integral_constant<int, t>{}, c, more, b##CUR, b##NXT)
#define X3_CHUNK(c, more, P, Q)                                                                                         \
    do {                                                                                                                \
        X3_TAP(0, c, more, P, Q); X3_TAP(1, c, more, Q, P); X3_TAP(2, c, more, P, Q); X3_TAP(3, c, more, Q, P);         \
        X3_TAP(4, c, more, P, Q); X3_TAP(5, c, more, Q, P); X3_TAP(6, c, more, P, Q); X3_TAP(7, c, more, Q, P);         \
        X3_TAP(8, c, more, P, Q);                                                                                       \
    } while (0)
        {
            int c = 0;
            for (; c + 1 < nch; c += 2) {
                X3_CHUNK(c, true, X, Y);
                X3_CHUNK(c + 1, c + 2 < nch, Y, X);
            }
            if (c < nch) X3_CHUNK(c, false, X, Y);
        }
#undef X3_CHUNK
#undef X3_TAP
#undef X3_MFMA_HALF
#undef X3_LOAD_A_HALF
#else
        uint4 af[TM][2][NP];                           // [fragment][16-channel step][piece]
        for (int c = 0; c < nch; ++c) {
            const int hs = c & 1;
            const char* hbase = smem + hs * HALO_B + a_lane;
            const bool more = c + 1 < nch;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const char* ap = hbase + (dy * ROWP + dx) * 16;
                if (!(dbg_noload & 4) || (tap == 0 && c == 0)) X3_LOAD_A(af, ap);
                if (!(dbg_noload & 1)) {
                if (tap < 8) C3_LOAD_BX(bnext, c, tap + 1);
                else if (more) C3_LOAD_BX(bnext, c + 1, 0);
                }
                if (more && !(dbg_noload & 2)) {
                    // the next chunk's halo: slot round j is requested at tap 2j and cut / written two taps later
                    if ((tap & 1) == 0 && tap >= 2 && tap / 2 - 1 < XJ) X3_STORE(tap / 2 - 1, hs ^ 1, c + 1, hb[(tap / 2 - 1) & 1]);
                    if ((tap & 1) == 0 && tap / 2 < XJ) X3_ISSUE(tap / 2, c + 1, hb[(tap / 2) & 1]);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {       // small partial products first
                    if constexpr (NP == 3) { X3_MFMA(af, bcur, s2, 2, 0); X3_MFMA(af, bcur, s2, 0, 2); X3_MFMA(af, bcur, s2, 1, 1); }
                    X3_MFMA(af, bcur, s2, 1, 0); X3_MFMA(af, bcur, s2, 0, 1); X3_MFMA(af, bcur, s2, 0, 0);
                }
                if (tap == 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!(dbg_noload & 1)) {
#pragma unroll
                    for (int q = 0; q < 2 * NP; ++q) bcur[q] = bnext[q];
                }
            }
        }
#endif
#undef X3_MFMA
#undef X3_LOAD_A
#undef X3_ISSUE
#undef X3_STORE
#undef X3_BNI
    } else {
    float4 bcur[4], bnext[4];
    if constexpr (PACKED) C3_LOAD_B(bcur, 0, 0);
    else issue_B(0, 0, 0);                       // the first weight slab goes out before any of the slot arithmetic
    unsigned hoff[7] = {OOB, OOB, OOB, OOB, OOB, OOB, OOB};   // (fixed size, first HJ used: see the note at boff)
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int ci = j * 4 + wave, q = ci * 64 + lane;     // q < 1664
        unsigned off = OOB;
        if (ci < HINS && q < 8 * HPL) {
            const int plane = SUBT == 2 ? (q * 5243) >> 20 : (q * 10486) >> 20;      // q / 200, q / 100 for q < 1700
            const int rem = q - plane * HPL;
            const int s = SUBT == 2 ? (rem >= 100 ? 1 : 0) : 0, hp = rem - s * 100;
            const int hy = (hp * 205) >> 11, hx = hp - hy * 10;                        // hp / 10 for hp < 100
            const int y = (s ? oy0[SUBT - 1] : oy0[0]) + hy, x = (s ? ox0[SUBT - 1] : ox0[0]) + hx;
            const int o = s ? org[SUBT - 1] : org[0];
            if (o != (int)0x80000000 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                off = ((unsigned)(o + hy * a.W + hx) * (unsigned)a.Kc + (unsigned)(plane * 4)) * 4u;
        }
        hoff[j] = off;
        C3_ISSUE_HALO(j, 0, 0);              // in flight while the next slot's offset is computed
    }
    // lane-constant parts of the fragment addresses (bytes)
    // GEMM row l31 of a wave's 32-row fragment <-> pixel (c3_strip_row(l31 >> 2), 4 * strip + (l31 & 3)) of the 8x8 sub-tile:
    // an 8-row x 4-column strip, strip = wh for the 32-row waves, = the fragment index i for the 64-row waves.
    // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), 64 banks of 4 B: with this
    // order the first group reads the even rows of the strip (halo slots 0-3, 20-23, 40-43, 60-63 = all sixteen 16-byte
    // slots of a 256-byte bank row), the second the odd rows - conflict-free for every tap shift; the row-major 4x8
    // block used before was 3-way conflicted (13.8 vs 8.2 LDS cycles per read, tools/lds_pattern_bench.hip).
    const int a_lane = (kh2 * HPL + wm * 100 + c3_strip_row(l31 >> 2) * 10 + wh * 4 + (l31 & 3)) * 16;
    const int b_lane = FLIP ? (kh2 * 4 * BN + wn * 32 + l31) * 4 : (kh2 * BN + wn * 32 + l31) * 16;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int bs = 0;
    if constexpr (PACKED) {
        for (int c = 0; c < nch; ++c) {
            const int hs = c & 1;
            const char* hbase = smem + hs * HALO_B + a_lane;
            const bool more = c + 1 < nch;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const char* ap = hbase + (dy * 10 + dx) * 16;
                float4 af[TM][4];
                // A fragments first: the compiler puts a vmcnt(0) in front of LDS reads that follow an LDS-DMA, and at
                // this point everything in flight (B of this tap, the previous tap's halo piece) is a whole tap old
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const float4*>(ap + q * (2 * HPL * 16) + i * 64);
                __builtin_amdgcn_sched_barrier(0);
                if (tap < 8) C3_LOAD_B(bnext, c, tap + 1);
                else if (more) C3_LOAD_B(bnext, c + 1, 0);
                if (tap < HJ && more) C3_ISSUE_HALO(tap, c + 1, hs ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF16) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 bb = c3_pack_bf16(bcur[2 * s2], bcur[2 * s2 + 1]);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3_pack_bf16(af[i][2 * s2], af[i][2 * s2 + 1]), bb, acc[i], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].x, bcur[q].x, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].y, bcur[q].y, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].z, bcur[q].z, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].w, bcur[q].w, acc[i], 0, 0, 0);
                    }
                }
                if (tap == 8) {
                    // chunk boundary: every wave's halo pieces of the next chunk have landed and nobody still reads this stage
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) bcur[q] = bnext[q];
            }
        }
    } else {
    for (int c = 0; c < nch; ++c) {
        const int hs = c & 1;
        const char* hbase = smem + hs * HALO_B + a_lane;
        const bool more = c + 1 < nch;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // all fragment reads of the step come first: the compiler orders every later LDS read behind an
            // outstanding LDS-DMA with a full vmcnt(0), so the prefetch is issued only after them
            const int dy = tap / 3, dx = tap - dy * 3;
            const char* ap = hbase + (dy * 10 + dx) * 16;
            const char* bp = smem + slab0 + bs * C3_B_BYTES + b_lane;
            float4 af[TM][4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const float4*>(ap + q * (2 * HPL * 16) + i * 64);
                if (!FLIP) b[q] = *reinterpret_cast<const float4*>(bp + q * (2 * BN * 16));
                else {
                    b[q].x = *reinterpret_cast<const float*>(bp + (q * 8 + 0) * (BN * 4));
                    b[q].y = *reinterpret_cast<const float*>(bp + (q * 8 + 1) * (BN * 4));
                    b[q].z = *reinterpret_cast<const float*>(bp + (q * 8 + 2) * (BN * 4));
                    b[q].w = *reinterpret_cast<const float*>(bp + (q * 8 + 3) * (BN * 4));
                }
            }
            if (!(dbg_noload & 1)) {
                if (tap < 8) issue_B(c, tap + 1, bs ^ 1);
                else if (more) issue_B(c + 1, 0, bs ^ 1);
            }
            if (tap < HJ && more && !(dbg_noload & 2)) C3_ISSUE_HALO(tap, c + 1, hs ^ 1);
            if constexpr (BF16) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 bb = c3_pack_bf16(b[2 * s2], b[2 * s2 + 1]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3_pack_bf16(af[i][2 * s2], af[i][2 * s2 + 1]), bb, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].x, b[q].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].y, b[q].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].z, b[q].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].w, b[q].w, acc[i], 0, 0, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the MFMAs of this step in front of the wait: they hide the DMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            bs ^= 1;
        }
    }
    }
    }
#undef C3_ISSUE_HALO
#undef issue_B
#undef C3_LOAD_B
#undef C3_LOAD_BX

    C3_STAMP(bx * gridDim.y + blockIdx.y, 2);
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = __builtin_ldexpf(acc[i][r], f16_kout);
    }
    double s1, s2;
    int img, g;
    c3_epilogue<BN, SUBT, TM, MAP4>(a, acc, bx, n0, wm, wn, wh, l31, kh2, s1, s2, img, g, AMX);
    if (a.bn_sums) {
        if (a.stat_acc) { S1 += s1; S2 += s2; Sbx = bx; if (g < Sg) { Sg = g; Simg = img; } }    // (out-of-range sub-tiles add zero)
        else c3_stats_merge<BN, 0>(a, smem, s1, s2, img, g, bx, n0, wave, lane, tid);
    }
#ifdef BH_TUNING
    if (a.dbg_ts) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (stores of the tile have left the wave)
#endif
    __syncthreads();            // the next tile's DMA overwrites the LDS this tile's statistics merge just read
    C3_STAMP(bx * gridDim.y + blockIdx.y, 3);
    }
    if (a.bn_sums && a.stat_acc) c3_stats_merge<BN, 0>(a, smem, S1, S2, Simg, Sg, Sbx, n0, wave, lane, tid);
    if (a.amax_out) {                                    // one integer atomic max per wave (bits of a non-negative float order like integers)
        const float m = wave_max(AMX);
        if (lane == 0) atomicMax(a.amax_out + ((blockIdx.x * 4 + wave) % BH_AMAX_SLOTS) * BH_AMAX_STRIDE, __builtin_bit_cast(unsigned, m));
    }
}

constexpr int C3_MIN_BLOCKS = 256;      // below this many workgroups the generic kernel's finer tiles fill the chip better
BH_KNOB(g_c3_noload, 0); BH_KNOB(g_c3_dbg_nch, -1); BH_KNOB(g_c3_subt, 2); BH_KNOB(g_c3_tpb, 2); BH_KNOB(g_c3_stamp, 0); BH_KNOB(g_c3_desync, 0); BH_KNOB(g_c3_walk32, 1); BH_KNOB(g_c3_pc, 1); BH_KNOB(g_c3_wt64, 512); BH_KNOB(g_c3_wt32, 512);
#ifdef BH_TUNING
// copies the phase time stamps of the last instrumented launch to the host (n entries of 8 x u64)
extern "C" int bh_debug_read_c3_stamps(unsigned long long* out, int n) {
    if (n > 16384) n = 16384;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c3_ts), (size_t)n * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
void bh_conv3x3_tune(int disable, int min_blocks) {
    (void)min_blocks;
    if (disable >= 500 && disable < 1000) { g_c3_stamp = disable - 500; return; }     // (-40, n): phase time stamps off (0) / on (halo kernel: any n > 0; persistent kernel: workgroup n - 1)
    if (disable >= 200 && disable < 264) { g_c3_desync = disable - 200; return; }
    if (disable == 300 || disable == 301) { g_c3_walk32 = disable - 300; return; }         // (-43, 0|1): several positions per workgroup in the 32-channel launches without statistics off / on
    if (disable >= 2000 && disable < 6000) { g_c3_wt64 = disable - 2000; return; }        // (-45, n): workgroups per statistics launch, 64-channel tile
    if (disable >= 6000 && disable < 10000) { g_c3_wt32 = disable - 6000; return; }       // (-46, n): the same, 32-channel tile (default 512)
    if (disable == 310 || disable == 311) { g_c3_pc = disable - 310; return; }            // (-44, 0|1): persistent producer / consumer kernel off / on
    if (disable <= -100) { g_c3_dbg_nch = -100 - disable - 1; return; }            // -100 -> -1 (all), -101 -> 0 chunks, -102 -> 1 ...
    if (disable >= 400 && disable < 464) { g_c3_noload = disable - 400; return; }      // (-18, bits): ablation bits (halo kernel 1 2 4; producer / consumer kernel 1 .. 16)
    if (disable >= 20 && disable < 24) { g_c3_tpb = disable - 20; return; }        // tile positions per workgroup on two-round launches (1 / 2)
    if (disable >= 11 && disable <= 13) { g_c3_subt = disable - 10; return; }      // 1 / 2 (automatic) / 3 (always two) sub-tiles per workgroup
}
#endif

// *taken = 1 when the shape is eligible and the launch was made; returns BH_OK or a hipError_t
int bh_conv3x3_try(const float* src, const float* w, const float* bias, float* out, const bh_conv_desc* d, int dgrad,
                   int accumulate, hipStream_t stream, int* taken, double* bn_sums, int groups, const float* res, int relu,
                   const bh_bn_reduce* bnr, const bh_bn_in* bni) {
    *taken = 0;
    // (packed weights only make sense to this kernel: a caller that passes them must have asked bh_conv_variant first)
    if ((d->route & BH_ROUTE_GENERIC_CONV) || d->transposed || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->in_nchw ||
        d->out_nchw || d->precision < 0 || d->precision > 4)
        return d->w_layout ? BH_E_UNSUPPORTED : 0;
    // split weights <-> their precision: w_layout 2 = three bf16 pieces (precision 2), 3 = two bf16 pieces (precision 3), 4 = two fp16
    // pieces with power-of-two scales (precision 4: needs the magnitude record of the source tensor, bh_conv_desc.a_bound)
    if (d->w_layout != 0 && ((d->w_layout == 2) != (d->precision == 2) || (d->w_layout == 3) != (d->precision == 3) ||
                             (d->w_layout == 4) != (d->precision == 4))) return BH_E_BADARG;
    if (d->w_layout == 4 && !d->a_bound) return BH_E_BADARG;
    // 4 x 4 maps (round 3: layer4 of the ResNet-34 regressor): split-operand form only, four images per sub-tile
    const bool map4 = d->Hi == 4 && d->Wi == 4 && d->Ho == 4 && d->Wo == 4 && d->N % 4 == 0 && (d->w_layout >= 2 && d->w_layout <= 4) &&
                      d->Co % 64 == 0 && d->Ci % 64 == 0 && !bni && (!bn_sums || (groups >= 1 && d->N % groups == 0 && (d->N / groups) % 4 == 0));
    if (!map4 && (d->Hi % 8 || d->Wi % 8 || d->Ho != d->Hi || d->Wo != d->Wi)) return d->w_layout ? BH_E_UNSUPPORTED : 0;
    const int Kc = dgrad ? d->Co : d->Ci, Nn = dgrad ? d->Ci : d->Co;
    if (Kc % 32 || Nn % 32) return d->w_layout ? BH_E_UNSUPPORTED : 0;
    const int bn_tile = (Nn % 64) ? 32 : 64;
    const bool packed = d->w_layout != 0;              // weights in bh_conv3x3_pack fragment order (this direction's buffer)
    const bool x3 = d->w_layout >= 2 && d->w_layout <= 4;   // ... cut into three / two bf16 pieces or two fp16 pieces (6 / 4 / 4 bytes per weight)
    const int np = d->w_layout >= 3 ? 2 : 3;
    const bool f16 = d->w_layout == 4;
    const bool bf16 = d->precision == 1;               // (precision 2 / 3 without split weights runs the exact fp32 form)
    const long long src_bytes = (long long)d->N * d->Hi * d->Wi * Kc * 4, w_bytes = (long long)d->Co * 9 * d->Ci * (x3 ? 2 * np : 4);
    const long long out_bytes = (long long)d->N * d->Hi * d->Wi * Nn * 4;
    if (src_bytes >= (1ll << 31) || w_bytes >= (1ll << 31) || out_bytes >= (1ll << 31)) return 0;
    C3Args a = {};
    a.Src = src; a.Wt = w; a.bias = bias; a.Out = out;
    a.N = d->N; a.H = d->Hi; a.W = d->Wi; a.Kc = Kc; a.Nn = Nn; a.Cw = d->Ci; a.accumulate = accumulate;
    a.amax_src = reinterpret_cast<const unsigned*>(d->a_bound);
    a.src_bytes = (unsigned)src_bytes; a.w_bytes = (unsigned)w_bytes; a.out_bytes = (unsigned)out_bytes;
    // sums: forward -> BatchNorm statistics of the output; dgrad + bnr -> that BatchNorm's backward sums; dgrad without
    // bnr -> plain per-channel (sum, sum of squares) of the gradient written (its column sums = the bias gradient of the
    // layer that produced this conv's input: bh_conv_dgrad_colsum), only on the non-accumulating store path
    if (bn_sums && (groups < 1 || d->N % groups || (!dgrad && bnr) || (dgrad && !bnr && accumulate))) return BH_E_BADARG;
    if (bnr) {
        if (!bnr->z || !bnr->stats) return BH_E_BADARG;
        a.bnr_z = bnr->z; a.bnr_y = bnr->y; a.bnr_stats = bnr->stats; a.bnr_gamma = bnr->gamma; a.bnr_beta = bnr->beta;
        a.bnr_eps = bnr->eps; a.bnr_relu = bnr->relu; a.bnr_rows = (d->N / groups) * d->Hi * d->Wi;
        a.amax_out = reinterpret_cast<unsigned*>(bnr->amax_d);
    }
    a.res = res; a.relu = relu; a.dbg_nch = g_c3_dbg_nch;
    a.bn_sums = bn_sums; a.imgs_per_group = bn_sums ? d->N / groups : 1; a.groups = groups;
    a.tiles_x = d->Wi / 8; a.tiles_per_img = (d->Hi / 8) * a.tiles_x; a.subtiles = d->N * a.tiles_per_img;
    if (map4) { a.tiles_x = 1; a.tiles_per_img = 1; a.subtiles = d->N / 4; }
    a.tx_shift = a.tpi_shift = -1;
    for (int b = 0; b < 24; ++b) {
        if (a.tiles_x == (1 << b)) a.tx_shift = b;
        if (a.tiles_per_img == (1 << b)) a.tpi_shift = b;
    }
    if (a.tx_shift < 0 || a.tpi_shift < 0) a.tx_shift = a.tpi_shift = -1;
    // one sub-tile per workgroup where two would leave half of the 512 workgroup slots empty (the 8x8x256-channel layers:
    // +6 %); elsewhere the 128-row tile is as fast or faster (measured, tools/conv3x3_check.py --subt1)
    const bool few = (long long)((a.subtiles + 1) / 2) * (Nn / bn_tile) <= 256;
    const int subt = map4 ? 1 : (bn_tile == 64 && (g_c3_subt == 1 || (d->route & BH_ROUTE_C3_ONE_SUBTILE) || (g_c3_subt == 2 && few))) ? 1 : 2;
    dim3 grid((a.subtiles + subt - 1) / subt, Nn / bn_tile);
    a.gx_total = (int)grid.x; a.tpb = 1;
    {   // exactly-two-rounds launches (two resident workgroups per CU with two sub-tiles, 512 slots): one round of two tiles
        const long long wgs = (long long)grid.x * grid.y;
        if (g_c3_tpb >= 2 && !(d->route & BH_ROUTE_C3_ONE_POSITION) && subt == 2 && Kc / 32 > 1 && wgs > 512 && wgs <= 1024) {
            a.tpb = 2; grid.x = (grid.x + 1) / 2;
        }
    }
    if (bn_sums && !map4 && !(d->route & BH_ROUTE_C3_ONE_POSITION) && groups >= 1 && a.subtiles % (subt * groups) == 0) {
        // statistics epilogues: every workgroup ends with one f64 atomic per (channel, moment) and the memory-side atomic unit takes
        // ~30 ns per same-address atomic - keep the workgroups per statistics group at <= 1024 by walking several tile positions per
        // workgroup (all inside one group) and adding their column sums in registers
        // (round 5, in the two-stream step: 512 workgroups for the 32-channel tile as well - 12.64 against 12.71 ms in four alternating runs,
        //  256 costs 0.16 ms; alone 1024 was the optimum.  64-channel tile: 256 / 512 / 1024 within 0.02 ms.  tools/ab_hook.sh "-46,n" / "-45,n")
        const int ppg = a.subtiles / (subt * groups);            // tile positions per statistics group
        int t = a.tpb;
        const long long wtarget = (x3 && bn_tile == 64) ? g_c3_wt64 : (x3 ? g_c3_wt32 : 2048);    // workgroups of the launch (all groups, all N tiles)
        while (ppg % (t * 2) == 0 && (ppg / t > 1024 || (long long)(ppg / t) * groups * grid.y > wtarget) && t < 64) t *= 2;
        if (ppg % t == 0 && (t > 1 || a.tpb == 1)) {
            if (t != a.tpb) { a.tpb = t; grid.x = (a.gx_total + t - 1) / t; }
            a.stat_acc = 1;
        }
    }
    if (x3 && !map4 && a.tpb == 1 && !bn_sums && g_c3_walk32 && !(d->route & BH_ROUTE_C3_ONE_POSITION)) {
        // split-operand launches without statistics (the dgrads of the big layers): several tile positions per workgroup as well - down to
        // ~1024 workgroups with the 32-channel tile (from 4-16 thousand: 252 -> 223 us at 128 x 128 x 32, 63 -> 60.5 us at 64 x 64 x 32),
        // ~512 with the 64-channel tile (142 -> 128 us at 64 x 64 x 64, 117 -> 111 us at 32 x 32 x 128, nothing at 16 x 16 x 256);
        // tools/c3_walk32.py, profiles/r04_c3_f16_ablation.txt (h).  (Requesting the next position's halo under the current
        // position's taps on top of the walk LOSES 5 %: same record.)
        const long long target = bn_tile == 32 ? 1024 : 512;
        int t = 1;
        while (a.gx_total % (t * 2) == 0 && (long long)(a.gx_total / (t * 2)) * grid.y >= target && t < 64) t *= 2;
        if (t > 1) { a.tpb = t; grid.x = a.gx_total / t; }
    }
    if (!(d->route & BH_ROUTE_HALO_SMALL) && (int)((map4 ? a.subtiles : (a.subtiles + 1) / 2) * grid.y) < (map4 ? C3_MIN_BLOCKS / 2 : C3_MIN_BLOCKS))
        return d->w_layout ? BH_E_UNSUPPORTED : 0;
    a.NW = Nn / 32;
    if (bni) {      // BatchNorm-on-load: the f32x3 forward only, table of <= 4 KB (two groups x 256 channels) in LDS
        if (!x3 || dgrad || !bni->table || bni->groups < 1 || d->N % bni->groups || (long long)bni->groups * Kc * 8 > 4096) return BH_E_UNSUPPORTED;
        a.bni = bni->table; a.bni_relu = bni->relu; a.bni_groups = bni->groups; a.bni_ipg = d->N / bni->groups;
    }
    // fp16-piece launches of the 64-channel tile: persistent producer / consumer workgroups (conv3x3_pc.hip, round 5) unless the call routes
    // to the one-workgroup-per-tile kernel below (BH_ROUTE_C3_TILE_WG: A/B measurements, the bit-identity tests)
    if (f16 && !map4 && bn_tile == 64 && !(d->route & (BH_ROUTE_C3_TILE_WG | BH_ROUTE_C3_ONE_SUBTILE | BH_ROUTE_C3_ONE_POSITION)) && g_c3_pc) {
        C3Args b = a;
        b.det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0; b.dbg_noload = g_c3_noload; b.dbg_ts = g_c3_stamp;
        const int st = bh_conv3x3_pc_launch(b, dgrad, bni ? bni->table : nullptr, bni ? bni->groups : 0, bni ? bni->relu : 0, (d->route & BH_ROUTE_C3_PC) != 0, stream);
        if (st != BH_E_UNSUPPORTED) { if (st == BH_OK) *taken = 1; return st; }
    }
    // (all ten template arguments, as rocprofv3 prints the symbol: FLIP, BN, BF16, SUBT, PACKED, X3, BNI, NP, MAP4, F16)
    if (bh_query("conv3x3_halo_kernel<%s,%d,%s,%d,%s,%s,%s,%d,%s,%s>", dgrad ? "true" : "false", bn_tile, bf16 ? "true" : "false", subt,
                 packed ? "true" : "false", x3 ? "true" : "false", bni ? "true" : "false", np, map4 ? "true" : "false", f16 ? "true" : "false")) {
        *taken = 1;
        return BH_OK;
    }
    static unsigned long long attr_devs = 0;             // devices on which the dynamic-LDS attributes have been set
    typedef void (*kern_t)(C3Args);
#define C3_ROW(P) conv3x3_halo_kernel<false, 64, false, 2, P>, conv3x3_halo_kernel<true, 64, false, 2, P>,   \
                  conv3x3_halo_kernel<false, 32, false, 2, P>, conv3x3_halo_kernel<true, 32, false, 2, P>,   \
                  conv3x3_halo_kernel<false, 64, true, 2, P>,  conv3x3_halo_kernel<true, 64, true, 2, P>,    \
                  conv3x3_halo_kernel<false, 32, true, 2, P>,  conv3x3_halo_kernel<true, 32, true, 2, P>,    \
                  conv3x3_halo_kernel<false, 64, false, 1, P>, conv3x3_halo_kernel<true, 64, false, 1, P>,   \
                  conv3x3_halo_kernel<false, 64, true, 1, P>,  conv3x3_halo_kernel<true, 64, true, 1, P>
#define C3_XROW(NP_, F_) conv3x3_halo_kernel<false, 64, false, 2, true, true, false, NP_, false, F_>, conv3x3_halo_kernel<true, 64, false, 2, true, true, false, NP_, false, F_>,  \
                     conv3x3_halo_kernel<false, 32, false, 2, true, true, false, NP_, false, F_>, conv3x3_halo_kernel<true, 32, false, 2, true, true, false, NP_, false, F_>,  \
                     conv3x3_halo_kernel<false, 64, false, 1, true, true, false, NP_, false, F_>, conv3x3_halo_kernel<true, 64, false, 1, true, true, false, NP_, false, F_>,  \
                     conv3x3_halo_kernel<false, 64, false, 2, true, true, true, NP_, false, F_>, conv3x3_halo_kernel<false, 32, false, 2, true, true, true, NP_, false, F_>,   \
                     conv3x3_halo_kernel<false, 64, false, 1, true, true, true, NP_, false, F_>
    // 0-23 fp32 fragments (LDS slab / packed), 24-32 three bf16 pieces, 33-41 two bf16 pieces, 42-45 the 4 x 4 map form (bf16 pieces),
    // 46-54 two fp16 pieces, 55-56 the 4 x 4 map form with fp16 pieces
    static const kern_t fns[57] = {C3_ROW(false), C3_ROW(true), C3_XROW(3, false), C3_XROW(2, false),
                                   conv3x3_halo_kernel<false, 64, false, 1, true, true, false, 3, true>, conv3x3_halo_kernel<true, 64, false, 1, true, true, false, 3, true>,
                                   conv3x3_halo_kernel<false, 64, false, 1, true, true, false, 2, true>, conv3x3_halo_kernel<true, 64, false, 1, true, true, false, 2, true>,
                                   C3_XROW(2, true),
                                   conv3x3_halo_kernel<false, 64, false, 1, true, true, false, 2, true, true>, conv3x3_halo_kernel<true, 64, false, 1, true, true, false, 2, true, true>};
#undef C3_XROW
#undef C3_ROW
    constexpr int HALO2 = 8 * 200 * 16, HALO1 = 8 * 100 * 16;          // one halo stage: two / one sub-tile per workgroup
    constexpr int LDS1 = 2 * HALO1 + 2 * C3_B_BYTES;                   // one sub-tile per workgroup: 41,984 B
    const int XHALO2 = 4 * np * 200 * 16, XHALO1 = 4 * np * 100 * 16;  // split form: 12 / 8 plane images per stage
    if (bh_device_once(attr_devs)) {
        for (int i = 42; i < 57; ++i) {          // 4 x 4 map form: 144 halo slots, 12 / 8 plane images, two stages
            if (i >= 46 && i < 55) continue;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (i < 44 ? 12 : 8) * 144 * 16);
            if (e != hipSuccess) return (int)e;
        }
        for (int i0 = 0; i0 < 51; ++i0) {
            const int i = i0 < 42 ? i0 : 33 + (i0 - 42);       // the fp16 rows (46-54) have the geometry of the two-piece bf16 rows (33-41)
            const int fi = i0 < 42 ? i0 : 46 + (i0 - 42);
            const int j = i % 12;
            const int xi = i >= 24 ? (i - 24) % 9 : 0, xh2 = (i >= 33 ? 8 : 12) * 200 * 16, xh1 = xh2 / 2;       // split rows: 0-5 plain, 6-8 BNI
            const int full = i >= 24 ? 2 * ((xi == 4 || xi == 5 || xi == 8) ? xh1 : xh2) + (xi >= 6 ? 4096 : 0)
                                     : i < 12 ? (j < 8 ? C3_LDS_BYTES : LDS1) : 2 * (j < 8 ? HALO2 : HALO1);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[fi]), hipFuncAttributeMaxDynamicSharedMemorySize, full);
            if (e != hipSuccess) return (int)e;
        }
    }
    const int xrow = f16 ? 46 : 24 + (np == 2 ? 9 : 0);
    const kern_t fn = map4 ? fns[(f16 ? 55 : 42 + (np == 2 ? 2 : 0)) + (dgrad ? 1 : 0)] : bni ? fns[xrow + 6 + (subt == 1 ? 2 : bn_tile == 64 ? 0 : 1)] : x3 ? fns[xrow + (subt == 1 ? 4 : bn_tile == 64 ? 0 : 2) + (dgrad ? 1 : 0)]
                         : fns[(packed ? 12 : 0) + (subt == 1 ? 8 + (bf16 ? 2 : 0) + (dgrad ? 1 : 0)
                                                              : (bf16 ? 4 : 0) + (bn_tile == 64 ? 0 : 2) + (dgrad ? 1 : 0))];
    a.dbg_noload = g_c3_noload; a.dbg_ts = g_c3_stamp; a.desync = g_c3_desync; a.det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0;
    const int stage = map4 ? 4 * np * 144 * 16 : x3 ? (subt == 1 ? XHALO1 : XHALO2) : subt == 1 ? HALO1 : HALO2;
    const int lds = packed ? (Kc / 32 > 1 ? 2 : 1) * stage                                    // halo stages only
                           : (subt == 1 ? LDS1 : C3_LDS_BYTES) - (Kc / 32 > 1 ? 0 : stage);   // single chunk: one halo stage
    if (bni) a.bni_lds = lds;
    hipLaunchKernelGGL(fn, grid, dim3(256), lds + (bni ? bni->groups * Kc * 8 : 0), stream, a);
    BH_LAUNCH_CHECK();
    *taken = 1;
    return BH_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight packing for the PACKED kernels: [Co][9][Ci] -> MFMA-fragment order, for the forward (K = Ci, N = Co) and for the
// dgrad (K = Co, N = Ci, taps flipped) of every 3x3 layer of a network in ONE launch (grid.y = layer).
//   packed[(((chunk*9 + tap)*NW + n_tile)*4 + q)*64 + lane] (float4) = B[k = chunk*32 + (2q + lane/32)*4 + e][n = n_tile*32 + lane%32]
//   forward: B[k][n] = W[n][tap][k]            dgrad: B[k][n] = W[k][8 - tap][n]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_conv3x3_weights_kernel(const bh_pack3x3_job* __restrict__ jobs) {
    const bh_pack3x3_job j = jobs[blockIdx.y];
    const int Co = j.Co, Ci = j.Ci;
    const long long n4 = (long long)Co * 9 * Ci / 4;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(j.w);
    if (j.split) {
        const int np = j.split == 1 ? 3 : 2;              // pieces per weight (split 1: the exact three-way cut; 2: two rounded bf16 pieces; 3: two fp16 pieces)
        float wscale = 1.0f;
        if (j.split == 3) {                               // 2^k with max |w| 2^k in [2^14, 2^15): the sixteen partial maxima behind the pieces
            const unsigned* rec = reinterpret_cast<const unsigned*>(j.pf ? j.pf : j.pd) + (long long)Co * 9 * Ci;
            unsigned m = 0;
            for (int i = 0; i < 16; ++i) m = rec[i] > m ? rec[i] : m;
            wscale = __builtin_bit_cast(float, (unsigned)(127 + bh_f16_scale_exp(m)) << 23);
        }
        // [chunk][tap][n tile][piece][16-channel step s2][lane] x (8 bf16): lane (l31, kh2) holds k = chunk*32 + (2*s2 + kh2)*8 + e
        const long long n8 = (long long)Co * 9 * Ci / 8;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
            const int lane = (int)(i & 63), s2 = (int)((i >> 6) & 1);
            const int l31 = lane & 31, kh2 = lane >> 5;
            const long long r = i >> 7;
            for (int dir = 0; dir < 2; ++dir) {
                float* const dst = dir ? j.pd : j.pf;
                if (!dst) continue;
                const int NW = (dir ? Ci : Co) / 32;
                const int nt = (int)(r % NW); const long long r2 = r / NW;
                const int tap = (int)(r2 % 9), c = (int)(r2 / 9);
                const int n = nt * 32 + l31, k0 = c * 32 + (2 * s2 + kh2) * 8;
                float4 u, v;
                if (!dir) {
                    const float4* src = reinterpret_cast<const float4*>(j.w + ((long long)(n * 9 + tap) * Ci + k0));
                    u = src[0]; v = src[1];
                } else {
                    const float* src = j.w + ((long long)k0 * 9 + (8 - tap)) * Ci + n;
                    const long long ks = 9ll * Ci;
                    u = make_float4(src[0], src[ks], src[2 * ks], src[3 * ks]);
                    v = make_float4(src[4 * ks], src[5 * ks], src[6 * ks], src[7 * ks]);
                }
                uint4 p0, p1, p2;
                uint4* const o = reinterpret_cast<uint4*>(dst) + (r * 2 * np + s2) * 64 + lane;
                if (np == 3) { bh_split8(u, v, p0, p1, p2); o[4 * 64] = p2; }
                else if (j.split == 3) bh_split8_f16(u, v, wscale, p0, p1);
                else bh_split8_2(u, v, p0, p1);
                o[0] = p0; o[2 * 64] = p1;
            }
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63), q = (int)((i >> 6) & 3);
        const int l31 = lane & 31, kh2 = lane >> 5;
        long long r = i >> 8;
        if (j.pf) {
            const int NW = Co / 32;
            const int nt = (int)(r % NW); long long r2 = r / NW;
            const int tap = (int)(r2 % 9), c = (int)(r2 / 9);
            const int n = nt * 32 + l31, k0 = c * 32 + (2 * q + kh2) * 4;
            reinterpret_cast<float4*>(j.pf)[i] = w4[((long long)(n * 9 + tap) * Ci + k0) >> 2];
        }
        if (j.pd) {
            const int NW = Ci / 32;
            const int nt = (int)(r % NW); long long r2 = r / NW;
            const int tap = (int)(r2 % 9), c = (int)(r2 / 9);
            const int n = nt * 32 + l31, k0 = c * 32 + (2 * q + kh2) * 4;
            const float* src = j.w + ((long long)k0 * 9 + (8 - tap)) * Ci + n;
            const long long ks = 9ll * Ci;
            reinterpret_cast<float4*>(j.pd)[i] = make_float4(src[0], src[ks], src[2 * ks], src[3 * ks]);
        }
    }
}

// split = 3 (fp16 pieces): the layer's max |w| as sixteen partial maxima (workgroup x -> word x behind the pieces of BOTH buffers; plain
// stores, nothing to zero), read by the pack kernel and by every consumer of the buffer
__global__ void __launch_bounds__(256) pack_weights_amax_kernel(const bh_pack3x3_job* __restrict__ jobs) {
    const bh_pack3x3_job j = jobs[blockIdx.y];
    if (j.split != 3) return;
    __shared__ float sm[4];
    const long long n4 = (long long)j.Co * 9 * j.Ci / 4;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(j.w);
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 16ll * 256) {
        const float4 v = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (j.pf) j.pf[n4 * 4 + blockIdx.x] = m;
        if (j.pd) j.pd[n4 * 4 + blockIdx.x] = m;
    }
}

// tables with split = 3 jobs: the maxima pass, then the pack (two launches for the whole network)
extern "C" int bh_conv3x3_pack_f16(const bh_pack3x3_job* jobs_dev, int njobs, void* stream) {
    if (!jobs_dev || njobs < 0) return BH_E_BADARG;
    if (njobs == 0) return BH_OK;
    hipLaunchKernelGGL(pack_weights_amax_kernel, dim3(16, njobs), dim3(256), 0, bh_stream(stream), jobs_dev);
    BH_LAUNCH_CHECK();
    return bh_conv3x3_pack(jobs_dev, njobs, stream);
}

extern "C" int bh_conv3x3_pack(const bh_pack3x3_job* jobs_dev, int njobs, void* stream) {
    if (!jobs_dev || njobs < 0) return BH_E_BADARG;
    if (njobs == 0) return BH_OK;
    hipLaunchKernelGGL(pack_conv3x3_weights_kernel, dim3(64, njobs), dim3(256), 0, bh_stream(stream), jobs_dev);
    BH_LAUNCH_CHECK();
    return BH_OK;
}
