// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions in "f32x3" arithmetic (bh_conv_desc.precision = 2), halo-tiled:
//   gw[co][tap][ci] += sum over pixels  gy[pixel][co] * x[pixel + tap shift][ci]
// The contraction runs over PIXELS, so on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: a lane supplies 8 consecutive K
// values of one row) both operands are needed "pixel-major per channel", while NHWC memory is channel-major per pixel.
// gfx950's transposing LDS read does that turn for free: ds_read_b64_tr_b16 hands lane i of a 16-lane group the i-th
// 16-bit COLUMN of the 4 x 16 block whose four rows the group's lanes point at (lane 4j + g -> row j, columns 4g..4g+3).
// So the LDS image stays [pixel][64 channels] (bf16, one image per piece of the three-way exact split x = hi + mid + lo,
// common.h bh_split8), a tap is a whole-pixel address shift (an immediate offset), and the four rows of a read are a 2x2
// pixel block: with rows padded by 64 B the four 64-byte row segments a half-wave reads fall on the four quarters of
// the 64 LDS banks - conflict-free for every tap.
//
// A workgroup (4 waves) owns one (64 output channels x 64 input channels) block of gw for ALL NINE taps - 36 accumulators
// of 32x32, nine per wave (wave = cout half x cin half) - and walks 8x8-pixel tiles: the gy tile (64 pixels) and the x halo
// (10x10 pixels) are loaded ONCE for the nine taps (the fp32-MFMA kernel wgrad_s1_kernel re-reads both per tap: 320 MB
// per launch against 50 MB algorithmic), cut into the three pieces in registers and written to LDS; the next tile's loads
// are in flight during the 216 MFMAs per wave of the current one.
// Per 16-pixel step and wave: 60 transposing reads (512 B each) for 54 MFMAs.
// SETS = 2 (default): ONE workgroup of eight waves per CU, two wave sets with an LDS image each.  The sets alternate between
// phases separated by a workgroup barrier: while one set runs the MFMAs of its tile the other cuts and writes its next
// tile, so the matrix pipe always has a computing wave on every SIMD and the split work of one set hides under the
// MFMAs of the other.  Afterwards the sets swap halves of their accumulators through LDS (set 0 ends up with taps 0-4 of
// both, set 1 with taps 5-8): a workgroup contributes ONE 64 x 9 x 64 block per launch instead of two - the split-K
// reduction is what the launch spends its tail on (147 KB per workgroup; 256 workgroups: 37.7 MB).
// The block is added to gw with fp32 atomics (same split-K scheme and float non-associativity as wgrad_s1_kernel), or -
// when the caller passes a workspace (bh_conv_wgrad_det) - stored as a partial tile that wgrad_x3_reduce_kernel sums in
// split order: bitwise reproducible, and cheaper than the atomics.
// SETS = 1: four waves, two workgroups per CU covering each other's staging (kept for comparison: twice the reduction).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) i16x4* lds_i16x4_ptr;

struct WX3Args {
    const float* X;
    const float* GY;
    float* Out;
    int N, H, W, Ci, Co;
    int tiles_x, tiles_per_img, ntiles;
    unsigned m_tx, m_tpi;          // floor(2^32 / tiles_x) + 1, floor(2^32 / tiles_per_img) + 1
    int cbi;                       // 64-channel input blocks (Ci / 64); blockIdx.x = (cout block * cbi + cin block) * nsplit + split
    int nsplit;
    unsigned x_bytes, gy_bytes;
    int noflush;
    float* partials;               // != NULL: partial[workgroup][tap][wave of the set][r][lane] instead of atomics
};

constexpr int WX_PIX = 128;                   // bytes of a pixel record: 64 bf16 channels
constexpr int WX_GROW = 8 * WX_PIX + 64;      // gy tile row (8 pixels + 64 B skew)
constexpr int WX_XROW = 10 * WX_PIX + 64;     // halo row (10 pixels + skew)
constexpr int WX_GP = 8 * WX_GROW;            // one piece image of the gy tile:  8,704 B
constexpr int WX_XP = 10 * WX_XROW;           // one piece image of the halo:    13,440 B
constexpr int WX_XBASE = 3 * WX_GP;
constexpr int WX_LDS = 3 * (WX_GP + WX_XP);   // 66,432 B per wave set
constexpr int WX_XCHG = 9 * 16384;            // accumulator exchange of the two sets: 5 + 4 taps x 16 KB

// SETS = 0: the pipelined form - ONE workgroup of four waves per CU, one wave per SIMD with the full 512-entry register file,
// TWO LDS images.  While the MFMAs of tile k read image k & 1, the same waves cut tile k + 1 out of the registers its loads
// arrived in and write it to the other image (two staging slots per 16-pixel step, in the MFMA stream), then request tile
// k + 2: no staging phase, one barrier per tile, and room to fetch the fragments of the next tap ahead of the MFMAs.
template <int SETS>
__global__ void __launch_bounds__(SETS == 0 ? 256 : 256 * SETS, SETS == 0 ? 1 : 2) wgrad_x3_kernel(WX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int set = SETS == 2 ? (int)(threadIdx.x >> 8) : 0;      // wave set: own LDS image, own tiles (every second one)
    char* const smem = smem_all + set * WX_LDS;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;                      // cout half, cin half of the 64 x 64 block
    const int split = blockIdx.x % a.nsplit, pair = blockIdx.x / a.nsplit;
    const int co0 = (pair / a.cbi) * 64, ci0 = (pair % a.cbi) * 64;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.GY), 0, a.gy_bytes, 0x00020000);

    // ---- staging slots of this thread (tile independent parts): slot = (pixel, 8-channel group) -> 2 dwordx4, 3 ds_write_b128 ----
    // gy: 64 pixels x 8 groups = 512 slots (2 per thread); halo: 100 x 8 = 800 slots (3 per thread + 32 threads a 4th)
    const int cg = tid & 7;
    int g_pix[2], g_lds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = (j * 256 + tid) >> 3;                       // 0..63
        g_pix[j] = (p >> 3) * a.W + (p & 7);
        g_lds[j] = (p >> 3) * WX_GROW + (p & 7) * WX_PIX + cg * 16;
    }
    int h_y[4], h_x[4], h_lds[4], h_cg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (slots 800..1023 of the fourth round repeat slot 799 - same address, same data - so that the staging code has no
        //  branch and the whole tile step stays one scheduling region)
        const int q = min(j * 256 + tid, 799);
        const int hp = q >> 3, cgq = q & 7;                       // 0..99
        const int hy = (hp * 205) >> 11, hx = hp - hy * 10;
        h_y[j] = hy - 1; h_x[j] = hx - 1;
        h_lds[j] = WX_XBASE + hy * WX_XROW + hx * WX_PIX + cgq * 16;
        h_cg[j] = cgq * 8;
    }

    float4 rg[2][2], rx[4][2];
    // loads of tile t: part 0 = the gy slots, part 1 / 2 = halo slots 0-1 / 2-3 (the pipelined form re-requests each register
    // group as soon as its slots have been written out, so every load has most of a tile of MFMAs to arrive)
    auto issue_part = [&](int t, auto PART) {
        constexpr int part = decltype(PART)::value;               // -1: all
        // (multiply-high division with host-made reciprocals: exact for t * divisor < 2^32; no branch in the tile step)
        const int img = a.tiles_per_img == 1 ? t : (int)__umulhi((unsigned)t, a.m_tpi), r = t - img * a.tiles_per_img;
        const int ty = a.tiles_x == 1 ? r : (int)__umulhi((unsigned)r, a.m_tx), tx = r - ty * a.tiles_x;
        const int org = (img * a.H + ty * 8) * a.W + tx * 8;     // pixel index of the tile's (0, 0)
        if constexpr (part <= 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned off = ((unsigned)(org + g_pix[j]) * (unsigned)a.Co + (unsigned)(co0 + cg * 8)) * 4u;
                rg[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
                rg[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off + 16u, 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (part >= 0 && (part == 0 || j / 2 != part - 1)) continue;
            const int y = ty * 8 + h_y[j], x = tx * 8 + h_x[j];
            const unsigned in = ((unsigned)(org + h_y[j] * a.W + h_x[j]) * (unsigned)a.Ci + (unsigned)(ci0 + h_cg[j])) * 4u;
            const unsigned off = ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) ? in : OOB;     // outside the image: zeros
            rx[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
            rx[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off + 16u, 0, 0));
        }
    };
    auto issue = [&](int t) { issue_part(t, std::integral_constant<int, -1>{}); };
    // slot 0, 1: gy; 2..5: halo
    auto stage_slot = [&](auto SLOT, char* img) {
        constexpr int sl = decltype(SLOT)::value;
        uint4 p0, p1, p2;
        if constexpr (sl < 2) {
            bh_split8(rg[sl][0], rg[sl][1], p0, p1, p2);
            *reinterpret_cast<uint4*>(img + g_lds[sl]) = p0;
            *reinterpret_cast<uint4*>(img + g_lds[sl] + WX_GP) = p1;
            *reinterpret_cast<uint4*>(img + g_lds[sl] + 2 * WX_GP) = p2;
        } else {
            constexpr int j = sl - 2;
            bh_split8(rx[j][0], rx[j][1], p0, p1, p2);
            *reinterpret_cast<uint4*>(img + h_lds[j]) = p0;
            *reinterpret_cast<uint4*>(img + h_lds[j] + WX_XP) = p1;
            *reinterpret_cast<uint4*>(img + h_lds[j] + 2 * WX_XP) = p2;
        }
    };
#define WX_SLOT(n, img) stage_slot(std::integral_constant<int, n>{}, img)
    auto stage = [&]() { WX_SLOT(0, smem); WX_SLOT(1, smem); WX_SLOT(2, smem); WX_SLOT(3, smem); WX_SLOT(4, smem); WX_SLOT(5, smem); };

    // ---- fragment addresses.  16-lane group (lane >> 4) & 1 -> channels 16..31 of the wave's 32; lane >> 5 = K half.
    // Within a group lane 4j + g points at row j of the 2x2 pixel block (j >> 1 down, j & 1 right), channels 4g..4g+3; the
    // K = 16 step ks covers tile rows 2ks, 2ks + 1: K half kh2 and read r take the block at columns 4 kh2 + 2r.
    const int L = lane & 15, jj = L >> 2, gq = L & 3, gsel = (lane >> 4) & 1, kh2 = lane >> 5;
    const int chan_b = (16 * gsel + 4 * gq) * 2;
    const int laneA = (jj >> 1) * WX_GROW + (4 * kh2 + (jj & 1)) * WX_PIX + chan_b + wm * 64;
    const int laneB = WX_XBASE + (jj >> 1) * WX_XROW + (4 * kh2 + (jj & 1)) * WX_PIX + chan_b + wn * 64;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#define WX_TR(base, off) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_ptr)(smem + (base) + (off)))
#define WX_OPER(lo_, hi_) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7))
#define WX_COMPUTE_KS(ks, la, lb) do { \
            bf16x8 af[3]; \
_Pragma("unroll") \
            for (int pc = 0; pc < 3; ++pc) { \
                const i16x4 v0 = WX_TR((la), pc * WX_GP + 2 * (ks) * WX_GROW); \
                const i16x4 v1 = WX_TR((la), pc * WX_GP + 2 * (ks) * WX_GROW + 2 * WX_PIX); \
                af[pc] = WX_OPER(v0, v1); \
            } \
_Pragma("unroll") \
            for (int tap = 0; tap < 9; ++tap) { \
                const int ty = tap / 3, tx = tap - ty * 3; \
                bf16x8 bf[3]; \
_Pragma("unroll") \
                for (int pc = 0; pc < 3; ++pc) { \
                    const i16x4 v0 = WX_TR((lb), pc * WX_XP + (2 * (ks) + ty) * WX_XROW + tx * WX_PIX); \
                    const i16x4 v1 = WX_TR((lb), pc * WX_XP + (2 * (ks) + ty) * WX_XROW + (tx + 2) * WX_PIX); \
                    bf[pc] = WX_OPER(v0, v1); \
                } \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[tap], 0, 0, 0); \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[tap], 0, 0, 0); \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[tap], 0, 0, 0); \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[tap], 0, 0, 0); \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[tap], 0, 0, 0); \
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[tap], 0, 0, 0); \
            } \
} while (0)

    // tiles of this workgroup: split, split + nsplit, ...; set s takes every SETS-th one starting at the s-th
    const int nt = a.ntiles > split ? (a.ntiles - split + a.nsplit - 1) / a.nsplit : 0;
    if constexpr (SETS == 0) {
        if (nt > 0) { issue(split); stage(); }
        __syncthreads();
        if (nt > 1) issue(split + a.nsplit);
        // One scheduling region per tile (no branches inside): the fragments of step (ks, tap) + 1 are requested before the six
        // MFMAs of step (ks, tap); the cut-and-write of the next tile (two slots per ks) and, in the last ks, the loads of the
        // tile after it are spread between the MFMAs by the sched_group_barrier pattern below.
#define WX_LOAD_A(dst, ks) _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                          \
        const i16x4 v0_ = WX_TR(la, pc * WX_GP + 2 * (ks) * WX_GROW);                                                   \
        const i16x4 v1_ = WX_TR(la, pc * WX_GP + 2 * (ks) * WX_GROW + 2 * WX_PIX);                                      \
        dst[pc] = WX_OPER(v0_, v1_); }
#define WX_LOAD_B(dst, ks, tap) _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                     \
        const i16x4 v0_ = WX_TR(lb, pc * WX_XP + (2 * (ks) + (tap) / 3) * WX_XROW + ((tap) % 3) * WX_PIX);              \
        const i16x4 v1_ = WX_TR(lb, pc * WX_XP + (2 * (ks) + (tap) / 3) * WX_XROW + ((tap) % 3 + 2) * WX_PIX);          \
        dst[pc] = WX_OPER(v0_, v1_); }
        for (int k = 0; k < nt; ++k) {
            const int cb = (k & 1) * WX_LDS;
            char* const nimg = smem_all + (WX_LDS - cb);                   // the other image
            const int la = laneA + cb, lb = laneB + cb;
            const int tnext = split + min(k + 2, nt - 1) * a.nsplit;       // (past the end: re-request the last tile, never used)
            bf16x8 af[2][3], bf[2][2][3];                                  // A by ks parity; B by step-pair parity and step parity
            WX_LOAD_A(af[0], 0);
            WX_LOAD_B(bf[0][0], 0, 0);
            WX_LOAD_B(bf[0][1], 0, 1);
            // steps (ks, tap) are taken two at a time so that consecutive MFMAs alternate between two accumulators (six
            // dependent MFMAs on one accumulator in a row leave the pipe waiting for its own result)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int pi = hp * 9 + q, s0 = 2 * pi, s1 = s0 + 1, pb = pi & 1;
#pragma unroll
                for (int f = s0 + 2; f <= s1 + 2; ++f)
                    if (f < 36) {
                        if (f % 9 == 0) { WX_LOAD_A(af[(f / 9) & 1], f / 9); }
                        WX_LOAD_B(bf[pb ^ 1][f & 1], f / 9, f % 9);
                    }
                if (pi == 3) { WX_SLOT(0, nimg); WX_SLOT(1, nimg); issue_part(tnext, std::integral_constant<int, 0>{}); }
                if (pi == 8) { WX_SLOT(2, nimg); WX_SLOT(3, nimg); issue_part(tnext, std::integral_constant<int, 1>{}); }
                if (pi == 13) { WX_SLOT(4, nimg); WX_SLOT(5, nimg); issue_part(tnext, std::integral_constant<int, 2>{}); }
                const int k0 = (s0 / 9) & 1, k1 = (s1 / 9) & 1, t0 = s0 % 9, t1 = s1 % 9;
#define WX_MM(PA, PB)                                                                                                   \
    acc[t0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k0][PA], bf[pb][0][PB], acc[t0], 0, 0, 0);                     \
    acc[t1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k1][PA], bf[pb][1][PB], acc[t1], 0, 0, 0)
                WX_MM(2, 0); WX_MM(0, 2); WX_MM(1, 1); WX_MM(1, 0); WX_MM(0, 1); WX_MM(0, 0);
#undef WX_MM
            }
            // desired issue order: one MFMA, then at most one fragment read, three VALU, and now and then a store / load
#pragma unroll
            for (int g = 0; g < 36; ++g)
#pragma unroll
                for (int h = 0; h < 6; ++h) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // VALU
                    if (h == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
                    if (h == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM read
                }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (LDS only: the requested tile stays in flight)
        }
#undef WX_LOAD_A
#undef WX_LOAD_B
    } else {
    if (set < nt) issue(split + set * a.nsplit);
    // phase p: the set with p % SETS == set cuts and writes tile p (its loads were issued a phase or more ago); the other set
    // (SETS = 2) runs the MFMAs of tile p - 1 and, first thing, requests tile p + 1
    for (int p = 0; p <= nt; ++p) {
        if (SETS == 1) {
            if (p == nt) break;
            stage();
            __syncthreads();
            if (p + 1 < nt) issue(split + (p + 1) * a.nsplit);
        }
        if (SETS == 1 || (((p & 1) != set) && p >= 1)) {
            if (SETS == 2 && p + 1 < nt) issue(split + (p + 1) * a.nsplit);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) WX_COMPUTE_KS(ks, laneA, laneB);
        } else if (SETS == 2 && (p & 1) == set && p < nt) {
            stage();
        }
        __syncthreads();
    }
    }
#undef WX_SLOT
#undef WX_COMPUTE_KS
#undef WX_TR
#undef WX_OPER
    const int l31 = lane & 31;
    int tap_lo = 0, tap_hi = 9;
    if constexpr (SETS == 2) {
        // swap: set 1 hands over taps 0-4, set 0 taps 5-8 ([tap][wave][r][lane] floats: lane-contiguous, conflict-free)
        float* const xc = reinterpret_cast<float*>(smem_all);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
            if ((tp < 5) == (set == 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xc[((tp * 4 + wave) * 16 + r) * 64 + lane] = acc[tp][r];
            }
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
            if ((tp < 5) == (set == 0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tp][r] += xc[((tp * 4 + wave) * 16 + r) * 64 + lane];
            }
        tap_lo = set == 0 ? 0 : 5; tap_hi = set == 0 ? 5 : 9;
    }
    if (a.noflush == 1) return;
    // C/D layout: column = lane & 31 (cin), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (cout).  All workgroups finish together and
    // every one adds to the same 64 x 9 x 64 block: each starts at another tap so that they do not queue on the same addresses.
    const int ntp = tap_hi - tap_lo;
    const int rot = (int)(blockIdx.x % (unsigned)ntp);
    for (int i = 0; i < ntp; ++i) {
        int tap = i + rot;
        if (tap >= ntp) tap -= ntp;
        tap += tap_lo;
        float* const o = a.Out + ((long long)(co0 + wm * 32 + 4 * kh2) * 9 + tap) * a.Ci + ci0 + wn * 32 + l31;
        float* const pp = a.partials ? a.partials + (((size_t)blockIdx.x * 9 + tap) * 4 + wave) * 1024 + lane : nullptr;
#define WX_FLUSH(T)                                                                                                     \
    case T:                                                                                                             \
        if (pp) {                                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) pp[r * 64] = acc[T][r];                                      \
        } else {                                                                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) atomicAdd(o + (long long)((r & 3) + 8 * (r >> 2)) * 9 * a.Ci, acc[T][r]); \
        }                                                                                                               \
        break;
        switch (tap) { WX_FLUSH(0) WX_FLUSH(1) WX_FLUSH(2) WX_FLUSH(3) WX_FLUSH(4) WX_FLUSH(5) WX_FLUSH(6) WX_FLUSH(7) WX_FLUSH(8) }
#undef WX_FLUSH
    }
}

// Sums the partial blocks of wgrad_x3_kernel in split order and adds the total to gw.  One workgroup = 64 consecutive
// elements of a block's partial order x 4 split groups (each thread adds every 4th... a contiguous quarter of the splits,
// 16 loads in flight), the four sub-sums are combined in fixed order through LDS: bitwise reproducible.
__global__ void __launch_bounds__(256) wgrad_x3_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int nsplit,
                                                              int cbi, int Ci) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;                           // [tap][wave][r][lane] index inside the block, < 36864
    const int pair = blockIdx.y;
    const int per = (nsplit + 3) / 4, s0 = grp * per, s1 = min(nsplit, s0 + per);
    const float* p = partials + ((size_t)pair * nsplit + s0) * 36864 + idx;
    float sum = 0.f;
    int sidx = s0;
    for (; sidx + 8 <= s1; sidx += 8, p += 8 * 36864) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)u * 36864];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; sidx < s1; ++sidx, p += 36864) sum += *p;
    red[grp][e] = sum;
    __syncthreads();
    if (grp == 0) {
        const float tot = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const int lane = idx & 63, r = (idx >> 6) & 15, wave = (idx >> 10) & 3, tap = idx >> 12;
        const int wm = wave & 1, wn = wave >> 1, kh2 = lane >> 5, l31 = lane & 31;
        const int co = (pair / cbi) * 64 + wm * 32 + 4 * kh2 + (r & 3) + 8 * (r >> 2), ci = (pair % cbi) * 64 + wn * 32 + l31;
        out[((long long)co * 9 + tap) * Ci + ci] += tot;
    }
}

BH_KNOB(g_wx3_target, 256); BH_KNOB(g_wx3_noflush, 0); BH_KNOB(g_wx3_sets, 0);
#ifdef BH_TUNING
void bh_wgrad_x3_tune(int what, int v) { if (what == 0) g_wx3_target = v; else if (what == 1) g_wx3_noflush = v; else g_wx3_sets = v; }
#endif

// *taken = 1 when the shape is eligible (3x3 / stride 1 / pad 1, NHWC, H and W multiples of 8, channels multiples of 64).
// ws != NULL: deterministic reduction through ws (ws_need != NULL: dry run that only reports the bytes needed)
int bh_wgrad_x3_try(const float* x, const float* gy, float* gw, const bh_conv_desc* d, hipStream_t stream, int* taken, float* ws,
                    long long ws_bytes, long long* ws_need) {
    *taken = 0;
    if (d->transposed || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->in_nchw || d->out_nchw) return BH_OK;
    if (d->Hi % 8 || d->Wi % 8 || d->Ho != d->Hi || d->Wo != d->Wi || d->Ci % 64 || d->Co % 64) return BH_OK;
    const long long xe = (long long)d->N * d->Hi * d->Wi * d->Ci, ge = (long long)d->N * d->Hi * d->Wi * d->Co;
    if (xe >= (1ll << 29) || ge >= (1ll << 29)) return BH_OK;
    WX3Args a = {};
    a.X = x; a.GY = gy; a.Out = gw;
    a.N = d->N; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.tiles_x = d->Wi / 8; a.tiles_per_img = (d->Hi / 8) * a.tiles_x; a.ntiles = d->N * a.tiles_per_img;
    if ((long long)a.ntiles * a.tiles_per_img >= (1ll << 32)) return BH_OK;
    a.m_tx = (unsigned)((1ull << 32) / (unsigned)a.tiles_x + 1ull);
    a.m_tpi = (unsigned)((1ull << 32) / (unsigned)a.tiles_per_img + 1ull);
    a.cbi = d->Ci / 64;
    const int pairs = a.cbi * (d->Co / 64);
    const int sets = g_wx3_sets;                               // 0: pipelined, one 4-wave workgroup per CU (default)
    int ns = g_wx3_target * (sets == 1 ? 2 : 1) / pairs;       // sets 1: two 4-wave workgroups per CU; 2: one of 8 waves
    const int per = sets == 2 ? 2 : 1;
    if (ns > (a.ntiles + per - 1) / per) ns = (a.ntiles + per - 1) / per;
    if (ns >= 8) ns = ns / 8 * 8;               // the workgroups of one split (same tiles, other channel blocks) land on one XCD
    if (ns < 1) ns = 1;
    a.nsplit = ns;
    a.x_bytes = (unsigned)(xe * 4); a.gy_bytes = (unsigned)(ge * 4);
    a.noflush = g_wx3_noflush;
    const long long need = (long long)pairs * ns * 36864 * 4;
    if (ws) {
        if (ws_need) *ws_need = need;
        else if (ws_bytes < need) return BH_E_BADARG;
        a.partials = ws;
    }
    if (bh_query(ws ? "wgrad_x3_kernel<%d>+wgrad_x3_reduce_kernel" : "wgrad_x3_kernel<%d>", sets)) { *taken = 1; return BH_OK; }
    static unsigned long long attr_devs = 0;
    constexpr int LDS2 = 2 * WX_LDS > WX_XCHG ? 2 * WX_LDS : WX_XCHG;
    if (bh_device_once(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, WX_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WX_LDS);
        if (e != hipSuccess) return (int)e;
    }
    if (sets == 0) hipLaunchKernelGGL(wgrad_x3_kernel<0>, dim3(pairs * ns), dim3(256), 2 * WX_LDS, stream, a);
    else if (sets == 1) hipLaunchKernelGGL(wgrad_x3_kernel<1>, dim3(pairs * ns), dim3(256), WX_LDS, stream, a);
    else hipLaunchKernelGGL(wgrad_x3_kernel<2>, dim3(pairs * ns), dim3(512), LDS2, stream, a);
    BH_LAUNCH_CHECK();
    if (ws) {
        hipLaunchKernelGGL(wgrad_x3_reduce_kernel, dim3(36864 / 64, pairs), dim3(256), 0, stream, ws, gw, ns, a.cbi, d->Ci);
        BH_LAUNCH_CHECK();
    }
    *taken = 1;
    return BH_OK;
}
