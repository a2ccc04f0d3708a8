// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions in "f32x3" arithmetic (bh_conv_desc.precision = 2), halo-tiled:
//   gw[co][tap][ci] += sum over pixels  gy[pixel][co] * x[pixel + tap shift][ci]
// The contraction runs over PIXELS, so on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: a lane supplies 8 consecutive K
// values of one row) both operands are needed "pixel-major per channel", while NHWC memory is channel-major per pixel.
// gfx950's transposing LDS read does that turn for free: ds_read_b64_tr_b16 hands lane i of a 16-lane group the i-th
// 16-bit COLUMN of the 4 x 16 block whose four rows the group's lanes point at (lane 4j + g -> row j, columns 4g..4g+3;
// tools/tr16_probe.hip).  So the LDS image stays [pixel][CB channels] (bf16, one image per piece of the three-way exact split
// x = hi + mid + lo, common.h bh_split8), a tap is a whole-pixel address shift (an immediate offset), and the four rows of
// a read are a 2x2 pixel block: with the image rows skewed the four 64-byte row segments a half-wave reads fall on the
// four quarters of the 64 LDS banks - conflict-free for every tap (SQ_LDS_BANK_CONFLICT = 0).
//
// A workgroup (4 waves, ONE per CU: one wave per SIMD with the whole 512-entry register file, 144 accumulator registers in
// AGPRs) owns one (CB output channels x CB input channels) block of gw for ALL NINE taps and walks 8x8-pixel tiles: the gy
// tile (64 pixels) and the x halo (10x10 pixels) are loaded ONCE for the nine taps (the fp32-MFMA kernel wgrad_s1_kernel
// re-reads both per tap: 320 MB per launch against 50 MB algorithmic), cut into the three pieces in registers and written
// to LDS.  Two LDS images: while the MFMAs of tile k read image k & 1, the same waves cut tile k + 1 out of the registers
// its loads arrived in and write it to the other image - inside the MFMA stream (multiply-high tile decoding, fragments
// double-buffered in source, two accumulators alternating) - then request tile k + 2; one barrier per tile.  (Spreading the
// cut-and-write VALU work evenly between the MFMAs - branch-free tile step, sched_group_barrier pattern of one MFMA : one
// LDS read : three VALU - measured 2-12 % SLOWER than leaving it in three bursts: an instruction between two MFMAs costs
// more than its issue slot.)
//   CB = 64: 36 accumulators of 32x32, nine per wave (wave = cout half x cin half), each wave runs the four 16-pixel steps
//            of a tile: 216 MFMAs and ~170 transposing reads per wave and tile.
//   CB = 32: 9 accumulators; the four waves split the PIXELS of a tile (wave w takes the w-th 16-pixel step) and each
//            keeps all nine - their partial blocks are four more terms of the split-K sum.
// The partial blocks (<= 256 workgroups x 147 KB) are added to gw with fp32 atomics, or - when the caller passes a workspace
// (bh_conv_wgrad_det; what the models do) - stored and summed in split order by wgrad_x3_reduce_kernel: bitwise
// reproducible, and cheaper than the atomics (every workgroup adds to the SAME block: same-address serialisation).
// Two other schedules were built and measured (two 4-wave workgroups per CU covering each other's staging; one 8-wave
// workgroup whose wave sets alternate between MFMA and staging phases): compute phase 55 / 66 us against 45 - DESIGN.md 3.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i16x4* lds_i16x4_ptr;

struct WX3Args {
    const float* X;
    const float* GY;
    float* Out;
    int N, H, W, Ci, Co;
    int tiles_x, tiles_per_img, ntiles;
    unsigned m_tx, m_tpi;          // floor(2^32 / tiles_x) + 1, floor(2^32 / tiles_per_img) + 1
    int cbi;                       // CB-channel input blocks (Ci / CB); blockIdx.x = (cout block * cbi + cin block) * nsplit + split
    int nsplit;
    unsigned x_bytes, gy_bytes;
    int noflush;
    float* partials;               // != NULL: partial[workgroup][tap][wave][r][lane] instead of atomics
    // BNI: X is the INPUT of a training-mode BatchNorm(+ReLU) whose output the convolution consumed: the halo staging applies
    // max(x * scale + shift, lo) per channel (zero padding stays zero).  bni: table[groups][Ci] x (scale, shift) (bh_bn_fwd_coeffs)
    const float* bni;
    int bni_relu, bni_ipg, bni_groups;
    // F16 (two fp16 pieces per operand, common.h F16X2): magnitude records of X (of the BatchNorm OUTPUT with BNI) and of GY
    const unsigned* amax_x;
    const unsigned* amax_gy;
    // BNA (round 6): GY is d, the gradient of a training-mode BatchNorm's OUTPUT (not masked), and the operand the kernel contracts is
    // that BatchNorm's adjoint, applied while the tile is staged:  g = sc (mask(d) - k1 - xhat(z) k2)  with z the BatchNorm's input
    // (= this convolution's output), the mask y > 0 of the ReLU behind it (y recomputed as z sc + sh, or read from bna_y when a residual was
    // added in front of the ReLU), k1 = sum mask(d) / rows, k2 = sum mask(d) xhat / rows from bna_sums (accumulated by the dgrad that made d:
    // bh_conv_dgrad_bnreduce) and (mean, 1 / std) from the forward sums bna_stats.  amax_gy is then the magnitude record of d; the scale of
    // the operand's fp16 pieces comes from the bound max_c |sc| (max |d| + |k1| + sqrt(rows) |k2|) over the workgroup's 64 channels.
    // Batch (round 6): up to four layers of ONE geometry in a launch - workgroup w belongs to layer w / wg_per_layer.  Each layer's pixel
    // range is then split over fewer workgroups: the launch writes (and wgrad_x3_reduce_kernel reads) one set of partial blocks for all of
    // them, where single launches write one set EACH, and three of four kernel / reduce launch pairs disappear.  Layer 0: the fields above.
    int nlayers, wg_per_layer;
    const float *X1, *X2, *X3, *GY1, *GY2, *GY3;
    float *Out1, *Out2, *Out3;
    const unsigned *amax_x1, *amax_x2, *amax_x3, *amax_gy1, *amax_gy2, *amax_gy3;
    const float *bni1, *bni2, *bni3;
    const float* bna_z;
    const float* bna_y;
    const double* bna_stats;
    const double* bna_sums;
    const float* bna_gamma;
    const float* bna_beta;
    float bna_eps;
    int bna_relu, bna_rows, bna_groups, bna_ipg;
};

// MAP4 (round 5, CB = 64): 4 x 4 feature maps (layer4 of the ResNet-34 regressor) - a tile is FOUR IMAGES laid out 2 x 2 as an 8 x 8 pixel block,
// each with its own zero border: the halo image is 12 x 12 (sub-map (a, b) at rows 6a .. 6a + 5, columns 6b .. 6b + 5)
template <int CB, int NP = 3, bool MAP4 = false>
struct WXGeom {
    static constexpr int XW = MAP4 ? 12 : 10;                  // halo rows / columns
    static constexpr int PIX = CB * 2;                        // bytes of a pixel record: CB bf16 channels
    // row skew: the four 64-byte segments of a transposing read (2x2 pixels) must start 16 banks apart
    static constexpr int GROW = 8 * PIX + (CB == 64 ? 64 : 128);      // gy tile row:  1088 / 640 B  (= 16 / 32 banks mod 64)
    static constexpr int XROW = XW * PIX + (CB == 64 ? 64 : 0);       // halo row:     1344 / 640 B (MAP4: 1600)
    static constexpr int GP = 8 * GROW;                        // one piece image of the gy tile
    static constexpr int XP = XW * XROW;                       // one piece image of the halo
    static constexpr int XBASE = NP * GP;
    static constexpr int LDS = NP * (GP + XP);                 // one image of a tile: 66,432 / 34,560 B (three pieces)
    static constexpr int NG = CB / 8;                          // 8-channel groups per pixel
    static constexpr int GS = 64 * NG / 256;                   // gy staging slots per thread (2 / 1)
    static constexpr int HS = (XW * XW * NG + 255) / 256;      // halo staging slots per thread (4 / 2; MAP4: 5)
};

// NP: bf16 pieces per operand - 3: the exact cut, six products (precision 2); 2: two rounded pieces, three products ("f32x2",
// precision 3, common.h bh_split8_2)
// F16 (NP = 2; round 4): the two pieces are fp16 numbers of the operand times a power-of-two scale per tensor ("f16x2", precision 4); the
// partial blocks are rescaled by 2^-(k_x + k_gy) when they are flushed
// PC (round 5; CB = 64, fp16 pieces): the workgroup has EIGHT waves with two roles - waves 0..3 (one per SIMD) only read fragments and issue
// MFMAs, waves 4..7 (one per SIMD, raised priority) own the staging: they wait for the loads of tile k + 1, cut them into pieces, write the
// other LDS image and request tile k + 2, all inside the consumers' tile k.  Same tiles per workgroup, same MFMA order, same partial
// blocks as the four-wave form: the results are bitwise the same; what changes is that the cut's VALU work and the LDS writes no longer
// sit in the instruction stream of the wave that feeds the matrix pipe (which is alone on its SIMD and stalls for every one of them).
// Two waves per SIMD leave 256 registers per wave: 144 accumulators + <= 112 for fragments and addresses.
template <int CB, bool BNI = false, int NP = 3, bool F16 = false, bool PC = false, bool MAP4 = false, bool BNA = false>
__global__ void __launch_bounds__(PC ? 512 : 256, 1) wgrad_x3_kernel(WX3Args a) {
    static_assert(!F16 || NP == 2, "fp16 pieces: two");
    static_assert(!BNA || (CB == 64 && F16 && !MAP4), "BatchNorm adjoint on load: 64-channel blocks of fp16 pieces");
    static_assert(!PC || (CB == 64 && F16), "producer / consumer form: 64-channel blocks of fp16 pieces");
    static_assert(!MAP4 || (CB == 64 && F16), "4 x 4 maps: 64-channel blocks of fp16 pieces");
    using G = WXGeom<CB, NP, MAP4>;
    constexpr int GS = G::GS, HS = G::HS, NG = G::NG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool producer = PC && threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // (PC: index within the role)
    const int wm = CB == 64 ? (wave & 1) : 0, wn = CB == 64 ? (wave >> 1) : 0;      // CB = 64: cout half, cin half of the block
    // (batched launches: this workgroup's layer and its operands - compile-time field names, scalar selects: no indexing into the arguments)
    int bid = (int)blockIdx.x;
    const float* aX = a.X; const float* aGY = a.GY; float* aOut = a.Out; const float* aBni = a.bni;
    const unsigned* aAmaxX = a.amax_x; const unsigned* aAmaxGY = a.amax_gy;
    if (a.nlayers > 1) {
        const int layer = bid / a.wg_per_layer;
        bid -= layer * a.wg_per_layer;
        if (layer == 1) { aX = a.X1; aGY = a.GY1; aOut = a.Out1; aBni = a.bni1; aAmaxX = a.amax_x1; aAmaxGY = a.amax_gy1; }
        else if (layer == 2) { aX = a.X2; aGY = a.GY2; aOut = a.Out2; aBni = a.bni2; aAmaxX = a.amax_x2; aAmaxGY = a.amax_gy2; }
        else if (layer == 3) { aX = a.X3; aGY = a.GY3; aOut = a.Out3; aBni = a.bni3; aAmaxX = a.amax_x3; aAmaxGY = a.amax_gy3; }
    }
    const int split = bid % a.nsplit, pair = bid / a.nsplit;
    const int co0 = (pair / a.cbi) * CB, ci0 = (pair % a.cbi) * CB;
    constexpr unsigned OOB = 0x80000000u;
    float f16_sx = 1.0f, f16_sg = 1.0f;
    int f16_kout = 0;
    if constexpr (F16) {
        const int kx = bh_f16_scale_exp(bh_amax_read(aAmaxX, lane)), kg = bh_f16_scale_exp(bh_amax_read(aAmaxGY, lane));
        f16_sx = __builtin_bit_cast(float, (unsigned)(127 + kx) << 23);
        f16_sg = __builtin_bit_cast(float, (unsigned)(127 + kg) << 23);
        f16_kout = -(kx + kg);
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aX), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aGY), 0, a.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNA ? a.bna_z : aGY), 0, a.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BNA && a.bna_y) ? a.bna_y : aGY), 0, a.gy_bytes, 0x00020000);

    // ---- staging slots of this thread (tile independent parts): slot = (pixel, 8-channel group) -> 2 dwordx4, 3 ds_write_b128 ----
    // gy: 64 pixels x NG groups (GS per thread); halo: 100 x NG slots (HS per thread; the slots past the last one repeat it -
    // same address, same data - so that the staging code has no branch and a tile step stays one scheduling region)
    int g_pix[GS], g_lds[GS], g_cg[GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) {
        const int q = j * 256 + tid, p = q / NG, cg = q % NG;     // p = 0..63
        if constexpr (MAP4) g_pix[j] = (((p >> 5) * 2 + ((p >> 2) & 1)) << 4) + (((p >> 3) & 3) << 2) + (p & 3);      // image (py / 4, px / 4), its pixel (py % 4, px % 4)
        else g_pix[j] = (p >> 3) * a.W + (p & 7);
        g_lds[j] = (p >> 3) * G::GROW + (p & 7) * G::PIX + cg * 16;
        g_cg[j] = cg * 8;
    }
    int h_y[HS], h_x[HS], h_lds[HS], h_cg[HS];
#pragma unroll
    for (int j = 0; j < HS; ++j) {
        const int q = min(j * 256 + tid, G::XW * G::XW * NG - 1);
        const int hp = q / NG, cg = q % NG;                       // hp = 0..99 (MAP4: 0..143)
        const int hy = MAP4 ? hp / 12 : (hp * 205) >> 11, hx = hp - hy * G::XW;
        if constexpr (MAP4) {
            // h_x: the image of the slot (0..3); h_y: its pixel offset from the tile's first image, or -1 on a border
            const int sy = hy / 6, iy = hy - sy * 6 - 1, sx = hx / 6, ix = hx - sx * 6 - 1;
            h_x[j] = sy * 2 + sx;
            h_y[j] = ((unsigned)iy < 4u && (unsigned)ix < 4u) ? ((sy * 2 + sx) << 4) + (iy << 2) + ix : -1;
        } else { h_y[j] = hy - 1; h_x[j] = hx - 1; }
        h_lds[j] = G::XBASE + hy * G::XROW + hx * G::PIX + cg * 16;
        h_cg[j] = cg * 8;
    }

    float4 rg[GS][2], rx[HS][2];
    float4 rz[BNA ? GS : 1][2], ry[BNA ? GS : 1][2];          // BNA: the BatchNorm's input (and saved output) of the gy slots
    int g_tb = 0;                                                 // BNA: byte offset of the tile's statistics group in the coefficient table
    const bool bna_rd_y = BNA && a.bna_y != nullptr;
    bool h_ok[HS];                                                // BNI: slot inside the image / statistics group of its image,
    int h_tb[HS];                                                 //      as byte offset of the slot's coefficients in the LDS table
#pragma unroll
    for (int j = 0; j < HS; ++j) { h_ok[j] = false; h_tb[j] = 0; }
    constexpr int TB0 = 2 * G::LDS;                               // [groups][CB] x (scale, shift) of this workgroup's input-channel block
    // BNA: [groups][8 channel groups][5 coefficients: sc, kz, kc, msc, msh][8 channels] floats behind the BNI table; one word for the bound
    const int TBA0 = TB0 + (BNI ? a.bni_groups * CB * 8 : 0);
    constexpr int TBA_GRP = (CB / 8) * 5 * 8 * 4;                 // bytes per statistics group
    if constexpr (BNA) {
        // coefficient table of the workgroup's 64 output channels (this thread: channel i % 64 of group i / 64) and the bound of |g|
        unsigned* const bnd = reinterpret_cast<unsigned*>(smem + TBA0 + a.bna_groups * TBA_GRP);
        if (threadIdx.x == 0) *bnd = 0u;
        __syncthreads();
        if (!PC || producer) {
            const float maxd = __builtin_bit_cast(float, bh_amax_read(aAmaxGY, lane));
            const double inv_rows = 1.0 / (double)a.bna_rows;
            const float invn = 1.0f / (float)a.bna_rows, sqn = sqrtf((float)a.bna_rows);
            for (int i = tid; i < a.bna_groups * CB; i += 256) {
                const int grp = i / CB, ch = i - grp * CB, c = co0 + ch;
                // (the arithmetic of bn_bwd_apply_kernel's coefficients, csrc/bn.hip: mean / variance in double, the rest in float)
                const double m = bn_sum_total(a.bna_stats, a.bna_groups, grp, a.Co, c, 0) * inv_rows;
                double var = bn_sum_total(a.bna_stats, a.bna_groups, grp, a.Co, c, 1) * inv_rows - m * m;
                if (var < 0) var = 0;
                const float mean = (float)m, invstd = 1.0f / sqrtf((float)var + a.bna_eps);
                const float sc = (a.bna_gamma ? a.bna_gamma[c] : 1.f) * invstd, sh = (a.bna_beta ? a.bna_beta[c] : 0.f) - mean * sc;
                const float k1 = (float)bn_sum_total(a.bna_sums, a.bna_groups, grp, a.Co, c, 0) * invn;
                const float k2 = (float)bn_sum_total(a.bna_sums, a.bna_groups, grp, a.Co, c, 1) * invn;
                // g = sc (dm - k1 - (z - mean) invstd k2) = sc dm + kz z + kc
                const float kz = -sc * invstd * k2, kc = sc * (mean * invstd * k2 - k1);
                float* const tb = reinterpret_cast<float*>(smem + TBA0 + grp * TBA_GRP + (ch >> 3) * 160) + (ch & 7);
                tb[0] = sc; tb[8] = kz; tb[16] = kc; tb[24] = sc; tb[32] = sh;
                const float bound = fabsf(sc) * (maxd + fabsf(k1) + sqn * fabsf(k2));
                atomicMax(bnd, __builtin_bit_cast(unsigned, bound));       // (bits of a non-negative float order like integers; NaN / inf: loud)
            }
        }
        __syncthreads();
        if constexpr (F16) {
            const int kg = bh_f16_scale_exp(*bnd);
            f16_sg = __builtin_bit_cast(float, (unsigned)(127 + kg) << 23);
            f16_kout = -(bh_f16_scale_exp(bh_amax_read(aAmaxX, lane)) + kg);
        }
    }
    if constexpr (BNI) {
        if (!PC || producer) {
            for (int i = tid; i < a.bni_groups * CB; i += 256) {
                const int grp = i / CB, ch = i - grp * CB;
                reinterpret_cast<float2*>(smem + TB0)[i] = reinterpret_cast<const float2*>(aBni)[grp * a.Ci + ci0 + ch];
            }
        }
    }
    if constexpr (BNI) __syncthreads();
    const float bni_lo = a.bni_relu ? 0.0f : -__builtin_inff();   // v_maximum3_f32: NaN propagates (fmaxf would drop it), -inf passes everything
    // loads of tile t.  part -1: all; 0: the gy slots; 1 / 2: the first / second half of the halo slots (each register group is
    // re-requested as soon as its slots have been written out, so every load has most of a tile of MFMAs to arrive)
    auto issue_part = [&](int t, auto PART) {
        constexpr int part = decltype(PART)::value;
        // (multiply-high division with host-made reciprocals: exact for t * divisor < 2^32; no branch in the tile step)
        const int img = a.tiles_per_img == 1 ? t : (int)__umulhi((unsigned)t, a.m_tpi), r = t - img * a.tiles_per_img;
        const int ty = a.tiles_x == 1 ? r : (int)__umulhi((unsigned)r, a.m_tx), tx = r - ty * a.tiles_x;
        const int org = MAP4 ? t * 64 : (img * a.H + ty * 8) * a.W + tx * 8;     // pixel index of the tile's (0, 0) (MAP4: of its first image)
        if constexpr (part <= 0) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const unsigned off = ((unsigned)(org + g_pix[j]) * (unsigned)a.Co + (unsigned)(co0 + g_cg[j])) * 4u;
                rg[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
                rg[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off + 16u, 0, 0));
                if constexpr (BNA) {
                    rz[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, off, 0, 0));
                    rz[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, off + 16u, 0, 0));
                    if (bna_rd_y) {
                        ry[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0));
                        ry[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, off + 16u, 0, 0));
                    }
                }
            }
            if constexpr (BNA) g_tb = TBA0 + (img / a.bna_ipg) * TBA_GRP;
        }
#pragma unroll
        for (int j = 0; j < HS; ++j) {
            if (part >= 0 && (part == 0 || (j < HS / 2 ? 1 : 2) != part)) continue;      // (part 1: the first HS / 2 halo slots, part 2: the rest)
            const int y = ty * 8 + h_y[j], x = tx * 8 + h_x[j];
            const unsigned in = MAP4 ? ((unsigned)(org + h_y[j]) * (unsigned)a.Ci + (unsigned)(ci0 + h_cg[j])) * 4u
                                     : ((unsigned)(org + h_y[j] * a.W + h_x[j]) * (unsigned)a.Ci + (unsigned)(ci0 + h_cg[j])) * 4u;
            const bool ok = MAP4 ? h_y[j] >= 0 : ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W);
            const unsigned off = ok ? in : OOB;                   // outside the image: zeros
            if constexpr (BNI) { h_ok[j] = ok; h_tb[j] = TB0 + (((MAP4 ? 4 * t + h_x[j] : img) / a.bni_ipg) * CB + h_cg[j]) * 8; }
            rx[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
            rx[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off + 16u, 0, 0));
        }
    };
    // slot 0 .. GS-1: gy; GS .. GS+HS-1: halo
    auto stage_slot = [&](auto SLOT, char* img) {
        constexpr int sl = decltype(SLOT)::value;
        uint4 p[3];
        if constexpr (sl < GS) {
            if constexpr (BNA) {
                // the BatchNorm adjoint of the slot's 8 channels: g = sc * mask(d) + kz * z + kc (a gy tile has no padding: every slot is a pixel)
                const float4* tb = reinterpret_cast<const float4*>(smem + g_tb + (g_cg[sl] >> 3) * 160);
                const float4 sc0 = tb[0], sc1 = tb[1], kz0 = tb[2], kz1 = tb[3], kc0 = tb[4], kc1 = tb[5], ms0 = tb[6], ms1 = tb[7], mh0 = tb[8], mh1 = tb[9];
                float4& d0 = rg[sl][0];
                float4& d1 = rg[sl][1];
                const float4 z0 = rz[sl][0], z1 = rz[sl][1];
                if (a.bna_relu) {
                    float4 y0, y1;
                    if (bna_rd_y) { y0 = ry[sl][0]; y1 = ry[sl][1]; }
                    else {
                        y0 = make_float4(__builtin_fmaf(z0.x, ms0.x, mh0.x), __builtin_fmaf(z0.y, ms0.y, mh0.y), __builtin_fmaf(z0.z, ms0.z, mh0.z), __builtin_fmaf(z0.w, ms0.w, mh0.w));
                        y1 = make_float4(__builtin_fmaf(z1.x, ms1.x, mh1.x), __builtin_fmaf(z1.y, ms1.y, mh1.y), __builtin_fmaf(z1.z, ms1.z, mh1.z), __builtin_fmaf(z1.w, ms1.w, mh1.w));
                    }
                    if (!(y0.x > 0.f)) d0.x = 0.f; if (!(y0.y > 0.f)) d0.y = 0.f; if (!(y0.z > 0.f)) d0.z = 0.f; if (!(y0.w > 0.f)) d0.w = 0.f;
                    if (!(y1.x > 0.f)) d1.x = 0.f; if (!(y1.y > 0.f)) d1.y = 0.f; if (!(y1.z > 0.f)) d1.z = 0.f; if (!(y1.w > 0.f)) d1.w = 0.f;
                }
                d0.x = __builtin_fmaf(sc0.x, d0.x, __builtin_fmaf(kz0.x, z0.x, kc0.x)); d0.y = __builtin_fmaf(sc0.y, d0.y, __builtin_fmaf(kz0.y, z0.y, kc0.y));
                d0.z = __builtin_fmaf(sc0.z, d0.z, __builtin_fmaf(kz0.z, z0.z, kc0.z)); d0.w = __builtin_fmaf(sc0.w, d0.w, __builtin_fmaf(kz0.w, z0.w, kc0.w));
                d1.x = __builtin_fmaf(sc1.x, d1.x, __builtin_fmaf(kz1.x, z1.x, kc1.x)); d1.y = __builtin_fmaf(sc1.y, d1.y, __builtin_fmaf(kz1.y, z1.y, kc1.y));
                d1.z = __builtin_fmaf(sc1.z, d1.z, __builtin_fmaf(kz1.z, z1.z, kc1.z)); d1.w = __builtin_fmaf(sc1.w, d1.w, __builtin_fmaf(kz1.w, z1.w, kc1.w));
            }
            bh_split8_any<NP, F16>(rg[sl][0], rg[sl][1], f16_sg, p);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<uint4*>(img + g_lds[sl] + pc * G::GP) = p[pc];
        } else if constexpr (sl < GS + HS) {
            constexpr int j = sl - GS;
            if constexpr (BNI) {
                const float4* tb = reinterpret_cast<const float4*>(smem + h_tb[j]);
                const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
                const bool ok = h_ok[j];
                float4& u = rx[j][0];
                float4& v = rx[j][1];
                u.x = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.x, t0.x, t0.y), bni_lo) : 0.f; u.y = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.y, t0.z, t0.w), bni_lo) : 0.f;
                u.z = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.z, t1.x, t1.y), bni_lo) : 0.f; u.w = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.w, t1.z, t1.w), bni_lo) : 0.f;
                v.x = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.x, t2.x, t2.y), bni_lo) : 0.f; v.y = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.y, t2.z, t2.w), bni_lo) : 0.f;
                v.z = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.z, t3.x, t3.y), bni_lo) : 0.f; v.w = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.w, t3.z, t3.w), bni_lo) : 0.f;
            }
            bh_split8_any<NP, F16>(rx[j][0], rx[j][1], f16_sx, p);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<uint4*>(img + h_lds[j] + pc * G::XP) = p[pc];
        }
    };
#define WX_SLOT(n, img) stage_slot(std::integral_constant<int, (n)>{}, img)
#define WX_PART(n, t) issue_part(t, std::integral_constant<int, (n)>{})

    // tiles of this workgroup: split, split + nsplit, ...
    const int nt = a.ntiles > split ? (a.ntiles - split + a.nsplit - 1) / a.nsplit : 0;
    if constexpr (PC) {
        if (producer) {
            // ---- staging waves: image k & 1 is complete at the barrier that opens the consumers' tile k ----
            __builtin_amdgcn_s_setprio(2);
            if (nt > 0) {
                WX_PART(-1, split);
                WX_SLOT(0, smem); WX_SLOT(1, smem); WX_SLOT(2, smem); WX_SLOT(3, smem); WX_SLOT(4, smem); WX_SLOT(5, smem); WX_SLOT(6, smem);
                if (nt > 1) WX_PART(-1, split + a.nsplit);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            for (int k = 0; k < nt; ++k) {
                if (k + 1 < nt) {
                    char* const nimg = smem + ((k + 1) & 1) * G::LDS;
                    WX_SLOT(0, nimg); WX_SLOT(1, nimg); WX_SLOT(2, nimg); WX_SLOT(3, nimg); WX_SLOT(4, nimg); WX_SLOT(5, nimg); WX_SLOT(6, nimg);
                    if (k + 2 < nt) WX_PART(-1, split + (k + 2) * a.nsplit);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            return;
        }
    }

    // ---- fragment addresses.  16-lane group (lane >> 4) & 1 -> channels 16..31 of the wave's 32; lane >> 5 = K half.
    // Within a group lane 4j + g points at row j of the 2x2 pixel block (j >> 1 down, j & 1 right), channels 4g..4g+3; the
    // K = 16 step ks covers tile rows 2ks, 2ks + 1: K half kh2 and read r take the block at columns 4 kh2 + 2r.
    // CB = 32: the wave's 16-pixel step (ks = wave) is folded into the lane base.
    const int L = lane & 15, jj = L >> 2, gq = L & 3, gsel = (lane >> 4) & 1, kh2 = lane >> 5;
    const int chan_b = (16 * gsel + 4 * gq) * 2;
    const int ksw = CB == 64 ? 0 : 2 * wave;
    const int laneA = ((jj >> 1) + ksw) * G::GROW + (4 * kh2 + (jj & 1)) * G::PIX + chan_b + wm * 64;
    const int laneB = G::XBASE + ((jj >> 1) + ksw) * G::XROW + ((MAP4 ? 6 : 4) * kh2 + (jj & 1)) * G::PIX + chan_b + wn * 64;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#define WX_TR(base, off) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_ptr)(smem + (base) + (off)))
#define WX_OPER(lo_, hi_) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7))
#define WX_LOAD_A(dst, ks) _Pragma("unroll") for (int pc = 0; pc < NP; ++pc) {                                          \
        const i16x4 v0_ = WX_TR(la, pc * G::GP + 2 * (ks) * G::GROW);                                                   \
        const i16x4 v1_ = WX_TR(la, pc * G::GP + 2 * (ks) * G::GROW + 2 * G::PIX);                                      \
        dst[pc] = WX_OPER(v0_, v1_); }
// (halo row of tile row 2 ks at kernel row 0: 2 ks - MAP4: rows 0..3 of the tile are sub-map rows 0..3 behind their border, rows 4..7 start at 6)
#define WX_BROW(ks) (MAP4 ? 6 * ((ks) >> 1) + 2 * ((ks) & 1) : 2 * (ks))
#define WX_LOAD_B(dst, ks, tap) _Pragma("unroll") for (int pc = 0; pc < NP; ++pc) {                                     \
        const i16x4 v0_ = WX_TR(lb, pc * G::XP + (WX_BROW(ks) + (tap) / 3) * G::XROW + ((tap) % 3) * G::PIX);            \
        const i16x4 v1_ = WX_TR(lb, pc * G::XP + (WX_BROW(ks) + (tap) / 3) * G::XROW + ((tap) % 3 + 2) * G::PIX);        \
        dst[pc] = WX_OPER(v0_, v1_); }

    if constexpr (!PC) {
        if (nt > 0) {
            WX_PART(-1, split);
            WX_SLOT(0, smem); WX_SLOT(1, smem); WX_SLOT(2, smem); WX_SLOT(3, smem); WX_SLOT(4, smem); WX_SLOT(5, smem); WX_SLOT(6, smem);   // (slots past GS + HS: nothing)
        }
    }
    if constexpr (PC) asm volatile("s_barrier" ::: "memory"); else __syncthreads();
    if constexpr (!PC) { if (nt > 1) WX_PART(-1, split + a.nsplit); }
    constexpr int NSTEP = CB == 64 ? 36 : 9;                       // (ks, tap) steps of a wave per tile
    constexpr int NQ = CB == 64 ? 9 : 5;                           // pairs per half of the (two-level, fully unrolled) pair loop
    for (int k = 0; k < nt; ++k) {
        const int cb = (k & 1) * G::LDS;
        char* const nimg = smem + (G::LDS - cb);                   // the other image
        const int la = laneA + cb, lb = laneB + cb;
        const int tnext = split + min(k + 2, nt - 1) * a.nsplit;   // (past the end: re-request the last tile, never used)
        bf16x8 af[2][NP], bf[2][2][NP];                            // A by ks parity; B by step-pair parity and step parity
        WX_LOAD_A(af[0], 0);
        WX_LOAD_B(bf[0][0], 0, 0);
        WX_LOAD_B(bf[0][1], 0, 1);
        // steps (ks, tap) are taken two at a time so that consecutive MFMAs alternate between two accumulators (six
        // dependent MFMAs on one accumulator in a row leave the pipe waiting for its own result)
#pragma unroll
        for (int hp = 0; hp < (CB == 64 ? 2 : 1); ++hp)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pi = hp * NQ + q, s0 = 2 * pi, s1 = s0 + 1, pb = pi & 1;
#pragma unroll
            for (int f = s0 + 2; f <= s1 + 2; ++f)
                if (f < NSTEP) {
                    if (f % 9 == 0) { WX_LOAD_A(af[(f / 9) & 1], f / 9); }
                    WX_LOAD_B(bf[pb ^ 1][f & 1], f / 9, f % 9);
                }
            // the next tile: cut and write a register group, then re-request it for the tile after
            if constexpr (PC) {
            } else if (CB == 64) {
                if (pi == 3) { WX_SLOT(0, nimg); WX_SLOT(1, nimg); WX_PART(0, tnext); }
                if (pi == 8) { WX_SLOT(2, nimg); WX_SLOT(3, nimg); WX_PART(1, tnext); }
                if (pi == 13) { WX_SLOT(4, nimg); WX_SLOT(5, nimg); WX_SLOT(6, nimg); WX_PART(2, tnext); }
            } else {
                if (pi == 1) { WX_SLOT(0, nimg); WX_PART(0, tnext); }
                if (pi == 2) { WX_SLOT(1, nimg); WX_PART(1, tnext); }
                if (pi == 3) { WX_SLOT(2, nimg); WX_PART(2, tnext); }
            }
            const int k0 = (s0 / 9) & 1, t0 = s0 % 9;
#define WX_MFMA(A_, B_, C_) (F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0) \
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0))
#define WX_MM0(PA, PB) acc[t0] = WX_MFMA(af[k0][PA], bf[pb][0][PB], acc[t0])
            if (s1 < NSTEP) {
                const int k1 = (s1 / 9) & 1, t1 = s1 % 9;
#define WX_MM1(PA, PB) acc[t1] = WX_MFMA(af[k1][PA], bf[pb][1][PB], acc[t1])
                // small partial products first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                if constexpr (NP == 3) { WX_MM0(2, 0); WX_MM1(2, 0); WX_MM0(0, 2); WX_MM1(0, 2); WX_MM0(1, 1); WX_MM1(1, 1); }
                WX_MM0(1, 0); WX_MM1(1, 0); WX_MM0(0, 1); WX_MM1(0, 1); WX_MM0(0, 0); WX_MM1(0, 0);
#undef WX_MM1
            } else {
                if constexpr (NP == 3) { WX_MM0(2, 0); WX_MM0(0, 2); WX_MM0(1, 1); }
                WX_MM0(1, 0); WX_MM0(0, 1); WX_MM0(0, 0);
            }
#undef WX_MM0
#undef WX_MFMA
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (LDS only: the requested tile stays in flight)
    }
#undef WX_LOAD_A
#undef WX_LOAD_B
#undef WX_BROW
#undef WX_SLOT
#undef WX_PART
#undef WX_TR
#undef WX_OPER
    if (a.noflush == 1) return;
    if constexpr (F16) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_ldexpf(acc[t][r], f16_kout);
    }
    // C/D layout: column = lane & 31 (cin), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (cout).  All workgroups finish together and
    // every one adds to the same block: each starts at another tap so that they do not queue on the same addresses.
    const int l31 = lane & 31;
    const int rot = (int)(blockIdx.x % 9u);
    for (int i = 0; i < 9; ++i) {
        int tap = i + rot;
        if (tap >= 9) tap -= 9;
        float* const o = aOut + ((long long)(co0 + wm * 32 + 4 * kh2) * 9 + tap) * a.Ci + ci0 + wn * 32 + l31;
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
// elements of a block's partial order x 4 split groups (each thread adds a contiguous quarter of the splits, 8 loads in
// flight), the four sub-sums are combined in fixed order through LDS: bitwise reproducible.
// CB = 64: a partial block is [tap][wave = quadrant][r][lane], an element has nsplit terms.
// CB = 32: the four waves of a workgroup hold four K-split terms of the same [tap][r][lane] element: 4 nsplit terms.
// (batched launches: blockIdx.y = layer * pairs_per_layer + block pair; out1 .. out3 the gradients of layers 1 .. 3)
template <int CB>
__global__ void __launch_bounds__(256) wgrad_x3_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int nsplit,
                                                              int cbi, int Ci, float* __restrict__ out1 = nullptr, float* __restrict__ out2 = nullptr,
                                                              float* __restrict__ out3 = nullptr, int pairs_per_layer = 1 << 30) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;                           // CB 64: [tap][wave][r][lane] < 36864; CB 32: [tap][r][lane] < 9216
    const int layer = (int)blockIdx.y / pairs_per_layer, pair = (int)blockIdx.y - layer * pairs_per_layer;
    if (layer == 1) out = out1; else if (layer == 2) out = out2; else if (layer == 3) out = out3;
    const int nterm = CB == 64 ? nsplit : 4 * nsplit;              // term s of CB = 32: workgroup s >> 2, wave s & 3
    const int per = (nterm + 3) / 4, s0 = grp * per, s1 = min(nterm, s0 + per);
    const float* const base = partials + (size_t)blockIdx.y * nsplit * 36864;
    auto term = [&](int s) -> const float* {
        if (CB == 64) return base + (size_t)s * 36864 + idx;
        const int tap = idx >> 10, rl = idx & 1023;
        return base + (size_t)(s >> 2) * 36864 + ((tap * 4 + (s & 3)) << 10) + rl;
    };
    float sum = 0.f;
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *term(s + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < s1; ++s) sum += *term(s);
    red[grp][e] = sum;
    __syncthreads();
    if (grp == 0) {
        const float tot = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const int lane = idx & 63, r = (idx >> 6) & 15;
        const int wave = CB == 64 ? (idx >> 10) & 3 : 0, tap = CB == 64 ? idx >> 12 : idx >> 10;
        const int wm = wave & 1, wn = wave >> 1, kh2 = lane >> 5, l31 = lane & 31;
        const int co = (pair / cbi) * CB + wm * 32 + 4 * kh2 + (r & 3) + 8 * (r >> 2), ci = (pair % cbi) * CB + wn * 32 + l31;
        out[((long long)co * 9 + tap) * Ci + ci] += tot;
    }
}

BH_KNOB(g_wx3_target, 256); BH_KNOB(g_wx3_noflush, 0); BH_KNOB(g_wx3_pc, 1); BH_KNOB(g_wx3_map4, 1);
#ifdef BH_TUNING
void bh_wgrad_x3_tune(int what, int v) { if (what == 0) g_wx3_target = v; else if (what == 1) g_wx3_noflush = v; else if (what == 2) g_wx3_pc = v; else if (what == 3) g_wx3_map4 = v; }
#endif

// *taken = 1 when the shape is eligible (3x3 / stride 1 / pad 1, NHWC, H and W multiples of 8, channels multiples of 32).
// ws != NULL: deterministic reduction through ws (ws_need != NULL: dry run that only reports the bytes needed)
// further layers of a batched launch (bh_conv_wgrad_batch): same geometry, arithmetic and BatchNorm-on-load form as layer 0
struct WX3Batch {
    int n;                                     // extra layers (1 .. 3)
    const float* x[3]; const float* gy[3]; float* gw[3];
    const void* a_bound[3]; const void* b_bound[3]; const float* bni[3];
};

int bh_wgrad_x3_try(const float* x, const float* gy, float* gw, const bh_conv_desc* d, hipStream_t stream, int* taken, float* ws,
                    long long ws_bytes, long long* ws_need, const bh_bn_in* bni, const bh_bn_adj* bna, const WX3Batch* extra) {
    *taken = 0;
    if (d->transposed || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->in_nchw || d->out_nchw) return BH_OK;
    // 4 x 4 maps (round 5: layer4 of the ResNet-34 regressor): fp16 pieces, 64-channel blocks, four images per tile
    const bool map4 = d->Hi == 4 && d->Wi == 4 && d->Ho == 4 && d->Wo == 4 && d->N % 4 == 0 && d->Ci % 64 == 0 && d->Co % 64 == 0 &&
                      d->precision == 4 && d->a_bound && d->b_bound && g_wx3_map4;
    if (!map4 && (d->Hi % 8 || d->Wi % 8 || d->Ho != d->Hi || d->Wo != d->Wi || d->Ci % 32 || d->Co % 32)) return BH_OK;
    const long long xe = (long long)d->N * d->Hi * d->Wi * d->Ci, ge = (long long)d->N * d->Hi * d->Wi * d->Co;
    if (xe >= (1ll << 29) || ge >= (1ll << 29)) return BH_OK;
    const int cb = (d->Ci % 64 == 0 && d->Co % 64 == 0) ? 64 : 32;
    WX3Args a = {};
    a.X = x; a.GY = gy; a.Out = gw;
    a.N = d->N; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.tiles_x = d->Wi / 8; a.tiles_per_img = (d->Hi / 8) * a.tiles_x; a.ntiles = d->N * a.tiles_per_img;
    if (map4) { a.tiles_x = 1; a.tiles_per_img = 1; a.ntiles = d->N / 4; }
    if ((long long)a.ntiles * a.tiles_per_img >= (1ll << 32)) return BH_OK;
    a.m_tx = (unsigned)((1ull << 32) / (unsigned)a.tiles_x + 1ull);
    a.m_tpi = (unsigned)((1ull << 32) / (unsigned)a.tiles_per_img + 1ull);
    a.cbi = d->Ci / cb;
    const int pairs = a.cbi * (d->Co / cb);
    // one workgroup per CU - or, when the caller says that another stream shares the GPU (BH_ROUTE_WX3_SHARED), 160: in-step sweep of
    // round 5 (tools/ab_hook.sh "-30,n"): 128-192 all beat 256 by 0.15-0.19 ms per step on configs[1], 160 also on configs[3]
    const int target = (d->route & BH_ROUTE_WX3_SHARED) && g_wx3_target == 256 ? 160 : g_wx3_target;
    const int L = extra ? extra->n + 1 : 1;     // layers of this launch
    if (extra && (map4 || cb != 64 || bna || !ws)) return BH_OK;
    int ns = target / (pairs * L);
    if (ns > a.ntiles) ns = a.ntiles;
    if (ns >= 8) ns = ns / 8 * 8;               // the workgroups of one split (same tiles, other channel blocks) land on one XCD
    if (ns < 1) ns = 1;
    a.nsplit = ns;
    a.x_bytes = (unsigned)(xe * 4); a.gy_bytes = (unsigned)(ge * 4);
    a.noflush = g_wx3_noflush;
    if (bni) {
        if (!bni->table || bni->groups < 1 || bni->groups > 4 || d->N % bni->groups) return BH_E_BADARG;
        a.bni = bni->table; a.bni_relu = bni->relu; a.bni_groups = bni->groups; a.bni_ipg = d->N / bni->groups;
    }
    if (bna) {
        // the BatchNorm adjoint on load: fp16 pieces, 64-channel blocks, whole images per statistics group, no 4 x 4 maps
        if (!bna->z || !bna->stats || !bna->sums || bna->groups < 1 || bna->groups > 4 || d->N % bna->groups) return BH_E_BADARG;
        if (cb != 64 || map4 || !(d->precision == 4 && d->a_bound && d->b_bound) || !ws) return BH_OK;
        a.bna_z = bna->z; a.bna_y = (bna->relu && bna->y) ? bna->y : nullptr; a.bna_stats = bna->stats; a.bna_sums = bna->sums;
        a.bna_gamma = bna->gamma; a.bna_beta = bna->beta; a.bna_eps = bna->eps; a.bna_relu = bna->relu ? 1 : 0;
        a.bna_groups = bna->groups; a.bna_ipg = d->N / bna->groups;
        a.bna_rows = a.bna_ipg * d->Hi * d->Wi;
    }
    const int tb_bytes = (bni ? bni->groups * cb * 8 : 0) + (bna ? bna->groups * 1280 + 16 : 0);
    const long long need = (long long)pairs * L * ns * 36864 * 4;
    if (extra) {
        a.nlayers = L; a.wg_per_layer = pairs * ns;
        a.X1 = extra->x[0]; a.GY1 = extra->gy[0]; a.Out1 = extra->gw[0]; a.bni1 = extra->bni[0];
        a.amax_x1 = reinterpret_cast<const unsigned*>(extra->a_bound[0]); a.amax_gy1 = reinterpret_cast<const unsigned*>(extra->b_bound[0]);
        if (L > 2) { a.X2 = extra->x[1]; a.GY2 = extra->gy[1]; a.Out2 = extra->gw[1]; a.bni2 = extra->bni[1];
                     a.amax_x2 = reinterpret_cast<const unsigned*>(extra->a_bound[1]); a.amax_gy2 = reinterpret_cast<const unsigned*>(extra->b_bound[1]); }
        if (L > 3) { a.X3 = extra->x[2]; a.GY3 = extra->gy[2]; a.Out3 = extra->gw[2]; a.bni3 = extra->bni[2];
                     a.amax_x3 = reinterpret_cast<const unsigned*>(extra->a_bound[2]); a.amax_gy3 = reinterpret_cast<const unsigned*>(extra->b_bound[2]); }
    }
    if (ws) {
        if (ws_need) *ws_need = need;
        else if (ws_bytes < need) return BH_E_BADARG;
        a.partials = ws;
    }
    const int np = d->precision >= 3 ? 2 : 3;
    // precision 4 (two fp16 pieces): with the magnitude records of both operands; without them the exact three-piece form runs
    const bool f16 = d->precision == 4 && d->a_bound && d->b_bound;
    a.amax_x = reinterpret_cast<const unsigned*>(d->a_bound); a.amax_gy = reinterpret_cast<const unsigned*>(d->b_bound);
    // fp16 pieces, 64-channel blocks: the eight-wave producer / consumer form (same partial blocks, bitwise the same sums)
    // (the eight-wave form: a per-call route bit - include/bihome.h BH_ROUTE_WX3_PC; the tuning build's knob can switch it off)
    const bool pc = f16 && cb == 64 && (d->route & BH_ROUTE_WX3_PC) && g_wx3_pc;
    // (all six template arguments, as rocprofv3 prints the symbol: CB, BNI, NP, F16, PC, MAP4)
    if (bh_query(ws ? "wgrad_x3_kernel<%d,%s,%d,%s%s>+wgrad_x3_reduce_kernel<%d>" : "wgrad_x3_kernel<%d,%s,%d,%s%s>", cb, bni ? "true" : "false",
                 (d->precision == 4 && !f16) ? 3 : np, f16 ? "true" : "false",
                 bna ? (pc ? ",true,false,true" : ",false,false,true") : map4 ? (pc ? ",true,true" : ",false,true") : (pc ? ",true,false" : ",false,false"), cb)) { *taken = 1; return BH_OK; }
    typedef void (*kern_t)(WX3Args);
    if (bna) {
        // (four instantiations: BatchNorm-on-load of x {no, yes} x {four-wave, eight-wave form})
        static const kern_t afn[4] = {wgrad_x3_kernel<64, false, 2, true, false, false, true>, wgrad_x3_kernel<64, true, 2, true, false, false, true>,
                                      wgrad_x3_kernel<64, false, 2, true, true, false, true>, wgrad_x3_kernel<64, true, 2, true, true, false, true>};
        constexpr int lds_a = 2 * WXGeom<64, 2>::LDS;
        static unsigned long long attr_devs_a = 0;
        if (bh_device_once(attr_devs_a)) {
            for (int i = 0; i < 4; ++i) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(afn[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         lds_a + 4 * 64 * 8 + 4 * 1280 + 16);
                if (e != hipSuccess) return (int)e;
            }
        }
        hipLaunchKernelGGL(afn[(pc ? 2 : 0) + (bni ? 1 : 0)], dim3(pairs * ns), dim3(pc ? 512 : 256), lds_a + tb_bytes, stream, a);
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(wgrad_x3_reduce_kernel<64>, dim3(36864 / 64, pairs), dim3(256), 0, stream, ws, gw, ns, a.cbi, d->Ci);
        BH_LAUNCH_CHECK();
        *taken = 1;
        return BH_OK;
    }
    static const kern_t fns[18] = {wgrad_x3_kernel<64, false, 3>, wgrad_x3_kernel<32, false, 3>, wgrad_x3_kernel<64, true, 3>, wgrad_x3_kernel<32, true, 3>,
                                   wgrad_x3_kernel<64, false, 2>, wgrad_x3_kernel<32, false, 2>, wgrad_x3_kernel<64, true, 2>, wgrad_x3_kernel<32, true, 2>,
                                   wgrad_x3_kernel<64, false, 2, true>, wgrad_x3_kernel<32, false, 2, true>, wgrad_x3_kernel<64, true, 2, true>, wgrad_x3_kernel<32, true, 2, true>,
                                   wgrad_x3_kernel<64, false, 2, true, true>, wgrad_x3_kernel<64, true, 2, true, true>,
                                   wgrad_x3_kernel<64, false, 2, true, true, true>, wgrad_x3_kernel<64, true, 2, true, true, true>,
                                   wgrad_x3_kernel<64, false, 2, true, false, true>, wgrad_x3_kernel<64, true, 2, true, false, true>};
    static const int lds_of[18] = {2 * WXGeom<64, 3>::LDS, 2 * WXGeom<32, 3>::LDS, 2 * WXGeom<64, 3>::LDS, 2 * WXGeom<32, 3>::LDS,
                                   2 * WXGeom<64, 2>::LDS, 2 * WXGeom<32, 2>::LDS, 2 * WXGeom<64, 2>::LDS, 2 * WXGeom<32, 2>::LDS,
                                   2 * WXGeom<64, 2>::LDS, 2 * WXGeom<32, 2>::LDS, 2 * WXGeom<64, 2>::LDS, 2 * WXGeom<32, 2>::LDS,
                                   2 * WXGeom<64, 2>::LDS, 2 * WXGeom<64, 2>::LDS, 2 * WXGeom<64, 2, true>::LDS, 2 * WXGeom<64, 2, true>::LDS,
                                   2 * WXGeom<64, 2, true>::LDS, 2 * WXGeom<64, 2, true>::LDS};
    static unsigned long long attr_devs = 0;
    if (bh_device_once(attr_devs)) {
        for (int i = 0; i < 18; ++i) {
            const bool tb = i < 12 ? (i & 2) != 0 : (i & 1) != 0;
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     lds_of[i] + (tb ? 4 * ((i < 12 && (i & 1)) ? 32 : 64) * 8 : 0));
            if (e != hipSuccess) return (int)e;
        }
    }
    const int ki = map4 ? (pc ? 14 : 16) + (bni ? 1 : 0) : pc ? 12 + (bni ? 1 : 0) : (f16 ? 8 : (np == 2 && d->precision != 4) ? 4 : 0) + (bni ? 2 : 0) + (cb == 64 ? 0 : 1);
    hipLaunchKernelGGL(fns[ki], dim3(pairs * L * ns), dim3(pc ? 512 : 256), lds_of[ki] + tb_bytes, stream, a);
    BH_LAUNCH_CHECK();
    if (ws) {
        if (cb == 64) hipLaunchKernelGGL(wgrad_x3_reduce_kernel<64>, dim3(36864 / 64, pairs * L), dim3(256), 0, stream, ws, gw, ns, a.cbi, d->Ci,
                                         a.Out1, a.Out2, a.Out3, pairs);
        else hipLaunchKernelGGL(wgrad_x3_reduce_kernel<32>, dim3(9216 / 64, pairs), dim3(256), 0, stream, ws, gw, ns, a.cbi, d->Ci);
        BH_LAUNCH_CHECK();
    }
    *taken = 1;
    return BH_OK;
}

// Round 6: the fp16-piece weight gradients of 2 .. 4 layers of ONE geometry in one launch (+ one reduce launch).  Same tiles, same MFMA
// order per workgroup; what changes is how many workgroups share a layer's pixels (fewer: one set of partial blocks per LAUNCH instead of
// per layer) - the sums are deterministic, not bitwise those of the single launches.
extern "C" int bh_conv_wgrad_batch(int n, const float* const* x, const float* const* gy, float* const* gw, const bh_conv_desc* const* descs, float* ws,
                                   long long ws_bytes, const bh_bn_in* const* bni, void* stream) {
    if (n < 1 || n > 4 || !x || !gy || !gw || !descs || !ws) return BH_E_BADARG;
    const bh_conv_desc* d0 = descs[0];
    if (!d0 || d0->precision != 4 || !d0->a_bound || !d0->b_bound) return BH_E_UNSUPPORTED;
    WX3Batch ex = {};
    ex.n = n - 1;
    for (int i = 0; i < n; ++i) {
        const bh_conv_desc* di = descs[i];
        if (!di || !x[i] || !gy[i] || !gw[i]) return BH_E_BADARG;
        if (di->N != d0->N || di->Hi != d0->Hi || di->Wi != d0->Wi || di->Ci != d0->Ci || di->Co != d0->Co || di->kh != d0->kh || di->kw != d0->kw ||
            di->stride != d0->stride || di->pad != d0->pad || di->transposed != d0->transposed || di->in_nchw != d0->in_nchw ||
            di->out_nchw != d0->out_nchw || di->precision != d0->precision || di->route != d0->route || !di->a_bound || !di->b_bound)
            return BH_E_BADARG;
        const bh_bn_in* bi = bni ? bni[i] : nullptr;
        const bh_bn_in* b0 = bni ? bni[0] : nullptr;
        if ((bi != nullptr) != (b0 != nullptr) || (bi && (bi->groups != b0->groups || bi->relu != b0->relu || !bi->table))) return BH_E_BADARG;
        if (i > 0) { ex.x[i - 1] = x[i]; ex.gy[i - 1] = gy[i]; ex.gw[i - 1] = gw[i]; ex.a_bound[i - 1] = di->a_bound; ex.b_bound[i - 1] = di->b_bound;
                     ex.bni[i - 1] = bi ? bi->table : nullptr; }
    }
    int taken = 0;
    const int rc = bh_wgrad_x3_try(x[0], gy[0], gw[0], d0, bh_stream(stream), &taken, ws, ws_bytes, nullptr, bni ? bni[0] : nullptr, nullptr,
                                   n > 1 ? &ex : nullptr);
    if (rc != BH_OK) return rc;
    return taken ? BH_OK : BH_E_UNSUPPORTED;
}
