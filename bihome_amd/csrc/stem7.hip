// Forward of the 7x7 / stride 2 / pad 3 stem convolutions (Rethinking.py:31 layer1, ResNet34.py:17 conv1, the extractor's
// resnet.conv1 of PerceptualHead.py:52-55): few input channels (1 grayscale patch, 2 stacked patches, 3 RGB, 6 stacked
// RGB) -> 64 channels, NCHW image planes in, NHWC out.
//
// As an implicit GEMM this is K = 49*Cin (49 ... 294) with a gather that no vector load serves; the generic kernel
// runs it on its scalar path at ~20 TFLOP/s.  Here a persistent workgroup keeps the whole transposed filter bank
// Wt[k = (c, ky, kx)][64] in LDS, and per 8x8 output tile stages the 21x21 input patch of every channel once; the A
// fragment of MFMA step kk is then ONE ds_read_b32 at (lane's pixel base) + (compile-time offset of tap k), the k loop
// is fully unrolled so those offsets are literals (the two half-waves take k = 2kk and 2kk+1).  The output (the only
// large stream: 64 channels x 4 B per pixel) is written as 128-byte rows.  fp32 MFMA (v_mfma_f32_32x32x2_f32) in
// every precision mode.
#include "common.h"
#include "warp_tap.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
BH_KNOB(g_stem_f16, 1);            // (tuning build: the stems' fp16-piece forms on / off - bh_debug_force_tile)
#ifdef BH_TUNING
void bh_stem7_tune(int v) { g_stem_f16 = v; }
#endif

struct Stem7Args {
    const float* x;        // [N][CIN][Hi][Wi]
    const float* w;        // [64][7][7][CIN]
    const float* bias;     // [64] or NULL
    float* y;              // [N][Ho][Wo][64]
    int N, Hi, Wi, Ho, Wo;
    int tiles_x, tiles_per_img, ntiles;
    int relu;              // epilogue ReLU (inference: BatchNorm folded into w / bias)
    // optional: per-channel (sum, sum of squares) of the output for the BatchNorm that follows, per statistics group (the images are
    // `groups` equal stacks): the persistent workgroup keeps the column sums of its tiles in registers (tiles are walked in image
    // order, so the group changes at most groups - 1 times) and adds them with one f64 atomic per channel and moment (round 3: the
    // separate bn_stats pass over the 134 MB output is gone)
    double* bn_sums;
    int groups, imgs_per_group, det;
    // stem7_fwd_f16_kernel<1, true> (round 6): x is not read - the input image is the homography warp of wsrc, made while the patch is fetched
    const float* wsrc;     // [N][1][Hi][Wi] source patches
    const double* H64;     // [N][9]
    float* warped;         // [N][1][Hi][Wi] the warped image (every pixel written once, by the workgroup that owns its tile) or NULL
    float* cov;            // [N][Hi/4][Wi/4] 4 x 4 average of the warped all-ones mask, or NULL
};

template <int CIN>
__global__ void __launch_bounds__(256) stem7_fwd_kernel(Stem7Args a) {
    constexpr int K = 49 * CIN;
    constexpr int KP = (K + 3) / 4 * 4;                 // 52, 100, 148, 296: zero-padded filter rows
    constexpr int PW = 21, PCH = PW * PW;               // input patch of an 8x8 output tile: (8-1)*2 + 7 = 21
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Wt = sm;                                     // [KP][64]
    float* patch = sm + KP * 64;                        // [CIN][21][21]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    for (int i = tid; i < KP * 64; i += 256) Wt[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 64 * K; i += 256) {           // w[n][t][c] -> Wt[c*49 + t][n]
        const int n = i / K, r = i - n * K, t = r / CIN, c = r - t * CIN;
        Wt[(c * 49 + t) * 64 + n] = a.w[i];
    }
    // lane's pixel inside the tile: rows [wm*32, wm*32+32) -> output (py, px) = (wm*4 + l31/8, l31%8)
    const int py = wm * 4 + (l31 >> 3), px = l31 & 7;
    const float* abase = patch + (2 * py) * PW + 2 * px;
    const float* bbase = Wt + wn * 32 + l31;
    const float bv = a.bias ? a.bias[wn * 32 + l31] : 0.f;

    double s1 = 0, s2 = 0;                             // column sums of this lane's channel over the tiles of statistics group cur_grp
    int cur_grp = -1;
    double* const red = reinterpret_cast<double*>(patch);      // [2 wm][64 n][2] (16-byte aligned: KP * 64 floats precede it)
    auto flush_stats = [&]() {
        // the two half-waves hold the two row halves of a column; the two wm waves of a channel range meet in LDS
        __syncthreads();                                 // (patch is free: every wave is past its fragment reads)
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (kh2 == 0) { red[(wm * 64 + wn * 32 + l31) * 2] = s1; red[(wm * 64 + wn * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 128 && cur_grp >= 0) {
            const int n = tid >> 1, mom = tid & 1;
            bh_acc_add(&a.bn_sums[bn_sum_index(0, a.groups, cur_grp, 64, n, mom)], red[n * 2 + mom] + red[(64 + n) * 2 + mom], a.det);
        }
        s1 = 0; s2 = 0;
    };
    // The input patch of the NEXT tile is requested into registers before this tile's MFMAs and written to LDS behind them (round 5: the
    // global-load latency of the staging sat between two barriers of every tile; with it in flight under the MFMAs: one plane 60 -> 58 us,
    // two planes 108 -> 97 us - tools/stem_time.py)
    constexpr int NPF = (CIN * PCH + 255) / 256;         // staged elements per thread
    float pf[NPF];
    auto fetch = [&](int tile_) {
        const int img_ = tile_ / a.tiles_per_img, t_ = tile_ - img_ * a.tiles_per_img;
        const int ty_ = t_ / a.tiles_x, tx_ = t_ - ty_ * a.tiles_x;
        const int iy0 = ty_ * 16 - 3, ix0 = tx_ * 16 - 3;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int i = tid + j * 256;
            const int c = i / PCH, r = i - c * PCH, yy = r / PW, xx = r - yy * PW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            float v = 0.f;
            if (tile_ < a.ntiles && i < CIN * PCH && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                v = a.x[(((size_t)img_ * CIN + c) * a.Hi + iy) * a.Wi + ix];
            pf[j] = v;
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int img = tile / a.tiles_per_img, t = tile - img * a.tiles_per_img;
        if (a.bn_sums) {
            const int grp = img / a.imgs_per_group;      // (workgroup-uniform)
            if (grp != cur_grp) { if (cur_grp >= 0) flush_stats(); cur_grp = grp; }
        }
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        __syncthreads();                                 // previous tile's fragment reads are done (and Wt is complete)
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if (tid + j * 256 < CIN * PCH) patch[tid + j * 256] = pf[j];
        __syncthreads();
        fetch(tile + (int)gridDim.x);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KP / 2; ++kk) {
            // taps of the two half-waves (compile-time): k -> (c, ky, kx) -> patch offset; padded k read offset 0 (Wt row is 0)
            const int k0 = 2 * kk, k1 = 2 * kk + 1;
            const int o0 = k0 < K ? (k0 / 49) * PCH + ((k0 % 49) / 7) * PW + (k0 % 49) % 7 : 0;
            const int o1 = k1 < K ? (k1 / 49) * PCH + ((k1 % 49) / 7) * PW + (k1 % 49) % 7 : 0;
            const float av = abase[kh2 ? o1 : o0];
            const float bw = bbase[(kh2 ? k1 : k0) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw, acc, 0, 0, 0);
        }
        // C/D layout: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            const int oy = ty * 8 + (m >> 3), ox = tx * 8 + (m & 7);
            float v = acc[r] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            a.y[(((size_t)img * a.Ho + oy) * a.Wo + ox) * 64 + wn * 32 + l31] = v;
            acc[r] = v;
        }
        if (a.bn_sums) {
            float q1 = 0.f, q2 = 0.f;                    // 16 elements in float, totals in double
#pragma unroll
            for (int r = 0; r < 16; ++r) { q1 += acc[r]; q2 = __builtin_fmaf(acc[r], acc[r], q2); }
            s1 += (double)q1; s2 += (double)q2;
        }
    }
    if (a.bn_sums && cur_grp >= 0) flush_stats();
}

// Round 6 (round-5 VERDICT item 9 / weak #15: "stems still run the fp32-input MFMA at <= 0.5 of a measured 145 TFLOP/s"): the same forward
// in the fp16-piece arithmetic of the 3x3 layers (common.h F16X2; bh_conv_desc.precision = 4) - every operand as TWO fp16 numbers of
// (value x 2^k), three v_mfma_f32_32x32x16_f16 products per product, fp32 accumulate, the result rescaled by 2^-(kx + kw) (exact).
//   * K order: a k-step of 16 is TWO tap rows (c, ky), (c, ky + 1) of 8 taps each - seven real kx and one pad tap with a zero weight -
//     so the eight consecutive k a lane supplies are eight consecutive patch pixels of one row: four ds_read_b32 per piece (the patch is
//     kept as two fp16 planes, row pitch 22 halfs, pad column zero).  One plane: 4 k-steps = 12 MFMAs of 32 cycles per 32 x 32 block
//     where the fp32-input form issues 26 of 64.
//   * scales: the filter bank's k from its maximum (every workgroup derives it from the 3136 CIN weights while it cuts them into LDS);
//     the patch's k PER TILE from the maximum of the 21 x 21 pixels the workgroup has just fetched (wave maxima through LDS behind the
//     barrier the staging needs anyway) - no magnitude record of the input images is needed, and a dark tile keeps its bits.
// Same tiles, same epilogue (bias, ReLU, BatchNorm sums) as stem7_fwd_kernel.
//
// WARP (one plane; round 6): the image is the homography warp of a source patch (src/data/utils.py:54-59 via PerceptualHead.py:371-401)
// and this stem is its consumer (:377,398).  The fetch of a tile's 21 x 21 patch - which runs one tile ahead, behind the MFMAs of the
// current one - makes the warped pixels instead of loading them: the pixel's tap (warp_tap.h: the arithmetic of warp_fwd4_kernel, bitwise),
// four gathered source pixels, the blend.  The 16 x 16 pixels a tile owns are written to `warped` by their workgroup (the image stays
// available to the caller; nobody reads it back on this path) and their bilinear weight sums go through 1 KB of LDS into the 4 x 4
// averages of the pooled all-ones mask (PerceptualHead.py:380-382,447-459), summed in warp_fwd4_kernel's order: the coverage is bitwise
// what that kernel writes.  The 5-pixel rim of the patch is warped again by the neighbouring tiles (441 / 256 of the taps) - VALU and L2
// gathers in the shadow of the matrix pipe; warp_fwd4_kernel's launch (~12 us for 17 MB: latency-bound) is gone.
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
template <int CIN, bool WARP = false>
__global__ void __launch_bounds__(256) stem7_fwd_f16_kernel(Stem7Args a, const double* __restrict__ Hp /* = a.H64 (WARP) */) {
    static_assert(!WARP || CIN == 1, "the warp is folded into the one-plane stem only");
    constexpr int ROWS = 7 * CIN;                       // tap rows (c, ky)
    constexpr int KS = (ROWS + 1) / 2;                  // k-steps of two tap rows
    constexpr int PW = 21, PP = 22;                     // patch width, row pitch in halfs (a row starts 4-byte aligned)
    constexpr int PLANE = CIN * PW * PP;                // halfs per fp16 plane
    constexpr int WB = KS * 4096;                       // bytes of one piece of the filter bank: [k-step][g 2][n 64] x 8 halfs
    extern __shared__ __attribute__((aligned(16))) char smc[];
    char* const Wp = smc;                               // [piece 2][KS][2][64] x 16 B
    _Float16* const ph = reinterpret_cast<_Float16*>(smc + 2 * WB);      // patch planes: hi, then lo
    float* const smx = reinterpret_cast<float*>(smc + 2 * WB + ((2 * PLANE * 2 + 15) & ~15));      // 4 wave maxima (+ 2 KB: statistics merge)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // ---- filter bank: maximum -> scale -> two fp16 pieces in fragment order ----
    float wmax = 0.f;
    for (int i = tid; i < 64 * 49 * CIN; i += 256) wmax = fmaxf(wmax, fabsf(a.w[i]));
    wmax = wave_max(wmax);
    if (lane == 0) smx[wave] = wmax;
    for (int i = tid; i < 2 * PLANE / 2; i += 256) reinterpret_cast<unsigned*>(ph)[i] = 0u;      // (pad column / unused slots stay zero)
    __syncthreads();
    wmax = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const int kw = bh_f16_scale_exp(__builtin_bit_cast(unsigned, wmax));
    const float sw = __builtin_bit_cast(float, (unsigned)(127 + kw) << 23);
    for (int i = tid; i < KS * 2 * 64; i += 256) {       // slot (s, g, n): eight taps of row r = 2 s + g for output channel n
        const int n = i & 63, g = (i >> 6) & 1, s_ = i >> 7;
        const int r = 2 * s_ + g, c = r / 7, ky = r - c * 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r < ROWS && j < 7) ? a.w[(n * 49 + ky * 7 + j) * CIN + c] : 0.f;
        uint4 hi, lo;
        bh_split8_f16(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sw, hi, lo);
        *reinterpret_cast<uint4*>(Wp + i * 16) = hi;
        *reinterpret_cast<uint4*>(Wp + WB + i * 16) = lo;
    }
    // lane's pixel inside the tile and its fragment addresses
    const int py = wm * 4 + (l31 >> 3), px = l31 & 7;
    const char* const abase = reinterpret_cast<const char*>(ph) + ((2 * py) * PP + 2 * px) * 2;
    const char* const bbase = Wp + (kh2 * 64 + wn * 32 + l31) * 16;
    const float bv = a.bias ? a.bias[wn * 32 + l31] : 0.f;

    double s1 = 0, s2 = 0;
    int cur_grp = -1;
    double* const red = reinterpret_cast<double*>(smx + 4);      // [2 wm][64 n][2] doubles: 2 KB behind the maxima (16-byte aligned)
    auto flush_stats = [&]() {
        __syncthreads();
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (kh2 == 0) { red[(wm * 64 + wn * 32 + l31) * 2] = s1; red[(wm * 64 + wn * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 128 && cur_grp >= 0) {
            const int n = tid >> 1, mom = tid & 1;
            bh_acc_add(&a.bn_sums[bn_sum_index(0, a.groups, cur_grp, 64, n, mom)], red[n * 2 + mom] + red[(64 + n) * 2 + mom], a.det);
        }
        s1 = 0; s2 = 0;
    };
    constexpr int PCH = PW * PW;
    constexpr int NPF = (CIN * PCH + 255) / 256;
    float pf[NPF];
    // WARP: the four gathered source pixels and their weights stay in registers until the top of the next tile (the gathers fly behind the
    // MFMAs and the epilogue of the current one, like the plain loads; no branch around them - a pixel outside the image or past the patch
    // gathers at an out-of-range offset, which a buffer load answers with 0, and carries zero weights)
    float pt[WARP ? NPF : 1][4], pwt[WARP ? NPF : 1][4];
    float pw[WARP ? NPF : 1];                           // the pixel's bilinear weight sum (the warped all-ones mask)
    float* const cw = reinterpret_cast<float*>(red + 256);      // WARP: [16][16] weight sums of the tile's own pixels (1 KB behind `red`)
    const unsigned plane = (unsigned)a.Hi * (unsigned)a.Wi;
    Hf Hnext = {};
    if constexpr (WARP) Hnext = load_h(Hp + (size_t)((int)blockIdx.x < a.ntiles ? (int)blockIdx.x / a.tiles_per_img : 0) * 9);
    auto fetch = [&](int tile_) {
        const int img_ = tile_ / a.tiles_per_img, t_ = tile_ - img_ * a.tiles_per_img;
        const int ty_ = t_ / a.tiles_x, tx_ = t_ - ty_ * a.tiles_x;
        const int iy0 = ty_ * 16 - 3, ix0 = tx_ * 16 - 3;
        if constexpr (WARP) {
            const bool live = tile_ < a.ntiles;
            const int im = live ? img_ : 0;
            const Hf Hm = Hnext;                        // (loaded one tile ahead: below)
            const __amdgpu_buffer_rsrc_t rs = plane_rsrc(a.wsrc + (size_t)im * plane, plane * 4u);
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + j * 256;
                const int yy = i / PW, xx = i - yy * PW;
                const int iy = iy0 + yy, ix = ix0 + xx;
                const bool ok = live && i < PCH && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                const Tap4 tp = make_tap4(Hm, ix, iy, a.Wi, a.Hi);
                float w00, w01, w10, w11, ws_;
                tap_weights(tp, w00, w01, w10, w11, ws_);
                pwt[j][0] = ok ? w00 : 0.f; pwt[j][1] = ok ? w01 : 0.f; pwt[j][2] = ok ? w10 : 0.f; pwt[j][3] = ok ? w11 : 0.f;
                pt[j][0] = ldtap(rs, ok ? tp.o00 : 0xFFFFFFFFu); pt[j][1] = ldtap(rs, ok ? tp.o01 : 0xFFFFFFFFu);
                pt[j][2] = ldtap(rs, ok ? tp.o10 : 0xFFFFFFFFu); pt[j][3] = ldtap(rs, ok ? tp.o11 : 0xFFFFFFFFu);
            }
            // the homography of the tile after this one: nine (wave-uniform) loads whose latency would otherwise sit in front of the taps
            const int nx = tile_ + (int)gridDim.x;
            Hnext = load_h(Hp + (size_t)(nx < a.ntiles ? nx / a.tiles_per_img : 0) * 9);
        } else {
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + j * 256;
                const int c = i / PCH, r = i - c * PCH, yy = r / PW, xx = r - yy * PW;
                const int iy = iy0 + yy, ix = ix0 + xx;
                float v = 0.f;
                if (tile_ < a.ntiles && i < CIN * PCH && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                    v = a.x[(((size_t)img_ * CIN + c) * a.Hi + iy) * a.Wi + ix];
                pf[j] = v;
            }
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int img = tile / a.tiles_per_img, t = tile - img * a.tiles_per_img;
        if (a.bn_sums) {
            const int grp = img / a.imgs_per_group;
            if (grp != cur_grp) { if (cur_grp >= 0) flush_stats(); cur_grp = grp; }
        }
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        if constexpr (WARP) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                pf[j] = tap_blend(pt[j][0], pt[j][1], pt[j][2], pt[j][3], pwt[j][0], pwt[j][1], pwt[j][2], pwt[j][3]);
                pw[j] = tap_wsum(pwt[j][0], pwt[j][1], pwt[j][2], pwt[j][3]);
            }
        }
        // the tile's maximum: wave maxima through LDS behind the barrier that also ends the previous tile's fragment reads
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < NPF; ++j) m = fmaxf(m, fabsf(pf[j]));
        m = wave_max(m);
        __syncthreads();                                 // previous tile's fragment reads (and its maxima reads) are done; Wp is complete
        if (lane == 0) smx[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        const int kx = bh_f16_scale_exp(__builtin_bit_cast(unsigned, m));
        const float sx = __builtin_bit_cast(float, (unsigned)(127 + kx) << 23);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int i = tid + j * 256;
            if (i < CIN * PCH) {
                const int c = i / PCH, r = i - c * PCH, yy = r / PW, xx = r - yy * PW;
                const _Float16 h = (_Float16)(pf[j] * sx);
                const _Float16 l = (_Float16)__builtin_fmaf(pf[j], sx, -(float)h);
                ph[(c * PW + yy) * PP + xx] = h;
                ph[PLANE + (c * PW + yy) * PP + xx] = l;
                if constexpr (WARP) {
                    if ((unsigned)(yy - 3) < 16u && (unsigned)(xx - 3) < 16u) {      // the 16 x 16 pixels this tile owns
                        cw[(yy - 3) * 16 + (xx - 3)] = pw[j];
                        if (a.warped) a.warped[(size_t)img * plane + (unsigned)(ty * 16 + yy - 3) * (unsigned)a.Wi + (tx * 16 + xx - 3)] = pf[j];
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (WARP) {
            if (a.cov && tid < 16) {
                // a 4 x 4 cell in warp_fwd4_kernel's order: rows left to right, then (r0 + r1) + (r2 + r3)
                const int cy = tid >> 2, cx = tid & 3;
                float rr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* q = cw + (4 * cy + r) * 16 + 4 * cx;
                    float cv = 0.0f;
                    cv += q[0]; cv += q[1]; cv += q[2]; cv += q[3];
                    rr[r] = cv;
                }
                a.cov[(size_t)img * (a.Hi / 4) * (a.Wi / 4) + (size_t)(ty * 4 + cy) * (a.Wi / 4) + tx * 4 + cx] =
                    ((rr[0] + rr[1]) + (rr[2] + rr[3])) * (1.0f / 16.0f);
            }
        }
        fetch(tile + (int)gridDim.x);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            // tap rows of the two half-waves (compile-time): r -> (c, ky) -> patch row; a row past the last one reads row 0 (its weights are 0)
            const int r0 = 2 * s_, r1 = 2 * s_ + 1;
            const int o0 = ((r0 / 7) * PW + r0 % 7) * PP * 2;
            const int o1 = r1 < ROWS ? ((r1 / 7) * PW + r1 % 7) * PP * 2 : 0;
            const char* const ap = abase + (kh2 ? o1 : o0);
            uint4 ah, al;
            ah.x = *reinterpret_cast<const unsigned*>(ap);      ah.y = *reinterpret_cast<const unsigned*>(ap + 4);
            ah.z = *reinterpret_cast<const unsigned*>(ap + 8);  ah.w = *reinterpret_cast<const unsigned*>(ap + 12);
            al.x = *reinterpret_cast<const unsigned*>(ap + PLANE * 2);      al.y = *reinterpret_cast<const unsigned*>(ap + PLANE * 2 + 4);
            al.z = *reinterpret_cast<const unsigned*>(ap + PLANE * 2 + 8);  al.w = *reinterpret_cast<const unsigned*>(ap + PLANE * 2 + 12);
            const uint4 bh = *reinterpret_cast<const uint4*>(bbase + s_ * 2048);
            const uint4 bl = *reinterpret_cast<const uint4*>(bbase + WB + s_ * 2048);
            // small products first: lo*hi, hi*lo, hi*hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, al), __builtin_bit_cast(st_f16x8, bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, ah), __builtin_bit_cast(st_f16x8, bl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, ah), __builtin_bit_cast(st_f16x8, bh), acc, 0, 0, 0);
        }
        // (the last products must have left the matrix pipe before their registers are read: spelled out, as in conv3x3_pc_kernel - this
        //  compiler's own padding did not survive every loop structure there)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
        const int kout = -(kx + kw);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mrow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            const int oy = ty * 8 + (mrow >> 3), ox = tx * 8 + (mrow & 7);
            float v = __builtin_ldexpf(acc[r], kout) + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            a.y[(((size_t)img * a.Ho + oy) * a.Wo + ox) * 64 + wn * 32 + l31] = v;
            acc[r] = v;
        }
        if (a.bn_sums) {
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q1 += acc[r]; q2 = __builtin_fmaf(acc[r], acc[r], q2); }
            s1 += (double)q1; s2 += (double)q2;
        }
    }
    if (a.bn_sums && cur_grp >= 0) flush_stats();
}

template <int CIN, bool WARP = false>
static int stem7_f16_launch(const Stem7Args& a, hipStream_t s) {
    constexpr int KS = (7 * CIN + 1) / 2;
    const size_t lds = (size_t)2 * KS * 4096 + ((2 * CIN * 21 * 22 * 2 + 15) & ~15) + 16 + 2048 + (WARP ? 1024 : 0);
    if (bh_query(WARP ? "stem7_fwd_f16_kernel<%d,true>" : "stem7_fwd_f16_kernel<%d>", CIN)) return BH_OK;
    static unsigned long long attr_devs = 0;
    if (bh_device_once(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7_fwd_f16_kernel<CIN, WARP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int per_cu = lds > 40 * 1024 ? (lds > 80 * 1024 ? 1 : 2) : 4;
    int blocks = 256 * per_cu;
    if (blocks > a.ntiles) blocks = a.ntiles;
    hipLaunchKernelGGL((stem7_fwd_f16_kernel<CIN, WARP>), dim3(blocks), dim3(256), lds, s, a, a.H64);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

template <int CIN>
static int stem7_launch(const Stem7Args& a, hipStream_t s) {
    constexpr int KP = (49 * CIN + 3) / 4 * 4;
    const size_t lds = sizeof(float) * (KP * 64 + (CIN * 21 * 21 > 512 ? CIN * 21 * 21 : 512));     // (>= 2 KB behind Wt: the statistics merge)
    if (bh_query("stem7_fwd_kernel<%d>", CIN)) return BH_OK;
    static unsigned long long attr_devs = 0;             // devices on which the dynamic-LDS attribute has been set
    if (bh_device_once(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7_fwd_kernel<CIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int per_cu = lds > 40 * 1024 ? (lds > 80 * 1024 ? 1 : 2) : 4;
    int blocks = 256 * per_cu;
    if (blocks > a.ntiles) blocks = a.ntiles;
    hipLaunchKernelGGL((stem7_fwd_kernel<CIN>), dim3(blocks), dim3(256), lds, s, a);
    BH_LAUNCH_CHECK();
    return BH_OK;
}


// *taken = 1 when the shape is a stem this kernel takes and the launch was made
int bh_stem7_try(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, int relu,
                 hipStream_t stream, int* taken, double* bn_sums, int groups) {
    *taken = 0;
    if ((d->route & BH_ROUTE_NO_STEM7) || d->transposed || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->Co != 64 ||
        d->out_nchw)
        return BH_OK;
    if (!(d->Ci == 1 || ((d->Ci == 2 || d->Ci == 3 || d->Ci == 6) && d->in_nchw))) return BH_OK;
    if (d->Ho % 8 || d->Wo % 8 || d->Ho * 2 != d->Hi || d->Wo * 2 != d->Wi) return BH_OK;
    Stem7Args a = {};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.relu = relu;
    a.bn_sums = bn_sums; a.groups = groups > 0 ? groups : 1; a.imgs_per_group = d->N / a.groups; a.det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0;
    a.N = d->N; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo;
    a.tiles_x = d->Wo / 8; a.tiles_per_img = (d->Ho / 8) * a.tiles_x; a.ntiles = d->N * a.tiles_per_img;
    if (a.ntiles < 256) return BH_OK;
    int rc;
    // precision 4 (the fp16-piece arithmetic of the 3x3 layers; round 6): the same tiles on v_mfma_f32_32x32x16_f16 - one and two planes
    // (the extractor's and the backbone's stems; the RGB stems keep the fp32-input form: their filter bank in pieces would take one workgroup per CU)
    const bool f16 = d->precision == 4 && d->Ci <= 2 && g_stem_f16;      // (deterministic calls too: the same arithmetic in both modes)
    switch (d->Ci) {
        case 1: rc = f16 ? stem7_f16_launch<1>(a, stream) : stem7_launch<1>(a, stream); break;
        case 2: rc = f16 ? stem7_f16_launch<2>(a, stream) : stem7_launch<2>(a, stream); break;
        case 3: rc = stem7_launch<3>(a, stream); break;
        default: rc = stem7_launch<6>(a, stream); break;
    }
    if (rc) return rc;
    *taken = 1;
    return BH_OK;
}

// The one-plane stem on the homography warp of src (stem7_fwd_f16_kernel<1, true> above): y = conv(warp(src, H)), the warped image and the
// pooled coverage written on the way (either may be NULL).  BH_E_UNSUPPORTED unless the fp16-piece stem applies and pool == 4.
extern "C" int bh_stem7_fwd_warp(const float* src, const double* H64, int pool, const float* w, const float* bias, float* y,
                                 const bh_conv_desc* d, float* warped, float* cov, double* bn_sums, int groups, void* stream) {
    if (!src || !H64 || !w || !y || !d) return BH_E_BADARG;
    if ((d->route & BH_ROUTE_NO_STEM7) || d->transposed || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->Co != 64 ||
        d->out_nchw || d->Ci != 1 || d->Ho % 8 || d->Wo % 8 || d->Ho * 2 != d->Hi || d->Wo * 2 != d->Wi)
        return BH_E_UNSUPPORTED;
    if (d->precision != 4 || !g_stem_f16 || pool != 4 || (long long)d->Hi * d->Wi * 4 >= (1ll << 31)) return BH_E_UNSUPPORTED;
    if (bn_sums && (groups < 1 || d->N % groups)) return BH_E_BADARG;
    Stem7Args a = {};
    a.x = nullptr; a.w = w; a.bias = bias; a.y = y; a.relu = 0;
    a.bn_sums = bn_sums; a.groups = groups > 0 ? groups : 1; a.imgs_per_group = d->N / a.groups; a.det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0;
    a.N = d->N; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo;
    a.tiles_x = d->Wo / 8; a.tiles_per_img = (d->Ho / 8) * a.tiles_x; a.ntiles = d->N * a.tiles_per_img;
    a.wsrc = src; a.H64 = H64; a.warped = warped; a.cov = cov;
    if (a.ntiles < 256) return BH_E_UNSUPPORTED;
    return stem7_f16_launch<1, true>(a, bh_stream(stream));
}

// ---------------------------------------------------------------------------------------------
// dgrad of the ONE-input-channel 7x7 / 2 stem (the frozen extractor's conv1 on a warped patch, PerceptualHead.py:52-55,377,398: the
// gradient has to reach the homography through the image) in one kernel (round 4).  It used to be a 1x1 GEMM gy x w^T into a per-pixel tap
// table (109 MB written and read back) + a col2im pass: 73 + 129 us.  Here a workgroup OWNS a 16 x 16 tile of the image: it stages the
// 11 x 11 gy pixels that reach it (x 64 channels) in LDS, multiplies them with the 64 x 49 filter bank (v_mfma_f32_32x32x2_f32: wave m =
// 32 of the 121 -> 128 pixels x 2 x 32 taps, K = 64 channels), leaves the 121 x 49 tap table in LDS (over the staged gy) and every thread
// sums the <= 16 taps of its image pixel.  No atomics (a first form that scattered 8 x 8 gy tiles into the image with float atomics spent
// its time in 3.6 M device-scope atomics: 182 us), 1.9x the GEMM work of the two-pass form and none of its 218 MB.  The same sums in another order
// than the two-pass form; deterministic by construction.  gy [N][Ho][Wo][64] NHWC, w [64][7][7][1], gx [N][1][2 Ho][2 Wo].
// ---------------------------------------------------------------------------------------------
// WARP (round 6; round-3/4/5 VERDICT: "the warp folded into the extractor stem's dgrad epilogue"): the image whose gradient this is was
// the homography warp of a source patch (src/data/utils.py:54-59 via PerceptualHead.py:371-401), and that gradient has ONE consumer - the
// warp's adjoint with respect to H.  The thread that has just summed the gradient of image pixel (iy, ix) applies it on the spot: the
// pixel's tap (warp_tap.h, the arithmetic of warp_bwd4_kernel), four gathered source pixels, the coverage term of the pooled all-ones
// mask, the nine sums of dL/dH in double - kept in registers while the workgroup's tiles stay in one image (the tiles are walked in image
// order here), reduced over the workgroup and added with nine f64 atomics when the image changes.  The 8.4 MB gradient is neither written
// nor read back, warp_bwd4_kernel's launch (~19 us for 17 MB: VALU- and latency-bound) disappears into the shadow of this kernel's MFMAs.
// gx may be NULL (the fused path), or is written as before (tests).
struct Stem7WarpArgs {
    const float* src;      // [N][1][Hi][Wi] source patches
    const double* H64;     // [N][9]
    const float* g_cov;    // [N][Hi / pool][Wi / pool] gradient of the pooled coverage, or NULL
    double* gH;            // [N][9], accumulated (+=)
    int pool_shift;        // log2(pool)
    float cov_scale;       // 1 / pool^2
    int wpi;               // workgroups per image (the grid is N * wpi)
};

// F16 (round 6; bh_conv_desc.precision = 4): the window GEMM in the fp16-piece arithmetic of the 3x3 layers - gy tile and filter bank as two
// fp16 pieces each (the tile's scale from ITS maximum, the bank's from the bank's: no magnitude record needed), three
// v_mfma_f32_32x32x16_f16 products per product: 24 MFMAs of 32 cycles per wave and tile where the fp32-input form issues 64 of 64.  The
// staged pieces are [pixel][64 channels + 8] halfs (a lane's eight consecutive k are one ds_read_b128), the bank [k-step][k half][tap] x 8 halfs.
template <bool WARP, bool F16 = false>
__global__ void __launch_bounds__(256, 3) stem7_dgrad_c1_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                             int Ho, int Wo, int tiles_x, int tiles_per_img, int ntiles, Stem7WarpArgs wa) {
    constexpr int GP = 68;                               // row pitch of the gy tile in LDS (floats): 16-byte rows for the staging stores
    constexpr int TP = 53;                               // row pitch of the tap table (49 used)
    constexpr int HP = 72;                               // F16: row pitch of a piece of the gy tile (halfs)
    __shared__ float Wt[64 * 64];                        // [k = channel][n = tap, 49 used]; F16: two pieces x [4 k-steps][2][64 taps] x 16 B
    // [gy pixel of the 11 x 11 window, 121 used][channel] (F16: two pieces of [128][HP] halfs = 36,864 B); then the tap table [121][TP]
    __shared__ __attribute__((aligned(16))) float gt[F16 ? (2 * 128 * HP * 2) / 4 : 128 * GP];
    __shared__ float smx[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh2 = lane >> 5;
    int kw = 0;
    if constexpr (!F16) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int k = i >> 6, n = i & 63;
            Wt[i] = n < 49 ? w[k * 49 + n] : 0.f;
        }
    } else {
        float wmax = 0.f;
        for (int i = tid; i < 64 * 49; i += 256) wmax = fmaxf(wmax, fabsf(w[i]));
        wmax = wave_max(wmax);
        if (lane == 0) smx[wave] = wmax;
        __syncthreads();
        wmax = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        kw = bh_f16_scale_exp(__builtin_bit_cast(unsigned, wmax));
        const float sw = __builtin_bit_cast(float, (unsigned)(127 + kw) << 23);
        // slot (s, g, n): channels k = 16 s + 8 g .. + 7 of tap n (w[k][7][7][1]: w[k * 49 + n]); taps 49 .. 63: zeros
        for (int i = tid; i < 4 * 2 * 64; i += 256) {
            const int n = i & 63, g = (i >> 6) & 1, s_ = i >> 7;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = n < 49 ? w[(16 * s_ + 8 * g + j) * 49 + n] : 0.f;
            uint4 hi, lo;
            bh_split8_f16(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sw, hi, lo);
            reinterpret_cast<uint4*>(Wt)[i] = hi;
            reinterpret_cast<uint4*>(Wt)[512 + i] = lo;
        }
    }
    const int Hi = 2 * Ho, Wi = 2 * Wo;
    const int py = tid >> 4, px = tid & 15;              // this thread's image pixel inside the tile
    // WARP: a contiguous range of tiles per workgroup (image order: the nine sums live in registers across the tiles of one image)
    // WARP: wa.wpi workgroups per image, workgroup j of an image takes its tiles j, j + wpi, ... - the nine sums stay in registers for the
    // workgroup's whole life (ONE reduction and nine atomics per workgroup), and the workgroups that run together work on neighbouring
    // tiles of the same images, so the 11 x 11 windows' overlap is served by the L2 as in the plain kernel's strided walk
    const int w_img = WARP ? (int)blockIdx.x / wa.wpi : 0, w_j = WARP ? (int)blockIdx.x - w_img * wa.wpi : 0;
    const int t_lo = WARP ? w_img * tiles_per_img + w_j : (int)blockIdx.x;
    const int t_hi = WARP ? (w_img + 1) * tiles_per_img : ntiles;
    const int t_step = WARP ? wa.wpi : (int)gridDim.x;
    __shared__ double wpart[4][9];
    // (Round 6 also tried the nine per-thread sums in float - a thread sees <= ~11 pixels - with the double starting at the wave reduction:
    //  no faster, and with the fp16-piece window GEMM in the same kernel the sums came out wrong in a few images per launch, differently
    //  from run to run (tests/test_head_kernels_gpu.py's multi-tile cases; not root-caused - the double form below is bit-stable).)
    double ws[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) ws[k] = 0.0;
    int cur_img = -1;
    Hf Hc = {};
    // the workgroup's sums for image `im` leave: wave sums -> LDS -> nine f64 atomics (uniform call: every thread of the workgroup)
    auto warp_flush = [&](int im) {
#pragma unroll
        for (int k = 0; k < 9; ++k) ws[k] = wave_sum(ws[k]);
        if (lane == 0)
            for (int k = 0; k < 9; ++k) wpart[wave][k] = ws[k];
        __syncthreads();
        if (tid < 9) atomicAdd(wa.gH + (size_t)im * 9 + tid, ((wpart[0][tid] + wpart[1][tid]) + wpart[2][tid]) + wpart[3][tid]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) ws[k] = 0.0;
    };
    for (int tile = t_lo; tile < t_hi; tile += t_step) {
        const int img = tile / tiles_per_img, t = tile - img * tiles_per_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int oy0 = ty * 8 - 1, ox0 = tx * 8 - 1;    // first gy pixel of the window: image rows 16 ty .. 16 ty + 15 see oy0 .. oy0 + 10
        // WARP: this thread's pixel, its tap and the four source pixels are requested early - they depend on the homography only, so the gathers'
        // latency (and the tap arithmetic) sits under the staging and the MFMAs below instead of behind the tile's last barrier - but BEHIND the
        // window's own loads (F16): the vector-memory counter is in order, and gathers requested first made every tile wait for them
        Tap4 tp = {};
        float wp00 = 0.f, wp01 = 0.f, wp10 = 0.f, wp11 = 0.f, wgc = 0.f;
        auto warp_top = [&]() {
            if (img != cur_img) {
                if (cur_img >= 0) warp_flush(cur_img);
                cur_img = img;
                Hc = load_h(wa.H64 + (size_t)img * 9);
            }
            const int iy_ = ty * 16 + py, ix_ = tx * 16 + px;
            tp = make_tap4(Hc, ix_, iy_, 2 * Wo, 2 * Ho);
            const unsigned plane = (unsigned)(4 * Ho) * (unsigned)Wo;
            const __amdgpu_buffer_rsrc_t rs = plane_rsrc(wa.src + (size_t)img * plane, plane * 4u);
            wp00 = ldtap(rs, tp.o00); wp01 = ldtap(rs, tp.o01); wp10 = ldtap(rs, tp.o10); wp11 = ldtap(rs, tp.o11);
            if (wa.g_cov)
                wgc = wa.g_cov[((size_t)img * ((2 * Ho) >> wa.pool_shift) + (iy_ >> wa.pool_shift)) * ((2 * Wo) >> wa.pool_shift) + (ix_ >> wa.pool_shift)] * wa.cov_scale;
        };
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        if constexpr (WARP && !F16) warp_top();
        if constexpr (!F16) {
        __syncthreads();                                 // the previous tile's tap table has been read (and Wt is complete)
        // 128 window slots x 16 float4: thread -> (slot, 4 channels); slots past 121 and pixels outside gy are zero
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = j * 256 + tid, slot = q >> 4, c4 = (q & 15) * 4;
            const int wy = slot / 11, wx = slot - wy * 11;
            const int oy = oy0 + wy, ox = ox0 + wx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < 121 && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo)
                v = *reinterpret_cast<const float4*>(gy + (((size_t)img * Ho + oy) * Wo + ox) * 64 + c4);
            *reinterpret_cast<float4*>(gt + slot * GP + c4) = v;
        }
        __syncthreads();
        const float* ap = gt + (wave * 32 + l31) * GP + kh2;
        const float* bp = Wt + kh2 * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float av = ap[2 * kk];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * kk * 64], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * kk * 64 + 32], acc1, 0, 0, 0);
        }
        } else {
        // the window into registers, its maximum over the workgroup, then the two fp16 pieces of (value x 2^kg) into LDS
        float4 sv[8];
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = j * 256 + tid, slot = q >> 4, c4 = (q & 15) * 4;
            const int wy = slot / 11, wx = slot - wy * 11;
            const int oy = oy0 + wy, ox = ox0 + wx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < 121 && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo)
                v = *reinterpret_cast<const float4*>(gy + (((size_t)img * Ho + oy) * Wo + ox) * 64 + c4);
            sv[j] = v;
        }
        if constexpr (WARP) warp_top();                   // (behind the window's loads in the in-order vector-memory queue)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(sv[j].x), fabsf(sv[j].y))), fmaxf(fabsf(sv[j].z), fabsf(sv[j].w)));
        m = wave_max(m);
        __syncthreads();                                 // the previous tile's tap table (and its maxima) have been read; the bank is complete
        if (lane == 0) smx[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        const int kg = bh_f16_scale_exp(__builtin_bit_cast(unsigned, m));
        const float sg = __builtin_bit_cast(float, (unsigned)(127 + kg) << 23);
        char* const gh = reinterpret_cast<char*>(gt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = j * 256 + tid, slot = q >> 4, c4 = (q & 15) * 4;
            uint2 hi, lo;
            bh_split2_pair_f16(sv[j].x, sv[j].y, sg, hi.x, lo.x);
            bh_split2_pair_f16(sv[j].z, sv[j].w, sg, hi.y, lo.y);
            *reinterpret_cast<uint2*>(gh + (slot * HP + c4) * 2) = hi;
            *reinterpret_cast<uint2*>(gh + 128 * HP * 2 + (slot * HP + c4) * 2) = lo;
        }
        __syncthreads();
        const char* const ap = gh + ((wave * 32 + l31) * HP + 8 * kh2) * 2;
        const char* const bp = reinterpret_cast<const char*>(Wt) + (kh2 * 64 + l31) * 16;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const uint4 ah = *reinterpret_cast<const uint4*>(ap + s_ * 32), al = *reinterpret_cast<const uint4*>(ap + 128 * HP * 2 + s_ * 32);
            const uint4 b0h = *reinterpret_cast<const uint4*>(bp + s_ * 2048), b0l = *reinterpret_cast<const uint4*>(bp + 8192 + s_ * 2048);
            const uint4 b1h = *reinterpret_cast<const uint4*>(bp + s_ * 2048 + 512), b1l = *reinterpret_cast<const uint4*>(bp + 8192 + s_ * 2048 + 512);
#define ST_MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, A_), __builtin_bit_cast(st_f16x8, B_), C_, 0, 0, 0)
            ST_MF(al, b0h, acc0); ST_MF(al, b1h, acc1);          // small products first, the two accumulators alternating
            ST_MF(ah, b0l, acc0); ST_MF(ah, b1l, acc1);
            ST_MF(ah, b0h, acc0); ST_MF(ah, b1h, acc1);
#undef ST_MF
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc0), "+v"(acc1));      // (as above: the products have left the matrix pipe)
        const int kout = -(kg + kw);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = __builtin_ldexpf(acc0[r], kout); acc1[r] = __builtin_ldexpf(acc1[r], kout); }
        }
        __syncthreads();                                 // every wave has read its gy rows: the table may overwrite them
        // C/D layout: col = lane&31 (tap), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (slot of the wave's 32)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            gt[slot * TP + l31] = acc0[r];
            if (l31 < 17) gt[slot * TP + 32 + l31] = acc1[r];
        }
        __syncthreads();
        // image pixel (iy, ix): taps (ky, kx) with iy + 3 - ky even, gy pixel ((iy + 3 - ky) / 2, (ix + 3 - kx) / 2)
        const int iy = ty * 16 + py, ix = tx * 16 + px;
        const int ky0 = (iy + 3) & 1, kx0 = (ix + 3) & 1;
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ky = ky0 + 2 * a;
            const int wy = ((iy + 3 - ky) >> 1) - oy0;   // (inside the window by construction; rows outside gy hold zeros)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int kx = kx0 + 2 * b;
                const int wx = ((ix + 3 - kx) >> 1) - ox0;
                if (ky < 7 && kx < 7) sum += gt[(wy * 11 + wx) * TP + ky * 7 + kx];
            }
        }
        if (!WARP || gx) gx[((size_t)img * Hi + iy) * Wi + ix] = sum;
        if constexpr (WARP) {
            // warp_bwd4_kernel's per-pixel arithmetic (csrc/warp.hip): float products, double sums
            const float sy = tp.wy0 + tp.wy1, sx = tp.wx0 + tp.wx1;
            float gu = wgc * ((tp.vx1 ? sy : 0.0f) - (tp.vx0 ? sy : 0.0f));
            float gv = wgc * ((tp.vy1 ? sx : 0.0f) - (tp.vy0 ? sx : 0.0f));
            gu += sum * ((wp01 - wp00) * tp.wy0 + (wp11 - wp10) * tp.wy1);
            gv += sum * ((wp10 - wp00) * tp.wx0 + (wp11 - wp01) * tp.wx1);
            const float a = gu * tp.iz, bq = gv * tp.iz;
            const float gz = tp.guard ? 0.0f : -(gu * tp.u + gv * tp.v) * tp.iz;
            const double fx = (double)ix, fy = (double)iy;
            ws[0] += (double)a * fx; ws[1] += (double)a * fy; ws[2] += (double)a;
            ws[3] += (double)bq * fx; ws[4] += (double)bq * fy; ws[5] += (double)bq;
            ws[6] += (double)gz * fx; ws[7] += (double)gz * fy; ws[8] += (double)gz;
        }
    }
    if constexpr (WARP) {
        if (cur_img >= 0) warp_flush(cur_img);
    }
}

extern "C" int bh_stem7_dgrad_c1(const float* gy, const float* w, float* gx, const bh_conv_desc* d, void* stream) {
    if (!gy || !w || !gx || !d) return BH_E_BADARG;
    if (d->transposed || d->Ci != 1 || d->Co != 64 || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->out_nchw ||
        d->Ho % 8 || d->Wo % 8 || d->Ho * 2 != d->Hi || d->Wo * 2 != d->Wi)
        return BH_E_UNSUPPORTED;
    if (d->N == 0) return BH_OK;
    if (bh_query((d->precision == 4 && g_stem_f16) ? "stem7_dgrad_c1_kernel<false,true>" : "stem7_dgrad_c1_kernel")) return BH_OK;
    hipStream_t s = bh_stream(stream);
    const int tiles_x = d->Wo / 8, tpi = (d->Ho / 8) * tiles_x, ntiles = d->N * tpi;
    int blocks = 256 * 3;
    if (blocks > ntiles) blocks = ntiles;
    if (d->precision == 4 && g_stem_f16)       // (round 6: the fp16-piece arithmetic of the 3x3 layers)
        hipLaunchKernelGGL((stem7_dgrad_c1_kernel<false, true>), dim3(blocks), dim3(256), 0, s, gy, w, gx, d->Ho, d->Wo, tiles_x, tpi, ntiles, Stem7WarpArgs{});
    else
        hipLaunchKernelGGL((stem7_dgrad_c1_kernel<false, false>), dim3(blocks), dim3(256), 0, s, gy, w, gx, d->Ho, d->Wo, tiles_x, tpi, ntiles, Stem7WarpArgs{});
    BH_LAUNCH_CHECK();
    return BH_OK;
}

// The same dgrad with the warp's adjoint applied to the gradient it makes (stem7_dgrad_c1_kernel<true>): gH[N][9] += dL/dH of
//   warped = warp(src, H)  (bh_warp_fwd, pool-averaged coverage included when g_cov is given)
// for the gradient gx = dgrad(gy) of `warped` - what bh_stem7_dgrad_c1 followed by bh_warp_bwd(src, H64, gx, g_cov, ...) computes, without
// the gradient image.  gx: NULL, or [N][Hi][Wi] written as by bh_stem7_dgrad_c1.  The f64 atomics make the last bits of gH depend on the
// order of the workgroups: deterministic callers use the two separate calls (bh_warp_bwd_f with BH_F_DETERMINISTIC).
extern "C" int bh_stem7_dgrad_c1_warp(const float* gy, const float* w, float* gx, const bh_conv_desc* d, const float* src, const double* H64,
                                      const float* g_cov, int pool, double* gH, void* stream) {
    if (!gy || !w || !d || !src || !H64 || !gH) return BH_E_BADARG;
    if (d->transposed || d->Ci != 1 || d->Co != 64 || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->out_nchw ||
        d->Ho % 8 || d->Wo % 8 || d->Ho * 2 != d->Hi || d->Wo * 2 != d->Wi)
        return BH_E_UNSUPPORTED;
    int shift = -1;
    for (int b = 0; b < 6; ++b) if (pool == (1 << b)) shift = b;
    if (shift < 0 || d->Hi % pool || d->Wi % pool || (long long)d->Hi * d->Wi * 4 >= (1ll << 31)) return BH_E_UNSUPPORTED;
    if (d->N == 0) return BH_OK;
    if (bh_query((d->precision == 4 && g_stem_f16) ? "stem7_dgrad_c1_kernel<true,true>" : "stem7_dgrad_c1_kernel<true>")) return BH_OK;
    hipStream_t s = bh_stream(stream);
    const int tiles_x = d->Wo / 8, tpi = (d->Ho / 8) * tiles_x, ntiles = d->N * tpi;
    // three workgroups per CU as the plain kernel, as whole workgroups per image (>= 1, <= one per tile)
    int wpi = (256 * 3) / d->N;
    if (wpi < 1) wpi = 1;
    if (wpi > tpi) wpi = tpi;
    const int blocks = d->N * wpi;
    Stem7WarpArgs wa = {src, H64, g_cov, gH, shift, 1.0f / (float)(pool * pool), wpi};
    if (d->precision == 4 && g_stem_f16)
        hipLaunchKernelGGL((stem7_dgrad_c1_kernel<true, true>), dim3(blocks), dim3(256), 0, s, gy, w, gx, d->Ho, d->Wo, tiles_x, tpi, ntiles, wa);
    else
        hipLaunchKernelGGL((stem7_dgrad_c1_kernel<true, false>), dim3(blocks), dim3(256), 0, s, gy, w, gx, d->Ho, d->Wo, tiles_x, tpi, ntiles, wa);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the 7x7 / 2 stem with CIN stacked image planes (the backbone's first conv: Rethinking.py:31, ResNet34.py:17) in a
// dedicated kernel (round 4; the generic split-K kernel ran this K = 49 CIN gather at 37 TFLOP/s: 176 us).  gW[co][tap] = sum over pixels
// of gy[pixel][co] * x[patch of the pixel][tap]: a persistent workgroup walks 8 x 8 tiles of gy, stages the tile ([64 pixels][64 channels])
// and its 21 x 21 input patch per plane in LDS and accumulates D[co][n = plane * 49 + tap] over the 64 pixels of the tile
// (v_mfma_f32_32x32x2_f32, k = pixel: the A fragment is gy[pixel][co] - one ds_read_b32 with a literal offset, the B fragment
// patch[tap offset of the lane + literal pixel offset], as in stem7_fwd_kernel); wave w owns taps [32 w, 32 w + 32) x all 64 co, the
// accumulators live across the workgroup's tiles.  Partial results go to a workspace [workgroups][64][128] and a second kernel adds them in
// workgroup order: no atomics, bitwise repeatable.  x [N][CIN][Hi][Wi], gy [N][Ho][Wo][64], gw [64][7][7][CIN] +=.
// ---------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) stem7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                          int Hi, int Wi, int Ho, int Wo, int tiles_x, int tiles_per_img, int ntiles) {
    static_assert(49 * CIN <= 128, "taps of all planes in 128 columns");
    constexpr int GP = 68, PW = 21, PCH = PW * PW;
    __shared__ __attribute__((aligned(16))) float gt[64 * GP];       // [pixel][co]
    __shared__ float patch[CIN * PCH + 64];                           // (+64: lanes past the last tap read in bounds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int n = wave * 32 + l31;
    const int tap_off = n < 49 * CIN ? (n / 49) * PCH + ((n % 49) / 7) * PW + (n % 49) % 7 : 0;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int i = tid; i < 64; i += 256) patch[CIN * PCH + i] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tiles_per_img, t = tile - img * tiles_per_img;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int iy0 = ty * 16 - 3, ix0 = tx * 16 - 3;
        __syncthreads();                                 // the previous tile's fragment reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = j * 256 + tid, pix = q >> 4, c4 = (q & 15) * 4;
            const int oy = ty * 8 + (pix >> 3), ox = tx * 8 + (pix & 7);
            *reinterpret_cast<float4*>(gt + pix * GP + c4) = *reinterpret_cast<const float4*>(gy + (((size_t)img * Ho + oy) * Wo + ox) * 64 + c4);
        }
        for (int i = tid; i < CIN * PCH; i += 256) {
            const int c = i / PCH, r = i - c * PCH, yy = r / PW, xx = r - yy * PW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) v = x[(((size_t)img * CIN + c) * Hi + iy) * Wi + ix];
            patch[i] = v;
        }
        __syncthreads();
        const float* ap = gt + kh2 * GP + l31;
        const float* bp = patch + tap_off + 2 * kh2;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            // pixel p = 2 kk + kh2 of the tile: (py, px) = (kk >> 2, ((2 kk) & 7) + kh2) -> patch offset 2 py * 21 + 2 px
            const float bv = bp[42 * (kk >> 2) + 2 * ((2 * kk) & 7)];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * GP], bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * GP + 32], bv, acc1, 0, 0, 0);
        }
    }
    // C/D layout: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co of the 32-block)
    float* out = ws + (size_t)blockIdx.x * 64 * 128;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * kh2;
        out[co * 128 + n] = acc0[r];
        out[(co + 32) * 128 + n] = acc1[r];
    }
}

template <int CIN>
__global__ void __launch_bounds__(256) stem7_wgrad_reduce_kernel(const float* __restrict__ ws, int nwg, float* __restrict__ gw) {
    // block = 16 outputs x 16 slices of the workgroup list: a thread adds every 16th partial (independent loads in flight), the slices
    // meet in LDS in slice order
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;                   // (co, n)
    const bool ok = i < 64 * 49 * CIN;
    const int co = ok ? i / (49 * CIN) : 0, n = ok ? i - co * (49 * CIN) : 0;
    float s = 0.f;
    if (ok)
        for (int g = sl; g < nwg; g += 16) s += ws[((size_t)g * 64 + co) * 128 + n];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][o];
        const int c = n / 49, tp = n - c * 49;
        gw[((size_t)co * 49 + tp) * CIN + c] += t;
    }
}

static bool stem7_wgrad_ok(const bh_conv_desc* d) {
    return d && !d->transposed && d->kh == 7 && d->kw == 7 && d->stride == 2 && d->pad == 3 && d->Co == 64 && d->Ci == 2 && d->in_nchw &&
           !d->out_nchw && d->Ho % 8 == 0 && d->Wo % 8 == 0 && d->Ho * 2 == d->Hi && d->Wo * 2 == d->Wi && d->N * (d->Ho / 8) * (d->Wo / 8) >= 512;
}
static int stem7_wgrad_blocks(const bh_conv_desc* d) {
    const int ntiles = d->N * (d->Ho / 8) * (d->Wo / 8);
    return ntiles < 1024 ? ntiles : 1024;
}

extern "C" size_t bh_stem7_wgrad_ws_bytes(const bh_conv_desc* d) {
    return stem7_wgrad_ok(d) ? (size_t)stem7_wgrad_blocks(d) * 64 * 128 * sizeof(float) : 0;
}

extern "C" int bh_stem7_wgrad(const float* x, const float* gy, float* gw, const bh_conv_desc* d, float* ws, size_t ws_bytes, void* stream) {
    if (!x || !gy || !gw || !d || !ws) return BH_E_BADARG;
    if (!stem7_wgrad_ok(d)) return BH_E_UNSUPPORTED;
    if (ws_bytes < bh_stem7_wgrad_ws_bytes(d)) return BH_E_BADARG;
    if (bh_query("stem7_wgrad_kernel<2>+stem7_wgrad_reduce_kernel<2>")) return BH_OK;
    hipStream_t s = bh_stream(stream);
    const int tiles_x = d->Wo / 8, tpi = (d->Ho / 8) * tiles_x, ntiles = d->N * tpi, blocks = stem7_wgrad_blocks(d);
    hipLaunchKernelGGL(stem7_wgrad_kernel<2>, dim3(blocks), dim3(256), 0, s, x, gy, ws, d->Hi, d->Wi, d->Ho, d->Wo, tiles_x, tpi, ntiles);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem7_wgrad_reduce_kernel<2>, dim3((64 * 98 + 15) / 16), dim3(256), 0, s, ws, blocks, gw);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

