// Streaming kernel for the POINTWISE layers of the decoder (round 5; round-4 VERDICT item 4): 1x1 / stride 1 convolutions and 2x2 / stride 2
// transposed convolutions with few input channels (K = Ci = 32 or 64) on the large maps - ResNet50DeconvBlock's upper 1x1 and both
// ConvTranspose2d (src/backbones/utils.py:60-82), Rethinking.py:144-147.  As a GEMM that is  y[pixel][n] = sum_k x[pixel][k] W[k][n]  with
// n = tap * Co + co (4 taps for the transposed conv, whose "columns" scatter to the 2x2 children of the pixel): 0.3 - 0.4 GB of traffic
// against 2 - 4 GFLOP.  The generic implicit-GEMM kernel ran these at 1 - 3 TB/s: 2k - 16k short-lived workgroups whose whole k loop is
// one or two iterations, LDS staging of an operand nobody shares, and - with BatchNorm statistics - one f64 atomic per workgroup and
// (channel, moment) on the SAME address (2048 of them cost 60 us).
//
// Here the whole weight panel of a wave lives in REGISTERS for the life of the wave (K x 32 NT floats, K/2 x NT per lane), the workgroups
// are persistent (<= 512: two per CU) and own a pixel range of ONE statistics group, and a wave needs no LDS and no barrier while it
// streams: per 32-pixel tile it loads K / 8 float4 per lane straight into the A layout of v_mfma_f32_32x32x2_f32 (lane = pixel row,
// half-wave = k parity class; the contraction order is free as long as A and B agree: step (j, e) takes k = 8 j + 4 kh2 + e), optionally
// applies the BatchNorm of its producer (BatchNorm-on-load: coefficients of the lane's channels in registers), runs K x NT / 2 MFMAs and
// writes 128-byte row segments; the next tile's loads are in flight meanwhile.  fp32-input MFMA: exactly the arithmetic of the generic
// kernel (sums in another order).  Statistics / magnitude record: per lane over the wave's life, one LDS round and ONE atomic per
// (channel, moment) and workgroup at the end (the taps of a channel are added in LDS first).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwArgs {
    const float* X;        // [M][K]
    const float* W;        // transposed conv: [K][N] (= Wt[ci][tap][co]); 1x1 conv: [N][K]
    const float* bias;     // [Co] or NULL
    float* Y;
    int M, N, Co, taps;    // N = taps * Co
    int Hs, Ws;            // source grid (= output grid for the 1x1 conv)
    int hw_shift, w_shift; // log2(Hs * Ws), log2(Ws) or -1
    int w_nk;              // 1: W[n][k]
    int slices;            // column slices of 32 NT columns (grid: slice fastest)
    int wg_per_group, groups, tiles_per_group;
    unsigned x_bytes, y_bytes;
    const float* bni;      // [groups][K] x (scale, shift) or NULL
    int bni_relu;
    double* bn_sums;       // or NULL
    int bn_det;
    unsigned* amax_out;    // or NULL
};

template <int K, int NT, bool BNI>
__global__ void __launch_bounds__(256, 2) pw_kernel(PwArgs a) {
    constexpr int KJ = K / 8;
    __shared__ double red[4 * NT * 32 * 2];
    __shared__ float sm_amax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh2 = lane >> 5;
    const int b = blockIdx.x;
    const int slice = b % a.slices, wi = (b / a.slices) % a.wg_per_group, grp = b / (a.slices * a.wg_per_group);
    const int n_base = slice * 32 * NT;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, a.y_bytes, 0x00020000);

    // ---- the wave's weight panel: bw[j][e][nt] = W(k = 8 j + 4 kh2 + e, n = n_base + 32 nt + l31) ----
    float bw[KJ][4][NT];
    float bv[NT];
    unsigned cp[NT];                                    // column part of the output byte offset (out of range: dropped)
    int cco[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + 32 * nt + l31;
        nok[nt] = n < a.N;
        const int nn = nok[nt] ? n : 0;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * j + 4 * kh2 + e;
                const float w = a.w_nk ? a.W[(size_t)nn * K + k] : a.W[(size_t)k * a.N + nn];
                bw[j][e][nt] = nok[nt] ? w : 0.f;
            }
        const int tap = nn / a.Co, co = nn - tap * a.Co;
        cco[nt] = co;
        bv[nt] = (a.bias && nok[nt]) ? a.bias[co] : 0.f;
        if (a.taps == 1) cp[nt] = (unsigned)nn * 4u;
        else cp[nt] = ((unsigned)((tap >> 1) * (2 * a.Ws) + (tap & 1)) * (unsigned)a.Co + (unsigned)co) * 4u;
    }
    // BatchNorm-on-load coefficients of the lane's channels (the workgroup's pixels lie in one group)
    float4 bsc[BNI ? KJ : 1], bsh[BNI ? KJ : 1];
    if constexpr (BNI) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const float4* tb = reinterpret_cast<const float4*>(a.bni + ((size_t)grp * K + 8 * j + 4 * kh2) * 2);
            const float4 t0 = tb[0], t1 = tb[1];        // (sc0, sh0, sc1, sh1), (sc2, sh2, sc3, sh3)
            bsc[j] = make_float4(t0.x, t0.z, t1.x, t1.z);
            bsh[j] = make_float4(t0.y, t0.w, t1.y, t1.w);
        }
    }
    const float lo = a.bni_relu ? 0.0f : -__builtin_inff();

    double st1[NT], st2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { st1[nt] = 0.0; st2[nt] = 0.0; }
    float vmax = 0.f;

    // tiles of this wave: group grp, tile index (wi * 4 + wave) + s * (wg_per_group * 4)
    const int tstride = a.wg_per_group * 4;
    int t = wi * 4 + wave;
    const int tile0 = grp * a.tiles_per_group;
    auto issue = [&](int tt, float4 (&av)[KJ]) {
        const unsigned off = ((unsigned)((tile0 + tt) * 32 + l31) * (unsigned)K + (unsigned)(4 * kh2)) * 4u;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
            av[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, tt < a.tiles_per_group ? off + (unsigned)j * 32u : 0xFFFFFFF0u, 0, 0));
    };
    float4 acur[KJ], anxt[KJ];
    issue(t, acur);
    for (; t < a.tiles_per_group; t += tstride) {
        issue(t + tstride, anxt);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            float4 x4 = acur[j];
            if constexpr (BNI) {
                x4.x = __builtin_elementwise_maximum(__builtin_fmaf(x4.x, bsc[j].x, bsh[j].x), lo);
                x4.y = __builtin_elementwise_maximum(__builtin_fmaf(x4.y, bsc[j].y, bsh[j].y), lo);
                x4.z = __builtin_elementwise_maximum(__builtin_fmaf(x4.z, bsc[j].z, bsh[j].z), lo);
                x4.w = __builtin_elementwise_maximum(__builtin_fmaf(x4.w, bsc[j].w, bsh[j].w), lo);
            }
            const float xe[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xe[e], bw[j][e][nt], acc[nt], 0, 0, 0);
        }
        // ---- epilogue: element (r, lane) = pixel 8 (r >> 2) + 4 kh2 + (r & 3) of the tile, column n_base + 32 nt + l31 ----
        const int pix0 = (tile0 + t) * 32;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float q1[NT], q2[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { q1[nt] = 0.f; q2[nt] = 0.f; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = rq * 4 + c;
                const int p = pix0 + 8 * rq + 4 * kh2 + c;
                unsigned rp;
                if (a.taps == 1) rp = (unsigned)p * (unsigned)a.N * 4u;
                else {
                    int nb, oy, ox;
                    if (a.hw_shift >= 0) {
                        nb = p >> a.hw_shift;
                        const int rr = p & ((1 << a.hw_shift) - 1);
                        oy = rr >> a.w_shift; ox = rr & ((1 << a.w_shift) - 1);
                    } else {
                        const int hw = a.Hs * a.Ws;
                        nb = p / hw;
                        const int rr = p - nb * hw;
                        oy = rr / a.Ws; ox = rr - oy * a.Ws;
                    }
                    rp = (((unsigned)(nb * (2 * a.Hs) + 2 * oy) * (unsigned)(2 * a.Ws) + (unsigned)(2 * ox)) * (unsigned)a.Co) * 4u;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float v = acc[nt][r] + bv[nt];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, nok[nt] ? rp + cp[nt] : 0xFFFFFFF0u, 0, 0);
                    const float vs = nok[nt] ? v : 0.f;
                    vmax = fmaxf(vmax, fabsf(vs));
                    q1[nt] += vs; q2[nt] = __builtin_fmaf(vs, vs, q2[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { st1[nt] += (double)q1[nt]; st2[nt] += (double)q2[nt]; }
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) acur[j] = anxt[j];
    }
    if (a.amax_out) bh_amax_commit(a.amax_out, vmax, blockIdx.x, sm_amax);
    if (a.bn_sums) {
        // half-waves by shuffle, the four waves and the taps of a channel through LDS, one atomic per (channel, moment) and workgroup
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double s1 = st1[nt] + __shfl_xor(st1[nt], 32, 64), s2 = st2[nt] + __shfl_xor(st2[nt], 32, 64);
            if (kh2 == 0) {
                red[((wave * NT + nt) * 32 + l31) * 2] = s1;
                red[((wave * NT + nt) * 32 + l31) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        // columns of this slice: n_base .. n_base + 32 NT; channel of column c: (n_base + c) % Co.  Thread (c, mom) with c the FIRST column of
        // its channel inside the slice adds the slice's other taps of that channel
        const int ncol = 32 * NT;
        if (tid < ncol * 2) {
            const int c = tid >> 1, mom = tid & 1, n = n_base + c;
            if (n < a.N) {
                const int co = n % a.Co;
                const bool first = c < a.Co || a.Co > ncol;          // no earlier column of the slice has this channel
                if (first) {
                    double tot = 0.0;
                    for (int c2 = c; c2 < ncol && n_base + c2 < a.N; c2 += a.Co)
#pragma unroll
                        for (int w = 0; w < 4; ++w) tot += red[((w * NT + (c2 >> 5)) * 32 + (c2 & 31)) * 2 + mom];
                    bh_acc_add(&a.bn_sums[bn_sum_index(0, a.groups, grp, a.Co, co, mom)], tot, a.bn_det);
                }
            }
        }
    }
}

template <int K, int NT, bool BNI>
static int pw_launch(const PwArgs& a, hipStream_t s, const char* name) {
    if (bh_query("pw_kernel<%d,%d,%s>", K, NT, BNI ? "true" : "false")) return BH_OK;
    (void)name;
    hipLaunchKernelGGL((pw_kernel<K, NT, BNI>), dim3(a.slices * a.wg_per_group * a.groups), dim3(256), 0, s, a);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

BH_KNOB(g_pw_on, 1); BH_KNOB(g_pw_wgs, 512);
#ifdef BH_TUNING
void bh_pointwise_tune(int what, int v) { if (what == 0) g_pw_on = v; else if (what == 1) g_pw_wgs = v; }
#endif

// *taken = 1 when the launch was made (or, under bh_conv_variant, would be) by pw_kernel.  groups: of the statistics / of the BatchNorm-on-load
// table (the same stacks of images).
int bh_pointwise_try(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, hipStream_t stream, int* taken,
                     double* bn_sums, int groups, float* amax_y, const bh_bn_in* bni) {
    *taken = 0;
    if (!g_pw_on || d->in_nchw || d->out_nchw || d->precision == 1) return BH_OK;
    const bool tconv = d->transposed && d->kh == 2 && d->kw == 2 && d->stride == 2 && d->pad == 0;
    const bool c1 = !d->transposed && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0;
    if (!tconv && !c1) return BH_OK;
    const int K = d->Ci, taps = tconv ? 4 : 1, N = taps * d->Co;
    if (K != 32 && K != 64) return BH_OK;
    if (bni && (!c1 || bni->groups < 1)) return BH_OK;
    // The kernel walks ONE partition of the images: with statistics AND a BatchNorm-on-load table the two group counts must agree (the
    // generic kernel keeps them apart and takes the launch otherwise); with a table alone the partition is the table's.
    if (bni && bn_sums && groups != bni->groups) return BH_OK;
    if (bni && !bn_sums) groups = bni->groups;
    if (groups < 1 || d->N % groups) return BH_OK;
    const long long M = (long long)d->N * d->Hi * d->Wi, mpg = M / groups;
    if (M < 65536 || mpg % 32) return BH_OK;             // (small maps: the generic kernel's many workgroups are fine there)
    const long long xe = M * K, ye = M * N;
    if (xe >= (1ll << 30) || ye >= (1ll << 30)) return BH_OK;
    // instance: (K, NT) with 32 NT columns per wave; K = 64 keeps NT <= 2 (registers)
    int NT;
    if (K == 32) NT = N > 64 ? 4 : (N > 32 ? 2 : 1);
    else NT = N > 32 ? 2 : 1;
    if (K == 32 && N > 128 && N % 128) return BH_OK;
    const int ncol = 32 * NT;
    if (N > ncol && N % ncol) return BH_OK;
    // statistics: a channel's taps must lie in one slice or be whole slices apart (Co divides the slice width or is a multiple of it)
    if (bn_sums && !((ncol % d->Co) == 0 || (d->Co % ncol) == 0)) return BH_OK;
    PwArgs a = {};
    a.X = x; a.W = w; a.bias = bias; a.Y = y;
    a.M = (int)M; a.N = N; a.Co = d->Co; a.taps = taps; a.Hs = d->Hi; a.Ws = d->Wi;
    a.hw_shift = a.w_shift = -1;
    for (int b = 0; b < 16; ++b) if (d->Wi == (1 << b)) a.w_shift = b;
    for (int b = 0; b < 31; ++b) if ((long long)d->Hi * d->Wi == (1ll << b)) a.hw_shift = b;
    if (a.w_shift < 0) a.hw_shift = -1;
    a.w_nk = c1 ? 1 : 0;
    a.slices = (N + ncol - 1) / ncol;
    a.groups = groups;
    a.tiles_per_group = (int)(mpg / 32);
    int wpg = g_pw_wgs / (a.slices * groups);
    if (wpg > (a.tiles_per_group + 3) / 4) wpg = (a.tiles_per_group + 3) / 4;
    if (wpg < 1) wpg = 1;
    a.wg_per_group = wpg;
    a.x_bytes = (unsigned)(xe * 4); a.y_bytes = (unsigned)(ye * 4);
    if (bni) { a.bni = bni->table; a.bni_relu = bni->relu; }
    a.bn_sums = bn_sums; a.bn_det = (d->route & BH_ROUTE_DETERMINISTIC) ? 1 : 0;
    a.amax_out = reinterpret_cast<unsigned*>(amax_y);
    int rc = BH_E_UNSUPPORTED;
    if (K == 32 && NT == 4 && !bni) rc = pw_launch<32, 4, false>(a, stream, "");
    else if (K == 32 && NT == 2 && !bni) rc = pw_launch<32, 2, false>(a, stream, "");
    else if (K == 32 && NT == 1 && !bni) rc = pw_launch<32, 1, false>(a, stream, "");
    else if (K == 32 && NT == 1 && bni) rc = pw_launch<32, 1, true>(a, stream, "");
    else if (K == 64 && NT == 2 && !bni) rc = pw_launch<64, 2, false>(a, stream, "");
    else if (K == 64 && NT == 1 && !bni) rc = pw_launch<64, 1, false>(a, stream, "");
    else if (K == 64 && NT == 1 && bni) rc = pw_launch<64, 1, true>(a, stream, "");
    else if (K == 64 && NT == 2 && bni) rc = pw_launch<64, 2, true>(a, stream, "");
    else if (K == 32 && NT == 2 && bni) rc = pw_launch<32, 2, true>(a, stream, "");
    if (rc == BH_E_UNSUPPORTED) return BH_OK;
    if (rc) return rc;
    *taken = 1;
    return BH_OK;
}
