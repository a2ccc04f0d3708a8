// Fused network tail of the Zeng backbone (src/backbones/Rethinking.py:145-147, `layer8`):
//     Conv1x1(Ci -> Cm, bias) -> BatchNorm2d(Cm, training) -> ReLU -> Conv1x1(Cm -> Co, bias) -> NCHW field
// at full resolution (128x128).  Unfused, the Cm = 128-channel intermediate is a [2B*16384, 128] fp32 tensor
// (1 GiB at B = 64) that is written, re-read for the statistics, normalised, read by the last conv and touched
// ~10 more times in the backward pass.  Here it never exists:
//   * a 1x1 conv is linear, so the batch statistics of its output follow from the first and second moments of
//     its 16-channel INPUT:  mean_y = W1 mean_x + b1,  var_y[c] = w_c^T Cov(x) w_c   (moments in double);
//   * forward  = one pass over x: y_c = w_c.x + b1_c -> scale/shift -> ReLU -> 2 dot products, per pixel;
//   * backward = one reduction pass (thread = channel: dgamma, dbeta, dW2 and S[c][k] = sum_m g_c[m] x_k[m])
//     + a tiny finalize that turns S and the x-moments into dW1 + one per-pixel pass for dx.
// Round 4:
//   * forward (Ci = 16, Co <= 2): the 16 -> Cm product runs on the matrix pipe in the exact three-piece bf16 arithmetic of the 3x3
//     layers (common.h bh_split8: six v_mfma_f32_32x32x16_bf16 per 32 pixels x 32 channels, K = 16 is ONE MFMA step), BatchNorm +
//     ReLU + the Cm -> Co dot products run on the accumulators in C/D layout (rows = channels, so the second convolution is an
//     in-lane sum + one cross-half-wave add): tail_fwd_mfma_kernel, ~4.5x less VALU work than the per-pixel loop;
//   * backward: the gradient that reaches the field is SPARSE on the biHomE path (only the DSAC-sampled points of `pf` feed the
//     DLT: <= 128 of 16384 pixels per image are non-zero) and the adjoint is linear in it apart from the BatchNorm mean terms.
//     The reduce pass skips pixels whose output gradient is exactly zero (they add exactly zero to every sum), and the dx pass
//     evaluates gx = c0 - M x (the BatchNorm mean terms: a 16 x 16 affine map of x, made once per group by tail_bwd_lin_kernel)
//     for every pixel plus the Cm-channel term only where the gradient is non-zero (wave-cooperative for a few pixels per wave,
//     the per-lane channel loop when a wave holds many: dense gradients - the supervised configs - cost what they did).
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TAIL_MAXCI 32
#define TAIL_CHUNKS 512

struct TailGeom {
    int groups, rows, hw, Ci, Cm, Co, nchunks, rows_per_chunk;
};

static bool tail_geom(int groups, int rows, int hw, int Ci, int Cm, int Co, TailGeom& g) {
    if (groups < 1 || rows < 1 || Ci % 4 || Ci > TAIL_MAXCI || Ci * Ci > 256 || Cm % 64 || Cm > 256 || Co < 1 || Co > 4 ||
        rows % hw)
        return false;
    g.groups = groups; g.rows = rows; g.hw = hw; g.Ci = Ci; g.Cm = Cm; g.Co = Co;
    int n = rows / 1024;
    if (n < 1) n = 1;
    if (n > TAIL_CHUNKS) n = TAIL_CHUNKS;
    g.rows_per_chunk = (rows + n - 1) / n;
    g.nchunks = (rows + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return true;
}

// workspace layout (doubles):  ystats[groups][Cm][2] | xmom[groups][Ci + Ci*Ci] | partials[groups][nchunks][Ci + Ci*Ci]
static inline size_t off_xmom(const TailGeom& g) { return (size_t)g.groups * g.Cm * 2; }
static inline size_t off_part(const TailGeom& g) { return off_xmom(g) + (size_t)g.groups * (g.Ci + g.Ci * g.Ci); }

// ---- first/second moments of x for Ci = 16 on the matrix pipe (round 4): X^T X is a GEMM.  v_mfma_f32_16x16x4_f32 takes A[i][k] and
// B[k][j] from lane 16 k + i / 16 k + j: with i, j = channel and k = pixel BOTH operands are the same register, and that register is one
// coalesced dword load - lane L reads element L of four consecutive pixels (64 floats).  No LDS, no barrier inside the loop (the form below
// staged 64-pixel slabs in LDS between two barriers: 72 us on 2 x 1M pixels, 1.9 TB/s).  The fp32 accumulators (D[i][j] in lane 16 (i / 4) + j,
// register i % 4 - the matrix is symmetric, so the orientation does not matter) and the per-lane first-moment sums are flushed into doubles
// every 256 pixels.  grid (nchunks, groups), block 256: the four waves interleave 4-pixel steps of the chunk. ----
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) tail_xmoments16_mfma_kernel(const float* __restrict__ x, TailGeom g, double* __restrict__ part) {
    __shared__ double red[4][272];
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const float* base = x + (size_t)grp * g.rows * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dxx[4] = {0, 0, 0, 0}, dx = 0;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    float sx = 0.f;
    // eight 4-pixel steps per trip: the eight loads are in flight together (a trip of one wave = 128 pixels, of the block = 512)
    for (int p0 = rbeg + 4 * wave; p0 < rend; p0 += 128) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 16 * u + (lane >> 4);
            a[u] = p < rend ? base[(size_t)p * 16 + (lane & 15)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], a[u], acc, 0, 0, 0);
            sx += a[u];
        }
        if ((((p0 - rbeg) >> 7) & 7) == 7) {             // every 8 trips = 256 pixels of this wave: fp32 blocks -> doubles
#pragma unroll
            for (int r = 0; r < 4; ++r) { dxx[r] += (double)acc[r]; acc[r] = 0.f; }
            dx += (double)sx; sx = 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dxx[r] += (double)acc[r];
    dx += (double)sx;
    dx += __shfl_xor(dx, 16, 64);                       // the four pixel rows of a step hold the same channel
    dx += __shfl_xor(dx, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * (lane >> 4) + r) * 16 + (lane & 15)] = dxx[r];
    if (lane < 16) red[wave][256 + lane] = dx;
    __syncthreads();
    double* o = part + ((size_t)grp * g.nchunks + chunk) * (16 + 256);
    for (int e = threadIdx.x; e < 272; e += 256) {
        const double tot = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        if (e < 256) o[16 + e] = tot; else o[e - 256] = tot;
    }
}

// ---- first/second moments of x: grid (nchunks, groups), block 256 ----
// Ci = 16 (the Zeng tail): a thread owns a 4 x 4 block of the 16 x 16 second-moment matrix for every 16th pixel of a 64-pixel
// slab (16 threads = one matrix, 16 pixel groups per block): two ds_read_b128 per 16 FMAs instead of two ds_read_b32 per FMA
// (round 3: 130 -> ~40 us on 2 x 1M pixels); the 16 groups are added through LDS once per chunk.  Other Ci: thread = (i, j) pair.
__global__ void __launch_bounds__(256) tail_xmoments_kernel(const float* __restrict__ x, TailGeom g, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float xs[64 * TAIL_MAXCI];
    const int Ci = g.Ci, grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const float* base = x + (size_t)grp * g.rows * Ci;
    if (Ci == 16) {
        __shared__ double red[16][16 * 17];
        const int t = threadIdx.x & 15, pg = threadIdx.x >> 4, ib = t >> 2, jb = t & 3;
        double sxx[16], sx[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) sxx[e] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) sx[e] = 0;
        for (int r0 = rbeg; r0 < rend; r0 += 64) {
            const int np = min(64, rend - r0);
            __syncthreads();
            for (int e = threadIdx.x; e < np * 4; e += 256)
                reinterpret_cast<float4*>(xs)[e] = reinterpret_cast<const float4*>(base + (size_t)r0 * 16)[e];
            __syncthreads();
            float a[16], ax[4];
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) ax[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = pg + 16 * q;
                if (p < np) {
                    const float4 vi = reinterpret_cast<const float4*>(xs)[p * 4 + ib], vj = reinterpret_cast<const float4*>(xs)[p * 4 + jb];
                    const float xi[4] = {vi.x, vi.y, vi.z, vi.w}, xj[4] = {vj.x, vj.y, vj.z, vj.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) a[u * 4 + v] = __builtin_fmaf(xi[u], xj[v], a[u * 4 + v]);
                        ax[u] += xi[u];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) sxx[e] += (double)a[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) sx[e] += (double)ax[e];
        }
        // pixel groups -> one matrix: [pg][16 x 16 second moments | 16 first moments] in fixed order
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int v = 0; v < 4; ++v) red[pg][(ib * 4 + u) * 16 + jb * 4 + v] = sxx[u * 4 + v];
            if (jb == 0) red[pg][256 + ib * 4 + u] = sx[u];
        }
        __syncthreads();
        double* p = part + ((size_t)grp * g.nchunks + chunk) * (16 + 256);
        for (int e = threadIdx.x; e < 272; e += 256) {
            double tot = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += red[q][e];
            if (e < 256) p[16 + e] = tot; else p[e - 256] = tot;
        }
        return;
    }
    const int i = threadIdx.x / Ci, j = threadIdx.x % Ci;
    const bool active = (int)threadIdx.x < Ci * Ci;
    double sxx = 0, sx = 0;
    for (int r0 = rbeg; r0 < rend; r0 += 64) {
        const int np = min(64, rend - r0);
        __syncthreads();
        for (int e = threadIdx.x; e < np * Ci / 4; e += 256)
            reinterpret_cast<float4*>(xs)[e] = reinterpret_cast<const float4*>(base + (size_t)r0 * Ci)[e];
        __syncthreads();
        if (active) {
            float axx = 0.f, ax = 0.f;
            for (int p = 0; p < np; ++p) {
                const float a = xs[p * Ci + i], b = xs[p * Ci + j];
                axx += a * b;
                ax += a;
            }
            sxx += axx; sx += ax;
        }
    }
    if (active) {
        double* p = part + ((size_t)grp * g.nchunks + chunk) * (Ci + Ci * Ci);
        p[Ci + i * Ci + j] = sxx;
        if (j == 0) p[i] = sx;
    }
}

// ---- reduce the per-chunk moment partials: grid (Ci + Ci*Ci, groups), one wave each ----
__global__ void __launch_bounds__(64) tail_xmom_reduce_kernel(const double* __restrict__ part, TailGeom g, double* __restrict__ xmom) {
    const int nm = g.Ci + g.Ci * g.Ci, e = blockIdx.x, grp = blockIdx.y;
    double s = 0;
    for (int k = threadIdx.x; k < g.nchunks; k += 64) s += part[((size_t)grp * g.nchunks + k) * nm + e];
    s = wave_sum(s);
    if (threadIdx.x == 0) xmom[(size_t)grp * nm + e] = s;
}

// ---- derive the BatchNorm statistics of y from the x moments, update running stats: Ci lanes per channel (lane i owns row i of
// the covariance form), 256 / Ci channels per block (round 4: the one-block, one-thread-per-channel form took 46 us of dependent
// double-precision latency) ----
__global__ void __launch_bounds__(256) tail_stats_finalize_kernel(const double* __restrict__ part, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, TailGeom g, float momentum,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, double* __restrict__ ws) {
    __shared__ double mom[TAIL_MAXCI + TAIL_MAXCI * TAIL_MAXCI];
    __shared__ float wl[256];
    const int Ci = g.Ci, nm = Ci + Ci * Ci, cpb = 256 / Ci;
    double* ystats = ws;
    double* xmom = ws + (size_t)g.groups * g.Cm * 2;
    const double n = (double)g.rows;
    const int i = threadIdx.x % Ci, cl = threadIdx.x / Ci, c = blockIdx.x * cpb + cl;
    const bool act = c < g.Cm;
    wl[threadIdx.x] = act ? w1[c * Ci + i] : 0.f;
    float rm = 0.f, rv = 1.f;
    if (act && i == 0) { rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 1.f; }
    for (int grp = 0; grp < g.groups; ++grp) {
        __syncthreads();
        for (int e = threadIdx.x; e < nm; e += 256) mom[e] = xmom[(size_t)grp * nm + e] / n;      // E[x_i], E[x_i x_j]
        __syncthreads();
        const double wi = wl[cl * Ci + i], mi = mom[i];
        double row = 0;
        for (int j = 0; j < Ci; ++j) row += (double)wl[cl * Ci + j] * (mom[Ci + j * Ci + i] - mi * mom[j]);     // (the moment matrix is symmetric)
        double eyy = wi * row, my = wi * mi;
        for (int off = Ci >> 1; off > 0; off >>= 1) { eyy += __shfl_xor(eyy, off, 64); my += __shfl_xor(my, off, 64); }
        if (act && i == 0) {
            my += b1 ? (double)b1[c] : 0.0;
            if (eyy < 0) eyy = 0;
            ystats[((size_t)grp * g.Cm + c) * 2] = my;
            ystats[((size_t)grp * g.Cm + c) * 2 + 1] = eyy;
            const float unb = (float)(n > 1 ? eyy * n / (n - 1) : eyy);
            rm = (1.f - momentum) * rm + momentum * (float)my;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
    }
    if (act && i == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

// per-channel constants in LDS: w1 row (Ci), b1, mean, invstd, gamma, beta, w2[0..Co)
struct TailLds {
    float* w1;    // [Cm][Ci]
    float* cst;   // [Cm][8]: b1, scale(=gamma*invstd), shift(=beta-mean*scale), w2_0, w2_1, w2_2, w2_3, invstd
    float* mu;    // [Cm]
};

__device__ __forceinline__ void tail_load_consts(TailLds& L, float* sm, const TailGeom& g, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ w2,
                                                 const double* __restrict__ ystats, const float* __restrict__ rmean,
                                                 const float* __restrict__ rvar, int use_running, int grp, float eps) {
    L.w1 = sm; L.cst = sm + g.Cm * g.Ci; L.mu = L.cst + g.Cm * 8;
    for (int e = threadIdx.x; e < g.Cm * g.Ci; e += blockDim.x) L.w1[e] = w1[e];
    for (int c = threadIdx.x; c < g.Cm; c += blockDim.x) {
        float mean, var;
        if (use_running) { mean = rmean[c]; var = rvar[c]; }
        else { mean = (float)ystats[((size_t)grp * g.Cm + c) * 2]; var = (float)ystats[((size_t)grp * g.Cm + c) * 2 + 1]; }
        const float invstd = 1.0f / sqrtf(var + eps);
        const float scale = (gamma ? gamma[c] : 1.f) * invstd;
        L.cst[c * 8 + 0] = b1 ? b1[c] : 0.f;
        L.cst[c * 8 + 1] = scale;
        L.cst[c * 8 + 2] = (beta ? beta[c] : 0.f) - mean * scale;
        for (int o = 0; o < 4; ++o) L.cst[c * 8 + 3 + o] = (o < g.Co) ? w2[o * g.Cm + c] : 0.f;
        L.cst[c * 8 + 7] = invstd;
        L.mu[c] = mean;
    }
    __syncthreads();
}

template <int CI>
__device__ __forceinline__ void load_x(const float* __restrict__ p, float (&xv)[CI]) {
#pragma unroll
    for (int k = 0; k < CI / 4; ++k) {
        const float4 v = reinterpret_cast<const float4*>(p)[k];
        xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
    }
}

// ---- forward: thread per pixel. grid (nblk, groups) ----
template <int CI>
__global__ void __launch_bounds__(256) tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, const double* __restrict__ ystats,
                                                       const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                       float* __restrict__ out, TailGeom g, float eps, int use_running) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    TailLds L;
    const int grp = blockIdx.y;
    tail_load_consts(L, sm, g, w1, b1, gamma, beta, w2, ystats, rmean, rvar, use_running, grp, eps);
    float bo[4];
    for (int o = 0; o < 4; ++o) bo[o] = (o < g.Co && b2) ? b2[o] : 0.f;
    // two pixels per thread, float2 arithmetic: the channel loop is FMA-bound and v_pk_fma_f32 issues two fp32 FMAs per
    // lane per cycle (the weight / constant operands are the same for both pixels)
    const int stride = gridDim.x * 256;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < g.rows; r += 2 * stride) {
        const int r1 = r + stride;
        const bool has1 = r1 < g.rows;
        const size_t m0 = (size_t)grp * g.rows + r, m1 = has1 ? (size_t)grp * g.rows + r1 : m0;
        float xa[CI], xb[CI];
        load_x<CI>(x + m0 * CI, xa);
        load_x<CI>(x + m1 * CI, xb);
        f32x2 xv[CI];
#pragma unroll
        for (int k = 0; k < CI; ++k) xv[k] = f32x2{xa[k], xb[k]};
        f32x2 o0 = f32x2{bo[0], bo[0]}, o1 = f32x2{bo[1], bo[1]}, o2 = f32x2{bo[2], bo[2]}, o3 = f32x2{bo[3], bo[3]};
        for (int c = 0; c < g.Cm; ++c) {
            const float* wr = L.w1 + c * CI;
            const float* cs = L.cst + c * 8;
            f32x2 y = f32x2{cs[0], cs[0]};
#pragma unroll
            for (int k = 0; k < CI; ++k) y = __builtin_elementwise_fma(f32x2{wr[k], wr[k]}, xv[k], y);
            f32x2 z = __builtin_elementwise_fma(y, f32x2{cs[1], cs[1]}, f32x2{cs[2], cs[2]});
            z = __builtin_elementwise_max(z, f32x2{0.f, 0.f});
            o0 = __builtin_elementwise_fma(z, f32x2{cs[3], cs[3]}, o0);
            o1 = __builtin_elementwise_fma(z, f32x2{cs[4], cs[4]}, o1);
            o2 = __builtin_elementwise_fma(z, f32x2{cs[5], cs[5]}, o2);
            o3 = __builtin_elementwise_fma(z, f32x2{cs[6], cs[6]}, o3);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !has1) break;
            const size_t m = h ? m1 : m0;
            const size_t img = m / g.hw, p = m - img * g.hw;
            float* op = out + img * g.Co * g.hw + p;
            op[0] = o0[h];
            if (g.Co > 1) op[(size_t)g.hw] = o1[h];
            if (g.Co > 2) op[2 * (size_t)g.hw] = o2[h];
            if (g.Co > 3) op[3 * (size_t)g.hw] = o3[h];
        }
    }
}

// ---- forward on the matrix pipe (Ci = 16, Cm % 32 == 0, Co <= 2, rows % 32 == 0, hw % 32 == 0): grid (nblk, groups), block 256 ----
// A wave owns 32 consecutive pixels per step.  D[m = channel][n = pixel] = sum_k W1[m][k] x[n][k] with K = 16 = ONE step of
// v_mfma_f32_32x32x16_bf16: A = W1 (lane (l31, kh2) holds W1[32 blk + l31][8 kh2 .. 8 kh2 + 8), cut once per wave into the three exact
// bf16 pieces and kept in registers), B = x (lane (l31, kh2) holds channels 8 kh2 .. 8 kh2 + 8 of pixel l31: the wave reads 2 KB of
// contiguous NHWC memory), six piece products per 32 x 32 block, small ones first (common.h bh_split8: fp32 accuracy).  C/D layout:
// lane (l31, kh2) holds pixel l31, register r holds channel 32 blk + (r & 3) + 8 (r >> 2) + 4 kh2 - so BatchNorm + ReLU read their
// per-channel constants as half-wave broadcasts from LDS and the Cm -> Co convolution is an in-lane sum over the registers plus ONE
// add across the half-waves.
typedef __bf16 tail_bf16x8 __attribute__((ext_vector_type(8)));
typedef float tail_f32x16 __attribute__((ext_vector_type(16)));
#define TAIL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tail_bf16x8, a), __builtin_bit_cast(tail_bf16x8, b), c, 0, 0, 0)
template <int NB>      // Cm / 32
__global__ void __launch_bounds__(256, 2) tail_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, const double* __restrict__ ystats,
                                                               const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                               float* __restrict__ out, TailGeom g, float eps, int use_running) {
    __shared__ __attribute__((aligned(16))) float4 cst[NB * 32];     // per channel: scale, shift (+ b1 * scale), w2[0][c], w2[1][c]
    const int grp = blockIdx.y, Cm = NB * 32;
    for (int c = threadIdx.x; c < Cm; c += 256) {
        float mean, var;
        if (use_running) { mean = rmean[c]; var = rvar[c]; }
        else { mean = (float)ystats[((size_t)grp * Cm + c) * 2]; var = (float)ystats[((size_t)grp * Cm + c) * 2 + 1]; }
        const float invstd = 1.0f / sqrtf(var + eps);
        const float scale = (gamma ? gamma[c] : 1.f) * invstd;
        const float shift = (beta ? beta[c] : 0.f) - mean * scale;
        cst[c] = make_float4(scale, fmaf(b1 ? b1[c] : 0.f, scale, shift), w2[c], g.Co > 1 ? w2[Cm + c] : 0.f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh2 = lane >> 5;
    // the three pieces of W1 in A-fragment order, [blk][piece][lane] x 16 B in LDS (a real loop over the channel blocks below: the
    // fully unrolled form with the fragments in registers made the scheduler hoist every constant read and spill)
    __shared__ __attribute__((aligned(16))) uint4 wpl[NB * 3 * 64];
    if (wave == 0) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const float4* src = reinterpret_cast<const float4*>(w1 + (size_t)(blk * 32 + l31) * 16 + kh2 * 8);
            uint4 p0, p1, p2;
            bh_split8(src[0], src[1], p0, p1, p2);
            wpl[(blk * 3 + 0) * 64 + lane] = p0; wpl[(blk * 3 + 1) * 64 + lane] = p1; wpl[(blk * 3 + 2) * 64 + lane] = p2;
        }
    }
    __syncthreads();
    const float bo = !b2 ? 0.f : (kh2 == 0 ? b2[0] : (g.Co > 1 ? b2[1] : 0.f));
    // a step = 64 pixels = two 32-pixel sets (u = 0, 1) that share every constant read
    const int nsteps = g.rows / 64, sstride = gridDim.x * 4;
    const float* xg = x + (size_t)grp * g.rows * 16 + (size_t)l31 * 16 + kh2 * 8;
    int si = blockIdx.x * 4 + wave;
    float4 xa[2], xb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { xa[u] = make_float4(0.f, 0.f, 0.f, 0.f); xb[u] = xa[u]; }
    if (si < nsteps) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4* px = reinterpret_cast<const float4*>(xg + ((size_t)si * 64 + u * 32) * 16);
            xa[u] = px[0]; xb[u] = px[1];
        }
    }
    for (; si < nsteps; si += sstride) {
        uint4 xp[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) bh_split8(xa[u], xb[u], xp[u][0], xp[u][1], xp[u][2]);
        const int sn = si + sstride;
        if (sn < nsteps) {                               // the next step's pixels arrive under this step's MFMAs
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4* px = reinterpret_cast<const float4*>(xg + ((size_t)sn * 64 + u * 32) * 16);
                xa[u] = px[0]; xb[u] = px[1];
            }
        }
        float o0[2] = {0.f, 0.f}, o1[2] = {0.f, 0.f};
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            const uint4 w0 = wpl[(blk * 3 + 0) * 64 + lane], wm = wpl[(blk * 3 + 1) * 64 + lane], wl = wpl[(blk * 3 + 2) * 64 + lane];
            const float4* cb = cst + blk * 32 + 4 * kh2;
            tail_f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
                acc[u] = TAIL_MFMA(wl, xp[u][0], acc[u]);      // lo * hi, hi * lo, mid * mid, mid * hi, hi * mid, hi * hi
                acc[u] = TAIL_MFMA(w0, xp[u][2], acc[u]);
                acc[u] = TAIL_MFMA(wm, xp[u][1], acc[u]);
                acc[u] = TAIL_MFMA(wm, xp[u][0], acc[u]);
                acc[u] = TAIL_MFMA(w0, xp[u][1], acc[u]);
                acc[u] = TAIL_MFMA(w0, xp[u][0], acc[u]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 c = cb[(r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float t = fmaxf(fmaf(acc[u][r], c.x, c.y), 0.f);
                    o0[u] = fmaf(t, c.z, o0[u]);
                    o1[u] = fmaf(t, c.w, o1[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float a0 = o0[u] + __shfl_xor(o0[u], 32, 64), a1 = o1[u] + __shfl_xor(o1[u], 32, 64);
            const size_t m = (size_t)grp * g.rows + (size_t)si * 64 + u * 32 + l31;
            const size_t img = m / g.hw, p = m - img * g.hw;
            if (kh2 == 0) out[(img * g.Co) * g.hw + p] = a0 + bo;
            else if (g.Co > 1) out[(img * g.Co + 1) * g.hw + p] = a1 + bo;
        }
    }
}

// ---- backward reduction: thread = channel c; grid (nchunks, groups), block Cm * nsub = 256 ----
// partial layout per (grp, chunk): [Cm][4 + CI] floats = { dbeta, dgamma, dW2_0..: see below }
//   q[0] = sum gbn, q[1] = sum gbn*yhat, q[2..2+Co) = sum g_o * z, q[6..6+CI) = sum gbn * x_k   (stride 6 + CI)
// Round 4: a pixel whose output gradient is exactly zero adds exactly zero to every one of these sums.  A slab of 256 pixels (one
// per thread) is scanned first - coalesced plane reads of gout - and the pixels that carry a gradient are compacted IN PIXEL ORDER
// (wave ballots + a four-entry prefix: the summation order is a function of the data, not of the scheduling); only their x rows
// are fetched.  biHomE: <= 128 DSAC-sampled points of 16384 per image; a dense gradient runs the same loop over all 256.
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_reduce_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ w2, const double* __restrict__ ystats,
                                                              const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                              float* __restrict__ part, TailGeom g, float eps, int use_running) {
    __shared__ __attribute__((aligned(16))) float xs[64 * CI];
    __shared__ __attribute__((aligned(16))) float gs[256 * 4];
    __shared__ int list[256];
    __shared__ int wcount[4];
    // block = nsub * Cm threads: thread (sub, c) accumulates channel c over every nsub-th listed pixel (more resident waves than one
    // thread per channel); each sub writes its own partial, the finalize sums them all
    const int grp = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x % g.Cm, sub = threadIdx.x / g.Cm;
    const int nsub = blockDim.x / g.Cm;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this thread's channel constants in registers
    float wr[CI];
#pragma unroll
    for (int k = 0; k < CI; ++k) wr[k] = w1[c * CI + k];
    float mean, var;
    if (use_running) { mean = rmean[c]; var = rvar[c]; }
    else { mean = (float)ystats[((size_t)grp * g.Cm + c) * 2]; var = (float)ystats[((size_t)grp * g.Cm + c) * 2 + 1]; }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f, bb = b1 ? b1[c] : 0.f;
    const float scale = gm * invstd, shift = bt - mean * scale;
    float w2c[4];
    for (int o = 0; o < 4; ++o) w2c[o] = (o < g.Co) ? w2[o * g.Cm + c] : 0.f;
    float q0 = 0.f, q1 = 0.f, qz[4] = {0.f, 0.f, 0.f, 0.f}, qs[CI];
#pragma unroll
    for (int k = 0; k < CI; ++k) qs[k] = 0.f;
    const float* xbase = x + (size_t)grp * g.rows * CI;
    const int SL = blockDim.x, nwaves = (blockDim.x + 63) >> 6;       // slab = one pixel per thread (Cm = 192: 192 threads)
    for (int r0 = rbeg; r0 < rend; r0 += SL) {
        const int np = min(SL, rend - r0);
        __syncthreads();                                  // the previous slab's readers of gs / list are done
        bool nzp = false;
        if ((int)threadIdx.x < np) {
            const size_t m = (size_t)grp * g.rows + r0 + threadIdx.x;
            const size_t img = m / g.hw, pp = m - img * g.hw;
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
            gv.x = gout[(img * g.Co) * g.hw + pp];
            if (g.Co > 1) gv.y = gout[(img * g.Co + 1) * g.hw + pp];
            if (g.Co > 2) gv.z = gout[(img * g.Co + 2) * g.hw + pp];
            if (g.Co > 3) gv.w = gout[(img * g.Co + 3) * g.hw + pp];
            reinterpret_cast<float4*>(gs)[threadIdx.x] = gv;
            nzp = gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f;      // (NaN != 0: kept)
        }
        const unsigned long long bal = __ballot(nzp);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = 0, n = 0;
        for (int w = 0; w < nwaves; ++w) { if (w < wave) base += wcount[w]; n += wcount[w]; }
        if (n == 0) continue;                             // (uniform: every thread read the same four counts)
        if (nzp) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = threadIdx.x;
        for (int l0 = 0; l0 < n; l0 += 64) {
            const int nn = min(64, n - l0);
            __syncthreads();                              // list complete / the previous round's readers of xs are done
            for (int e = threadIdx.x; e < nn * (CI / 4); e += blockDim.x) {
                const int q = e / (CI / 4), f = e % (CI / 4);
                reinterpret_cast<float4*>(xs)[e] = reinterpret_cast<const float4*>(xbase + (size_t)(r0 + list[l0 + q]) * CI)[f];
            }
            __syncthreads();
            for (int q = sub; q < nn; q += nsub) {
                const float* xv = xs + q * CI;
                const float4 gq = reinterpret_cast<const float4*>(gs)[list[l0 + q]];
                float y = bb;
#pragma unroll
                for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xv[k], y);
                const float yh = (y - mean) * invstd;
                const float zz = fmaf(y, scale, shift);          // same expression as the forward kernel (same ReLU mask)
                const float z = fmaxf(zz, 0.f);
                const float g0 = gq.x, g1 = gq.y, g2 = gq.z, g3 = gq.w;
                float gz = g0 * w2c[0] + g1 * w2c[1] + g2 * w2c[2] + g3 * w2c[3];
                const float gbn = (zz > 0.f) ? gz : 0.f;
                q0 += gbn; q1 = fmaf(gbn, yh, q1);
                qz[0] = fmaf(g0, z, qz[0]); qz[1] = fmaf(g1, z, qz[1]); qz[2] = fmaf(g2, z, qz[2]); qz[3] = fmaf(g3, z, qz[3]);
#pragma unroll
                for (int k = 0; k < CI; ++k) qs[k] = fmaf(gbn, xv[k], qs[k]);
            }
        }
    }
    float* q = part + ((((size_t)grp * g.nchunks + chunk) * nsub + sub) * g.Cm + c) * (6 + CI);
    q[0] = q0; q[1] = q1; q[2] = qz[0]; q[3] = qz[1]; q[4] = qz[2]; q[5] = qz[3];
#pragma unroll
    for (int k = 0; k < CI; ++k) q[6 + k] = qs[k];
}

// ---- backward finalize: one block of 256 per channel (grid Cm): the four waves split the partial rows, lane k < CI of wave 0
// forms dW1[c][k] (round 4: the one-wave form spent 50 us on 16 dependent rounds of scattered loads + 256 redundant double FMAs) ----
// coef[groups][Cm][4] = { A = gamma*invstd, kbeta = dbeta_g/n, kgamma = dgamma_g/n, unused }
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_finalize_kernel(const float* __restrict__ part, int nsub,
                                                                const float* __restrict__ w1, const float* __restrict__ b1,
                                                                const float* __restrict__ gamma, const double* __restrict__ ws,
                                                                const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                                TailGeom g, float eps, int use_running, float* __restrict__ gw1,
                                                                float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                                float* __restrict__ gw2, float* __restrict__ coef) {
    __shared__ double red[4][6 + CI];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nm = CI + CI * CI;
    const double* ystats = ws;
    const double* xmom = ws + (size_t)g.groups * g.Cm * 2;
    const double n = (double)g.rows;
    double tgam = 0, tbet = 0, tw2[4] = {0, 0, 0, 0}, tw1 = 0;      // (wave 0; tw1: lane k < CI)
    for (int grp = 0; grp < g.groups; ++grp) {
        double acc[6 + CI];
#pragma unroll
        for (int e = 0; e < 6 + CI; ++e) acc[e] = 0;
        for (int k = threadIdx.x; k < g.nchunks * nsub; k += 256) {
            const float* q = part + (((size_t)grp * g.nchunks * nsub + k) * g.Cm + c) * (6 + CI);
            float v[6 + CI];
#pragma unroll
            for (int e = 0; e < (6 + CI) / 2; ++e) { const float2 t = reinterpret_cast<const float2*>(q)[e]; v[2 * e] = t.x; v[2 * e + 1] = t.y; }
#pragma unroll
            for (int e = 0; e < 6 + CI; ++e) acc[e] += (double)v[e];
        }
#pragma unroll
        for (int e = 0; e < 6 + CI; ++e) acc[e] = wave_sum(acc[e]);
        __syncthreads();                                  // (the previous group's readers of red are done)
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 6 + CI; ++e) red[wave][e] = acc[e];
        }
        __syncthreads();
        if (wave == 0) {
            const double a0 = red[0][0] + red[1][0] + red[2][0] + red[3][0], a1 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
            double mean, var;
            if (use_running) { mean = rmean[c]; var = rvar[c]; }
            else { mean = ystats[((size_t)grp * g.Cm + c) * 2]; var = ystats[((size_t)grp * g.Cm + c) * 2 + 1]; }
            const double invstd = 1.0 / sqrt((double)(float)var + (double)eps);
            const double gm = gamma ? (double)gamma[c] : 1.0;
            const double A = gm * invstd;
            const double kb = use_running ? 0.0 : a0 / n, kg = use_running ? 0.0 : a1 / n;
            tbet += a0; tgam += a1;
            for (int o = 0; o < 4; ++o) tw2[o] += red[0][2 + o] + red[1][2 + o] + red[2][2 + o] + red[3][2 + o];
            // dW1[c][k] = A * ( S[c][k] - kb * sum_m x_k - kg * sum_m yhat_c x_k ),
            //   sum_m yhat_c x_k = invstd * ( sum_j W1[c][j] Sxx[j][k] + (b1_c - mean) * Sx[k] )
            if (lane < CI) {
                const int k = lane;
                const double* mo = xmom + (size_t)grp * nm;
                const double sk = red[0][6 + k] + red[1][6 + k] + red[2][6 + k] + red[3][6 + k];
                double syx = ((b1 ? (double)b1[c] : 0.0) - mean) * mo[k];
                for (int j = 0; j < CI; ++j) syx += (double)w1[c * CI + j] * mo[CI + j * CI + k];
                syx *= invstd;
                tw1 += A * (sk - kb * mo[k] - kg * syx);
            }
            if (lane == 0) {
                float* cf = coef + ((size_t)grp * g.Cm + c) * 4;
                cf[0] = (float)A; cf[1] = (float)kb; cf[2] = (float)kg; cf[3] = 0.f;
            }
        }
    }
    if (wave == 0) {
        if (lane == 0) {
            if (ggamma) ggamma[c] += (float)tgam;
            if (gbeta) gbeta[c] += (float)tbet;
            if (gw2) for (int o = 0; o < g.Co; ++o) gw2[o * g.Cm + c] += (float)tw2[o];
        }
        if (gw1 && lane < CI) gw1[c * CI + lane] += (float)tw1;
    }
}

// ---- the BatchNorm mean terms of the input gradient as an affine map of x: grid (CI + 1, groups), block 256 ----
// gy_c = A_c (gbn_c - kb_c - yhat_c kg_c) and yhat_c = invstd_c (w_c . x + b1_c - mean_c) is linear in x, so for every pixel
//   gx_k = sum_c gy_c W1[c][k] = c0_k - sum_j M[j][k] x_j + sum_c A_c gbn_c W1[c][k],
//   M[j][k] = sum_c A_c kg_c invstd_c W1[c][j] W1[c][k],   c0_k = -sum_c A_c (kb_c + kg_c invstd_c (b1_c - mean_c)) W1[c][k].
// lin[groups][CI*CI + CI] = { -M (row j, column k), c0 }.  Block j < CI makes row j of M, block CI makes c0: thread (k, part)
// sums every (256 / CI)-th channel, the parts are added through LDS in a fixed order.
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_lin_kernel(const float* __restrict__ coef, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const double* __restrict__ ystats,
                                                           const float* __restrict__ rmean, const float* __restrict__ rvar, TailGeom g,
                                                           float eps, int use_running, float* __restrict__ lin) {
    constexpr int NP = 256 / CI;
    __shared__ double red[NP][CI];
    const int grp = blockIdx.y, j = blockIdx.x, k = threadIdx.x % CI, part = threadIdx.x / CI;
    const bool mat = j < CI;
    double acc = 0;
    for (int c = part; c < g.Cm; c += NP) {
        const float* cf = coef + ((size_t)grp * g.Cm + c) * 4;
        double mean, var;
        if (use_running) { mean = rmean[c]; var = rvar[c]; }
        else { mean = ystats[((size_t)grp * g.Cm + c) * 2]; var = ystats[((size_t)grp * g.Cm + c) * 2 + 1]; }
        const double invstd = 1.0 / sqrt((double)(float)var + (double)eps);
        const double A = cf[0], kb = cf[1], kg = cf[2], wk = w1[c * CI + k];
        if (mat) acc += A * kg * invstd * (double)w1[c * CI + j] * wk;
        else acc += A * (kb + kg * invstd * ((b1 ? (double)b1[c] : 0.0) - mean)) * wk;
    }
    red[part][k] = acc;
    __syncthreads();
    if (part == 0) {
        double t = 0;
        for (int q = 0; q < NP; ++q) t += red[q][k];
        lin[(size_t)grp * (CI * CI + CI) + (mat ? j * CI + k : CI * CI + k)] = (float)(-t);
    }
}

// ---- backward dx: thread per pixel, grid (nblk, groups).  gx = c0 - M x for every pixel (16 x 16 FMAs), plus the Cm-channel term
// sum_c A_c gbn_c W1[c][k] where the output gradient is non-zero: a wave with <= TAIL_SPARSE_MAX such pixels handles them one by one
// with its 64 lanes over the channels (wave_sum of the CI partial sums), a wave with more runs the per-lane channel loop ----
#define TAIL_SPARSE_MAX 16
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_dx_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w2, const double* __restrict__ ystats,
                                                          const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                          const float* __restrict__ coef, const float* __restrict__ lin,
                                                          float* __restrict__ gx, TailGeom g, float eps, int use_running) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    TailLds L;
    const int grp = blockIdx.y;
    tail_load_consts(L, sm, g, w1, b1, gamma, beta, w2, ystats, rmean, rvar, use_running, grp, eps);
    float* cf = L.mu + g.Cm;                 // [Cm][4] copy of coef for this group
    float* ln = cf + g.Cm * 4;               // [CI*CI + CI]: -M, c0
    float* w1p = ln + CI * CI + CI;          // [Cm][CI + 1]: W1 with a padded row pitch (per-lane rows without bank conflicts)
    for (int e = threadIdx.x; e < g.Cm * 4; e += 256) cf[e] = coef[(size_t)grp * g.Cm * 4 + e];
    for (int e = threadIdx.x; e < CI * CI + CI; e += 256) ln[e] = lin[(size_t)grp * (CI * CI + CI) + e];
    for (int e = threadIdx.x; e < g.Cm * CI; e += 256) w1p[(e / CI) * (CI + 1) + e % CI] = w1[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int rb = blockIdx.x * 256 + (threadIdx.x & ~63); rb < g.rows; rb += gridDim.x * 256) {       // (wave-uniform trip count)
        const int r = rb + lane;
        const bool valid = r < g.rows;
        const size_t m = (size_t)grp * g.rows + (valid ? r : rb);
        float xv[CI], gxv[CI];
        load_x<CI>(x + m * CI, xv);
        const size_t img = m / g.hw, p = m - img * g.hw;
        const float* gp = gout + img * g.Co * g.hw + p;
        const float g0 = gp[0], g1 = g.Co > 1 ? gp[(size_t)g.hw] : 0.f, g2 = g.Co > 2 ? gp[2 * (size_t)g.hw] : 0.f,
                    g3 = g.Co > 3 ? gp[3 * (size_t)g.hw] : 0.f;
        int zo = 0;                          // (a zero the compiler cannot see through: the CI * CI map stays in LDS, not in 272 hoisted VGPRs)
        asm volatile("" : "+v"(zo));
        const float* lz = ln + zo;
#pragma unroll
        for (int k = 0; k < CI; ++k) gxv[k] = lz[CI * CI + k];
#pragma unroll
        for (int j = 0; j < CI; ++j)
#pragma unroll
            for (int k = 0; k < CI; ++k) gxv[k] = fmaf(lz[j * CI + k], xv[j], gxv[k]);
        const bool nzl = valid && (g0 != 0.f || g1 != 0.f || g2 != 0.f || g3 != 0.f);
        unsigned long long mask = __ballot(nzl);
        if (mask) {
            if (__popcll(mask) > TAIL_SPARSE_MAX) {
                for (int c = 0; c < g.Cm; ++c) {
                    const float* wr = L.w1 + c * CI;
                    const float* cs = L.cst + c * 8;
                    float y = cs[0];
#pragma unroll
                    for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xv[k], y);
                    const float zz = fmaf(y, cs[1], cs[2]);
                    const float gz = g0 * cs[3] + g1 * cs[4] + g2 * cs[5] + g3 * cs[6];
                    const float gy = (zz > 0.f) ? cf[c * 4] * gz : 0.f;
#pragma unroll
                    for (int k = 0; k < CI; ++k) gxv[k] = fmaf(gy, wr[k], gxv[k]);
                }
            } else {
                while (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    float xs_[CI];
#pragma unroll
                    for (int k = 0; k < CI; ++k) xs_[k] = __shfl(xv[k], src, 64);
                    const float s0 = __shfl(g0, src, 64), s1 = __shfl(g1, src, 64), s2 = __shfl(g2, src, 64), s3 = __shfl(g3, src, 64);
                    float acc[CI];
#pragma unroll
                    for (int k = 0; k < CI; ++k) acc[k] = 0.f;
                    for (int c = lane; c < g.Cm; c += 64) {
                        const float* wr = w1p + c * (CI + 1);
                        const float* cs = L.cst + c * 8;
                        float y = cs[0];
#pragma unroll
                        for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xs_[k], y);
                        const float zz = fmaf(y, cs[1], cs[2]);
                        const float gz = s0 * cs[3] + s1 * cs[4] + s2 * cs[5] + s3 * cs[6];
                        const float gy = (zz > 0.f) ? cf[c * 4] * gz : 0.f;
#pragma unroll
                        for (int k = 0; k < CI; ++k) acc[k] = fmaf(gy, wr[k], acc[k]);
                    }
#pragma unroll
                    for (int k = 0; k < CI; ++k) {
                        const float t = wave_sum(acc[k]);
                        if (lane == src) gxv[k] += t;
                    }
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < CI / 4; ++k)
                reinterpret_cast<float4*>(gx + m * CI)[k] = make_float4(gxv[4 * k], gxv[4 * k + 1], gxv[4 * k + 2], gxv[4 * k + 3]);
        }
    }
}

// ---- backward dx on the matrix pipe (Ci = 16, Co <= 2, rows % 32 == 0, hw % 32 == 0): grid (nblk, groups), block 256 ----
// The affine part gx = c0 - M x is ONE MFMA step per 32 pixels in the exact three-piece arithmetic: A = -M (lane (k = l31 < 16, kh2)
// holds -M[k][8 kh2 .. 8 kh2 + 8), M is symmetric), B = x (as in tail_fwd_mfma_kernel), C = c0 per row.  C/D: lane (pixel l31, kh2),
// register r < 8 holds gx[pixel][k = (r & 3) + 8 (r >> 2) + 4 kh2] - two float4 stores per lane, 64 contiguous bytes per pixel from
// the two half-waves.  The Cm-channel term of the pixels that carry an output gradient is added in registers before the store: wave-
// cooperatively for <= TAIL_SPARSE_MAX such pixels per 32-pixel step, else by the two lanes of every pixel over half the channels each.
__global__ void __launch_bounds__(256, 2) tail_bwd_dx_mfma_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ w2, const double* __restrict__ ystats,
                                                                  const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                                  const float* __restrict__ coef, const float* __restrict__ lin,
                                                                  float* __restrict__ gx, TailGeom g, float eps, int use_running) {
    constexpr int CI = 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    TailLds L;
    const int grp = blockIdx.y;
    tail_load_consts(L, sm, g, w1, b1, gamma, beta, w2, ystats, rmean, rvar, use_running, grp, eps);
    float* cf = L.mu + g.Cm;                 // [Cm][4] copy of coef for this group
    float* w1p = cf + g.Cm * 4;              // [Cm][CI + 1]: W1 with a padded row pitch (per-lane rows without bank conflicts)
    for (int e = threadIdx.x; e < g.Cm * 4; e += 256) cf[e] = coef[(size_t)grp * g.Cm * 4 + e];
    for (int e = threadIdx.x; e < g.Cm * CI; e += 256) w1p[(e / CI) * (CI + 1) + e % CI] = w1[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh2 = lane >> 5;
    const float* lg = lin + (size_t)grp * (CI * CI + CI);
    uint4 mp[3];
    {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (l31 < CI) { const float4* src = reinterpret_cast<const float4*>(lg + l31 * CI + kh2 * 8); a = src[0]; b = src[1]; }
        bh_split8(a, b, mp[0], mp[1], mp[2]);
    }
    float cinit[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) cinit[r] = lg[CI * CI + (r & 3) + 8 * (r >> 2) + 4 * kh2];
    const int nsteps = g.rows / 32, sstride = gridDim.x * 4;
    const float* xg = x + (size_t)grp * g.rows * CI + (size_t)l31 * CI + kh2 * 8;
    int si = blockIdx.x * 4 + wave;
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
    float go = 0.f;
    if (si < nsteps) {
        const float4* px = reinterpret_cast<const float4*>(xg + (size_t)si * 32 * CI);
        xa = px[0]; xb = px[1];
        const size_t m = (size_t)grp * g.rows + (size_t)si * 32 + l31;
        const size_t img = m / g.hw, p = m - img * g.hw;
        if (kh2 < g.Co) go = gout[(img * g.Co + kh2) * g.hw + p];
    }
    for (; si < nsteps; si += sstride) {
        const float xm[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};     // channels 8 kh2 .. 8 kh2 + 8 of pixel l31
        const float gme = go;                                                      // output gradient plane kh2 of pixel l31
        uint4 xp[3];
        bh_split8(xa, xb, xp[0], xp[1], xp[2]);
        const size_t m = (size_t)grp * g.rows + (size_t)si * 32 + l31;
        const int sn = si + sstride;
        if (sn < nsteps) {
            const float4* px = reinterpret_cast<const float4*>(xg + (size_t)sn * 32 * CI);
            xa = px[0]; xb = px[1];
            const size_t mn = (size_t)grp * g.rows + (size_t)sn * 32 + l31;
            const size_t img = mn / g.hw, p = mn - img * g.hw;
            go = (kh2 < g.Co) ? gout[(img * g.Co + kh2) * g.hw + p] : 0.f;
        }
        tail_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r < 8 ? cinit[r] : 0.f;
        acc = TAIL_MFMA(mp[2], xp[0], acc);
        acc = TAIL_MFMA(mp[0], xp[2], acc);
        acc = TAIL_MFMA(mp[1], xp[1], acc);
        acc = TAIL_MFMA(mp[1], xp[0], acc);
        acc = TAIL_MFMA(mp[0], xp[1], acc);
        acc = TAIL_MFMA(mp[0], xp[0], acc);
        float gxv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) gxv[r] = acc[r];
        const unsigned long long bal = __ballot(gme != 0.f);
        unsigned pm = (unsigned)(bal | (bal >> 32));                               // pixels of this step that carry a gradient
        if (pm) {
            if (__popc(pm) > TAIL_SPARSE_MAX) {
                // dense: the two lanes of a pixel take the even / odd channels
                float xf[CI];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xo = __shfl_xor(xm[i], 32, 64);
                    xf[i] = kh2 ? xo : xm[i];
                    xf[8 + i] = kh2 ? xm[i] : xo;
                }
                const float gp_ = __shfl_xor(gme, 32, 64);
                const float g0 = kh2 ? gp_ : gme, g1 = kh2 ? gme : gp_;
                float a16[CI];
#pragma unroll
                for (int k = 0; k < CI; ++k) a16[k] = 0.f;
                for (int c = kh2; c < g.Cm; c += 2) {
                    const float* wr = L.w1 + c * CI;
                    const float* cs = L.cst + c * 8;
                    float y = cs[0];
#pragma unroll
                    for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xf[k], y);
                    const float zz = fmaf(y, cs[1], cs[2]);
                    const float gz = g0 * cs[3] + g1 * cs[4];
                    const float gy = (zz > 0.f) ? cf[c * 4] * gz : 0.f;
#pragma unroll
                    for (int k = 0; k < CI; ++k) a16[k] = fmaf(gy, wr[k], a16[k]);
                }
#pragma unroll
                for (int k = 0; k < CI; ++k) a16[k] += __shfl_xor(a16[k], 32, 64);
#pragma unroll
                for (int r = 0; r < 8; ++r) gxv[r] += kh2 ? a16[(r & 3) + 8 * (r >> 2) + 4] : a16[(r & 3) + 8 * (r >> 2)];
            } else {
                while (pm) {
                    const int src = __ffs((int)pm) - 1;
                    pm &= pm - 1;
                    float xs_[CI];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { xs_[i] = __shfl(xm[i], src, 64); xs_[8 + i] = __shfl(xm[i], src + 32, 64); }
                    const float s0 = __shfl(gme, src, 64), s1 = __shfl(gme, src + 32, 64);
                    float a16[CI];
#pragma unroll
                    for (int k = 0; k < CI; ++k) a16[k] = 0.f;
                    for (int c = lane; c < g.Cm; c += 64) {
                        const float* wr = w1p + c * (CI + 1);
                        const float* cs = L.cst + c * 8;
                        float y = cs[0];
#pragma unroll
                        for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xs_[k], y);
                        const float zz = fmaf(y, cs[1], cs[2]);
                        const float gz = s0 * cs[3] + s1 * cs[4];
                        const float gy = (zz > 0.f) ? cf[c * 4] * gz : 0.f;
#pragma unroll
                        for (int k = 0; k < CI; ++k) a16[k] = fmaf(gy, wr[k], a16[k]);
                    }
#pragma unroll
                    for (int k = 0; k < CI; ++k) a16[k] = wave_sum(a16[k]);
                    if (l31 == src) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) gxv[r] += kh2 ? a16[(r & 3) + 8 * (r >> 2) + 4] : a16[(r & 3) + 8 * (r >> 2)];
                    }
                }
            }
        }
        float4* dst = reinterpret_cast<float4*>(gx + m * CI);
        dst[kh2] = make_float4(gxv[0], gxv[1], gxv[2], gxv[3]);
        dst[2 + kh2] = make_float4(gxv[4], gxv[5], gxv[6], gxv[7]);
    }
}

// sum over all pixels of each NCHW output-gradient plane (bias gradient of the last conv): grid (64, Co), block 256
__global__ void __launch_bounds__(256) tail_gb2_kernel(const float* __restrict__ gout, int N, int Co, int hw, float* __restrict__ gb2) {
    __shared__ double sm[4];
    const int o = blockIdx.y;
    double acc = 0;
    // one (image, channel) plane per workgroup pass: contiguous, no per-element division
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        const float* pl = gout + ((size_t)n * Co + o) * hw;
        if ((hw & 3) == 0) {
            for (int p = threadIdx.x * 4; p < hw; p += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(pl + p);
                acc += (double)((v.x + v.y) + (v.z + v.w));
            }
        } else {
            for (int p = threadIdx.x; p < hw; p += 256) acc += pl[p];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gb2 + o, (float)(sm[0] + sm[1] + sm[2] + sm[3]));
}

static int fwd_blocks(const TailGeom& g) {
    int nb = (g.rows + 255) / 256;
    int cap = 4096 / g.groups;
    return nb > cap ? cap : nb;
}

extern "C" {

int bh_tail_ws_doubles(int groups, int Ci, int Cm) {
    return groups * Cm * 2 + groups * (Ci + Ci * Ci) * (1 + TAIL_CHUNKS);
}
int bh_tail_scratch_floats(int groups, int Ci, int Cm) { return groups * TAIL_CHUNKS * 4 * Cm * (6 + Ci) + groups * Cm * 4 + groups * (Ci * Ci + Ci); }

int bh_tail_fwd(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, float* running_mean,
                float* running_var, const float* w2, const float* b2, float* out, double* ws, int groups, int rows, int hw,
                int Ci, int Cm, int Co, float eps, float momentum, int use_running, void* stream) {
    return bh_tail_fwd_route(x, w1, b1, gamma, beta, running_mean, running_var, w2, b2, out, ws, groups, rows, hw, Ci, Cm, Co, eps,
                             momentum, use_running, 0, stream);
}

int bh_tail_fwd_route(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const float* w2, const float* b2, float* out, double* ws, int groups, int rows, int hw,
                      int Ci, int Cm, int Co, float eps, float momentum, int use_running, int tail_route, void* stream) {
    TailGeom g;
    if (!x || !w1 || !w2 || !out || !ws) return BH_E_BADARG;
    if (!tail_geom(groups, rows, hw, Ci, Cm, Co, g) || (Ci != 16 && Ci != 32 && Ci != 8)) return BH_E_UNSUPPORTED;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    if (!use_running) {
        if (Ci == 16 && !(tail_route & 2))
            hipLaunchKernelGGL(tail_xmoments16_mfma_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, ws + off_part(g));
        else
            hipLaunchKernelGGL(tail_xmoments_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, ws + off_part(g));
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(tail_xmom_reduce_kernel, dim3(Ci + Ci * Ci, groups), dim3(64), 0, s, ws + off_part(g), g,
                           ws + off_xmom(g));
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(tail_stats_finalize_kernel, dim3((Cm + 256 / Ci - 1) / (256 / Ci)), dim3(256), 0, s, ws + off_part(g), w1, b1, g, momentum,
                           running_mean, running_var, ws);
        BH_LAUNCH_CHECK();
    }
    const size_t lds = sizeof(float) * (Cm * Ci + Cm * 8 + Cm);
    dim3 grid(fwd_blocks(g), groups);
#define TAIL_FWD(CI_)                                                                                                       \
    hipLaunchKernelGGL((tail_fwd_kernel<CI_>), grid, dim3(256), lds, s, x, w1, b1, gamma, beta, w2, b2, ws, running_mean,  \
                       running_var, out, g, eps, use_running)
    // round 4: the matrix-pipe form where the shape allows (the Zeng tail: 16 -> 128 -> 2 on 128 x 128 fields)
    const bool mfma = Ci == 16 && Co <= 2 && rows % 64 == 0 && hw % 32 == 0 && (Cm == 64 || Cm == 128 || Cm == 256) &&
                      !(tail_route & 1);
    if (mfma) {
        // persistent: four workgroups per CU (114 VGPRs) walk the 64-pixel steps (no dispatch rounds, one constant / weight set-up each)
        int nb = (rows / 64 + 3) / 4;
        const int cap = (1024 + groups - 1) / groups;
        dim3 mgrid(nb > cap ? cap : nb, groups);
#define TAIL_FWD_M(NB_)                                                                                                      \
    hipLaunchKernelGGL((tail_fwd_mfma_kernel<NB_>), mgrid, dim3(256), 0, s, x, w1, b1, gamma, beta, w2, b2, ws, running_mean, \
                       running_var, out, g, eps, use_running)
        if (Cm == 64) TAIL_FWD_M(2); else if (Cm == 128) TAIL_FWD_M(4); else TAIL_FWD_M(8);
#undef TAIL_FWD_M
    } else if (Ci == 16) TAIL_FWD(16); else if (Ci == 32) TAIL_FWD(32); else TAIL_FWD(8);
#undef TAIL_FWD
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_tail_bwd(const float* gout, const float* x, const float* w1, const float* b1, const float* gamma, const float* beta,
                const float* w2, const double* ws, const float* running_mean, const float* running_var, float* gx, float* gw1,
                float* ggamma, float* gbeta, float* gw2, float* gb2, float* scratch, int groups, int rows, int hw, int Ci, int Cm,
                int Co, float eps, int use_running, void* stream) {
    return bh_tail_bwd_f(gout, x, w1, b1, gamma, beta, w2, ws, running_mean, running_var, gx, gw1, ggamma, gbeta, gw2, gb2, scratch, groups, rows,
                         hw, Ci, Cm, Co, eps, use_running, 0, stream);
}

int bh_tail_bwd_f(const float* gout, const float* x, const float* w1, const float* b1, const float* gamma, const float* beta,
                  const float* w2, const double* ws, const float* running_mean, const float* running_var, float* gx, float* gw1,
                  float* ggamma, float* gbeta, float* gw2, float* gb2, float* scratch, int groups, int rows, int hw, int Ci, int Cm,
                  int Co, float eps, int use_running, int flags, void* stream) {
    TailGeom g;
    if (!gout || !x || !w1 || !w2 || !ws || !scratch) return BH_E_BADARG;
    if (!tail_geom(groups, rows, hw, Ci, Cm, Co, g) || (Ci != 16 && Ci != 32 && Ci != 8)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    const int nsub = 256 / Cm;                  // Cm in {64,128,256} -> 4, 2, 1 pixel sub-streams per block
    float* part = scratch;
    float* coef = scratch + (size_t)groups * TAIL_CHUNKS * 4 * Cm * (6 + Ci);
    float* lin = coef + (size_t)groups * Cm * 4;
    const size_t lds = sizeof(float) * (Cm * Ci + Cm * 8 + Cm + Cm * 4 + Ci * Ci + Ci + Cm * (Ci + 1));
    dim3 grid(fwd_blocks(g), groups);
#define TAIL_BWD(CI_)                                                                                                        \
    do {                                                                                                                     \
        hipLaunchKernelGGL((tail_bwd_reduce_kernel<CI_>), dim3(g.nchunks, groups), dim3(Cm * nsub), 0, s, gout, x, w1, b1,  \
                           gamma, beta, w2, ws, running_mean, running_var, part, g, eps, use_running);                       \
        BH_LAUNCH_CHECK();                                                                                                   \
        hipLaunchKernelGGL((tail_bwd_finalize_kernel<CI_>), dim3(Cm), dim3(256), 0, s, part, nsub, w1, b1, gamma, ws,        \
                           running_mean, running_var, g, eps, use_running, gw1, ggamma, gbeta, gw2, coef);                   \
        BH_LAUNCH_CHECK();                                                                                                   \
        if (gx) {                                                                                                            \
            hipLaunchKernelGGL((tail_bwd_lin_kernel<CI_>), dim3(CI_ + 1, groups), dim3(256), 0, s, coef, w1, b1, ws, running_mean,   \
                               running_var, g, eps, use_running, lin);                                                       \
            BH_LAUNCH_CHECK();                                                                                               \
            if (CI_ == 16 && Co <= 2 && rows % 32 == 0 && hw % 32 == 0) {                                                     \
                int nb_ = (rows / 32 + 3) / 4;                                                                               \
                const int cap_ = (512 + groups - 1) / groups;                                                                \
                hipLaunchKernelGGL(tail_bwd_dx_mfma_kernel, dim3(nb_ > cap_ ? cap_ : nb_, groups), dim3(256),                 \
                                   sizeof(float) * (Cm * 16 + Cm * 8 + Cm + Cm * 4 + Cm * 17), s, gout, x, w1, b1, gamma, beta, w2, \
                                   ws, running_mean, running_var, coef, lin, gx, g, eps, use_running);                       \
            } else                                                                                                           \
            hipLaunchKernelGGL((tail_bwd_dx_kernel<CI_>), grid, dim3(256), lds, s, gout, x, w1, b1, gamma, beta, w2, ws,     \
                               running_mean, running_var, coef, lin, gx, g, eps, use_running);                               \
            BH_LAUNCH_CHECK();                                                                                               \
        }                                                                                                                    \
    } while (0)
    if (Ci == 16) TAIL_BWD(16); else if (Ci == 32) TAIL_BWD(32); else TAIL_BWD(8);
#undef TAIL_BWD
    if (gb2) {
        const int nimg = groups * rows / hw;
        hipLaunchKernelGGL(tail_gb2_kernel, dim3((flags & BH_F_DETERMINISTIC) ? 1 : (nimg < 256 ? nimg : 256), Co), dim3(256), 0, s, gout, nimg, Co, hw, gb2);
        BH_LAUNCH_CHECK();
    }
    return BH_OK;
}

}  // extern "C"
