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
// K = 16 is far too small for MFMA tiles to pay; these are VALU kernels bounded by reading x (64 B/pixel).
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

// ---- one block of 256: derive the BatchNorm statistics of y from the x moments, update running stats ----
__global__ void __launch_bounds__(256) tail_stats_finalize_kernel(const double* __restrict__ part, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, TailGeom g, float momentum,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, double* __restrict__ ws) {
    __shared__ double mom[TAIL_MAXCI + TAIL_MAXCI * TAIL_MAXCI];
    const int Ci = g.Ci, nm = Ci + Ci * Ci;
    double* ystats = ws;
    double* xmom = ws + (size_t)g.groups * g.Cm * 2;
    const double n = (double)g.rows;
    float rm = 0.f, rv = 1.f;
    const int c = threadIdx.x;
    if (c < g.Cm) { rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 1.f; }
    for (int grp = 0; grp < g.groups; ++grp) {
        __syncthreads();
        for (int e = threadIdx.x; e < nm; e += 256) mom[e] = xmom[(size_t)grp * nm + e] / n;      // E[x_i], E[x_i x_j]
        __syncthreads();
        if (c < g.Cm) {
            double my = b1 ? (double)b1[c] : 0.0, eyy = 0;
            for (int i = 0; i < Ci; ++i) {
                const double wi = w1[c * Ci + i], mi = mom[i];
                my += wi * mi;
                double row = 0;
                for (int j = 0; j < Ci; ++j) row += (double)w1[c * Ci + j] * (mom[Ci + i * Ci + j] - mi * mom[j]);
                eyy += wi * row;
            }
            if (eyy < 0) eyy = 0;
            ystats[((size_t)grp * g.Cm + c) * 2] = my;
            ystats[((size_t)grp * g.Cm + c) * 2 + 1] = eyy;
            const float unb = (float)(n > 1 ? eyy * n / (n - 1) : eyy);
            rm = (1.f - momentum) * rm + momentum * (float)my;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
    }
    if (c < g.Cm) {
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

// ---- backward reduction: thread = channel c; grid (nchunks, groups), block Cm ----
// partial layout per (grp, chunk): [Cm][4 + CI] floats = { dbeta, dgamma, dW2_0..: see below }
//   q[0] = sum gbn, q[1] = sum gbn*yhat, q[2..2+Co) = sum g_o * z, q[6..6+CI) = sum gbn * x_k   (stride 6 + CI)
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_reduce_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ w2, const double* __restrict__ ystats,
                                                              const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                              float* __restrict__ part, TailGeom g, float eps, int use_running) {
    __shared__ __attribute__((aligned(16))) float xs[64 * CI];
    __shared__ float gs[64 * 4];
    // block = nsub * Cm threads: thread (sub, c) accumulates channel c over every nsub-th pixel of the chunk (more
    // resident waves than one thread per channel); each sub writes its own partial, the finalize sums them all
    const int grp = blockIdx.y, chunk = blockIdx.x, c = threadIdx.x % g.Cm, sub = threadIdx.x / g.Cm;
    const int nsub = blockDim.x / g.Cm;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
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
    for (int r0 = rbeg; r0 < rend; r0 += 64) {
        const int np = min(64, rend - r0);
        __syncthreads();
        for (int e = threadIdx.x; e < np * CI / 4; e += blockDim.x)
            reinterpret_cast<float4*>(xs)[e] = reinterpret_cast<const float4*>(xbase + (size_t)r0 * CI)[e];
        for (int e = threadIdx.x; e < np * 4; e += blockDim.x) {
            const int p = e >> 2, o = e & 3;
            const size_t m = (size_t)grp * g.rows + r0 + p;
            const size_t img = m / g.hw, pp = m - img * g.hw;
            gs[e] = (o < g.Co) ? gout[(img * g.Co + o) * g.hw + pp] : 0.f;
        }
        __syncthreads();
        for (int p = sub; p < np; p += nsub) {
            const float* xv = xs + p * CI;
            float y = bb;
#pragma unroll
            for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xv[k], y);
            const float yh = (y - mean) * invstd;
            const float zz = fmaf(y, scale, shift);          // same expression as the forward kernel (same ReLU mask)
            const float z = fmaxf(zz, 0.f);
            const float g0 = gs[p * 4], g1 = gs[p * 4 + 1], g2 = gs[p * 4 + 2], g3 = gs[p * 4 + 3];
            float gz = g0 * w2c[0] + g1 * w2c[1] + g2 * w2c[2] + g3 * w2c[3];
            const float gbn = (zz > 0.f) ? gz : 0.f;
            q0 += gbn; q1 = fmaf(gbn, yh, q1);
            qz[0] = fmaf(g0, z, qz[0]); qz[1] = fmaf(g1, z, qz[1]); qz[2] = fmaf(g2, z, qz[2]); qz[3] = fmaf(g3, z, qz[3]);
#pragma unroll
            for (int k = 0; k < CI; ++k) qs[k] = fmaf(gbn, xv[k], qs[k]);
        }
    }
    float* q = part + ((((size_t)grp * g.nchunks + chunk) * nsub + sub) * g.Cm + c) * (6 + CI);
    q[0] = q0; q[1] = q1; q[2] = qz[0]; q[3] = qz[1]; q[4] = qz[2]; q[5] = qz[3];
#pragma unroll
    for (int k = 0; k < CI; ++k) q[6 + k] = qs[k];
}

// ---- backward finalize: one wave per channel (grid Cm, block 64) ----
// coef[groups][Cm][4] = { A = gamma*invstd, kbeta = dbeta_g/n, kgamma = dgamma_g/n, unused }
template <int CI>
__global__ void __launch_bounds__(64) tail_bwd_finalize_kernel(const float* __restrict__ part, int nsub,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ gamma, const double* __restrict__ ws,
                                                               const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                               TailGeom g, float eps, int use_running, float* __restrict__ gw1,
                                                               float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                               float* __restrict__ gw2, float* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int nm = CI + CI * CI;
    const double* ystats = ws;
    const double* xmom = ws + (size_t)g.groups * g.Cm * 2;
    const double n = (double)g.rows;
    double tgam = 0, tbet = 0, tw2[4] = {0, 0, 0, 0}, tw1[CI];
    for (int k = 0; k < CI; ++k) tw1[k] = 0;
    for (int grp = 0; grp < g.groups; ++grp) {
        double acc[6 + CI];
        for (int e = 0; e < 6 + CI; ++e) acc[e] = 0;
        for (int k = lane; k < g.nchunks * nsub; k += 64) {
            const float* q = part + (((size_t)grp * g.nchunks * nsub + k) * g.Cm + c) * (6 + CI);
            for (int e = 0; e < 6 + CI; ++e) acc[e] += (double)q[e];
        }
        for (int e = 0; e < 6 + CI; ++e) acc[e] = wave_sum(acc[e]);
        double mean, var;
        if (use_running) { mean = rmean[c]; var = rvar[c]; }
        else { mean = ystats[((size_t)grp * g.Cm + c) * 2]; var = ystats[((size_t)grp * g.Cm + c) * 2 + 1]; }
        const double invstd = 1.0 / sqrt((double)(float)var + (double)eps);
        const double gm = gamma ? (double)gamma[c] : 1.0;
        const double A = gm * invstd;
        const double kb = use_running ? 0.0 : acc[0] / n, kg = use_running ? 0.0 : acc[1] / n;
        tbet += acc[0]; tgam += acc[1];
        for (int o = 0; o < 4; ++o) tw2[o] += acc[2 + o];
        // dW1[c][k] = A * ( S[c][k] - kb * sum_m x_k - kg * sum_m yhat_c x_k ),
        //   sum_m yhat_c x_k = invstd * ( sum_j W1[c][j] Sxx[j][k] + (b1_c - mean) * Sx[k] )
        const double* mo = xmom + (size_t)grp * nm;
        for (int k = 0; k < CI; ++k) {
            double syx = ((b1 ? (double)b1[c] : 0.0) - mean) * mo[k];
            for (int j = 0; j < CI; ++j) syx += (double)w1[c * CI + j] * mo[CI + j * CI + k];
            syx *= invstd;
            tw1[k] += A * (acc[6 + k] - kb * mo[k] - kg * syx);
        }
        if (lane == 0) {
            float* cf = coef + ((size_t)grp * g.Cm + c) * 4;
            cf[0] = (float)A; cf[1] = (float)kb; cf[2] = (float)kg; cf[3] = 0.f;
        }
    }
    if (lane == 0) {
        if (ggamma) ggamma[c] += (float)tgam;
        if (gbeta) gbeta[c] += (float)tbet;
        if (gw2) for (int o = 0; o < g.Co; ++o) gw2[o * g.Cm + c] += (float)tw2[o];
        if (gw1) for (int k = 0; k < CI; ++k) gw1[c * CI + k] += (float)tw1[k];
    }
}

// ---- backward dx: thread per pixel. grid (nblk, groups) ----
template <int CI>
__global__ void __launch_bounds__(256) tail_bwd_dx_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w2, const double* __restrict__ ystats,
                                                          const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                          const float* __restrict__ coef, float* __restrict__ gx, TailGeom g,
                                                          float eps, int use_running) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    TailLds L;
    const int grp = blockIdx.y;
    tail_load_consts(L, sm, g, w1, b1, gamma, beta, w2, ystats, rmean, rvar, use_running, grp, eps);
    float* cf = L.mu + g.Cm;                 // [Cm][4] copy of coef for this group
    for (int e = threadIdx.x; e < g.Cm * 4; e += 256) cf[e] = coef[(size_t)grp * g.Cm * 4 + e];
    __syncthreads();
    for (int r = blockIdx.x * 256 + threadIdx.x; r < g.rows; r += gridDim.x * 256) {
        const size_t m = (size_t)grp * g.rows + r;
        float xv[CI], gxv[CI];
        load_x<CI>(x + m * CI, xv);
#pragma unroll
        for (int k = 0; k < CI; ++k) gxv[k] = 0.f;
        const size_t img = m / g.hw, p = m - img * g.hw;
        const float* gp = gout + img * g.Co * g.hw + p;
        const float g0 = gp[0], g1 = g.Co > 1 ? gp[(size_t)g.hw] : 0.f, g2 = g.Co > 2 ? gp[2 * (size_t)g.hw] : 0.f,
                    g3 = g.Co > 3 ? gp[3 * (size_t)g.hw] : 0.f;
        for (int c = 0; c < g.Cm; ++c) {
            const float* wr = L.w1 + c * CI;
            const float* cs = L.cst + c * 8;
            float y = cs[0];
#pragma unroll
            for (int k = 0; k < CI; ++k) y = fmaf(wr[k], xv[k], y);
            const float yh = (y - L.mu[c]) * cs[7];
            const float zz = fmaf(y, cs[1], cs[2]);
            const float gz = g0 * cs[3] + g1 * cs[4] + g2 * cs[5] + g3 * cs[6];
            const float gbn = (zz > 0.f) ? gz : 0.f;
            const float gy = cf[c * 4] * (gbn - cf[c * 4 + 1] - yh * cf[c * 4 + 2]);
#pragma unroll
            for (int k = 0; k < CI; ++k) gxv[k] = fmaf(gy, wr[k], gxv[k]);
        }
#pragma unroll
        for (int k = 0; k < CI / 4; ++k)
            reinterpret_cast<float4*>(gx + m * CI)[k] = make_float4(gxv[4 * k], gxv[4 * k + 1], gxv[4 * k + 2], gxv[4 * k + 3]);
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
int bh_tail_scratch_floats(int groups, int Ci, int Cm) { return groups * TAIL_CHUNKS * 4 * Cm * (6 + Ci) + groups * Cm * 4; }

int bh_tail_fwd(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, float* running_mean,
                float* running_var, const float* w2, const float* b2, float* out, double* ws, int groups, int rows, int hw,
                int Ci, int Cm, int Co, float eps, float momentum, int use_running, void* stream) {
    TailGeom g;
    if (!x || !w1 || !w2 || !out || !ws) return BH_E_BADARG;
    if (!tail_geom(groups, rows, hw, Ci, Cm, Co, g) || (Ci != 16 && Ci != 32 && Ci != 8)) return BH_E_UNSUPPORTED;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    if (!use_running) {
        hipLaunchKernelGGL(tail_xmoments_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, ws + off_part(g));
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(tail_xmom_reduce_kernel, dim3(Ci + Ci * Ci, groups), dim3(64), 0, s, ws + off_part(g), g,
                           ws + off_xmom(g));
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(tail_stats_finalize_kernel, dim3(1), dim3(256), 0, s, ws + off_part(g), w1, b1, g, momentum,
                           running_mean, running_var, ws);
        BH_LAUNCH_CHECK();
    }
    const size_t lds = sizeof(float) * (Cm * Ci + Cm * 8 + Cm);
    dim3 grid(fwd_blocks(g), groups);
#define TAIL_FWD(CI_)                                                                                                       \
    hipLaunchKernelGGL((tail_fwd_kernel<CI_>), grid, dim3(256), lds, s, x, w1, b1, gamma, beta, w2, b2, ws, running_mean,  \
                       running_var, out, g, eps, use_running)
    if (Ci == 16) TAIL_FWD(16); else if (Ci == 32) TAIL_FWD(32); else TAIL_FWD(8);
#undef TAIL_FWD
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_tail_bwd(const float* gout, const float* x, const float* w1, const float* b1, const float* gamma, const float* beta,
                const float* w2, const double* ws, const float* running_mean, const float* running_var, float* gx, float* gw1,
                float* ggamma, float* gbeta, float* gw2, float* gb2, float* scratch, int groups, int rows, int hw, int Ci, int Cm,
                int Co, float eps, int use_running, void* stream) {
    TailGeom g;
    if (!gout || !x || !w1 || !w2 || !ws || !scratch) return BH_E_BADARG;
    if (!tail_geom(groups, rows, hw, Ci, Cm, Co, g) || (Ci != 16 && Ci != 32 && Ci != 8)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    const int nsub = 256 / Cm;                  // Cm in {64,128,256} -> 4, 2, 1 pixel sub-streams per block
    float* part = scratch;
    float* coef = scratch + (size_t)groups * TAIL_CHUNKS * 4 * Cm * (6 + Ci);
    const size_t lds = sizeof(float) * (Cm * Ci + Cm * 8 + Cm + Cm * 4);
    dim3 grid(fwd_blocks(g), groups);
#define TAIL_BWD(CI_)                                                                                                        \
    do {                                                                                                                     \
        hipLaunchKernelGGL((tail_bwd_reduce_kernel<CI_>), dim3(g.nchunks, groups), dim3(Cm * nsub), 0, s, gout, x, w1, b1,  \
                           gamma, beta, w2, ws, running_mean, running_var, part, g, eps, use_running);                       \
        BH_LAUNCH_CHECK();                                                                                                   \
        hipLaunchKernelGGL((tail_bwd_finalize_kernel<CI_>), dim3(Cm), dim3(64), 0, s, part, nsub, w1, b1, gamma, ws,         \
                           running_mean, running_var, g, eps, use_running, gw1, ggamma, gbeta, gw2, coef);                   \
        BH_LAUNCH_CHECK();                                                                                                   \
        if (gx) {                                                                                                            \
            hipLaunchKernelGGL((tail_bwd_dx_kernel<CI_>), grid, dim3(256), lds, s, gout, x, w1, b1, gamma, beta, w2, ws,     \
                               running_mean, running_var, coef, gx, g, eps, use_running);                                    \
            BH_LAUNCH_CHECK();                                                                                               \
        }                                                                                                                    \
    } while (0)
    if (Ci == 16) TAIL_BWD(16); else if (Ci == 32) TAIL_BWD(32); else TAIL_BWD(8);
#undef TAIL_BWD
    if (gb2) {
        const int nimg = groups * rows / hw;
        hipLaunchKernelGGL(tail_gb2_kernel, dim3(bh_deterministic() ? 1 : (nimg < 256 ? nimg : 256), Co), dim3(256), 0, s, gout, nimg, Co, hw, gb2);
        BH_LAUNCH_CHECK();
    }
    return BH_OK;
}

}  // extern "C"
