// BatchNorm2d over ONE channel (round 3): the last layer of the Zhang feature extractor, Conv2d(8, 1) -> BatchNorm2d(1) -> ReLU
// (src/backbones/ContentAware.py:68-70) and of its mask predictor (:24-26).  The general kernels (bn.hip) are float4-over-channels
// (C % 4 == 0); with C = 1 the tensor is a plain vector of `rows` pixels per statistics group, read as float4 over PIXELS.
// Same contract as bh_bn_fwd / bh_bn_bwd (they dispatch here for C == 1): sums in the padded [groups][1][2] entries (caller-zeroed,
// deterministic-mode aware), running statistics updated group after group, backward = per-chunk partial sums (fixed order) + apply.
#include "common.h"

#define BN1_MAX_CHUNKS 256

struct Bn1Geom {
    int groups, nchunks, det;
    long long rows, rows_per_chunk;
};

static bool bn1_geom(int groups, int rows, Bn1Geom& g, int det) {
    if (groups < 1 || rows < 1) return false;
    g.groups = groups; g.rows = rows; g.det = det;
    long long n = rows / 4096;
    if (n < 1) n = 1;
    if (n > BN1_MAX_CHUNKS) n = BN1_MAX_CHUNKS;
    g.rows_per_chunk = ((rows + n - 1) / n + 3) / 4 * 4;
    g.nchunks = (int)((rows + g.rows_per_chunk - 1) / g.rows_per_chunk);
    return true;
}

__device__ __forceinline__ double bn1_block_sum(double v, double* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// grid (nchunks, groups)
__global__ void __launch_bounds__(256) bn1_stats_kernel(const float* __restrict__ x, Bn1Geom g, double* __restrict__ sums) {
    __shared__ double sm[4];
    const int grp = blockIdx.y;
    const long long rbeg = (long long)blockIdx.x * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const float* base = x + (size_t)grp * g.rows;
    double s1 = 0, s2 = 0;
    const bool vec = (((size_t)grp * g.rows) & 3) == 0;
    if (vec) {
        for (long long r = rbeg + threadIdx.x * 4; r + 3 < rend; r += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(base + r);
            s1 += (double)((v.x + v.y) + (v.z + v.w));
            s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        const long long tail = rbeg + (rend - rbeg) / 4 * 4;
        for (long long r = tail + threadIdx.x; r < rend; r += 256) { const float v = base[r]; s1 += v; s2 += (double)v * v; }
    } else {
        for (long long r = rbeg + threadIdx.x; r < rend; r += 256) { const float v = base[r]; s1 += v; s2 += (double)v * v; }
    }
    s1 = bn1_block_sum(s1, sm);
    s2 = bn1_block_sum(s2, sm);
    if (threadIdx.x == 0) {
        bh_acc_add(&sums[bn_sum_index(0, g.groups, grp, 1, 0, 0)], s1, g.det);
        bh_acc_add(&sums[bn_sum_index(0, g.groups, grp, 1, 0, 1)], s2, g.det);
    }
}

__device__ __forceinline__ void bn1_coeffs(const double* __restrict__ stats, const float* gamma, const float* beta, const float* rmean,
                                           const float* rvar, int use_running, const Bn1Geom& g, int grp, float eps, float& mean,
                                           float& invstd, float& scale, float& shift) {
    if (use_running) { mean = rmean[0]; invstd = 1.0f / sqrtf(rvar[0] + eps); }
    else {
        const double m = bn_sum_total(stats, g.groups, grp, 1, 0, 0, g.det) / (double)g.rows;
        double var = bn_sum_total(stats, g.groups, grp, 1, 0, 1, g.det) / (double)g.rows - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        invstd = 1.0f / sqrtf((float)var + eps);
    }
    scale = (gamma ? gamma[0] : 1.f) * invstd;
    shift = (beta ? beta[0] : 0.f) - mean * scale;
}

// grid (nblk, groups): y = act(x * scale + shift + res)
__global__ void __launch_bounds__(256) bn1_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ rmean,
                                                        const float* __restrict__ rvar, const float* __restrict__ res,
                                                        float* __restrict__ y, const double* __restrict__ stats, Bn1Geom g, float eps,
                                                        int flags, int use_running, float momentum, float* __restrict__ upd_mean,
                                                        float* __restrict__ upd_var) {
    const int grp = blockIdx.y;
    float mean, invstd, sc, sh;
    bn1_coeffs(stats, gamma, beta, rmean, rvar, use_running, g, grp, eps, mean, invstd, sc, sh);
    const bool relu = flags & 1;
    const size_t off = (size_t)grp * g.rows;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < g.rows; r += (long long)gridDim.x * 256) {
        float v = __builtin_fmaf(x[off + r], sc, sh);
        if (res) v += res[off + r];
        if (relu) v = __builtin_elementwise_maximum(v, 0.0f);
        y[off + r] = v;
    }
    if (upd_mean && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // running statistics: one momentum update per group, in group order (consecutive nn.BatchNorm2d calls upstream)
        float rm = upd_mean[0], rv = upd_var[0];
        const double n = (double)g.rows;
        for (int q = 0; q < g.groups; ++q) {
            const double m = bn_sum_total(stats, g.groups, q, 1, 0, 0, g.det) / n;
            double var = bn_sum_total(stats, g.groups, q, 1, 0, 1, g.det) / n - m * m;
            if (var < 0) var = 0;
            const float unb = (float)(n > 1 ? var * n / (n - 1) : var);
            rm = (1.f - momentum) * rm + momentum * (float)m;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        upd_mean[0] = rm; upd_var[0] = rv;
    }
}

// backward, pass 1: per-chunk partial sums of (g * mask, g * mask * xhat).  grid (nchunks, groups); part[groups][nchunks][2]
__global__ void __launch_bounds__(256) bn1_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const double* __restrict__ stats,
                                                             const float* __restrict__ rmean, const float* __restrict__ rvar, Bn1Geom g,
                                                             float eps, int flags, int use_running, double* __restrict__ part) {
    __shared__ double sm[4];
    const int grp = blockIdx.y;
    float mean, invstd, sc, sh;
    bn1_coeffs(stats, gamma, beta, rmean, rvar, use_running, g, grp, eps, mean, invstd, sc, sh);
    const bool relu = flags & 1, from_x = flags & 4;
    const long long rbeg = (long long)blockIdx.x * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const size_t off = (size_t)grp * g.rows;
    double q0 = 0, q1 = 0;
    for (long long r = rbeg + threadIdx.x; r < rend; r += 256) {
        const float xv = x[off + r];
        float gm = gy[off + r];
        if (relu) { const float yv = from_x ? __builtin_fmaf(xv, sc, sh) : y[off + r]; if (!(yv > 0.f)) gm = 0.f; }
        q0 += (double)gm;
        q1 += (double)(gm * ((xv - mean) * invstd));
    }
    q0 = bn1_block_sum(q0, sm);
    q1 = bn1_block_sum(q1, sm);
    if (threadIdx.x == 0) { part[((size_t)grp * g.nchunks + blockIdx.x) * 2] = q0; part[((size_t)grp * g.nchunks + blockIdx.x) * 2 + 1] = q1; }
}

// backward, pass 2.  grid (nblk, groups): gx = A (gm - kb - xhat kg), gres = gm; block (0, 0) adds dgamma / dbeta
__global__ void __launch_bounds__(256) bn1_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const double* __restrict__ stats,
                                                            const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                            const double* __restrict__ part, float* __restrict__ gx,
                                                            float* __restrict__ gres, Bn1Geom g, float eps, int flags, int use_running,
                                                            float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int grp = blockIdx.y;
    float mean, invstd, sc, sh;
    bn1_coeffs(stats, gamma, beta, rmean, rvar, use_running, g, grp, eps, mean, invstd, sc, sh);
    double q0 = 0, q1 = 0;
    for (int c = 0; c < g.nchunks; ++c) { q0 += part[((size_t)grp * g.nchunks + c) * 2]; q1 += part[((size_t)grp * g.nchunks + c) * 2 + 1]; }
    const float A = (gamma ? gamma[0] : 1.f) * invstd;
    const float kb = use_running ? 0.f : (float)(q0 / (double)g.rows), kg = use_running ? 0.f : (float)(q1 / (double)g.rows);
    const bool relu = flags & 1, from_x = flags & 4;
    const size_t off = (size_t)grp * g.rows;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < g.rows; r += (long long)gridDim.x * 256) {
        const float xv = x[off + r];
        float gm = gy[off + r];
        if (relu) { const float yv = from_x ? __builtin_fmaf(xv, sc, sh) : y[off + r]; if (!(yv > 0.f)) gm = 0.f; }
        if (gres) gres[off + r] = gm;
        gx[off + r] = A * (gm - kb - (xv - mean) * invstd * kg);
    }
    if ((ggamma || gbeta) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        double tb = 0, tg = 0;
        for (int q = 0; q < g.groups; ++q)
            for (int c = 0; c < g.nchunks; ++c) { tb += part[((size_t)q * g.nchunks + c) * 2]; tg += part[((size_t)q * g.nchunks + c) * 2 + 1]; }
        if (ggamma) ggamma[0] += (float)tg;
        if (gbeta) gbeta[0] += (float)tb;
    }
}

static int bn1_blocks(const Bn1Geom& g) {
    long long nb = (g.rows + 1023) / 1024;
    const long long cap = 512 / g.groups > 0 ? 512 / g.groups : 1;
    return (int)(nb > cap ? cap : nb);
}

int bn1_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, const float* res, float* y,
            double* stats, int groups, int rows, float eps, float momentum, int flags, int use_running, hipStream_t s) {
    Bn1Geom g;
    if (!bn1_geom(groups, rows, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    if (!use_running && !(flags & 8)) {
        hipLaunchKernelGGL(bn1_stats_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, stats);
        BH_LAUNCH_CHECK();
    }
    const bool upd = !use_running && running_mean && running_var;
    hipLaunchKernelGGL(bn1_apply_kernel, dim3(bn1_blocks(g), groups), dim3(256), 0, s, x, gamma, beta, running_mean, running_var, res, y,
                       stats, g, eps, flags, use_running, momentum, upd ? running_mean : nullptr, upd ? running_var : nullptr);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bn1_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats, float* gx,
            float* gres, float* ggamma, float* gbeta, double* scratch, int groups, int rows, float eps, int flags, int use_running,
            const float* running_mean, const float* running_var, hipStream_t s) {
    Bn1Geom g;
    if (!bn1_geom(groups, rows, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    if (flags & 16) return BH_E_UNSUPPORTED;                 // (no conv epilogue produces the sums of a one-channel BatchNorm)
    hipLaunchKernelGGL(bn1_bwd_reduce_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, gy, y, x, gamma, beta, stats, running_mean,
                       running_var, g, eps, flags, use_running, scratch);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn1_bwd_apply_kernel, dim3(bn1_blocks(g), groups), dim3(256), 0, s, gy, y, x, gamma, beta, stats, running_mean,
                       running_var, (const double*)scratch, gx, gres, g, eps, flags, use_running, ggamma, gbeta);
    BH_LAUNCH_CHECK();
    return BH_OK;
}
