// Training-mode BatchNorm2d on NHWC feature maps ([groups*rows, C], C contiguous), with `groups`
// independent sub-batches stacked along the row axis (one group per reference forward call).
// HBM-bound streaming kernels: float4 per lane along channels, a lane keeps its 4 channels for its
// whole row range, statistics are accumulated in double and combined through per-chunk partials
// (deterministic; no atomics).
//   forward : stats partials -> finalize (mean/var, running stats) -> apply (+residual, +ReLU)
//   backward: reduce partials (sum dy, sum dy*xhat) -> finalize (dgamma/dbeta) -> apply (dx, dres)
#include "common.h"

#define BN_MAX_CHUNKS 256

struct BnGeom {
    int groups, rows, C, C4, LPR, RPP, nchunks, rows_per_chunk;
};

static bool bn_geom(int groups, int rows, int C, BnGeom& g) {
    if (C % 4 || groups < 1 || rows < 1) return false;
    g.groups = groups; g.rows = rows; g.C = C; g.C4 = C / 4;
    if (g.C4 > 256 || (256 % g.C4)) return false;
    g.LPR = g.C4; g.RPP = 256 / g.LPR;
    int n = rows / (g.RPP * 4);          // >= 4 rows per lane per chunk; up to 256 chunks x groups workgroups
    if (n < 1) n = 1;
    if (n > BN_MAX_CHUNKS) n = BN_MAX_CHUNKS;
    g.rows_per_chunk = (rows + n - 1) / n;
    g.nchunks = (rows + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return true;
}

// block-level reduction of NV doubles per thread over threads sharing the same channel quad
template <int NV>
__device__ __forceinline__ void reduce_rows(double (&v)[NV], int LPR, int RPP, double* sm /* [256*NV] */) {
    for (int i = 0; i < NV; ++i) sm[threadIdx.x * NV + i] = v[i];
    __syncthreads();
    if ((int)threadIdx.x < LPR) {
        for (int r = 1; r < RPP; ++r)
            for (int i = 0; i < NV; ++i) v[i] += sm[(threadIdx.x + r * LPR) * NV + i];
    }
}

// grid (nchunks, groups)
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x, BnGeom g, double* __restrict__ part) {
    __shared__ double sm[256 * 8];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const float* base = x + ((size_t)grp * g.rows) * g.C + cq * 4;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rbeg + r0; r < rend; r += g.RPP) {
        float4 a = *reinterpret_cast<const float4*>(base + (size_t)r * g.C);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += (double)a.x * a.x; v[5] += (double)a.y * a.y; v[6] += (double)a.z * a.z; v[7] += (double)a.w * a.w;
    }
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 2;
        for (int i = 0; i < 4; ++i) { p[i * 2] = v[i]; p[i * 2 + 1] = v[4 + i]; }
    }
}

// one wavefront per channel (lanes stride over the chunk partials, shuffle-reduce); groups are processed in
// order so that the running statistics see the same sequence of momentum updates as consecutive
// nn.BatchNorm2d calls
__global__ void __launch_bounds__(64) bn_finalize_kernel(const double* __restrict__ part, BnGeom g, float momentum,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         double* __restrict__ stats) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 1.f;
    const double n = (double)g.rows;
    for (int grp = 0; grp < g.groups; ++grp) {
        double s = 0, ss = 0;
        for (int k = lane; k < g.nchunks; k += 64) {
            const double* p = part + (((size_t)grp * g.nchunks + k) * g.C + c) * 2;
            s += p[0]; ss += p[1];
        }
        s = wave_sum(s); ss = wave_sum(ss);
        const double mean = s / n;
        double var = ss / n - mean * mean;
        if (var < 0) var = 0;
        if (lane == 0) {
            stats[((size_t)grp * g.C + c) * 2] = mean;
            stats[((size_t)grp * g.C + c) * 2 + 1] = var;
        }
        const float unb = (float)(n > 1 ? var * n / (n - 1) : var);
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        rv = (1.f - momentum) * rv + momentum * unb;
    }
    if (lane == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

__device__ __forceinline__ void bn_coeffs(const double* __restrict__ stats, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, const float* __restrict__ rmean,
                                          const float* __restrict__ rvar, int use_running, int grp, int C, int c, float eps,
                                          float& mean, float& invstd, float& scale, float& shift) {
    if (use_running) { mean = rmean[c]; invstd = 1.0f / sqrtf(rvar[c] + eps); }
    else {
        mean = (float)stats[((size_t)grp * C + c) * 2];
        invstd = 1.0f / sqrtf((float)stats[((size_t)grp * C + c) * 2 + 1] + eps);
    }
    const float gm = gamma ? gamma[c] : 1.f;
    scale = gm * invstd;
    shift = (beta ? beta[c] : 0.f) - mean * scale;
}

// grid (nblk, groups)
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ rmean,
                                                       const float* __restrict__ rvar, const float* __restrict__ res,
                                                       float* __restrict__ y, const double* __restrict__ stats, BnGeom g,
                                                       float eps, int flags, int use_running) {
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m, is;
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, grp, g.C, cq * 4 + i, eps, m, is, sc[i], sh[i]);
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, addres = (flags & 2) && res;
    for (int r = blockIdx.x * g.RPP + r0; r < g.rows; r += gridDim.x * g.RPP) {
        const size_t off = gbase + (size_t)r * g.C;
        float4 a = *reinterpret_cast<const float4*>(x + off);
        float4 o = make_float4(a.x * sc[0] + sh[0], a.y * sc[1] + sh[1], a.z * sc[2] + sh[2], a.w * sc[3] + sh[3]);
        if (addres) {
            float4 q = *reinterpret_cast<const float4*>(res + off);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + off) = o;
    }
}

// backward reduce: grid (nchunks, groups)
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const double* __restrict__ stats,
                                                            const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                            BnGeom g, float eps, int flags, int use_running,
                                                            double* __restrict__ part) {
    __shared__ double sm[256 * 8];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    float mean[4], invstd[4], sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, grp, g.C, cq * 4 + i, eps, mean[i], invstd[i], sc[i], sh[i]);
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, mask_from_x = flags & 4;     // bit 2: no residual -> y = relu(x*sc+sh), recompute the mask
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rbeg + r0; r < rend; r += g.RPP) {
        const size_t off = gbase + (size_t)r * g.C;
        float4 d = *reinterpret_cast<const float4*>(gy + off);
        float4 a = *reinterpret_cast<const float4*>(x + off);
        if (relu) {
            float4 o;
            if (mask_from_x) o = make_float4(a.x * sc[0] + sh[0], a.y * sc[1] + sh[1], a.z * sc[2] + sh[2], a.w * sc[3] + sh[3]);
            else o = *reinterpret_cast<const float4*>(y + off);
            if (!(o.x > 0.f)) d.x = 0.f;
            if (!(o.y > 0.f)) d.y = 0.f;
            if (!(o.z > 0.f)) d.z = 0.f;
            if (!(o.w > 0.f)) d.w = 0.f;
        }
        v[0] += d.x; v[1] += d.y; v[2] += d.z; v[3] += d.w;
        v[4] += (double)(d.x * ((a.x - mean[0]) * invstd[0]));
        v[5] += (double)(d.y * ((a.y - mean[1]) * invstd[1]));
        v[6] += (double)(d.z * ((a.z - mean[2]) * invstd[2]));
        v[7] += (double)(d.w * ((a.w - mean[3]) * invstd[3]));
    }
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 2;
        for (int i = 0; i < 4; ++i) { p[i * 2] = v[i]; p[i * 2 + 1] = v[4 + i]; }
    }
}

__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const double* __restrict__ part, BnGeom g,
                                                             float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                             double* __restrict__ sums) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double tg = 0, tb = 0;
    for (int grp = 0; grp < g.groups; ++grp) {
        double s1 = 0, s2 = 0;
        for (int k = lane; k < g.nchunks; k += 64) {
            const double* p = part + (((size_t)grp * g.nchunks + k) * g.C + c) * 2;
            s1 += p[0]; s2 += p[1];
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) {
            sums[((size_t)grp * g.C + c) * 2] = s1;
            sums[((size_t)grp * g.C + c) * 2 + 1] = s2;
        }
        tb += s1; tg += s2;
    }
    if (lane == 0) {
        if (ggamma) ggamma[c] += (float)tg;
        if (gbeta) gbeta[c] += (float)tb;
    }
}

// grid (nblk, groups)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const double* __restrict__ stats, const float* __restrict__ rmean,
                                                           const float* __restrict__ rvar, const double* __restrict__ sums,
                                                           float* __restrict__ gx, float* __restrict__ gres, BnGeom g,
                                                           float eps, int flags, int use_running) {
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    float mean[4], invstd[4], sc[4], sh[4], k1[4], k2[4];
    const float invn = 1.0f / (float)g.rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cq * 4 + i;
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, grp, g.C, c, eps, mean[i], invstd[i], sc[i], sh[i]);
        if (use_running) { k1[i] = 0.f; k2[i] = 0.f; }
        else {
            k1[i] = (float)(sums[((size_t)grp * g.C + c) * 2]) * invn;
            k2[i] = (float)(sums[((size_t)grp * g.C + c) * 2 + 1]) * invn;
        }
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, mask_from_x = flags & 4;
    for (int r = blockIdx.x * g.RPP + r0; r < g.rows; r += gridDim.x * g.RPP) {
        const size_t off = gbase + (size_t)r * g.C;
        float4 d = *reinterpret_cast<const float4*>(gy + off);
        float4 a = *reinterpret_cast<const float4*>(x + off);
        if (relu) {
            float4 o;
            if (mask_from_x) o = make_float4(a.x * sc[0] + sh[0], a.y * sc[1] + sh[1], a.z * sc[2] + sh[2], a.w * sc[3] + sh[3]);
            else o = *reinterpret_cast<const float4*>(y + off);
            if (!(o.x > 0.f)) d.x = 0.f;
            if (!(o.y > 0.f)) d.y = 0.f;
            if (!(o.z > 0.f)) d.z = 0.f;
            if (!(o.w > 0.f)) d.w = 0.f;
        }
        if (gres) *reinterpret_cast<float4*>(gres + off) = d;
        float4 o;
        o.x = sc[0] * (d.x - k1[0] - (a.x - mean[0]) * invstd[0] * k2[0]);
        o.y = sc[1] * (d.y - k1[1] - (a.y - mean[1]) * invstd[1] * k2[1]);
        o.z = sc[2] * (d.z - k1[2] - (a.z - mean[2]) * invstd[2] * k2[2]);
        o.w = sc[3] * (d.w - k1[3] - (a.w - mean[3]) * invstd[3] * k2[3]);
        *reinterpret_cast<float4*>(gx + off) = o;
    }
}

static int apply_blocks(const BnGeom& g) {
    int nb = (g.rows + g.RPP * 4 - 1) / (g.RPP * 4);
    int cap = 2048 / g.groups;
    if (cap < 1) cap = 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" {

int bh_bn_stats_doubles(int groups, int C) { return groups * C * 2 * (1 + BN_MAX_CHUNKS); }

int bh_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
              const float* res, float* y, double* stats, int groups, int rows, int C, float eps, float momentum,
              int flags, int use_running, void* stream) {
    BnGeom g;
    if (!x || !y || !stats) return BH_E_BADARG;
    if (!bn_geom(groups, rows, C, g)) return BH_E_UNSUPPORTED;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    if (!use_running) {
        double* part = stats + (size_t)groups * C * 2;
        hipLaunchKernelGGL(bn_stats_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, part);
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part, g, momentum, running_mean, running_var, stats);
        BH_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, x, gamma, beta, running_mean,
                       running_var, res, y, stats, g, eps, flags, use_running);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats, float* gx,
              float* gres, float* ggamma, float* gbeta, double* scratch, int groups, int rows, int C, float eps, int flags,
              int use_running, const float* running_mean, const float* running_var, void* stream) {
    BnGeom g;
    if (!gy || !x || !gx || !stats || !scratch || ((flags & 1) && !(flags & 4) && !y)) return BH_E_BADARG;
    if ((flags & 4) && (flags & 2)) return BH_E_BADARG;        // the mask can only be recomputed without a residual input
    if (!bn_geom(groups, rows, C, g)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    double* part = scratch + (size_t)groups * C * 2;
    if (!use_running || ggamma || gbeta) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, gy, y, x, gamma, beta, stats,
                           running_mean, running_var, g, eps, flags, use_running, part);
        BH_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, g, ggamma, gbeta, scratch);
        BH_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, gy, y, x, gamma, beta, stats,
                       running_mean, running_var, scratch, gx, gres, g, eps, flags, use_running);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
