// Training-mode BatchNorm2d on NHWC feature maps ([groups*rows, C], C contiguous), with `groups`
// independent sub-batches stacked along the row axis (one group per reference forward call).
// HBM-bound streaming kernels: float4 per lane along channels, a lane keeps its 4 channels for its
// whole row range.  Per-channel sums are accumulated in double: every workgroup reduces its rows and
// adds one double per channel into a caller-zeroed [groups][C][2] buffer with hardware f64 atomics
// (global_atomic_add_f64), so there is no finalize launch - the consumer kernels derive mean / variance
// (forward) and the two gradient sums (backward) from those totals on the fly.
//   forward : [stats sums, unless the producing conv already accumulated them] -> apply (+residual, +ReLU,
//             running statistics updated by workgroup 0)
//   backward: reduce sums (sum dy, sum dy*xhat) -> apply (dx, dres; dgamma/dbeta added by workgroup 0)
#include "common.h"

#define BN_MAX_CHUNKS 256

struct BnGeom {
    int groups, rows, C, C4, LPR, RPP, nchunks, rows_per_chunk;
    int det;               // deterministic mode: cross-workgroup sums through integer limbs (common.h bh_det_add)
};

// det: 1 = this call accumulates / reads through the integer limbs (BH_BN_DETERMINISTIC), 0 = word 0 only, -1 = readers look
static bool bn_geom(int groups, int rows, int C, BnGeom& g, int det) {
    if (C % 4 || groups < 1 || rows < 1) return false;
    g.groups = groups; g.rows = rows; g.C = C; g.C4 = C / 4; g.det = det;
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
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x, BnGeom g, double* __restrict__ sums) {
    __shared__ double sm[256 * 8];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    const float* base = x + ((size_t)grp * g.rows) * g.C + cq * 4;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto accumulate = [&](const float4& a) {
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += (double)a.x * a.x; v[5] += (double)a.y * a.y; v[6] += (double)a.z * a.z; v[7] += (double)a.w * a.w;
    };
    int r = rbeg + r0;
    for (; r + 7 * g.RPP < rend; r += 8 * g.RPP) {          // eight rows per lane in flight
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(base + (size_t)(r + u * g.RPP) * g.C);
#pragma unroll
        for (int u = 0; u < 8; ++u) accumulate(a[u]);
    }
    for (; r < rend; r += g.RPP) accumulate(*reinterpret_cast<const float4*>(base + (size_t)r * g.C));
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        const int slot = blockIdx.x % BH_BN_SUM_SLOTS;
        for (int i = 0; i < 4; ++i) {
            bh_acc_add(&sums[bn_sum_index(slot, g.groups, grp, g.C, cq * 4 + i, 0)], v[i], g.det);
            bh_acc_add(&sums[bn_sum_index(slot, g.groups, grp, g.C, cq * 4 + i, 1)], v[4 + i], g.det);
        }
    }
}

// running statistics: groups are processed in order so that the buffers see the same sequence of momentum
// updates as consecutive nn.BatchNorm2d calls (one call per group upstream)
__device__ __forceinline__ void bn_update_running(const double* __restrict__ sums, const BnGeom& g, float momentum,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var) {
    const double n = (double)g.rows;
    for (int c = threadIdx.x; c < g.C; c += blockDim.x) {
        float rm = running_mean[c], rv = running_var[c];
        for (int grp = 0; grp < g.groups; ++grp) {
            const double mean = bn_sum_total(sums, g.groups, grp, g.C, c, 0, g.det) / n;
            double var = bn_sum_total(sums, g.groups, grp, g.C, c, 1, g.det) / n - mean * mean;
            if (var < 0) var = 0;
            const float unb = (float)(n > 1 ? var * n / (n - 1) : var);
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

__device__ __forceinline__ void bn_coeffs(const double* __restrict__ stats, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, const float* __restrict__ rmean,
                                          const float* __restrict__ rvar, int use_running, int groups, int grp, int C, int c,
                                          float eps, double rows, float& mean, float& invstd, float& scale, float& shift, int det = -1) {
    if (use_running) { mean = rmean[c]; invstd = 1.0f / sqrtf(rvar[c] + eps); }
    else {                                                      // stats = [groups][C][2] sums (x, x^2) over `rows` rows
        const double m = bn_sum_total(stats, groups, grp, C, c, 0, det) / rows;
        double var = bn_sum_total(stats, groups, grp, C, c, 1, det) / rows - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        invstd = 1.0f / sqrtf((float)var + eps);
    }
    const float gm = gamma ? gamma[c] : 1.f;
    scale = gm * invstd;
    shift = (beta ? beta[c] : 0.f) - mean * scale;
}

// The same in two phases, so that a streaming kernel can put its first data loads BETWEEN the (tiny) table loads and the
// arithmetic: VALU work is not hidden by other waves on this chip, and with eight waves per SIMD starting together the
// ~300 instructions of four channels' coefficients used to delay the first load of every wave by microseconds.
struct BnRaw { double s1, s2; float gm, bt; };
__device__ __forceinline__ BnRaw bn_raw(const double* __restrict__ stats, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, const float* __restrict__ rmean,
                                        const float* __restrict__ rvar, int use_running, int groups, int grp, int C, int c, int det = -1) {
    BnRaw r;
    if (use_running) { r.s1 = rmean[c]; r.s2 = rvar[c]; }
    else { r.s1 = bn_sum_total(stats, groups, grp, C, c, 0, det); r.s2 = bn_sum_total(stats, groups, grp, C, c, 1, det); }
    r.gm = gamma ? gamma[c] : 1.f;
    r.bt = beta ? beta[c] : 0.f;
    return r;
}
__device__ __forceinline__ void bn_coeffs_from_raw(const BnRaw& r, int use_running, float eps, double inv_rows, float& mean,
                                                   float& invstd, float& scale, float& shift) {
    if (use_running) { mean = (float)r.s1; invstd = 1.0f / sqrtf((float)r.s2 + eps); }
    else {
        const double m = r.s1 * inv_rows;
        double var = r.s2 * inv_rows - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        invstd = 1.0f / sqrtf((float)var + eps);
    }
    scale = r.gm * invstd;
    shift = r.bt - mean * scale;
}

// grid (nblk, groups)
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ rmean,
                                                       const float* __restrict__ rvar, const float* __restrict__ res,
                                                       float* __restrict__ y, const double* __restrict__ stats, BnGeom g,
                                                       float eps, int flags, int use_running, float momentum,
                                                       float* __restrict__ upd_mean, float* __restrict__ upd_var,
                                                       unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    float vmax = 0.f;                                         // amax != NULL: max |y| of this thread's elements (magnitude record, common.h F16X2)
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    BnRaw raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[i] = bn_raw(stats, gamma, beta, rmean, rvar, use_running, g.groups, grp, g.C, cq * 4 + i, g.det);
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, addres = (flags & 2) && res;
    // four rows per lane and pass; the loads of the first pass are issued before the coefficient arithmetic
    const int stride = gridDim.x * g.RPP;
    int r = blockIdx.x * g.RPP + r0;
    float4 a[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rr = r + u * stride;
        a[u] = rr < g.rows ? *reinterpret_cast<const float4*>(x + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[u] = (addres && rr < g.rows) ? *reinterpret_cast<const float4*>(res + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sc[4], sh[4];
    const double inv_rows = 1.0 / (double)g.rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m, is;
        bn_coeffs_from_raw(raw[i], use_running, eps, inv_rows, m, is, sc[i], sh[i]);
    }
    if (upd_mean && blockIdx.x == 0 && blockIdx.y == 0) bn_update_running(stats, g, momentum, upd_mean, upd_var);
    while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            float4 o = make_float4(a[u].x * sc[0] + sh[0], a[u].y * sc[1] + sh[1], a[u].z * sc[2] + sh[2], a[u].w * sc[3] + sh[3]);
            if (addres) { o.x += q[u].x; o.y += q[u].y; o.z += q[u].z; o.w += q[u].w; }
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(y + gbase + (size_t)rr * g.C) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        r += 4 * stride;
        if (r >= g.rows) break;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            a[u] = rr < g.rows ? *reinterpret_cast<const float4*>(x + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
            q[u] = (addres && rr < g.rows) ? *reinterpret_cast<const float4*>(res + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
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
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, g.groups, grp, g.C, cq * 4 + i, eps, (double)g.rows, mean[i], invstd[i], sc[i], sh[i], g.det);
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, mask_from_x = flags & 4;     // bit 2: no residual -> y = relu(x*sc+sh), recompute the mask
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool need_y = relu && !mask_from_x;
    auto accumulate = [&](float4 d, const float4& a, const float4& yv) {
        if (relu) {
            float4 o;
            if (mask_from_x) o = make_float4(a.x * sc[0] + sh[0], a.y * sc[1] + sh[1], a.z * sc[2] + sh[2], a.w * sc[3] + sh[3]);
            else o = yv;
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
    };
    // four rows per lane in flight (a one-row loop keeps 2-3 loads outstanding per lane and runs at HBM latency, not
    // bandwidth: 37 us for two 33.5 MB tensors)
    int r = rbeg + r0;
    for (; r + 3 * g.RPP < rend; r += 4 * g.RPP) {
        float4 d[4], a[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t off = gbase + (size_t)(r + u * g.RPP) * g.C;
            d[u] = *reinterpret_cast<const float4*>(gy + off);
            a[u] = *reinterpret_cast<const float4*>(x + off);
            yv[u] = need_y ? *reinterpret_cast<const float4*>(y + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accumulate(d[u], a[u], yv[u]);
    }
    for (; r < rend; r += g.RPP) {
        const size_t off = gbase + (size_t)r * g.C;
        const float4 d = *reinterpret_cast<const float4*>(gy + off);
        const float4 a = *reinterpret_cast<const float4*>(x + off);
        const float4 yv = need_y ? *reinterpret_cast<const float4*>(y + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        accumulate(d, a, yv);
    }
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 2;
        for (int i = 0; i < 4; ++i) { p[i * 2] = v[i]; p[i * 2 + 1] = v[4 + i]; }
    }
}

// one wavefront per channel: totals of the chunk partials (deterministic), dgamma / dbeta, and the compact
// coefficient table the apply kernel starts from: coef[grp][c] = (mean, invstd, mean(dy), mean(dy * xhat))
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const double* __restrict__ part, BnGeom g,
                                                             const double* __restrict__ stats, const float* __restrict__ rmean,
                                                             const float* __restrict__ rvar, float eps, int use_running,
                                                             float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                             float4* __restrict__ coef) {
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
            float mean, invstd, sc, sh;
            bn_coeffs(stats, nullptr, nullptr, rmean, rvar, use_running, g.groups, grp, g.C, c, eps, (double)g.rows, mean, invstd, sc, sh, g.det);
            const float invn = 1.0f / (float)g.rows;
            coef[(size_t)grp * g.C + c] = use_running ? make_float4(mean, invstd, 0.f, 0.f)
                                                      : make_float4(mean, invstd, (float)s1 * invn, (float)s2 * invn);
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
                                                           const float4* __restrict__ coef,
                                                           float* __restrict__ gx, float* __restrict__ gres, BnGeom g,
                                                           int flags, const double* __restrict__ stats,
                                                           const double* __restrict__ sums, float eps,
                                                           float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                           unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    float vmax = 0.f;                                         // amax != NULL: max |gx| (magnitude record of the gradient, common.h F16X2)
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    // table loads first, then the data loads of the first pass, then the coefficient arithmetic (see bn_raw)
    BnRaw raw[4];
    float4 cf4[4];
    double rs1[4], rs2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cq * 4 + i;
        cf4[i] = make_float4(0.f, 0.f, 0.f, 0.f); rs1[i] = 0; rs2[i] = 0;
        if (coef) {
            cf4[i] = coef[(size_t)grp * g.C + c];        // (mean, invstd, mean dy, mean dy*xhat) from the finalize kernel
            raw[i].gm = gamma ? gamma[c] : 1.f; raw[i].bt = beta ? beta[c] : 0.f; raw[i].s1 = 0; raw[i].s2 = 0;
        } else {
            // the gradient sums were accumulated by the dgrad that produced gy (bh_conv_dgrad_bnreduce): no reduce /
            // finalize launches, the coefficients come straight from the two padded sums tables
            raw[i] = bn_raw(stats, gamma, beta, nullptr, nullptr, 0, g.groups, grp, g.C, c, g.det);
            rs1[i] = bn_sum_total(sums, g.groups, grp, g.C, c, 0, g.det);
            rs2[i] = bn_sum_total(sums, g.groups, grp, g.C, c, 1, g.det);
        }
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const bool relu = flags & 1, mask_from_x = flags & 4;
    const bool rd_y = relu && !mask_from_x;
    const int stride = gridDim.x * g.RPP;
    int r = blockIdx.x * g.RPP + r0;
    float4 dv[4], av[4], yv[4];
    auto issue = [&](int rb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = rb + u * stride;
            const size_t off = gbase + (size_t)rr * g.C;
            const bool ok = rr < g.rows;
            dv[u] = ok ? *reinterpret_cast<const float4*>(gy + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            av[u] = ok ? *reinterpret_cast<const float4*>(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            yv[u] = (ok && rd_y) ? *reinterpret_cast<const float4*>(y + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    issue(r);
    float mean[4], invstd[4], sc[4], sh[4], k1[4], k2[4];
    const double inv_rows = 1.0 / (double)g.rows;
    const float invn = 1.0f / (float)g.rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (coef) {
            mean[i] = cf4[i].x; invstd[i] = cf4[i].y; k1[i] = cf4[i].z; k2[i] = cf4[i].w;
            sc[i] = raw[i].gm * cf4[i].y;
            sh[i] = raw[i].bt - cf4[i].x * sc[i];
        } else {
            bn_coeffs_from_raw(raw[i], 0, eps, inv_rows, mean[i], invstd[i], sc[i], sh[i]);
            k1[i] = (float)rs1[i] * invn;
            k2[i] = (float)rs2[i] * invn;
        }
    }
    if (!coef && (ggamma || gbeta) && blockIdx.x == 0 && blockIdx.y == 0) {       // dgamma / dbeta: totals over the groups
        for (int c = threadIdx.x; c < g.C; c += blockDim.x) {
            double tb = 0, tg = 0;
            for (int q = 0; q < g.groups; ++q) {
                tb += bn_sum_total(sums, g.groups, q, g.C, c, 0, g.det);
                tg += bn_sum_total(sums, g.groups, q, g.C, c, 1, g.det);
            }
            if (ggamma) ggamma[c] += (float)tg;
            if (gbeta) gbeta[c] += (float)tb;
        }
    }
    while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            const size_t off = gbase + (size_t)rr * g.C;
            float4 d = dv[u];
            const float4 a = av[u];
            if (relu) {
                float4 o;
                if (mask_from_x) o = make_float4(a.x * sc[0] + sh[0], a.y * sc[1] + sh[1], a.z * sc[2] + sh[2], a.w * sc[3] + sh[3]);
                else o = yv[u];
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
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        r += 4 * stride;
        if (r >= g.rows) break;
        issue(r);
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one pass (round 4: the stem of the backbone and of the perceptual extractor,
// Rethinking.py:31-36 / torchvision resnet conv1-bn1-relu-maxpool): the 64 x 64 x 64 activation between BatchNorm and pooling
// (134 MB at 128 images) is never written or read - the adjoint needs the BatchNorm INPUT only (ReLU mask recomputed) and the
// arg-max positions.  y[N,Ho,Wo,C], idx: window position (0..8) of the first maximum per element (the ATen tie rule, as
// maxpool_fwd_kernel).  grid-stride over output float4s; coefficient table [groups][C] x (scale, shift) in LDS.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ rmean,
                                                             const float* __restrict__ rvar, const double* __restrict__ stats,
                                                             float* __restrict__ y, unsigned char* __restrict__ idx, BnGeom g, int N,
                                                             int Hi, int Wi, int Ho, int Wo, float eps, int relu, int use_running,
                                                             float momentum, float* __restrict__ upd_mean, float* __restrict__ upd_var,
                                                             unsigned* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float tb[];          // [groups][C][2]
    __shared__ float sm_amax[4];
    for (int e = threadIdx.x; e < g.groups * g.C; e += 256) {
        float m, is, sc, sh;
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, g.groups, e / g.C, g.C, e % g.C, eps, (double)g.rows, m, is, sc, sh, g.det);
        tb[2 * e] = sc; tb[2 * e + 1] = sh;
    }
    __syncthreads();
    if (upd_mean && blockIdx.x == 0) bn_update_running(stats, g, momentum, upd_mean, upd_var);
    const int C4 = g.C4, ipg = N / g.groups;
    const float lo = relu ? 0.0f : -INFINITY;
    float vmax = 0.f;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float4* t4 = reinterpret_cast<const float4*>(tb + ((size_t)(n / ipg) * g.C + c * 4) * 2);
        const float4 t0 = t4[0], t1 = t4[1];                           // (sc0, sh0, sc1, sh1), (sc2, sh2, sc3, sh3)
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 w = make_uchar4(0, 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= Wi) continue;
                const unsigned char t = (unsigned char)(ky * 3 + kx);
                float4 v = *reinterpret_cast<const float4*>(x + ((((size_t)n * Hi + iy) * Wi + ix) * C4 + c) * 4);
                v.x = fmaxf(__builtin_fmaf(v.x, t0.x, t0.y), lo); v.y = fmaxf(__builtin_fmaf(v.y, t0.z, t0.w), lo);
                v.z = fmaxf(__builtin_fmaf(v.z, t1.x, t1.y), lo); v.w = fmaxf(__builtin_fmaf(v.w, t1.z, t1.w), lo);
                if (v.x > m.x) { m.x = v.x; w.x = t; }
                if (v.y > m.y) { m.y = v.y; w.y = t; }
                if (v.z > m.z) { m.z = v.z; w.z = t; }
                if (v.w > m.w) { m.w = v.w; w.w = t; }
            }
        }
        *reinterpret_cast<float4*>(y + i * 4) = m;
        if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = w;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(m.x), fabsf(m.y))), fmaxf(fabsf(m.z), fabsf(m.w)));
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x, sm_amax);
}

// ---------------------------------------------------------------------------------------------
// Two-branch join (round 4): y = relu(bn_a(xa) + bn_b(xb)) - the end of every residual unit whose lower branch ends in its own
// BatchNorm (ResNet50DeconvBlock / the strided ResNet34ConvBlock, src/backbones/utils.py:60-82,85-112).  Unfused that is
// bn(xb) -> l (read + write), bn(xa, res = l) -> y (two reads + write): the normalised lower branch is written and read back
// for nothing.  Here both BatchNorms are applied in ONE pass (two reads, one write), and the adjoint - d = gy [y > 0],
// gxa = sc_a (d - k1 - xhat_a k2a), gxb = sc_b (d - k1 - xhat_b k2b) with the SHARED k1 = mean d - is one reduce pass over
// (gy, y, xa, xb) for the three sums and one apply pass, instead of reduce + apply per branch with the masked gradient
// written and read in between.  grid (nblk, groups)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_join_apply_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                            const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                                            const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                                            const double* __restrict__ stats_a, const double* __restrict__ stats_b,
                                                            float* __restrict__ y, BnGeom g, float eps_a, float eps_b, int relu,
                                                            float mom_a, float mom_b, float* __restrict__ rm_a, float* __restrict__ rv_a,
                                                            float* __restrict__ rm_b, float* __restrict__ rv_b, unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    float vmax = 0.f;
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    BnRaw ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = bn_raw(stats_a, gamma_a, beta_a, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, g.det);
        rb[i] = bn_raw(stats_b, gamma_b, beta_b, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, g.det);
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const int stride = gridDim.x * g.RPP;
    int r = blockIdx.x * g.RPP + r0;
    float4 a[4], q[4];
    auto issue = [&](int rbase) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = rbase + u * stride;
            const bool ok = rr < g.rows;
            a[u] = ok ? *reinterpret_cast<const float4*>(xa + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
            q[u] = ok ? *reinterpret_cast<const float4*>(xb + gbase + (size_t)rr * g.C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    issue(r);
    float sca[4], scb[4], sh[4];
    const double inv_rows = 1.0 / (double)g.rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m, is, sha, shb;
        bn_coeffs_from_raw(ra[i], 0, eps_a, inv_rows, m, is, sca[i], sha);
        bn_coeffs_from_raw(rb[i], 0, eps_b, inv_rows, m, is, scb[i], shb);
        sh[i] = sha + shb;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (rm_a) bn_update_running(stats_a, g, mom_a, rm_a, rv_a);
        if (rm_b) bn_update_running(stats_b, g, mom_b, rm_b, rv_b);
    }
    while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            float4 o = make_float4(__builtin_fmaf(a[u].x, sca[0], __builtin_fmaf(q[u].x, scb[0], sh[0])),
                                   __builtin_fmaf(a[u].y, sca[1], __builtin_fmaf(q[u].y, scb[1], sh[1])),
                                   __builtin_fmaf(a[u].z, sca[2], __builtin_fmaf(q[u].z, scb[2], sh[2])),
                                   __builtin_fmaf(a[u].w, sca[3], __builtin_fmaf(q[u].w, scb[3], sh[3])));
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(y + gbase + (size_t)rr * g.C) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        r += 4 * stride;
        if (r >= g.rows) break;
        issue(r);
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
}

// reduce: per chunk (sum d, sum d xhat_a, sum d xhat_b) per channel, d = gy [y > 0] (relu) - grid (nchunks, groups); part[grp][chunk][C][3]
__global__ void __launch_bounds__(256) bn_join_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                 const float* __restrict__ xa, const float* __restrict__ xb,
                                                                 const double* __restrict__ stats_a, const double* __restrict__ stats_b,
                                                                 BnGeom g, float eps_a, float eps_b, int relu, double* __restrict__ part,
                                                                 const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                                                 const float* __restrict__ gamma_b, const float* __restrict__ beta_b, int remask) {
    __shared__ double sm[256 * 12];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    float ma[4], ia[4], mb[4], ib[4], t0, t1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bn_coeffs(stats_a, nullptr, nullptr, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, eps_a, (double)g.rows, ma[i], ia[i], t0, t1, g.det);
        bn_coeffs(stats_b, nullptr, nullptr, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, eps_b, (double)g.rows, mb[i], ib[i], t0, t1, g.det);
    }
    // remask (round 5): the ReLU mask is recomputed from xa, xb with the forward kernel's own arithmetic (bn_join_apply_kernel: the same
    // coefficient expressions, the same two FMAs) - y, a fourth 134 MB stream at the full-resolution join, is not read
    float sca[4] = {0.f, 0.f, 0.f, 0.f}, scb[4] = {0.f, 0.f, 0.f, 0.f}, shm[4] = {0.f, 0.f, 0.f, 0.f};
    if (remask) {
        const double inv_rows = 1.0 / (double)g.rows;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const BnRaw ra = bn_raw(stats_a, gamma_a, beta_a, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, g.det);
            const BnRaw rb = bn_raw(stats_b, gamma_b, beta_b, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, g.det);
            float m, is, sha, shb;
            bn_coeffs_from_raw(ra, 0, eps_a, inv_rows, m, is, sca[i], sha);
            bn_coeffs_from_raw(rb, 0, eps_b, inv_rows, m, is, scb[i], shb);
            shm[i] = sha + shb;
        }
    }
    const bool rd_y = relu && !remask;
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    double v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto accumulate = [&](float4 d, float4 yv, const float4& a, const float4& b) {
        if (remask) yv = make_float4(__builtin_fmaf(a.x, sca[0], __builtin_fmaf(b.x, scb[0], shm[0])), __builtin_fmaf(a.y, sca[1], __builtin_fmaf(b.y, scb[1], shm[1])),
                                     __builtin_fmaf(a.z, sca[2], __builtin_fmaf(b.z, scb[2], shm[2])), __builtin_fmaf(a.w, sca[3], __builtin_fmaf(b.w, scb[3], shm[3])));
        if (relu) {
            if (!(yv.x > 0.f)) d.x = 0.f;
            if (!(yv.y > 0.f)) d.y = 0.f;
            if (!(yv.z > 0.f)) d.z = 0.f;
            if (!(yv.w > 0.f)) d.w = 0.f;
        }
        v[0] += d.x; v[1] += d.y; v[2] += d.z; v[3] += d.w;
        v[4] += (double)(d.x * ((a.x - ma[0]) * ia[0])); v[5] += (double)(d.y * ((a.y - ma[1]) * ia[1]));
        v[6] += (double)(d.z * ((a.z - ma[2]) * ia[2])); v[7] += (double)(d.w * ((a.w - ma[3]) * ia[3]));
        v[8] += (double)(d.x * ((b.x - mb[0]) * ib[0])); v[9] += (double)(d.y * ((b.y - mb[1]) * ib[1]));
        v[10] += (double)(d.z * ((b.z - mb[2]) * ib[2])); v[11] += (double)(d.w * ((b.w - mb[3]) * ib[3]));
    };
    int r = rbeg + r0;
    for (; r + 3 * g.RPP < rend; r += 4 * g.RPP) {
        float4 d[4], yv[4], a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t off = gbase + (size_t)(r + u * g.RPP) * g.C;
            d[u] = *reinterpret_cast<const float4*>(gy + off);
            yv[u] = rd_y ? *reinterpret_cast<const float4*>(y + off) : make_float4(1.f, 1.f, 1.f, 1.f);
            a[u] = *reinterpret_cast<const float4*>(xa + off);
            b[u] = *reinterpret_cast<const float4*>(xb + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accumulate(d[u], yv[u], a[u], b[u]);
    }
    for (; r < rend; r += g.RPP) {
        const size_t off = gbase + (size_t)r * g.C;
        accumulate(*reinterpret_cast<const float4*>(gy + off), rd_y ? *reinterpret_cast<const float4*>(y + off) : make_float4(1.f, 1.f, 1.f, 1.f),
                   *reinterpret_cast<const float4*>(xa + off), *reinterpret_cast<const float4*>(xb + off));
    }
    reduce_rows<12>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 3;
        for (int i = 0; i < 4; ++i) { p[i * 3] = v[i]; p[i * 3 + 1] = v[4 + i]; p[i * 3 + 2] = v[8 + i]; }
    }
}

// one wavefront per channel: totals (deterministic), dgamma / dbeta of both BatchNorms, coef[grp][c] = (mean d, mean d xhat_a, mean d xhat_b, 0)
__global__ void __launch_bounds__(64) bn_join_bwd_finalize_kernel(const double* __restrict__ part, BnGeom g, float* __restrict__ gg_a,
                                                                  float* __restrict__ gb_a, float* __restrict__ gg_b, float* __restrict__ gb_b,
                                                                  float4* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double t1 = 0, t2a = 0, t2b = 0;
    for (int grp = 0; grp < g.groups; ++grp) {
        double s1 = 0, s2a = 0, s2b = 0;
        for (int k = lane; k < g.nchunks; k += 64) {
            const double* p = part + (((size_t)grp * g.nchunks + k) * g.C + c) * 3;
            s1 += p[0]; s2a += p[1]; s2b += p[2];
        }
        s1 = wave_sum(s1); s2a = wave_sum(s2a); s2b = wave_sum(s2b);
        if (lane == 0) {
            const float invn = 1.0f / (float)g.rows;
            coef[(size_t)grp * g.C + c] = make_float4((float)s1 * invn, (float)s2a * invn, (float)s2b * invn, 0.f);
        }
        t1 += s1; t2a += s2a; t2b += s2b;
    }
    if (lane == 0) {
        if (gg_a) gg_a[c] += (float)t2a;
        if (gb_a) gb_a[c] += (float)t1;
        if (gg_b) gg_b[c] += (float)t2b;
        if (gb_b) gb_b[c] += (float)t1;
    }
}

// apply: grid (nblk, groups)
__global__ void __launch_bounds__(256) bn_join_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                const float* __restrict__ xa, const float* __restrict__ xb,
                                                                const float* __restrict__ gamma_a, const float* __restrict__ gamma_b,
                                                                const double* __restrict__ stats_a, const double* __restrict__ stats_b,
                                                                const float4* __restrict__ coef, float* __restrict__ gxa,
                                                                float* __restrict__ gxb, BnGeom g, float eps_a, float eps_b, int relu,
                                                                unsigned* __restrict__ amax_a, unsigned* __restrict__ amax_b,
                                                                const float* __restrict__ beta_a, const float* __restrict__ beta_b, int remask) {
    __shared__ float sm_amax[4];
    float vmax_a = 0.f, vmax_b = 0.f;
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    BnRaw ra[4], rb[4];
    float4 cf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cq * 4 + i;
        ra[i] = bn_raw(stats_a, gamma_a, remask ? beta_a : nullptr, nullptr, nullptr, 0, g.groups, grp, g.C, c, g.det);
        rb[i] = bn_raw(stats_b, gamma_b, remask ? beta_b : nullptr, nullptr, nullptr, 0, g.groups, grp, g.C, c, g.det);
        cf[i] = coef[(size_t)grp * g.C + c];
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const int stride = gridDim.x * g.RPP;
    int r = blockIdx.x * g.RPP + r0;
    float4 dv[4], yv[4], av[4], bv[4];
    auto issue = [&](int rbase) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = rbase + u * stride;
            const size_t off = gbase + (size_t)rr * g.C;
            const bool ok = rr < g.rows;
            // (loads under an `if`, not `ok ? *p : z` with a NAMED zero: the compiler made that a select between the global address and
            //  the address of z in scratch memory, i.e. flat loads and a scratch store per operand - 32 bytes of private segment)
            dv[u] = make_float4(0.f, 0.f, 0.f, 0.f); av[u] = dv[u]; bv[u] = dv[u];
            yv[u] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (ok) {
                dv[u] = *reinterpret_cast<const float4*>(gy + off);
                if (relu && !remask) yv[u] = *reinterpret_cast<const float4*>(y + off);
                av[u] = *reinterpret_cast<const float4*>(xa + off);
                bv[u] = *reinterpret_cast<const float4*>(xb + off);
            }
        }
    };
    issue(r);
    float ma[4], ia[4], sca[4], mb[4], ib[4], scb[4], shm[4];
    const double inv_rows = 1.0 / (double)g.rows;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float sha, shb;
        bn_coeffs_from_raw(ra[i], 0, eps_a, inv_rows, ma[i], ia[i], sca[i], sha);
        bn_coeffs_from_raw(rb[i], 0, eps_b, inv_rows, mb[i], ib[i], scb[i], shb);
        shm[i] = sha + shb;                                    // (remask: the forward kernel's shift - with beta = NULL it is not used)
    }
    while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            const size_t off = gbase + (size_t)rr * g.C;
            const float d0[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
            const float y0[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
            const float a[4] = {av[u].x, av[u].y, av[u].z, av[u].w}, b[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
            float oa[4], ob[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (scalars, not arrays written under a condition: those ended up in scratch memory - 32 bytes of private segment)
                const float yi = remask ? __builtin_fmaf(a[i], sca[i], __builtin_fmaf(b[i], scb[i], shm[i])) : y0[i];
                const float di = (relu && !(yi > 0.f)) ? 0.f : d0[i];
                const float e = di - cf[i].x;
                oa[i] = sca[i] * (e - (a[i] - ma[i]) * ia[i] * cf[i].y);
                ob[i] = scb[i] * (e - (b[i] - mb[i]) * ib[i] * cf[i].z);
                vmax_a = fmaxf(vmax_a, fabsf(oa[i])); vmax_b = fmaxf(vmax_b, fabsf(ob[i]));
            }
            *reinterpret_cast<float4*>(gxa + off) = make_float4(oa[0], oa[1], oa[2], oa[3]);
            *reinterpret_cast<float4*>(gxb + off) = make_float4(ob[0], ob[1], ob[2], ob[3]);
        }
        r += 4 * stride;
        if (r >= g.rows) break;
        issue(r);
    }
    if (amax_a) bh_amax_commit(amax_a, vmax_a, blockIdx.x + blockIdx.y * 7u, sm_amax);
    if (amax_b) { __syncthreads(); bh_amax_commit(amax_b, vmax_b, blockIdx.x + blockIdx.y * 5u, sm_amax); }
}

BH_KNOB(g_bn_apply_cap, 512);        // workgroups per apply launch: two per CU that loop beat a one-shot grid of 2048 by 15-30 % (tools/bn_apply_sweep.py; tuning: bh_debug_force_tile(-20, n))
#ifdef BH_TUNING
void bh_bn_tune(int cap) { g_bn_apply_cap = cap; }
#endif
static int apply_blocks(const BnGeom& g) {
    int nb = (g.rows + g.RPP * 4 - 1) / (g.RPP * 4);
    int cap = g_bn_apply_cap / g.groups;
    if (cap < 1) cap = 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

// Coefficient table of a training-mode BatchNorm whose APPLY rides in its consumer (bh_conv_fwd_bnin / bh_conv_wgrad_bnin):
// table[grp][c] = (scale, shift) with y = x * scale + shift, from the forward sums; workgroup 0 also makes the running-statistics
// update that bh_bn_fwd would have made.  grid ceil(groups * C / 256).
__global__ void __launch_bounds__(256) bn_fwd_coeffs_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, BnGeom g, float eps, float momentum,
                                                            float* __restrict__ upd_mean, float* __restrict__ upd_var,
                                                            float2* __restrict__ table, unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    // (the running-statistics update has a workgroup of its own - the LAST one: behind the table it doubled the length of a launch that sits
    //  between every conv and its BatchNorm-on-load consumer, 27 times per step)
    if (upd_mean && blockIdx.x == gridDim.x - 1) { bn_update_running(stats, g, momentum, upd_mean, upd_var); return; }
    const int i = blockIdx.x * 256 + threadIdx.x;
    float bound = 0.f;
    if (i < g.groups * g.C) {
        const int grp = i / g.C, c = i - grp * g.C;
        float mean, invstd, sc, sh;
        bn_coeffs(stats, gamma, beta, nullptr, nullptr, 0, g.groups, grp, g.C, c, eps, (double)g.rows, mean, invstd, sc, sh, g.det);
        table[i] = make_float2(sc, sh);
        // |y| = |gamma xhat + beta| <= |gamma| sqrt(rows - 1) + |beta| for ANY data (a sample is at most sqrt(rows - 1) standard deviations
        // from the mean of its rows); one binade of margin for the rounding of the sums and of the fused multiply-add
        bound = 2.0f * (fabsf(gamma ? gamma[c] : 1.f) * sqrtf((float)g.rows) + fabsf(beta ? beta[c] : 0.f));
    }
    if (amax) bh_amax_commit(amax, bound, blockIdx.x, sm_amax);
}

// per-channel sums of an NHWC tensor into caller-zeroed sums[groups][C][2] (used by bh_bn_fwd and by the conv entry
// point that hands the statistics to the BatchNorm that follows it)
int bn_launch_stats(const float* x, int groups, int rows, int C, double* sums, hipStream_t s, int det) {
    BnGeom g;
    if (!bn_geom(groups, rows, C, g, det ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, sums);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

// C == 1 (bn1.hip)
int bn1_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, const float* res, float* y,
            double* stats, int groups, int rows, float eps, float momentum, int flags, int use_running, hipStream_t s);
int bn1_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats, float* gx,
            float* gres, float* ggamma, float* gbeta, double* scratch, int groups, int rows, float eps, int flags, int use_running,
            const float* running_mean, const float* running_var, hipStream_t s);

extern "C" {

int bh_bn_stats_doubles(int groups, int C) { return (int)BH_BN_SUM_DOUBLES(groups, C); }

int bh_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
              const float* res, float* y, double* stats, int groups, int rows, int C, float eps, float momentum,
              int flags, int use_running, void* stream) {
    return bh_bn_fwd_amax(x, gamma, beta, running_mean, running_var, res, y, stats, groups, rows, C, eps, momentum, flags, use_running, nullptr, stream);
}

int bh_bn_fwd_amax(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   const float* res, float* y, double* stats, int groups, int rows, int C, float eps, float momentum,
                   int flags, int use_running, float* amax_y, void* stream) {
    BnGeom g;
    if (!x || !y || !stats) return BH_E_BADARG;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    if (C == 1) return amax_y ? BH_E_UNSUPPORTED : bn1_fwd(x, gamma, beta, running_mean, running_var, res, y, stats, groups, rows, eps, momentum, flags, use_running, bh_stream(stream));
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    if (!use_running && !(flags & 8)) {                       // bit 3: the producer already accumulated the sums
        hipLaunchKernelGGL(bn_stats_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, stats);
        BH_LAUNCH_CHECK();
    }
    const bool upd = !use_running && running_mean && running_var;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, x, gamma, beta, running_mean,
                       running_var, res, y, stats, g, eps, flags, use_running, momentum, upd ? running_mean : nullptr,
                       upd ? running_var : nullptr, reinterpret_cast<unsigned*>(amax_y));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_fwd_coeffs(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var, int groups,
                     int rows, int C, float eps, float momentum, float* table, void* stream) {
    return bh_bn_fwd_coeffs_amax(stats, gamma, beta, running_mean, running_var, groups, rows, C, eps, momentum, table, nullptr, stream);
}

int bh_bn_fwd_coeffs_amax(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var, int groups,
                          int rows, int C, float eps, float momentum, float* table, float* amax_y, void* stream) {
    BnGeom g;
    if (!stats || !table) return BH_E_BADARG;
    if (!bn_geom(groups, rows, C, g, -1)) return BH_E_UNSUPPORTED;        // (a reader: looks at the limbs of whatever mode wrote the sums)
    const bool upd = running_mean && running_var;
    hipLaunchKernelGGL(bn_fwd_coeffs_kernel, dim3((groups * C + 255) / 256 + (upd ? 1 : 0)), dim3(256), 0, bh_stream(stream), stats, gamma, beta, g, eps,
                       momentum, upd ? running_mean : nullptr, upd ? running_var : nullptr, reinterpret_cast<float2*>(table),
                       reinterpret_cast<unsigned*>(amax_y));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats, float* gx,
              float* gres, float* ggamma, float* gbeta, double* scratch, int groups, int rows, int C, float eps, int flags,
              int use_running, const float* running_mean, const float* running_var, void* stream) {
    return bh_bn_bwd_amax(gy, y, x, gamma, beta, stats, gx, gres, ggamma, gbeta, scratch, groups, rows, C, eps, flags, use_running, running_mean,
                          running_var, nullptr, stream);
}

int bh_bn_bwd_amax(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats, float* gx,
                   float* gres, float* ggamma, float* gbeta, double* scratch, int groups, int rows, int C, float eps, int flags,
                   int use_running, const float* running_mean, const float* running_var, float* amax_gx, void* stream) {
    BnGeom g;
    if (!gy || !x || !gx || !stats || !scratch || ((flags & 1) && !(flags & 4) && !y)) return BH_E_BADARG;
    if ((flags & 4) && (flags & 2)) return BH_E_BADARG;        // the mask can only be recomputed without a residual input
    if (C == 1) return amax_gx ? BH_E_UNSUPPORTED : bn1_bwd(gy, y, x, gamma, beta, stats, gx, gres, ggamma, gbeta, scratch, groups, rows, eps, flags, use_running,
                               running_mean, running_var, bh_stream(stream));
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    if (flags & 16) {            // scratch = padded sums accumulated by bh_conv_dgrad_bnreduce (training mode only)
        if (use_running) return BH_E_BADARG;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, gy, y, x, gamma, beta,
                           (const float4*)nullptr, gx, gres, g, flags, stats, (const double*)scratch, eps, ggamma, gbeta,
                           reinterpret_cast<unsigned*>(amax_gx));
        BH_LAUNCH_CHECK();
        return BH_OK;
    }
    // scratch: [groups][C] float4 coefficient table, then the per-chunk partial sums
    float4* coef = reinterpret_cast<float4*>(scratch);
    double* part = scratch + (size_t)groups * C * 2;
    const bool need_sums = !use_running || ggamma || gbeta;
    if (need_sums) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, gy, y, x, gamma, beta, stats,
                           running_mean, running_var, g, eps, flags, use_running, part);
        BH_LAUNCH_CHECK();
    }
    // (eval mode without trainable affine parameters: nchunks = 0 makes the finalize kernel write the table only)
    BnGeom gf = g;
    if (!need_sums) gf.nchunks = 0;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, gf, stats, running_mean, running_var, eps,
                       use_running, ggamma, gbeta, coef);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, gy, y, x, gamma, beta, coef,
                       gx, gres, g, flags, (const double*)nullptr, (const double*)nullptr, eps, (float*)nullptr,
                       (float*)nullptr, reinterpret_cast<unsigned*>(amax_gx));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_scratch_doubles(int groups, int C) { return groups * C * 2 * (1 + (BN_MAX_CHUNKS > 256 ? BN_MAX_CHUNKS : 256)); }

int bh_bn_stats(const float* x, double* stats, int groups, int rows, int C, int flags, void* stream) {
    if (!x || !stats) return BH_E_BADARG;
    return bn_launch_stats(x, groups, rows, C, stats, bh_stream(stream), (flags & BH_BN_DETERMINISTIC) ? 1 : 0);
}

int bh_bn_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                      unsigned char* idx, double* stats, int groups, int N, int Hi, int Wi, int C, float eps, float momentum, int flags,
                      int use_running, float* amax_y, void* stream) {
    BnGeom g;
    if (!x || !y || !stats || N < 1 || groups < 1 || N % groups) return BH_E_BADARG;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    const int rows = (N / groups) * Hi * Wi;
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0) || groups * C * 8 > 32768) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    if (!use_running && !(flags & 8)) {                       // bit 3: the producer already accumulated the sums
        hipLaunchKernelGGL(bn_stats_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, x, g, stats);
        BH_LAUNCH_CHECK();
    }
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    size_t nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    nb = (nb * 256 / (C / 4)) * (C / 4) / 256;               // (keeps gridDim * 256 a multiple of C / 4: any value is correct, this is tidy)
    if (nb < 1) nb = 1;
    const bool upd = !use_running && running_mean && running_var;
    hipLaunchKernelGGL(bn_maxpool_fwd_kernel, dim3((unsigned)nb), dim3(256), (size_t)groups * C * 8, s, x, gamma, beta, running_mean, running_var,
                       stats, y, idx, g, N, Hi, Wi, Ho, Wo, eps, flags & 1, use_running, momentum, upd ? running_mean : nullptr,
                       upd ? running_var : nullptr, reinterpret_cast<unsigned*>(amax_y));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

// ---------------------------------------------------------------------------------------------
// Adjoint of bn_maxpool_fwd in two passes over the BatchNorm INPUT (round 5).  Unfused it was bh_maxpool3s2_bwd (writes the 134 MB
// full-resolution gradient of the extractor stem) followed by bh_bn_bwd (reads it twice, next to x): 850 MB per call, three calls per step.
// Here the full-resolution gradient never exists: both passes rebuild d(pixel) = sum of the pooled gradients of the (<= 4) windows whose
// recorded arg-max points at the pixel (the gather form of maxpool_bwd_kernel; gy and idx are a quarter of the size and stay in L2),
// mask it with the recomputed ReLU and either accumulate (sum d, sum d xhat) - chunk partials in fixed order, bn_bwd_finalize_kernel
// makes the coefficient table - or write gx = scale (d - mean d - xhat mean(d xhat)).  490 MB per call.
// ---------------------------------------------------------------------------------------------
struct PoolGeom { int Hi, Wi, Ho, Wo, ipg; };      // ipg: images per group

// The (<= 4) windows that contain input pixel (iy, ix): rows iy >> 1 and (iy + 1) >> 1 (the same for even iy), columns alike.  Branch-free:
// all four candidates are loaded (clamped to the first where they do not exist) so that the loads of several pixels can be in flight
// together; they are added in the order of maxpool_bwd_kernel's loops (bitwise the same d).
struct PoolTaps {
    uchar4 w[4];
    float4 g[4];
    unsigned char t[4];
    bool ok[4];
};
__device__ __forceinline__ void pooled_taps_load(PoolTaps& q, const unsigned char* __restrict__ idx, const float* __restrict__ gy, const PoolGeom& pg,
                                                 int n, int iy, int ix, int C4, int c, bool live) {
    const int oyA = iy >> 1, oyB = (iy + 1) >> 1, oxA = ix >> 1, oxB = (ix + 1) >> 1;
    const bool vy = oyB != oyA && oyB < pg.Ho, vx = oxB != oxA && oxB < pg.Wo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool by = k >> 1, bx = k & 1;
        q.ok[k] = live && (!by || vy) && (!bx || vx);
        const int oy = (by && vy) ? oyB : oyA, ox = (bx && vx) ? oxB : oxA;
        q.t[k] = (unsigned char)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));
        const size_t o = live ? (((size_t)n * pg.Ho + oy) * pg.Wo + ox) * C4 + c : 0;
        q.w[k] = *reinterpret_cast<const uchar4*>(idx + o * 4);
        q.g[k] = *reinterpret_cast<const float4*>(gy + o * 4);
    }
}
__device__ __forceinline__ float4 pooled_taps_sum(const PoolTaps& q) {
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (q.ok[k] && q.w[k].x == q.t[k]) d.x += q.g[k].x;
        if (q.ok[k] && q.w[k].y == q.t[k]) d.y += q.g[k].y;
        if (q.ok[k] && q.w[k].z == q.t[k]) d.z += q.g[k].z;
        if (q.ok[k] && q.w[k].w == q.t[k]) d.w += q.g[k].w;
    }
    return d;
}

// grid (nchunks, groups): rows of a group = its images' pixels, as bn_bwd_reduce_kernel; four rows per lane in flight
__global__ void __launch_bounds__(256) bn_maxpool_bwd_reduce_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                                    const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const double* __restrict__ stats,
                                                                    const float* __restrict__ rmean, const float* __restrict__ rvar, BnGeom g,
                                                                    PoolGeom pg, float eps, int relu, int use_running, double* __restrict__ part) {
    __shared__ double sm[256 * 8];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    float mean[4], invstd[4], sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bn_coeffs(stats, gamma, beta, rmean, rvar, use_running, g.groups, grp, g.C, cq * 4 + i, eps, (double)g.rows, mean[i], invstd[i], sc[i], sh[i], g.det);
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const int hw = pg.Hi * pg.Wi;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rbeg + r0; r < rend; r += 4 * g.RPP) {
        float4 a[4];
        PoolTaps q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * g.RPP;
            const bool live = rr < rend;
            const int rc = live ? rr : rbeg;
            const int nl = rc / hw, p = rc - nl * hw, iy = p / pg.Wi, ix = p - iy * pg.Wi;
            a[u] = *reinterpret_cast<const float4*>(x + gbase + (size_t)rc * g.C);
            pooled_taps_load(q[u], idx, gy, pg, grp * pg.ipg + nl, iy, ix, g.C4, cq, live);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + u * g.RPP >= rend) break;
            float4 d = pooled_taps_sum(q[u]);
            const float4 av = a[u];
            if (relu) {
                if (!(av.x * sc[0] + sh[0] > 0.f)) d.x = 0.f;
                if (!(av.y * sc[1] + sh[1] > 0.f)) d.y = 0.f;
                if (!(av.z * sc[2] + sh[2] > 0.f)) d.z = 0.f;
                if (!(av.w * sc[3] + sh[3] > 0.f)) d.w = 0.f;
            }
            v[0] += d.x; v[1] += d.y; v[2] += d.z; v[3] += d.w;
            v[4] += (double)(d.x * ((av.x - mean[0]) * invstd[0]));
            v[5] += (double)(d.y * ((av.y - mean[1]) * invstd[1]));
            v[6] += (double)(d.z * ((av.z - mean[2]) * invstd[2]));
            v[7] += (double)(d.w * ((av.w - mean[3]) * invstd[3]));
        }
    }
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 2;
        for (int i = 0; i < 4; ++i) { p[i * 2] = v[i]; p[i * 2 + 1] = v[4 + i]; }
    }
}

// grid (nblk, groups); coef[grp][c] = (mean, invstd, mean d, mean d xhat) from bn_bwd_finalize_kernel
__global__ void __launch_bounds__(256) bn_maxpool_bwd_apply_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                                   const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float4* __restrict__ coef,
                                                                   float* __restrict__ gx, BnGeom g, PoolGeom pg, int relu,
                                                                   unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    float vmax = 0.f;
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    float mean[4], invstd[4], sc[4], sh[4], k1[4], k2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cq * 4 + i;
        const float4 cf = coef[(size_t)grp * g.C + c];
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        mean[i] = cf.x; invstd[i] = cf.y; k1[i] = cf.z; k2[i] = cf.w;
        sc[i] = gm * cf.y;
        sh[i] = bt - cf.x * sc[i];
    }
    const size_t gbase = ((size_t)grp * g.rows) * g.C + cq * 4;
    const int hw = pg.Hi * pg.Wi;
    const int stride = gridDim.x * g.RPP;
    for (int r = blockIdx.x * g.RPP + r0; r < g.rows; r += 4 * stride) {
        float4 a[4];
        PoolTaps q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            const bool live = rr < g.rows;
            const int rc = live ? rr : 0;
            const int nl = rc / hw, p = rc - nl * hw, iy = p / pg.Wi, ix = p - iy * pg.Wi;
            a[u] = *reinterpret_cast<const float4*>(x + gbase + (size_t)rc * g.C);
            pooled_taps_load(q[u], idx, gy, pg, grp * pg.ipg + nl, iy, ix, g.C4, cq, live);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            float4 d = pooled_taps_sum(q[u]);
            const float4 av = a[u];
            if (relu) {
                if (!(av.x * sc[0] + sh[0] > 0.f)) d.x = 0.f;
                if (!(av.y * sc[1] + sh[1] > 0.f)) d.y = 0.f;
                if (!(av.z * sc[2] + sh[2] > 0.f)) d.z = 0.f;
                if (!(av.w * sc[3] + sh[3] > 0.f)) d.w = 0.f;
            }
            float4 o;
            o.x = sc[0] * (d.x - k1[0] - (av.x - mean[0]) * invstd[0] * k2[0]);
            o.y = sc[1] * (d.y - k1[1] - (av.y - mean[1]) * invstd[1] * k2[1]);
            o.z = sc[2] * (d.z - k1[2] - (av.z - mean[2]) * invstd[2] * k2[2]);
            o.w = sc[3] * (d.w - k1[3] - (av.w - mean[3]) * invstd[3] * k2[3]);
            *reinterpret_cast<float4*>(gx + gbase + (size_t)rr * g.C) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
}

int bh_bn_maxpool_bwd(const float* gy, const unsigned char* idx, const float* x, const float* gamma, const float* beta, const double* stats,
                      float* gx, float* ggamma, float* gbeta, double* scratch, int groups, int N, int Hi, int Wi, int C, float eps, int flags,
                      int use_running, const float* running_mean, const float* running_var, float* amax_gx, void* stream) {
    BnGeom g;
    if (!gy || !idx || !x || !gx || !stats || !scratch || N < 1 || groups < 1 || N % groups) return BH_E_BADARG;
    if (use_running && (!running_mean || !running_var)) return BH_E_BADARG;
    const long long rows = (long long)(N / groups) * Hi * Wi;
    if (rows >= (1ll << 31) || !bn_geom(groups, (int)rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    PoolGeom pg = {Hi, Wi, (Hi - 1) / 2 + 1, (Wi - 1) / 2 + 1, N / groups};
    hipStream_t s = bh_stream(stream);
    // scratch as in bh_bn_bwd: [groups][C] float4 coefficient table, then the per-chunk partial sums
    float4* coef = reinterpret_cast<float4*>(scratch);
    double* part = scratch + (size_t)groups * C * 2;
    const bool need_sums = !use_running || ggamma || gbeta;
    if (need_sums) {
        hipLaunchKernelGGL(bn_maxpool_bwd_reduce_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, gy, idx, x, gamma, beta, stats, running_mean,
                           running_var, g, pg, eps, flags & 1, use_running, part);
        BH_LAUNCH_CHECK();
    }
    BnGeom gf = g;
    if (!need_sums) gf.nchunks = 0;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, gf, stats, running_mean, running_var, eps, use_running, ggamma,
                       gbeta, coef);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_maxpool_bwd_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, gy, idx, x, gamma, beta, coef, gx, g, pg,
                       flags & 1, reinterpret_cast<unsigned*>(amax_gx));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C" (templates below)

// ---------------------------------------------------------------------------------------------
// BatchNorm adjoint whose output gradient is the dgrad of a 1x1 convolution with FEW output channels (round 5: the decoder units'
// BatchNorm + ReLU in front of the 1x1 conv that halves the channels, src/backbones/utils.py:60-82, at full resolution: C = 32 -> 16 on
// 128 x 128 maps).  Unfused: the 1x1 dgrad writes g[M][C] (268 MB) and bh_bn_bwd reads it twice.  Here g is never stored: both passes
// rebuild  g[p][c] = sum_k gs[p][k] w[k][c]  from the HALF-SIZE gradient of the conv's output (KC = 16 floats per pixel; the lane's four
// weight columns stay in registers: KC float4) - 16 FMAs per element, far below what the two streams leave room for.  Passes as in
// bh_bn_bwd: chunk partials in fixed order + bn_bwd_finalize_kernel, then apply.  The mask is recomputed from x (no residual: flags bit2
// semantics).  1742 -> 1072 MB per call.
// ---------------------------------------------------------------------------------------------
template <int KC>
__device__ __forceinline__ float4 grad_from_1x1(const float4 (&gs)[KC / 4], const float4 (&w)[KC]) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < KC / 4; ++q) {
        const float e[4] = {gs[q].x, gs[q].y, gs[q].z, gs[q].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 wk = w[q * 4 + i];
            g.x = __builtin_fmaf(e[i], wk.x, g.x); g.y = __builtin_fmaf(e[i], wk.y, g.y);
            g.z = __builtin_fmaf(e[i], wk.z, g.z); g.w = __builtin_fmaf(e[i], wk.w, g.w);
        }
    }
    return g;
}

// grid (nchunks, groups)
template <int KC>
__global__ void __launch_bounds__(256) bn_bwd_1x1_reduce_kernel(const float* __restrict__ gs, const float* __restrict__ w,
                                                                const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const double* __restrict__ stats, BnGeom g,
                                                                float eps, int relu, double* __restrict__ part) {
    __shared__ double sm[256 * 8];
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const int rbeg = chunk * g.rows_per_chunk, rend = min(g.rows, rbeg + g.rows_per_chunk);
    float4 wr[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (size_t)k * g.C + cq * 4);
    float mean[4], invstd[4], sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bn_coeffs(stats, gamma, beta, nullptr, nullptr, 0, g.groups, grp, g.C, cq * 4 + i, eps, (double)g.rows, mean[i], invstd[i], sc[i], sh[i], g.det);
    const size_t rbase = (size_t)grp * g.rows;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rbeg + r0; r < rend; r += 2 * g.RPP) {
        float4 a[2], q[2][KC / 4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = r + u * g.RPP;
            const size_t row = rbase + (size_t)(rr < rend ? rr : rbeg);
            a[u] = *reinterpret_cast<const float4*>(x + row * g.C + cq * 4);
#pragma unroll
            for (int j = 0; j < KC / 4; ++j) q[u][j] = *reinterpret_cast<const float4*>(gs + row * KC + j * 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (r + u * g.RPP >= rend) break;
            float4 d = grad_from_1x1<KC>(q[u], wr);
            const float4 av = a[u];
            if (relu) {
                if (!(av.x * sc[0] + sh[0] > 0.f)) d.x = 0.f;
                if (!(av.y * sc[1] + sh[1] > 0.f)) d.y = 0.f;
                if (!(av.z * sc[2] + sh[2] > 0.f)) d.z = 0.f;
                if (!(av.w * sc[3] + sh[3] > 0.f)) d.w = 0.f;
            }
            v[0] += d.x; v[1] += d.y; v[2] += d.z; v[3] += d.w;
            v[4] += (double)(d.x * ((av.x - mean[0]) * invstd[0]));
            v[5] += (double)(d.y * ((av.y - mean[1]) * invstd[1]));
            v[6] += (double)(d.z * ((av.z - mean[2]) * invstd[2]));
            v[7] += (double)(d.w * ((av.w - mean[3]) * invstd[3]));
        }
    }
    reduce_rows<8>(v, g.LPR, g.RPP, sm);
    if ((int)threadIdx.x < g.LPR) {
        double* p = part + (((size_t)grp * g.nchunks + chunk) * g.C + cq * 4) * 2;
        for (int i = 0; i < 4; ++i) { p[i * 2] = v[i]; p[i * 2 + 1] = v[4 + i]; }
    }
}

// grid (nblk, groups); coef from bn_bwd_finalize_kernel
template <int KC>
__global__ void __launch_bounds__(256) bn_bwd_1x1_apply_kernel(const float* __restrict__ gs, const float* __restrict__ w,
                                                               const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float4* __restrict__ coef,
                                                               float* __restrict__ gx, BnGeom g, int relu, unsigned* __restrict__ amax) {
    __shared__ float sm_amax[4];
    float vmax = 0.f;
    const int cq = threadIdx.x % g.LPR, r0 = threadIdx.x / g.LPR;
    const int grp = blockIdx.y;
    float4 wr[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (size_t)k * g.C + cq * 4);
    float mean[4], invstd[4], sc[4], sh[4], k1[4], k2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cq * 4 + i;
        const float4 cf = coef[(size_t)grp * g.C + c];
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        mean[i] = cf.x; invstd[i] = cf.y; k1[i] = cf.z; k2[i] = cf.w;
        sc[i] = gm * cf.y;
        sh[i] = bt - cf.x * sc[i];
    }
    const size_t rbase = (size_t)grp * g.rows;
    const int stride = gridDim.x * g.RPP;
    for (int r = blockIdx.x * g.RPP + r0; r < g.rows; r += 2 * stride) {
        float4 a[2], q[2][KC / 4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = r + u * stride;
            const size_t row = rbase + (size_t)(rr < g.rows ? rr : 0);
            a[u] = *reinterpret_cast<const float4*>(x + row * g.C + cq * 4);
#pragma unroll
            for (int j = 0; j < KC / 4; ++j) q[u][j] = *reinterpret_cast<const float4*>(gs + row * KC + j * 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = r + u * stride;
            if (rr >= g.rows) break;
            float4 d = grad_from_1x1<KC>(q[u], wr);
            const float4 av = a[u];
            if (relu) {
                if (!(av.x * sc[0] + sh[0] > 0.f)) d.x = 0.f;
                if (!(av.y * sc[1] + sh[1] > 0.f)) d.y = 0.f;
                if (!(av.z * sc[2] + sh[2] > 0.f)) d.z = 0.f;
                if (!(av.w * sc[3] + sh[3] > 0.f)) d.w = 0.f;
            }
            float4 o;
            o.x = sc[0] * (d.x - k1[0] - (av.x - mean[0]) * invstd[0] * k2[0]);
            o.y = sc[1] * (d.y - k1[1] - (av.y - mean[1]) * invstd[1] * k2[1]);
            o.z = sc[2] * (d.z - k1[2] - (av.z - mean[2]) * invstd[2] * k2[2]);
            o.w = sc[3] * (d.w - k1[3] - (av.w - mean[3]) * invstd[3] * k2[3]);
            *reinterpret_cast<float4*>(gx + (rbase + (size_t)rr) * g.C + cq * 4) = o;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (amax) bh_amax_commit(amax, vmax, blockIdx.x + blockIdx.y * 7u, sm_amax);
}

extern "C" {

int bh_bn_bwd_from_1x1(const float* gs, const float* w, int KC, const float* x, const float* gamma, const float* beta, const double* stats,
                       float* gx, float* ggamma, float* gbeta, double* scratch, int groups, int rows, int C, float eps, int flags,
                       float* amax_gx, void* stream) {
    BnGeom g;
    if (!gs || !w || !x || !gx || !stats || !scratch) return BH_E_BADARG;
    if (KC != 16 && KC != 32) return BH_E_UNSUPPORTED;
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    float4* coef = reinterpret_cast<float4*>(scratch);
    double* part = scratch + (size_t)groups * C * 2;
    unsigned* am = reinterpret_cast<unsigned*>(amax_gx);
    if (KC == 16) hipLaunchKernelGGL(bn_bwd_1x1_reduce_kernel<16>, dim3(g.nchunks, groups), dim3(256), 0, s, gs, w, x, gamma, beta, stats, g, eps, flags & 1, part);
    else hipLaunchKernelGGL(bn_bwd_1x1_reduce_kernel<32>, dim3(g.nchunks, groups), dim3(256), 0, s, gs, w, x, gamma, beta, stats, g, eps, flags & 1, part);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, g, stats, (const float*)nullptr, (const float*)nullptr, eps, 0, ggamma,
                       gbeta, coef);
    BH_LAUNCH_CHECK();
    if (KC == 16) hipLaunchKernelGGL(bn_bwd_1x1_apply_kernel<16>, dim3(apply_blocks(g), groups), dim3(256), 0, s, gs, w, x, gamma, beta, coef, gx, g, flags & 1, am);
    else hipLaunchKernelGGL(bn_bwd_1x1_apply_kernel<32>, dim3(apply_blocks(g), groups), dim3(256), 0, s, gs, w, x, gamma, beta, coef, gx, g, flags & 1, am);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_join_scratch_doubles(int groups, int C) { return groups * C * 2 + groups * C * 3 * (BN_MAX_CHUNKS > 256 ? BN_MAX_CHUNKS : 256); }

int bh_bn_join_fwd(const float* xa, const float* xb, const float* gamma_a, const float* beta_a, float* rmean_a, float* rvar_a,
                   const float* gamma_b, const float* beta_b, float* rmean_b, float* rvar_b, const double* stats_a, const double* stats_b,
                   float* y, int groups, int rows, int C, float eps_a, float eps_b, float momentum_a, float momentum_b, int flags,
                   float* amax_y, void* stream) {
    BnGeom g;
    if (!xa || !xb || !y || !stats_a || !stats_b) return BH_E_BADARG;
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipLaunchKernelGGL(bn_join_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, bh_stream(stream), xa, xb, gamma_a, beta_a, gamma_b,
                       beta_b, stats_a, stats_b, y, g, eps_a, eps_b, flags & 1, momentum_a, momentum_b, (rmean_a && rvar_a) ? rmean_a : nullptr,
                       rvar_a, (rmean_b && rvar_b) ? rmean_b : nullptr, rvar_b, reinterpret_cast<unsigned*>(amax_y));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

static int bn_join_bwd_impl(const float* gy, const float* y, const float* xa, const float* xb, const float* gamma_a, const float* gamma_b,
                            const float* beta_a, const float* beta_b, int remask,
                            const double* stats_a, const double* stats_b, float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b,
                            float* gbeta_b, double* scratch, int groups, int rows, int C, float eps_a, float eps_b, int flags, float* amax_gxa,
                            float* amax_gxb, void* stream) {
    BnGeom g;
    if (!gy || !xa || !xb || !gxa || !gxb || !stats_a || !stats_b || !scratch || ((flags & 1) && !remask && !y)) return BH_E_BADARG;
    if (!bn_geom(groups, rows, C, g, (flags & BH_BN_DETERMINISTIC) ? 1 : 0)) return BH_E_UNSUPPORTED;
    hipStream_t s = bh_stream(stream);
    float4* coef = reinterpret_cast<float4*>(scratch);
    double* part = scratch + (size_t)groups * C * 2;
    hipLaunchKernelGGL(bn_join_bwd_reduce_kernel, dim3(g.nchunks, groups), dim3(256), 0, s, gy, y, xa, xb, stats_a, stats_b, g, eps_a, eps_b,
                       flags & 1, part, gamma_a, beta_a, gamma_b, beta_b, remask);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_join_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, g, ggamma_a, gbeta_a, ggamma_b, gbeta_b, coef);
    BH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_join_bwd_apply_kernel, dim3(apply_blocks(g), groups), dim3(256), 0, s, gy, y, xa, xb, gamma_a, gamma_b, stats_a,
                       stats_b, coef, gxa, gxb, g, eps_a, eps_b, flags & 1, reinterpret_cast<unsigned*>(amax_gxa),
                       reinterpret_cast<unsigned*>(amax_gxb), beta_a, beta_b, remask);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_bn_join_bwd(const float* gy, const float* y, const float* xa, const float* xb, const float* gamma_a, const float* gamma_b,
                   const double* stats_a, const double* stats_b, float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b,
                   float* gbeta_b, double* scratch, int groups, int rows, int C, float eps_a, float eps_b, int flags, float* amax_gxa,
                   float* amax_gxb, void* stream) {
    return bn_join_bwd_impl(gy, y, xa, xb, gamma_a, gamma_b, nullptr, nullptr, 0, stats_a, stats_b, gxa, gxb, ggamma_a, gbeta_a, ggamma_b, gbeta_b,
                            scratch, groups, rows, C, eps_a, eps_b, flags, amax_gxa, amax_gxb, stream);
}

int bh_bn_join_bwd_remask(const float* gy, const float* xa, const float* xb, const float* gamma_a, const float* beta_a, const float* gamma_b,
                          const float* beta_b, const double* stats_a, const double* stats_b, float* gxa, float* gxb, float* ggamma_a,
                          float* gbeta_a, float* ggamma_b, float* gbeta_b, double* scratch, int groups, int rows, int C, float eps_a, float eps_b,
                          int flags, float* amax_gxa, float* amax_gxb, void* stream) {
    return bn_join_bwd_impl(gy, nullptr, xa, xb, gamma_a, gamma_b, beta_a, beta_b, 1, stats_a, stats_b, gxa, gxb, ggamma_a, gbeta_a, ggamma_b,
                            gbeta_b, scratch, groups, rows, C, eps_a, eps_b, flags, amax_gxa, amax_gxb, stream);
}

}  // extern "C"
