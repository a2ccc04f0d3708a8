// Trained content masks of the Zhang baseline (round 4: FIX_MASK False).  The mask predictor's conv / BatchNorm stack runs on the conv
// executor; this file holds what follows its last BatchNorm (src/backbones/ContentAware.py:24-26,28-35,47-50,128-134):
//   s = sigmoid(y)                                                       (nn.Sigmoid of layer5)
//   m = clamp(s / (max_p s * strength), 0, 1)   if strength > 0          (__normalize_mask: per-sample maximum)      else m = s
//   g = m * f                                                            (G = mask * features, the resnet's input)
// and the adjoint (gradient of m from the head - the masks weight the triplet loss, directly and through the warp - plus the
// gradient of g from the resnet) back to y and f.  The maximum's gradient goes to ONE pixel per sample (the first one that attains
// it: torch's max(1)[0] backward is an index_select).  One workgroup per sample: per-sample reductions have a single writer, so the
// kernels are the same in deterministic mode.  [N, P] tensors, P = h * w pixels of one-channel maps.
#include "common.h"

__device__ __forceinline__ float mask_sigmoid(float y) { return 1.0f / (1.0f + expf(-y)); }

// smax[N] (the per-sample maximum of s; written also without normalisation), imax[N] (its first pixel)
__global__ void __launch_bounds__(256) mask_fwd_kernel(const float* __restrict__ y, const float* __restrict__ f, int P, float strength,
                                                       float* __restrict__ m, float* __restrict__ g, float* __restrict__ smax,
                                                       int* __restrict__ imax) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int n = blockIdx.x;
    const size_t o = (size_t)n * P;
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int p = threadIdx.x; p < P; p += 256) {
        const float s = mask_sigmoid(y[o + p]);
        if (s > best) { best = s; bi = p; }              // (ascending p per thread: the first pixel of the thread's maximum)
    }
    for (int off = 32; off; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
    for (int k = 1; k < 4; ++k)
        if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
    if (threadIdx.x == 0) { smax[n] = best; imax[n] = bi; }
    const float den = best * strength;                   // mask / (max_value * strength), :33
    for (int p = threadIdx.x; p < P; p += 256) {
        const float s = mask_sigmoid(y[o + p]);
        const float v = strength > 0.0f ? fminf(fmaxf(s / den, 0.0f), 1.0f) : s;
        m[o + p] = v;
        if (g) g[o + p] = v * f[o + p];
    }
}

// g_m (may be NULL: no gradient from the head, e.g. the biHomE head on this backbone) and g_g (may be NULL) -> g_y, g_f (may be NULL)
__global__ void __launch_bounds__(256) mask_bwd_kernel(const float* __restrict__ y, const float* __restrict__ f, const float* __restrict__ m,
                                                       const float* __restrict__ smax, const int* __restrict__ imax,
                                                       const float* __restrict__ g_m, const float* __restrict__ g_g, int P, float strength,
                                                       float* __restrict__ g_y, float* __restrict__ g_f) {
    __shared__ double sm[4];
    const int n = blockIdx.x;
    const size_t o = (size_t)n * P;
    const float M = smax[n], den = M * strength;
    double gM = 0.0;                                     // d loss / d max = - sum_p gv_p * v_p / M over the pixels the clamp passes
    if (strength > 0.0f) {
        for (int p = threadIdx.x; p < P; p += 256) {
            const float s = mask_sigmoid(y[o + p]);
            const float v = s / den;
            const float gm = (g_m ? g_m[o + p] : 0.0f) + (g_g ? g_g[o + p] * f[o + p] : 0.0f);
            if (v >= 0.0f && v <= 1.0f) gM -= (double)gm * (double)(v / M);
        }
        gM = wave_sum(gM);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = gM;
        __syncthreads();
        gM = sm[0] + sm[1] + sm[2] + sm[3];
    }
    const int pi = imax[n];
    for (int p = threadIdx.x; p < P; p += 256) {
        const float s = mask_sigmoid(y[o + p]);
        const float gg = g_g ? g_g[o + p] : 0.0f;
        const float gm = (g_m ? g_m[o + p] : 0.0f) + gg * (f ? f[o + p] : 0.0f);
        float gs = gm;
        if (strength > 0.0f) {
            const float v = s / den;
            gs = (v >= 0.0f && v <= 1.0f) ? gm / den : 0.0f;
            if (p == pi) gs += (float)gM;
        }
        g_y[o + p] = gs * s * (1.0f - s);
        if (g_f) g_f[o + p] = gg * m[o + p];
    }
}

extern "C" {

int bh_mask_fwd(const float* y, const float* f, int N, int P, float strength, float* m, float* g, float* smax, int* imax, void* stream) {
    if (!y || !m || !smax || !imax || (g && !f) || N < 0 || P < 1) return BH_E_BADARG;
    if (N == 0) return BH_OK;
    hipLaunchKernelGGL(mask_fwd_kernel, dim3(N), dim3(256), 0, bh_stream(stream), y, f, P, strength, m, g, smax, imax);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_mask_bwd(const float* y, const float* f, const float* m, const float* smax, const int* imax, const float* g_m, const float* g_g,
                int N, int P, float strength, float* g_y, float* g_f, void* stream) {
    if (!y || !m || !smax || !imax || !g_y || ((g_g || g_f) && !f) || (g_f && !g_g) || N < 0 || P < 1) return BH_E_BADARG;
    if (N == 0) return BH_OK;
    hipLaunchKernelGGL(mask_bwd_kernel, dim3(N), dim3(256), 0, bh_stream(stream), y, f, m, smax, imax, g_m, g_g, P, strength, g_y, g_f);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
