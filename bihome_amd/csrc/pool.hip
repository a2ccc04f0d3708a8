// NHWC pooling and elementwise glue kernels (all HBM-bound, float4 per lane along channels).
#include "common.h"

// MaxPool2d(3, stride 2, pad 1): grid-stride over output float4s; also records the window position (0..8) of
// the FIRST maximum (row-major scan, the ATen tie rule) per element, one byte each
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int N, int Hi, int Wi, int C4,
                                                          int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 w = make_uchar4(0, 0, 0, 0);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= Hi) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= Wi) continue;
                const unsigned char t = (unsigned char)(ky * 3 + kx);
                float4 v = *reinterpret_cast<const float4*>(x + ((((size_t)n * Hi + iy) * Wi + ix) * C4 + c) * 4);
                if (v.x > m.x) { m.x = v.x; w.x = t; }
                if (v.y > m.y) { m.y = v.y; w.y = t; }
                if (v.z > m.z) { m.z = v.z; w.z = t; }
                if (v.w > m.w) { m.w = v.w; w.w = t; }
            }
        }
        *reinterpret_cast<float4*>(y + i * 4) = m;
        if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = w;
    }
}

// adjoint, gather form (no atomics): an input pixel receives gy of every window (<= 4) whose recorded
// arg-max position points at it
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const unsigned char* __restrict__ idx, const float* __restrict__ gy,
                                                          float* __restrict__ gx, int N, int Hi, int Wi, int C4, int Ho,
                                                          int Wo) {
    const size_t total = (size_t)N * Hi * Wi * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ix = (int)(p % Wi); p /= Wi;
        const int iy = (int)(p % Hi);
        const int n = (int)(p / Hi);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        const int oy0 = max(0, iy / 2), oy1 = min(Ho - 1, (iy + 1) / 2);
        const int ox0 = max(0, ix / 2), ox1 = min(Wo - 1, (ix + 1) / 2);
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                const unsigned char t = (unsigned char)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));
                const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C4 + c;
                const uchar4 w = *reinterpret_cast<const uchar4*>(idx + o * 4);
                const float4 go = *reinterpret_cast<const float4*>(gy + o * 4);
                if (w.x == t) g.x += go.x;
                if (w.y == t) g.y += go.y;
                if (w.z == t) g.z += go.z;
                if (w.w == t) g.w += go.w;
            }
        *reinterpret_cast<float4*>(gx + i * 4) = g;
    }
}

// global average pool: grid (N), block 256: y[n][c] = mean_p x[n][p][c]
__global__ void __launch_bounds__(256) gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        double acc = 0;
        for (int p = 0; p < HW; ++p) acc += x[((size_t)n * HW + p) * C + c];
        y[(size_t)n * C + c] = (float)(acc / HW);
    }
}

__global__ void __launch_bounds__(256) gap_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int N, int HW,
                                                      int C) {
    const size_t total = (size_t)N * HW * C;
    const float inv = 1.0f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t n = i / ((size_t)HW * C);
        gx[i] = gy[n * C + c] * inv;
    }
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ o, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = (n4 << 2) + threadIdx.x;
        o[i] = a[i] + b[i];
    }
}

// max |x| into a magnitude record (common.h F16X2): 512 looping workgroups, eight float4 per lane in flight, one integer atomic max each
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, size_t n4, size_t n, unsigned* __restrict__ rec) {
    __shared__ float sm[4];
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (; i < n4; i += stride) { const float4 v = x4[i]; m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
    bh_amax_commit(rec, m, blockIdx.x, sm);
}

static int nblocks(size_t work) {
    size_t b = (work + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" {

int bh_maxpool3s2_fwd(const float* x, float* y, unsigned char* argmax, int N, int Hi, int Wi, int C, void* stream) {
    if (!x || !y || C % 4) return C % 4 ? BH_E_UNSUPPORTED : BH_E_BADARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblocks((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, bh_stream(stream), x,
                       y, argmax, N, Hi, Wi, C / 4, Ho, Wo);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_maxpool3s2_bwd(const unsigned char* argmax, const float* gy, float* gx, int N, int Hi, int Wi, int C, void* stream) {
    if (!argmax || !gy || !gx || C % 4) return C % 4 ? BH_E_UNSUPPORTED : BH_E_BADARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks((size_t)N * Hi * Wi * (C / 4))), dim3(256), 0, bh_stream(stream),
                       argmax, gy, gx, N, Hi, Wi, C / 4, Ho, Wo);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_gap_fwd(const float* x, float* y, int N, int HW, int C, void* stream) {
    if (!x || !y) return BH_E_BADARG;
    if (N == 0) return BH_OK;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(N), dim3(256), 0, bh_stream(stream), x, y, HW, C);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_gap_bwd(const float* gy, float* gx, int N, int HW, int C, void* stream) {
    if (!gy || !gx) return BH_E_BADARG;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(nblocks((size_t)N * HW * C)), dim3(256), 0, bh_stream(stream), gy, gx, N, HW, C);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_absmax(const float* x, long long n, float* record, void* stream) {
    if (!x || !record || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return BH_E_BADARG;
    if (n == 0) return BH_OK;
    const size_t n4 = (size_t)n / 4;
    int nb = (int)((n4 + 256 * 8 - 1) / (256 * 8));
    nb = nb > 512 ? 512 : (nb < 1 ? 1 : nb);
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, bh_stream(stream), x, n4, (size_t)n, reinterpret_cast<unsigned*>(record));
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return BH_E_BADARG;
    if (n == 0) return BH_OK;
    hipLaunchKernelGGL(add_kernel, dim3(nblocks((size_t)n / 4 + 1)), dim3(256), 0, bh_stream(stream), a, b, out,
                       (size_t)n / 4, (size_t)n);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
