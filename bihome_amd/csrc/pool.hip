// NHWC pooling and elementwise glue kernels (all HBM-bound, float4 per lane along channels).
#include "common.h"

// MaxPool2d(3, stride 2, pad 1): grid-stride over output float4s
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Hi,
                                                          int Wi, int C4, int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= Hi) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= Wi) continue;
                float4 v = *reinterpret_cast<const float4*>(x + ((((size_t)n * Hi + iy) * Wi + ix) * C4 + c) * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(y + i * 4) = m;
    }
}

// adjoint, gather form (no atomics): an input pixel receives gy of every window whose FIRST maximum
// (row-major scan, the ATen tie rule) it is.
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gx, int N, int Hi, int Wi, int C4, int Ho,
                                                          int Wo) {
    const size_t total = (size_t)N * Hi * Wi * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ix = (int)(p % Wi); p /= Wi;
        const int iy = (int)(p % Hi);
        const int n = (int)(p / Hi);
        const float4 me = *reinterpret_cast<const float4*>(x + i * 4);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows containing (iy, ix): oy in [ceil((iy-1)/2), floor((iy+1)/2)]
        const int oy0 = max(0, (iy) / 2), oy1 = min(Ho - 1, (iy + 1) / 2);
        const int ox0 = max(0, (ix) / 2), ox1 = min(Wo - 1, (ix + 1) / 2);
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                // is (iy, ix) the first maximum of window (oy, ox)?
                bool fx = true, fy = true, fz = true, fw = true;
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if (yy < 0 || yy >= Hi) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if (xx < 0 || xx >= Wi) continue;
                        if (yy == iy && xx == ix) continue;
                        const float4 v = *reinterpret_cast<const float4*>(x + ((((size_t)n * Hi + yy) * Wi + xx) * C4 + c) * 4);
                        const bool before = (yy < iy) || (yy == iy && xx < ix);
                        // an earlier element wins ties, a later one only if strictly greater
                        if (before ? (v.x >= me.x) : (v.x > me.x)) fx = false;
                        if (before ? (v.y >= me.y) : (v.y > me.y)) fy = false;
                        if (before ? (v.z >= me.z) : (v.z > me.z)) fz = false;
                        if (before ? (v.w >= me.w) : (v.w > me.w)) fw = false;
                    }
                }
                const float4 go = *reinterpret_cast<const float4*>(gy + ((((size_t)n * Ho + oy) * Wo + ox) * C4 + c) * 4);
                if (fx) g.x += go.x;
                if (fy) g.y += go.y;
                if (fz) g.z += go.z;
                if (fw) g.w += go.w;
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

static int nblocks(size_t work) {
    size_t b = (work + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" {

int bh_maxpool3s2_fwd(const float* x, float* y, int N, int Hi, int Wi, int C, void* stream) {
    if (!x || !y || C % 4) return C % 4 ? BH_E_UNSUPPORTED : BH_E_BADARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblocks((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, bh_stream(stream), x,
                       y, N, Hi, Wi, C / 4, Ho, Wo);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

int bh_maxpool3s2_bwd(const float* x, const float* gy, float* gx, int N, int Hi, int Wi, int C, void* stream) {
    if (!x || !gy || !gx || C % 4) return C % 4 ? BH_E_UNSUPPORTED : BH_E_BADARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks((size_t)N * Hi * Wi * (C / 4))), dim3(256), 0, bh_stream(stream), x,
                       gy, gx, N, Hi, Wi, C / 4, Ho, Wo);
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

int bh_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return BH_E_BADARG;
    if (n == 0) return BH_OK;
    hipLaunchKernelGGL(add_kernel, dim3(nblocks((size_t)n / 4 + 1)), dim3(256), 0, bh_stream(stream), a, b, out,
                       (size_t)n / 4, (size_t)n);
    BH_LAUNCH_CHECK();
    return BH_OK;
}

}  // extern "C"
