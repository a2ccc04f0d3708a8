// Do VALU instructions issue under running MFMAs on gfx950?  A loop of 32 v_mfma_f32_32x32x2_f32 per iteration (two
// accumulators) with NV independent v_fma_f32 interleaved, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec_bench.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = s;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = x + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(m * NV + j) & 7] = __builtin_fmaf(v[(m * NV + j) & 7], s, 1.0f);
        }
    }
    float t = 0;
    for (int r = 0; r < 16; ++r) t += a0[r] + a1[r];
    for (int j = 0; j < 8; ++j) t += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NV>
void run(int blocks_per_cu, float* d_out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k<NV>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d_out, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NV>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d_out, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mf = (double)256 * blocks_per_cu * 4 * iters * 32;     // wave-level MFMAs
    const double tf = mf * 4096 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz
    printf("waves/SIMD %d  VALU per MFMA %.1f : %8.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD)\n", blocks_per_cu, NV / 2.0, ms, tf,
           ms * 1e-3 * 2.4e9 / (iters * 32.0 * blocks_per_cu));
}

int main() {
    float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>(w, d_out); run<2>(w, d_out); run<4>(w, d_out); run<8>(w, d_out); run<16>(w, d_out); run<24>(w, d_out);
    }
    return 0;
}
