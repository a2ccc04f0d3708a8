// What issues under a running MFMA on gfx950?  32 v_mfma_f32_32x32x2_f32 per iteration (two accumulators) with, per
// MFMA pair: MODE 0 nothing, 1: 4 v_fma_f32, 3: 1 ds_read_b128 (result used at the end of the iteration),
// 4: 2 ds_read_b32, 5: 1 workgroup barrier per 32 MFMAs, 6: 1 buffer_load_dwordx4 (L2-resident) per 4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* src, int iters, float s) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = s;
    float v[4] = {x, x + 1, x + 2, x + 3};
    float4 lacc = make_float4(0, 0, 0, 0);
    unsigned sacc = 0;
    const float* lp = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(v[j], s, 1.0f);
            } else if (MODE == 2) {
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7" : "+s"(sacc));
            } else if (MODE == 3) {
                // plain C++ LDS read: the compiler places the lgkmcnt wait before the use at the end of the iteration
                const float4 t = *reinterpret_cast<const float4*>(lp + (((m + it) & 3) << 6));
                lacc.x += t.x;
            } else if (MODE == 4) {
                const float t0 = lp[((m + it) & 3) << 6], t1 = lp[(((m + it) & 3) << 6) + 64];
                lacc.x += t0 + t1;
            } else if (MODE == 6) {
                if ((m & 1) == 0) {
                    float4 t = *reinterpret_cast<const float4*>(src + ((threadIdx.x + m * 256 + (it & 7) * 4096) & 32767) * 4);
                    lacc.x += t.x;
                }
            }
        }
        if (MODE == 5) __builtin_amdgcn_s_barrier();
    }
    float t = lacc.x + (float)sacc;
    for (int r = 0; r < 16; ++r) t += a0[r] + a1[r];
    for (int j = 0; j < 4; ++j) t += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, int bpc, float* d_out, float* d_src) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, d_out, d_src, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, d_out, d_src, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves/SIMD %d  %-44s %8.3f ms  %.1f cycles per MFMA per SIMD\n", bpc, name, ms, ms * 1e-3 * 2.4e9 / (iters * 32.0 * bpc));
}

int main() {
    float *d_out, *d_src; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&d_src, 32768 * 16);
    hipMemset(d_src, 0, 32768 * 16);
    for (int w = 1; w <= 2; ++w) {
        run<0>("MFMA only", w, d_out, d_src);
        run<1>("+ 2 v_fma_f32 per MFMA", w, d_out, d_src);
        run<3>("+ 0.5 ds_read_b128 (+ 0.5 v_add) per MFMA", w, d_out, d_src);
        run<4>("+ 1 ds_read_b32 (+ 1 v_add) per MFMA", w, d_out, d_src);
        run<5>("+ 1 s_barrier per 32 MFMA", w, d_out, d_src);
        run<6>("+ 1 global float4 load per 4 MFMA (+ 1 v_add)", w, d_out, d_src);
    }
    return 0;
}
