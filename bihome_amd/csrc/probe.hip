// Measured peaks for bench.py's roofline ("measure achievable ... MFMA ... peaks with a microbenchmark in the same run and
// report fractions of both vendor and measured peaks").  A bare stream of dependent-free v_mfma_f32_32x32x16_bf16 on the whole chip, two
// workgroups of four waves per CU (the occupancy of the f32x3 kernels), with constant or with random-bit operands: on MI355X the second is
// what real tensors look like to the matrix pipe, and its sustained rate is 25-35 % below the first - the pipe is power-limited once its
// operands toggle (tools/mfma_clock_probe.hip for the clock-resolved version).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned probe_mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) probe_mfma_kernel(float* sink, int iters, int random) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        // random sign / mantissa bits with exponents near 1.0 (nothing overflows), or the constant 1.0
        auto rnd = [&](unsigned k) { const unsigned h = probe_mix(threadIdx.x * 131u + u * 17u + k); return random ? ((h & 0x807f807fu) | 0x3f003f00u) : 0x3f803f80u; };
        a[u] = make_uint4(rnd(1), rnd(2), rnd(3), rnd(4)); b[u] = make_uint4(rnd(5), rnd(6), rnd(7), rnd(8));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u + i) & 7]), __builtin_bit_cast(bf16x8, b[(u + 2 * i) & 7]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int bh_probe_mfma_bf16(int random_operands, float* sink_dev, double* tflops, void* stream) {
    if (!sink_dev || !tflops) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    const int iters = 1024, nwg = 512;                       // 32 MFMAs per iteration and wave, ~1 ms per launch
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);       // warm-up (clock settles)
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);
    (void)hipEventRecord(e0, s);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_mfma_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);
    (void)hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return (int)e;
    *tflops = 3.0 * nwg * 4.0 * iters * 32.0 * 32768.0 / (ms * 1e-3) * 1e-12;
    return BH_OK;
}

// the same for the fp32-input instruction the generic / stem / small-channel kernels issue (v_mfma_f32_32x32x2_f32, 64 cycles): round-3
// ADVICE - rooflines bound by the fp32 matrix pipe had no measured peak to relate to
__global__ void __launch_bounds__(256) probe_mfma_f32_kernel(float* sink, int iters, int random) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        auto rnd = [&](unsigned k) {
            const unsigned h = probe_mix(threadIdx.x * 131u + u * 17u + k);
            return random ? __builtin_bit_cast(float, (h & 0x807fffffu) | 0x3f000000u) : 1.0f;
        };
        a[u] = rnd(1); b[u] = rnd(5);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 2 * i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int bh_probe_mfma_f32(int random_operands, float* sink_dev, double* tflops, void* stream) {
    if (!sink_dev || !tflops) return BH_E_BADARG;
    hipStream_t s = bh_stream(stream);
    const int iters = 512, nwg = 512;                        // 32 MFMAs of 4096 flops per iteration and wave
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);
    hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);
    (void)hipEventRecord(e0, s);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(nwg), dim3(256), 0, s, sink_dev, iters, random_operands);
    (void)hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return (int)e;
    *tflops = 3.0 * nwg * 4.0 * iters * 32.0 * 4096.0 / (ms * 1e-3) * 1e-12;
    return BH_OK;
}

