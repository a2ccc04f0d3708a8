// How fast does a dense stream of v_mfma_f32_32x32x16_bf16 REALLY run, and what lowers the clock it runs at?  One workgroup of four
// waves per CU (or two), N dependent-free MFMAs on four accumulators; the kernel reads the shader-clock counter (s_memtime) and the
// 100 MHz wall clock (s_memrealtime) around the stream: cycles per MFMA (pipe occupancy) and ns per MFMA (wall) separately - their
// ratio is the clock the SIMD actually ran at.  Modes: 0 constant operands, 1 eight sets of random-bit operands (data toggling),
// 2 = 1 + one ds_read_b128 per two MFMAs (the f32x3 3x3 kernel's LDS traffic), 3 = 2 + one L2-resident buffer_load_dwordx4 per four MFMAs.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o tools/mfma_clock_probe.bin && tools/mfma_clock_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, float* sink, const uint4* __restrict__ gsrc, int iters) {
    __shared__ uint4 lds[2048];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        const unsigned s = MODE == 0 ? 0x3f803f80u : 0;
        // random sign / mantissa bits, exponents near 1.0 so that nothing overflows: 0x3f80 | random low 7 bits | random sign
        auto rnd = [&](unsigned k) { const unsigned h = mix(threadIdx.x * 131u + u * 17u + k); return MODE == 0 ? s : ((h & 0x807f807fu) | 0x3f003f00u); };
        a[u] = make_uint4(rnd(1), rnd(2), rnd(3), rnd(4)); b[u] = make_uint4(rnd(5), rnd(6), rnd(7), rnd(8));
    }
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(mix(i), mix(i + 7), mix(i + 13), mix(i + 29));
    __syncthreads();
    uint4 l0 = a[0], g0 = b[0];
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE >= 2 && (u & 1) == 0) { l0 = lds[(threadIdx.x + 64 * u + 16 * it) & 2047]; }
            if (MODE >= 2 && (u & 1) == 1) { const uint4 l1 = lds[(threadIdx.x * 3 + 64 * u + 16 * it) & 2047]; a[u].x ^= (l0.x ^ l1.y) & 0x007f007fu; }
            if (MODE >= 3 && (u & 3) == 0) g0 = gsrc[(blockIdx.x * 4096 + threadIdx.x + 256 * ((it + u) & 15)) & ((1 << 20) - 1)];
            if (MODE >= 3 && (u & 3) == 3) b[u].y ^= g0.x & 0x007f007fu;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u + i) & 7]), __builtin_bit_cast(bf16x8, b[(u + 2 * i) & 7]), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x % 64 == 0) { const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[w * 2] = c1 - c0; out[w * 2 + 1] = w1 - w0; }
    if (s == 12345.678f) sink[0] = s;
}
int main(int argc, char** argv) {
    unsigned long long* d; float* sink; uint4* gsrc;
    hipMalloc(&d, 1 << 20); hipMalloc(&sink, 4); hipMalloc(&gsrc, 16 << 20); hipMemset(gsrc, 0x3c, 16 << 20);
    const char* names[4] = {"constant operands", "random-bit operands", "random + 0.5 ds_read_b128 / MFMA", "random + LDS + 0.25 buffer_load / MFMA"};
    const int ncu = argc > 1 ? atoi(argv[1]) : 256;       // compute units to occupy (workgroups = ncu x workgroups per CU)
    for (int mode = 0; mode < 4; ++mode) for (int wg_per_cu : {1, 2}) {
        const int iters = 2048, nwg = ncu * wg_per_cu, nw = nwg * 4;
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), 0, 0, d, sink, gsrc, iters);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, 0, d, sink, gsrc, iters);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(256), 0, 0, d, sink, gsrc, iters);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(nwg), dim3(256), 0, 0, d, sink, gsrc, iters);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw * 2);
        hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ns;
        const double n = 32.0 * iters;
        for (int w = 0; w < nw; ++w) { cyc.push_back(h[2 * w] / n); ns.push_back(h[2 * w + 1] * 10.0 / n); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
        printf("%-40s %d WG/CU: ticks per MFMA per wave p50 %.2f; wall ns per MFMA per wave p50 %.2f; clock %.0f MHz; chip rate %.0f TFLOP/s\n",
               names[mode], wg_per_cu, cyc[nw / 2], ns[nw / 2], 1e3 * cyc[nw / 2] / ns[nw / 2], 4.0 * ncu * wg_per_cu * 32768.0 / ns[nw / 2] * 1e-3);
    }
    return 0;
}
