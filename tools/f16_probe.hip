// What the two-piece fp16 split ("f16x2", bh_conv_desc.precision = 4) relies on, settled on the hardware:
//   1. v_mfma_f32_32x32x16_f16 does NOT flush subnormal fp16 inputs (the low piece of a small element is subnormal);
//   2. products of fp16 numbers (11 x 11 significand bits) are exact in the fp32 accumulate, and a K = 16 sum against float64;
//   3. the sustained rate of the f16 MFMA on random-bit operands against the bf16 one (the pipe is power-limited on toggling
//      operands, DESIGN.md 5.2: an 11-bit multiplier array may draw more than an 8-bit one).
// hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o tools/f16_probe.bin && tools/f16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __host__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one wave: D = A (32 x 16, row = lane & 31, k = 8 (lane >> 5) + e) x B (16 x 32, col = lane & 31) + 0
__global__ void mm_probe(const unsigned short* a, const unsigned short* b, float* d) {
    const int lane = threadIdx.x;
    f16x8 av, bv;
    for (int e = 0; e < 8; ++e) {
        av[e] = __builtin_bit_cast(_Float16, a[(lane & 31) * 16 + 8 * (lane >> 5) + e]);
        bv[e] = __builtin_bit_cast(_Float16, b[(8 * (lane >> 5) + e) * 32 + (lane & 31)]);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

static double h2d(unsigned short h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp(1.0 + m / 1024.0, e - 15);
    return s ? -v : v;
}

template <bool F16, int MODE>
__global__ void __launch_bounds__(256) rate_probe(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        auto rnd = [&](unsigned k) {
            const unsigned h = mix(threadIdx.x * 131u + u * 17u + k);
            if (MODE == 0) return F16 ? 0x3c003c00u : 0x3f803f80u;
            return F16 ? ((h & 0x83ff83ffu) | 0x38003800u) : ((h & 0x807f807fu) | 0x3f003f00u);      // random sign + mantissa, |x| in [0.5, 1)
        };
        a[u] = make_uint4(rnd(1), rnd(2), rnd(3), rnd(4)); b[u] = make_uint4(rnd(5), rnd(6), rnd(7), rnd(8));
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + i) & 7]), __builtin_bit_cast(f16x8, b[(u + 2 * i) & 7]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u + i) & 7]), __builtin_bit_cast(bf16x8, b[(u + 2 * i) & 7]), acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x % 64 == 0) { const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[w * 2] = c1 - c0; out[w * 2 + 1] = w1 - w0; }
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    unsigned short ha[512], hb[512];
    float hd[1024];
    unsigned short *da, *db; float* dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 4096);
    auto run = [&]() {
        hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost);
    };
    // 1. subnormal inputs: A = 2^-20 (fp16 subnormal 0x0010), B = 1 -> 16 * 2^-20 = 2^-16; A = B = 2^-20 -> 16 * 2^-40
    for (int i = 0; i < 512; ++i) { ha[i] = 0x0010; hb[i] = 0x3c00; }
    run();
    printf("subnormal A x 1.0      : D[0] = %.9g (expected %.9g)  %s\n", hd[0], std::ldexp(1.0, -16), hd[0] == std::ldexp(1.0f, -16) ? "NOT flushed" : "FLUSHED / wrong");
    for (int i = 0; i < 512; ++i) { ha[i] = 0x0010; hb[i] = 0x0010; }
    run();
    printf("subnormal A x subn. B  : D[0] = %.9g (expected %.9g)  %s\n", hd[0], std::ldexp(1.0, -36), hd[0] == std::ldexp(1.0f, -36) ? "NOT flushed" : "FLUSHED / wrong");
    for (int i = 0; i < 512; ++i) { ha[i] = 0x0001; hb[i] = 0x0001; }
    run();
    printf("min subnormal squared  : D[0] = %.9g (expected %.9g)\n", hd[0], std::ldexp(16.0, -48));
    for (int i = 0; i < 512; ++i) { ha[i] = 0x7bff; hb[i] = 0x7bff; }       // 65504^2 * 16 = 6.9e10: fine in fp32
    run();
    printf("max finite squared x16 : D[0] = %.9g (expected %.9g)\n", hd[0], 16.0 * 65504.0 * 65504.0);
    // 2. random operands over 16 binades, all mantissa bits: error of the K = 16 sum against float64, relative to sum |a b|
    double worst = 0, worst_exact = 0;
    for (int trial = 0; trial < 64; ++trial) {
        for (int i = 0; i < 512; ++i) {
            const unsigned r = mix(trial * 1024 + i), q = mix(77777 + trial * 1024 + i);
            ha[i] = (unsigned short)((r & 0x83ff) | ((7 + ((r >> 16) % 16)) << 10));
            hb[i] = (unsigned short)((q & 0x83ff) | ((7 + ((q >> 16) % 16)) << 10));
        }
        run();
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            double s = 0, sa = 0;
            float chain = 0.f;
            for (int k = 0; k < 16; ++k) { const double p = h2d(ha[m * 16 + k]) * h2d(hb[k * 32 + n]); s += p; sa += std::fabs(p); chain = fmaf((float)h2d(ha[m * 16 + k]), (float)h2d(hb[k * 32 + n]), chain); }
            worst = std::max(worst, std::fabs(hd[m * 32 + n] - s) / sa);
            worst_exact = std::max(worst_exact, std::fabs((double)chain - s) / sa);
        }
    }
    printf("random K=16 sums       : max |D - f64| / sum|ab| = %.3g (2^%.1f); an fp32 fmaf chain: %.3g\n", worst, std::log2(worst), worst_exact);
    // 3. rate on the whole chip: f16 against bf16, constant and random-bit operands
    unsigned long long* d; float* sink;
    hipMalloc(&d, 1 << 20); hipMalloc(&sink, 4);
    for (int f16 = 0; f16 < 2; ++f16) for (int mode = 0; mode < 2; ++mode) for (int wg_per_cu : {1, 2}) {
        const int iters = 2048, nwg = 256 * wg_per_cu, nw = nwg * 4;
        for (int rep = 0; rep < 3; ++rep) {
            if (!f16 && mode == 0) hipLaunchKernelGGL((rate_probe<false, 0>), dim3(nwg), dim3(256), 0, 0, d, sink, iters);
            if (!f16 && mode == 1) hipLaunchKernelGGL((rate_probe<false, 1>), dim3(nwg), dim3(256), 0, 0, d, sink, iters);
            if (f16 && mode == 0) hipLaunchKernelGGL((rate_probe<true, 0>), dim3(nwg), dim3(256), 0, 0, d, sink, iters);
            if (f16 && mode == 1) hipLaunchKernelGGL((rate_probe<true, 1>), dim3(nwg), dim3(256), 0, 0, d, sink, iters);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw * 2);
        hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ns;
        const double n = 32.0 * iters;
        for (int w = 0; w < nw; ++w) { cyc.push_back(h[2 * w] / n); ns.push_back(h[2 * w + 1] * 10.0 / n); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
        printf("%s %-20s %d WG/CU: ticks per MFMA p50 %.2f; ns per MFMA p50 %.2f; clock %.0f MHz; chip rate %.0f TFLOP/s\n", f16 ? "f16 " : "bf16",
               mode ? "random-bit operands" : "constant operands", wg_per_cu, cyc[nw / 2], ns[nw / 2], 1e3 * cyc[nw / 2] / ns[nw / 2],
               4.0 * 256 * wg_per_cu * 32768.0 / ns[nw / 2] * 1e-3);
    }
    return 0;
}
