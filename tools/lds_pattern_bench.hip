// ds_read_b128 bank-conflict probe: every lane reads 16 bytes at off[lane] (+ a rotating multiple of 4 KB) from a 64 KB LDS
// array; patterns are built on the host.  hipcc --offload-arch=gfx950 -O3 tools/lds_pattern_bench.hip -o /tmp/ldsb && /tmp/ldsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

__global__ void __launch_bounds__(256) probe(const int* __restrict__ off, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = (float)i;
    __syncthreads();
    const int o = off[threadIdx.x & 63];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = *reinterpret_cast<const float4*>(sm + ((o + u * 4096 + ((it & 31) << 8)) & 65535));   // + multiples of 256 B: same banks
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    struct Pat { std::string name; std::vector<int> off; };
    std::vector<Pat> pats;
    auto mk = [&](const char* name, auto f) { Pat p; p.name = name; for (int l = 0; l < 64; ++l) p.off.push_back(f(l)); pats.push_back(p); };
    mk("contiguous 16B/lane", [](int l) { return l * 16; });
    mk("halo A: 8px rows, pitch 160B, kh2 +3200B", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 10 + (l31 & 7)) * 16; });
    mk("halo A: 8px rows, pitch 160B, kh2 +1600B", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 100 + (l31 >> 3) * 10 + (l31 & 7)) * 16; });
    mk("8px rows, pitch 144B (9 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 9 + (l31 & 7)) * 16; });
    mk("8px rows, pitch 176B (11 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 11 + (l31 & 7)) * 16; });
    mk("8px rows, pitch 192B (12 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 12 + (l31 & 7)) * 16; });
    mk("8px rows, pitch 256B (16 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 16 + (l31 & 7)) * 16; });
    mk("8px rows, pitch 128B (dense)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 3) * 8 + (l31 & 7)) * 16; });
    mk("16px rows, pitch 288B (18 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 4) * 18 + (l31 & 15)) * 16; });
    mk("4px rows, pitch 96B (6 px)", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 200 + (l31 >> 2) * 6 + (l31 & 3)) * 16; });
    mk("8px rows pitch 160B, kh2 +3200+64B", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 204 + (l31 >> 3) * 10 + (l31 & 7)) * 16; });
    mk("8px rows pitch 160B, kh2 +3200+128B", [](int l) { int l31 = l & 31, kh2 = l >> 5; return (kh2 * 208 + (l31 >> 3) * 10 + (l31 & 7)) * 16; });
    mk("halo A new: 8x4 strip, rows permuted, pitch 160B", [](int l) { int l31 = l & 31, kh2 = l >> 5, q = l31 >> 2; int row = q ^ (((q >> 1) ^ (q >> 2)) & 1); return (kh2 * 200 + row * 10 + (l31 & 3)) * 16; });
    mk("8x4 strip, rows in order, pitch 160B", [](int l) { int l31 = l & 31, kh2 = l >> 5, q = l31 >> 2; return (kh2 * 200 + q * 10 + (l31 & 3)) * 16; });
    mk("all lanes same address (broadcast)", [](int l) { return 0; });
    mk("stride 32B", [](int l) { return l * 32; });
    mk("stride 64B", [](int l) { return l * 64; });
    int* d_off; float* d_out;
    hipMalloc(&d_off, 64 * sizeof(int)); hipMalloc(&d_out, 256 * 256 * sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    for (auto& p : pats) {
        hipMemcpy(d_off, p.off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, d_off, 64, d_out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, d_off, iters, d_out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per CU: 4 waves x iters x 8 reads; cycles at 2.4 GHz
        const double reads = 4.0 * iters * 8;
        printf("%-44s %8.3f ms  %6.2f cycles/read (2.4 GHz)  %6.1f B/clk/CU\n", p.name.c_str(), ms, ms * 1e-3 * 2.4e9 / reads, reads * 1024 / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
