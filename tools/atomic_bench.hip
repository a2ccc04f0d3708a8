// Device-scope atomicAdd throughput on MI355X: `blocks` workgroups x 256 threads each add `per_thread` values to a table
// of `naddr` floats (thread t of every block hits addresses t, t+256, ...): time against contention per address.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench
// Measured (round 1): ~270 G fp32 atomics/s chip-wide when spread over 36,864 addresses (i.e. ~1 lane-atomic per L2
// channel per clock, independent of agent / workgroup / wavefront scope); ~28 ns per atomic when 1024 workgroups hit
// the same address; f64: 512 workgroups x 128 addresses 14.5 us.  These set the design of the split-K wgrad flush and
// of the padded BatchNorm sums tables (csrc/common.h).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void burst(float* tab, int naddr, int per_thread) {
    for (int i = 0; i < per_thread; ++i) atomicAdd(&tab[(threadIdx.x + i * 256) % naddr], 1.0f);
}
__global__ void burstd(double* tab, int naddr, int per_thread) {
    for (int i = 0; i < per_thread; ++i) atomicAdd(&tab[(threadIdx.x + i * 256) % naddr], 1.0);
}
int main() {
    float* tab;
    if (hipMalloc(&tab, 1 << 24) != hipSuccess || hipMemset(tab, 0, 1 << 24) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int cfgs[][3] = {{512, 36864, 144}, {256, 36864, 144}, {64, 36864, 144}, {512, 4096, 16}, {512, 128, 1}, {1024, 256, 1}};
    for (auto& c : cfgs) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            burst<<<c[0], 256>>>(tab, c[1], c[2]);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)c[0] * 256 * c[2];
        printf("f32 blocks %4d naddr %6d per_thread %3d : %8.1f us  %.1f G atomics/s  %d per address -> %.1f ns each\n", c[0], c[1],
               c[2], ms * 1e3, n / ms / 1e6, (int)(n / c[1]), ms * 1e6 / (n / c[1]));
    }
    int cd[][2] = {{512, 128}, {64, 128}};
    for (auto& c : cd) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            burstd<<<c[0], 128>>>((double*)tab, c[1], 1);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("f64 blocks %4d naddr %d: %.1f us\n", c[0], c[1], ms * 1e3);
    }
    return 0;
}
