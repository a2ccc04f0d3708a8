// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bihome.h"

#define BH_LAUNCH_CHECK()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

// BatchNorm sums buffers: entry (group, channel, moment) lives at double index ((group*C + channel)*2 + moment) *
// BH_BN_SUM_STRIDE, i.e. on its own 128-byte line - f64 atomics to one line are serialised by the memory-side atomic
// unit, and a whole reduce kernel's workgroups arrive within a few microseconds of each other
#define BH_BN_SUM_STRIDE 16
// (replicating the table over several slots was measured too: the consumers' extra strided loads cost more than the
// shorter atomic tail saves, so there is one slot)
#define BH_BN_SUM_SLOTS 1
#define BH_BN_SUM_DOUBLES(groups, C) ((size_t)BH_BN_SUM_SLOTS * (groups) * (C) * 2 * BH_BN_SUM_STRIDE)

static inline hipStream_t bh_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Tuning knobs: compile-time constants in the product library, process-global variables only under -DBH_TUNING
#ifdef BH_TUNING
#define BH_KNOB(name, val) int name = val
#define BH_KNOB_EXTERN(name) extern int name
#else
#define BH_KNOB(name, val) static constexpr int name = val
#define BH_KNOB_EXTERN(name) static_assert(true, "")
#endif

// bh_conv_variant: while a query is active on this thread, dispatchers record the kernel they would launch and return
struct BhQuery { char name[256]; int len; };
extern thread_local BhQuery* bh_query_ctx;
bool bh_query(const char* fmt, ...);       // true (and the name appended) when a query is active: the caller skips its launch

// per-device one-time flags (hipFuncSetAttribute is per device)
static inline bool bh_device_once(unsigned long long& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    if (mask & (1ull << dev)) return false;
    mask |= 1ull << dev;
    return true;
}

__device__ __forceinline__ size_t bn_sum_index(int slot, int groups, int grp, int C, int c, int mom) {
    return ((((size_t)slot * groups + grp) * C + c) * 2 + mom) * BH_BN_SUM_STRIDE;
}
__device__ __forceinline__ double bn_sum_total(const double* __restrict__ buf, int groups, int grp, int C, int c, int mom) {
    double t = 0;
#pragma unroll
    for (int s = 0; s < BH_BN_SUM_SLOTS; ++s) t += buf[bn_sum_index(s, groups, grp, C, c, mom)];
    return t;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// 3x3 row-major helpers (double)
__device__ __forceinline__ void mat3_mul(const double* a, const double* b, double* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
