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

// ---- deterministic mode (bh_set_deterministic(1), include/bihome.h): every cross-workgroup accumulation becomes order-independent ----
// A floating-point atomic add depends on the order in which the workgroups arrive.  In deterministic mode a sum lives in an ENTRY
// of four 8-byte words: word 0 a plain double (the default mode's atomic target; in deterministic mode only the escape for
// non-finite / huge addends), words 1-3 three int64 limbs on the grids 2^0, 2^-40, 2^-80.  An addend v is cut EXACTLY into
// trunc(v), trunc(frac * 2^40), rint(rest * 2^40) and the three integers are added with integer atomics - integer addition is
// associative, so the total is the same bit pattern whatever the arrival order (range 2^62, resolution 2^-80 ~ 8e-25; each
// addend's own rounding onto that grid is a function of the addend alone).  Readers add word 0 and the limbs; a caller-zeroed
// entry that was only written in the default mode has zero limbs, so one reader serves both modes.
#define BH_ACC_WORDS 4
// (round 4: the mode is a PER-CALL bit - BH_ROUTE_DETERMINISTIC in bh_conv_desc.route, BH_BN_DETERMINISTIC in the BatchNorm flags,
//  BH_F_DETERMINISTIC in the flags argument of the bh_*_f entry points; the library holds no mode of its own)
__device__ __forceinline__ void bh_det_add(double* entry, double v) {
    if (!(fabs(v) < 0x1p62)) { atomicAdd(entry, v); return; }  // inf / NaN / out of range: stays visible in word 0
    const double hi = trunc(v);
    const double r1 = (v - hi) * 0x1p40;                       // exact
    const double mid = trunc(r1);
    const double lo = rint((r1 - mid) * 0x1p40);
    unsigned long long* const w = reinterpret_cast<unsigned long long*>(entry);
    if (hi != 0.0) atomicAdd(w + 1, (unsigned long long)(long long)hi);
    if (mid != 0.0) atomicAdd(w + 2, (unsigned long long)(long long)mid);
    if (lo != 0.0) atomicAdd(w + 3, (unsigned long long)(long long)lo);
}
// det: 1 = the limbs are in use, 0 = word 0 only (the writer's mode is known: no limb loads), -1 = look (cold paths)
__device__ __forceinline__ double bh_acc_read(const double* __restrict__ entry, int det = -1) {
    double t = entry[0];
#ifndef BH_ACC_LOOK                                             // (A/B builds: always look at the limbs, as the first version did)
    if (det == 0) return t;
#endif
    const long long* const w = reinterpret_cast<const long long*>(entry);
    const long long l0 = w[1], l1 = w[2], l2 = w[3];
    if (l0 | l1 | l2) t += ((double)l2 * 0x1p-40 + (double)l1) * 0x1p-40 + (double)l0;      // smallest limb first
    return t;
}
// accumulate into entry e of a sums buffer in either mode
__device__ __forceinline__ void bh_acc_add(double* entry, double v, int det) {
    if (det) bh_det_add(entry, v); else atomicAdd(entry, v);
}

__device__ __forceinline__ size_t bn_sum_index(int slot, int groups, int grp, int C, int c, int mom) {
    return ((((size_t)slot * groups + grp) * C + c) * 2 + mom) * BH_BN_SUM_STRIDE;
}
// (an entry of a BatchNorm sums buffer is the first BH_ACC_WORDS words of its 128-byte line)
__device__ __forceinline__ double bn_sum_total(const double* __restrict__ buf, int groups, int grp, int C, int c, int mom, int det = -1) {
    double t = 0;
#pragma unroll
    for (int s = 0; s < BH_BN_SUM_SLOTS; ++s) t += bh_acc_read(buf + bn_sum_index(s, groups, grp, C, c, mom), det);
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

// X3 ("f32x3": fp32 result accuracy on the bf16 matrix pipe): every fp32 operand x is cut into three bf16 pieces
// x = hi + mid + lo (truncation: 8 + 8 + 8 significand bits, so the sum is EXACT), and a product a*b is evaluated as the six
// partial products of total order <= 2 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi) on v_mfma_f32_32x32x16_bf16 with fp32
// accumulate; the three dropped ones are below 2^-23 |a*b| - the size of one fp32 rounding of the product.  Six 32-cycle
// MFMAs contract 16 channels that cost eight 64-cycle v_mfma_f32_32x32x2_f32: 2.67x less matrix-pipe time.
// bh_split8: 8 floats -> the 8 bf16 of each piece, element e in the low / high half of dword e/2 (4.5 VALU per element).
typedef __bf16 bh_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bh_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bh_split8(const float4& u, const float4& v, uint4& hi, uint4& mid, uint4& lo) {
#ifdef BH_SPLIT_TRUNC
    // (round 2 form, A/B builds: all three pieces by truncation)
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned xb[8], rb[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        xb[e] = __builtin_bit_cast(unsigned, x[e]);
        const float r = x[e] - __builtin_bit_cast(float, xb[e] & 0xFFFF0000u);        // exact
        rb[e] = __builtin_bit_cast(unsigned, r);
        const float t = r - __builtin_bit_cast(float, rb[e] & 0xFFFF0000u);           // exact, <= 8 significant bits
        sb[e] = __builtin_bit_cast(unsigned, t);
    }
    // v_perm_b32: bytes 7,6 of {S0,S1} = high half of S0, bytes 3,2 = high half of S1
    hi = make_uint4(__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u),
                    __builtin_amdgcn_perm(xb[5], xb[4], 0x07060302u), __builtin_amdgcn_perm(xb[7], xb[6], 0x07060302u));
    mid = make_uint4(__builtin_amdgcn_perm(rb[1], rb[0], 0x07060302u), __builtin_amdgcn_perm(rb[3], rb[2], 0x07060302u),
                     __builtin_amdgcn_perm(rb[5], rb[4], 0x07060302u), __builtin_amdgcn_perm(rb[7], rb[6], 0x07060302u));
    lo = make_uint4(__builtin_amdgcn_perm(sb[1], sb[0], 0x07060302u), __builtin_amdgcn_perm(sb[3], sb[2], 0x07060302u),
                    __builtin_amdgcn_perm(sb[5], sb[4], 0x07060302u), __builtin_amdgcn_perm(sb[7], sb[6], 0x07060302u));
#else
    // Round 3: hi by truncation (no overflow near FLT_MAX), mid and lo by v_cvt_pk_bf16_f32 (round to nearest even, two elements per
    // instruction) - 4.5 instead of 5.5 VALU per element, and STILL EXACT: r = x - hi has <= 16 significant bits below hi's last place;
    // mid = RNE(r) keeps r's top 8, so t = r - mid is at most half a unit of mid's last place - <= 8 significant bits and a sign, a bf16
    // number - and lo = RNE(t) = t.  (A carry in the rounding makes mid a power of two and t negative: still <= 8 bits.)
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[2 * e]), b1 = __builtin_bit_cast(unsigned, x[2 * e + 1]);
        h[e] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);       // high halves of (x[2e+1], x[2e]): element 2e in the low half
        const bh_f32x2 r = {x[2 * e] - __builtin_bit_cast(float, b0 & 0xFFFF0000u), x[2 * e + 1] - __builtin_bit_cast(float, b1 & 0xFFFF0000u)};
        m[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bh_bf16x2));
        const bh_f32x2 t = {r[0] - __builtin_bit_cast(float, m[e] << 16), r[1] - __builtin_bit_cast(float, m[e] & 0xFFFF0000u)};      // exact
        l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bh_bf16x2));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    mid = make_uint4(m[0], m[1], m[2], m[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
#endif
}

// four floats -> the four bf16 of each piece (the generic implicit-GEMM kernel stages float4s), and one float (its transposed B path)
__device__ __forceinline__ void bh_split4(const float4& u, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {u.x, u.y, u.z, u.w};
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned b0 = __builtin_bit_cast(unsigned, x[2 * e]), b1 = __builtin_bit_cast(unsigned, x[2 * e + 1]);
        h[e] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const bh_f32x2 r = {x[2 * e] - __builtin_bit_cast(float, b0 & 0xFFFF0000u), x[2 * e + 1] - __builtin_bit_cast(float, b1 & 0xFFFF0000u)};
        m[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bh_bf16x2));
        const bh_f32x2 t = {r[0] - __builtin_bit_cast(float, m[e] << 16), r[1] - __builtin_bit_cast(float, m[e] & 0xFFFF0000u)};
        l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bh_bf16x2));
    }
    hi = make_uint2(h[0], h[1]); mid = make_uint2(m[0], m[1]); lo = make_uint2(l[0], l[1]);
}
__device__ __forceinline__ void bh_split1(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned b = __builtin_bit_cast(unsigned, x);
    hi = (unsigned short)(b >> 16);
    const float r = x - __builtin_bit_cast(float, b & 0xFFFF0000u);
    const __bf16 m = (__bf16)r;
    mid = __builtin_bit_cast(unsigned short, m);
    const float t = r - (float)m;
    lo = __builtin_bit_cast(unsigned short, (__bf16)t);
}

// X2 ("f32x2": bh_conv_desc.precision = 3): two bf16 pieces per operand, both ROUNDED to nearest even (v_cvt_pk_bf16_f32):
// x = hi + mid + e with |e| <= 2^-18 |x| and zero mean, and a product a*b is evaluated as hi*hi + hi*mid + mid*hi (three MFMAs;
// the dropped mid*mid is <= 2^-18 |a*b| as well).  Error ~4e-6 per product: ~13x the fp32 rounding, ~500x below the bf16-operand
// mode - a separately reported arithmetic, never the default.  2.5 VALU per element.
__device__ __forceinline__ void bh_split2_pair(float a, float b, unsigned& hi, unsigned& mid) {
    const bh_f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bh_bf16x2));
    const bh_f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u)};     // exact
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bh_bf16x2));
}
__device__ __forceinline__ void bh_split8_2(const float4& u, const float4& v, uint4& hi, uint4& mid) {
    bh_split2_pair(u.x, u.y, hi.x, mid.x);
    bh_split2_pair(u.z, u.w, hi.y, mid.y);
    bh_split2_pair(v.x, v.y, hi.z, mid.z);
    bh_split2_pair(v.z, v.w, hi.w, mid.w);
}
// NP pieces of 8 floats: p[0] = hi, p[1] = mid (, p[2] = lo)
template <int NP>
__device__ __forceinline__ void bh_split8_np(const float4& u, const float4& v, uint4 (&p)[3]) {
    if constexpr (NP == 3) bh_split8(u, v, p[0], p[1], p[2]);
    else bh_split8_2(u, v, p[0], p[1]);
}

// F16X2 ("f16x2": bh_conv_desc.precision = 4, round 4): two FP16 pieces per operand with a power-of-two scale per tensor.
//   x * 2^k = hi + lo + e,  hi = rn16(x 2^k), lo = rn16(x 2^k - hi)  (the difference is exact in fp32): 11 + 11 significand bits and a
// sign, |e| <= 2^-23 |x 2^k| as long as lo is a normal fp16 number, <= 2^-25 (absolute, scaled units) below that.  A product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (each piece product exact in the fp32 accumulate; tools/f16_probe.hip: subnormal
// fp16 inputs are NOT flushed); the dropped lo*lo is <= 2^-22 |a b|.  Per product ~2^-22: 16x below f32x2, one or two fp32 roundings.
// The scale makes max |x 2^k| land in [2^14, 2^15) (fp16 overflows at 65504): k comes from a MAGNITUDE RECORD of the tensor - an
// upper bound of max |x| that its producer left (BatchNorm apply kernels: measured; BatchNorm-on-load: |gamma| sqrt(rows) + |beta|;
// bh_absmax: a streaming pass) - BH_AMAX_SLOTS words, one per 128-byte line, combined with max (workgroups of a producer spread their
// integer atomic max over the slots).  Elements within 2^-18 of the bound keep all 22 bits; smaller ones degrade to an absolute error of
// 2^-40 of the bound.  Results are rescaled by 2^-(ka + kb) in the epilogue (v_ldexp_f32: exact).
#define BH_AMAX_SLOTS 16
#define BH_AMAX_STRIDE 32                   // words between slots (128 bytes)
#define BH_AMAX_WORDS (BH_AMAX_SLOTS * BH_AMAX_STRIDE)
typedef _Float16 bh_f16x2 __attribute__((ext_vector_type(2)));
// scale exponent for a tensor whose magnitudes are bounded by the float with these bits (>= 0): bound * 2^k in [2^14, 2^15)
__host__ __device__ __forceinline__ int bh_f16_scale_exp(unsigned bound_bits) {
    const int e = (int)((bound_bits >> 23) & 0xFFu);          // biased exponent (0: zero / subnormal bound)
    int k = 14 - (e - 127);
    if (e == 0xFF) k = 0;                                      // inf / NaN bound: no scaling, the non-finite values propagate
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
// all 64 lanes: the record's maximum (lanes 0..15 load one slot each)
__device__ __forceinline__ unsigned bh_amax_read(const unsigned* __restrict__ rec, int lane) {
    unsigned v = lane < BH_AMAX_SLOTS ? rec[lane * BH_AMAX_STRIDE] : 0u;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, off, 64); v = o > v ? o : v; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// a workgroup's contribution: every thread passes its own maximum of |values| (as float); one atomic per workgroup
__device__ __forceinline__ void bh_amax_commit(unsigned* __restrict__ rec, float m, unsigned slot_seed, float* sm4 /* LDS, >= blockDim/64 floats */) {
    m = wave_max(m);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) sm4[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, sm4[w]);
        // NaN: fmaxf drops it; a NaN in the data shows up through the products themselves.  (bits of a non-negative float order like integers)
        atomicMax(rec + (slot_seed % BH_AMAX_SLOTS) * BH_AMAX_STRIDE, __builtin_bit_cast(unsigned, m));
    }
}
__device__ __forceinline__ void bh_split2_pair_f16(float a, float b, float s, unsigned& hi, unsigned& lo) {
    const bh_f32x2 v = {a * s, b * s};
    const bh_f16x2 h = __builtin_convertvector(v, bh_f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const bh_f32x2 r = {__builtin_fmaf(a, s, -(float)h[0]), __builtin_fmaf(b, s, -(float)h[1])};      // exact
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bh_f16x2));
}
__device__ __forceinline__ void bh_split8_f16(const float4& u, const float4& v, float s, uint4& hi, uint4& lo) {
    bh_split2_pair_f16(u.x, u.y, s, hi.x, lo.x);
    bh_split2_pair_f16(u.z, u.w, s, hi.y, lo.y);
    bh_split2_pair_f16(v.x, v.y, s, hi.z, lo.z);
    bh_split2_pair_f16(v.z, v.w, s, hi.w, lo.w);
}
// NP pieces of 8 floats in the arithmetic of the kernel: F16 = false -> bf16 pieces (s unused), true -> two fp16 pieces of x * s
template <int NP, bool F16>
__device__ __forceinline__ void bh_split8_any(const float4& u, const float4& v, float s, uint4 (&p)[3]) {
    if constexpr (F16) { static_assert(NP == 2, "fp16 pieces: two"); bh_split8_f16(u, v, s, p[0], p[1]); }
    else bh_split8_np<NP>(u, v, p);
}
