// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) in the fp16-piece arithmetic ("f16x2", bh_conv_desc.precision = 4) on PERSISTENT
// workgroups with SPECIALISED waves (round 5).
//
// conv3x3_halo_kernel (conv3x3.hip) runs a tile's three phases - stage-in, tap loop, store - one after the other in the same four waves, and
// the two workgroups of a CU run them in lockstep: on 128 x 32 x 32 x 64 the launch costs 15 us of stores + first stage, 15 us of MFMA and 11 us
// of in-loop loads, added up (profiles/r04_c3_f16_ablation.txt).  Here ONE workgroup of twelve waves owns a CU for the whole launch and walks
// its share of the (128-pixel x 64-channel) tiles; every wave has one job:
//   waves 0-3    C  consumers: ds_read_b128 fragments + v_mfma_f32_32x32x16_f16, nothing else.  Wave (wm, wn) owns sub-tile wm x channels
//                   [32 wn, 32 wn + 32) as before; a tap is two halves of six MFMAs (16-channel step 0 / 1), the fragments of the next half
//                   are requested while the current half multiplies (two register sets, no copies).  At a tile's end the accumulators go
//                   to an LDS hand-over tile (32 ds_write_b32 per wave) and the next tile starts at once.
//   waves 4-6    H  halo staging: a 32-channel chunk's two 10 x 10 halos global -> registers -> BatchNorm-on-load (optional) -> two fp16
//                   pieces -> LDS.  TWO chunks are in flight (a register set per halo stage): a slot round is cut, written, and the same
//                   registers request the round of the chunk after next - nothing else in these waves waits on the vector-memory counter,
//                   so every load has two chunk times (~3.5 us) to arrive.  Lane = (halo pixel, 8-channel plane): whole 128-byte lines.
//   wave  7      D  the packed weight fragments of kernel row r + 2 by LDS-DMA into a ring of three row slots (24 pieces of 1 KB per row;
//                   the only vector-memory traffic of this wave, so "the previous row has landed" is an exact counter wait).
//   waves 8-11   E  the previous tile's epilogue out of the hand-over tile: 2^-(ka + kb) rescale, bias, accumulate / residual / ReLU,
//                   BatchNorm statistics or BatchNorm-backward sums, with 16-byte loads and stores (lane = pixel x four channels: 1 KB
//                   contiguous per instruction; the halo kernel stored 4 bytes per lane).  What a unit reads from HBM is requested four
//                   units ahead; stores are never waited for.  The epilogue's options are a template argument (EM): no dead branches.
// Synchronisation is one s_barrier per kernel row (3 taps = 36 MFMAs per consumer wave), placed right after a consumer has REQUESTED the
// row's last fragments: the row's weight slot and - on the last row of a chunk - the chunk's halo stage are then free for the producers.
// Two halo stages, three weight-row slots, one hand-over tile: 158.5 KB of the CU's 160.
// The MFMA order per accumulator (chunk, tap, 16-channel step, products lo*hi, hi*lo, hi*hi) is that of conv3x3_halo_kernel: the
// convolution results are bit-identical to it (tests/test_conv_pc_gpu.py); the statistics sums are accumulated in double per element (the halo
// kernel: float per fragment quad, then double) and agree to rounding.
#include "conv3x3_args.h"
#include <type_traits>

namespace {
constexpr int PC_HPL = 200;                          // halo slots per k-plane: two sub-tiles x 10 x 10
constexpr int PC_PS = PC_HPL * 16 + 32;              // plane stride: +32 B so that the H waves' ds_write_b128 (lane = pixel x plane) spread over all banks
constexpr int PC_PIECE = 4 * PC_PS;                  // four 8-channel planes per fp16 piece
constexpr int PC_STAGE = 2 * PC_PIECE;               // one halo stage (hi / lo pieces): 25,856 B
constexpr int PC_BTAP = 8192;                        // one tap's weight fragments: [n tile 2][piece 2][16-channel step 2][lane] x 16 B
constexpr int PC_BROW = 3 * PC_BTAP;                 // one kernel row
constexpr int PC_OFF_B = 2 * PC_STAGE;
constexpr int PC_OFF_EPI = PC_OFF_B + 3 * PC_BROW;
constexpr int PC_EPI_BYTES = 128 * 64 * 4;           // hand-over tile [128 pixels][64 channels] fp32
constexpr int PC_OFF_RED = PC_OFF_EPI + PC_EPI_BYTES;
constexpr int PC_RED_BYTES = 4 * 128 * 8;            // statistics partials of the four E waves: [wave][channel 64][moment 2] doubles
constexpr int PC_LDS = PC_OFF_RED + PC_RED_BYTES;    // 162,304 B of 163,840
constexpr unsigned PC_XOOB = 0x80000000u;            // out-of-range buffer offset (tensor sizes are below 2^31): the load returns zeros

#ifdef BH_TUNING
// barrier time stamps of one workgroup (shader clock): [wave 0 .. 11][barrier number][arrival, release] (round 6: every wave, not one per role -
// tools/pc_timeline.py names the wave the others wait for)
__device__ unsigned long long g_pc_ts[12 * 160 * 2];
#define PC_BARRIER()                                                                                                    \
    do {                                                                                                                \
        const bool st_ = a.dbg_ts && blockIdx.x == (unsigned)(a.dbg_ts - 1) && (tid & 63) == 0;                        \
        if (st_ && pc_nb < 160) g_pc_ts[(wave * 160 + pc_nb) * 2] = __builtin_readcyclecounter();                       \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                \
        if (st_ && pc_nb < 160) g_pc_ts[(wave * 160 + pc_nb) * 2 + 1] = __builtin_readcyclecounter();                   \
        ++pc_nb;                                                                                                        \
    } while (0)
#else
#define PC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#ifndef PC_EMU
#define PC_EMU 0
#endif
#define PC_H_ASM 0        // (1: the H waves' loads as inline assembly with hand-counted waits - see the note in the H role; not yet sound)
#define PC_BARRIER_VM() asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// maximum over lanes 0 .. 15 of a wave (the sixteen slots of a magnitude record / the sixteen partial maxima behind packed weights)
__device__ __forceinline__ unsigned pc_amax_reduce(unsigned v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, off, 64); v = o > v ? o : v; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ void pc_decode(const C3Args& a, int g, int& img, int& ty, int& tx) {
    if (a.tpi_shift >= 0) { img = g >> a.tpi_shift; const int t = g & (a.tiles_per_img - 1); ty = t >> a.tx_shift; tx = t & (a.tiles_x - 1); }
    else { img = g / a.tiles_per_img; const int t = g - img * a.tiles_per_img; ty = t / a.tiles_x; tx = t - ty * a.tiles_x; }
}

// a.tpb: tiles per workgroup; a.gx_total: tile positions (two 8 x 8 sub-tiles each); tile index wt -> channel tile wt / gx_total, position
// wt % gx_total (a workgroup's consecutive tiles share their weights and their statistics entries)
// EM (epilogue mode, E waves): bit 0 statistics (forward sums / column sums / BatchNorm-backward sums), bit 1 one tensor added (the old gradient
// of a join, or the residual of the inference path), bit 2 BatchNorm-backward form (reads that BatchNorm's input z), bit 3 ... with the ReLU
// mask taken from its saved output y
template <bool DGRAD, bool BNI, int EM>
__global__ void __launch_bounds__(768) conv3x3_pc_kernel(C3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NY = a.Nn >> 6;
    const int total = a.gx_total * NY;
    const int wt0 = blockIdx.x * a.tpb;
    const int Tw = min(a.tpb, total - wt0);              // tiles of this workgroup (>= 1)
    const int nch = a.Kc >> 5;
    const int K = Tw * nch;                              // chunks
    const int R = K * 3;                                 // kernel rows = barrier phases
#ifdef BH_TUNING
    int pc_nb = 0;
#endif

    // wait until at most n vector-memory operations (the youngest n) are outstanding; everything the caller must not leave behind is older
    auto wait_vm = [&](int n) {
#define PC_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        switch (n < 0 ? 0 : (n > 48 ? 48 : n)) {
            PC_W(0) PC_W(1) PC_W(2) PC_W(3) PC_W(4) PC_W(5) PC_W(6) PC_W(7) PC_W(8) PC_W(9) PC_W(10) PC_W(11) PC_W(12) PC_W(13) PC_W(14) PC_W(15)
            PC_W(16) PC_W(17) PC_W(18) PC_W(19) PC_W(20) PC_W(21) PC_W(22) PC_W(23) PC_W(24) PC_W(25) PC_W(26) PC_W(27) PC_W(28) PC_W(29) PC_W(30)
            PC_W(31) PC_W(32) PC_W(33) PC_W(34) PC_W(35) PC_W(36) PC_W(37) PC_W(38) PC_W(39) PC_W(40) PC_W(41) PC_W(42) PC_W(43) PC_W(44) PC_W(45)
            PC_W(46) PC_W(47) PC_W(48)
        }
#undef PC_W
    };

    if (wave < 4) {
        // =============================== C: consumers ===============================
        const int l31 = lane & 31, kh2 = lane >> 5;
        const int wm = wave & 1, wn = wave >> 1;
        const int sr_ = c3_strip_row(l31 >> 2);
        const char* const aL = smem + kh2 * PC_PS + (wm * 100 + sr_ * 10 + (l31 & 3)) * 16;
        const char* const bL = smem + PC_OFF_B + wn * 4096 + lane * 16;
        int eoff[4];                                     // hand-over tile: byte offset of (strip row of register quad rq, x = 0, this lane's channel)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) eoff[rq] = PC_OFF_EPI + ((wm * 64 + c3_strip_row(2 * rq + kh2) * 8) * 64 + wn * 32 + l31) * 4;
        // two fragment sets ([fragment][piece] + [piece] of one 16-channel step): a half tap multiplies one set while the fragments of the next
        // half tap are requested into the other (six MFMAs ~ 200 cycles ahead; three sets / twelve MFMAs measured the same)
        uint4 FA0[2][2], FB0[2], FA1[2][2], FB1[2];
        f32x16 acc[2];
#ifdef BH_TUNING
        const int dbg = a.dbg_noload;
#else
        constexpr int dbg = 0;
#endif
        // fragments of one half tap: ap = this lane's halo position of the tap, bp = this lane's slot of the tap's weight image
#define PC_LD_A(A_, ap, s2, i_, pc_) A_[i_][pc_] = *reinterpret_cast<const uint4*>((ap) + (pc_) * PC_PIECE + 2 * (s2) * PC_PS + (i_) * 64)
#define PC_LD_B(B_, bp, s2, pc_) B_[pc_] = *reinterpret_cast<const uint4*>((bp) + ((pc_) * 2 + (s2)) * 1024)
#define PC_LOADH(A_, B_, ap, bp, s2)                                                                                    \
    do {                                                                                                                \
        PC_LD_A(A_, ap, s2, 0, 1); PC_LD_B(B_, bp, s2, 0); PC_LD_A(A_, ap, s2, 1, 1);                                   \
        PC_LD_A(A_, ap, s2, 0, 0); PC_LD_B(B_, bp, s2, 1); PC_LD_A(A_, ap, s2, 1, 0);                                   \
    } while (0)
        // One half tap: six products - lo*hi, hi*lo, hi*hi (small first) for fragment 0 and 1, the order of conv3x3_halo_kernel's X3_MFMA per
        // accumulator - ALTERNATING between the two accumulators, each followed by one fragment request of the next half tap (in the order
        // that half will use them).  sched_barrier(0) pins this order: left alone the compiler issues the three products of an accumulator
        // back to back, and a v_mfma_f32_32x32x16_f16 that depends on the one in front of it waits ~20 cycles beyond the 32 of the
        // pipe (tools/pc_timeline.py: 2.0 kilo-cycles per kernel row of 36 MFMAs where 1.15 are the pipe's).
#define PC_STEP(ACC, AV, BV, LD)                                                                                        \
    do {                                                                                                                \
        ACC = c3_mfma16<true>(AV, BV, ACC);                                                                             \
        LD;                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define PC_HALF(UA, UB, LA, LB, ap, bp, s2)                                                                             \
    do {                                                                                                                \
        PC_STEP(acc[0], UA[0][1], UB[0], PC_LD_A(LA, ap, s2, 0, 1));                                                    \
        PC_STEP(acc[1], UA[1][1], UB[0], PC_LD_B(LB, bp, s2, 0));                                                       \
        PC_STEP(acc[0], UA[0][0], UB[1], PC_LD_A(LA, ap, s2, 1, 1));                                                    \
        PC_STEP(acc[1], UA[1][0], UB[1], PC_LD_A(LA, ap, s2, 0, 0));                                                    \
        PC_STEP(acc[0], UA[0][0], UB[0], PC_LD_B(LB, bp, s2, 1));                                                       \
        PC_STEP(acc[1], UA[1][0], UB[0], PC_LD_A(LA, ap, s2, 1, 0));                                                    \
    } while (0)
        // the half tap in front of a barrier: its six requests (the row's LAST fragments) go out first, so that they have returned when the
        // six products are through and the lgkmcnt(0) in front of the barrier does not drain the pipe
#define PC_HALF_PRE(UA, UB, LA, LB, ap, bp, s2)                                                                         \
    do {                                                                                                                \
        PC_LOADH(LA, LB, ap, bp, s2);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        PC_STEP(acc[0], UA[0][1], UB[0], (void)0);                                                                      \
        PC_STEP(acc[1], UA[1][1], UB[0], (void)0);                                                                      \
        PC_STEP(acc[0], UA[0][0], UB[1], (void)0);                                                                      \
        PC_STEP(acc[1], UA[1][0], UB[1], (void)0);                                                                      \
        PC_STEP(acc[0], UA[0][0], UB[0], (void)0);                                                                      \
        PC_STEP(acc[1], UA[1][0], UB[0], (void)0);                                                                      \
    } while (0)
        PC_BARRIER();                                    // B(-1): halo stage 0 and weight row 0 are in LDS
        PC_LOADH(FA0, FB0, aL, bL, 0);
#if PC_EMU
        // (timing experiment, wrong results: what the row time would be with fewer fragment reads per MFMA.  Both fragment sets are loaded
        //  once for real, then 1: the B reads are dropped - 4 reads per 6 MFMAs, the ratio of a 2 x 2 accumulator block; 2: every read is
        //  dropped - the matrix pipe with the other roles around it; 3: the A reads of fragment 1 are dropped as well - 2 per 6)
        PC_LOADH(FA1, FB1, aL, bL, 1);
#undef PC_LD_B
#define PC_LD_B(B_, bp, s2, pc_) (void)0
#if PC_EMU >= 2
#undef PC_LD_A
#if PC_EMU == 2
#define PC_LD_A(A_, ap, s2, i_, pc_) (void)0
#else
#define PC_LD_A(A_, ap, s2, i_, pc_) do { if ((i_) == 0) A_[i_][pc_] = *reinterpret_cast<const uint4*>((ap) + (pc_) * PC_PIECE + 2 * (s2) * PC_PS + (i_) * 64); } while (0)
#endif
#endif
#endif
        int kap = 0;                                     // chunk counter: halo stage = kap & 1; the weight slot of kernel row r is slot r
        for (int ti = 0; ti < Tw; ++ti) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            for (int c = 0; c < nch; ++c, ++kap) {
                const int par = kap & 1;
                const bool more = kap + 1 < K;
#pragma unroll
                for (int row = 0; row < 3; ++row) {
                    const char* const rp = aL + par * PC_STAGE + row * 160;          // tap (row, 0) of this chunk's halo stage
                    const char* const sp = bL + row * PC_BROW;
                    // tap 0 of the next kernel row (next chunk: the other stage; after the last chunk the request is repeated on this row -
                    // harmless, and the loop body stays branch-free)
                    const char* const nrp = row < 2 ? rp + 160 : (more ? aL + (par ^ 1) * PC_STAGE : rp);
                    const char* const nsp = row < 2 ? sp + PC_BROW : bL;
                    if (dbg & 1) { PC_BARRIER(); continue; }       // (ablation: barriers only)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        // half 0: step 0 multiplies, step 1 of this tap is requested
                        if (t < 2) PC_HALF(FA0, FB0, FA1, FB1, rp + t * 16, sp + t * PC_BTAP, 1);
                        else {
                            PC_HALF_PRE(FA0, FB0, FA1, FB1, rp + t * 16, sp + t * PC_BTAP, 1);
                            PC_BARRIER();                // B(rho): the row's last fragments are in registers - its weight slot (row 2: and the halo stage) are free
                        }
                        // half 1: step 1 multiplies, step 0 of the next tap is requested
                        if (t < 2) PC_HALF(FA1, FB1, FA0, FB0, rp + (t + 1) * 16, sp + (t + 1) * PC_BTAP, 0);
                        else PC_HALF(FA1, FB1, FA0, FB0, nrp, nsp, 0);
                    }
                }
            }
            // hand the accumulators over: element (i, r) = pixel (strip row of (r >> 2, kh2), x = 4 i + (r & 3)), channel wn * 32 + l31.
            // The last products must have left the matrix pipe before their registers are read: the compiler's hazard padding did not
            // survive the loop structure in every instantiation (dgrad + BatchNorm sums: sporadic stale quads in the first row of a tile),
            // so the wait states are spelled out - 32 of them per tile, tied to both accumulators.
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(smem + eoff[r >> 2] + (4 * i + (r & 3)) * 256) = acc[i][r];
        }
        PC_BARRIER();                                    // F1: the last tile is in the hand-over tile
        PC_BARRIER();                                    // F2
        PC_BARRIER();                                    // F3
        PC_BARRIER();                                    // F4
#undef PC_HALF_PRE
#undef PC_HALF
#undef PC_STEP
#undef PC_LOADH
#undef PC_LD_A
#undef PC_LD_B
        return;
    }

    if (wave < 7) {
        // =============================== H: halo staging (3 waves) ===============================
        // Slot round j (0 .. 4) of a chunk: halo pixel hp = j * 48 + (ht >> 2) of the [2 sub-tiles][10][10] image, this thread's 8-channel
        // plane (ht & 3).  Chunk k lives in register set k & 1 (= its halo stage): round j of chunk k + 1 is cut and written in the row
        // phases listed below and the same registers immediately request round j of chunk k + 3.
        //
        // The loads are inline assembly with hand-counted waits: left to the compiler, a load whose use lies two chunk loops ahead is waited
        // for with vmcnt(0) - i.e. together with everything requested since, one HBM round trip per row phase (tools/pc_timeline.py: the H
        // waves were the last at 22 of 29 barriers).  nis counts this wave's loads; seq[j] = nis right after round j's requests, so
        // "nis - seq[j]" younger loads may stay in flight when round j is needed.  Between a request and its wait statement - which names
        // the registers - the values are only carried, never read.  So that the compiler has no reason to COPY a register whose load is
        // still in flight, every request and every wait is unconditional straight-line code (a round that has nothing to load asks for an
        // out-of-range offset: zeros, at once); only the cut arithmetic and the LDS stores sit under conditions, and the chunk loop is
        // unrolled by two so that the register set is a compile-time choice.
        // (the producers' instruction streams are short and must not wait for issue slots behind the consumer wave of their SIMD, which is
        //  older and would win every arbitration: tools/pc_timeline.py showed ~20 cycles per H / E instruction at equal priority)
        __builtin_amdgcn_s_setprio(2);
        const int ht = tid - 256;
        const int plane = ht & 3;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Src), 0, a.src_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNI ? a.bni : a.Src), 0,
                                                                              BNI ? (unsigned)(a.bni_groups * a.Kc * 8) : 0u, 0x00020000);
        constexpr int NR = 5;
        const int lo0 = plane * PC_PS + (ht >> 2) * 16;   // LDS byte offset of round 0's slot inside a piece image (round j: + j * 48 * 16)
        unsigned xoff[NR];                                // global byte offset of the slot's 32 bytes in chunk 0 of the tile being loaded (PC_XOOB: padding)
        auto h_tile = [&](int wt) {
            const int bx = wt % a.gx_total;
            int org[2], oy0[2], ox0[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int g = bx * 2 + s_;
                int img, ty, tx;
                pc_decode(a, g < a.subtiles ? g : 0, img, ty, tx);
                oy0[s_] = ty * 8 - 1; ox0[s_] = tx * 8 - 1;
                org[s_] = g < a.subtiles ? (img * a.H + oy0[s_]) * a.W + ox0[s_] : (int)0x80000000;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int hp = j * 48 + (ht >> 2);
                const int s_ = hp >= 100 ? 1 : 0, p_ = hp - 100 * s_;
                const int hy = (p_ * 205) >> 11, hx = p_ - hy * 10;
                const int y = (s_ ? oy0[1] : oy0[0]) + hy, x = (s_ ? ox0[1] : ox0[0]) + hx;
                const int o = s_ ? org[1] : org[0];
                unsigned off = PC_XOOB;
                if (hp < PC_HPL && o != (int)0x80000000 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                    off = ((unsigned)(o + hy * a.W + hx) * (unsigned)a.Kc + (unsigned)(plane * 8)) * 4u;
                xoff[j] = off;
            }
        };
        typedef float pc_f4 __attribute__((ext_vector_type(4)));
        pc_f4 hA[NR][2], hB[NR][2];                       // the rounds in flight: even chunks / odd chunks
        pc_f4 tb[4];                                      // BNI: (scale, shift) of this plane's 8 channels of the chunk being cut (ONE set: it is
                                                          // requested when the previous chunk's last round has been cut, a row phase ahead)
        unsigned okA = 0, okB = 0;                        // bit j: slot round j of that chunk lies inside the image
        int nis = 0, seq_tb = 0, seqA[NR] = {0, 0, 0, 0, 0}, seqB[NR] = {0, 0, 0, 0, 0};
        int li_t = 0, li_c = 0;                           // (tile, chunk of the tile) whose rounds are being requested
        // the coefficients of chunk number k (its tile's statistics group, its 32 channels, this thread's plane); k >= K: nothing (zeros)
        auto tb_issue = [&](int k) {
            const int ti_ = k / nch, c_ = k - ti_ * nch;
            const int bx = (wt0 + (k < K ? ti_ : 0)) % a.gx_total;
            int img, ty, tx;
            pc_decode(a, bx * 2 < a.subtiles ? bx * 2 : 0, img, ty, tx);
            const unsigned vo = k < K ? (unsigned)((img / a.bni_ipg) * a.Kc + plane * 8) * 8u : PC_XOOB, so = (unsigned)(c_ * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                tb[q] = __builtin_bit_cast(pc_f4, __builtin_amdgcn_raw_buffer_load_b128(rsT, vo + (unsigned)q * 16u, so, 0));
            nis += 4; seq_tb = nis;
        };
        // round j of the next chunk in sequence (en: there is one - else an out-of-range request) -> register set h
        auto h_issue = [&](auto J, pc_f4 (&h)[NR][2], unsigned& okm, int (&seq)[NR], bool en) {
            constexpr int j = decltype(J)::value;
            if (j == 0 && en && li_c == 0) h_tile(wt0 + li_t);
            const unsigned vo = en ? xoff[j] : PC_XOOB, so = en ? (unsigned)(li_c * 128) : 0u;
            h[j][0] = __builtin_bit_cast(pc_f4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so, 0));
            h[j][1] = __builtin_bit_cast(pc_f4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo + 16u, so, 0));
            nis += 2; seq[j] = nis;
            okm = (okm & ~(1u << j)) | ((vo != PC_XOOB ? 1u : 0u) << j);
            if (j == NR - 1 && en) { if (++li_c == nch) { li_c = 0; ++li_t; } }
        };
        float f16_s = 1.0f;
        const float bni_lo = a.bni_relu ? 0.0f : -__builtin_inff();
        // round j of register set h -> fp16 pieces in halo stage `stage` (en: the chunk exists)
        auto h_cut = [&](auto J, int stage, pc_f4 (&h)[NR][2], unsigned okm, const int (&seq)[NR], bool en) {
            constexpr int j = decltype(J)::value;
            if constexpr (BNI) {
                if (j == 0 && PC_H_ASM) {                  // this chunk's coefficients (younger than its rounds: everything older has then arrived too)
                    wait_vm(nis - seq_tb);
                    asm volatile("" : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]));
                }
            }
            if (PC_H_ASM) {
                wait_vm(nis - seq[j]);                     // round j has arrived
                asm volatile("" : "+v"(h[j][0]), "+v"(h[j][1]));
            }
            if (en && j * 48 + (ht >> 2) < PC_HPL) {
                char* const st = smem + stage * PC_STAGE;
                float4 u = make_float4(h[j][0][0], h[j][0][1], h[j][0][2], h[j][0][3]), v = make_float4(h[j][1][0], h[j][1][1], h[j][1][2], h[j][1][3]);
                if constexpr (BNI) {
                    // y = max(x * scale + shift, lo) on the slot's 8 channels; padding stays zero
                    const bool ok = (okm >> j) & 1u;
                    u.x = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.x, tb[0][0], tb[0][1]), bni_lo) : 0.f;
                    u.y = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.y, tb[0][2], tb[0][3]), bni_lo) : 0.f;
                    u.z = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.z, tb[1][0], tb[1][1]), bni_lo) : 0.f;
                    u.w = ok ? __builtin_elementwise_maximum(__builtin_fmaf(u.w, tb[1][2], tb[1][3]), bni_lo) : 0.f;
                    v.x = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.x, tb[2][0], tb[2][1]), bni_lo) : 0.f;
                    v.y = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.y, tb[2][2], tb[2][3]), bni_lo) : 0.f;
                    v.z = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.z, tb[3][0], tb[3][1]), bni_lo) : 0.f;
                    v.w = ok ? __builtin_elementwise_maximum(__builtin_fmaf(v.w, tb[3][2], tb[3][3]), bni_lo) : 0.f;
                }
                uint4 p0, p1;
                bh_split8_f16(u, v, f16_s, p0, p1);
                *reinterpret_cast<uint4*>(st + lo0 + j * 768) = p0;
                *reinterpret_cast<uint4*>(st + PC_PIECE + lo0 + j * 768) = p1;
            }
        };
#ifdef BH_TUNING
        const bool hon = !(a.dbg_noload & 4);             // ablation: no halo staging at all (timing only)
#else
        constexpr bool hon = true;
#endif
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
        using J3 = std::integral_constant<int, 3>; using J4 = std::integral_constant<int, 4>;
        // the three row phases of chunk number kap_: chunk kap_ + 1 (register set / halo stage SET) -> LDS in three shares, chunk kap_ + 3
        // into flight behind each share; BatchNorm-on-load: chunk kap_ + 2's coefficients once chunk kap_ + 1 is through
#define PC_H_CHUNK(kap_, h, okm, seq, SET, AHEAD)                                                                       \
    do {                                                                                                                \
        const bool cut_ = (kap_) + 1 < K && hon, iss_ = (kap_) + (AHEAD) < K && hon;                                    \
        h_cut(J0{}, SET, h, okm, seq, cut_); h_issue(J0{}, h, okm, seq, iss_);                                          \
        h_cut(J1{}, SET, h, okm, seq, cut_); h_issue(J1{}, h, okm, seq, iss_);                                          \
        PC_BARRIER();                                 /* row 0 */                                                       \
        h_cut(J2{}, SET, h, okm, seq, cut_); h_issue(J2{}, h, okm, seq, iss_);                                          \
        h_cut(J3{}, SET, h, okm, seq, cut_); h_issue(J3{}, h, okm, seq, iss_);                                          \
        PC_BARRIER();                                 /* row 1 */                                                       \
        h_cut(J4{}, SET, h, okm, seq, cut_); h_issue(J4{}, h, okm, seq, iss_);                                          \
        if constexpr (BNI) tb_issue(hon ? (kap_) + 2 : K);                                                              \
        PC_BARRIER();                                 /* row 2 */                                                       \
    } while (0)
#define PC_H_ISSUE_ALL(h, okm, seq, en)                                                                                 \
    do { h_issue(J0{}, h, okm, seq, en); h_issue(J1{}, h, okm, seq, en); h_issue(J2{}, h, okm, seq, en);                \
         h_issue(J3{}, h, okm, seq, en); h_issue(J4{}, h, okm, seq, en); } while (0)
        if constexpr (!BNI) {
            // chunks 0 and 1 requested at once (then the source tensor's scale, whose loads the compiler waits for - and with them, being
            // younger, for both chunks); chunk 0 cut; chunk 2 requested
            const unsigned rec_ = lane < BH_AMAX_SLOTS ? a.amax_src[lane * BH_AMAX_STRIDE] : 0u;      // (the source tensor's magnitude record: in flight with the halos)
            PC_H_ISSUE_ALL(hA, okA, seqA, hon);
            PC_H_ISSUE_ALL(hB, okB, seqB, 1 < K && hon);
            f16_s = __builtin_bit_cast(float, (unsigned)(127 + bh_f16_scale_exp(pc_amax_reduce(rec_))) << 23);
            h_cut(J0{}, 0, hA, okA, seqA, hon); h_cut(J1{}, 0, hA, okA, seqA, hon); h_cut(J2{}, 0, hA, okA, seqA, hon);
            h_cut(J3{}, 0, hA, okA, seqA, hon); h_cut(J4{}, 0, hA, okA, seqA, hon);
            PC_H_ISSUE_ALL(hA, okA, seqA, 2 < K && hon);
            PC_BARRIER();                                 // B(-1)
            for (int kap = 0; kap < K; kap += 2) {
                PC_H_CHUNK(kap, hB, okB, seqB, 1, 3);     // chunk kap + 1 (odd) lives in set B / stage 1
                if (kap + 1 < K) PC_H_CHUNK(kap + 1, hA, okA, seqA, 0, 3);
            }
        } else {
            // BatchNorm-on-load: ONE register set (the coefficient registers and the transform's temporaries take the room of the second):
            // a round of chunk kap + 1 is cut and the same registers request that round of chunk kap + 2 - one chunk time in flight
            (void)hB; (void)okB; (void)seqB;
            const unsigned rec_ = lane < BH_AMAX_SLOTS ? a.amax_src[lane * BH_AMAX_STRIDE] : 0u;
            tb_issue(hon ? 0 : K);
            PC_H_ISSUE_ALL(hA, okA, seqA, hon);
            f16_s = __builtin_bit_cast(float, (unsigned)(127 + bh_f16_scale_exp(pc_amax_reduce(rec_))) << 23);
            h_cut(J0{}, 0, hA, okA, seqA, hon); h_cut(J1{}, 0, hA, okA, seqA, hon); h_cut(J2{}, 0, hA, okA, seqA, hon);
            h_cut(J3{}, 0, hA, okA, seqA, hon); h_cut(J4{}, 0, hA, okA, seqA, hon);
            tb_issue(hon ? 1 : K);
            PC_H_ISSUE_ALL(hA, okA, seqA, 1 < K && hon);
            PC_BARRIER();                                 // B(-1)
            for (int kap = 0; kap < K; ++kap) PC_H_CHUNK(kap, hA, okA, seqA, (kap + 1) & 1, 2);
        }
        PC_BARRIER();                                     // F1
        PC_BARRIER();                                     // F2
        PC_BARRIER();                                     // F3
        PC_BARRIER();                                     // F4
#undef PC_H_ISSUE_ALL
#undef PC_H_CHUNK
        return;
    }

    if (wave < 8) {
        // =============================== D: weight rows by LDS-DMA (1 wave) ===============================
        // Kernel row r of a chunk lives in ring slot r: 3 taps x [2 n tiles][2 pieces][2 steps] x 1 KB = 24 pieces.  Row rho + 2 is requested
        // in phase rho (its slot was left at B(rho - 1)) and must have landed at B(rho + 1): nothing else in this wave touches the
        // vector-memory counter, so "all but this phase's 24" is exact.
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wt), 0, a.w_bytes, 0x00020000);
#ifdef BH_TUNING
        const bool doff = a.dbg_noload & 8;               // ablation: no weights (timing only)
#else
        constexpr bool doff = false;
#endif
        auto dma_num = [&](int r) {                       // weight row number r = (tile, chunk, kernel row)
            const int kp = r / 3, row = r - kp * 3, ti_ = kp / nch, c_ = kp - ti_ * nch;
            const int ny = (wt0 + ti_) / a.gx_total;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int piece = 0; piece < 8; ++piece) {
                    const unsigned so = (unsigned)(((c_ * 9 + row * 3 + t) * a.NW + ny * 2) * 4096 + piece * 1024);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(smem + PC_OFF_B + row * PC_BROW + t * PC_BTAP + piece * 1024), 16,
                                                             (unsigned)lane * 16u, so, 0, 0);
                }
        };
        int pro = 0;
        if (!doff) { dma_num(0); if (R > 1) { dma_num(1); ++pro; } if (R > 2) { dma_num(2); ++pro; } }
        wait_vm(24 * pro);                                // row 0 has landed
        PC_BARRIER();                                     // B(-1)
        for (int rho = 0; rho < R; ++rho) {
            int nd = 0;
            if (rho >= 1 && rho + 2 < R && !doff) { dma_num(rho + 2); nd = 24; }
            wait_vm(rho == 0 ? ((R > 2 && !doff) ? 24 : 0) : nd);       // row rho + 1 has landed
            PC_BARRIER();                                 // B(rho)
        }
        PC_BARRIER();                                     // F1
        PC_BARRIER();                                     // F2
        PC_BARRIER();                                     // F3
        PC_BARRIER();                                     // F4
        return;
    }

    // =============================== E: epilogue (4 waves) ===============================
    {
        __builtin_amdgcn_s_setprio(1);
        const int ew = wave - 8;
        const int cq = lane & 15, pl = lane >> 4;         // channel quad (channels 4 cq .. 4 cq + 3 of the tile), pixel of the unit
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.Out, 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.Out), 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bnr_z ? a.bnr_z : a.Out), 0, a.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bnr_y ? a.bnr_y : a.Out), 0, a.out_bytes, 0x00020000);
        constexpr bool stats = EM & 1, use_ext = EM & 2, rd_z = EM & 4, rd_y = EM & 8;
        const bool rd_res = a.res != nullptr;             // (use_ext: the residual when there is one, else the old gradient)
        const unsigned nn4 = (unsigned)a.Nn * 4u;
#ifdef BH_TUNING
        const bool eoff_ = a.dbg_noload & 16;             // ablation: no epilogue (timing only)
#else
        constexpr bool eoff_ = false;
#endif

        // ---- state of the tile whose epilogue RUNS, and of the tile whose operands are being FETCHED (the same or the next) ----
        unsigned tbase[2] = {0u, 0u}, fbase[2] = {0u, 0u};   // byte offset of (sub-tile origin, channel n0) in the output tensor
        bool tvalid[2] = {false, false}, fvalid[2] = {false, false};
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r_invstd = bv, r_cx0 = bv, r_sc = bv, r_sh = bv;      // BatchNorm-reduce coefficients of the lane's four channels
        int cur_grp = -1, cur_n0 = 0;
        double S1[4] = {0.0, 0.0, 0.0, 0.0}, S2[4] = {0.0, 0.0, 0.0, 0.0};
        float amx = 0.0f;                                 // (a.amax_out, BatchNorm-backward forms: max |mask(d)| of what this wave wrote)
        bool have = false;                                // S1 / S2 hold something
        // Statistics leave the workgroup with ONE f64 atomic per (channel, moment): stash() puts this wave's sums (reduced over its four pixel
        // lanes) into LDS, commit() - behind the next barrier - lets E wave 0 add the four waves' sums and issue the atomics.  Happens when the
        // workgroup's tile range crosses a statistics group or channel tile, and at the end.
        int pend_grp = 0, pend_n0 = 0;
        bool pending = false;
        auto stash = [&]() {
            double* const red = reinterpret_cast<double*>(smem + PC_OFF_RED);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double s1 = S1[e], s2 = S2[e];
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                if (pl == 0) { red[(ew * 64 + cq * 4 + e) * 2] = s1; red[(ew * 64 + cq * 4 + e) * 2 + 1] = s2; }
                S1[e] = 0.0; S2[e] = 0.0;
            }
            pend_grp = cur_grp; pend_n0 = cur_n0; pending = true; have = false;
        };
        auto commit = [&]() {
            if (!pending) return;
            if (ew == 0) {
                const double* const red = reinterpret_cast<const double*>(smem + PC_OFF_RED);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = q * 64 + lane;            // (channel, moment)
                    const double tot = ((red[idx] + red[128 + idx]) + red[256 + idx]) + red[384 + idx];
                    bh_acc_add(&a.bn_sums[bn_sum_index(0, a.groups, pend_grp, a.Nn, pend_n0 + (idx >> 1), idx & 1)], tot, a.det);
                }
            }
            pending = false;
        };
        auto tile_geom = [&](int wt, unsigned (&base)[2], bool (&valid)[2], int& grp, int& n0) {
            const int ny_ = wt / a.gx_total, bx = wt - ny_ * a.gx_total;
            n0 = ny_ * 64; grp = 0;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int g = bx * 2 + s_;
                int img, ty, tx;
                pc_decode(a, g < a.subtiles ? g : 0, img, ty, tx);
                valid[s_] = g < a.subtiles;
                base[s_] = ((unsigned)((img * a.H + ty * 8) * a.W + tx * 8) * (unsigned)a.Nn + (unsigned)n0) * 4u;
                if (s_ == 0) grp = img / a.imgs_per_group;
            }
        };
        auto f_tile = [&](int wt) { int g_, n_; tile_geom(wt, fbase, fvalid, g_, n_); };
        auto e_tile = [&](int wt) {
            int grp, n0;
            tile_geom(wt, tbase, tvalid, grp, n0);
            if (stats && have && (grp != cur_grp || n0 != cur_n0)) stash();
            if (grp != cur_grp || n0 != cur_n0) {
                cur_grp = grp; cur_n0 = n0;
                const int n = n0 + cq * 4;
                bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (rd_z) {
                    float isd[4], cx[4], sc[4], sh[4];
                    const double rows = (double)a.bnr_rows;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double mu = bn_sum_total(a.bnr_stats, a.groups, grp, a.Nn, n + e, 0, a.det) / rows;
                        double var = bn_sum_total(a.bnr_stats, a.groups, grp, a.Nn, n + e, 1, a.det) / rows - mu * mu;
                        if (var < 0) var = 0;
                        const float mean = (float)mu;
                        isd[e] = 1.0f / sqrtf((float)var + a.bnr_eps);
                        sc[e] = (a.bnr_gamma ? a.bnr_gamma[n + e] : 1.f) * isd[e];
                        sh[e] = (a.bnr_beta ? a.bnr_beta[n + e] : 0.f) - mean * sc[e];
                        cx[e] = -mean * isd[e];
                    }
                    r_invstd = make_float4(isd[0], isd[1], isd[2], isd[3]); r_cx0 = make_float4(cx[0], cx[1], cx[2], cx[3]);
                    r_sc = make_float4(sc[0], sc[1], sc[2], sc[3]); r_sh = make_float4(sh[0], sh[1], sh[2], sh[3]);
                }
            }
        };
        // Epilogue unit idx (0 .. 7) of this wave = unit u = ew + 4 idx of the tile: four pixels x 64 channels = 1 KB of the hand-over tile and
        // of the output row.  What a unit reads from HBM (old gradient, residual, BatchNorm input / output) is requested FOUR units ahead
        // into register set idx & 3 (epi_fetch), two or more barrier phases before epi_run needs it; the stores are never waited for.
        // Addressing: unit u = ew + 4 idx covers pixels (sub-tile idx >> 2, row (ew >> 1) + 2 (idx & 3), x = 4 (ew & 1) + pl): the lane part of the
        // byte offset is ONE tile-independent VGPR, the rest (sub-tile origin + row step) rides in the scalar offset of the buffer instruction.
        float4 pf_a[4], pf_z[4], pf_y[4];                  // pf_a: the old gradient (dgrad joins) OR the residual (inference forward) - never both
        constexpr int nld = (use_ext ? 1 : 0) + (rd_z ? 1 : 0) + (rd_y ? 1 : 0);      // loads per unit
        const unsigned voff = (unsigned)((ew >> 1) * a.W + ((ew & 1) << 2) + pl) * nn4 + (unsigned)cq * 16u;
        const unsigned rowstep2 = 2u * (unsigned)a.W * nn4;
        const char* const epi_l = smem + PC_OFF_EPI + ew * 1024 + lane * 16;
        const __amdgpu_buffer_rsrc_t rsA_ = rd_res ? rsR : rsO;
        auto epi_fetch = [&](auto Q, int idx) {            // unit idx of the FETCH tile -> register set q
            constexpr int q = decltype(Q)::value;
            if (nld == 0 || !((idx >> 2) ? fvalid[1] : fvalid[0])) return;
            const unsigned so = ((idx >> 2) ? fbase[1] : fbase[0]) + (unsigned)(idx & 3) * rowstep2;
            if constexpr (use_ext) pf_a[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA_, voff, so, 0));
            if constexpr (rd_z) pf_z[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, voff, so, 0));
            if constexpr (rd_y) pf_y[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voff, so, 0));
        };
        int kout = 0;
        auto epi_run = [&](auto Q, int idx) {              // unit idx of the RUN tile, operands in register set q
            constexpr int q = decltype(Q)::value;
            if (!((idx >> 2) ? tvalid[1] : tvalid[0])) return;
            const unsigned so = ((idx >> 2) ? tbase[1] : tbase[0]) + (unsigned)(idx & 3) * rowstep2;
            const float4 av = *reinterpret_cast<const float4*>(epi_l + idx * 4096);
            const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 o = use_ext ? pf_a[q] : z0, zz = rd_z ? pf_z[q] : z0, yy = rd_y ? pf_y[q] : z0;
            const float accv[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
            const float oo[4] = {o.x, o.y, o.z, o.w};
            const float z4[4] = {zz.x, zz.y, zz.z, zz.w}, y4[4] = {yy.x, yy.y, yy.z, yy.w};
            const float isd[4] = {r_invstd.x, r_invstd.y, r_invstd.z, r_invstd.w}, cx[4] = {r_cx0.x, r_cx0.y, r_cx0.z, r_cx0.w};
            const float sc[4] = {r_sc.x, r_sc.y, r_sc.z, r_sc.w}, sh[4] = {r_sh.x, r_sh.y, r_sh.z, r_sh.w};
            float out[4];
            const bool do_relu = a.relu != 0, mask_on = rd_z && a.bnr_relu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = __builtin_ldexpf(accv[e], kout) + bb[e];
                if constexpr (use_ext) v += oo[e];         // (0 + old, or 0 + residual: the halo kernel's `ext`)
                v = do_relu ? fmaxf(v, 0.0f) : v;
                out[e] = v;
                if constexpr (stats) {
                    // forward / column sums: (v, v^2); BatchNorm backward: (g, g xhat) with g = v under the ReLU mask of that BatchNorm
                    float vm = v, t = v;
                    if constexpr (rd_z) {
                        const float yv = rd_y ? y4[e] : __builtin_fmaf(z4[e], sc[e], sh[e]);
                        vm = (mask_on && !(yv > 0.f)) ? 0.f : v;
                        t = __builtin_fmaf(z4[e], isd[e], cx[e]);
                        amx = fmaxf(amx, fabsf(vm));
                    }
                    S1[e] += (double)vm;
                    S2[e] = __builtin_fma((double)vm, (double)t, S2[e]);
                }
            }
            // gfx950 hazard, found the hard way (round 5): a 128-bit buffer store whose soffset is an SGPR reads its data registers over several
            // cycles AFTER it issues - a VALU instruction that overwrites the first data register right behind the store (the compiler reused it
            // as the next unit's ReLU temporary) reached some lanes first: sporadic wrong first channels in lanes 12..15 of each 16.  LLVM's hazard
            // table holds this case for safe ("no hazard when soffset is a register") and pads nothing, so the wait states are spelled out, tied
            // to the data registers (which therefore stay live until they have passed).
            const __attribute__((ext_vector_type(4))) unsigned ov =
                __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(out[0], out[1], out[2], out[3]));
            __builtin_amdgcn_raw_buffer_store_b128(ov, rsO, voff, so, 0);
            asm volatile("s_nop 3" :: "v"(ov));
            have = true;
        };
        // ---- schedule: two wave GROUPS take turns.  Group g = ew >> 1 is active in the phases k = 1 .. nph - 1 of a tile with k odd (g = 0) /
        // even (g = 1); in an active phase a wave runs one BATCH of <= 4 of its eight units (register set = position in the batch) and then
        // requests the operands of its next batch - which it will need two barrier phases (~3.5 kilo-cycles) later.  The compiler waits
        // for a load whose use lies in a later loop iteration with vmcnt(0): with this turn-taking the only loads a wave has in flight at
        // that wait ARE the batch it needs (requested two phases ago), so the conservative wait costs nothing - with all four waves working
        // every phase it cost one HBM round trip per phase (tools/pc_timeline.py: the E waves last at 16 of 29 barriers).
        using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>; using Q2 = std::integral_constant<int, 2>;
        using Q3 = std::integral_constant<int, 3>;
        const int nph = 3 * nch;                          // phases per tile
        const int grp = ew >> 1;
        const int nA = grp ? (nph - 1) / 2 : nph / 2;     // active phases per tile (>= 1)
        constexpr int UB = 4, NB = 2;                     // units per batch, batches per tile
        const int BPP = (NB + nA - 1) / nA;               // batches per active phase (2 only for one-chunk tiles)
        auto run_batch = [&](int b) {
            const int i0 = b * UB;
            epi_run(Q0{}, i0 + 0); epi_run(Q1{}, i0 + 1); epi_run(Q2{}, i0 + 2); epi_run(Q3{}, i0 + 3);
        };
        auto fetch_batch = [&](int b) {
            const int i0 = b * UB;
            epi_fetch(Q0{}, i0 + 0); epi_fetch(Q1{}, i0 + 1); epi_fetch(Q2{}, i0 + 2); epi_fetch(Q3{}, i0 + 3);
        };
        // batch b of the run tile, then the request of the batch after it: b + 1 of the same tile, or batch 0 of tile `nxt` (-1: none)
        // (Round 6 tried refilling a unit's registers right after that unit has been consumed, so that the requests leave a batch's run
        //  time earlier: the compiler then waits vmcnt(0) in front of EVERY unit - a register loaded in an earlier loop iteration counts as
        //  pending whatever has been waited for since - i.e. for the refill just issued: dgrad + BatchNorm sums + accumulate 57.4 us
        //  against 48.8 on the 32 x 32 shape, barrier time line 123 against 103 kilo-cycles.  The whole-batch order stays.)
        auto step_batch = [&](int b, int nxt) {
            run_batch(b);
            if (nld == 0) return;
            if (b + 1 < NB) fetch_batch(b + 1);
            else if (nxt >= 0) { f_tile(nxt); fetch_batch(0); }
        };

        if (!eoff_) {
            // the scales of the fp16 pieces (2^ka source, 2^kw weights), then batch 0 of the first tile goes into flight
            const unsigned* const wrec = reinterpret_cast<const unsigned*>(a.Wt) + (a.w_bytes >> 2);
            const unsigned wv = lane < 16 ? wrec[lane] : 0u;
            const unsigned rec_ = lane < BH_AMAX_SLOTS ? a.amax_src[lane * BH_AMAX_STRIDE] : 0u;
            kout = -(bh_f16_scale_exp(pc_amax_reduce(wv)) + bh_f16_scale_exp(pc_amax_reduce(rec_)));
            asm volatile("" :: "s"(kout) : "memory");       // (the requests below stay behind the scale loads)
            f_tile(wt0);
            fetch_batch(0);
        }
        PC_BARRIER();                                     // B(-1)
        for (int ti = 0; ti < Tw; ++ti) {
            for (int k = 0; k < nph; ++k) {               // phase k of this tile: ends at B(rho)
                // this phase's share of the PREVIOUS tile's epilogue (the hand-over tile is readable from phase 1 on)
                if (ti > 0 && !eoff_) {
                    if (k == 0) e_tile(wt0 + ti - 1);
                    else {
                        if (k == 1) commit();
                        if ((k & 1) != grp) {             // this group's turn: active phase number a_ of the tile
                            const int a_ = grp ? (k >> 1) - 1 : (k >> 1);
                            for (int bb = 0; bb < BPP; ++bb) { const int b = a_ * BPP + bb; if (b < NB) step_batch(b, wt0 + ti); }
                        }
                    }
                }
                PC_BARRIER();                             // B(rho)
            }
        }
        PC_BARRIER();                                     // F1: the last tile's accumulators are in the hand-over tile
        if (!eoff_) e_tile(wt0 + Tw - 1);
        PC_BARRIER();                                     // F2
        commit();
        PC_BARRIER();                                     // F3 (the sums buffer in LDS is free again)
        if (!eoff_) {
            for (int b = 0; b < NB; ++b) step_batch(b, -1);
            if (stats) stash();
        }
        PC_BARRIER();                                     // F4
        commit();
        if (rd_z && a.amax_out) {                         // one integer atomic max per E wave (bits of a non-negative float order like integers)
            const float m = wave_max(amx);
            if (lane == 0) atomicMax(a.amax_out + ((blockIdx.x * 4 + ew) % BH_AMAX_SLOTS) * BH_AMAX_STRIDE, __builtin_bit_cast(unsigned, m));
        }
    }
}
}  // namespace

#ifdef BH_TUNING
extern "C" int bh_debug_read_pc_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_ts), sizeof(unsigned long long) * 12 * 160 * 2, 0, hipMemcpyDeviceToHost);
}
#endif

int bh_conv3x3_pc_launch(C3Args& a, int dgrad, const float* bni_table, int bni_groups, int bni_relu, bool force, hipStream_t stream) {
    if (a.Kc % 32 || a.Nn % 64 || a.subtiles < 1) return BH_E_UNSUPPORTED;
    // both sub-tiles of a tile position lie in one statistics group / BatchNorm-on-load group
    const long long sub_per_group = (long long)a.imgs_per_group * a.tiles_per_img;
    if ((a.bn_sums || a.bnr_z) && (sub_per_group % 2)) return BH_E_UNSUPPORTED;
    if (a.accumulate && a.res) return BH_E_UNSUPPORTED;          // (one added tensor: the old gradient of a join OR the inference path's residual)
    if (a.bnr_z && !a.bn_sums) return BH_E_UNSUPPORTED;
    if (bni_table) {
        if (dgrad || bni_groups < 1 || a.N % bni_groups || ((long long)(a.N / bni_groups) * a.tiles_per_img) % 2) return BH_E_UNSUPPORTED;
        a.bni = bni_table; a.bni_groups = bni_groups; a.bni_relu = bni_relu; a.bni_ipg = a.N / bni_groups;
    }
    const int em = (a.bn_sums ? 1 : 0) | ((a.accumulate || a.res) ? 2 : 0) | (a.bnr_z ? 4 : 0) | ((a.bnr_z && a.bnr_relu && a.bnr_y) ? 8 : 0);
    // instantiated: forward {plain, statistics, residual} x BatchNorm-on-load {no, yes (no residual)}; dgrad {plain, column sums, accumulate,
    // BatchNorm sums [+ accumulate] [+ saved output]}
    typedef void (*kern_t)(C3Args);
    struct Row { int dgrad, bni, em; kern_t fn; };
#define PC_ROW(D, B, E) {D, B, E, conv3x3_pc_kernel<(D) != 0, (B) != 0, E>}
    static const Row rows[] = {PC_ROW(0, 0, 0), PC_ROW(0, 0, 1), PC_ROW(0, 0, 2), PC_ROW(0, 1, 0), PC_ROW(0, 1, 1),
                               PC_ROW(1, 0, 0), PC_ROW(1, 0, 1), PC_ROW(1, 0, 2), PC_ROW(1, 0, 5), PC_ROW(1, 0, 7), PC_ROW(1, 0, 13), PC_ROW(1, 0, 15)};
#undef PC_ROW
    constexpr int NROWS = sizeof(rows) / sizeof(rows[0]);
    kern_t fn = nullptr;
    for (int i = 0; i < NROWS; ++i)
        if (rows[i].dgrad == (dgrad ? 1 : 0) && rows[i].bni == (bni_table ? 1 : 0) && rows[i].em == em) fn = rows[i].fn;
    if (!fn) return BH_E_UNSUPPORTED;
    a.gx_total = (a.subtiles + 1) / 2;
    const long long total = (long long)a.gx_total * (a.Nn / 64);
    const int cus = 256;
    const int T = (int)((total + cus - 1) / cus);
    // Where the persistent kernel is the faster one today (tests/test_conv_pc_gpu.py __main__, one MI355X): every forward form and the plain
    // dgrad; the dgrad forms whose epilogue READS tensors (old gradient, BatchNorm input / output) only with one tile per workgroup - with
    // several, the epilogue waves' loads (compiler-counted, one HBM round trip per barrier phase) hold the consumers up and the halo kernel
    // wins by 0 - 5 %.  BH_ROUTE_C3_PC takes the launch regardless (tests, A/B).
    if (!force && (em & 14) && dgrad && T > 1) return BH_E_UNSUPPORTED;
    const int grid = (int)((total + T - 1) / T);
    a.tpb = T;
    a.NW = a.Nn / 32;
    if (bh_query("conv3x3_pc_kernel<%s,%s,%d>", dgrad ? "true" : "false", bni_table ? "true" : "false", em)) return BH_OK;
    static unsigned long long attr_devs = 0;
    if (bh_device_once(attr_devs)) {
        for (int i = 0; i < NROWS; ++i) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
            if (e != hipSuccess) return (int)e;
        }
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(768), PC_LDS, stream, a);
    BH_LAUNCH_CHECK();
    return BH_OK;
}
