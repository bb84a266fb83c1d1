// Winograd F(4x4, 3x3) convolution with the channel sum on the BF16 matrix pipe, float32-accurate: both operands are split
// exactly into three bf16 pieces (x = x1 + x2 + x3, 8 significant bits each) and the six products above 2^-24 relative are
// accumulated in float32 by v_mfma_f32_32x32x16_bf16 - 32 pipe cycles for K = 16 where v_mfma_f32_32x32x2_f32 needs 64 for K = 2
// (round 6; option "winograd" = 3, `precision: split`; the fp32-MFMA kernel of wino4_kernel.hip stays the default).
//
//   sum_c V[c] U[c]  ~  sum_c  v1 u1 + (v1 u2 + v2 u1) + (v1 u3 + v2 u2 + v3 u1)        dropped: v2 u3 + v3 u2 + v3 u3 <= 3 x 2^-24 |V U|
//
// "K-folded" operands: a lane holds 4 input channels of its tile (one 16-byte slot); the 8 K slots of a lane are those 4 channels x 2
// pieces, so 8-channel groups stay as in the fp32 kernel and one MFMA sums TWO of the six products over the 8 channels of a group:
//     A = [v3|v1] x B = [u1|u3]      A = [v1|v2] x B = [u2|u1]      A = [v2|v1] x B = [u2|u1]
//
// Wave = (transform ROW xi, HALF of its six points) x all 64 output channels of the workgroup: the column transform and the split of
// a point (5.5 vector instructions per value: v_and, v_sub, v_and, v_sub + 1.5 v_perm) feed two 32-channel MFMA column blocks.
// Per 8-channel group a wave runs P0, P1, P2 (one point each: column transform, split, 6 MFMAs, LDS-DMA of a later filter stage
// behind them).  What the kernel looked like on the way here, with numbers, is in EXPERIMENTS.md (round 6):
//   v1  every wave reads the raw halo rows of its transform row itself (fp32 kernel's scheme): 518 KB of LDS traffic per group;
//   v2  the ROW transform done once per workgroup and group by all 768 threads (row_pass) into a t image in LDS, which a wave reads
//       as five 16-byte columns: 355 KB of LDS traffic, 40 % fewer vector instructions - and slower, until
//   v3  (this file) the three waves of a SIMD were rotated against the two barriers again, the filter stream went two stages ahead
//       and the raw halo image put a pixel's two 16-byte halves next to each other (32 instead of 64 cache lines per LDS-DMA).
// The bound is none of the pipes: the LDS-DMA path of a CU sustains ~70 GB/s from L2-resident data (tools/micro/ldsdma_rate.hip; 34
// from the Infinity Cache) and this kernel needs 131 KB per group and workgroup - 110 KB of it filter, six bytes per element, used
// for 32 tiles only (36 points x 96 accumulator registers bound M x N of a fused Winograd workgroup).  Without any DMA the K loop runs
// 2.2x the fp32 kernel's rate, with it 1.27x (tools/experiments/w4s_ablate.sh): 256 -> 256 at 64 x 64 x 280 windows 4.29 -> 3.45 ms.
//
// Filter image (written on the device from the fp32 image of winograd4_filter by wino4s_filter_kernel): per 64-channel output
// block, 8-channel group, point slot and wave one 3-KB stage = three self-contained 1-KB LDS-DMA pieces: [block 0: 64 lanes x [u2|u1]]
// [block 1: 64 x [u2|u1]] [64 x u3 of block 0, 64 x u3 of block 1].  Output stage: a wave folds its half row, the two halves of a row
// meet in the exchange image (half 0 writes, half 1 adds), then wino4_combine.inc as in the fp32 kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_util.h"

namespace ecseg {

#include "wino4_consts.inc"
#define W4_HALO_RING 2
// raw image of this kernel (only row_pass reads it): plain row / column order, the two 16-byte channel halves of a pixel NEXT to each other -
// neighbouring lanes of a halo LDS-DMA then read the 32 contiguous bytes of one pixel (the fp32 kernel's image, laid out for its MFMA
// operand reads, puts them 18 lanes apart: 64 cache lines per instruction; measured on this kernel: -0.3 ms of 3.6 on 256 -> 256 at 64 x 64)
#define W4_HALO_SLOT(r, cc) const int h = (cc) & 1, hy = (r), hx = (cc) >> 1
#define W4_HALO_UPPER(cc) ((cc) & 1)

#ifndef ECSEG_W4_TSLOTS
#define ECSEG_W4_TSLOTS 4
#endif
// Timing-only ablations (no filter DMA, no halo DMA, no MFMAs, no split arithmetic, hot / contiguous sources) exist only in A/B builds
// with -DECSEG_W4S_ABL=<bits> (tools/w4s_variants.sh, csrc/wino4s_diag.inc); the product translation unit has ONE code path: every
// hook below is empty.
#ifdef ECSEG_W4S_ABL
#include "wino4s_diag.inc"
#else
#define W4_DIAG_SKIP_HALO_DMA()
#define W4_DIAG_HALO_OFFSET(off, a)
#define W4S_DIAG_SKIP_FILTER_DMA()
#define W4S_DIAG_STAGE(stage) (stage)
#define W4S_DIAG_NO_SPLIT 0
#define W4S_MFMA(CB, A, B) acc[P][CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[P][CB], 0, 0, 0)
#endif
#define ESTAMP(i)
#define W4_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

namespace {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef u32x4 __attribute__((aligned(8))) u32x4_a8;

constexpr int W4S_STAGE = 3072;      // bytes of one filter stage: 2 column blocks x 64 lanes x 24
constexpr int W4S_TS = 2 * 6 * 4 * 36;   // slots of the t image: 2 regions x 6 transform rows x 4 tile rows x (18 columns x 2 channel halves)

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bfloat(unsigned v) { return __builtin_bit_cast(float, v); }
// the high halves of two floats = their bf16 truncations, packed (lo in bits 0-15)
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(fbits(hi), fbits(lo), 0x07060302u); }
}  // namespace

template <bool HEAD>
__global__ __launch_bounds__(768) void conv_wino4s_kernel(ConvParams p, int regs_x, int regs_y, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [2][W4_HS]          raw halo (group g -> buffer g & 1)
    f32x4* Ts = Hs + 2 * W4_HS;                              // [W4S_TS]            row-transformed halo of ONE group: t[region][xi][tile row][18 columns x 2 halves]
    char* Bs = reinterpret_cast<char*>(Ts + W4S_TS);         // [12][2][W4S_STAGE]  per-wave filter stages

    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave % 6, hr = wave / 6;                  // transform row, half row (points 0 1 2 | 5 3 4 in stage order)
    const int li = lane & 31, lh = lane >> 5;

#include "wino4_region.inc"
    (void)nstages;
    // ---- filter DMA: image [nb][group][point slot][wave][3072 B]; scalar base + the lane's constant 16-byte offset ----
    const unsigned long long w_base = (unsigned long long)(size_t)p.wt + ((size_t)nb * (3 * ngroups) * 12 + wave) * W4S_STAGE;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma_filter_piece = [&](int stage, int buf, auto kk) __attribute__((always_inline)) {
        constexpr int k = decltype(kk)::value;
        W4S_DIAG_SKIP_FILTER_DMA();
        const unsigned long long g = w_base + (unsigned long long)W4S_DIAG_STAGE(stage) * (12 * W4S_STAGE);
        const unsigned dst = lds_base + (unsigned)((2 * W4_HS + W4S_TS) * 16 + (wave * 2 + buf) * W4S_STAGE);
        const unsigned l16 = lane16;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(dst), "s"(g), "n"(k * 1024) : "memory");
    };

    // ---- the lane's tile (as in the fp32 kernel: the b128 lane groups of a region read conflict-free) and its slot in the t image ----
    const int q8 = li >> 2, tx = li & 3;
    const int tg = (0x96 >> q8) & 1;
    const int ty = (q8 == 0 || q8 == 1) ? 0 : (q8 == 2 || q8 == 3) ? 1 : (q8 == 4 || q8 == 5) ? 2 : 3;
    // t image: slot(region, xi, tile row, column, half) = ((region * 6 + xi) * 4 + tile row) * 36 + half * 18 + P(column): the row
    // stride 36 = 4 (mod 16) and the regrouped columns make the 16 lanes of a b128 group hit 16 different bank groups, as in the raw halo
    const int t_lane = ((tg * 6 + xi) * 4 + ty) * 36 + lh * 18 + tx;
    // halo columns of a half row: half 0 needs columns 0..4 (points 0, +-a), half 1 columns 1..5 (points +-b, inf).  Both keep
    // columns 1..4 in t[1..4]; t[0] is column 0 (half 0) or column 5 (half 1): ONE wave-uniform slot offset, no second code path
    const int xcol = hr ? w4_cpos(5) : w4_cpos(0);
    // column transform of a half row, with wave-uniform constants: e = t4 - kq t2, o = t3 - kq t1, points e +- kr o;
    // the third point is 0 (half 0: KP t0 + KS t2 + t4) or infinity (half 1: KP t1 + KS t3 + t0): stage 0 of both halves
    const float kq = hr ? KA2 : KB2, kr = hr ? KB : KA;

    f32x16 acc[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[v][cb][e] = 0.f;

    // ---- the row transform, ONCE per workgroup and group (round 6, second version).  In the fp32 kernel - and in this kernel's first
    //      version - every wave reads the raw halo rows of its transform row itself: 20 ds_read_b128 per wave and group, 240 KB of LDS
    //      reads per group for 21 KB of data, and with the channel sum three times faster that traffic (+ 144 KB of filter reads +
    //      134 KB of LDS-DMA writes = 518 KB at 128 B / clock) - not the matrix pipe, not the vector pipe - set the pace
    //      (tools/experiments/w4s_ablate.sh).  Now 288 threads transform the group cooperatively: item = (region, tile row, column,
    //      channel half): six raw rows in, t[xi = 0..5] out (12 fmas per channel, the +- rows share their even / odd parts), written
    //      to the t image; a wave then reads the five columns of ITS row: raw 28 KB + t 28 KB written + 60 KB read. ----
    auto row_pass = [&](int grp) __attribute__((always_inline)) {
        // 576 half items (region, tile row, channel half, column, channel PAIR) over the 12 waves x 48 lanes: every wave carries the
        // same share (a pass run by five waves alone left the other seven waiting at the barrier behind it), 8-byte accesses,
        // neighbouring lanes on neighbouring addresses
        if (lane >= 48) return;
        const int item = wave * 48 + lane, cpair = item & 1, h = (item >> 1) & 1, k = item >> 2;
        const int x = k % 18, r2 = k / 18, tyy = r2 & 3, tgg = r2 >> 2;
        const f32x2* R = reinterpret_cast<const f32x2*>(Hs + (grp & 1) * W4_HS + (tgg * 18 + 4 * tyy) * 36 + 2 * x + h) + cpair;     // raw row 4 tyy + i: + 36 i slots
        const f32x2 d0 = R[2 * 36 * 0], d1 = R[2 * 36 * 1], d2 = R[2 * 36 * 2], d3 = R[2 * 36 * 3], d4 = R[2 * 36 * 4], d5 = R[2 * 36 * 5];
        f32x2* T = reinterpret_cast<f32x2*>(Ts + ((tgg * 6) * 4 + tyy) * 36 + h * 18 + w4_pos(x)) + cpair;                            // + xi * 144 slots
        // t[xi] = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3] as the SAME fma chains conv_wino4_kernel's per-wave row transform runs (innermost term
        // first): bit-identical t, hence bit-identical results in the fp32 kernel.  (A first version shared the even / odd parts of the +- rows,
        // 12 instead of 16 fmas per channel: the smooth fixture model's wrong-pixel total rose from 11 to 19 of ~15 hard pixels per image.)
        f32x2 o;
#define W4_ROW3(XI, A0, DA, A1, DB, DC) do { _Pragma("unroll") for (int c = 0; c < 2; ++c) o[c] = __builtin_fmaf(A0, DA[c], __builtin_fmaf(A1, DB[c], DC[c])); T[2 * (XI) * 144] = o; } while (0)
#define W4_ROW4(XI, A0, DA, A1, DB, A2, DC, DD) do { _Pragma("unroll") for (int c = 0; c < 2; ++c) \
            o[c] = __builtin_fmaf(A0, DA[c], __builtin_fmaf(A1, DB[c], __builtin_fmaf(A2, DC[c], DD[c]))); T[2 * (XI) * 144] = o; } while (0)
        W4_ROW3(0, KP, d0, KS, d2, d4);
        W4_ROW3(5, KP, d1, KS, d3, d5);
        W4_ROW4(1, -KA * KB2, d1, -KB2, d2, KA, d3, d4);
        W4_ROW4(2, KA * KB2, d1, -KB2, d2, -KA, d3, d4);
        W4_ROW4(3, -KA2 * KB, d1, -KA2, d2, KB, d3, d4);
        W4_ROW4(4, KA2 * KB, d1, -KA2, d2, -KB, d3, d4);
#undef W4_ROW3
#undef W4_ROW4
    };
    f32x4 t[5];
    auto load_t = [&]() __attribute__((always_inline)) {
        const f32x4* A = Ts + t_lane;
        t[0] = A[xcol];
        t[1] = A[w4_cpos(1)]; t[2] = A[w4_cpos(2)]; t[3] = A[w4_cpos(3)]; t[4] = A[w4_cpos(4)];
    };
    // ---- one point: column transform of the lane's 4 channels, exact 3-way bf16 split, 6 MFMAs; the three pieces of the next
    //      filter stage go out behind the first MFMAs ----
    float Vm[4];                                             // the "-" point of the +- pair (computed in P1 with its partner, used in P2)
    auto point = [&](auto pp, int fbuf, int ahead_stage) __attribute__((always_inline)) {
        constexpr int P = decltype(pp)::value;
        // Filter stage: [column block 0: 64 x [u2|u1]] [column block 1: 64 x [u2|u1]] [64 x u3 of block 0][64 x u3 of block 1]: one aligned
        // 16-byte and one 8-byte read per block - 24 bytes per lane (the first version read [u2|u1] and [u1|u3] as two overlapping
        // 16-byte windows: 32).  Register budget (168 at three waves per SIMD; 96 accumulators + 20 of t[] are always live): block 0's
        // fragments are read first, the split runs under their latency, block 1's are read when the split's temporaries are dead -
        // the order is pinned (sched_barrier): the compiler's own schedule hoists all reads and spills.
        const char* fp = Bs + (wave * 2 + fbuf) * W4S_STAGE;
        const u32x4 b12a = reinterpret_cast<const u32x4*>(fp)[lane];
        const u32x2 u3a = reinterpret_cast<const u32x2*>(fp + 2048)[lane];
        __builtin_amdgcn_sched_barrier(0);
        float V[4];
        if (P == 0) {
            if (hr == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) V[c] = __builtin_fmaf(KP, t[0][c], __builtin_fmaf(KS, t[2][c], t[4][c]));
            } else {
                // (the empty asm keeps the two arms apart: merged, they become ONE fma chain over t[hr], t[2 + hr], ... - a dynamically
                // indexed t[] lives in scratch memory)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = t[1][c];
                    asm volatile("" : "+v"(x));
                    V[c] = __builtin_fmaf(KP, x, __builtin_fmaf(KS, t[3][c], t[0][c]));
                }
            }
        } else if (P == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = __builtin_fmaf(-kq, t[2][c], t[4][c]), o = __builtin_fmaf(-kq, t[1][c], t[3][c]);
                V[c] = __builtin_fmaf(kr, o, e);
                Vm[c] = __builtin_fmaf(-kr, o, e);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) V[c] = Vm[c];
        }
        // exact split by truncation: v1 = high half of v, v2 = high half of (v - v1), v3 = v - v1 - v2 (8 bits left: exact)
        u32x8 a8;
        {
            float r[4], s[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = W4S_DIAG_NO_SPLIT ? V[c] : V[c] - bfloat(fbits(V[c]) & 0xffff0000u);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = W4S_DIAG_NO_SPLIT ? V[c] : r[c] - bfloat(fbits(r[c]) & 0xffff0000u);
            const unsigned p1a = pack_hi(V[0], V[1]), p1b = pack_hi(V[2], V[3]);
            a8 = u32x8{pack_hi(s[0], s[1]), pack_hi(s[2], s[3]), p1a, p1b, pack_hi(r[0], r[1]), pack_hi(r[2], r[3]), p1a, p1b};
        }
        const bf16x8 A3 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3));      // [v3|v1]
        const bf16x8 A2 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 2, 3, 4, 5));      // [v1|v2]
        const bf16x8 A1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 4, 5, 6, 7));      // [v2|v1]
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 b12b = reinterpret_cast<const u32x4*>(fp + 1024)[lane];
        const u32x2 u3b = reinterpret_cast<const u32x2*>(fp + 2048 + 512)[lane];
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 B12a = __builtin_bit_cast(bf16x8, b12a), B3a = __builtin_bit_cast(bf16x8, u32x4{b12a[2], b12a[3], u3a[0], u3a[1]});
        const bf16x8 B12b = __builtin_bit_cast(bf16x8, b12b), B3b = __builtin_bit_cast(bf16x8, u32x4{b12b[2], b12b[3], u3b[0], u3b[1]});
        // The stage streamed from here is TWO stages ahead and goes into the buffer this phase is reading: piece 0 (block 0's [u2|u1]) once
        // those fragments are in registers (behind the second MFMA), pieces 1 and 2 (block 1's, the u3 of both) behind the first MFMAs
        // of block 1.  One stage ahead (the first version) left a stage ~0.8 of a phase to land and every phase began with a wait.
        W4S_MFMA(0, A3, B3a);
        W4S_MFMA(0, A2, B12a);
        __builtin_amdgcn_sched_barrier(0);
        dma_filter_piece(ahead_stage, fbuf, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        W4S_MFMA(0, A1, B12a);
        W4S_MFMA(1, A3, B3b);
        __builtin_amdgcn_sched_barrier(0);
        dma_filter_piece(ahead_stage, fbuf, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        W4S_MFMA(1, A2, B12b);
        __builtin_amdgcn_sched_barrier(0);
        dma_filter_piece(ahead_stage, fbuf, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        W4S_MFMA(1, A1, B12b);
        __builtin_amdgcn_sched_barrier(0);                   // (MFMAs are no memory operations: without this they sink below the s_barrier that follows)
    };
#define W4_BARRIER() asm volatile("s_barrier" ::: "memory")
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
    // Barriers and phases.  Per group g: barrier Y(g-1) - the t image holds group g and raw buffer g & 1 is free (this wave's halo pieces
    // of group g + 2 go out right behind it) - and barrier X(g) - nobody needs the t image of group g any more and the raw halo of
    // group g + 1 has landed: row_pass(g + 1) follows it.  Between them every wave runs its three point phases, but the three waves
    // of a SIMD (w, w + 4, w + 8: one of each class) sit at different points of the sequence, so the LDS latencies of one run under
    // the split arithmetic and MFMAs of the others:
    //   class 0:   Y(g-1) | t <- image, P0(g), P1(g)         | X(g) | row_pass(g+1), P2(g)
    //   class 1:   Y(g-1) | t <- image, P2(g-1), P0(g)       | X(g) | row_pass(g+1), P1(g)
    //   class 2:   Y(g-1) | P1(g-1), t <- image, P2(g-1)     | X(g) | row_pass(g+1), P0(g)
    // (P1 is the last reader of t[], P2 needs only Vm[].)  Filter stage s = 3 g + P lives in buffer s & 1 and is streamed TWO stages
    // ahead (point()).  Waits - a wave counts only its own LDS-DMAs, in issue order: phase s needs stage s; younger than it are the three
    // pieces of stage s + 1 and, when barrier Y lies between phase s - 2 and phase s, the two halo pieces issued behind Y: vmcnt(5) for
    // the two phases that follow Y, vmcnt(3) for the third (and for all of them once no halo is left to fetch).
#define W4_PH(PP, s, H) do { W4_SB(); if (H) W4_WAIT(5); else W4_WAIT(3); W4_SB(); \
                             point(std::integral_constant<int, PP>{}, (s) & 1, (s) + 2 < 3 * ngroups ? (s) + 2 : 3 * ngroups - 1); } while (0)
#define W4_Y() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_BARRIER(); W4_SB(); } while (0)      /* (own share of the row pass is written) */
#define W4_X() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_BARRIER(); W4_SB(); } while (0)      /* (own t[] is loaded) */
#define W4_HALO(g) do { dma_halo_piece((g), std::integral_constant<int, 0>{}); dma_halo_piece((g), std::integral_constant<int, 1>{}); } while (0)
    const int cls = wave >> 2;
    W4_HALO(0);
    if (ngroups > 1) W4_HALO(1);
    dma_filter_piece(0, 0, std::integral_constant<int, 0>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 1>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 2>{});
    dma_filter_piece(1, 1, std::integral_constant<int, 0>{});
    dma_filter_piece(1, 1, std::integral_constant<int, 1>{});
    dma_filter_piece(1, 1, std::integral_constant<int, 2>{});
    if (ngroups > 1) W4_WAIT(8); else W4_WAIT(6);            // raw group 0 has landed
    W4_BARRIER();
    row_pass(0);
    if (cls == 0) {
        for (int grp = 0; grp < ngroups; ++grp) {
            W4_Y();
            load_t();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            W4_PH(0, 3 * grp, mh);
            W4_PH(1, 3 * grp + 1, mh);
            if (grp + 1 < ngroups) { W4_X(); row_pass(grp + 1); W4_SB(); }
            W4_PH(2, 3 * grp + 2, false);
        }
    } else if (cls == 1) {
        {
            W4_Y();
            load_t();
            const bool mh = 2 < ngroups;
            if (mh) W4_HALO(2);
            W4_PH(0, 0, mh);
            if (1 < ngroups) { W4_X(); row_pass(1); W4_SB(); }
            W4_PH(1, 1, mh);                                 // (stricter than needed when halo pieces went out: they are older than stage 2 here)
        }
        for (int grp = 1; grp < ngroups; ++grp) {
            W4_Y();
            load_t();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            W4_PH(2, 3 * grp - 1, mh);
            W4_PH(0, 3 * grp, mh);
            if (grp + 1 < ngroups) { W4_X(); row_pass(grp + 1); W4_SB(); }
            W4_PH(1, 3 * grp + 1, false);
        }
        W4_PH(2, 3 * ngroups - 1, false);
    } else {
        {
            W4_Y();
            const bool mh = 2 < ngroups;
            if (mh) W4_HALO(2);
            load_t();
            if (1 < ngroups) {
                if (mh) W4_WAIT(8); else W4_WAIT(6);         // raw group 1 has landed (this class has not waited for anything since the prologue)
                W4_X(); row_pass(1); W4_SB();
            }
            W4_PH(0, 0, mh);
        }
        for (int grp = 1; grp < ngroups; ++grp) {
            W4_Y();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            W4_PH(1, 3 * grp - 2, mh);
            W4_SB();
            load_t();
            W4_PH(2, 3 * grp - 1, mh);
            if (grp + 1 < ngroups) { W4_X(); row_pass(grp + 1); W4_SB(); }
            W4_PH(0, 3 * grp, false);
        }
        W4_PH(1, 3 * ngroups - 2, false);
        W4_PH(2, 3 * ngroups - 1, false);
    }
#undef W4_PH
#undef W4_Y
#undef W4_X
#undef W4_HALO
#undef W4_SB
#undef W4_BARRIER

    // ---- output stage: two passes (column blocks) through the [xi][x][tile][32 couts] exchange image ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the compiler does not see the asm LDS-DMAs
    float* Rs = reinterpret_cast<float*>(smem);
    const int Cout = p.out.c;
    float hl[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) hl[a][b][c] = 0.f;
    f32x4 bvp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (p.bias != nullptr) {
        bvp[0] = *reinterpret_cast<const f32x4*>(p.bias + nb * 64 + 4 * (tid & 7));
        if (nb * 64 + 32 < Cout) bvp[1] = *reinterpret_cast<const f32x4*>(p.bias + nb * 64 + 32 + 4 * (tid & 7));
    }
    // R = M[xi][:] A of a HALF row (A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a2 a2 b2 b2 0; 0 a3 -a3 b3 -b3 1]):
    //   half 0 (m0, m+a, m-a):   r0 = m0 + s, r1 = a d, r2 = a2 s, r3 = a3 d            s = m+ + m-, d = m+ - m-
    //   half 1 (minf, m+b, m-b): r0 = s, r1 = b d, r2 = b2 s, r3 = b3 d + minf
    const float k1 = hr ? KB : KA, k2 = hr ? KB2 : KA2, k3 = hr ? KB3 : KA3;
    auto write_R = [&](auto cbc, auto add_c) __attribute__((always_inline)) {
        constexpr int cb = decltype(cbc)::value;
        constexpr bool add = decltype(add_c)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float me = acc[0][cb][e], mp = acc[1][cb][e], mm = acc[2][cb][e];
            const float sm = mp + mm, df = mp - mm;
            float* o = Rs + (xi * 4) * W4_RPLANE + tl * 32 + li;
            const float r1 = k1 * df, r2 = k2 * sm;
            if (add) {
                o[0 * W4_RPLANE] += sm; o[1 * W4_RPLANE] += r1; o[2 * W4_RPLANE] += r2; o[3 * W4_RPLANE] += __builtin_fmaf(k3, df, me);
                // (four accumulator rows at a time: left alone, the compiler reads all 64 words first and spills accumulators for them)
                if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            } else {
                o[0 * W4_RPLANE] = me + sm; o[1 * W4_RPLANE] = r1; o[2 * W4_RPLANE] = r2; o[3 * W4_RPLANE] = k3 * df;
            }
        }
    };
    // (two straight-line passes, not a loop: in a loop the fold arithmetic of BOTH column blocks is loop-invariant, gets hoisted in
    // front of it and spills ~90 registers)
    auto do_pass = [&](auto passc) __attribute__((always_inline)) {
        constexpr int pass = decltype(passc)::value;
        __syncthreads();                                     // main-loop LDS reads / previous pass's combine are done
        if (hr == 0) write_R(passc, std::false_type{});
        __syncthreads();
        if (hr == 1) write_R(passc, std::true_type{});
        __syncthreads();
#include "wino4_combine.inc"
    };
    do_pass(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    do_pass(std::integral_constant<int, 1>{});
#include "wino4_head.inc"
}

// The fp32 image of winograd4_filter (api.hip) -> the bf16x3 stage image of conv_wino4s_kernel; one thread per (block, stage, wave,
// column block, lane): 4 channels x 3 pieces.
__global__ __launch_bounds__(256) void wino4s_filter_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int nblk, int ngroups) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nblk * 3 * ngroups * 12 * 128;
    if (id >= total) return;
    const int lane = (int)(id & 63), cb = (int)((id >> 6) & 1);
    long long r = id >> 7;
    const int wave = (int)(r % 12); r /= 12;
    const int st = (int)(r % (3 * ngroups)); const int nb = (int)(r / (3 * ngroups));
    const int g = st / 3, ps = st % 3;
    const int xi = wave % 6, hr = wave / 6;
    const int nu = hr ? (ps == 0 ? 5 : ps == 1 ? 3 : 4) : ps;
    const int m = lane & 31, kb = lane >> 5;
    // stage layout: [block 0: 64 lanes x [u2|u1]] [block 1: 64 x [u2|u1]] [64 x u3 of block 0] [64 x u3 of block 1]
    unsigned short* o = dst + (((long long)(nb * 3 * ngroups + st) * 12 + wave) * (W4S_STAGE / 2));
    unsigned short* o12 = o + cb * 512 + lane * 8;
    unsigned short* o3 = o + 1024 + cb * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ss = c >> 1, e = c & 1;
        const long long idx = (((long long)nb * (2 * ngroups) + (2 * g + ss)) * 12 + (cb * 6 + xi)) * 768 + ((((long long)(nu >> 1) * 64 + kb * 32 + m) * 2 + (nu & 1)) * 2) + e;
        const float u = src[idx];
        const unsigned b1 = fbits(u) & 0xffff0000u;
        const float r1 = u - bfloat(b1);
        const unsigned b2 = fbits(r1) & 0xffff0000u;
        const float r2 = r1 - bfloat(b2);
        o12[c] = (unsigned short)(b2 >> 16);                 // u2
        o12[4 + c] = (unsigned short)(b1 >> 16);             // u1
        o3[c] = (unsigned short)(fbits(r2) >> 16);           // u3
    }
}

size_t wino4s_image_bytes(int cin, int cout) { return (size_t)((cout + 63) / 64) * 3 * ((cin + 7) / 8) * 12 * W4S_STAGE; }

hipError_t launch_wino4s_filter(const float* wt_wino4, void* dst, int cin, int cout, hipStream_t s) {
    const int nblk = (cout + 63) / 64, ngroups = (cin + 7) / 8;
    const long long total = (long long)nblk * 3 * ngroups * 12 * 128;
    hipLaunchKernelGGL(wino4s_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wt_wino4, reinterpret_cast<unsigned short*>(dst), nblk, ngroups);
    return hipGetLastError();
}

// Same geometry rules as conv_wino4_kernel, and whole 64-channel output blocks (a lone 32-channel block would spend half of every
// MFMA pair on padding: such layers stay on the fp32 kernel's SPLIT variant).
bool conv_wino4s_supported(const ConvParams& p) { return conv_wino4_supported(p) && p.out.c % 64 == 0; }

hipError_t launch_conv_wino4s(const ConvParams& p, hipStream_t s) {
    const int regs_x = p.out.w / 16, regs_y = p.out.h / 16;
    const size_t nreg = p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * regs_x * regs_y;
    const size_t npairs = (nreg + 1) / 2;
    const size_t grid = npairs * (size_t)(p.out.c / 64);
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull || !conv_wino4_span_ok(p, p.lut != nullptr ? p.per_image : 2)) return hipErrorInvalidValue;
    if (p.head_w != nullptr && (!p.head_only || p.pool.p != nullptr)) return hipErrorInvalidValue;     // (the HEAD kernels write neither the features nor a pool)
    size_t lds = (size_t)(2 * W4_HS + W4S_TS) * 16 + (size_t)12 * 2 * W4S_STAGE;
    const size_t lds_epi = (size_t)24 * W4_RPLANE * 4;
    if (lds_epi > lds) lds = lds_epi;
    void (*kern)(ConvParams, int, int, int) = p.head_w != nullptr ? conv_wino4s_kernel<true> : conv_wino4s_kernel<false>;
    static DeviceOnce attr_set[2];
    const hipError_t ea = attr_set[p.head_w != nullptr ? 1 : 0].run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(768), lds, s, p, regs_x, regs_y, (int)npairs);
    return hipGetLastError();
}

}  // namespace ecseg
