// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores (gfx950): 36 instead of 144 multiplies per 4x4 output
// block, i.e. 4x fewer MFMAs than the direct convolution and 1.78x fewer than F(2x2,3x3), still exact-f32 fma chains
// (v_mfma_f32_32x32x2_f32).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 6x6 input tile, g: 3x3 filter, Y: 4x4 outputs
//
// The filter transform U = G g G^T is done once at model load (float64 on the host, api.hip: winograd4_filter).
//
// Workgroup = 12 waves = 2 regions of 16x16 output pixels (32 Winograd tiles = the MFMA M dimension) x 64 output
// channels.  Wave w = (channel half ch = w / 6, transform ROW xi = w % 6): it reads the raw halo rows its row transform
// needs straight from LDS, forms t[j] = B^T[xi,:] d[:,j] and the six column points V[xi][0..5] in registers and
// multiplies them with U[xi][nu] on the MFMA (6 points x 16 accumulators = 96 registers); no transformed input ever
// touches LDS or HBM.  After the K loop every wave folds its own row (R = M[xi][:] A) in registers, the six rows
// meet through LDS, and Y = A^T R + bias, activation is written as 16-byte stores (128-B segments per pixel).
//
// Pipeline: the halo arrives 8 input channels at a time by LDS-DMA into a 3-deep ring, two groups ahead of its use
// (every wave issues two pieces per group, one at a time behind pinned MFMAs; nobody waits for a piece to land; ONE
// s_barrier per 8 channels).  The filter fragments are private to a wave (nobody else reads them), so every wave
// streams its own 3-KB stage (6 points x 4 input channels x 32 output channels) by LDS-DMA into a private double
// buffer, ordered by its own counted vmcnt only - no barrier on the filter path.  All LDS-DMA goes through inline asm:
// the compiler orders every ds_read behind a builtin LDS-DMA with vmcnt(0).  The barrier sits at a different
// point of the phase sequence (transform, MFMA stage 0, MFMA stage 1) for the three waves of a SIMD (phase rotation).
//
// The output stage can also write the 2x2 max-pool of its result, finish a 1x1 head (<= 4 classes) and take its region
// list from a look-up table (demand-driven cropping) - see ConvParams in common.h.
//
// LDS halo image, 16-byte slots (4 channels): slot(g, y, x, h) = (g * 18 + P(y)) * 36 + h * 18 + P(x), where
// P(v) = {0, 5, 10, 14}[v % 4] + v / 4 regroups the 18 halo rows / columns by their phase modulo the tile stride 4.
// For a fixed tile offset (i, j) the 16 lanes of a ds_read_b128 group then read slots 36 * ty + tx + const:
// 36 = 4 (mod 16), so all 16 land on different 16-byte bank groups: conflict-free without padding.  The A-operand
// lane -> tile map follows the hardware's b128 lane groups (see gty below).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_util.h"

namespace ecseg {

#include "wino4_consts.inc"
#define W4_HALO_RING 3
// raw image of this kernel: rows and columns regrouped by their phase modulo the tile stride (see the header), the two channel halves 18 slots apart
#define W4_HALO_SLOT(r, cc) const int h = (cc) >= 18 ? 1 : 0, hy = w4_inv(r), hx = w4_inv((cc) - h * 18)
#define W4_HALO_UPPER(cc) ((cc) >= 18)

// Row-transform pipeline depth (slots of in-flight halo reads beside the six direct ones; 0: rounds 1-3, column by column)
#ifndef ECSEG_W4_TSLOTS
#define ECSEG_W4_TSLOTS 4
#endif

// Diagnostics (timing-only ablations, in-kernel cycle stamps) live in wino4_diag.inc and exist only in the -DECSEG_DIAG
// build (tools/build_variants.sh diag); the product translation unit has ONE code path: every hook below is empty.
#ifdef ECSEG_DIAG
#include "wino4_diag.inc"
#else
#define W4_TEMPLATE template <bool HEAD = false, bool SPLIT = false>
#define W4_DIAG_ENTRY()
#define W4_DIAG_SKIP_HALO_DMA()
#define W4_DIAG_HALO_OFFSET(off, a)
#define W4_DIAG_SKIP_FILTER_DMA()
#define W4_DIAG_FAKE_TRANSFORM(grp)
#define W4_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0)
#define W4_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define W4_KSTAMP_BEGIN()
#define WSTAMP(i)
#define W4_KSTAMP_DUMP()
#define W4_ESTAMP_BEGIN()
#define ESTAMP(i)
#define W4_ESTAMP_DUMP()
#define W4_DIAG_SELECT(kern, p, lds)
#endif

// SPLIT (a lone 32-channel output block, Cout == 32): the two channel-half waves of a transform row would otherwise
// multiply real channels (ch = 0) and zero padding (ch = 1).  Instead both work on the SAME 32 outputs and split the 8
// input channels of a group: wave (ch, xi) runs only filter stage ch (channels 2 ch, 2 ch + 1 of both halo planes) of every
// group, from the ch = 0 slot of the unchanged filter image; the partial sums meet in the exchange image of the output
// stage (ch = 0 writes, ch = 1 adds).  Per group a wave has two phases (T, S) instead of three; waves 4-11 run S one
// group late (phase rotation).
W4_TEMPLATE
__global__ __launch_bounds__(768) void conv_wino4_kernel(ConvParams p, int regs_x, int regs_y, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [3][W4_HS]        halo ring (group g -> buffer g % 3)
    f32x4* Bs = Hs + 3 * W4_HS;                              // [12][2][W4_BWS]   per-wave filter stages

    W4_DIAG_ENTRY();
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;   // LDS byte address of the dynamic segment
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave % 6, ch = wave / 6;
    const int li = lane & 31, lh = lane >> 5;

#include "wino4_region.inc"
    // ---- filter DMA: wt4[nb][stage][wave][point pair nu / 2][lane = h * 32 + cout][nu % 2][k 2], 768 floats per wave and stage; the
    //      address is a scalar base (advanced per stage by scalar adds) + the lane's constant 16-byte offset ----
    const unsigned long long w_base = (unsigned long long)(size_t)(p.wt + ((size_t)nb * nstages * 12 + (SPLIT ? xi : wave)) * 768);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4* Bw = Bs + wave * 2 * W4_BWS;
    auto dma_filter_piece = [&](int stage, int buf, auto kk) __attribute__((always_inline)) {
        W4_DIAG_SKIP_FILTER_DMA();
        constexpr int k = decltype(kk)::value;
        const unsigned long long g = w_base + (unsigned long long)stage * (12 * 768 * 4);
        const unsigned dst = lds_base + (unsigned)(3 * W4_HS + (wave * 2 + buf) * W4_BWS) * 16u;
        const unsigned l16 = lane16;
        unsigned keep;
        // the instruction offset advances the global AND the LDS address: one M0 for the three pieces
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(dst), "s"(g), "n"(k * 1024) : "memory");
    };

    // ---- A-operand lane -> tile.  ds_read_b128 serves lanes {0-3,12-15,20-27} and {4-11,16-19,28-31} of each half
    //      in separate LDS cycles; give each of those groups the 16 tiles of ONE region ----
    const int q8 = li >> 2, tx = li & 3;
    const int tg = (0x96 >> q8) & 1;                              // region of lane quad q8: 0,1,1,0,1,0,0,1
    const int ty = (q8 == 0 || q8 == 1) ? 0 : (q8 == 2 || q8 == 3) ? 1 : (q8 == 4 || q8 == 5) ? 2 : 3;
    const int a_lane = (tg * 18 + ty) * 36 + lh * 18 + tx;       // slot of halo pixel (4 ty, 4 tx) of the lane's tile

    // row transform of wave xi (row xi of B^T): t = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3]  (rows 0 and 5: three terms)
    int rr0, rr1, rr2, rr3; float c0, c1, c2;
    switch (xi) {
        case 0:  rr0 = 0; rr1 = 2; rr2 = 4; rr3 = 4; c0 = KP;        c1 = KS;   c2 = 1.f; break;
        case 1:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -KA * KB2; c1 = -KB2; c2 = KA;  break;
        case 2:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = KA * KB2;  c1 = -KB2; c2 = -KA; break;
        case 3:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -KA2 * KB; c1 = -KA2; c2 = KB;  break;
        case 4:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = KA2 * KB;  c1 = -KA2; c2 = -KB; break;
        default: rr0 = 1; rr1 = 3; rr2 = 5; rr3 = 5; c0 = KP;        c1 = KS;   c2 = 1.f; break;
    }
    const int ro0 = 36 * w4_pos(rr0), ro1 = 36 * w4_pos(rr1), ro2 = 36 * w4_pos(rr2), ro3 = 36 * w4_pos(rr3);

    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;

    f32x4 t[6];
    if (SPLIT) {                                             // (components 2, 3 are never written in this mode: keep the vectors defined)
#pragma unroll
        for (int j = 0; j < 6; ++j) t[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- row transform of group grp: t[j] for the six halo columns of the lane's tile, 4 channels each ----
    // Rows 1-4 of B^T end in +1 (t = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3]: three fmas), rows 0 and 5 have only three
    // terms, the last with +1 (t = c0 d[r0] + c1 d[r1] + d[r2]: three reads, two fmas) - for any point set of this shape.
    const bool inner_row = xi >= 1 && xi <= 4;
    auto transform = [&](int grp, auto c_src, auto c_num) __attribute__((always_inline)) {   // t[j][k] = row transform of channel c_src + k of the lane's slot, k < c_num (SPLIT: the wave's two channels, in components 0 and 1)
        constexpr int CS = decltype(c_src)::value, CN = decltype(c_num)::value;
        const f32x4* A = Hs + (grp % 3) * W4_HS + a_lane;
        constexpr int cp[6] = {w4_cpos(0), w4_cpos(1), w4_cpos(2), w4_cpos(3), w4_cpos(4), w4_cpos(5)};
        W4_DIAG_FAKE_TRANSFORM(grp);
        // scalar fmas on purpose (file is built with -fno-slp-vectorize): packed f32 VALU ops (v_pk_fma_f32) stall the
        // SIMD beside MFMAs, single v_fma_f32 hide in the matrix pipe's shadow
        // Software-pipelined (round 4): the phase used to be six LDS round trips behind each other (one per halo column:
        // 3 - 4 reads, wait, 12 fmas) and took 2200 - 3200 cycles per group - the three transforms of a SIMD's waves add up to the
        // group period (in-kernel stamps).  Now the LAST term of every column is read straight into t[j] (it enters the
        // chain with coefficient 1), the other reads go through NS rotating slots: each step waits for ONE read, accumulates it
        // into its column (four fmas) and re-issues the slot, so NS reads are always in flight.  The fma chain of a column runs in
        // the same order as before (innermost term first): results are bit-identical.
        constexpr int NS = ECSEG_W4_TSLOTS;
        // CN == 4: the lane's whole 16-byte slots; CN == 2 (SPLIT, round 4): only the wave's two channels CS, CS + 1 of every slot
        // (8-byte reads, half the fmas - a SPLIT wave multiplies half as often per group, so the row transform weighed twice as
        // much per MFMA there: 64 -> 32 channels ran at 0.315 of the peak, half the rate of the unsplit layers)
        static_assert((CN == 4 && CS == 0) || (CN == 2 && (CS == 0 || CS == 2)), "whole slots or one channel pair");
        typedef typename std::conditional<CN == 4, f32x4, f32x2>::type vt;
        auto rd = [&](int slot) __attribute__((always_inline)) -> vt {
            if constexpr (CN == 4) return A[slot];
            else return reinterpret_cast<const f32x2*>(A + slot)[CS / 2];
        };
        vt sl[NS];
        if (inner_row) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const vt d = rd(ro3 + cp[j]);
#pragma unroll
                for (int c = 0; c < CN; ++c) t[j][CS + c] = d[c];
            }
#define W4_TRD(i) rd(((i) % 3 == 0 ? ro2 : (i) % 3 == 1 ? ro1 : ro0) + cp[(i) / 3])
#pragma unroll
            for (int k = 0; k < NS; ++k) sl[k] = W4_TRD(k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const float cf = i % 3 == 0 ? c2 : i % 3 == 1 ? c1 : c0;
#pragma unroll
                for (int c = 0; c < CN; ++c) t[i / 3][CS + c] = __builtin_fmaf(cf, sl[i % NS][c], t[i / 3][CS + c]);
                if (i + NS < 18) sl[i % NS] = W4_TRD(i + NS);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef W4_TRD
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const vt d = rd(ro2 + cp[j]);
#pragma unroll
                for (int c = 0; c < CN; ++c) t[j][CS + c] = d[c];
            }
#define W4_TRD(i) rd(((i) % 2 == 0 ? ro1 : ro0) + cp[(i) / 2])
#pragma unroll
            for (int k = 0; k < NS; ++k) sl[k] = W4_TRD(k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const float cf = i % 2 == 0 ? c1 : c0;
#pragma unroll
                for (int c = 0; c < CN; ++c) t[i / 2][CS + c] = __builtin_fmaf(cf, sl[i % NS][c], t[i / 2][CS + c]);
                if (i + NS < 12) sl[i % NS] = W4_TRD(i + NS);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef W4_TRD
        }
    };
    // ---- one filter stage (2 of the group's 4 channel pairs; buffer ss): column transform + 12 MFMAs ----
    auto mfma_stage = [&](int ss, int fbuf, int next_stage, int halo_grp) __attribute__((always_inline)) {     // ss: channel pair of the group, fbuf: filter buffer; halo_grp: group to prefetch, < 0: none
        const int nbuf = fbuf ^ 1;
        float V[6][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {   // V[nu] = sum_j B^T[nu][j] t[j], scalar ops (see transform)
            const int c = 2 * ss + e;
            const float u0 = t[0][c], u1 = t[1][c], u2 = t[2][c], u3 = t[3][c], u4 = t[4][c], u5 = t[5][c];
            // points +-a share an even part (u4 - b2 u2) and an odd part (u3 - b2 u1), points +-b likewise with a2
            const float ea = __builtin_fmaf(-KB2, u2, u4), oa = __builtin_fmaf(-KB2, u1, u3);
            const float eb = __builtin_fmaf(-KA2, u2, u4), ob = __builtin_fmaf(-KA2, u1, u3);
            V[0][e] = __builtin_fmaf(KP, u0, __builtin_fmaf(KS, u2, u4));
            V[1][e] = __builtin_fmaf(KA, oa, ea);
            V[2][e] = __builtin_fmaf(-KA, oa, ea);
            V[3][e] = __builtin_fmaf(KB, ob, eb);
            V[4][e] = __builtin_fmaf(-KB, ob, eb);
            V[5][e] = __builtin_fmaf(KP, u1, __builtin_fmaf(KS, u3, u5));
        }
        f32x2 w2[6];
        const f32x4* Bp = Bw + fbuf * W4_BWS + lane;        // three 16-byte reads: the fragments of point pairs (0, 1), (2, 3), (4, 5)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x4 w4 = Bp[k * 64];
            w2[2 * k] = f32x2{w4[0], w4[1]};
            w2[2 * k + 1] = f32x2{w4[2], w4[3]};
        }
        // 12 MFMAs, channel-major: consecutive MFMAs hit different accumulators (dependency distance 6), so even a lone
        // wave keeps the matrix pipe full.  The next stage's three filter pieces go out one at a time behind MFMAs 2, 4
        // and 6 (pinned): the wave's issue slot is free while the pipe works, and the load path never sees a burst.
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                W4_MFMA(acc[v], V[v][e], w2[v][e]);
                if (e == 0 && (v == 1 || v == 3 || v == 5)) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (v == 1) dma_filter_piece(next_stage, nbuf, std::integral_constant<int, 0>{});
                    if (v == 3) dma_filter_piece(next_stage, nbuf, std::integral_constant<int, 1>{});
                    if (v == 5) dma_filter_piece(next_stage, nbuf, std::integral_constant<int, 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (e == 1 && (v == 1 || v == 3) && halo_grp >= 0) {       // behind MFMAs 8 and 10
                    __builtin_amdgcn_sched_barrier(0);
                    if (v == 1) dma_halo_piece(halo_grp, std::integral_constant<int, 0>{});
                    if (v == 3) dma_halo_piece(halo_grp, std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
#define W4_BARRIER() asm volatile("s_barrier" ::: "memory")

    W4_KSTAMP_BEGIN();
    // Phase rotation.  Per 8-channel group a wave has three phases: T (halo LDS reads + row transform, latency-bound),
    // S0 and S1 (12 MFMAs each).  The barrier would keep the three waves of a SIMD (w, w + 4, w + 8) in the same phase,
    // with the matrix pipe idle while all of them transform.  So the barrier sits at a different point of each wave's
    // phase sequence: class 0 (waves 0-3, also the halo loaders) runs T(g) S0(g) S1(g) after barrier g, class 1 runs
    // S1(g-1) T(g) S0(g), class 2 runs S0(g-1) S1(g-1) T(g): at any time one wave of a SIMD transforms while the other
    // two feed the matrix pipe.  t[] lives in registers across the barrier; every class executes ngroups barriers and
    // reads halo group g only between barriers g and g+1.
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
// T(g): raised priority for the few long-latency instructions of the transform; afterwards the MFMA phases run at a
// priority that orders the three waves of a SIMD (class 2 first): the wave that is latest in the rotation gets the pipe
// Priorities (round 4, A/B on one box, per-layer tables in gpurun_out/r04_ab1): the MATRIX phases run above the transform
// (transform 0, matrix phases 1 / 2 / 3 by rotation class): 1024 -> 512 at 32x32 13.83 -> 13.35 ms, 512 -> 512 6.92-7.11 -> 6.73-6.78,
// every other layer +-0.5 %; the transform is latency-bound on its LDS reads and loses nothing at priority 0, a ready MFMA
// no longer waits behind another wave's burst of 12 transform fmas.  (Rounds 1-3: transform at 3, matrix phases 0 / 1 / 2; no
// priorities at all: -6 %.)
#define W4_PT 0
#define W4_PS(PR) ((PR) + 1)
#define W4_T(g, PR) do { __builtin_amdgcn_s_setprio(W4_PT); transform(g, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}); __builtin_amdgcn_s_setprio(W4_PS(PR)); } while (0)
    // S0(g): filter stage 2g has landed (it is the youngest thing this wave issued) -> vmcnt(0); streams stage 2g+1 and
    // the halo of group g+2.  S1(g): only the two halo pieces issued after stage 2g+1 may still fly -> vmcnt(2).
#define W4_S0(g) do { W4_SB(); W4_WAIT(0); W4_SB(); mfma_stage(0, 0, 2 * (g) + 1, (g) + 2 < ngroups ? (g) + 2 : -1); } while (0)
#define W4_S1(g) do { W4_SB(); if ((g) + 2 < ngroups) W4_WAIT(2); else W4_WAIT(0); W4_SB(); \
                      mfma_stage(1, 1, (g) + 1 < ngroups ? 2 * (g) + 2 : 2 * (g), -1); } while (0)
// SPLIT: T(g) of the wave's two channels 2 ch, 2 ch + 1 only (round 4, in the slot-pipelined form; the column-by-column two-channel
// transform of round 2 made the register allocator spill ~100 registers); S(g) = its one filter stage of group g (stage 2 g + ch of the image, private
// buffer g & 1): everything this wave issued has landed (vmcnt(0): the filter of this group and its halo pieces of group
// g + 1); streams the filter of group g + 1 and the halo of group g + 2
#define W4_TS(g, PR) do { __builtin_amdgcn_s_setprio(W4_PT); \
                          transform(g, std::integral_constant<int, 2 * CH>{}, std::integral_constant<int, 2>{}); \
                          __builtin_amdgcn_s_setprio(W4_PS(PR)); } while (0)
#define W4_SS(g) do { W4_SB(); W4_WAIT(0); W4_SB(); \
                      mfma_stage(CH, (g) & 1, ((g) + 1 < ngroups ? 2 * (g) + 2 : 2 * (g)) + CH, (g) + 2 < ngroups ? (g) + 2 : -1); } while (0)
    const int cls = wave >> 2;
    dma_halo_piece(0, std::integral_constant<int, 0>{});
    dma_halo_piece(0, std::integral_constant<int, 1>{});
    if (ngroups > 1) {
        dma_halo_piece(1, std::integral_constant<int, 0>{});
        dma_halo_piece(1, std::integral_constant<int, 1>{});
    }
    dma_filter_piece(SPLIT ? ch : 0, 0, std::integral_constant<int, 0>{});
    dma_filter_piece(SPLIT ? ch : 0, 0, std::integral_constant<int, 1>{});
    dma_filter_piece(SPLIT ? ch : 0, 0, std::integral_constant<int, 2>{});
    if (ngroups > 1) W4_WAIT(5); else W4_WAIT(3);            // halo group 0 has landed (group 1: before barrier 1, below)
    if (SPLIT) {
        // two phases per group.  Waves 0-3 run T(g) S(g) after barrier g; waves 4-11 run S(g - 1) T(g): while one class
        // transforms, the other feeds the matrix pipe.  The late waves' halo pieces of group g + 1 go out in period g (inside
        // S(g - 1)) and are waited for before barrier g + 1.
        auto kloop = [&](auto chc) __attribute__((always_inline)) {      // (one copy of the loop per channel half: no branch inside)
            constexpr int CH = decltype(chc)::value;
            if (cls == 0) {
                for (int grp = 0; grp < ngroups; ++grp) {
                    W4_BARRIER();
                    W4_TS(grp, 0);
                    W4_SS(grp);
                }
            } else {
                W4_BARRIER();
                W4_TS(0, 1);
                for (int grp = 1; grp < ngroups; ++grp) {
                    W4_WAIT(0);
                    W4_BARRIER();
                    W4_SS(grp - 1);
                    W4_TS(grp, 1);
                }
                W4_SS(ngroups - 1);
            }
        };
        if (ch == 0) kloop(std::integral_constant<int, 0>{}); else kloop(std::integral_constant<int, 1>{});
    } else if (cls == 0) {
        for (int grp = 0; grp < ngroups; ++grp) {
            WSTAMP(0);
            W4_BARRIER();
            WSTAMP(1);
            W4_T(grp, 0);
            WSTAMP(2);
            W4_S0(grp);
            WSTAMP(3);
            W4_S1(grp);
            WSTAMP(4);
        }
    } else if (cls == 1) {
        W4_BARRIER();
        W4_T(0, 1);
        W4_S0(0);
        for (int grp = 1; grp < ngroups; ++grp) {
            WSTAMP(0);
            W4_BARRIER();
            WSTAMP(1);
            W4_S1(grp - 1);
            WSTAMP(2);
            W4_T(grp, 1);
            WSTAMP(3);
            W4_S0(grp);
            WSTAMP(4);
        }
        W4_S1(ngroups - 1);
    } else {
        W4_BARRIER();
        W4_T(0, 2);
        for (int grp = 1; grp < ngroups; ++grp) {
            W4_WAIT(3);                                      // own halo pieces of group grp + 0/1 landed (3 filter pieces may fly)
            WSTAMP(0);
            W4_BARRIER();
            WSTAMP(1);
            W4_S0(grp - 1);
            WSTAMP(2);
            W4_S1(grp - 1);
            WSTAMP(3);
            W4_T(grp, 2);
            WSTAMP(4);
        }
        W4_S0(ngroups - 1);
        W4_S1(ngroups - 1);
    }
#undef W4_T
#undef W4_S0
#undef W4_S1
#undef W4_TS
#undef W4_SS
#undef W4_SB
    W4_KSTAMP_DUMP();
#undef W4_BARRIER

    // ---- output stage: two passes (channel halves) through a [xi][x][tile][32 couts] exchange image ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the compiler does not see the asm LDS-DMAs
    float* Rs = reinterpret_cast<float*>(smem);
    const int Cout = p.out.c;
    W4_ESTAMP_BEGIN();
    // Work split of the combine step: wino4_combine.inc (1024 whole items per pass, a wave owns tile pairs).
    float hl[2][4][4];                                       // fused 1x1 head: partial logits [round][row of the tile][class]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) hl[a][b][c] = 0.f;
    // the bias quads of both passes are fetched here, under the K loop's drain and the first barrier: a global load inside
    // the combine step would queue behind the previous pass's output stores (one in-order vmcnt)
    f32x4 bvp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (p.bias != nullptr) {
        bvp[0] = *reinterpret_cast<const f32x4*>(p.bias + nb * 64 + 4 * (tid & 7));
        if (nb * 64 + 32 < Cout) bvp[1] = *reinterpret_cast<const f32x4*>(p.bias + nb * 64 + 32 + 4 * (tid & 7));
    }
    // fold the wave's own row (R = M[xi][:] A) into the exchange image; `add` (SPLIT, ch = 1): onto the partner's partial sums
    auto write_R = [&](auto add_c) __attribute__((always_inline)) {
        constexpr bool add = decltype(add_c)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = (e & 3) + 8 * (e >> 2) + 4 * lh;               // accumulator row = tile slot
            const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            float* o = Rs + (xi * 4) * W4_RPLANE + tl * 32 + li;
            const float r0 = m0 + s12 + s34, r1 = __builtin_fmaf(KA, d12, KB * d34), r2 = __builtin_fmaf(KA2, s12, KB2 * s34),
                        r3 = __builtin_fmaf(KA3, d12, __builtin_fmaf(KB3, d34, m5));
            if (add) {                                           // (this lane's four words: nobody else touches them in this phase)
                o[0 * W4_RPLANE] += r0; o[1 * W4_RPLANE] += r1; o[2 * W4_RPLANE] += r2; o[3 * W4_RPLANE] += r3;
            } else {
                o[0 * W4_RPLANE] = r0; o[1 * W4_RPLANE] = r1; o[2 * W4_RPLANE] = r2; o[3 * W4_RPLANE] = r3;
            }
        }
    };
    for (int pass = 0; pass < (SPLIT ? 1 : 2); ++pass) {
        __syncthreads();                                     // main-loop LDS reads / previous pass's combine are done
        ESTAMP(0);                                           // [0] barrier (K-loop skew / previous combine)
        if (SPLIT) {
            if (ch == 0) write_R(std::false_type{});
            __syncthreads();
            if (ch == 1) write_R(std::true_type{});
        } else if (ch == pass) {
            write_R(std::false_type{});
        }
        ESTAMP(1);                                           // [1] fold own row + write R to LDS
        __syncthreads();
        ESTAMP(2);                                           // [2] barrier
#include "wino4_combine.inc"
        ESTAMP(3);                                           // [3] combine + output stores issued
    }
#include "wino4_head.inc"
    W4_ESTAMP_DUMP();
}

// Eligibility beyond "3x3, stride 1, pad 1, same size" (checked by the caller): extents multiples of 16, input channels
// a multiple of 4 (>= 8), output channels a multiple of 32 (the filter image is zero padded to 8 / 64), 16-byte aligned views.
// The halo goes through a raw buffer descriptor with 32-bit lane offsets relative to the lower of a workgroup's two windows and
// 0x7fff0000 bytes of records: two neighbouring windows (whole launches) must fit that range - conv_wino4_span_ok(p, 2) - and so
// must the windows of one image that a region list can pair (cropped launches: conv_wino4_span_ok(p, per_image), checked by the
// caller before it hands over a list).  Beyond it the hardware would silently return zeros.
bool conv_wino4_span_ok(const ConvParams& p, int windows) {
    return (size_t)windows * p.in.h * p.in.w * (size_t)p.in.cs * 4 < 0x7fff0000ull;
}
bool conv_wino4_supported(const ConvParams& p) {
    return p.in.h == p.out.h && p.in.w == p.out.w && p.out.h % 16 == 0 && p.out.w % 16 == 0 && p.in.c % 4 == 0 &&
           p.in.c >= 8 && p.out.c % 32 == 0 && p.in.cs % 4 == 0 && p.out.cs % 4 == 0 && p.zero != nullptr && conv_wino4_span_ok(p, 2);
}

hipError_t launch_conv_wino4(const ConvParams& p, hipStream_t s) {
    const int regs_x = p.out.w / 16, regs_y = p.out.h / 16;
    const size_t nreg = p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * regs_x * regs_y;
    const size_t npairs = (nreg + 1) / 2;
    const size_t grid = npairs * (size_t)((p.out.c + 63) / 64);
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull || !conv_wino4_span_ok(p, p.lut != nullptr ? p.per_image : 2)) return hipErrorInvalidValue;
    if (p.head_w != nullptr && (!p.head_only || p.pool.p != nullptr)) return hipErrorInvalidValue;     // (the HEAD kernels write neither the features nor a pool)
    size_t lds = (size_t)(3 * W4_HS + 12 * 2 * W4_BWS) * 16;
    const size_t lds_epi = (size_t)24 * W4_RPLANE * 4;
    if (lds_epi > lds) lds = lds_epi;
    void (*kern)(ConvParams, int, int, int) = conv_wino4_kernel<false, false>;
    if (p.head_w != nullptr) kern = conv_wino4_kernel<true, false>;
    else if (p.out.c == 32 && p.w4_split) kern = conv_wino4_kernel<false, true>;   // a lone 32-channel block: split K
    W4_DIAG_SELECT(kern, p, lds);
    static DeviceOnce attr_set[3];                          // the attribute is per device
    const int which = p.head_w != nullptr ? 1 : (p.out.c == 32 && p.w4_split) ? 2 : 0;
    const hipError_t ea = attr_set[which].run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(768), lds, s, p, regs_x, regs_y, (int)npairs);
    return hipGetLastError();
}

}  // namespace ecseg
