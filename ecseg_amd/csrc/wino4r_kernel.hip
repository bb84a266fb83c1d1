// Winograd F(4x4, 3x3) on the fp32 matrix cores with the ROW transform done once per workgroup (round 6): conv_wino4_kernel's arithmetic -
// the same filter image, the same fma chains in the row and column transforms, MFMA order and output stage: results are BIT-IDENTICAL
// (tools/w4r_time.py) - in the loop structure of conv_wino4s_kernel (wino4s_kernel.hip): per 8-channel group all 768 threads
// turn the raw halo into a t image in LDS (row_pass: 576 half items, 24 fmas each), a wave reads the six 16-byte columns of ITS row instead
// of 24 raw slots + 72 fmas; two barriers per group, raw halo double-buffered and fetched two groups ahead.  See DESIGN.md 5.1 for what
// it measures against the kernel it came from.  (A SPLIT variant for lone 32-channel blocks was built and measured 2 - 6 % SLOWER than
// conv_wino4_kernel<false, true>, which keeps those layers: tools/experiments/wino4r_split_variant.patch.)
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_util.h"

namespace ecseg {

#include "wino4_consts.inc"
#define W4_HALO_RING 2
// raw image (only row_pass reads it): plain row / column order, the two 16-byte channel halves of a pixel next to each other
#define W4_HALO_SLOT(r, cc) const int h = (cc) & 1, hy = (r), hx = (cc) >> 1
#define W4_HALO_UPPER(cc) ((cc) & 1)
// (timing-only ablations exist only in A/B builds, tools/w4r_variants.sh -DECSEG_W4R_ABL=<bits>: 1 no filter DMA, 2 no halo DMA, 4 no MFMAs)
#ifdef ECSEG_W4R_ABL
#define W4_DIAG_SKIP_HALO_DMA() do { if (ECSEG_W4R_ABL & 2) return; } while (0)
#define W4_DIAG_SKIP_FILTER_DMA() do { if (ECSEG_W4R_ABL & 1) return; } while (0)
#else
#define W4_DIAG_SKIP_HALO_DMA()
#define W4_DIAG_SKIP_FILTER_DMA()
#endif
#define W4_DIAG_HALO_OFFSET(off, a)
#if defined(ECSEG_W4R_ABL) && (ECSEG_W4R_ABL & 4)
#define W4_MFMA(ACC, A, B) asm volatile("" :: "v"(A), "v"(B))
#else
#define W4_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0)
#endif
#define W4_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define ESTAMP(i)
#define W4_ESTAMP_BEGIN()
#define W4_ESTAMP_DUMP()

namespace {
constexpr int W4R_TS = 2 * 6 * 4 * 36;   // slots of the t image: 2 regions x 6 transform rows x 4 tile rows x (18 columns x 2 channel halves)
}

template <bool HEAD>
__global__ __launch_bounds__(768) void conv_wino4r_kernel(ConvParams p, int regs_x, int regs_y, int npairs) {
    constexpr bool SPLIT = false;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [2][W4_HS]        raw halo (group g -> buffer g & 1)
    f32x4* Ts = Hs + 2 * W4_HS;                              // [W4R_TS]          row-transformed halo of one group
    f32x4* Bs = Ts + W4R_TS;                                 // [12][2][W4_BWS]   per-wave filter stages

    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
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
        const unsigned dst = lds_base + (unsigned)(2 * W4_HS + W4R_TS + (wave * 2 + buf) * W4_BWS) * 16u;
        const unsigned l16 = lane16;
        unsigned keep;
        // the instruction offset advances the global AND the LDS address: one M0 for the three pieces
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(dst), "s"(g), "n"(k * 1024) : "memory");
    };


    const int q8 = li >> 2, tx = li & 3;
    const int tg = (0x96 >> q8) & 1;
    const int ty = (q8 == 0 || q8 == 1) ? 0 : (q8 == 2 || q8 == 3) ? 1 : (q8 == 4 || q8 == 5) ? 2 : 3;
    const int t_lane = ((tg * 6 + xi) * 4 + ty) * 36 + lh * 18 + tx;      // the lane's tile in the t image (wino4s_kernel.hip)

    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;

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
    f32x4 t[6];
    auto load_t = [&]() __attribute__((always_inline)) {
        const f32x4* A = Ts + t_lane;
#pragma unroll
        for (int j = 0; j < 6; ++j) t[j] = A[w4_cpos(j)];
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
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
    // Per group g: barrier Y(g-1) (the t image holds group g, raw buffer g & 1 is free: this wave's two halo pieces of group g + 2 go out) and
    // barrier X(g) (nobody needs the t image of group g any more, the raw halo of group g + 1 has landed: row_pass(g + 1) follows).  The
    // three waves of a SIMD sit at different points of the sequence:
    //   class 0:   Y(g-1) | t <- image, S0(g)        | X(g) | row_pass(g+1), S1(g)
    //   class 1:   Y(g-1) | S1(g-1), t <- image      | X(g) | row_pass(g+1), S0(g)
    //   class 2:   Y(g-1) | t <- image, S0(g)        | X(g) | S1(g), row_pass(g+1)
    // Filter stage s = 2 g + ss lives in buffer ss and is streamed one stage ahead behind the MFMAs of the stage before (as in
    // conv_wino4_kernel).  Waits: the phase right behind Y has the two halo pieces issued behind Y younger than its stage -> vmcnt(2);
    // the other phase's stage is the youngest thing the wave issued -> vmcnt(0).  (Two stages ahead, as the split kernel streams: -1 %,
    // tools/experiments/wino4r_filter_two_stages_ahead.patch; s_setprio by rotation class as in conv_wino4_kernel: +-0.)
#define W4_S(ss, g, H) do { W4_SB(); if (H) W4_WAIT(2); else W4_WAIT(0); W4_SB(); \
                            mfma_stage(ss, ss, (ss) == 0 ? 2 * (g) + 1 : ((g) + 1 < ngroups ? 2 * (g) + 2 : 2 * (g)), -1); W4_SB(); } while (0)
#define W4_Y() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_BARRIER(); W4_SB(); } while (0)
#define W4_X() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_BARRIER(); W4_SB(); } while (0)
#define W4_HALO(g) do { dma_halo_piece((g), std::integral_constant<int, 0>{}); dma_halo_piece((g), std::integral_constant<int, 1>{}); } while (0)
    const int cls = wave >> 2;
    W4_HALO(0);
    if (ngroups > 1) W4_HALO(1);
    dma_filter_piece(0, 0, std::integral_constant<int, 0>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 1>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 2>{});
    if (ngroups > 1) W4_WAIT(5); else W4_WAIT(3);            // raw group 0 has landed
    W4_BARRIER();
    row_pass(0);
    if (cls == 0) {
        for (int grp = 0; grp < ngroups; ++grp) {
            W4_Y();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            load_t();
            W4_S(0, grp, mh);
            if (grp + 1 < ngroups) { W4_X(); row_pass(grp + 1); W4_SB(); }
            W4_S(1, grp, false);
        }
    } else if (cls == 1) {
        {
            W4_Y();
            const bool mh = 2 < ngroups;
            if (mh) W4_HALO(2);
            load_t();
            if (1 < ngroups) {
                if (mh) W4_WAIT(5); else W4_WAIT(3);         // raw group 1 has landed (this class has not waited for anything since the prologue)
                W4_X(); row_pass(1); W4_SB();
            }
            W4_S(0, 0, false);
        }
        for (int grp = 1; grp < ngroups; ++grp) {
            W4_Y();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            W4_S(1, grp - 1, mh);
            load_t();
            if (grp + 1 < ngroups) { W4_X(); row_pass(grp + 1); W4_SB(); }
            W4_S(0, grp, false);
        }
        W4_S(1, ngroups - 1, false);
    } else {
        for (int grp = 0; grp < ngroups; ++grp) {
            W4_Y();
            const bool mh = grp + 2 < ngroups;
            if (mh) W4_HALO(grp + 2);
            load_t();
            W4_S(0, grp, mh);
            if (grp + 1 < ngroups) { W4_X(); }
            W4_S(1, grp, false);
            if (grp + 1 < ngroups) { row_pass(grp + 1); W4_SB(); }
        }
    }
#undef W4_S
#undef W4_Y
#undef W4_X
#undef W4_HALO
#undef W4_SB
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
    if constexpr (!HEAD && !SPLIT) {
        // Passes by TILE half instead of channel half (round 6): pass P takes the accumulator rows e = 8 P .. 8 P + 7 (tiles 16 P .. 16 P + 15) of ALL
        // 64 output channels, exchange image [xi][x][16 tiles][64 couts] - every wave folds and writes in BOTH passes (half as many rows each)
        // where the channel-half passes left six of the twelve waves waiting behind the other six's fold.  The same sums element by element:
        // bit-identical.  (Not for a fused head: its per-pixel class sums would then be spread over two waves.)
        const int cq = tid & 7, cx = (tid >> 3) & 3, nlo = (tid >> 5) & 1;          // channel quad, column of the tile, tile of the pair
        const int cwave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const float* rlane = Rs + cx * W4_RPLANE + nlo * 64 + 4 * cq;
        const unsigned out_row = (unsigned)(W * p.out.cs);
        auto tile_pass = [&](auto pc) __attribute__((always_inline)) {
            constexpr int P = decltype(pc)::value;
            __syncthreads();                                 // main-loop LDS reads / previous pass's combine are done
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * P + e8;
                const int tl = (e8 & 3) + 8 * (e8 >> 2) + 4 * lh;         // tile slot within the half: 0..15
                const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                float* o = Rs + (xi * 4) * W4_RPLANE + tl * 64 + ch * 32 + li;
                const float r0 = m0 + s12 + s34, r1 = __builtin_fmaf(KA, d12, KB * d34), r2 = __builtin_fmaf(KA2, s12, KB2 * s34),
                            r3 = __builtin_fmaf(KA3, d12, __builtin_fmaf(KB3, d34, m5));
                o[0 * W4_RPLANE] = r0; o[1 * W4_RPLANE] = r1; o[2 * W4_RPLANE] = r2; o[3 * W4_RPLANE] = r3;
            }
            __syncthreads();
            if (P == 0) asm volatile("" :: "v"(bvp[0]), "v"(bvp[1]));      // (the bias wait in straight-line code: wino4_combine.inc)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = cwave + 12 * k;                // unit: tile pair u & 7 of this half x channel half u >> 3
                if (u >= 16) break;
                const int tp = u & 7, chh = u >> 3;
                const int nq8 = 4 * P + (tp >> 1);           // tile row pair index of the 32 tiles
                const int g = (0x96 >> nq8) & 1, nty = nq8 >> 1;
                const int img = g ? r_img[1] : r_img[0];
                if (img < 0) continue;
                const int co = nb * 64 + chh * 32 + 4 * cq;
                if (co + 3 >= Cout) continue;                // (the missing channel half of a Cout % 64 == 32 block; wave-uniform per lane group: co_ok is all or nothing for chh)
                const f32x4 bv = chh ? bvp[1] : bvp[0];
                const float* r = rlane + tp * 128 + chh * 32;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(r);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(r + 1 * 4 * W4_RPLANE);
                const f32x4 q2 = *reinterpret_cast<const f32x4*>(r + 2 * 4 * W4_RPLANE);
                const f32x4 q3 = *reinterpret_cast<const f32x4*>(r + 3 * 4 * W4_RPLANE);
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(r + 4 * 4 * W4_RPLANE);
                const f32x4 q5 = *reinterpret_cast<const f32x4*>(r + 5 * 4 * W4_RPLANE);
                const f32x4 s12 = q1 + q2, d12 = q1 - q2, s34 = q3 + q4, d34 = q3 - q4;
                f32x4 y[4];                                  // (wino4_combine.inc's expressions, term for term)
                y[0] = q0 + s12 + s34 + bv;
                y[1] = KA * d12 + KB * d34 + bv;
                y[2] = KA2 * s12 + KB2 * s34 + bv;
                y[3] = KA3 * d12 + KB3 * d34 + q5 + bv;
                if (p.act == ECSEG_ACT_RELU) {
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float tv = y[yy][c];
                            asm("v_max_f32_e32 %0, 0, %1" : "=v"(tv) : "v"(tv));
                            y[yy][c] = tv;
                        }
                } else if (p.act != ECSEG_ACT_LINEAR) {
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[yy][c] = apply_act_core(y[yy][c], p.act, p.alpha);
                }
                const int oy = (g ? r_y0[1] : r_y0[0]) + 4 * nty, ox = (g ? r_x0[1] : r_x0[0]) + 8 * (tp & 1);       // wave-uniform
                const unsigned out_lane = (unsigned)((4 * nlo + cx) * p.out.cs + co);
                float* ob = p.out.p + (((size_t)img * H + oy) * W + ox) * p.out.cs;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy)
                    __builtin_nontemporal_store(y[yy], reinterpret_cast<f32x4*>(ob + (size_t)(out_lane + (unsigned)yy * out_row)));
                if (p.pool.p != nullptr) {                  // fused MaxPooling2D(2x2, stride 2), as in wino4_combine.inc
                    float* pb = p.pool.p + (((size_t)img * p.pool.h + (oy >> 1)) * p.pool.w + (ox >> 1)) * p.pool.cs;
                    const unsigned pool_lane = (unsigned)((2 * nlo + (cx >> 1)) * p.pool.cs + co);
#pragma unroll
                    for (int yp = 0; yp < 2; ++yp) {
                        f32x4 m;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float a = fmaxf(y[2 * yp][c], y[2 * yp + 1][c]);
                            m[c] = fmaxf(a, __shfl_xor(a, 8));
                        }
                        if (!(cx & 1)) *reinterpret_cast<f32x4*>(pb + (size_t)(pool_lane + (unsigned)(yp * p.pool.w * p.pool.cs))) = m;
                    }
                }
            }
        };
        tile_pass(std::integral_constant<int, 0>{});
        tile_pass(std::integral_constant<int, 1>{});
    } else
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


bool conv_wino4r_supported(const ConvParams& p) { return conv_wino4_supported(p) && !(p.out.c == 32 && p.w4_split); }

hipError_t launch_conv_wino4r(const ConvParams& p, hipStream_t s) {
    const int regs_x = p.out.w / 16, regs_y = p.out.h / 16;
    const size_t nreg = p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * regs_x * regs_y;
    const size_t npairs = (nreg + 1) / 2;
    const size_t grid = npairs * (size_t)((p.out.c + 63) / 64);
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull || !conv_wino4_span_ok(p, p.lut != nullptr ? p.per_image : 2)) return hipErrorInvalidValue;
    if (p.head_w != nullptr && (!p.head_only || p.pool.p != nullptr)) return hipErrorInvalidValue;
    size_t lds = (size_t)(2 * W4_HS + W4R_TS + 12 * 2 * W4_BWS) * 16;
    const size_t lds_epi = (size_t)24 * W4_RPLANE * 4;
    if (lds_epi > lds) lds = lds_epi;
    void (*kern)(ConvParams, int, int, int) = p.head_w != nullptr ? conv_wino4r_kernel<true> : conv_wino4r_kernel<false>;
    static DeviceOnce attr_set[2];
    const hipError_t ea = attr_set[p.head_w != nullptr ? 1 : 0].run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(768), lds, s, p, regs_x, regs_y, (int)npairs);
    return hipGetLastError();
}

}  // namespace ecseg
