// Winograd F(4x4, 3x3) on the fp32 matrix cores, "half workgroup" form: TWO independent 6-wave workgroups per CU.
//
// Same arithmetic as wino4_kernel.hip (36 multiplies per 4x4 output block, exact-f32 v_mfma_f32_32x32x2_f32, filter
// transform U = G g G^T in float64 at model load), different work split:
//
//   wino4_kernel:  workgroup = 12 waves = 2 regions x 64 output channels, 156 KB of LDS -> ONE workgroup per CU; nothing
//                  overlaps its prologue, its barrier skew or its output stage (15 - 25 % of a short-K workgroup's life).
//   this kernel:   workgroup = 6 waves (transform rows xi) = 2 regions (32 tiles) x 32 output channels, 78 KB of LDS ->
//                  TWO workgroups per CU that share nothing and drift apart: one's output stage, prologue and barrier
//                  waits run under the other's MFMAs.  The three waves of a SIMD come from both workgroups, which
//                  gives the de-phasing that wino4_kernel builds by hand (phase rotation).
//
// K loop in stages of 4 input channels (one MFMA K pair per lane half): per stage and wave 24 ds_read_b64 of raw halo
// pixels (all issued at once: with two channels per lane instead of four the row transform needs 12 registers, not
// 24, so the reads of all six columns can be in flight together - one LDS round trip per stage instead of six),
// 36 + 26 scalar FMAs / adds (row + column transform), 6 ds_read_b64 of filter fragments, 12 MFMAs.  Halo (3-deep
// ring, 12 KB per stage) and the wave-private filter stages (double buffer) arrive by LDS-DMA issued one piece at a
// time behind MFMAs, ordered by counted vmcnt waits; ONE s_barrier per stage.
//
// LDS halo image: 16-byte slot = the stage's 4 channels of one pixel; slot(g, y, x) = (g * 18 + P(y)) * 20 + P(x) with
// P(v) = {0, 5, 10, 14}[v % 4] + v / 4 (rows / columns regrouped by their phase modulo the tile stride): the 16 lanes
// of a read phase hit slots 20 ty + tx + const, 20 = 4 (mod 16) -> distinct bank groups, no padding, no conflicts.
// Lane halves read the two 8-byte halves of a slot (channels 0,1 / 2,3 = the two K values of the MFMA).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_util.h"

namespace ecseg {

namespace {

constexpr int WH_HS = 768;           // halo slots per buffer: 2 regions x 18 rows x 20 = 720 used
constexpr int WH_BWS = 192;          // filter slots per wave and stage: 6 points x 2 halves x 32 couts x 2 k / 4
constexpr int WH_RPLANE = 544;       // floats per (xi, x) plane of the output exchange image: 16 tiles x 32 couts + 32

__device__ __forceinline__ int wh_pos(int v) { const int m = v & 3; return (m == 0 ? 0 : m == 1 ? 5 : m == 2 ? 10 : 14) + (v >> 2); }
__device__ __forceinline__ int wh_inv(int r) { return r < 5 ? 4 * r : r < 10 ? 4 * (r - 5) + 1 : r < 14 ? 4 * (r - 10) + 2 : 4 * (r - 14) + 3; }
constexpr int wh_cpos(int v) { return ((v & 3) == 0 ? 0 : (v & 3) == 1 ? 5 : (v & 3) == 2 ? 10 : 14) + (v >> 2); }

typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA piece: 64 lanes x 16 bytes, global (per-lane address) -> LDS bytes [lds_dst + 16 * lane] (see wino4_kernel.hip)
template <int OFF>
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst), "n"(OFF) : "memory");
}

}  // namespace

__global__ __launch_bounds__(384, 3) void conv_wino4h_kernel(ConvParams p, int regs_x, int regs_y, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [3][WH_HS]       halo ring (stage s -> buffer s % 3)
    f32x4* Bs = Hs + 3 * WH_HS;                              // [6][2][WH_BWS]   per-wave filter stages

    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = transform row
    const int li = lane & 31, lh = lane >> 5;

    // channel half fastest: the two workgroups that share a region pair are dispatched next to each other (same XCD / L2)
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ch = (int)(bid & 1u);
    const unsigned rest = bid >> 1;
    const int pair = (int)(rest % (unsigned)npairs), nb = (int)(rest / (unsigned)npairs);
    const int Cout = p.out.c;
    const int n0 = nb * 64 + ch * 32;                        // first output channel of this workgroup
    if (n0 >= Cout) return;                                  // Cout % 64 == 32: the last block has one half only
    const int H = p.in.h, W = p.in.w;
    const int nstages = (p.in.c + 3) >> 2;                   // 4 input channels each

    int r_img[2], r_y0[2], r_x0[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int rid = 2 * pair + g;
        int rx, ry, img;
        if (p.lut != nullptr) {                              // cropped launch: the regions some later stage reads
            const int i = rid / p.lut_len, v = p.lut[rid - i * p.lut_len];
            img = i * p.per_image + (v >> 16); ry = (v >> 8) & 255; rx = v & 255;      // origins in 4-pixel tiles
            if (i >= p.n / p.per_image) img = p.n;
            r_img[g] = img < p.n ? img : -1;
            r_y0[g] = ry * 4; r_x0[g] = rx * 4;
            continue;
        } else {
            rx = rid % regs_x;
            const int t = rid / regs_x;
            ry = t % regs_y; img = t / regs_y;
        }
        r_img[g] = img < p.n ? img : -1;
        r_y0[g] = ry * 16; r_x0[g] = rx * 16;
    }

    // ---- halo DMA: wave w fills slots 64 k + lane for k = w and w + 6; the per-lane source pointers are computed once
    //      and parked in LDS (bit 0 = "advance with the stage"; padding / out-of-image lanes read the zero page) ----
    unsigned long long* Hd = reinterpret_cast<unsigned long long*>(Bs + 6 * 2 * WH_BWS) + tid;   // [2][384]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a = 64 * (xi + 6 * i) + lane;
        unsigned long long d = (unsigned long long)(size_t)p.zero;
        if (a < 720) {
            const int g = a >= 360 ? 1 : 0, rem = a - g * 360;
            const int r = rem / 20, c = rem - r * 20;
            if (c < 18) {
                const int img = g ? r_img[1] : r_img[0];
                const int iy = (g ? r_y0[1] : r_y0[0]) - 1 + wh_inv(r), ix = (g ? r_x0[1] : r_x0[0]) - 1 + wh_inv(c);
                if (img >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W)
                    d = (unsigned long long)(size_t)(p.in.p + (((size_t)img * H + iy) * W + ix) * p.in.cs) | 1ull;
            }
        }
        Hd[i * 384] = d;
    }
    auto dma_halo_piece = [&](int stage, auto ii) {
        constexpr int i = decltype(ii)::value;
        const unsigned long long d = Hd[i * 384];
        const float* src = reinterpret_cast<const float*>((size_t)(d & ~1ull)) + (d & 1ull ? stage * 4 : 0);
        glds16<0>(src, lds_base + (unsigned)((stage % 3) * WH_HS + 64 * (xi + 6 * i)) * 16u);
    };
    // ---- filter DMA: wt4h[nb][stage][ch * 6 + xi][point nu][h][cout 32][k 2], 768 floats per wave and stage ----
    const float* w_src = p.wt + ((size_t)nb * nstages * 12 + ch * 6 + xi) * 768 + lane * 4;
    f32x4* Bw = Bs + xi * 2 * WH_BWS;
    auto dma_filter_piece = [&](int stage, int buf, auto kk) {
        constexpr int k = decltype(kk)::value;
        const float* g = w_src + (size_t)stage * (12 * 768);
        glds16<k * 1024>(g, lds_base + (unsigned)(3 * WH_HS + (xi * 2 + buf) * WH_BWS) * 16u);
    };

    // ---- A-operand lane -> tile (same lane groups as wino4_kernel: each ds_read phase sees the 16 tiles of one region) ----
    const int q8 = li >> 2, tx = li & 3;
    const int tg = (0x96 >> q8) & 1;
    const int ty = (q8 == 0 || q8 == 1) ? 0 : (q8 == 2 || q8 == 3) ? 1 : (q8 == 4 || q8 == 5) ? 2 : 3;
    const int a_lane = 2 * ((tg * 18 + ty) * 20 + tx) + lh;  // in 8-byte units: slot of halo pixel (4 ty, 4 tx), lane half

    int rr0, rr1, rr2, rr3; float c0, c1, c2;
    switch (xi) {                                            // row xi of B^T: t = c0 d[r0] + c1 d[r1] + c2 d[r2] (+ d[r3])
        case 0:  rr0 = 0; rr1 = 2; rr2 = 4; rr3 = 4; c0 = 4.f;  c1 = -5.f; c2 = 1.f;  break;
        case 1:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -4.f; c1 = -4.f; c2 = 1.f;  break;
        case 2:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = 4.f;  c1 = -4.f; c2 = -1.f; break;
        case 3:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -2.f; c1 = -1.f; c2 = 2.f;  break;
        case 4:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = 2.f;  c1 = -1.f; c2 = -2.f; break;
        default: rr0 = 1; rr1 = 3; rr2 = 5; rr3 = 5; c0 = 4.f;  c1 = -5.f; c2 = 1.f;  break;
    }
    const int ro0 = 40 * wh_pos(rr0), ro1 = 40 * wh_pos(rr1), ro2 = 40 * wh_pos(rr2), ro3 = 40 * wh_pos(rr3);   // 8-byte units
    const bool inner_row = xi >= 1 && xi <= 4;

    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;

#define WH_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WH_BARRIER() asm volatile("s_barrier" ::: "memory")
#define WH_SB() __builtin_amdgcn_sched_barrier(0)

    dma_halo_piece(0, std::integral_constant<int, 0>{});
    dma_halo_piece(0, std::integral_constant<int, 1>{});
    if (nstages > 1) {
        dma_halo_piece(1, std::integral_constant<int, 0>{});
        dma_halo_piece(1, std::integral_constant<int, 1>{});
    }
    dma_filter_piece(0, 0, std::integral_constant<int, 0>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 1>{});
    dma_filter_piece(0, 0, std::integral_constant<int, 2>{});
    if (nstages > 1) WH_WAIT(5); else WH_WAIT(3);            // own pieces of halo stage 0 have landed

    for (int s = 0; s < nstages; ++s) {
        WH_BARRIER();                                        // halo stage s complete and visible; buffer (s + 2) % 3 is free
        // ---- row transform: t[j] = B^T[xi, :] d[:, j] for the six halo columns of the lane's tile, 2 channels ----
        const f32x2* A2 = reinterpret_cast<const f32x2*>(Hs + (s % 3) * WH_HS) + a_lane;
        constexpr int cp[6] = {2 * wh_cpos(0), 2 * wh_cpos(1), 2 * wh_cpos(2), 2 * wh_cpos(3), 2 * wh_cpos(4), 2 * wh_cpos(5)};
        f32x2 t[6];
        __builtin_amdgcn_s_setprio(3);
        if (inner_row) {
            f32x2 d0[6], d1[6], d2[6], d3[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { d0[j] = A2[ro0 + cp[j]]; d1[j] = A2[ro1 + cp[j]]; d2[j] = A2[ro2 + cp[j]]; d3[j] = A2[ro3 + cp[j]]; }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    t[j][c] = __builtin_fmaf(c0, d0[j][c], __builtin_fmaf(c1, d1[j][c], __builtin_fmaf(c2, d2[j][c], d3[j][c])));
        } else {
            f32x2 d0[6], d1[6], d2[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { d0[j] = A2[ro0 + cp[j]]; d1[j] = A2[ro1 + cp[j]]; d2[j] = A2[ro2 + cp[j]]; }
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c) t[j][c] = __builtin_fmaf(c0, d0[j][c], __builtin_fmaf(c1, d1[j][c], d2[j][c]));
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- column transform: V[nu] = sum_j B^T[nu][j] t[j] (scalar ops: packed f32 VALU stalls beside MFMAs) ----
        float V[6][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float u0 = t[0][e], u1 = t[1][e], u2 = t[2][e], u3 = t[3][e], u4 = t[4][e], u5 = t[5][e];
            const float a42 = __builtin_fmaf(-4.f, u2, u4), a31 = __builtin_fmaf(-4.f, u1, u3);
            const float b42 = u4 - u2, b31 = u3 - u1;
            V[0][e] = __builtin_fmaf(4.f, u0, __builtin_fmaf(-5.f, u2, u4));
            V[1][e] = a42 + a31;
            V[2][e] = a42 - a31;
            V[3][e] = __builtin_fmaf(2.f, b31, b42);
            V[4][e] = __builtin_fmaf(-2.f, b31, b42);
            V[5][e] = __builtin_fmaf(4.f, u1, __builtin_fmaf(-5.f, u3, u5));
        }
        // ---- filter stage s has landed: only the two halo pieces issued after it (stage s + 1's) may still fly ----
        WH_SB();
        if (s >= 1 && s + 1 < nstages) WH_WAIT(2); else WH_WAIT(0);
        WH_SB();
        const f32x2* Bp = reinterpret_cast<const f32x2*>(Bw + (s & 1) * WH_BWS) + lane;
        f32x2 w2[6];
#pragma unroll
        for (int v = 0; v < 6; ++v) w2[v] = Bp[v * 64];
        const bool more_f = s + 1 < nstages, more_h = s + 2 < nstages;
        // 12 MFMAs, channel-major (dependency distance 6); the next filter stage's three pieces go out behind MFMAs 2, 4, 6,
        // the halo pieces of stage s + 2 behind MFMAs 8 and 10 - one at a time, never a burst on the load path
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v][e], w2[v][e], acc[v], 0, 0, 0);
                if (e == 0 && (v == 1 || v == 3 || v == 5) && more_f) {
                    WH_SB();
                    if (v == 1) dma_filter_piece(s + 1, (s + 1) & 1, std::integral_constant<int, 0>{});
                    if (v == 3) dma_filter_piece(s + 1, (s + 1) & 1, std::integral_constant<int, 1>{});
                    if (v == 5) dma_filter_piece(s + 1, (s + 1) & 1, std::integral_constant<int, 2>{});
                    WH_SB();
                }
                if (e == 1 && (v == 1 || v == 3) && more_h) {
                    WH_SB();
                    if (v == 1) dma_halo_piece(s + 2, std::integral_constant<int, 0>{});
                    if (v == 3) dma_halo_piece(s + 2, std::integral_constant<int, 1>{});
                    WH_SB();
                }
            }
        // own halo pieces of stage s + 1 (issued during stage s - 1 / the prologue) must have landed before the next
        // barrier: behind them at most filter stage s + 1 (3) and halo stage s + 2 (2) were issued
        WH_SB();
        if (more_h) WH_WAIT(5); else WH_WAIT(0);
        WH_SB();
    }
#undef WH_WAIT
#undef WH_BARRIER
#undef WH_SB

    // ---- output stage: two passes (tile halves) through a [xi][x][16 tiles][32 couts] exchange image ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the compiler does not see the asm LDS-DMAs
    float* Rs = reinterpret_cast<float*>(smem);
    const int q = tid & 7;                                   // (384 % 8 == 0: the channel quad of a thread is fixed)
    const int co = n0 + 4 * q;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && co + 3 < Cout) bv = *reinterpret_cast<const f32x4*>(p.bias + co);
    for (int th = 0; th < 2; ++th) {
        __syncthreads();                                     // K-loop LDS reads / previous pass's combine are done
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const int e = 8 * th + e8;                       // accumulator rows of tiles 16 th .. 16 th + 15
            const int tl = (e8 & 3) + 8 * (e8 >> 2) + 4 * lh;             // tile slot inside the half
            const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            float* o = Rs + (xi * 4) * WH_RPLANE + tl * 32 + li;
            o[0 * WH_RPLANE] = m0 + s12 + s34;
            o[1 * WH_RPLANE] = d12 + 2.f * d34;
            o[2 * WH_RPLANE] = s12 + 4.f * s34;
            o[3 * WH_RPLANE] = d12 + 8.f * d34 + m5;
        }
        __syncthreads();
        // combine: 1024 half items (channel quad q, column x, tile, row pair yh) over 384 threads
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int item = tid + k * 384;
            if (item >= 1024) break;
            const int x = (item >> 3) & 3, nloc = (item >> 5) & 15, yh = item >> 9;
            const int n = 16 * th + nloc;
            const int nq8 = n >> 2, ntx = n & 3;
            const int g = (0x96 >> nq8) & 1;
            const int nty = nq8 < 2 ? 0 : nq8 < 4 ? 1 : nq8 < 6 ? 2 : 3;
            const int img = g ? r_img[1] : r_img[0];
            if (img < 0) continue;
            const float* r = Rs + x * WH_RPLANE + nloc * 32 + 4 * q;
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(r + 1 * 4 * WH_RPLANE);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(r + 2 * 4 * WH_RPLANE);
            const f32x4 q3 = *reinterpret_cast<const f32x4*>(r + 3 * 4 * WH_RPLANE);
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(r + 4 * 4 * WH_RPLANE);
            const f32x4 qe = *reinterpret_cast<const f32x4*>(r + (yh ? 5 : 0) * 4 * WH_RPLANE);
            const f32x4 s12 = q1 + q2, d12 = q1 - q2, s34 = q3 + q4, d34 = q3 - q4;
            f32x4 y[2];
            if (yh == 0) {
                y[0] = qe + s12 + s34 + bv;
                y[1] = d12 + 2.f * d34 + bv;
            } else {
                y[0] = s12 + 4.f * s34 + bv;
                y[1] = d12 + 8.f * d34 + qe + bv;
            }
            const int oy = (g ? r_y0[1] : r_y0[0]) + 4 * nty + 2 * yh, ox = (g ? r_x0[1] : r_x0[0]) + 4 * ntx + x;
            float* o = p.out.p + (((size_t)img * H + oy) * W + ox) * p.out.cs + co;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                y[yy] = apply_act4(y[yy], p.act, p.alpha);
                if (co + 3 < Cout) *reinterpret_cast<f32x4*>(o + (size_t)yy * W * p.out.cs) = y[yy];
            }
            if (p.pool.p != nullptr) {
                // fused MaxPooling2D(2x2, stride 2): the row pair is in registers, the column partner (x ^ 1) is lane ^ 8 of
                // the same tile, hence of the same region: it is active whenever this lane is
                f32x4 m;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = fmaxf(y[0][c], y[1][c]);
                    m[c] = fmaxf(a, __shfl_xor(a, 8));
                }
                if (!(x & 1) && co + 3 < Cout)
                    *reinterpret_cast<f32x4*>(p.pool.p + (((size_t)img * p.pool.h + (oy >> 1)) * p.pool.w + (ox >> 1)) * p.pool.cs + co) = m;
            }
        }
    }
}

// Same eligibility as conv_wino4_supported, but Cin % 4 == 0 needs no tail handling here (stages of 4 channels).
bool conv_wino4h_supported(const ConvParams& p) {
    return p.in.h == p.out.h && p.in.w == p.out.w && p.out.h % 16 == 0 && p.out.w % 16 == 0 && p.in.c % 4 == 0 &&
           p.in.c >= 8 && p.out.c % 32 == 0 && p.in.cs % 4 == 0 && p.out.cs % 4 == 0 && p.zero != nullptr && p.head_w == nullptr;
}

hipError_t launch_conv_wino4h(const ConvParams& p, hipStream_t s) {
    const int regs_x = p.out.w / 16, regs_y = p.out.h / 16;
    const size_t nreg = p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * regs_x * regs_y;
    const size_t npairs = (nreg + 1) / 2;
    const size_t grid = npairs * (size_t)((p.out.c + 63) / 64) * 2;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    size_t lds = (size_t)(3 * WH_HS + 6 * 2 * WH_BWS) * 16 + 2 * 384 * 8;      // 79,872 B: two workgroups per CU
    const size_t lds_epi = (size_t)24 * WH_RPLANE * 4;
    if (lds_epi > lds) lds = lds_epi;
    static DeviceOnce attr_set;
    if (attr_set.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { attr_set.reset(); return e; }
    }
    static const bool dbg = getenv("ECSEG_DEBUG_OCC") != nullptr;
    if (dbg) {
        int nblk = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(conv_wino4h_kernel), 384, lds);
        fprintf(stderr, "wino4h: occupancy %d blocks/CU (rc %d), lds %zu, grid %zu\n", nblk, (int)e, lds, grid);
    }
    hipLaunchKernelGGL(conv_wino4h_kernel, dim3((unsigned)grid), dim3(384), lds, s, p, regs_x, regs_y, (int)npairs);
    return hipGetLastError();
}

}  // namespace ecseg
