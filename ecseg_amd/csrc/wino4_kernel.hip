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
// Pipeline: the halo arrives 8 input channels at a time by LDS-DMA (double-buffered, ONE s_barrier per 8 channels);
// the filter fragments are private to a wave (nobody else reads them), so every wave streams its own 3-KB stage
// (6 points x 4 input channels x 32 output channels) by LDS-DMA into a private double buffer, ordered by its own
// counted vmcnt only - no barrier on the filter path.
//
// LDS halo image, 16-byte slots (4 channels): slot(g, y, x, h) = (g * 18 + P(y)) * 36 + h * 18 + P(x), where
// P(v) = {0, 5, 10, 14}[v % 4] + v / 4 regroups the 18 halo rows / columns by their phase modulo the tile stride 4.
// For a fixed tile offset (i, j) the 16 lanes of a ds_read_b128 group then read slots 36 * ty + tx + const:
// 36 = 4 (mod 16), so all 16 land on different 16-byte bank groups: conflict-free without padding.  The A-operand
// lane -> tile map follows the hardware's b128 lane groups (see gty below).
#include <cstdlib>

#include "common.h"
#include "device_util.h"

namespace ecseg {

namespace {

constexpr int W4_HS = 1536;          // halo slots per buffer: 2 regions x 18 rows x 36 = 1296 used, padded to 24 x 64
constexpr int W4_BWS = 192;          // filter slots per wave and stage: 6 points x 2 halves x 32 couts x 2 k / 4
constexpr int W4_RPLANE = 1056;      // floats per (xi, x) plane of the output exchange image: 32 tiles x 32 couts + 32

__device__ __forceinline__ int w4_pos(int v) {           // 0..17 -> regrouped position
    const int m = v & 3;
    return (m == 0 ? 0 : m == 1 ? 5 : m == 2 ? 10 : 14) + (v >> 2);
}
__device__ __forceinline__ int w4_inv(int r) {           // regrouped position -> 0..17
    return r < 5 ? 4 * r : r < 10 ? 4 * (r - 5) + 1 : r < 14 ? 4 * (r - 10) + 2 : 4 * (r - 14) + 3;
}
constexpr int w4_cpos(int v) { return ((v & 3) == 0 ? 0 : (v & 3) == 1 ? 5 : (v & 3) == 2 ? 10 : 14) + (v >> 2); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

}  // namespace

__global__ __launch_bounds__(768) void conv_wino4_kernel(ConvParams p, int regs_x, int regs_y, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [2][W4_HS]        halo, double-buffered
    f32x4* Bs = Hs + 2 * W4_HS;                              // [12][2][W4_BWS]   per-wave filter stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave % 6, ch = wave / 6;
    const int li = lane & 31, lh = lane >> 5;

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = (int)(bid % (unsigned)npairs), nb = (int)(bid / (unsigned)npairs);
    const int H = p.in.h, W = p.in.w;                        // output extent == input extent
    const int ngroups = p.cin_chunks;                        // 8 input channels each
    const int nstages = 2 * ngroups;

    // the two 16 x 16 regions of this workgroup: consecutive in (patch, region row, region column) order
    int r_img[2], r_y0[2], r_x0[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int rid = 2 * pair + g;
        const int rx = rid % regs_x, t = rid / regs_x;
        const int ry = t % regs_y, img = t / regs_y;
        r_img[g] = img < p.n ? img : -1;
        r_y0[g] = ry * 16; r_x0[g] = rx * 16;
    }

    // ---- halo DMA descriptors: this lane fills slots 64 * k + lane, k = wave and wave + 12 ----
    const float* h_src[2];
    int h_step[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int a = 64 * (wave + 12 * k) + lane;
        h_src[k] = p.zero; h_step[k] = 0;
        if (a < 2 * 18 * 36) {
            const int g = a / 648, rem = a - g * 648;
            const int r = rem / 36, cc = rem - r * 36;
            const int h = cc / 18, c = cc - h * 18;
            const int img = g ? r_img[1] : r_img[0];
            const int iy = (g ? r_y0[1] : r_y0[0]) - 1 + w4_inv(r), ix = (g ? r_x0[1] : r_x0[0]) - 1 + w4_inv(c);
            if (img >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                h_src[k] = p.in.p + (((size_t)img * H + iy) * W + ix) * p.in.cs + 4 * h;
                h_step[k] = 8;
            }
        }
    }
    auto dma_halo = [&](int grp, int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(h_src[k] + grp * h_step[k]),
                                             (lptr_t)(Hs + buf * W4_HS + 64 * (wave + 12 * k) + lane), 16, 0, 0);
    };
    // ---- filter DMA: wt4[nb][stage][wave][point nu][h][cout 32][k 2], 768 floats per wave and stage ----
    const float* w_src = p.wt + ((size_t)nb * nstages * 12 + wave) * 768 + lane * 4;
    f32x4* Bw = Bs + wave * 2 * W4_BWS;
    auto dma_filter = [&](int stage, int buf) {
        const float* g = w_src + (size_t)stage * (12 * 768);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + k * 256), (lptr_t)(Bw + buf * W4_BWS + k * 64 + lane), 16, 0, 0);
    };

    // ---- A-operand lane -> tile.  ds_read_b128 serves lanes {0-3,12-15,20-27} and {4-11,16-19,28-31} of each half
    //      in separate LDS cycles; give each of those groups the 16 tiles of ONE region ----
    const int q8 = li >> 2, tx = li & 3;
    const int tg = (0x96 >> q8) & 1;                              // region of lane quad q8: 0,1,1,0,1,0,0,1
    const int ty = (q8 == 0 || q8 == 1) ? 0 : (q8 == 2 || q8 == 3) ? 1 : (q8 == 4 || q8 == 5) ? 2 : 3;
    const int a_lane = (tg * 18 + ty) * 36 + lh * 18 + tx;       // slot of halo pixel (4 ty, 4 tx) of the lane's tile

    // row transform of wave xi: t = c0 d[r0] + c1 d[r1] + c2 d[r2] + c3 d[r3]
    int rr0, rr1, rr2, rr3; float c0, c1, c2, c3;
    switch (xi) {
        case 0:  rr0 = 0; rr1 = 2; rr2 = 4; rr3 = 4; c0 = 4.f;  c1 = -5.f; c2 = 1.f;  c3 = 0.f; break;
        case 1:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -4.f; c1 = -4.f; c2 = 1.f;  c3 = 1.f; break;
        case 2:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = 4.f;  c1 = -4.f; c2 = -1.f; c3 = 1.f; break;
        case 3:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = -2.f; c1 = -1.f; c2 = 2.f;  c3 = 1.f; break;
        case 4:  rr0 = 1; rr1 = 2; rr2 = 3; rr3 = 4; c0 = 2.f;  c1 = -1.f; c2 = -2.f; c3 = 1.f; break;
        default: rr0 = 1; rr1 = 3; rr2 = 5; rr3 = 5; c0 = 4.f;  c1 = -5.f; c2 = 1.f;  c3 = 0.f; break;
    }
    const int ro0 = 36 * w4_pos(rr0), ro1 = 36 * w4_pos(rr1), ro2 = 36 * w4_pos(rr2), ro3 = 36 * w4_pos(rr3);

    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;

    dma_halo(0, 0);
    dma_filter(0, 0);
    for (int grp = 0; grp < ngroups; ++grp) {
        const bool more = grp + 1 < ngroups;
        // this wave's pieces of halo group grp have landed (only the 3 pieces of filter stage 2 grp may be in flight);
        // the barrier publishes everybody's pieces and retires the other buffer's readers
        asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        if (more) dma_halo(grp + 1, (grp + 1) & 1);

        // ---- row transform: t[j] for the six halo columns of the lane's tile, 4 channels each ----
        const f32x4* A = Hs + (grp & 1) * W4_HS + a_lane;
        f32x4 t[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            constexpr int cp[6] = {w4_cpos(0), w4_cpos(1), w4_cpos(2), w4_cpos(3), w4_cpos(4), w4_cpos(5)};
            const f32x4 d0 = A[ro0 + cp[j]], d1 = A[ro1 + cp[j]], d2 = A[ro2 + cp[j]], d3 = A[ro3 + cp[j]];
            t[j] = c0 * d0 + c1 * d1 + c2 * d2 + c3 * d3;
            asm volatile("" : "+v"(t[j]));                   // finish this column here: 16 transient registers, not 96
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            // stage 2 grp + ss lives in filter buffer ss; start the next stage's stream, then wait for this one
            if (ss == 0) {
                dma_filter(2 * grp + 1, 1);
                if (more) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // halo grp+1 (2) + stage+1 (3) may fly
                else      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                if (more) { dma_filter(2 * grp + 2, 0); asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
                else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // column transform of this stage's two channels: V[nu] = sum_j B^T[nu][j] t[j]
            f32x2 u[6], V[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { u[j][0] = t[j][2 * ss]; u[j][1] = t[j][2 * ss + 1]; }
            {
                const f32x2 a42 = u[4] - 4.f * u[2], a31 = u[3] - 4.f * u[1];
                const f32x2 b42 = u[4] - u[2], b31 = 2.f * (u[3] - u[1]);
                V[0] = 4.f * u[0] - 5.f * u[2] + u[4];
                V[1] = a42 + a31;
                V[2] = a42 - a31;
                V[3] = b42 + b31;
                V[4] = b42 - b31;
                V[5] = 4.f * u[1] - 5.f * u[3] + u[5];
            }
            const f32x2* Bp = reinterpret_cast<const f32x2*>(Bw + ss * W4_BWS) + lane;
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                const f32x2 w2 = Bp[v * 64];
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v][0], w2[0], acc[v], 0, 0, 0);
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v][1], w2[1], acc[v], 0, 0, 0);
            }
        }
    }

    // ---- output stage: two passes (channel halves) through a [xi][x][tile][32 couts] exchange image ----
    float* Rs = reinterpret_cast<float*>(smem);
    const int Cout = p.out.c;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                     // main-loop LDS reads / previous pass's combine are done
        if (ch == pass) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tl = (e & 3) + 8 * (e >> 2) + 4 * lh;           // accumulator row = tile slot
                const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                float* o = Rs + (xi * 4) * W4_RPLANE + tl * 32 + li;
                o[0 * W4_RPLANE] = m0 + s12 + s34;
                o[1 * W4_RPLANE] = d12 + 2.f * d34;
                o[2 * W4_RPLANE] = s12 + 4.f * s34;
                o[3 * W4_RPLANE] = d12 + 8.f * d34 + m5;
            }
        }
        __syncthreads();
        const int n0 = nb * 64 + pass * 32;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + k * 768;
            if (item >= 1024) break;
            const int q = item & 7, x = (item >> 3) & 3, n = item >> 5;
            const int nq8 = n >> 2, ntx = n & 3;
            const int g = (0x96 >> nq8) & 1;
            const int nty = nq8 < 2 ? 0 : nq8 < 4 ? 1 : nq8 < 6 ? 2 : 3;
            const int img = g ? r_img[1] : r_img[0];
            if (img < 0) continue;
            const float* r = Rs + x * W4_RPLANE + n * 32 + 4 * q;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(r + 0 * 4 * W4_RPLANE);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(r + 1 * 4 * W4_RPLANE);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(r + 2 * 4 * W4_RPLANE);
            const f32x4 q3 = *reinterpret_cast<const f32x4*>(r + 3 * 4 * W4_RPLANE);
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(r + 4 * 4 * W4_RPLANE);
            const f32x4 q5 = *reinterpret_cast<const f32x4*>(r + 5 * 4 * W4_RPLANE);
            const int co = n0 + 4 * q;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + co);
            const f32x4 s12 = q1 + q2, d12 = q1 - q2, s34 = q3 + q4, d34 = q3 - q4;
            f32x4 y[4];
            y[0] = q0 + s12 + s34 + bv;
            y[1] = d12 + 2.f * d34 + bv;
            y[2] = s12 + 4.f * s34 + bv;
            y[3] = d12 + 8.f * d34 + q5 + bv;
            const int oy = (g ? r_y0[1] : r_y0[0]) + 4 * nty, ox = (g ? r_x0[1] : r_x0[0]) + 4 * ntx + x;
            float* o = p.out.p + (((size_t)img * H + oy) * W + ox) * p.out.cs + co;
            if (co + 3 < Cout) {
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    f32x4 v = y[yy];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.alpha);
                    *reinterpret_cast<f32x4*>(o + (size_t)yy * W * p.out.cs) = v;
                }
            }
        }
    }
}

// Eligibility beyond "3x3, stride 1, pad 1, same size" (checked by the caller): extents multiples of 16, channels
// multiples of 8 / 64, 16-byte aligned views.
bool conv_wino4_supported(const ConvParams& p) {
    return p.in.h == p.out.h && p.in.w == p.out.w && p.out.h % 16 == 0 && p.out.w % 16 == 0 && p.in.c % 8 == 0 &&
           p.in.c >= 8 && p.out.c % 64 == 0 && p.in.cs % 4 == 0 && p.out.cs % 4 == 0 && p.zero != nullptr;
}

hipError_t launch_conv_wino4(const ConvParams& p, hipStream_t s) {
    const int regs_x = p.out.w / 16, regs_y = p.out.h / 16;
    const size_t nreg = (size_t)p.n * regs_x * regs_y;
    const size_t npairs = (nreg + 1) / 2;
    const size_t grid = npairs * (size_t)(p.out.c / 64);
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    size_t lds = (size_t)(2 * W4_HS + 12 * 2 * W4_BWS) * 16;
    const size_t lds_epi = (size_t)24 * W4_RPLANE * 4;
    if (lds_epi > lds) lds = lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)grid), dim3(768), lds, s, p, regs_x, regs_y, (int)npairs);
    return hipGetLastError();
}

}  // namespace ecseg
