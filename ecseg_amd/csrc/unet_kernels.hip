// U-Net layer kernels for gfx950 (MI355X).  NHWC float32 everywhere.
//
// The hot kernel is conv_mfma: an im2col-free implicit-GEMM convolution on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32, exact f32 fma chains).  Per workgroup: 128 output pixels (a TH x TW spatial tile of one
// patch) x BN output channels; the input halo tile and the filter slab of one 8-channel K-chunk are staged in LDS and
// every one of the R*S taps is served from that single staged halo (no im2col buffer, no re-read of the input).
//
//   M (MFMA rows)  = output pixels, lane i = l & 31 -> pixel i of the wave's 32-pixel strip
//   N (MFMA cols)  = output channels
//   K              = (tap, input channel); one MFMA consumes channels {e, 4 + e} of the chunk for one tap, so a
//                    single ds_read_b128 per operand feeds four MFMAs (k-order inside a chunk is permuted
//                    identically for A and B, which only re-orders the exact f32 accumulation).
//
// LDS images (conflict-free ds_read_b128: consecutive lanes read consecutive 16-byte slots)
//   As[half h][halo pixel][4 ch]           channels 4h .. 4h+3 of the chunk
//   Bs[tap][half h][out channel][4 ch]
// Global filter layout (re-laid out once at model load): wt[tap][chunk][h][N padded][4].
#include <cstdlib>

#include "common.h"
#include "device_util.h"

namespace ecseg {

template <int NT, int TW, int R, int S, int KCH = 1, int ST = 1>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p, int tiles_x, int tiles_y, int nblk_n, int np_total) {
    // KCH = 8-channel K-chunks staged per barrier interval (4 for 1x1 taps: a single tap gives a wave only 4*NT MFMAs
    // per chunk, too few to amortise the barrier / staging cost).  ST = output stride (2: strided convolutions - the halo
    // tile covers (TH - 1) ST + R rows and a lane's A operand sits ST halo pixels from its neighbour's).
    constexpr int TH = 128 / TW;
    constexpr int HH = (TH - 1) * ST + R, HW = (TW - 1) * ST + S;
    constexpr int NPIX = HH * HW;
    constexpr int BN = NT * 32;
    constexpr int A_PIECES = NPIX * 2;
    constexpr int A_PER_T = (A_PIECES * KCH + 255) / 256;
    constexpr int B_PIECES = R * S * 2 * BN;
    constexpr int B_PER_T = (B_PIECES * KCH + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* As = reinterpret_cast<f32x4*>(smem);   // [KCH][2][NPIX]
    f32x4* Bs = As + KCH * 2 * NPIX;              // [KCH][R*S][2][BN]

    const int tid = threadIdx.x;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % nblk_n; bid /= nblk_n;
    int tx0, ty0, img;
    if (p.lut != nullptr) {                                  // cropped launch: tile origins (in 4-pixel units) from the list
        const int i = bid / p.lut_len, v = p.lut[bid - i * p.lut_len];
        img = i * p.per_image + (v >> 16); ty0 = ((v >> 8) & 255) * 4; tx0 = (v & 255) * 4;
    } else {
        tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
        ty0 = (bid % tiles_y) * TH; bid /= tiles_y;
        img = bid;
    }
    const int n0 = nb * BN;

    const int Hin = p.in.h, Win = p.in.w, Cin = p.in.c;
    const size_t in_img = (size_t)img * Hin * Win * p.in.cs;

    // ---- per-thread staging descriptors (fixed over the K loop) ----
    long a_off[A_PER_T];     // float offset of this thread's 16-byte piece for stage 0, or -1 when zero-filled
    int a_ch[A_PER_T];       // first channel of the piece inside stage 0
    int a_lds[A_PER_T];
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int q = tid + k * 256;
        a_off[k] = -1; a_lds[k] = -1; a_ch[k] = 0;
        if (q < A_PIECES * KCH) {
            const int kc = q / A_PIECES, qq = q - kc * A_PIECES;
            const int pix = qq >> 1, h = qq & 1;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int iy = ty0 * ST - p.pad_top + hy, ix = tx0 * ST - p.pad_left + hx;
            a_lds[k] = (kc * 2 + h) * NPIX + pix;
            a_ch[k] = kc * 8 + h * 4;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win)
                a_off[k] = (long)(in_img + ((size_t)iy * Win + ix) * p.in.cs + kc * 8 + h * 4);
        }
    }
    const size_t chunk_stride = (size_t)p.wt_chunk_stride;         // floats between consecutive chunks of one tap
    const size_t tap_stride = (size_t)p.wt_tap_stride;
    long b_off[B_PER_T];
    int b_kc[B_PER_T];
#pragma unroll
    for (int k = 0; k < B_PER_T; ++k) {
        const int q = tid + k * 256;
        b_off[k] = -1; b_kc[k] = 0;
        if (q < B_PIECES * KCH) {
            const int kc = q / B_PIECES, qq = q - kc * B_PIECES;
            const int tap = qq / (2 * BN), rem = qq - tap * 2 * BN;
            const int h = rem / BN, j = rem - h * BN;
            b_kc[k] = kc;
            b_off[k] = (long)(tap * tap_stride + kc * chunk_stride + ((size_t)h * np_total + n0 + j) * 4);
        }
    }
    const int nstages = (p.cin_chunks + KCH - 1) / KCH;

    f32x4 a_reg[A_PER_T], b_reg[B_PER_T];
    auto load_chunk = [&](int st) {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_off[k] >= 0 && st * KCH * 8 + a_ch[k] < Cin)
                v = *reinterpret_cast<const f32x4*>(p.in.p + a_off[k] + (size_t)st * KCH * 8);
            a_reg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_off[k] >= 0 && st * KCH + b_kc[k] < p.cin_chunks)
                v = *reinterpret_cast<const f32x4*>(p.wt + b_off[k] + (size_t)st * KCH * chunk_stride);
            b_reg[k] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k)
            if (a_lds[k] >= 0) As[a_lds[k]] = a_reg[k];
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const int q = tid + k * 256;
            if (q < B_PIECES * KCH) Bs[q] = b_reg[k];
        }
    };

    // ---- wave / lane roles ----
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int ty, tx;
    if (TW == 32) { ty = wave; tx = li; } else { ty = 2 * wave + (li >> 4); tx = li & 15; }
    const f32x4* Ap = As + lh * NPIX + ty * ST * HW + tx * ST;
    const f32x4* Bp = Bs + lh * BN + li;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    // Sub-pixel transposed convolutions (round 6): 7 of the 16 (tap, output phase) filter blocks of a 3x3 / stride-2 layer are all zero
    // - the phases have 4, 2, 2 and 1 taps - and the up-sample + 2x2 decoder step lowered to such a layer lives on exactly those 9.
    // skip[nt]: bit t set = tap t of this workgroup's 32-column tile nt multiplies zeros only (a tile that straddles two phases -
    // coutp = 16 - skips what both have in common).  Wave-uniform: one scalar branch per (tap, tile).
    unsigned skip[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        skip[nt] = 0;
        if (R == 2 && S == 2 && p.convt && p.tap_zero_mask) {
            const int ab0 = (n0 + nt * 32) / p.coutp, ab1 = min((n0 + nt * 32 + 31) / p.coutp, p.kT * p.kT - 1);
#pragma unroll
            for (int t = 0; t < R * S; ++t) {
                bool z = true;
                for (int ab = ab0; ab <= ab1; ++ab) z = z && ((p.tap_zero_mask >> (t * 4 + ab)) & 1);
                if (z) skip[nt] |= 1u << t;
            }
        }
    }

    load_chunk(0);
    for (int st = 0; st < nstages; ++st) {
        store_chunk();
        __syncthreads();
        if (st + 1 < nstages) load_chunk(st + 1);   // next stage's HBM/L2 latency hides under this stage's MFMAs
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const f32x4 a = Ap[kc * 2 * NPIX + r * HW + s];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (R == 2 && S == 2 && ((skip[nt] >> (r * S + s)) & 1)) continue;
                        const f32x4 b = Bp[(kc * R * S + r * S + s) * 2 * BN + nt * 32];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[nt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: each wave turns its 32 pixel x 32 channel accumulator tile around through a private 4-KB LDS tile
    //      (the K loop's last barrier has retired the staging buffers) so that a lane finishes 4 consecutive channels of
    //      one pixel: bias + activation, one 16-byte store; a wave instruction writes 8 pixels x 128 B instead of 64 x 4 B ----
    const int Hout = p.out.h, Wout = p.out.w, Cout = p.out.c;
    const int Ht = p.convt ? Hin + p.convt_ext : Hout, Wt = p.convt ? Win + p.convt_ext : Wout;   // extent the tiles walk over
    float* Xs = reinterpret_cast<float*>(smem) + wave * 1024;
    const bool vec_ok = (p.out.cs % 4 == 0) && (Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out.p) & 15) == 0);
    const int quad = lane & 7;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int co = n0 + nt * 32 + 4 * quad, oa = 0, ob = 0;
        if (p.convt) {
            // N index = (a * kT + b) * coutp + co, decoded per lane: with coutp = 16 a 32-column tile holds BOTH column phases
            // b = 0, 1 of one row phase a, i.e. the two output pixels (2x, 2x + 1) a lane octet writes are neighbours in memory
            const int ab = co / p.coutp;
            co -= ab * p.coutp;
            oa = ab / p.kT; ob = ab - oa * p.kT;
            oa += p.phase_a; ob += p.phase_b;                  // (phase-by-phase launches: N holds one phase, ab = 0)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) Xs[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + li] = acc[nt][e];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (co + c < Cout) bv[c] = p.bias[co + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 3) + 8 * r;               // pixel index inside the wave's 32-pixel strip
            f32x4 v = *reinterpret_cast<const f32x4*>(Xs + row * 32 + 4 * quad) + bv;
            v = apply_act4(v, p.act, p.alpha);
            int py, px;
            if (TW == 32) { py = wave; px = row; } else { py = 2 * wave + (row >> 4); px = row & 15; }
            const int tyy = ty0 + py, txx = tx0 + px;          // position in the tiled extent
            int oy = tyy, ox = txx;
            bool ok = tyy < Ht && txx < Wt;                     // the tile may overhang the extent
            if (p.convt) {
                oy = tyy * p.kT + oa - p.crop_top;
                ox = txx * p.kT + ob - p.crop_left;
                ok = ok && oy >= 0 && oy < Hout && ox >= 0 && ox < Wout;
            }
            if (ok) {
                float* o = p.out.p + (((size_t)img * Hout + oy) * Wout + ox) * p.out.cs + co;
                if (vec_ok && co + 3 < Cout) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));   // (streamed once, read by a later launch)
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (co + c < Cout) o[c] = v[c];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) on the fp32 matrix cores: 16 instead of 36 multiplies per 2x2 output block (2.25x fewer MFMAs).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 outputs
//
// The filter transform U = G g G^T is done once at model load (float64 on the host).  A workgroup owns 4 x 8 tiles
// (8 x 16 output pixels) x BN output channels; its four waves split the 16 transform points by ROW a of the 4x4
// transform domain: wave a reads the two raw input rows its row transform needs straight from the staged halo in LDS,
// forms t = B^T[a,:] d and the four column points V[a][0..3] in registers (32 packed adds per 32 MFMAs - free beside
// the matrix pipe) and multiplies them with U[a][b] on the MFMA.  No transformed input ever touches LDS or HBM.
// After the K loop each wave folds its own row (R_a = M[a][:] A) in registers, the four rows meet once through LDS,
// and Y = A^T R (+ bias, activation) is written as 128-byte row segments.
//
// LDS: As[h][col parity][10 rows][12 (9 used)][4ch] - even / odd halo columns in separate planes and a row pitch of
// 12 slots make every ds_read_b128 of the 4 x 8 tile lanes conflict-free; Bs[point 16][h][BN][4ch].
// Winograd output stage shared by the kernels below.  The four waves (transform rows a = 0..3) of a row tile fold
// their own row in registers (R_a = M[a][:] A), meet through LDS in a [a][jp][tile][channel] image, and every lane then
// finishes 4 consecutive output channels of one pixel pair: Y = A^T R + bias, activation, two 16-byte stores.  A wave
// instruction writes 4 pixels x 256 B: 8 store instructions per wave instead of 64 dword stores.
template <int NT>
__device__ __forceinline__ void wino_write_R(float* Rs, const f32x16 (&acc)[4][NT], int wa, int lane) {
    constexpr int BN = NT * 32;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int t = (e & 3) + 8 * (e >> 2) + 4 * lh;           // accumulator row = tile index inside the row tile
            const float ra = acc[0][nt][e] + acc[1][nt][e] + acc[2][nt][e];
            const float rb = acc[1][nt][e] - acc[2][nt][e] - acc[3][nt][e];
            Rs[((wa * 2 + 0) * 32 + t) * BN + nt * 32 + li] = ra;
            Rs[((wa * 2 + 1) * 32 + t) * BN + nt * 32 + li] = rb;
        }
}

template <int NT>
__device__ __forceinline__ void wino_store_Y(const float* Rs, const ConvParams& p, int img, int oy0, int ox0, int n0, int wa,
                                             int lane) {
    constexpr int BN = NT * 32;
    constexpr int QN = BN / 4;                 // channel quads per pixel
    constexpr int TPI = 64 / QN;               // tiles per wave instruction
    const int q = lane % QN, tsub = lane / QN;
    const int Hout = p.out.h, Wout = p.out.w, Cout = p.out.c;
    const int co = n0 + 4 * q;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && co + 3 < Cout) bv = *reinterpret_cast<const f32x4*>(p.bias + co);
    f32x4 pm = {0.f, 0.f, 0.f, 0.f};
    // 32 tiles x 2 column parities, 4 waves x TPI tiles per step; a wave does both column parities of its tiles (the
    // fused 2x2 max-pool below needs the 2 x 2 outputs of a tile in one thread)
    for (int it = 2 * wa; it < (32 / TPI) * 2; it += (it & 1) ? 7 : 1) {
        const int jp = it & 1, t = (it >> 1) * TPI + tsub;
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(Rs + ((0 * 2 + jp) * 32 + t) * BN + 4 * q);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(Rs + ((1 * 2 + jp) * 32 + t) * BN + 4 * q);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(Rs + ((2 * 2 + jp) * 32 + t) * BN + 4 * q);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(Rs + ((3 * 2 + jp) * 32 + t) * BN + 4 * q);
        f32x4 y0 = q0 + q1 + q2 + bv, y1 = q1 - q2 - q3 + bv;
        y0 = apply_act4(y0, p.act, p.alpha); y1 = apply_act4(y1, p.act, p.alpha);
        const int oy = oy0 + 2 * (t >> 3), ox = ox0 + 2 * (t & 7) + jp;
        if (co + 3 < Cout && ox < Wout) {
            float* o = p.out.p + (((size_t)img * Hout + oy) * Wout + ox) * p.out.cs + co;
            if (oy < Hout) *reinterpret_cast<f32x4*>(o) = y0;
            if (oy + 1 < Hout) *reinterpret_cast<f32x4*>(o + (size_t)Wout * p.out.cs) = y1;
        }
        if (p.pool.p != nullptr) {
            // fused MaxPooling2D(2x2, stride 2): even extents (checked by the caller), so a tile is inside or outside as a whole
#pragma unroll
            for (int c = 0; c < 4; ++c) pm[c] = jp ? fmaxf(pm[c], fmaxf(y0[c], y1[c])) : fmaxf(y0[c], y1[c]);
            if (jp && co + 3 < Cout && ox < Wout && oy + 1 < Hout)
                *reinterpret_cast<f32x4*>(p.pool.p + (((size_t)img * p.pool.h + (oy >> 1)) * p.pool.w + (ox >> 1)) * p.pool.cs + co) = pm;
        }
    }
}

template <int NT, int MT>
__global__ __launch_bounds__(256, (MT == 1 ? 2 : 1)) void conv_wino_kernel(ConvParams p, int tiles_x, int tiles_y, int nblk_n) {
    constexpr int TTY = 4 * MT, TTX = 8;                     // MT MFMA row tiles of 4 x 8 Winograd tiles each
    constexpr int HR = 2 * TTY + 2, HC = 2 * TTX + 2;       // halo: (8 MT + 2) x 18 pixels
    constexpr int CS = 12;                                   // slots per (plane, row): 9 used
    constexpr int PLANE = HR * CS;
    constexpr int A_SLOTS = 4 * PLANE;                       // [h][parity]
    constexpr int BN = NT * 32;
    constexpr int A_PIECES = HR * HC * 2;
    constexpr int A_PER_T = (A_PIECES + 255) / 256;
    constexpr int B_PIECES = 16 * 2 * BN;
    constexpr int B_PER_T = B_PIECES / 256;

    // double-buffered: [As0 | As1 | Bs0 | Bs1]; the filter slab goes global -> LDS directly (LDS-DMA, no VGPR staging)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* As = reinterpret_cast<f32x4*>(smem);
    f32x4* Bs = As + 2 * A_SLOTS;

    const int tid = threadIdx.x;
    // spatial position fastest, output-channel block slowest: the workgroups that run together on one XCD stream the
    // same filter slabs
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % tiles_x; bid /= tiles_x;
    const int by = bid % tiles_y; bid /= tiles_y;
    const int img = bid % p.n; bid /= p.n;
    const int nb = bid;
    const int ox0 = bx * (2 * TTX), oy0 = by * (2 * TTY);
    const int n0 = nb * BN;
    const int Hin = p.in.h, Win = p.in.w, Cin = p.in.c;
    const size_t in_img = (size_t)img * Hin * Win * p.in.cs;

    // rows / columns this window may read (ConvParams::in_box; the whole image otherwise): zero outside
    int lo_y = 0, hi_y = Hin - 1, lo_x = 0, hi_x = Win - 1;
    if (p.in_box != nullptr) {
        const int32_t* bx = p.in_box + 4 * ((img + p.box_first) % p.per_image);
        lo_y = bx[0]; hi_y = bx[1]; lo_x = bx[2]; hi_x = bx[3];
    }
    long a_off[A_PER_T];
    int a_ch[A_PER_T], a_lds[A_PER_T];
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int q = tid + k * 256;
        a_off[k] = -1; a_lds[k] = -1; a_ch[k] = 0;
        if (q < A_PIECES) {
            const int pix = q >> 1, h = q & 1;
            const int hy = pix / HC, hx = pix - hy * HC;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            a_lds[k] = (h * 2 + (hx & 1)) * PLANE + hy * CS + (hx >> 1);
            a_ch[k] = h * 4;
            if (iy >= lo_y && iy <= hi_y && ix >= lo_x && ix <= hi_x)
                a_off[k] = (long)(in_img + ((size_t)iy * Win + ix) * p.in.cs + h * 4);
        }
    }
    const size_t chunk_stride = (size_t)p.wt_chunk_stride;
    const size_t tap_stride = (size_t)p.wt_tap_stride;
    long b_off[B_PER_T];
#pragma unroll
    for (int k = 0; k < B_PER_T; ++k) {
        const int q = tid + k * 256;
        const int tap = q / (2 * BN), rem = q - tap * 2 * BN;
        const int h = rem / BN, j = rem - h * BN;
        b_off[k] = (long)(tap * tap_stride + ((size_t)h * p.coutp + n0 + j) * 4);
    }
    f32x4 a_reg[A_PER_T];
    auto load_a = [&](int c) {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_off[k] >= 0 && c * 8 + a_ch[k] < Cin)
                v = *reinterpret_cast<const f32x4*>(p.in.p + a_off[k] + (size_t)c * 8);
            a_reg[k] = v;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k)
            if (a_lds[k] >= 0) As[buf * A_SLOTS + a_lds[k]] = a_reg[k];
    };
    auto dma_b = [&](int c, int buf) {          // LDS destination = wave base + lane * 16: the slab image is lane-linear
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const float* g = p.wt + b_off[k] + (size_t)c * chunk_stride;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + buf * B_PIECES + tid + k * 256), 16, 0, 0);
        }
    };

    const int lane = tid & 63, wa = tid >> 6;              // wave = transform row a
    const int li = lane & 31, lh = lane >> 5;
    const int ti = li >> 3, tj = li & 7;
    // row transform of wave a: t = s0 * d[r0] + s1 * d[r1]
    const int r0 = (wa == 0) ? 0 : 1, r1 = (wa == 3) ? 3 : 2;
    const float s0 = (wa == 2) ? -1.f : 1.f, s1 = (wa == 0 || wa == 3) ? -1.f : 1.f;
    const int a0_off = (lh * 2) * PLANE + (2 * ti + r0) * CS + tj;         // even columns (j = 0, 2)
    const int a1_off = (lh * 2) * PLANE + (2 * ti + r1) * CS + tj;
    const int b_lane = (wa * 4 * 2 + lh) * BN + li;

    f32x16 acc[MT][4][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][b][nt][e] = 0.f;

    load_a(0);
    dma_b(0, 0);
    store_a(0);
    __syncthreads();                                       // (hipcc drains vmcnt before the barrier: slab 0 has landed)
    for (int c = 0; c < p.cin_chunks; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < p.cin_chunks;
        if (more) { load_a(c + 1); dma_b(c + 1, cur ^ 1); }   // next chunk streams in under this chunk's MFMAs
        const f32x4* A0 = As + cur * A_SLOTS + a0_off;
        const f32x4* A1 = As + cur * A_SLOTS + a1_off;
        const f32x4* Bp = Bs + cur * B_PIECES + b_lane;
        // issue the LDS reads of the chunk first (8 halo fragments per row tile + 4*NT filter fragments), then
        // transform, then 32*MT MFMAs back to back; the second row tile's transform overlaps the first one's MFMAs
        f32x4 d[MT][8];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ro = m * 8 * CS;                     // row tile m starts 8 halo rows further down
            d[m][0] = A0[ro]; d[m][1] = A1[ro]; d[m][2] = A0[ro + PLANE]; d[m][3] = A1[ro + PLANE];
            d[m][4] = A0[ro + 1]; d[m][5] = A1[ro + 1]; d[m][6] = A0[ro + PLANE + 1]; d[m][7] = A1[ro + PLANE + 1];
        }
        f32x4 w[4][NT];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[b][nt] = Bp[b * 2 * BN + nt * 32];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            // halo column j of the tile: plane (j & 1), slot + (j >> 1)
            const f32x4 t0 = s0 * d[m][0] + s1 * d[m][1];
            const f32x4 t1 = s0 * d[m][2] + s1 * d[m][3];
            const f32x4 t2 = s0 * d[m][4] + s1 * d[m][5];
            const f32x4 t3 = s0 * d[m][6] + s1 * d[m][7];
            f32x4 V[4];
            V[0] = t0 - t2; V[1] = t1 + t2; V[2] = t2 - t1; V[3] = t1 - t3;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[m][b][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[b][e], w[b][nt][e], acc[m][b][nt], 0, 0, 0);
                }
            }
        }
        if (more) store_a(cur ^ 1);
        __syncthreads();
    }

    // ---- output stage, one row tile at a time (wino_write_R / wino_store_Y) ----
    float* Rs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m) __syncthreads();
        wino_write_R<NT>(Rs, acc[m], wa, lane);
        __syncthreads();
        wino_store_Y<NT>(Rs, p, img, oy0 + 8 * m, ox0, n0, wa, lane);
    }
}

template <int NT, int MT>
static hipError_t launch_conv_wino_t(const ConvParams& p, hipStream_t s) {
    constexpr int BN = NT * 32;
    constexpr int TTY = 4 * MT;
    const int tiles_x = (p.out.w + 15) / 16, tiles_y = (p.out.h + 2 * TTY - 1) / (2 * TTY);
    const int nblk_n = p.coutp / BN;
    size_t lds = (size_t)2 * (4 * (2 * TTY + 2) * 12 + 16 * 2 * BN) * 16;
    const size_t lds_epi = (size_t)4 * 2 * NT * 4 * 64 * 16;
    if (lds_epi > lds) lds = lds_epi;
    const size_t grid = (size_t)p.n * tiles_x * tiles_y * nblk_n;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    static DeviceOnce attr_set;                             // the attribute is per device
    const hipError_t ea = attr_set.run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<NT, MT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((conv_wino_kernel<NT, MT>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y, nblk_n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Filter-resident F(2x2, 3x3) for NARROW layers (Cin <= 32, Cout <= 32: the full-resolution levels of base-16 / base-32
// U-Nets).  conv_wino_kernel streams a 16-point filter slab per 8 input channels for every 8 x 16 output pixels; with
// <= 32 channels that slab (16 KB) is larger than the input tile it is multiplied with, and the layer runs at the
// filter stream's L2 rate (1.6 TB/s of HBM-side traffic, 0.16-0.23 of the MFMA peak).  Here the whole transformed
// filter of a wave's transform row (4 points x Cin x 32 output channels = 16 * CH registers) is loaded ONCE into
// registers and the workgroup walks `tpw` tiles of a tile row: per tile only the input halo (all channels, one LDS
// image) and the outputs move.  Same arithmetic, lane maps and output stage as conv_wino_kernel<1, 1> (bit-identical
// results): wave a = transform row a, 32 MFMAs per 8 input channels and tile.
// Pipeline per tile (two barriers): the next tile's halo is fetched into registers before the MFMAs of the current one,
// stored to LDS after the exchange barrier (the halo image is dead by then) and before the output stores are issued.
// ------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256, 2) void conv_wino_res_kernel(ConvParams p, int tiles_x, int tiles_y, int segs_x, int tpw) {
    constexpr int HR = 10, HC = 18;                          // halo of 4 x 8 Winograd tiles: 10 x 18 pixels
    constexpr int CS = 12;                                   // slots per (plane, row): 9 used
    constexpr int PLANE = HR * CS;
    constexpr int A_SLOTS = 4 * PLANE;                       // per 8-channel chunk: [h][column parity]
    constexpr int PIECES = HR * HC * 2 * CH;                 // 16-byte pieces of one halo (all channels)
    constexpr int A_PER_T = (PIECES + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* As = reinterpret_cast<f32x4*>(smem);              // [CH][A_SLOTS]
    float* Rs = reinterpret_cast<float*>(As + CH * A_SLOTS); // [4][2][32][32] output exchange image

    const int tid = threadIdx.x;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = bid % segs_x; bid /= segs_x;
    const int by = bid % tiles_y; bid /= tiles_y;
    const int img = bid;
    const int bx0 = seg * tpw;
    const int ntile = min(tpw, tiles_x - bx0);
    const int oy0 = by * 8;
    const int Hin = p.in.h, Win = p.in.w, Cin = p.in.c;
    const size_t in_img = (size_t)img * Hin * Win * p.in.cs;

    // rows / columns this window may read (ConvParams::in_box; the whole image otherwise): zero outside
    int lo_y = 0, hi_y = Hin - 1, lo_x = 0, hi_x = Win - 1;
    if (p.in_box != nullptr) {
        const int32_t* bx = p.in_box + 4 * ((img + p.box_first) % p.per_image);
        lo_y = bx[0]; hi_y = bx[1]; lo_x = bx[2]; hi_x = bx[3];
    }
    // halo pieces of this thread: consecutive threads take the 2 * CH channel quads of one pixel (Cin * 4 contiguous bytes)
    long a_off[A_PER_T];                                     // offset of the piece for tile column 0 (may be "negative" at x = -1)
    int a_lds[A_PER_T], a_hx[A_PER_T];                       // LDS slot (-1: no piece), halo column | 0x100 when the row / channels are outside
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int q = tid + k * 256;
        a_off[k] = 0; a_lds[k] = -1; a_hx[k] = 0x100;
        if (q < PIECES) {
            const int pix = q / (2 * CH), sub = q - pix * (2 * CH);
            const int c = sub >> 1, h = sub & 1;
            const int hy = pix / HC, hx = pix - hy * HC;
            const int iy = oy0 - 1 + hy;
            a_lds[k] = c * A_SLOTS + (h * 2 + (hx & 1)) * PLANE + hy * CS + (hx >> 1);
            const bool ok = iy >= lo_y && iy <= hi_y && c * 8 + h * 4 < Cin;
            a_hx[k] = hx | (ok ? 0 : 0x100);
            a_off[k] = (long)in_img + ((long)iy * Win + (long)(bx0 * 16 - 1 + hx)) * p.in.cs + c * 8 + h * 4;
        }
    }
    f32x4 a_reg[A_PER_T];
    auto load_a = [&](int t) {                               // tile t of this workgroup's walk
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int ix = (bx0 + t) * 16 - 1 + (a_hx[k] & 0xff);
            if (!(a_hx[k] & 0x100) && ix >= lo_x && ix <= hi_x)
                v = *reinterpret_cast<const f32x4*>(p.in.p + a_off[k] + (long)t * 16 * p.in.cs);
            a_reg[k] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k)
            if (a_lds[k] >= 0) As[a_lds[k]] = a_reg[k];
    };

    const int lane = tid & 63, wa = tid >> 6;                // wave = transform row a
    const int li = lane & 31, lh = lane >> 5;
    const int ti = li >> 3, tj = li & 7;
    const int r0 = (wa == 0) ? 0 : 1, r1 = (wa == 3) ? 3 : 2;
    const float s0 = (wa == 2) ? -1.f : 1.f, s1 = (wa == 0 || wa == 3) ? -1.f : 1.f;
    const int a0_off = (lh * 2) * PLANE + (2 * ti + r0) * CS + tj;
    const int a1_off = (lh * 2) * PLANE + (2 * ti + r1) * CS + tj;

    load_a(0);
    // the wave's filter fragments: [chunk][column point b]: 4 k-steps of the MFMA B operand (lane = (channel half lh, cout li))
    f32x4 w[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            w[c][b] = *reinterpret_cast<const f32x4*>(p.wt + (size_t)(wa * 4 + b) * (size_t)p.wt_tap_stride +
                                                      (size_t)c * (size_t)p.wt_chunk_stride + ((size_t)lh * p.coutp + li) * 4);
    store_a();
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const bool more = t + 1 < ntile;
        if (more) load_a(t + 1);                             // lands under this tile's MFMAs
        f32x16 acc[4][1];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][0][e] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const f32x4* A0 = As + c * A_SLOTS + a0_off;
            const f32x4* A1 = As + c * A_SLOTS + a1_off;
            const f32x4 d0 = A0[0], d1 = A1[0], d2 = A0[PLANE], d3 = A1[PLANE];
            const f32x4 d4 = A0[1], d5 = A1[1], d6 = A0[PLANE + 1], d7 = A1[PLANE + 1];
            const f32x4 t0 = s0 * d0 + s1 * d1;
            const f32x4 t1 = s0 * d2 + s1 * d3;
            const f32x4 t2 = s0 * d4 + s1 * d5;
            const f32x4 t3 = s0 * d6 + s1 * d7;
            f32x4 V[4];
            V[0] = t0 - t2; V[1] = t1 + t2; V[2] = t2 - t1; V[3] = t1 - t3;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[b][e], w[c][b][e], acc[b][0], 0, 0, 0);
        }
        wino_write_R<1>(Rs, acc, wa, lane);
        __syncthreads();                                     // R complete; every wave is done with this tile's halo image
        if (more) store_a();
        wino_store_Y<1>(Rs, p, img, oy0, (bx0 + t) * 16, 0, wa, lane);
        __syncthreads();                                     // next halo visible; Rs free again
    }
}

template <int CH>
static hipError_t launch_conv_wino_res_t(const ConvParams& p, hipStream_t s) {
    const int tiles_x = (p.out.w + 15) / 16, tiles_y = (p.out.h + 7) / 8;
    // tiles per workgroup: as many as keep >= ~2 workgroups per CU-slot in the launch
    int tpw = 16;
    while (tpw > 1 && (size_t)p.n * tiles_y * ((tiles_x + tpw - 1) / tpw) < 2048) tpw >>= 1;
    const int segs_x = (tiles_x + tpw - 1) / tpw;
    const size_t grid = (size_t)p.n * tiles_y * segs_x;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)CH * 4 * 10 * 12 * 16 + (size_t)4 * 2 * 32 * 32 * 4;
    hipLaunchKernelGGL((conv_wino_res_kernel<CH>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y, segs_x, tpw);
    return hipGetLastError();
}

int conv_wino_ntile(int cout) { return (cout % 64 == 0 || cout > 64) ? 64 : 32; }

hipError_t launch_conv_wino(const ConvParams& p, hipStream_t s) {
    if (p.resident && p.coutp == 32 && p.cin_chunks >= 1 && p.cin_chunks <= 4) {
        switch (p.cin_chunks) {
            case 1: return launch_conv_wino_res_t<1>(p, s);
            case 2: return launch_conv_wino_res_t<2>(p, s);
            case 3: return launch_conv_wino_res_t<3>(p, s);
            default: return launch_conv_wino_res_t<4>(p, s);
        }
    }
    return conv_wino_ntile(p.out.c) == 64 ? launch_conv_wino_t<2, 1>(p, s) : launch_conv_wino_t<1, 1>(p, s);
}

int conv_mfma_ntile(int cout) {
    if (cout % 128 == 0) return 128;
    if (cout % 64 == 0 || cout > 64) return 64;
    return 32;
}

bool conv_mfma_supported(const ConvParams& p) {
    const bool taps_ok = (p.R == 3 && p.S == 3) || (p.R == 2 && p.S == 2) || (p.R == 1 && p.S == 1) || (p.stride <= 1 && ((p.R == 2 && p.S == 1) || (p.R == 1 && p.S == 2)));
    const bool align_ok = (p.in.cs % 4 == 0) && (p.in.c % 4 == 0) && (((uintptr_t)p.in.p) % 16 == 0);
    return taps_ok && align_ok && p.in.c >= 8 && p.out.c >= 16;
}

template <int NT, int TW, int R, int S, int KCH = 1, int ST = 1>
static hipError_t launch_conv_mfma_t(const ConvParams& p, hipStream_t s) {
    constexpr int TH = 128 / TW;
    constexpr int BN = NT * 32;
    const int ext_h = p.convt ? p.in.h + p.convt_ext : p.out.h, ext_w = p.convt ? p.in.w + p.convt_ext : p.out.w;
    const int tiles_x = (ext_w + TW - 1) / TW, tiles_y = (ext_h + TH - 1) / TH;
    const int np_total = p.convt == 1 ? p.kT * p.kT * p.coutp : p.coutp;        // (convt == 2: one output phase per launch)
    const int nblk_n = np_total / BN;
    size_t lds = (size_t)KCH * (2 * ((TH - 1) * ST + R) * ((TW - 1) * ST + S) + R * S * 2 * BN) * 16;
    if (lds < 4 * 4096) lds = 4 * 4096;                       // the output stage needs a 4-KB exchange tile per wave
    const size_t grid = (p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * tiles_x * tiles_y) * nblk_n;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_mfma_kernel<NT, TW, R, S, KCH, ST>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y,
                       nblk_n, np_total);
    return hipGetLastError();
}

template <int NT, int TW>
static hipError_t launch_conv_mfma_rs(const ConvParams& p, hipStream_t s) {
    if (p.stride == 2 && !p.convt) {                          // strided halo gather (classifier stems, down-sampling convolutions)
        if (p.R == 3) return launch_conv_mfma_t<NT, TW, 3, 3, 1, 2>(p, s);
        if (p.R == 2) return launch_conv_mfma_t<NT, TW, 2, 2, 1, 2>(p, s);
        return launch_conv_mfma_t<NT, TW, 1, 1, 2, 2>(p, s);
    }
    if (p.R == 3) return launch_conv_mfma_t<NT, TW, 3, 3>(p, s);
    if (p.R == 2 && p.S == 2) return launch_conv_mfma_t<NT, TW, 2, 2>(p, s);
    if (p.R == 2 && p.S == 1) return launch_conv_mfma_t<NT, TW, 2, 1, 2>(p, s);     // (the two-tap phases of a 3x3 / stride-2 transposed convolution)
    if (p.R == 1 && p.S == 2) return launch_conv_mfma_t<NT, TW, 1, 2, 2>(p, s);
    return launch_conv_mfma_t<NT, TW, 1, 1, 2>(p, s);      // K-chunks per barrier: 1 / 2 / 4 / 8 measured 0 / +0.2 / -0.2 / -1.5 % (A/B)
}

hipError_t launch_conv_mfma(const ConvParams& p, hipStream_t s) {
    // transposed convolutions whose kT x kT phases x padded channels fit one 128-column tile (Cout <= 32 at 2x2): ONE workgroup
    // computes every output phase of its input tile - the tile is staged once instead of once per phase, and the per-thread
    // staging descriptors / prologue of these short-K layers are amortised over 2 - 4x the work
    const int np_t = p.convt == 1 ? p.kT * p.kT * p.coutp : 0;
    const int bn = (p.convt && np_t <= 128 && np_t % 32 == 0) ? np_t : conv_mfma_ntile(p.out.c);
    const bool wide = p.force_tw ? p.force_tw == 32 : (p.convt ? p.in.w : p.out.w) >= 32;
    if (bn == 128) return wide ? launch_conv_mfma_rs<4, 32>(p, s) : launch_conv_mfma_rs<4, 16>(p, s);
    if (bn == 64) return wide ? launch_conv_mfma_rs<2, 32>(p, s) : launch_conv_mfma_rs<2, 16>(p, s);
    return wide ? launch_conv_mfma_rs<1, 32>(p, s) : launch_conv_mfma_rs<1, 16>(p, s);
}

// ------------------------------------------------------------------------------------------------------------
// Direct kernels (HBM-bound layers: first conv with Cin <= 4, the 1x1 head, and generic fall-backs)
// ------------------------------------------------------------------------------------------------------------

// Cin <= 4, Cout % 4 == 0: thread = (pixel, 4 output channels); a wave writes 1 KiB contiguous.
__global__ __launch_bounds__(256) void conv_small_cin_kernel(TView in, TView out, const float* __restrict__ w,
                                                             const float* __restrict__ bias, size_t total, int R, int S,
                                                             int pad_top, int pad_left, int act, float alpha) {
    const int quads = out.c >> 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(t % quads);
        size_t pix = t / quads;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bias) acc = *reinterpret_cast<const f32x4*>(bias + q * 4);
        for (int r = 0; r < R; ++r) {
            const int iy = oy - pad_top + r;
            if (iy < 0 || iy >= in.h) continue;
            for (int s = 0; s < S; ++s) {
                const int ix = ox - pad_left + s;
                if (ix < 0 || ix >= in.w) continue;
                const float* ip = in.p + ((img * in.h + iy) * in.w + ix) * in.cs;
                for (int ci = 0; ci < in.c; ++ci) {
                    const float v = ip[ci];
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ((size_t)((r * S + s) * in.c + ci)) * out.c + q * 4);
                    acc[0] = fmaf(v, wv[0], acc[0]); acc[1] = fmaf(v, wv[1], acc[1]);
                    acc[2] = fmaf(v, wv[2], acc[2]); acc[3] = fmaf(v, wv[3], acc[3]);
                }
            }
        }
        float* op = out.p + ((img * out.h + oy) * out.w + ox) * out.cs + q * 4;
        f32x4 o;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
        o = apply_act4(o, act, alpha);
        *reinterpret_cast<f32x4*>(op) = o;
    }
}

// First layer of the U-Net (Cin = 1, 3x3, 4.4 flop/B: HBM-bound on its output).  Thread = 4 output channels x 8 consecutive
// ROWS of one pixel column: consecutive lanes are the channel quads of a pixel, then the next pixel of the row, so that every
// store instruction of a wave writes 1 KiB of CONTIGUOUS output (16 pixels x 64 B at 16 channels, 4 pixels x 256 B at 64) -
// with the 8 pixels of a thread along the row instead (rounds 1-2) a store instruction wrote 16 separate 64-byte pieces
// 512 bytes apart at 16 channels and the layer ran at 2.1 - 2.5 TB/s.  The nine filter taps live in registers; the 10 x 3
// input window streams through three registers per row (lanes of one pixel read the same address: one broadcast fetch).
__global__ __launch_bounds__(256) void conv_first_kernel(TView in, TView out, const float* __restrict__ w,
                                                         const float* __restrict__ bias, size_t total, int pad_top,
                                                         int pad_left, int act, float alpha) {
    constexpr int PY = 8;
    const int quads = out.c >> 2;
    const int gy = (out.h + PY - 1) / PY;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(t % quads);
        size_t g = t / quads;
        const int ox = (int)(g % out.w); g /= out.w;
        const int y0 = (int)(g % gy) * PY;
        const size_t img = g / gy;
        f32x4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(w + (size_t)k * out.c + q * 4);
        f32x4 acc[PY];
        const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PY; ++i) acc[i] = b4;
        const float* col = in.p + (img * in.h * in.w) * in.cs;
#pragma unroll
        for (int r = 0; r < PY + 2; ++r) {                   // input row r of the window feeds output rows r - 2 .. r
            const int iy = y0 - pad_top + r;
            float v[3];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int ix = ox - pad_left + sx;
                v[sx] = (iy >= 0 && iy < in.h && ix >= 0 && ix < in.w) ? col[((size_t)iy * in.w + ix) * in.cs] : 0.f;
            }
#pragma unroll
            for (int kr = 0; kr < 3; ++kr) {
                const int i = r - kr;                         // output row of the thread that sees this input row as tap row kr
                if (i < 0 || i >= PY) continue;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const f32x4 k4 = wt[kr * 3 + sx];
                    acc[i][0] = fmaf(v[sx], k4[0], acc[i][0]); acc[i][1] = fmaf(v[sx], k4[1], acc[i][1]);
                    acc[i][2] = fmaf(v[sx], k4[2], acc[i][2]); acc[i][3] = fmaf(v[sx], k4[3], acc[i][3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            if (y0 + i < out.h) {
                f32x4 o;
                o[0] = acc[i][0]; o[1] = acc[i][1]; o[2] = acc[i][2]; o[3] = acc[i][3];
                o = apply_act4(o, act, alpha);
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(out.p + ((img * out.h + y0 + i) * out.w + ox) * out.cs + q * 4));
            }
        }
    }
}

hipError_t launch_conv_small_cin(const TView& in, const TView& out, const float* w, const float* bias, int n, int R, int S,
                                 int pad_top, int pad_left, int act, float alpha, hipStream_t s) {
    if (in.c == 1 && R == 3 && S == 3) {
        const size_t tot = (size_t)n * ((out.h + 7) / 8) * out.w * (out.c / 4);
        if (!tot) return hipSuccess;
        const unsigned g = (unsigned)((tot + 255) / 256 > 65536 * 16 ? 65536 * 16 : (tot + 255) / 256);
        hipLaunchKernelGGL(conv_first_kernel, dim3(g), dim3(256), 0, s, in, out, w, bias, tot, pad_top, pad_left, act, alpha);
        return hipGetLastError();
    }
    const size_t total = (size_t)n * out.h * out.w * (out.c / 4);
    if (!total) return hipSuccess;
    const unsigned grid = (unsigned)((total + 255) / 256 > 65536 * 16 ? 65536 * 16 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_small_cin_kernel, dim3(grid), dim3(256), 0, s, in, out, w, bias, total, R, S, pad_top,
                       pad_left, act, alpha);
    return hipGetLastError();
}

// 1x1 conv with few outputs (the 4-class head) + optional channel softmax.  LPP lanes share one pixel (16 at Cin >= 64,
// 8 / 4 for narrower inputs - with 16 lanes on a 16-channel pixel twelve of them idled and the layer ran at 1 TB/s): each
// lane reads float4 slices of the pixel's channel vector (256 B contiguous per pixel at Cin = 64), partial sums are
// reduced with wave shuffles, lane 0 of the group finishes bias / softmax and writes the pixel.
template <int COUT, int LPP>
__global__ __launch_bounds__(256) void conv_head_kernel(TView in, TView out, const float* __restrict__ w,
                                                        const float* __restrict__ bias, size_t npix, int act, float alpha) {
    constexpr int PPW = 64 / LPP;                              // pixels per wave
    const int sub = threadIdx.x & (LPP - 1);
    const int c4n = in.c >> 2;
    for (size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPP; ; pix += ((size_t)gridDim.x * blockDim.x) / LPP) {
        // all LPP lanes of a group share `pix`; groups of one wave may run out at different times, so keep the whole
        // wave in the loop and mask the work instead of breaking (shuffles need every lane present)
        const size_t wave_first = pix - (((size_t)threadIdx.x & 63) / LPP);
        if (wave_first >= npix) break;
        (void)PPW;
        const bool live = pix < npix;
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
        if (live) {
            const float* ip = in.p + pix * in.cs;
            for (int c4 = sub; c4 < c4n; c4 += LPP) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ip + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* wr = w + (size_t)(c4 * 4 + e) * COUT;
#pragma unroll
                    for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v[e], wr[o], acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float v = acc[o];
#pragma unroll
            for (int d = LPP / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            acc[o] = v;
        }
        if (live && sub == 0) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] += bias ? bias[o] : 0.f;
            if (act == ECSEG_ACT_SOFTMAX) {
                float m = acc[0];
#pragma unroll
                for (int o = 1; o < COUT; ++o) m = fmaxf(m, acc[o]);
                float sum = 0.f;
#pragma unroll
                for (int o = 0; o < COUT; ++o) { acc[o] = expf(acc[o] - m); sum += acc[o]; }
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = acc[o] / sum;
            } else {
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = apply_act(acc[o], act, alpha);
            }
            float* op = out.p + pix * out.cs;
#pragma unroll
            for (int o = 0; o < COUT; ++o) op[o] = acc[o];
        }
    }
}

template <int LPP>
static hipError_t launch_conv_head_l(const TView& in, const TView& out, const float* w, const float* bias, size_t npix, int act,
                                     float alpha, hipStream_t s) {
    size_t blocks = (npix * LPP + 255) / 256;
    if (blocks > 65536 * 8) blocks = 65536 * 8;
#define HEAD_CASE(C) case C: hipLaunchKernelGGL((conv_head_kernel<C, LPP>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, w, bias, npix, act, alpha); break;
    switch (out.c) {
        HEAD_CASE(1) HEAD_CASE(2) HEAD_CASE(3) HEAD_CASE(4) HEAD_CASE(5) HEAD_CASE(6) HEAD_CASE(7) HEAD_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef HEAD_CASE
    return hipGetLastError();
}

hipError_t launch_conv_head(const TView& in, const TView& out, const float* w, const float* bias, int n, int act,
                            float alpha, hipStream_t s) {
    const size_t npix = (size_t)n * out.h * out.w;
    if (!npix) return hipSuccess;
    const int c4n = in.c >> 2;                               // lanes per pixel: as many as there are 4-channel slices, up to 16
    if (c4n >= 16) return launch_conv_head_l<16>(in, out, w, bias, npix, act, alpha, s);
    if (c4n >= 8) return launch_conv_head_l<8>(in, out, w, bias, npix, act, alpha, s);
    return launch_conv_head_l<4>(in, out, w, bias, npix, act, alpha, s);
}

// Generic direct convolution: thread = (pixel, output channel).  Correctness fall-back for shapes the MFMA kernel
// does not take (odd channel counts, large taps).
__global__ __launch_bounds__(256) void conv_generic_kernel(TView in, TView out, const float* __restrict__ w,
                                                           const float* __restrict__ bias, size_t total, int R, int S,
                                                           int stride, int pad_top, int pad_left, int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float acc = bias ? bias[co] : 0.f;
        for (int r = 0; r < R; ++r) {
            const int iy = oy * stride - pad_top + r;
            if (iy < 0 || iy >= in.h) continue;
            for (int s = 0; s < S; ++s) {
                const int ix = ox * stride - pad_left + s;
                if (ix < 0 || ix >= in.w) continue;
                const float* ip = in.p + ((img * in.h + iy) * in.w + ix) * in.cs;
                const float* wp = w + (size_t)(r * S + s) * in.c * out.c + co;
                for (int ci = 0; ci < in.c; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * out.c], acc);
            }
        }
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + co] = apply_act(acc, act, alpha);
    }
}

hipError_t launch_conv_generic(const TView& in, const TView& out, const float* w, const float* bias, int n, int R, int S,
                               int stride, int pad_top, int pad_left, int act, float alpha, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65536 * 16) blocks = 65536 * 16;
    hipLaunchKernelGGL(conv_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, w, bias, total, R, S, stride,
                       pad_top, pad_left, act, alpha);
    return hipGetLastError();
}

// Generic transposed convolution (Keras kernel layout (kh, kw, out, in)): thread = (output pixel, output channel),
// gathers every (input pixel, tap) pair that lands on it.
__global__ __launch_bounds__(256) void convt_generic_kernel(TView in, TView out, const float* __restrict__ w,
                                                            const float* __restrict__ bias, size_t total, int R, int S,
                                                            int stride, int crop_top, int crop_left, int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float acc = bias ? bias[co] : 0.f;
        const int fy = oy + crop_top, fx = ox + crop_left;   // coordinates in the uncropped output
        for (int a = 0; a < R; ++a) {
            const int ny = fy - a;
            if (ny < 0 || ny % stride) continue;
            const int iy = ny / stride;
            if (iy >= in.h) continue;
            for (int b = 0; b < S; ++b) {
                const int nx = fx - b;
                if (nx < 0 || nx % stride) continue;
                const int ix = nx / stride;
                if (ix >= in.w) continue;
                const float* ip = in.p + ((img * in.h + iy) * in.w + ix) * in.cs;
                const float* wp = w + ((size_t)(a * S + b) * out.c + co) * in.c;
                for (int ci = 0; ci < in.c; ++ci) acc = fmaf(ip[ci], wp[ci], acc);
            }
        }
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + co] = apply_act(acc, act, alpha);
    }
}

hipError_t launch_convt_generic(const TView& in, const TView& out, const float* w, const float* bias, int n, int R, int S,
                                int stride, int crop_top, int crop_left, int act, float alpha, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65536 * 16) blocks = 65536 * 16;
    hipLaunchKernelGGL(convt_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, w, bias, total, R, S, stride,
                       crop_top, crop_left, act, alpha);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Element-wise / resampling kernels (HBM-bound; float4 fast path when every view is 16-byte aligned)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool view_vec4(const TView& v) {
    return (v.c % 4 == 0) && (v.cs % 4 == 0) && ((((uintptr_t)v.p) & 15) == 0);
}

// MaxPooling2D (mode 0) / AveragePooling2D (mode 1), 'valid'
template <bool VEC>
__global__ __launch_bounds__(256) void maxpool_kernel(TView in, TView out, size_t total, int kh, int kw, int stride, int mode) {
    constexpr int V = VEC ? 4 : 1;
    const int cq = out.c / V;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % cq) * V;
        size_t pix = t / cq;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = mode ? 0.f : -INFINITY;
        for (int r = 0; r < kh; ++r)
            for (int s = 0; s < kw; ++s) {
                const float* ip = in.p + ((img * in.h + (oy * stride + r)) * in.w + (ox * stride + s)) * in.cs + c;
                if (VEC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ip);
#pragma unroll
                    for (int e = 0; e < V; ++e) m[e] = mode ? m[e] + v[e] : fmaxf(m[e], v[e]);
                } else {
                    m[0] = mode ? m[0] + ip[0] : fmaxf(m[0], ip[0]);
                }
            }
        if (mode) {
            const float inv = 1.f / (float)(kh * kw);
#pragma unroll
            for (int e = 0; e < V; ++e) m[e] *= inv;
        }
        float* op = out.p + ((img * out.h + oy) * out.w + ox) * out.cs + c;
        if (VEC) {
            f32x4 o; o[0] = m[0]; o[1] = m[V > 1 ? 1 : 0]; o[2] = m[V > 2 ? 2 : 0]; o[3] = m[V > 3 ? 3 : 0];
            *reinterpret_cast<f32x4*>(op) = o;
        } else {
            op[0] = m[0];
        }
    }
}

static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 65536 * 16) b = 65536 * 16;
    return (unsigned)(b ? b : 1);
}

hipError_t launch_maxpool(const TView& in, const TView& out, int n, int kh, int kw, int stride, int mode, hipStream_t s) {
    const bool vec = (in.c % 4 == 0) && (in.cs % 4 == 0) && (out.cs % 4 == 0) && ((((uintptr_t)in.p) | ((uintptr_t)out.p)) & 15) == 0;
    const size_t total = (size_t)n * out.h * out.w * (vec ? out.c / 4 : out.c);
    if (!total) return hipSuccess;
    if (vec) hipLaunchKernelGGL(maxpool_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, in, out, total, kh, kw, stride, mode);
    else hipLaunchKernelGGL(maxpool_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, in, out, total, kh, kw, stride, mode);
    return hipGetLastError();
}

// GlobalAveragePooling2D (mode 1) / GlobalMaxPooling2D (mode 0): workgroup = (patch, 64 channels); lane = channel
// (coalesced 256-B rows), the four waves split the pixels and meet in LDS.  Fixed summation order: run-to-run identical.
__global__ __launch_bounds__(256) void global_pool_kernel(TView in, TView out, int mode) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const size_t img = blockIdx.y;
    const int npx = in.h * in.w;
    float acc = mode ? 0.f : -INFINITY;
    if (c < in.c) {
        const float* ip = in.p + img * (size_t)npx * in.cs + c;
        for (int q = wave; q < npx; q += 4) {
            const float v = ip[(size_t)q * in.cs];
            acc = mode ? acc + v : fmaxf(acc, v);
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < in.c) {
        float r = part[0][lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) r = mode ? r + part[k][lane] : fmaxf(r, part[k][lane]);
        if (mode) r *= 1.f / (float)npx;
        out.p[img * (size_t)out.cs + c] = r;
    }
}

hipError_t launch_global_pool(const TView& in, const TView& out, int n, int mode, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(global_pool_kernel, dim3((unsigned)((in.c + 63) / 64), (unsigned)n), dim3(256), 0, s, in, out, mode);
    return hipGetLastError();
}

template <bool VEC>
__global__ __launch_bounds__(256) void upsample_kernel(TView in, TView out, size_t total, int f, int mode) {
    constexpr int V = VEC ? 4 : 1;
    const int cq = out.c / V;
    const float inv = 1.f / (float)f;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % cq) * V;
        size_t pix = t / cq;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float o[V];
        if (mode == 0) {
            const float* ip = in.p + ((img * in.h + oy / f) * in.w + ox / f) * in.cs + c;
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = ip[e];
        } else {
            // half-pixel centres: src = (dst + 0.5) / f - 0.5, edges clamped
            const float sy = ((float)oy + 0.5f) * inv - 0.5f, sx = ((float)ox + 0.5f) * inv - 0.5f;
            const float fy = floorf(sy), fx = floorf(sx);
            const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), in.h - 1);
            const int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), in.w - 1);
            const float ly = sy - fy, lx = sx - fx;
            const float* r0 = in.p + (img * in.h + y0) * (size_t)in.w * in.cs + c;
            const float* r1 = in.p + (img * in.h + min(y1, in.h - 1)) * (size_t)in.w * in.cs + c;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float tl = r0[(size_t)x0 * in.cs + e], tr = r0[(size_t)x1 * in.cs + e];
                const float bl = r1[(size_t)x0 * in.cs + e], br = r1[(size_t)x1 * in.cs + e];
                const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
                o[e] = top + (bot - top) * ly;
            }
        }
        float* op = out.p + ((img * out.h + oy) * out.w + ox) * out.cs + c;
#pragma unroll
        for (int e = 0; e < V; ++e) op[e] = o[e];
    }
}

hipError_t launch_upsample(const TView& in, const TView& out, int n, int factor, int mode, hipStream_t s) {
    const bool vec = (in.c % 4 == 0);
    const size_t total = (size_t)n * out.h * out.w * (vec ? out.c / 4 : out.c);
    if (!total) return hipSuccess;
    if (vec) hipLaunchKernelGGL(upsample_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, in, out, total, factor, mode);
    else hipLaunchKernelGGL(upsample_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, in, out, total, factor, mode);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void affine_kernel(TView in, TView out, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, size_t total, int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        const size_t pix = t / out.c;
        float v = in.p[pix * in.cs + c];
        if (scale) v = v * scale[c] + shift[c];
        out.p[pix * out.cs + c] = apply_act_ext(v, act, alpha);
    }
}

hipError_t launch_affine(const TView& in, const TView& out, const float* scale, const float* shift, int n, int act,
                         float alpha, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(affine_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, scale, shift, total, act, alpha);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_kernel(TView a, TView b, TView out, size_t total, int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        const size_t pix = t / out.c;
        out.p[pix * out.cs + c] = apply_act_ext(a.p[pix * a.cs + c] + b.p[pix * b.cs + c], act, alpha);
    }
}

hipError_t launch_add(const TView& a, const TView& b, const TView& out, int n, int act, float alpha, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, b, out, total, act, alpha);
    return hipGetLastError();
}

// y[oy][ox] = x[oy - off_y][ox - off_x] or 0 outside: ZeroPadding2D (off > 0), Cropping2D (off < 0), plain copy (0)
__global__ __launch_bounds__(256) void copy_kernel(TView in, TView out, size_t total, int off_y, int off_x) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        const int iy = oy - off_y, ix = ox - off_x;
        float v = 0.f;
        if (iy >= 0 && iy < in.h && ix >= 0 && ix < in.w) v = in.p[((img * in.h + iy) * in.w + ix) * in.cs + c];
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + c] = v;
    }
}

hipError_t launch_copy(const TView& in, const TView& out, int n, int off_y, int off_x, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, total, off_y, off_x);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void softmax_kernel(TView in, TView out, size_t npix) {
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
        const float* ip = in.p + pix * in.cs;
        float* op = out.p + pix * out.cs;
        float m = ip[0];
        for (int c = 1; c < in.c; ++c) m = fmaxf(m, ip[c]);
        float sum = 0.f;
        for (int c = 0; c < in.c; ++c) sum += expf(ip[c] - m);
        for (int c = 0; c < in.c; ++c) op[c] = expf(ip[c] - m) / sum;
    }
}

hipError_t launch_softmax(const TView& in, const TView& out, int n, hipStream_t s) {
    const size_t npix = (size_t)n * out.h * out.w;
    if (!npix) return hipSuccess;
    hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(npix)), dim3(256), 0, s, in, out, npix);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t count) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (size_t)gridDim.x * blockDim.x)
        out[t] = (float)in[t];
}

hipError_t launch_u8_to_f32(const uint8_t* in, float* out, size_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(count)), dim3(256), 0, s, in, out, count);
    return hipGetLastError();
}

// im2patches_overlap (reference src/image_tools.py:181-184): patch k of image i = gray[i, y0:y0+256, x0:x0+256]
__global__ __launch_bounds__(256) void tile_patches_kernel(const uint8_t* __restrict__ gray, int H, int W,
                                                           const int32_t* __restrict__ pos, int n_pos,
                                                           float* __restrict__ out, size_t total) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(t & 255), y = (int)((t >> 8) & 255);
        const size_t pk = t >> 16;
        const int k = (int)(pk % n_pos);
        const size_t img = pk / n_pos;
        out[t] = (float)gray[(img * H + pos[2 * k] + y) * W + pos[2 * k + 1] + x];
    }
}

hipError_t launch_tile_patches(const uint8_t* gray, int n_img, int H, int W, const int32_t* pos_yx, int n_pos, float* out,
                               hipStream_t s) {
    const size_t total = (size_t)n_img * n_pos * 65536;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(tile_patches_kernel, dim3(grid_for(total)), dim3(256), 0, s, gray, H, W, pos_yx, n_pos, out, total);
    return hipGetLastError();
}

}  // namespace ecseg
