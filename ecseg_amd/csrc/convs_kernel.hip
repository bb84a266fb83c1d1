// One-tap implicit GEMM on the BF16 matrix pipe with 3-way split operands, float32-accurate (round 6; option "winograd" = 3): the
// 2x2 / stride-2 transposed convolutions of the U-Net decoder (kT x kT phases x Cout columns, K = Cin) and any 1x1 convolution.
//
// Unlike the fused Winograd kernel (wino4s_kernel.hip), where every wave transforms and splits its own operand and the split costs
// 5.5 vector instructions per value and per 32 output channels, a plain GEMM shares its A tile through LDS: a pixel x channel value is
// split ONCE per workgroup, when the tile is staged (global -> registers -> split -> three bf16 planes in LDS), and then read by every
// wave as ready 16-byte MFMA operands.  The filter is split at load (launch_convs_filter, from the fp32 image of relayout_convt /
// relayout_conv).  Per 16-channel chunk and 32-column tile: six v_mfma_f32_32x32x16_bf16 (a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 -
// everything above 2^-24 relative) = 192 pipe cycles where the fp32 kernel's eight 32x32x2 take 512.
//
// Workgroup = 4 waves = 128 pixels (4 x 32 or 8 x 16 tile of the INPUT extent) x BN = NT x 32 columns, as conv_mfma_kernel; the
// output stage (bias, activation, scatter of the kT x kT phases, 16-byte stores) is the same.
#include "common.h"
#include "device_util.h"

namespace ecseg {

namespace {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bfloat(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(fbits(hi), fbits(lo), 0x07060302u); }
}  // namespace

// LDS: As[piece 3][k-half 2][128 pixels] x 16 B (8 channels bf16) | Bs[piece 3][k-half 2][BN columns] x 16 B
template <int NT, int TW>
__global__ __launch_bounds__(256) void convs_kernel(ConvParams p, int tiles_x, int tiles_y, int nblk_n, int np_total) {
    constexpr int TH = 128 / TW;
    constexpr int BN = NT * 32;
    constexpr int B_SLOTS = 3 * 2 * BN;                      // 16-byte slots of one filter chunk
    constexpr int B_PER_T = (B_SLOTS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);              // [3][2][128]
    u32x4* Bs = As + 3 * 2 * 128;                            // [3][2][BN]

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

    // ---- staging: thread = (pixel, channel quad) twice: quads q and q + 2 of the 16-channel chunk (q = tid & 1) ----
    const int s_pix = tid >> 1, s_q = tid & 1;
    const int s_py = TW == 32 ? s_pix >> 5 : s_pix >> 4, s_px = TW == 32 ? s_pix & 31 : s_pix & 15;
    const int s_iy = ty0 + s_py, s_ix = tx0 + s_px;
    const bool s_in = s_iy < Hin && s_ix < Win;
    const float* s_src = p.in.p + in_img + ((size_t)s_iy * Win + s_ix) * p.in.cs;
    const int nchunks = (Cin + 15) / 16;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.wt);      // [chunk][piece][k-half][np_total] x 16 B
    f32x4 a_reg[2];
    u32x4 b_reg[B_PER_T];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = c * 16 + 4 * (s_q + 2 * k);
            a_reg[k] = (s_in && ch < Cin) ? *reinterpret_cast<const f32x4*>(s_src + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const int q = tid + k * 256;
            if (q < B_SLOTS) {
                const int pk = q / BN, j = q - pk * BN;           // pk = piece * 2 + k-half
                b_reg[k] = wsrc[((size_t)c * 6 + pk) * np_total + n0 + j];
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // exact split by truncation: v1 = high half, v2 = high half of (v - v1), v3 = the rest (8 bits: exact)
            const f32x4 v = a_reg[k];
            float r[4], s[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = v[c] - bfloat(fbits(v[c]) & 0xffff0000u);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = r[c] - bfloat(fbits(r[c]) & 0xffff0000u);
            const int quad = s_q + 2 * k, kh = quad >> 1, half = quad & 1;      // channels 4 quad .. of the chunk: k-half kh, 8-byte half `half`
            u32x2* d = reinterpret_cast<u32x2*>(As + kh * 128 + s_pix) + half;
            d[0 * 2 * 2 * 128] = u32x2{pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
            d[1 * 2 * 2 * 128] = u32x2{pack_hi(r[0], r[1]), pack_hi(r[2], r[3])};
            d[2 * 2 * 2 * 128] = u32x2{pack_hi(s[0], s[1]), pack_hi(s[2], s[3])};
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const int q = tid + k * 256;
            if (q < B_SLOTS) Bs[q] = b_reg[k];
        }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const u32x4* Ap = As + lh * 128 + wave * 32 + li;        // the wave's 32 pixels: staging order = (row, column) of the tile = MFMA row order
    const u32x4* Bp = Bs + lh * BN + li;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, Ap[0 * 256]), a2 = __builtin_bit_cast(bf16x8, Ap[1 * 256]), a3 = __builtin_bit_cast(bf16x8, Ap[2 * 256]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bp[0 * 2 * BN + nt * 32]), b2 = __builtin_bit_cast(bf16x8, Bp[1 * 2 * BN + nt * 32]),
                         b3 = __builtin_bit_cast(bf16x8, Bp[2 * 2 * BN + nt * 32]);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[nt], 0, 0, 0);     // small terms first
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[nt], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- output stage (conv_mfma_kernel's): a wave turns its 32 x 32 accumulator tile around through a private 4-KB LDS tile ----
    const int Hout = p.out.h, Wout = p.out.w, Cout = p.out.c;
    const int Ht = p.convt ? Hin + p.convt_ext : Hout, Wt = p.convt ? Win + p.convt_ext : Wout;
    float* Xs = reinterpret_cast<float*>(smem) + wave * 1024;
    const bool vec_ok = (p.out.cs % 4 == 0) && (Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out.p) & 15) == 0);
    const int quad = lane & 7;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int co = n0 + nt * 32 + 4 * quad, oa = 0, ob = 0;
        if (p.convt) {
            const int ab = co / p.coutp;
            co -= ab * p.coutp;
            oa = ab / p.kT + p.phase_a; ob = ab - (ab / p.kT) * p.kT + p.phase_b;
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
            const int row = (lane >> 3) + 8 * r;
            f32x4 v = *reinterpret_cast<const f32x4*>(Xs + row * 32 + 4 * quad) + bv;
            v = apply_act4(v, p.act, p.alpha);
            int py, px;
            if (TW == 32) { py = wave; px = row; } else { py = 2 * wave + (row >> 4); px = row & 15; }
            const int tyy = ty0 + py, txx = tx0 + px;
            int oy = tyy, ox = txx;
            bool ok = tyy < Ht && txx < Wt;
            if (p.convt) {
                oy = tyy * p.kT + oa - p.crop_top;
                ox = txx * p.kT + ob - p.crop_left;
                ok = ok && oy >= 0 && oy < Hout && ox >= 0 && ox < Wout;
            }
            if (ok) {
                float* o = p.out.p + (((size_t)img * Hout + oy) * Wout + ox) * p.out.cs + co;
                if (vec_ok && co + 3 < Cout) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (co + c < Cout) o[c] = v[c];
                }
            }
        }
    }
}

// fp32 one-tap filter image (relayout_convt / relayout_conv with R = S = 1: [chunk of 8][half][np][4 ch], padded pitch) -> the split
// image [chunk of 16][piece][k-half][np] x 8 bf16; one thread per (chunk, k-half, column)
__global__ __launch_bounds__(256) void convs_filter_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int chunks8, int chunks16,
                                                           int np, long chunk_pitch) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)chunks16 * 2 * np) return;
    const int n = (int)(id % np);
    const int kh = (int)((id / np) & 1), c16 = (int)(id / (2l * np));
    const int c8 = 2 * c16 + kh;
    unsigned short o[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float u = c8 < chunks8 ? src[(long)c8 * chunk_pitch + ((long)(e >> 2) * np + n) * 4 + (e & 3)] : 0.f;
        const unsigned b1 = fbits(u) & 0xffff0000u;
        const float r1 = u - bfloat(b1);
        const unsigned b2 = fbits(r1) & 0xffff0000u;
        const float r2 = r1 - bfloat(b2);
        o[0][e] = (unsigned short)(b1 >> 16); o[1][e] = (unsigned short)(b2 >> 16); o[2][e] = (unsigned short)(fbits(r2) >> 16);
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        unsigned short* d = dst + ((((long)c16 * 3 + pc) * 2 + kh) * np + n) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = o[pc][e];
    }
}

size_t convs_image_bytes(int cin, int np) { return (size_t)((cin + 15) / 16) * 6 * np * 16; }

hipError_t launch_convs_filter(const float* wt_fp32, void* dst, int cin, int np, hipStream_t s) {
    const int chunks8 = (cin + 7) / 8, chunks16 = (cin + 15) / 16;
    const long total = (long)chunks16 * 2 * np;
    hipLaunchKernelGGL(convs_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wt_fp32, reinterpret_cast<unsigned short*>(dst), chunks8,
                       chunks16, np, wt_chunk_pitch(np));
    return hipGetLastError();
}

// one tap, stride 1, whole 16-byte channel quads, >= 32 columns per phase block (the narrow decoder steps keep conv_mfma_kernel's
// all-phases-in-one-tile form)
bool convs_supported(const ConvParams& p) {
    const int np = p.convt == 1 ? p.kT * p.kT * p.coutp : p.coutp;
    return p.R == 1 && p.S == 1 && p.stride <= 1 && p.in.c >= 16 && p.in.c % 4 == 0 && p.in.cs % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(p.in.p) & 15) == 0 && p.coutp % 32 == 0 && np % 32 == 0 && p.pad_top == 0 && p.pad_left == 0;
}

template <int NT, int TW>
static hipError_t launch_convs_t(const ConvParams& p, hipStream_t s) {
    constexpr int TH = 128 / TW, BN = NT * 32;
    const int ext_h = p.convt ? p.in.h + p.convt_ext : p.out.h, ext_w = p.convt ? p.in.w + p.convt_ext : p.out.w;
    const int tiles_x = (ext_w + TW - 1) / TW, tiles_y = (ext_h + TH - 1) / TH;
    const int np_total = p.convt == 1 ? p.kT * p.kT * p.coutp : p.coutp;
    const int nblk_n = np_total / BN;
    size_t lds = (size_t)(3 * 2 * 128 + 3 * 2 * BN) * 16;
    if (lds < 4 * 4096) lds = 4 * 4096;
    const size_t grid = (p.lut != nullptr ? (size_t)(p.n / p.per_image) * p.lut_len : (size_t)p.n * tiles_x * tiles_y) * nblk_n;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((convs_kernel<NT, TW>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y, nblk_n, np_total);
    return hipGetLastError();
}

hipError_t launch_convs(const ConvParams& p, hipStream_t s) {
    if (!convs_supported(p)) return hipErrorInvalidValue;
    const int np_total = p.convt == 1 ? p.kT * p.kT * p.coutp : p.coutp;
    const bool wide = p.force_tw ? p.force_tw == 32 : (p.convt ? p.in.w : p.out.w) >= 32;
    if (np_total % 128 == 0) return wide ? launch_convs_t<4, 32>(p, s) : launch_convs_t<4, 16>(p, s);
    if (np_total % 64 == 0) return wide ? launch_convs_t<2, 32>(p, s) : launch_convs_t<2, 16>(p, s);
    return wide ? launch_convs_t<1, 32>(p, s) : launch_convs_t<1, 16>(p, s);
}

}  // namespace ecseg
