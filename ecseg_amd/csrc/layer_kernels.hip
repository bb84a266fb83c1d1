// Kernels of the wider Keras layer vocabulary (round 5): whatever metaseg.h5 / interseg_models/* turn out to contain
// must load (reference src/utils.py:27-33, src/interseg.py:96-98 call tf.keras.models.load_model on an unknown file).
//
//   conv_mfma_tap_kernel   Conv2D with ANY taps / stride / dilation_rate on the fp32 matrix cores (implicit GEMM, the tile of
//                          every tap staged on its own - dilated taps share no halo worth keeping)
//   dwconv_kernel          DepthwiseConv2D / the depthwise half of SeparableConv2D / Conv2D(groups = Cin): HBM-bound,
//                          16-byte channel-quad accesses, LDS-staged halo
//   binary_kernel          Add / Multiply / Subtract / Maximum / Minimum with broadcasting (squeeze-and-excite x * s)
//   prelu_kernel, layernorm_kernel, pool_pad_kernel ('same' pooling)
// NHWC float32 views everywhere (common.h: TView).
#include "common.h"
#include "device_util.h"

namespace ecseg {

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, tap by tap.  Workgroup = 128 output pixels (TH x TW tile of one patch) x BN output channels,
// as conv_mfma_kernel (unet_kernels.hip), but the K loop walks (tap, group of KCH 8-channel chunks) and stages, per step,
// the 128 input pixels THAT tap reads: pixel (oy * stride - pad + r * dil, ox * stride - pad + s * dil).  With a dilation
// rate of 6 - 18 (ASPP heads) the union of the taps' footprints is (TH + 2 dil) x (TW + 2 dil) pixels of which every tap
// uses 128: a shared halo would not fit LDS and would be read once per tap anyway.  Input re-reads (R * S times) are served
// by L2; the MFMA work per staged byte equals the 1x1 kernel's.
//
//   M = output pixels (lane & 31 -> pixel of the wave's 32-pixel strip), N = output channels, K = (tap, input channel);
//   one MFMA (v_mfma_f32_32x32x2_f32) consumes channels {e, 4 + e} of a chunk, so one ds_read_b128 per operand feeds four.
// LDS: As[kc][half][128 pixels][4 ch], Bs[kc][half][BN][4 ch]; global filter layout = relayout_conv (api.hip):
// wt[tap][chunk][half][N padded][4].
// ------------------------------------------------------------------------------------------------------------
template <int NT, int TW>
__global__ __launch_bounds__(256) void conv_mfma_tap_kernel(ConvParams p, int tiles_x, int tiles_y, int nblk_n, int dil) {
    constexpr int KCH = 4;
    constexpr int BN = NT * 32;
    constexpr int B_PER_T = (2 * BN * KCH) / 256;            // = NT

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* As = reinterpret_cast<f32x4*>(smem);              // [KCH][2][128]
    f32x4* Bs = As + KCH * 2 * 128;                          // [KCH][2][BN]

    const int tid = threadIdx.x;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % nblk_n; bid /= nblk_n;
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * (128 / TW); bid /= tiles_y;
    const int img = bid;
    const int n0 = nb * BN;
    const int Hin = p.in.h, Win = p.in.w, Cin = p.in.c;
    const int st = p.stride > 0 ? p.stride : 1;
    const size_t in_img = (size_t)img * Hin * Win * p.in.cs;

    // this thread's A piece: pixel a_pix of the tile, channel half a_h, for every one of the KCH chunks of a step
    const int a_pix = tid >> 1, a_h = tid & 1;
    const int a_iy0 = (ty0 + a_pix / TW) * st - p.pad_top, a_ix0 = (tx0 + a_pix % TW) * st - p.pad_left;
    const size_t chunk_stride = (size_t)p.wt_chunk_stride, tap_stride = (size_t)p.wt_tap_stride;
    // B pieces: q = tid + k * 256 over [kc][half][BN]
    const int ngrp = (p.cin_chunks + KCH - 1) / KCH;
    const int nsteps = p.R * p.S * ngrp;

    f32x4 a_reg[KCH], b_reg[B_PER_T];
    auto load_step = [&](int step) {
        const int tap = step / ngrp, grp = step - tap * ngrp;
        const int r = tap / p.S, s = tap - r * p.S;
        const int iy = a_iy0 + r * dil, ix = a_ix0 + s * dil;
        const bool inside = iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        const float* ap = p.in.p + in_img + ((size_t)(inside ? iy : 0) * Win + (inside ? ix : 0)) * p.in.cs + a_h * 4;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            const int ch = (grp * KCH + kc) * 8 + a_h * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inside && ch < Cin) v = *reinterpret_cast<const f32x4*>(ap + (grp * KCH + kc) * 8);
            a_reg[kc] = v;
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const int q = tid + k * 256;
            const int kc = q / (2 * BN), rem = q - kc * 2 * BN;
            const int h = rem / BN, j = rem - h * BN;
            const int chunk = grp * KCH + kc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (chunk < p.cin_chunks)
                v = *reinterpret_cast<const f32x4*>(p.wt + tap * tap_stride + chunk * chunk_stride + ((size_t)h * p.coutp + n0 + j) * 4);
            b_reg[k] = v;
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) As[(kc * 2 + a_h) * 128 + a_pix] = a_reg[kc];
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) Bs[tid + k * 256] = b_reg[k];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const f32x4* Ap = As + lh * 128 + wave * 32 + li;        // (row-major TH x TW tile: pixel index = wave * 32 + li for both tile shapes)
    const f32x4* Bp = Bs + lh * BN + li;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    load_step(0);
    for (int step = 0; step < nsteps; ++step) {
        store_step();
        __syncthreads();
        if (step + 1 < nsteps) load_step(step + 1);
        const int grp = step % ngrp;
        const int live = min(KCH, p.cin_chunks - grp * KCH);   // chunks of this step that exist (uniform)
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            if (kc < live) {
                const f32x4 a = Ap[kc * 2 * 128];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 b = Bp[kc * 2 * BN + nt * 32];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // output stage: every wave turns its 32 x 32 accumulator tile around through a private 4-KB LDS tile so that a lane
    // finishes 4 consecutive channels of one pixel (bias, activation, one 16-byte store)
    const int Hout = p.out.h, Wout = p.out.w, Cout = p.out.c;
    float* Xs = reinterpret_cast<float*>(smem) + wave * 1024;
    const bool vec_ok = (p.out.cs % 4 == 0) && (Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out.p) & 15) == 0);
    const int quad = lane & 7;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 32 + 4 * quad;
#pragma unroll
        for (int e = 0; e < 16; ++e) Xs[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + li] = acc[nt][e];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (co + c < Cout) bv[c] = p.bias[co + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 3) + 8 * r;               // pixel inside the wave's 32-pixel strip
            f32x4 v = *reinterpret_cast<const f32x4*>(Xs + row * 32 + 4 * quad) + bv;
            v = apply_act4(v, p.act, p.alpha);
            const int pix = wave * 32 + row;
            const int oy = ty0 + pix / TW, ox = tx0 + pix % TW;
            if (oy < Hout && ox < Wout) {
                float* o = p.out.p + (((size_t)img * Hout + oy) * Wout + ox) * p.out.cs + co;
                if (vec_ok && co + 3 < Cout) *reinterpret_cast<f32x4*>(o) = v;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (co + c < Cout) o[c] = v[c];
                }
            }
        }
    }
}

bool conv_mfma_tap_supported(const ConvParams& p) {
    return (p.in.cs % 4 == 0) && (p.in.c % 4 == 0) && ((((uintptr_t)p.in.p) & 15) == 0) && p.in.c >= 8 && p.R >= 1 && p.S >= 1;
}

template <int NT, int TW>
static hipError_t launch_conv_mfma_tap_t(const ConvParams& p, int dil, hipStream_t s) {
    constexpr int TH = 128 / TW, BN = NT * 32;
    const int tiles_x = (p.out.w + TW - 1) / TW, tiles_y = (p.out.h + TH - 1) / TH;
    const int nblk_n = p.coutp / BN;
    size_t lds = (size_t)4 * (2 * 128 + 2 * BN) * 16;
    if (lds < 4 * 4096) lds = 4 * 4096;
    const size_t grid = (size_t)p.n * tiles_x * tiles_y * nblk_n;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_mfma_tap_kernel<NT, TW>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y, nblk_n, dil);
    return hipGetLastError();
}

hipError_t launch_conv_mfma_tap(const ConvParams& p, int dil, hipStream_t s) {
    if (dil < 1) dil = 1;
    const int bn = conv_mfma_ntile(p.out.c);
    if (p.coutp % bn != 0) return hipErrorInvalidValue;
    const bool wide = p.out.w >= 32;
    if (bn == 128) return wide ? launch_conv_mfma_tap_t<4, 32>(p, dil, s) : launch_conv_mfma_tap_t<4, 16>(p, dil, s);
    if (bn == 64) return wide ? launch_conv_mfma_tap_t<2, 32>(p, dil, s) : launch_conv_mfma_tap_t<2, 16>(p, dil, s);
    return wide ? launch_conv_mfma_tap_t<1, 32>(p, dil, s) : launch_conv_mfma_tap_t<1, 16>(p, dil, s);
}

// ------------------------------------------------------------------------------------------------------------
// DepthwiseConv2D (depth multiplier 1, channels % 4 == 0): HBM-bound - every input value is used kh * kw times by
// neighbouring outputs of ITS channel only, so the work is one pass over the tensor if the halo is kept on chip.
// Workgroup = DW_TY x DW_TX output pixels x CQ channel quads (<= 64 channels): the input halo of the tile is staged in
// LDS with 16-byte loads (consecutive lanes = consecutive channel quads of a pixel: 256 contiguous bytes per pixel at 64
// channels), every work item then reads its kh * kw taps from LDS (consecutive lanes = consecutive 16-byte slots:
// conflict-free) and writes 4 channels with one 16-byte store.  Filter [tap][channel] (Keras' (kh, kw, cin, 1)) in LDS too.
// LDS == false: the same arithmetic straight from global memory (L2), for halos that do not fit 64 KB.
// ------------------------------------------------------------------------------------------------------------
constexpr int DW_TY = 8, DW_TX = 8;

template <bool LDS>
__global__ __launch_bounds__(256) void dwconv_kernel(TView in, TView out, const float* __restrict__ w, const float* __restrict__ bias,
                                                     int kh, int kw, int stride, int dil, int pad_top, int pad_left, int act,
                                                     float alpha, int tiles_x, int tiles_y, int cq, int nblk_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    unsigned bid = blockIdx.x;
    const int cb = bid % nblk_c; bid /= nblk_c;
    const int bx = bid % tiles_x; bid /= tiles_x;
    const int by = bid % tiles_y; bid /= tiles_y;
    const size_t img = bid;
    const int q0 = cb * cq;                                   // first channel quad of this workgroup
    const int nq = min(cq, in.c / 4 - q0);
    const int oy0 = by * DW_TY, ox0 = bx * DW_TX;
    const int hh = (DW_TY - 1) * stride + (kh - 1) * dil + 1, hw = (DW_TX - 1) * stride + (kw - 1) * dil + 1;
    const int iy0 = oy0 * stride - pad_top, ix0 = ox0 * stride - pad_left;
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [hh][hw][cq]
    f32x4* Ws = Hs + (LDS ? hh * hw * cq : 0);               // [kh * kw][cq]
    const float* ip = in.p + img * (size_t)in.h * in.w * in.cs + q0 * 4;
    for (int t = tid; t < kh * kw * cq; t += 256) {
        const int tap = t / cq, q = t - tap * cq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < nq) v = *reinterpret_cast<const f32x4*>(w + (size_t)tap * in.c + (q0 + q) * 4);
        Ws[t] = v;
    }
    if (LDS) {
        for (int t = tid; t < hh * hw * cq; t += 256) {
            const int q = t % cq, pix = t / cq;
            const int hy = pix / hw, hx = pix - hy * hw;
            const int iy = iy0 + hy, ix = ix0 + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < nq && iy >= 0 && iy < in.h && ix >= 0 && ix < in.w)
                v = *reinterpret_cast<const f32x4*>(ip + ((size_t)iy * in.w + ix) * in.cs + q * 4);
            Hs[t] = v;
        }
    }
    __syncthreads();
    for (int t = tid; t < DW_TY * DW_TX * cq; t += 256) {
        const int q = t % cq, pix = t / cq;
        const int py = pix / DW_TX, px = pix - py * DW_TX;
        const int oy = oy0 + py, ox = ox0 + px;
        if (q >= nq || oy >= out.h || ox >= out.w) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) acc = *reinterpret_cast<const f32x4*>(bias + (q0 + q) * 4);
        for (int r = 0; r < kh; ++r)
            for (int s = 0; s < kw; ++s) {
                f32x4 v;
                if (LDS) v = Hs[((py * stride + r * dil) * hw + px * stride + s * dil) * cq + q];
                else {
                    const int iy = iy0 + py * stride + r * dil, ix = ix0 + px * stride + s * dil;
                    v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (iy >= 0 && iy < in.h && ix >= 0 && ix < in.w) v = *reinterpret_cast<const f32x4*>(ip + ((size_t)iy * in.w + ix) * in.cs + q * 4);
                }
                const f32x4 k4 = Ws[(r * kw + s) * cq + q];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[c], k4[c], acc[c]);
            }
        acc = apply_act_ext4(acc, act, alpha);
        *reinterpret_cast<f32x4*>(out.p + ((img * out.h + oy) * out.w + ox) * out.cs + (q0 + q) * 4) = acc;
    }
}

// Any depth multiplier / channel count / alignment: thread = (output pixel, output channel); kernel (kh, kw, cin, mult)
__global__ __launch_bounds__(256) void dwconv_generic_kernel(TView in, TView out, const float* __restrict__ w, const float* __restrict__ bias,
                                                             size_t total, int kh, int kw, int stride, int dil, int pad_top, int pad_left,
                                                             int mult, int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        const int ci = co / mult;
        float acc = bias ? bias[co] : 0.f;
        for (int r = 0; r < kh; ++r) {
            const int iy = oy * stride - pad_top + r * dil;
            if (iy < 0 || iy >= in.h) continue;
            for (int s = 0; s < kw; ++s) {
                const int ix = ox * stride - pad_left + s * dil;
                if (ix < 0 || ix >= in.w) continue;
                acc = fmaf(in.p[((img * in.h + iy) * in.w + ix) * in.cs + ci], w[(size_t)(r * kw + s) * out.c + co], acc);
            }
        }
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + co] = apply_act_ext(acc, act, alpha);
    }
}

static unsigned lk_grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 65536 * 16) b = 65536 * 16;
    return (unsigned)(b ? b : 1);
}

hipError_t launch_dwconv(const TView& in, const TView& out, const float* w, const float* bias, int n, int kh, int kw, int stride,
                         int dil, int pad_top, int pad_left, int mult, int act, float alpha, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (dil < 1) dil = 1;
    const bool vec = mult == 1 && in.c % 4 == 0 && in.cs % 4 == 0 && out.cs % 4 == 0 && out.c == in.c &&
                     ((((uintptr_t)in.p) | ((uintptr_t)out.p)) & 15) == 0 && (bias == nullptr || (((uintptr_t)bias) & 15) == 0) &&
                     (((uintptr_t)w) & 15) == 0;
    if (!vec) {
        const size_t total = (size_t)n * out.h * out.w * out.c;
        if (!total) return hipSuccess;
        hipLaunchKernelGGL(dwconv_generic_kernel, dim3(lk_grid_for(total)), dim3(256), 0, s, in, out, w, bias, total, kh, kw, stride, dil,
                           pad_top, pad_left, mult, act, alpha);
        return hipGetLastError();
    }
    const int quads = in.c / 4;
    const int cq = quads < 16 ? quads : 16;
    const int nblk_c = (quads + cq - 1) / cq;
    const int tiles_x = (out.w + DW_TX - 1) / DW_TX, tiles_y = (out.h + DW_TY - 1) / DW_TY;
    const size_t grid = (size_t)n * tiles_x * tiles_y * nblk_c;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t hh = (size_t)(DW_TY - 1) * stride + (size_t)(kh - 1) * dil + 1, hw = (size_t)(DW_TX - 1) * stride + (size_t)(kw - 1) * dil + 1;
    const size_t halo = hh * hw * cq * 16, wbytes = (size_t)kh * kw * cq * 16;
    if (wbytes > 60000) {                                       // (a > 60 x 60-tap depthwise kernel)
        const size_t total = (size_t)n * out.h * out.w * out.c;
        hipLaunchKernelGGL(dwconv_generic_kernel, dim3(lk_grid_for(total)), dim3(256), 0, s, in, out, w, bias, total, kh, kw, stride, dil,
                           pad_top, pad_left, mult, act, alpha);
        return hipGetLastError();
    }
    if (halo + wbytes <= 64 * 1024)
        hipLaunchKernelGGL(dwconv_kernel<true>, dim3((unsigned)grid), dim3(256), halo + wbytes, s, in, out, w, bias, kh, kw, stride, dil,
                           pad_top, pad_left, act, alpha, tiles_x, tiles_y, cq, nblk_c);
    else
        hipLaunchKernelGGL(dwconv_kernel<false>, dim3((unsigned)grid), dim3(256), wbytes, s, in, out, w, bias, kh, kw, stride, dil,
                           pad_top, pad_left, act, alpha, tiles_x, tiles_y, cq, nblk_c);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// y = act(a (+) b) with numpy-style broadcasting of extents of 1 (h, w, c of either input)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bin_op(float a, float b, int mode) {
    switch (mode) {
        case ECSEG_BIN_MUL: return a * b;
        case ECSEG_BIN_SUB: return a - b;
        case ECSEG_BIN_MAX: return fmaxf(a, b);
        case ECSEG_BIN_MIN: return fminf(a, b);
        default: return a + b;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void binary_kernel(TView a, TView b, TView out, size_t total, int mode, int act, float alpha) {
    constexpr int V = VEC ? 4 : 1;
    const int cq = out.c / V;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % cq) * V;
        size_t pix = t / cq;
        const int x = (int)(pix % out.w); pix /= out.w;
        const int y = (int)(pix % out.h);
        const size_t img = pix / out.h;
        const float* ap = a.p + ((img * a.h + (a.h > 1 ? y : 0)) * a.w + (a.w > 1 ? x : 0)) * a.cs + (a.c > 1 ? c : 0);
        const float* bp = b.p + ((img * b.h + (b.h > 1 ? y : 0)) * b.w + (b.w > 1 ? x : 0)) * b.cs + (b.c > 1 ? c : 0);
        float* op = out.p + ((img * out.h + y) * out.w + x) * out.cs + c;
        if (VEC) {                                            // (both inputs carry all channels)
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap), bv = *reinterpret_cast<const f32x4*>(bp);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bin_op(av[e], bv[e], mode);
            *reinterpret_cast<f32x4*>(op) = apply_act_ext4(o, act, alpha);
        } else {
            op[0] = apply_act_ext(bin_op(ap[0], bp[0], mode), act, alpha);
        }
    }
}

hipError_t launch_binary(const TView& a, const TView& b, const TView& out, int n, int mode, int act, float alpha, hipStream_t s) {
    auto v4 = [&](const TView& v) { return v.c == out.c && v.c % 4 == 0 && v.cs % 4 == 0 && ((((uintptr_t)v.p) & 15) == 0); };
    const bool vec = v4(a) && v4(b) && v4(out);
    const size_t total = (size_t)n * out.h * out.w * (vec ? out.c / 4 : out.c);
    if (!total) return hipSuccess;
    if (vec) hipLaunchKernelGGL(binary_kernel<true>, dim3(lk_grid_for(total)), dim3(256), 0, s, a, b, out, total, mode, act, alpha);
    else hipLaunchKernelGGL(binary_kernel<false>, dim3(lk_grid_for(total)), dim3(256), 0, s, a, b, out, total, mode, act, alpha);
    return hipGetLastError();
}

// PReLU: slope per channel (per_element == 0) or per (y, x, channel) of the patch
__global__ __launch_bounds__(256) void prelu_kernel(TView in, TView out, const float* __restrict__ slope, size_t total, int per_element) {
    const size_t per_patch = (size_t)out.h * out.w;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        const size_t pix = t / out.c;
        const float v = in.p[pix * in.cs + c];
        const float a = slope[per_element ? (pix % per_patch) * out.c + c : (size_t)c];
        out.p[pix * out.cs + c] = v > 0.f ? v : a * v;
    }
}

hipError_t launch_prelu(const TView& in, const TView& out, const float* slope, int n, int per_element, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(prelu_kernel, dim3(lk_grid_for(total)), dim3(256), 0, s, in, out, slope, total, per_element);
    return hipGetLastError();
}

// LayerNormalization over the channels of a pixel: LPP lanes share one pixel (strided channel reads, shuffle reductions);
// mean, then variance of the centred values (tf.nn.moments), then x * inv + (beta - mean * inv) with inv = rsqrt(var + eps) * gamma
template <int LPP>
__global__ __launch_bounds__(256) void layernorm_kernel(TView in, TView out, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        size_t npix, float eps) {
    const int sub = threadIdx.x & (LPP - 1);
    const float inv_c = 1.f / (float)in.c;
    for (size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPP; ; pix += ((size_t)gridDim.x * blockDim.x) / LPP) {
        const size_t wave_first = pix - (((size_t)threadIdx.x & 63) / LPP);
        if (wave_first >= npix) break;                        // (whole waves stay in the loop: the shuffles need every lane)
        const bool live = pix < npix;
        const float* ip = in.p + (live ? pix : 0) * in.cs;
        float sum = 0.f;
        if (live) for (int c = sub; c < in.c; c += LPP) sum += ip[c];
#pragma unroll
        for (int d = LPP / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        const float mean = sum * inv_c;
        float sq = 0.f;
        if (live) for (int c = sub; c < in.c; c += LPP) { const float dlt = ip[c] - mean; sq = fmaf(dlt, dlt, sq); }
#pragma unroll
        for (int d = LPP / 2; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
        const float rstd = rsqrtf(sq * inv_c + eps);
        if (live) {
            float* op = out.p + pix * out.cs;
            for (int c = sub; c < in.c; c += LPP) {
                const float inv = rstd * (gamma ? gamma[c] : 1.f);
                op[c] = fmaf(ip[c], inv, (beta ? beta[c] : 0.f) - mean * inv);
            }
        }
    }
}

hipError_t launch_layernorm(const TView& in, const TView& out, const float* gamma, const float* beta, int n, float eps, hipStream_t s) {
    const size_t npix = (size_t)n * out.h * out.w;
    if (!npix) return hipSuccess;
    if (in.c >= 64) {
        size_t blocks = (npix * 64 + 255) / 256; if (blocks > 65536 * 8) blocks = 65536 * 8;
        hipLaunchKernelGGL(layernorm_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, gamma, beta, npix, eps);
    } else if (in.c >= 16) {
        size_t blocks = (npix * 16 + 255) / 256; if (blocks > 65536 * 8) blocks = 65536 * 8;
        hipLaunchKernelGGL(layernorm_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, gamma, beta, npix, eps);
    } else {
        size_t blocks = (npix * 4 + 255) / 256; if (blocks > 65536 * 8) blocks = 65536 * 8;
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, gamma, beta, npix, eps);
    }
    return hipGetLastError();
}

// MaxPooling2D / AveragePooling2D with padding='same': the window starts (pad_top, pad_left) before the input; the maximum
// and the average run over the pixels that lie inside it (TensorFlow's average excludes the padding from the divisor)
__global__ __launch_bounds__(256) void pool_pad_kernel(TView in, TView out, size_t total, int kh, int kw, int stride, int pad_top,
                                                       int pad_left, int mode) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float m = mode ? 0.f : -INFINITY;
        int cnt = 0;
        for (int r = 0; r < kh; ++r) {
            const int iy = oy * stride - pad_top + r;
            if (iy < 0 || iy >= in.h) continue;
            for (int s = 0; s < kw; ++s) {
                const int ix = ox * stride - pad_left + s;
                if (ix < 0 || ix >= in.w) continue;
                const float v = in.p[((img * in.h + iy) * in.w + ix) * in.cs + c];
                m = mode ? m + v : fmaxf(m, v);
                ++cnt;
            }
        }
        if (mode) m = cnt ? m / (float)cnt : 0.f;
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + c] = m;
    }
}

hipError_t launch_pool_pad(const TView& in, const TView& out, int n, int kh, int kw, int stride, int pad_top, int pad_left, int mode,
                           hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(pool_pad_kernel, dim3(lk_grid_for(total)), dim3(256), 0, s, in, out, total, kh, kw, stride, pad_top, pad_left, mode);
    return hipGetLastError();
}

// conv_generic_kernel (unet_kernels.hip) with dilation rates and per-axis strides: the scalar fall-back of dilated layers the MFMA kernel does
// not take (Cin % 4 != 0, unaligned views) and of every layer with ANISOTROPIC strides / dilation rates (round 6)
__global__ __launch_bounds__(256) void conv_generic_dil_kernel(TView in, TView out, const float* __restrict__ w, const float* __restrict__ bias,
                                                               size_t total, int R, int S, int stride, int stride_x, int dil, int dil_x, int pad_top, int pad_left,
                                                               int act, float alpha) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(t % out.c);
        size_t pix = t / out.c;
        const int ox = (int)(pix % out.w); pix /= out.w;
        const int oy = (int)(pix % out.h);
        const size_t img = pix / out.h;
        float acc = bias ? bias[co] : 0.f;
        for (int r = 0; r < R; ++r) {
            const int iy = oy * stride - pad_top + r * dil;
            if (iy < 0 || iy >= in.h) continue;
            for (int s = 0; s < S; ++s) {
                const int ix = ox * stride_x - pad_left + s * dil_x;
                if (ix < 0 || ix >= in.w) continue;
                const float* ip = in.p + ((img * in.h + iy) * in.w + ix) * in.cs;
                const float* wp = w + (size_t)(r * S + s) * in.c * out.c + co;
                for (int ci = 0; ci < in.c; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * out.c], acc);
            }
        }
        out.p[((img * out.h + oy) * out.w + ox) * out.cs + co] = apply_act(acc, act, alpha);
    }
}

hipError_t launch_conv_generic_dil(const TView& in, const TView& out, const float* w, const float* bias, int n, int R, int S, int stride, int stride_x,
                                   int dil, int dil_x, int pad_top, int pad_left, int act, float alpha, hipStream_t s) {
    const size_t total = (size_t)n * out.h * out.w * out.c;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(conv_generic_dil_kernel, dim3(lk_grid_for(total)), dim3(256), 0, s, in, out, w, bias, total, R, S, stride, stride_x,
                       dil < 1 ? 1 : dil, dil_x < 1 ? 1 : dil_x, pad_top, pad_left, act, alpha);
    return hipGetLastError();
}

}  // namespace ecseg
