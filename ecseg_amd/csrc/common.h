// Shared declarations of libecseg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ecseg_hip.h"

namespace ecseg {

// NHWC float32 view into a device buffer: consecutive pixels are `cs` floats apart, `p` points at the view's first
// channel of pixel (n=0, y=0, x=0).
struct TView {
    float* p;
    int h, w, c, cs;
};

struct ConvParams {
    TView in, out;
    const float* wt;    // re-laid-out kernel (see relayout_* in api.hip)
    const float* bias;  // may be null
    int n;              // patches in the batch
    int R, S;           // taps
    int pad_top, pad_left;
    int act;
    float alpha;
    int cin_chunks;     // ceil(Cin / 8)
    int coutp;          // Cout padded to a multiple of the N tile
    // transposed-conv mode (R = S = 1 in the GEMM, kT x kT stride-kT scatter in the epilogue)
    int convt;          // 0 | 1 | 2 (2: ONE output phase per launch - N = coutp, the phase in phase_a / phase_b)
    int kT, crop_top, crop_left;
    int convt_ext;      // transposed mode: tiles walk the input extent + convt_ext (sub-pixel form of k x k / stride 2: 1)
    int phase_a, phase_b;   // transposed mode with ONE output phase per launch (N = coutp): the launch writes output (kT i + phase_a - crop_top, kT j + phase_b - crop_left)
    int tap_zero_mask;  // sub-pixel form (R = S = 2, kT = 2): bit (tap * 4 + phase) set = that (tap, output phase) block of the filter is all zero
                        // (7 of 16 at k = 3: the phases have 4, 2, 2 and 1 taps): conv_mfma_kernel skips its MFMAs.  0: multiply everything
    int stride;         // forward convolution: output stride (0 / 1: dense; 2: conv_mfma gathers a strided halo)
    const float* zero;  // >= 16 bytes of zeros in device memory (LDS-DMA source for padding / out-of-image pixels)
    TView pool;         // pool.p != null: also write MaxPooling2D(2x2, stride 2) of the activated output (conv_wino4 only)
    // fused 1x1 head (conv_wino4 only, Cout == 64): head_w != null: logits = act(out) . head_w[64][4] + head_b[4] (classes
    // padded to 4), head_act (softmax | linear ...) over head_k classes, written to head_out; the 64-channel output itself
    // is not written when head_only != 0
    const float* head_w; const float* head_b; TView head_out; int head_k, head_act, head_only;
    // demand-driven cropping (conv_wino4 only): lut != null: only the 16x16 regions listed are computed, the same list
    // for every image of `per_image` consecutive patches; entry = patch-in-image << 16 | (y origin / 4) << 8 | (x origin / 4)
    const int32_t* lut; int lut_len, per_image;
    // Winograd kernels of a cropped plan: in_box != null: (y0, y1, x0, x1), inclusive, per window of an image - the receptive
    // field of the outputs a later stage reads; input pixels outside it are read as ZERO.  A Winograd tile mixes its whole
    // 6x6 (4x4) input tile into every output: pixels outside an output's 3x3 support cancel only up to rounding, and
    // outside the box a cropped producer has left whatever the buffer held before - the results would depend (in the last
    // bits) on the history of the buffer.  Window of patch i: (i + box_first) % per_image.
    const int32_t* in_box; int box_first;
    // conv_mfma (transposed convolutions): the same list idea over its TH x TW tiles of the INPUT extent; force_tw = 16 | 32
    // selects the tile shape the list was built for (0: the launcher's own choice)
    int force_tw;
    // conv_wino (F(2x2)): 1: layers with <= 32 input and <= 32 output channels take the filter-resident kernel
    int resident;
    // conv_wino4: 1: a layer with exactly 32 output channels splits the input channels of every 8-channel group between the
    // two channel-half waves of a transform row instead of multiplying zero padding
    int w4_split;
    // conv_wino16 (16 -> 16 channels), round 5: first_w != null: p.in is the network's 1-channel INPUT and this launch also computes
    // the layer in front - Conv2D 3x3 'same', 1 -> 16 channels, kernel first_w[9][16] (HWIO), bias first_b (may be null),
    // activation first_act - on the matrix cores, straight into its own halo buffer: the 16-channel tensor between the two
    // layers never exists in memory
    const float* first_w; const float* first_b; int first_act; float first_alpha;
    // filter image strides in floats: [tap][chunk][half][N padded][4] with padded chunk / tap pitches (power-of-two
    // pitches put the 16 transform points of a K-chunk on the same L2 channel and set)
    long wt_chunk_stride, wt_tap_stride;
};

// Winograd F(4x4,3x3) interpolation points {0, +-W4_PA, +-W4_PB, inf}, shared by the host filter transform (api.hip:
// winograd4_filter) and the kernel's input / output transforms (wino4_kernel.hip).  The textbook set is {0, +-1, +-2, inf};
// the rounding error of the result is dominated by the float32 channel sum of the transformed products on the matrix cores,
// whose magnitude the points set: tools/wino_points.py replays the kernel's arithmetic on the CPU and measures, against a
// float64 convolution, 4.4x the error of a sequential float32 direct convolution for {1, 2} and 2.0x for {5/8, 3/2} (the
// best pair on a 1/16 grid; both dyadic, so every constant of B^T and A^T stays exact in float32) - at the same number of
// VALU instructions (the +-1 rows' additions become fmas).
// (-DECSEG_W4_PA=1 -DECSEG_W4_PB=2 rebuilds the textbook kernel for A/B measurements: tools/build_variants.sh points12)
#ifndef ECSEG_W4_PA
#define ECSEG_W4_PA 0.625
#define ECSEG_W4_PB 1.5
#endif
constexpr double W4_PA = ECSEG_W4_PA, W4_PB = ECSEG_W4_PB;

// pitches used by relayout_* (api.hip) and the kernels
inline long wt_chunk_pitch(int np_total) { return (long)2 * np_total * 4 + 32; }
inline long wt_tap_pitch(int np_total, int chunks) { return wt_chunk_pitch(np_total) * chunks + 96; }

// ---- launchers implemented in unet_kernels.hip --------------------------------------------------------------
hipError_t launch_conv_mfma(const ConvParams& p, hipStream_t s);
bool       conv_mfma_supported(const ConvParams& p);
int        conv_mfma_ntile(int cout);   // N tile (32 | 64 | 128) used for a given Cout
// Winograd F(2x2,3x3) variant for 3x3 / stride 1 / pad 1 convolutions; p.wt = 16 transformed taps, p.coutp padded to
// conv_wino_ntile()
hipError_t launch_conv_wino(const ConvParams& p, hipStream_t s);
int        conv_wino_ntile(int cout);
// Winograd F(2x2,3x3) for 16 / 32 input and output channels on 16x16x4 MFMAs (wino16_kernel.hip); p.wt = image written by
// relayout_wino16 (api.hip)
hipError_t launch_conv_wino16(const ConvParams& p, hipStream_t s);
bool       conv_wino16_supported(const ConvParams& p);
bool       conv_wino16_first_supported(const ConvParams& p);   // with ConvParams::first_w: the network's first layer computed into the halo
// Winograd F(4x4,3x3) (wino4_kernel.hip); p.wt = image written by winograd4_filter (api.hip)
hipError_t launch_conv_wino4(const ConvParams& p, hipStream_t s);
bool       conv_wino4_supported(const ConvParams& p);
bool       conv_wino4_span_ok(const ConvParams& p, int windows);   // the halo's buffer descriptor reaches `windows` consecutive windows
// F(4x4,3x3) on the fp32 matrix cores with the row transform done once per workgroup (wino4r_kernel.hip, round 6); p.wt as conv_wino4_kernel
hipError_t launch_conv_wino4r(const ConvParams& p, hipStream_t s);
bool       conv_wino4r_supported(const ConvParams& p);
// F(4x4,3x3) with 3-way bf16 split operands on the bf16 matrix pipe (wino4s_kernel.hip, round 6); p.wt = the stage image written
// on the device by launch_wino4s_filter from the fp32 image of winograd4_filter (wino4s_image_bytes bytes)
hipError_t launch_conv_wino4s(const ConvParams& p, hipStream_t s);
bool       conv_wino4s_supported(const ConvParams& p);
size_t     wino4s_image_bytes(int cin, int cout);
hipError_t launch_wino4s_filter(const float* wt_wino4, void* dst, int cin, int cout, hipStream_t s);
// One-tap GEMM (2x2 / stride-2 transposed convolutions, 1x1 convolutions) with 3-way bf16 split operands (convs_kernel.hip, round 6);
// p.wt = the split image launch_convs_filter writes from the fp32 one-tap image (np = kT * kT * coutp columns; convs_image_bytes bytes)
hipError_t launch_convs(const ConvParams& p, hipStream_t s);
bool       convs_supported(const ConvParams& p);
size_t     convs_image_bytes(int cin, int np);
hipError_t launch_convs_filter(const float* wt_fp32, void* dst, int cin, int np, hipStream_t s);

hipError_t launch_conv_small_cin(const TView& in, const TView& out, const float* w_hwio, const float* bias, int n,
                                 int R, int S, int pad_top, int pad_left, int act, float alpha, hipStream_t s);
hipError_t launch_conv_head(const TView& in, const TView& out, const float* w_io, const float* bias, int n, int act,
                            float alpha, hipStream_t s);
hipError_t launch_conv_generic(const TView& in, const TView& out, const float* w_hwio, const float* bias, int n,
                               int R, int S, int stride, int pad_top, int pad_left, int act, float alpha, hipStream_t s);
hipError_t launch_convt_generic(const TView& in, const TView& out, const float* w_hwoi, const float* bias, int n,
                                int R, int S, int stride, int crop_top, int crop_left, int act, float alpha,
                                hipStream_t s);
hipError_t launch_maxpool(const TView& in, const TView& out, int n, int kh, int kw, int stride, int mode, hipStream_t s);
hipError_t launch_global_pool(const TView& in, const TView& out, int n, int mode, hipStream_t s);
hipError_t launch_upsample(const TView& in, const TView& out, int n, int factor, int mode, hipStream_t s);
hipError_t launch_affine(const TView& in, const TView& out, const float* scale, const float* shift, int n, int act,
                         float alpha, hipStream_t s);
hipError_t launch_add(const TView& a, const TView& b, const TView& out, int n, int act, float alpha, hipStream_t s);
hipError_t launch_copy(const TView& in, const TView& out, int n, int off_y, int off_x, hipStream_t s);
hipError_t launch_softmax(const TView& in, const TView& out, int n, hipStream_t s);
hipError_t launch_u8_to_f32(const uint8_t* in, float* out, size_t count, hipStream_t s);

// ---- launchers implemented in layer_kernels.hip (the wider Keras vocabulary) -------------------------------------
// Conv2D with any taps / stride / dilation on the matrix cores; p.wt = relayout_conv image, p.coutp padded to conv_mfma_ntile()
hipError_t launch_conv_mfma_tap(const ConvParams& p, int dilation, hipStream_t s);
bool       conv_mfma_tap_supported(const ConvParams& p);
hipError_t launch_conv_generic_dil(const TView& in, const TView& out, const float* w_hwio, const float* bias, int n, int R, int S,
                                   int stride, int stride_x, int dilation, int dilation_x, int pad_top, int pad_left, int act, float alpha,
                                   hipStream_t s);       // scalar kernel: vertical / horizontal stride and dilation rate may differ
// DepthwiseConv2D: kernel (kh, kw, cin, mult), output channel = input channel * mult + j
hipError_t launch_dwconv(const TView& in, const TView& out, const float* w, const float* bias, int n, int kh, int kw, int stride,
                         int dilation, int pad_top, int pad_left, int mult, int act, float alpha, hipStream_t s);
// y = act(a (+) b), mode = ECSEG_BIN_*, extents of 1 broadcast
hipError_t launch_binary(const TView& a, const TView& b, const TView& out, int n, int mode, int act, float alpha, hipStream_t s);
hipError_t launch_prelu(const TView& in, const TView& out, const float* slope, int n, int per_element, hipStream_t s);
hipError_t launch_layernorm(const TView& in, const TView& out, const float* gamma, const float* beta, int n, float eps, hipStream_t s);
hipError_t launch_pool_pad(const TView& in, const TView& out, int n, int kh, int kw, int stride, int pad_top, int pad_left, int mode,
                           hipStream_t s);
// im2patches_overlap on the device: (n_img, H, W) uint8 -> (n_img * n_pos, 256, 256, 1) float32
hipError_t launch_tile_patches(const uint8_t* gray, int n_img, int H, int W, const int32_t* pos_yx, int n_pos,
                               float* out, hipStream_t s);

// ---- launchers implemented in post_kernels.hip -----------------------------------------------------------------
// stitch + img_as_ubyte + argmax; src_map: (H*W) int32 = (patch << 16) | (y << 8) | x, or -1 when never written
// tie_risk (may be null): per image, the pixels whose two largest quantised values differ by at most 1; tie_shards: scratch of
// n_img * G_SHARDS * G_STRIDE ints (replicated counters, one 128-B line each)
hipError_t launch_stitch_argmax(const float* probs, int prob_cs, const int32_t* src_map, int n_img, int n_pos,
                                int H, int W, uint8_t* labels, hipStream_t s, int32_t* tie_risk = nullptr, int32_t* tie_shards = nullptr);
// the stitched probabilities themselves: float32 (n_img, H, W, 4); never-written canvas pixels are 0
hipError_t launch_stitch_probs(const float* probs, int prob_cs, const int32_t* src_map, int n_img, int n_pos,
                               int H, int W, float* out, hipStream_t s);

struct PostWorkspace {
    // all sized for `cap_img` images of `cap_px` pixels
    int32_t* L;          // union-find parents / final root index per pixel
    uint32_t* area;      // per-root slots (indexed like L)
    unsigned long long* sumy;
    unsigned long long* sumx;
    uint32_t* flag;      // per-root bit flags
    uint8_t* tmpA;       // scratch label images
    uint8_t* tmpB;
    int32_t* list;       // per image: compacted root lists for the nucleus-in-metaphase test
    int32_t* g;          // G_SLOTS x cap_img blocks of per-image counters (G_STRIDE ints x G_SHARDS replicas each)
    uint8_t* tile_any;   // per image and 64 x 32 labelling tile: the tile holds a keyed pixel (written by ccl_local)
    uint32_t* own_bits;  // per image and tile: 2048 bits, bit = the pixel is the root of a tile component ("owner"; ccl_local -> ccl_resolve)
    double* binned;      // per image and axis: the chromosome centroids' coordinates grouped by integer bin (nucleus test)
    int32_t* binstart;   // per image and axis: first entry of every bin in `binned` (NUCLEUS_BIN_EXTENT + 2 ints)
    size_t binned_cap;   // entries per image and axis in `binned`
    int cap_img;
    size_t cap_px;
};
enum { G_STRIDE = 32, G_SHARDS = 16 };
enum { G_SLOTS = 6 };                  // counter blocks in PostWorkspace::g: one per labelling of run_meta_inference that produces counters (zeroed by ONE memset)
enum { NUCLEUS_BIN_EXTENT = 32768 };   // largest image extent for which the nucleus test runs on binned coordinates (LDS histogram)

// meta_inference on n_img uint8 label images, in place; n_ec receives count_cc(img==3)[0] per image
hipError_t run_meta_inference(PostWorkspace& ws, uint8_t* img, int n_img, int H, int W, int32_t* n_ec_dev,
                              hipStream_t s);
hipError_t run_count_cc(PostWorkspace& ws, const uint8_t* mask, int n_img, int H, int W, int32_t* n_dev,
                        long long* px_dev, hipStream_t s);
hipError_t run_ccl_labels(PostWorkspace& ws, const uint8_t* mask, int n_img, int H, int W, int conn,
                          int32_t* labels_dev, hipStream_t s);
hipError_t run_count_coloc(PostWorkspace& ws, const uint8_t* ob1, const uint8_t* ob2, int n_img, int H, int W,
                           int32_t* n_dev, hipStream_t s);
hipError_t run_count_hsr(PostWorkspace& ws, const uint8_t* chrom, const uint8_t* fish, int n_img, int H, int W,
                         int thr, int32_t* n_dev, hipStream_t s);
hipError_t run_overlay(PostWorkspace& ws, const uint8_t* labels, const uint8_t* rgb, int n_img, int H, int W, int C,
                       int sens, int hsr_thr, long long* out_dev, hipStream_t s);
hipError_t launch_u16_to_u8(const uint16_t* in, uint8_t* out, size_t count, hipStream_t s);
hipError_t run_preprocess(const void* img, int n_img, int H, int W, int C, int bps, uint8_t* gray, int32_t* inverted,
                          uint32_t* hist_ws, hipStream_t s);

}  // namespace ecseg
