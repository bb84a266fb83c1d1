/*
 * ecseg_hip.h - C ABI of libecseg_hip.so: the MI355X (gfx950) implementation of ecSeg's metaseg hot path.
 *
 * The reference (UCRajkumar/ecSeg) is pure Python and has no FFI of its own; the boundary it exposes is three
 * Python call shapes plus a file contract.  Each entry point below names the reference
 * interface it stands in for (file:line relative to the reference repository).  The ctypes binding that a
 * maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers and sizes, no C++/torch types;
 *   - every call returns 0 on success or a negative ECSEG_E_* code; the text of the last failure is available
 *     from ecseg_last_error(); nothing throws across the ABI;
 *   - the caller owns host memory, the handle owns device memory and its HIP stream;
 *   - one handle per GPU; a handle is not thread-safe, different handles may be driven from different threads;
 *   - all calls are synchronous with respect to the host (results are complete on return);
 *   - "_dev" variants take DEVICE pointers (e.g. torch tensors' data_ptr) and run on the handle's stream without
 *     host copies; they still return only after the stream has drained unless stated otherwise.
 *   - images are row-major (H, W[, C]); label images are uint8 with values 0..3
 *     (0 background, 1 nucleus, 2 chromosome, 3 ecDNA: src/utils.py:128-131).
 */
#ifndef ECSEG_HIP_H
#define ECSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): ecseg_get_conv_launch_profile kinds 3 / 4 and fusion bits, images_per_group 0 = automatic, op code 9
 * (GLOBALPOOL), MAXPOOL honours `mode`, CONV accepts stride != 1, ecseg_npy_write_i64 / ecseg_png_write* /
 * ecseg_tiff_* / ecseg_allgather_records added.  ecseg_amd/_lib.py refuses a library whose version differs from the one it was written for. */
/* 3 (round 4): ecseg_segment_images_ex (per-image tie-risk counts, stitched probabilities). */
/* 4 (round 5): ecseg_op_desc gains `dilation`; op codes 10-12 (DWCONV, PRELU, LAYERNORM); ADD takes `mode` (add / multiply / subtract /
 * maximum / minimum) and broadcasts extents of 1; MAXPOOL honours pad_top / pad_left ('same' pooling); activation codes 7-14;
 * ECSEG_COMM_TIMEOUT_S bounds ecseg_comm_create / ecseg_allgather_records*. */
/* 5 (round 5): ecseg_meta_segment (pre-process + segment in one call), ecseg_prefetch_input, ecseg_host_alloc / ecseg_host_free
 * (page-locked host buffers); ecseg_create sets the device's scheduling flag to hipDeviceScheduleBlockingSync (see there). */
/* (round 6, still 5 - additions a round-5 caller never triggers: option "winograd" = 3 and "wino4_rowpass"; ecseg_get_conv_launch_profile kinds 5 / 6
 * (the split kernels); CONV ops read their so far unused `mode` word as the horizontal stride / dilation rate (0 = as before); CONVT kernels larger than
 * their stride run phase by phase.) */
#define ECSEG_ABI_VERSION 5

#define ECSEG_OK             0
#define ECSEG_E_INVALID     -1   /* bad argument / shape / plan */
#define ECSEG_E_HIP         -2   /* a HIP runtime call failed */
#define ECSEG_E_NOMODEL     -3   /* a model-dependent call before ecseg_model_load */
#define ECSEG_E_NOMEM       -4
#define ECSEG_E_UNSUPPORTED -5
#define ECSEG_E_IO          -6   /* a file could not be opened / read / written (host I/O entry points) */

typedef struct ecseg_ctx ecseg_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
int         ecseg_abi_version(void);
/* SIDE EFFECT ON THE PROCESS: ecseg_create sets hipDeviceScheduleBlockingSync on the device - a device-wide, process-wide
 * setting: every HIP user of the process (torch tensors of an embedding application too) then SLEEPS while it waits for the GPU
 * instead of spinning (same wall time, one core less per waiting call).  ECSEG_SPIN_WAIT=1 in the environment leaves the runtime's
 * default alone - at your own risk: under the default mode hipFree was seen to hang for ever in ecseg_destroy after several
 * handles had been created and closed in one process (csrc/api.hip: ecseg_create).  ECSEG_DEBUG_CALLS=1 prints a host-side
 * timeline of every ecseg_meta_segment call on stderr. */
int         ecseg_create(ecseg_ctx** out, int device_id);
void        ecseg_destroy(ecseg_ctx* h);
const char* ecseg_last_error(ecseg_ctx* h);        /* h may be NULL: error of the last failed ecseg_create */
int         ecseg_device_name(ecseg_ctx* h, char* buf, int buflen);
/* The HIP stream all work of this handle is launched on (hipStream_t as void*), for event timing by the caller. */
void*       ecseg_stream(ecseg_ctx* h);

/* ---- model plan: replaces tf.keras.models.load_model (src/utils.py:27-33) -------------------------------- */
/* Kernel-level operators of the U-Net plan.  The host (ecseg_amd/keras_plan.py) lowers the Keras model_config
 * found in metaseg.h5 to this list; tensors are NHWC float32 "views" into device buffers so that Concatenate
 * costs nothing (producers write straight into the concatenated buffer). */
enum {
    ECSEG_OP_CONV      = 1,  /* Conv2D kh x kw, stride s, dilation d, zero padding (pad_top, pad_left), bias, activation; a Dense
                                layer is the 1x1 case on a (1, 1, features) tensor; a grouped convolution is one CONV per group
                                on channel views */
    ECSEG_OP_CONVT     = 2,  /* Conv2DTranspose kh x kw, stride s, crop (pad_top, pad_left), bias, activation */
    ECSEG_OP_MAXPOOL   = 3,  /* MaxPooling2D (mode 0) / AveragePooling2D (mode 1) kh x kw stride s; 'same': (pad_top, pad_left) window
                                positions before the input (the maximum / the average runs over the pixels inside the input) */
    ECSEG_OP_UPSAMPLE  = 4,  /* UpSampling2D x s, mode: 0 nearest, 1 bilinear (half-pixel centres) */
    ECSEG_OP_AFFINE    = 5,  /* y = act(x * scale[c] + shift[c]): BatchNormalization (inference), Rescaling */
    ECSEG_OP_ACT       = 6,  /* y = act(x) */
    ECSEG_OP_ADD       = 7,  /* y = act(a (+) b), (+) = `mode` (ECSEG_BIN_*: Add / Multiply / Subtract / Maximum / Minimum); an extent of
                                1 in either input (h, w or c) is broadcast - the squeeze-and-excite x * s(1, 1, c) */
    ECSEG_OP_COPY      = 8,  /* y = x (materialise a view, ZeroPadding2D / Cropping2D via offsets) */
    ECSEG_OP_GLOBALPOOL = 9, /* GlobalMaxPooling2D (mode 0) / GlobalAveragePooling2D (mode 1): (h, w, c) -> (1, 1, c) */
    ECSEG_OP_DWCONV    = 10, /* DepthwiseConv2D kh x kw, stride s, dilation, zero padding (pad_top, pad_left), depth multiplier
                                `mode` (>= 1; output channel = input channel * mode + j), kernel (kh, kw, cin, mode), bias,
                                activation; SeparableConv2D = DWCONV followed by a 1x1 CONV */
    ECSEG_OP_PRELU     = 11, /* y = x > 0 ? x : a * x; w0 = a: mode 0 one slope per channel (shared_axes [1, 2]), mode 1 one per
                                (y, x, channel) of the patch */
    ECSEG_OP_LAYERNORM = 12  /* LayerNormalization over the channel axis of every pixel: (x - mean) / sqrt(var + alpha) * w0[c] +
                                w1[c] (alpha = epsilon; w0 / w1 = -1: no scale / no centre) */
};
/* RELU_CLIP: min(max(x, 0), alpha) (ReLU(max_value) - relu6); ELU: x > 0 ? x : alpha (exp(x) - 1); HARD_SIGMOID: Keras' clip(0.2 x +
 * 0.5, 0, 1); GELU: the exact erf form (Keras' default approximate=False) */
enum { ECSEG_ACT_LINEAR = 0, ECSEG_ACT_RELU = 1, ECSEG_ACT_SOFTMAX = 2, ECSEG_ACT_SIGMOID = 3,
       ECSEG_ACT_LEAKY = 4, ECSEG_ACT_TANH = 5, ECSEG_ACT_ELU = 6, ECSEG_ACT_RELU_CLIP = 7, ECSEG_ACT_SWISH = 8,
       ECSEG_ACT_HARD_SIGMOID = 9, ECSEG_ACT_SOFTPLUS = 10, ECSEG_ACT_SELU = 11, ECSEG_ACT_GELU = 12, ECSEG_ACT_EXP = 13,
       ECSEG_ACT_SOFTSIGN = 14 };
/* ECSEG_OP_ADD `mode` */
enum { ECSEG_BIN_ADD = 0, ECSEG_BIN_MUL = 1, ECSEG_BIN_SUB = 2, ECSEG_BIN_MAX = 3, ECSEG_BIN_MIN = 4 };

typedef struct ecseg_tensor_desc {
    int32_t buffer;     /* index of the device buffer this view lives in */
    int32_t h, w, c;    /* per-patch logical shape */
    int32_t c_stride;   /* floats between consecutive pixels in the buffer (>= c_offset + c) */
    int32_t c_offset;   /* first channel of the view inside a pixel */
} ecseg_tensor_desc;

typedef struct ecseg_op_desc {
    int32_t op;
    int32_t in0, in1;           /* tensor indices (in1 = -1 when unused) */
    int32_t out;
    int32_t kh, kw, stride;
    int32_t pad_top, pad_left;  /* CONV: zero padding before; CONVT: rows/cols cropped from the full output;
                                   COPY: offset of the input inside the output (>0) or crop (<0) */
    int32_t act;
    int32_t mode;               /* UPSAMPLE interpolation; MAXPOOL / GLOBALPOOL: 0 max, 1 average; ADD: ECSEG_BIN_*; DWCONV: depth
                                   multiplier; PRELU: 0 per channel, 1 per element; CONV (round 6): per-axis strides / dilation
                                   rates - bits 0-7 the HORIZONTAL stride, bits 8-15 the HORIZONTAL dilation rate where they differ
                                   from the vertical ones in `stride` / `dilation` (0: the same; such layers run on the scalar
                                   kernel, any taps / channels) */
    int32_t w0, w1;             /* weight array indices: CONV/CONVT kernel + bias (-1 none); AFFINE scale + shift */
    float   alpha;              /* LEAKY slope / RELU_CLIP maximum / ELU alpha; LAYERNORM epsilon */
    int32_t dilation;           /* CONV / DWCONV: dilation_rate (0 or 1: none) */
} ecseg_op_desc;

/* weights[i] is a host float32 array of weight_len[i] elements, Keras layout
 * (Conv2D kernel HWIO; Conv2DTranspose kernel (kh, kw, out, in)).  The library re-lays kernels out for its MFMA
 * kernels once, on the device.  input_tensor must be (256, 256, 1)-shaped per patch for the segment calls. */
int ecseg_model_load(ecseg_ctx* h,
                     const ecseg_tensor_desc* tensors, int n_tensors, int n_buffers,
                     const ecseg_op_desc* ops, int n_ops,
                     const float* const* weights, const int64_t* weight_len, int n_weights,
                     int input_tensor, int output_tensor);
int ecseg_model_flops_per_patch(ecseg_ctx* h, double* flops);   /* algorithmic 2*MAC count of the loaded plan */

/* ---- model.predict_on_batch(uint8[N,256,256,C]) -> float32[N,256,256,K] (src/utils.py:115) --------------- */
int ecseg_forward_patches(ecseg_ctx* h, const uint8_t* patches_nhwc, int n, float* out_nhwc);
/* Same with float32 inputs (N, H, W, C): the interSeg classifier ecseg_c is fed normalised floats
 * (preprocess_ecseg_c, src/utils.py:166-173; src/interseg.py:167-168). */
int ecseg_forward_patches_f32(ecseg_ctx* h, const float* patches_nhwc, int n, float* out_nhwc);
/* Debug/parity: copy any plan tensor (compact NHWC float32) after the last forward of n patches. */
int ecseg_read_tensor(ecseg_ctx* h, int tensor, int n, float* out_nhwc);

/* ---- meta_segment minus file I/O (src/utils.py:111-119) + count_cc(I==3)[0] (src/metaseg.py:46) ---------- */
/* gray: n_img pre-processed uint8 images (H, W) (the output of meta_preprocess).  Steps on the device:
 * im2patches_overlap (src/image_tools.py:148-186) -> U-Net -> patches2im_overlap (:188-252) -> img_as_ubyte ->
 * argmax (src/utils.py:117-118) -> meta_inference (src/image_tools.py:15-84) -> count_cc(I==3)[0].
 * labels_raw (optional, may be NULL) receives the argmax labels, labels_post the post-processed labels,
 * n_ec one int32 per image. */
int ecseg_segment_images(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W,
                         uint8_t* labels_raw, uint8_t* labels_post, int32_t* n_ec);
int ecseg_segment_images_dev(ecseg_ctx* h, const uint8_t* gray_dev, int n_img, int H, int W,
                             uint8_t* labels_raw_dev, uint8_t* labels_post_dev, int32_t* n_ec_dev);
/* ecseg_segment_images with two optional extra outputs (either may be NULL):
 *   tie_risk  one int32 per image: the number of pixels whose two largest uint8-quantised probabilities (the values
 *             np.argmax sees, src/utils.py:117-118) differ by at most 1 - the pixels whose label a last-bit difference between
 *             two float32 evaluations of the network (this library, TensorFlow, a CPU port) can flip; an upper bound on the
 *             raw-label disagreement of such evaluations for this image;
 *   probs     float32 (n_img, H, W, 4): the stitched probabilities themselves (patches2im_overlap, src/utils.py:116; never-
 *             written canvas pixels are 0), for comparison with the reference's own stitched output. */
int ecseg_segment_images_ex(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W,
                            uint8_t* labels_raw, uint8_t* labels_post, int32_t* n_ec, int32_t* tie_risk, float* probs);
/* Upper bound on images (of 35 windows) per internal U-Net launch group.  Default and n == 0: automatic - as many as fit
 * ~48 GB of activations, between 16 and 64 (16 for the canonical base-64 U-Net, 32 for base 32, 64 for base 16). */
int ecseg_set_images_per_group(ecseg_ctx* h, int n);
/* Tuning knobs: "overlap_post" (1: clean-up + count of group g run on a second stream beside the U-Net of group g+1;
 * 0 (default): everything on one stream - measured equal, the MFMA convs already fill the chip), "post_chunk"
 * (images per post-processing launch set), "images_per_group", "winograd" (3x3 / stride-1 / 'same'
 * convolutions: 2 (default) Winograd F(4x4,3x3) where the layer allows it (extents % 16, Cin % 4 and >= 8, Cout % 32), else
 * F(2x2,3x3); 1: F(2x2,3x3); 0: the direct implicit-GEMM kernel; all three are fp32 MFMA kernels; 3 (round 6): as 2, with the F(4x4) layers of whole
 * 64-channel output blocks and the 2x2 / stride-2 up-convolutions on the BF16 matrix pipe - both operands split exactly into three bf16 pieces, six of the
 * nine piece products, float32 accumulation: float32-accurate; the split filter images are made on the device when the option is set), "fuse_first" (1 (default): the network's first layer - Conv2D 3x3 'same', 1 -> 16 channels - is computed
 * by the 16 -> 16 conv_wino16_kernel convolution behind it, on the matrix cores, straight into that kernel's halo buffer; the 16-channel tensor
 * between the two never exists in memory; 0: conv_first_kernel writes it), "wino16" (1 (default):
 * F(2x2,3x3) layers with 16 or 32 input and output channels and extents >= 16 x 32 take conv_wino16_kernel - 16x16x4 MFMAs,
 * register output stage; 0: the 32-wide F(2x2) kernels), "wino_resident" (1 (default): the remaining F(2x2) layers with <= 32
 * input and <= 32 output channels keep their filter in registers and walk a tile row; 0: the streaming F(2x2) kernel - the
 * results of the two are identical), "wino4_split" (1 (default): an F(4x4) layer with exactly 32 output channels splits the 8 input
 * channels of a group between the two channel-half waves of a transform row; 0: the upper waves multiply the zero padding
 * of the 64-channel block), "fuse_pool" (1
 * (default): a MaxPooling2D(2x2, stride 2) that directly follows a Winograd (F(4x4) or F(2x2)) convolution is written by that
 * convolution's output stage; 0: separate max-pool kernel), "fuse_head" (1 (default): a 1x1 convolution with <= 4 output channels that
 * is the only reader of a 64-channel F(4x4) convolution (or of a 16 / 32-channel conv_wino16_kernel convolution) is computed by
 * that convolution's output stage and the feature tensor is never written; 0: separate head kernel), "crop" (1 (default): in ecseg_segment_images the last
 * full-resolution convolutions (F(4x4): 16x16 regions; conv_wino16_kernel: 16x32 blocks; 2x2 up-convolutions: input tiles)
 * compute only the parts of every window that the stitch (or the halo of the convolutions behind them) reads - 72 % of them
 * at 1040x1392, and their Winograd tiles read zeros outside the receptive field of those parts ("crop_mask", 1 (default)), so the result
 * is a pure function of the image: independent of batch position, window lanes and of what ran before; it agrees with whole windows
 * (0) to float32 rounding, i.e. labels can differ at near-ties of the quantised probabilities), "unet_lanes" (0 (default): automatic -
 * a launch group of <= 70 windows (one or two 1040x1392 images) runs its U-Net as two window lanes on their own streams, which fills
 * the half-empty last round of workgroups of the deep layers (one image: 11.3 -> 10.4 ms); 1..8: that many lanes; results are
 * bit-identical for every value), "blocking_wait" (1 (default): the wait for a launch group sleeps on an event created
 * with hipEventBlockingSync; 0: hipStreamSynchronize), "post_graph" (1: the
 * ~60 short kernels of meta_inference + count are captured once per (buffers, geometry) into a HIP graph and replayed;
 * 0 (default): plain launches - measured equal, the asynchronous launch queue already hides the launch gaps). */
int ecseg_set_option(ecseg_ctx* h, const char* key, int value);

/* ---- meta_preprocess (src/image_tools.py:86-101) ---------------------------------------------------------- */
/* img: n_img images (H, W, C) of uint8 (bytes_per_sample 1) or uint16 (2), C in {1,3,4}.  u16 -> u8 as
 * cv2.convertScaleAbs(alpha=255/65535); channel 2 when C > 1; Otsu; inverted when more than half is white.
 * gray_out: (n_img, H, W) uint8; inverted_out (optional) one flag per image. */
int ecseg_preprocess(ecseg_ctx* h, const void* img, int n_img, int H, int W, int C, int bytes_per_sample,
                     uint8_t* gray_out, int32_t* inverted_out);

/* ---- meta_segment of a batch in one call (src/utils.py:105-124 minus imread / imwrite, + src/metaseg.py:46) ---------- */
/* img as for ecseg_preprocess.  pre_process (src/utils.py:112) -> patches -> U-Net -> stitch -> argmax -> meta_inference ->
 * count_cc(I==3)[0] without the pre-processed images leaving the device in between: what `make metaseg` calls per batch.
 * gray_out (optional): the pre-processed images (src/utils.py:122-123 writes dapi/<name> from them), copied back under
 * the U-Net; labels_post, n_ec, tie_risk (optional) as for ecseg_segment_images_ex.  Results are identical to
 * ecseg_preprocess followed by ecseg_segment_images_ex. */
int ecseg_meta_segment(ecseg_ctx* h, const void* img, int n_img, int H, int W, int C, int bytes_per_sample,
                       uint8_t* gray_out, uint8_t* labels_post, int32_t* n_ec, int32_t* tie_risk);

/* Names the raw images of the ecseg_meta_segment call AFTER the coming one (page-locked memory, same layout, `bytes` in total):
 * the coming call uploads them on a stream of their own under its kernels, the call after it recognises them by (pointer, size)
 * and skips its own upload - the host <-> device copies of batch k + 1 overlap the kernels of batch k on ONE handle.  The
 * images must not change between the two calls; a call with other images uploads as always and drops what was sent ahead.
 * Results are those of calls without it. */
int ecseg_prefetch_input(ecseg_ctx* h, const void* img, size_t bytes);

/* ---- page-locked host buffers ------------------------------------------------------------------------------ */
/* Host pointers handed to any entry point may be ordinary (pageable) memory.  Buffers from ecseg_host_alloc make the
 * host <-> device copies DMA transfers (about 2x the pageable rate, and the gray_out copy of ecseg_meta_segment then really
 * overlaps the U-Net).  Free with ecseg_host_free before ecseg_destroy.  Unlike every other entry point these two may be called
 * from another thread while the handle is inside a call (they use only its device number); a failure is reported by the
 * return code alone (ECSEG_E_NOMEM / ECSEG_E_HIP), not through ecseg_last_error. */
int ecseg_host_alloc(ecseg_ctx* h, size_t bytes, void** out);
int ecseg_host_free(ecseg_ctx* h, void* p);

/* u16_to_u8 alone (src/image_tools.py:98-101, used by split_FISH_channels :142): count uint16 samples ->
 * uint8 with cv2.convertScaleAbs(alpha = 255/65535) rounding. */
int ecseg_u16_to_u8(ecseg_ctx* h, const uint16_t* in, long long count, uint8_t* out);

/* ---- stitched probabilities -> labels only (src/utils.py:116-118), for parity of the tail in isolation ---- */
/* probs: (n_img * n_patches, 256, 256, 4) float32 patch predictions in reference patch order. */
int ecseg_stitch_argmax(ecseg_ctx* h, const float* probs, int n_img, int H, int W, uint8_t* labels_raw);

/* ---- meta_inference(I) (src/image_tools.py:15-84) + count_cc(I==3)[0] on given label images -------------- */
int ecseg_meta_inference(ecseg_ctx* h, const uint8_t* labels_in, int n_img, int H, int W,
                         uint8_t* labels_out, int32_t* n_ec);
int ecseg_meta_inference_dev(ecseg_ctx* h, const uint8_t* labels_in_dev, int n_img, int H, int W,
                             uint8_t* labels_out_dev, int32_t* n_ec_dev);

/* ---- counting (src/image_tools.py:103-134) ---------------------------------------------------------------- */
/* masks are uint8 (non-zero = True), (n_img, H, W). */
/* count_cc (:114-119): n_out = number of 8-connected components; px_out = total pixels, or -1 where the
 * reference returns float 0.0 (no component, or a mask without any background pixel). */
int ecseg_count_cc(ecseg_ctx* h, const uint8_t* mask, int n_img, int H, int W, int32_t* n_out, int64_t* px_out);
/* Connected-component labels themselves: 0 background, else 1 + raster index of the component's first pixel
 * (connectivity 4 or 8); used by the parity tests of the union-find kernels. */
int ecseg_ccl_labels(ecseg_ctx* h, const uint8_t* mask, int n_img, int H, int W, int connectivity,
                     int32_t* labels_out);
/* count_colocalization (:126-134) */
int ecseg_count_colocalization(ecseg_ctx* h, const uint8_t* ob1, const uint8_t* ob2, int n_img, int H, int W,
                               int32_t* n_out);
/* count_HSR (:103-112) */
int ecseg_count_hsr(ecseg_ctx* h, const uint8_t* chrom, const uint8_t* fish, int n_img, int H, int W,
                    int size_threshold, int32_t* n_out);

/* ---- meta_overlay row (src/meta_overlay.py:59-95, src/image_tools.py:136-146) ----------------------------- */
/* labels: (n_img, H, W) uint8 post-processed labels (what read_seg loads, src/utils.py:125-132);
 * rgb: (n_img, H, W, C>=2) uint8, channel 0 red, channel 1 green; sensitivity: color_sensitivity.
 * out: n_img rows of 12 int64, in final CSV column order (src/meta_overlay.py:98-100):
 *   [0,1]  count_cc(ec)                 (n, px)   px = -1 stands for the reference's float 0.0
 *   [2,3]  count_cc(green & ~nuclei & ~chrom)
 *   [4,5]  count_cc(red   & ~nuclei & ~chrom)
 *   [6]    coloc(ec, green')   [7] coloc(ec, red')   [8] coloc(green' & ~chrom, red' & ~chrom)
 *   [9]    coloc(ec, red' & green')   [10] HSR(red)   [11] HSR(green)          (x' = x & ~nuclei) */
int ecseg_overlay(ecseg_ctx* h, const uint8_t* labels, const uint8_t* rgb, int n_img, int H, int W, int C,
                  int sensitivity, int hsr_size_threshold, int64_t* out);

/* ---- per-stage device timings of the last segment call (milliseconds, HIP events on the handle's stream) -- */
/* ECSEG_T_COUNT: device time of the kernels of the last ecseg_overlay / ecseg_preprocess / ecseg_count_* call (inputs
 * already resident, copies excluded). */
enum { ECSEG_T_TILE = 0, ECSEG_T_UNET = 1, ECSEG_T_TAIL = 2, ECSEG_T_POST = 3, ECSEG_T_COUNT = 4, ECSEG_T_N = 5 };
int ecseg_get_timings(ecseg_ctx* h, float* ms_out /* [ECSEG_T_N] */);
/* Average duration (ms) and launch count of the dominant kernel (MFMA conv) over the last segment/forward call,
 * measured with HIP events around every launch when profiling is enabled (adds a little launch overhead). */
int ecseg_set_kernel_profiling(ecseg_ctx* h, int enabled);
int ecseg_get_conv_profile(ecseg_ctx* h, double* total_ms, int64_t* launches, double* flops);
/* FLOPs the matrix cores actually executed in those launches (Winograd F(2x2,3x3) issues 16/36 of the algorithmic
 * multiplies of a 3x3 convolution; the direct kernel issues all of them). */
int ecseg_get_conv_executed_flops(ecseg_ctx* h, double* flops);
/* Per-launch records of the same profile, in launch order: plan operator index, kind, duration, algorithmic and executed
 * FLOPs.  kind bits 0-7 = kernel family (0 direct implicit GEMM conv_mfma_kernel, 1 Winograd F(2x2,3x3) conv_wino_kernel,
 * 2 Winograd F(4x4,3x3) conv_wino4_kernel, 3 filter-resident F(2x2) conv_wino_res_kernel, 4 F(2x2) on 16x16x4 MFMAs
 * conv_wino16_kernel); bit 8 (0x100): the launch also wrote the 2x2 max-pool that follows in the plan; bit 9 (0x200): it
 * also finished the 1x1 head that follows (the convolution's own output was not written); bit 10 (0x400): it also computed the
 * network's first layer (the plan operator in front of op_index) into its own halo ("fuse_first").  Returns the number of records
 * written (<= max_records) or a negative error. */
int ecseg_get_conv_launch_profile(ecseg_ctx* h, int max_records, int32_t* op_index, int32_t* kind, float* ms,
                                  double* flops, double* executed_flops);
/* Diagnostics only (-DECSEG_DIAG builds; ECSEG_E_UNSUPPORTED otherwise): up to 240 floats of in-kernel cycle stamps written by the ECSEG_WINO_STAMP build of the conv kernel. */
int ecseg_debug_peek(ecseg_ctx* h, float* out, int n);

/* ---- host-side byte codecs for the file I/O around the path (no GPU work) ---------------------------------- */
/* TIFF LZW (MSB-first, 9..12-bit codes, early change): inputs read by imread (src/utils.py:110) and the
 * dapi/<name>.tif written by cv2.imwrite (src/utils.py:122-123).  Return bytes written, or -1 on error. */
long long ecseg_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap);
long long ecseg_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap);

/* ---- whole-file readers / writers of `make metaseg` / `make meta_overlay` (host only; thread-safe, no handle) --------
 * Called through ctypes they run without the interpreter lock, so the I/O threads of ecseg_amd/metaseg.py scale over the
 * host cores.  Return ECSEG_OK, ECSEG_E_INVALID (bad argument / corrupt file), ECSEG_E_IO, or - readers only -
 * ECSEG_E_UNSUPPORTED for a valid TIFF layout that is left to the Python reader (tiles, BigTIFF, PackBits, float). */
/* np.save(labels/<stem>.npy, I.astype(int64)) (src/metaseg.py:53): byte-identical to numpy's format-1.0 writer. */
int ecseg_npy_write_i64(const char* path, const uint8_t* labels, int H, int W);
/* plt.imsave(labels/<stem>.png, I, cmap=ListedColormap([4 colours]), vmin=0, vmax=4) (src/metaseg.py:47-52): RGBA. */
int ecseg_png_write_labels(const char* path, const uint8_t* labels, int H, int W);
/* 8-bit gray / RGB / RGBA PNG (channels 1 / 3 / 4; zlib level 0..9, or -1: the settings of cv2.imwrite without parameters -
 * SUB filter, Z_BEST_SPEED, Z_RLE strategy): red/ and green/ of split_FISH_channels (src/image_tools.py:136-146). */
int ecseg_png_write(const char* path, const uint8_t* pixels, int H, int W, int channels, int level);
/* cv2.imwrite(red/<name>.png, cv2.bitwise_not(np.uint8(I[..., c]))) (src/image_tools.py:143-144): channel `channel` of an interleaved
 * 8-bit (H, W, channels) image as a gray PNG, inverted when invert != 0, with cv2's default encoder settings. */
int ecseg_png_write_channel(const char* path, const uint8_t* pixels, int H, int W, int channels, int channel, int invert);
/* np.load(labels/<stem>.npy) narrowed to uint8 while it is read (read_seg, src/utils.py:125-132; src/meta_overlay.py:59): C-order
 * 2-D integer arrays (the int64 file `make metaseg` writes, src/metaseg.py:53); ECSEG_E_UNSUPPORTED for any other layout. */
int ecseg_npy_label_info(const char* path, int* H, int* W);
int ecseg_npy_read_labels_u8(const char* path, uint8_t* dst, int H, int W);
/* cv2.imwrite(dapi/<name>.tif, gray) (src/utils.py:122-123): LZW + predictor 2, strips of 8192 / W rows; invert != 0
 * stores 255 - img (cv2.bitwise_not, src/utils.py:112). */
int ecseg_tiff_write_gray8(const char* path, const uint8_t* img, int H, int W, int invert);
/* skimage.io.imread of a baseline TIFF (src/utils.py:110): shape first, then the samples as native-endian (H, W, spp). */
int ecseg_tiff_info(const char* path, int* H, int* W, int* samples_per_pixel, int* bits_per_sample);
int ecseg_tiff_read(const char* path, void* dst, long long dst_bytes);

/* ---- the path's one exchange step: all-gather of per-image result records over RCCL (xGMI) ---------------------------
 * Image-parallel sharding needs no data-path collective (every image is independent: src/metaseg.py:42 is a serial loop);
 * at the end every rank contributes its block of fixed-size records and rank 0 writes ec_quantification.csv
 * (src/metaseg.py:44-57).  Record = ECSEG_RECORD_INT64 int64: [0] global image index (-1 = padding of the last shard)
 * [1] status (0 ok) [2] n_ec [3..14] the twelve fields of ecseg_overlay [15] reserved.  Shards are padded to equal length.
 * librccl.so is loaded on first use (no link-time dependency); ECSEG_E_UNSUPPORTED when it is not installed.
 * Every wait is bounded: ecseg_comm_create and the synchronous all-gathers return ECSEG_E_HIP with a message when the peers do
 * not answer within ECSEG_COMM_TIMEOUT_S seconds (environment variable, default 300) - the communicator is aborted and unusable
 * afterwards (ecseg_comm_destroy it).
 * Rendezvous: rank 0 obtains ECSEG_COMM_ID_BYTES from ecseg_comm_unique_id and hands them to the other ranks by any channel
 * of the host's (file, environment, socket); then every rank calls ecseg_comm_create (collective).  One process per GPU. */
#define ECSEG_RECORD_INT64  16
#define ECSEG_COMM_ID_BYTES 128
typedef struct ecseg_comm ecseg_comm;
int         ecseg_comm_unique_id(void* out, int out_bytes);
int         ecseg_comm_create(ecseg_comm** out, const void* unique_id, int rank, int world, int device_id);
void        ecseg_comm_destroy(ecseg_comm* c);
const char* ecseg_comm_last_error(void);          /* text of the calling thread's last failed ecseg_comm_* / all-gather call */
/* host buffers: n_records records in, world * n_records out (rank-major); synchronous */
int ecseg_allgather_records(ecseg_comm* c, const int64_t* send, int n_records, int64_t* recv);
/* device buffers; stream (hipStream_t as void*) non-null: returns once enqueued on it; null: own stream, synchronous */
int ecseg_allgather_records_dev(ecseg_comm* c, const int64_t* send_dev, int n_records, int64_t* recv_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECSEG_HIP_H */
