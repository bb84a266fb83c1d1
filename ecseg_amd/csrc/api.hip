// C ABI of libecseg_hip.so (see include/ecseg_hip.h).  Host-side orchestration only: buffer management, the layer plan
// interpreter, the image pipeline (tile -> U-Net -> stitch/argmax -> meta_inference -> count) and timing.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

using namespace ecseg;

namespace {

std::string g_create_error;

struct OpRt {                 // run-time form of one plan operator
    ecseg_op_desc d;
    int path;                 // which kernel family
    float* wt = nullptr;      // device weights in the layout the chosen kernel wants
    float* wt_wino = nullptr; // Winograd-transformed filter (16 points) when the op is eligible
    float* wt_wino4 = nullptr; // F(4x4,3x3) filter image (36 points, per-wave stage layout of wino4_kernel.hip)
    float* wt_wino16 = nullptr; // F(2x2,3x3) filter image of wino16_kernel.hip (Cin, Cout in {16, 32})
    void* wt_split1 = nullptr; // bf16x3 image of convs_kernel.hip (one-tap GEMM: the 2x2 / stride-2 up-convolutions), made from `wt` when "winograd" = 3 is asked for
    int s1_cin = 0, s1_np = 0; // its K and column count (0: not eligible)
    void* wt_wino4s = nullptr; // bf16x3 stage image of wino4s_kernel.hip, made on the device from wt_wino4 when "winograd" = 3 is asked for
    int w4_cin = 0, w4_cout = 0; // the layer wt_wino4 was made for
    int coutp_wino = 0;
    float* bias = nullptr;
    float* head_w4 = nullptr; // PATH_HEAD with <= 4 classes: [cin][4] / [4] zero-padded copies for the fused output stage
    float* head_b4 = nullptr;
    // demand-driven cropping: which part of this op's OUTPUT somebody reads, as a recipe applied to the stitch's per-window
    // bounding box: 'd' = grow by one pixel (a 3x3 convolution behind), 'h' = halve (a 2x2 / stride-2 up-convolution
    // behind); crop_ok = false: everything is needed (or unknown)
    bool crop_ok = false;
    std::string crop_code;
    float* scale = nullptr;   // AFFINE
    float* shift = nullptr;
    int cin_chunks = 0, coutp = 0;
    int subpixel = 0;         // CONVT kh x kw / stride 2 with k in {3, 4} as a 2x2-tap convolution over the input (relayout_convt_subpixel)
    // round 6: the same layers with >= 32 output channels run PHASE BY PHASE instead - output rows y = 2 j + c come from the kernel rows
    // kh = c + crop (mod 2) at input offsets (c + crop - kh) / 2: four forward convolutions of 1 or 2 taps per axis on the input extent
    // (9 instead of 16 tap x phase products at k = 3, no zero blocks staged or multiplied, no overhanging tile row), each with a
    // strided scatter store.  ph_wt[c_y * 2 + c_x] != null: that path; ph_R / ph_S / ph_pt / ph_pl: taps and leading pad of each phase
    float* ph_wt[4] = {nullptr, nullptr, nullptr, nullptr};
    int ph_R[4] = {0, 0, 0, 0}, ph_S[4] = {0, 0, 0, 0}, ph_pt[4] = {0, 0, 0, 0}, ph_pl[4] = {0, 0, 0, 0};
    double flops = 0.0;       // algorithmic 2*MAC per patch
};
enum { PATH_MFMA = 1, PATH_SMALL_CIN = 2, PATH_HEAD = 3, PATH_GENERIC = 4, PATH_OTHER = 5, PATH_TAP = 6 /* conv_mfma_tap_kernel: any taps / stride / dilation */ };

struct CropLut {                  // size: extent (pixels) of the tensors it applies to; start[w]: first entry of window w (entries are window-major)
    int32_t* dev = nullptr; int len = 0; int size = 0; std::vector<int> start;
};
struct StitchPlan {
    int n_pos = 0;
    int32_t* pos_dev = nullptr;   // (n_pos, 2) window origins (row, col), reference order
    int32_t* map_dev = nullptr;   // (H*W) source map
    // demand-driven cropping: per window the bounding box (y0, y1, x0, x1) of the pixels the stitch reads, and the region
    // lists built from it on first use, keyed by the op's crop recipe (OpRt::crop_code)
    std::vector<int> box;
    std::map<std::string, CropLut> luts;
    std::map<std::string, int32_t*> boxes;   // per recipe: device (n_pos, 4) need boxes (get_crop_box)
};

}  // namespace

struct ecseg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // post-processing of group g overlaps the U-Net of group g+1
    std::vector<hipStream_t> lane_streams;   // window lanes of small batches (run_plan)
    std::vector<hipEvent_t> lane_events;
    int unet_lanes = 0;              // 0: automatic (2 lanes up to lane_auto_windows windows per group, else 1)
    int lane_auto_windows = 70;
    bool lanes_ok = false;           // the input's and the output's buffers hold nothing of another per-window size (run_plan)
    std::string err;
    char devname[256] = {0};

    bool has_model = false;
    std::vector<ecseg_tensor_desc> tensors;
    std::vector<OpRt> ops;
    int n_buffers = 0;
    std::vector<size_t> buf_floats;       // floats per patch of every buffer
    std::vector<float*> bufs;
    int cap_patches = 0;
    std::vector<float*> dev_allocs;       // weight allocations (freed on reload / destroy)
    float* zero_page = nullptr;           // 1 KiB: 64 bytes of zeros + diagnostics scratch
    int input_tensor = -1, output_tensor = -1;
    std::vector<int> consumers;   // per tensor: number of operators reading it
    double flops_per_patch = 0.0, mfma_flops_per_patch = 0.0;

    // image pipeline
    int images_per_group = 0;    // images (of 35 windows) per U-Net launch; 0 = automatic: ~48 GB of activations (windows_per_group)
    std::map<std::pair<int, int>, StitchPlan> stitch;
    uint8_t* d_gray = nullptr; size_t d_gray_cap = 0;
    uint8_t* d_raw = nullptr; size_t d_raw_cap = 0;
    int32_t* d_tie = nullptr; size_t d_tie_cap = 0;        // per-image tie-risk counts of the last segment call
    int32_t* d_tie_sh = nullptr; size_t d_tie_sh_cap = 0;  // their replicated counters (one launch group)
    float* d_sprobs = nullptr; size_t d_sprobs_cap = 0;    // stitched probabilities of one launch group (ecseg_segment_images_ex)
    uint8_t* d_post = nullptr; size_t d_post_cap = 0;
    uint8_t* d_aux8 = nullptr; size_t d_aux8_cap = 0;      // second uint8 input (masks, rgb)
    // ecseg_prefetch_input: the raw images of the NEXT ecseg_meta_segment call, uploaded on their own stream while this call computes
    uint8_t* d_pre = nullptr; size_t d_pre_cap = 0;
    const void* pre_host = nullptr; size_t pre_bytes = 0;  // what d_pre holds (host pointer + size are the key); nullptr: nothing
    const void* next_host = nullptr; size_t next_bytes = 0;   // registered by ecseg_prefetch_input for the coming call to send ahead
    hipStream_t stream_in = nullptr;
    hipEvent_t ev_pre = nullptr;
    uint8_t* d_u8in = nullptr; size_t d_u8in_cap = 0;      // uint8 patches of forward_patches
    int32_t* d_i32 = nullptr; size_t d_i32_cap = 0;        // small int outputs
    long long* d_i64 = nullptr; size_t d_i64_cap = 0;
    float* d_probs_in = nullptr; size_t d_probs_cap = 0;
    uint32_t* d_hist = nullptr; size_t d_hist_cap = 0;
    PostWorkspace ws{};
    size_t ws_list_bytes = 0;
    // meta_inference is ~60 short dependent kernels: captured once per (buffers, geometry) into a HIP graph and replayed
    struct PostGraph { uint8_t* img; int32_t* nec; int n, H, W; hipStream_t s; hipGraphExec_t exec; unsigned long long stamp; };
    std::vector<PostGraph> post_graphs;
    unsigned long long post_graph_clock = 0;
    int post_graph = 0;       // measured +-0 % at 4 / 16 / 64 images per call (the launch queue already hides the gaps): off by default
    int post_chunk = 64;
    int overlap_post = 0;
    int blocking_wait = 1;    // the long waits (a whole launch group) sleep on a blocking event instead of spinning on the stream
    hipEvent_t ev_block = nullptr;
    int fuse_pool = 1;        // 2x2 max-pool written by the producing F(4x4) convolution's output stage
    int crop = 1;             // segment path: skip output regions of the last full-resolution convolutions that the stitch never reads
    int crop_mask = 1;        // cropped plan: Winograd kernels read zeros outside the receptive field of the needed outputs (0: A/B measurements only - results then depend on stale buffer contents in the last bits)
    int fuse_head = 1;        // 1x1 head (<= 4 classes) computed by the output stage of the last F(4x4) convolution
    int wino4_rowpass = 1;    // 1 (default since round 6: +1.8 % on the base-64 step, 1 - 6 % per layer): F(4x4) fp32 layers (no lone 32-channel block) on conv_wino4r_kernel - row transform once per workgroup (round 6 A/B)
    int use_winograd = 2;     // 0 direct, 1 Winograd F(2x2,3x3), 2 F(4x4,3x3) where eligible (else F(2x2)), 3: F(4x4) with 3-way bf16 split operands on the bf16 matrix pipe where eligible (else as 2)
    int wino4_split = 1;      // F(4x4) layers with exactly 32 output channels: split-K over the channel-half waves
    int wino16 = 1;           // F(2x2) layers with 16 / 32 input and output channels: conv_wino16_kernel (16x16x4 MFMA, register output stage)
    int wino_resident = 1;    // F(2x2) layers with <= 32 input and output channels: filter-resident kernel (conv_wino_res_kernel)
    int fuse_first = 1;       // the network's first layer (3x3, 1 -> 16 channels) computed into the halo of the 16 -> 16 convolution behind it (conv_wino16_kernel FIRST)

    // timing
    hipEvent_t ev[ECSEG_T_N + 1] = {};
    float stage_ms[ECSEG_T_N] = {};
    bool profile_kernels = false;
    std::vector<hipEvent_t> prof_events;   // pairs
    std::vector<hipEvent_t> grp_events;    // 6 per image group of segment_dev
    size_t prof_used = 0;
    double prof_flops = 0.0, prof_exec_flops = 0.0;
    struct ProfRec { int op; int kind; double flops, exec_flops; float ms; };   // kind: 0 direct, 1 F(2x2), 2 F(4x4), 3 filter-resident F(2x2), 4 F(2x2) on 16x16x4 MFMAs (wino16), 5 F(4x4) with bf16x3 split operands (wino4s; exec_flops = the fp32-equivalent products, each issued as 6 bf16 products), 6 one-tap GEMM with bf16x3 split operands (convs_kernel)
    std::vector<ProfRec> prof_recs;        // one per profiled launch of the last segment / forward call
    double last_conv_ms = 0.0; long long last_conv_launches = 0; double last_conv_flops = 0.0, last_conv_exec_flops = 0.0;
};

namespace {

int fail(ecseg_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}
int fail_hip(ecseg_ctx* h, hipError_t e, const char* what) {
    return fail(h, ECSEG_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail_hip((h), _e, #call); } while (0)

template <typename T>
int ensure(ecseg_ctx* h, T*& ptr, size_t& cap, size_t need_elems) {
    if (need_elems <= cap && ptr) return ECSEG_OK;
    if (ptr) { hipError_t e = hipFree(ptr); ptr = nullptr; cap = 0; if (e != hipSuccess) return fail_hip(h, e, "hipFree"); }
    if (need_elems == 0) need_elems = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), need_elems * sizeof(T));
    if (e != hipSuccess) { ptr = nullptr; return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    cap = need_elems;
    return ECSEG_OK;
}

TView view_of(const ecseg_ctx* h, int t) {
    const ecseg_tensor_desc& d = h->tensors[t];
    TView v;
    v.p = h->bufs[d.buffer] + d.c_offset;
    v.h = d.h; v.w = d.w; v.c = d.c; v.cs = d.c_stride;
    return v;
}

void free_model(ecseg_ctx* h) {
    for (float* p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear();
    for (float* p : h->bufs) if (p) (void)hipFree(p);
    h->bufs.clear();
    h->cap_patches = 0;
    h->ops.clear(); h->tensors.clear();
    h->has_model = false;
}

int upload(ecseg_ctx* h, const std::vector<float>& host, float** dev) {
    float* p = nullptr;
    const size_t n = host.empty() ? 1 : host.size();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float));
    if (e != hipSuccess) return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc(weights): ") + hipGetErrorString(e));
    h->dev_allocs.push_back(p);
    if (!host.empty()) {
        e = hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail_hip(h, e, "hipMemcpy(weights)");
    }
    *dev = p;
    return ECSEG_OK;
}

// Keras HWIO kernel -> wt[tap][chunk][half][NP][4] (zero padded; padded chunk / tap pitches, see common.h)
std::vector<float> relayout_conv(const float* w, int R, int S, int cin, int cout, int chunks, int np) {
    const size_t cp = (size_t)wt_chunk_pitch(np), tp = (size_t)wt_tap_pitch(np, chunks);
    std::vector<float> o((size_t)R * S * tp, 0.f);
    for (int t = 0; t < R * S; ++t)
        for (int ci = 0; ci < cin; ++ci) {
            const int chunk = ci / 8, hh = (ci % 8) / 4, e = ci % 4;
            const float* src = w + ((size_t)t * cin + ci) * cout;
            float* dst = o.data() + (size_t)t * tp + (size_t)chunk * cp + ((size_t)hh * np) * 4 + e;
            for (int co = 0; co < cout; ++co) dst[(size_t)co * 4] = src[co];
        }
    return o;
}
// Filter image of conv_wino16_kernel: MFMA A fragments [point 16][Cin / 16][Cout / 16][lane 64][k-step 4]; lane =
// (channel quad kq = lane / 16, output channel m = lane % 16) holds U[point][16 kc + 4 kq + s][16 nb + m] for s = 0..3
std::vector<float> relayout_wino16(const std::vector<float>& u, int cin, int cout) {
    const int KC = cin / 16, NB = cout / 16;
    std::vector<float> o((size_t)16 * KC * NB * 64 * 4);
    for (int pt = 0; pt < 16; ++pt)
        for (int kc = 0; kc < KC; ++kc)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int kq = lane >> 4, m = lane & 15;
                        o[((((size_t)pt * KC + kc) * NB + nb) * 64 + lane) * 4 + s] =
                            u[((size_t)pt * cin + 16 * kc + 4 * kq + s) * cout + 16 * nb + m];
                    }
    return o;
}
// Winograd F(2x2,3x3) filter transform U = G g G^T (float64), as 16 "taps" in HWIO order [a*4+b][cin][cout]
std::vector<float> winograd_filter(const float* w, int cin, int cout) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> u((size_t)16 * cin * cout);
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co) {
            double g[3][3], t[4][3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) g[r][c] = w[((size_t)(r * 3 + c) * cin + ci) * cout + co];
            for (int a = 0; a < 4; ++a)
                for (int c = 0; c < 3; ++c) t[a][c] = G[a][0] * g[0][c] + G[a][1] * g[1][c] + G[a][2] * g[2][c];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b)
                    u[((size_t)(a * 4 + b) * cin + ci) * cout + co] =
                        (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        }
    return u;
}

// Winograd F(4x4,3x3) filter transform U = G g G^T (float64, 36 points) written straight in the per-wave stage layout
// of conv_wino4_kernel: wt4[cout block of 64][stage = 4 input channels][wave = half * 6 + xi][nu][h][cout 32][e], where
// stage s of 8-channel group s / 2 holds input channels 8 (s / 2) + 4 h + 2 (s % 2) + e.
std::vector<float> winograd4_filter(const float* w, int cin, int cout) {
    // G row of a finite point p: [1, p, p^2] / prod_{q != p} (p - q) over the finite points {0, +-a, +-b}; infinity: [0, 0, 1]
    // (textbook values for a = 1, b = 2: 1/4, -1/6, 1/24)
    const double a = W4_PA, b = W4_PB, a2 = a * a, b2 = b * b;
    const double n0 = a2 * b2, na = 2 * a2 * (a2 - b2), nb_ = 2 * b2 * (b2 - a2);
    const double G[6][3] = {{1 / n0, 0, 0},           {1 / na, a / na, a2 / na},   {1 / na, -a / na, a2 / na},
                            {1 / nb_, b / nb_, b2 / nb_}, {1 / nb_, -b / nb_, b2 / nb_}, {0, 0, 1}};
    // zero padded to whole 64-channel output blocks and whole 8-channel input groups (Cout % 64 == 32: the second
    // channel-half waves of the last block multiply zeros; Cin % 8 == 4: the second half of the last group is zero)
    const int nblk = (cout + 63) / 64, nstages = 2 * ((cin + 7) / 8);
    std::vector<float> o((size_t)nblk * nstages * 12 * 768, 0.f);
    for (int ci = 0; ci < cin; ++ci) {
        const int grp = ci / 8, r8 = ci % 8;
        const int hh = r8 / 4, ss = (r8 % 4) / 2, e = r8 % 2;
        const int stage = 2 * grp + ss;
        for (int co = 0; co < cout; ++co) {
            double g[3][3], t[6][3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) g[r][c] = w[((size_t)(r * 3 + c) * cin + ci) * cout + co];
            for (int a = 0; a < 6; ++a)
                for (int c = 0; c < 3; ++c) t[a][c] = G[a][0] * g[0][c] + G[a][1] * g[1][c] + G[a][2] * g[2][c];
            const int nb = co / 64, half = (co % 64) / 32, m = co % 32;
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double u = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                    // per (block, stage, wave): [point pair b / 2][lane = hh * 32 + m][point b % 2][channel e] - ONE ds_read_b128 per lane
                    // and point pair delivers the B operands of four MFMAs (round 4: three 16-byte reads per stage instead of six
                    // 8-byte ones; an LDS read beside the MFMA stream costs the matrix pipe ~14 cycles whatever its width)
                    const size_t idx = (((size_t)nb * nstages + stage) * 12 + (half * 6 + a)) * 768 + ((((size_t)(b >> 1) * 64 + hh * 32 + m) * 2 + (b & 1)) * 2) + e;
                    o[idx] = (float)u;
                }
        }
    }
    return o;
}

// Keras Conv2DTranspose kernel (kh, kw, out, in) -> one-tap GEMM filter over N = (a*kT + b) * coutp + co
std::vector<float> relayout_convt(const float* w, int kT, int cin, int cout, int chunks, int coutp) {
    const int np = kT * kT * coutp;
    const size_t cp = (size_t)wt_chunk_pitch(np), tp = (size_t)wt_tap_pitch(np, chunks);
    std::vector<float> o(tp, 0.f);
    for (int ab = 0; ab < kT * kT; ++ab)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                const int chunk = ci / 8, hh = (ci % 8) / 4, e = ci % 4;
                o[(size_t)chunk * cp + ((size_t)hh * np + (size_t)ab * coutp + co) * 4 + e] = w[((size_t)ab * cout + co) * cin + ci];
            }
    return o;
}

// Keras Conv2DTranspose kernel (k, k, out, in), stride 2, k in {3, 4}, as the filter of a 2x2-tap convolution over the INPUT
// that produces all four output phases of a 2x2 output block at once: output (2 i + a, 2 j + b) of the full (uncropped)
// result sums w[a - 2 d][b - 2 e] x in(i + d, j + e) over d, e in {-1, 0} (taps with kernel index outside [0, k) are zero:
// 5 of 16 at k = 3, none at k = 4).  Layout as relayout_conv with tap t = (d + 1) * 2 + (e + 1) and N = (a * 2 + b) * coutp + co.
std::vector<float> relayout_convt_subpixel(const float* w, int k, int cin, int cout, int chunks, int coutp) {
    const int np = 4 * coutp;
    const size_t cp = (size_t)wt_chunk_pitch(np), tp = (size_t)wt_tap_pitch(np, chunks);
    std::vector<float> o((size_t)4 * tp, 0.f);
    for (int d = -1; d <= 0; ++d)
        for (int e = -1; e <= 0; ++e)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int kh = a - 2 * d, kw = b - 2 * e;
                    if (kh >= k || kw >= k) continue;
                    const int t = (d + 1) * 2 + (e + 1);
                    for (int co = 0; co < cout; ++co)
                        for (int ci = 0; ci < cin; ++ci) {
                            const int chunk = ci / 8, hh = (ci % 8) / 4, ee = ci % 4;
                            o[(size_t)t * tp + (size_t)chunk * cp + ((size_t)hh * np + (size_t)(a * 2 + b) * coutp + co) * 4 + ee] =
                                w[((size_t)(kh * k + kw) * cout + co) * cin + ci];
                        }
                }
    return o;
}

int ensure_patches(ecseg_ctx* h, int n) {
    if (n <= h->cap_patches) return ECSEG_OK;
    for (float*& p : h->bufs) { if (p) (void)hipFree(p); p = nullptr; }
    h->bufs.assign(h->n_buffers, nullptr);
    h->cap_patches = 0;
    for (int b = 0; b < h->n_buffers; ++b) {
        const size_t bytes = std::max<size_t>(h->buf_floats[b], 4) * (size_t)n * sizeof(float);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->bufs[b]), bytes);
        if (e != hipSuccess) return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc(activations): ") + hipGetErrorString(e));
    }
    h->cap_patches = n;
    return ECSEG_OK;
}

// Windows per U-Net launch group.  An explicit images_per_group counts 35-window images (1040 x 1392).  Automatic: as many
// windows as fit ~48 GB of activations, between 16 and 64 such images - 16 for the canonical base-64 U-Net (82 MB per
// window), 32 for base 32, 64 for base 16, whose short kernels gain 5-6 % from the longer launches (base-16 bench model:
// 799 / 836 / 851 images/s at 16 / 32 / 64 images per group).
int windows_per_group(const ecseg_ctx* h) {
    if (h->images_per_group > 0) return h->images_per_group * 35;
    size_t per_window = 0;
    for (size_t f : h->buf_floats) per_window += std::max<size_t>(f, 4) * sizeof(float);
    const size_t budget = (size_t)48 << 30;
    size_t img = per_window ? budget / (per_window * 35) : 16;
    int g = 16;
    while (g < 64 && (size_t)(2 * g) <= img) g *= 2;
    return g * 35;
}

hipEvent_t* prof_pair(ecseg_ctx* h) {
    if (h->prof_used + 2 > h->prof_events.size()) {
        for (int k = 0; k < 2; ++k) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            h->prof_events.push_back(e);
        }
    }
    hipEvent_t* p = &h->prof_events[h->prof_used];
    h->prof_used += 2;
    return p;
}

// Run the whole plan on n patches whose input tensor has already been written.
// Region list of a crop recipe: per window the stitch's bounding box is grown / halved as the recipe says, then covered
// by 16x16 regions whose origins are multiples of 4 pixels (the Winograd tile) and stay inside the tensor.  Entry =
// window << 16 | (y origin / 4) << 8 | (x origin / 4).  len 0: nothing to gain (or an extent the kernel cannot take).
// Need box of window i under a crop recipe: the stitch's bounding box, grown by one pixel per 'd' (a 3x3 convolution
// behind) and halved per 'h' (a stride-2 up-convolution behind).  False: nothing of this window is ever read.
bool recipe_box(const StitchPlan* sp, const std::string& code, int i, int b[4]) {
    for (int k = 0; k < 4; ++k) b[k] = sp->box[4 * i + k];
    if (b[1] < 0) return false;
    int sz = 256;
    for (char c : code) {
        if (c == 'd') { b[0] = std::max(b[0] - 1, 0); b[1] = std::min(b[1] + 1, sz - 1); b[2] = std::max(b[2] - 1, 0); b[3] = std::min(b[3] + 1, sz - 1); }
        else { sz /= 2; for (int k = 0; k < 4; ++k) b[k] /= 2; }
    }
    return true;
}

// Device table (n_pos, 4) of the need boxes of a recipe (ConvParams::in_box); null on allocation failure (no masking).
const int32_t* get_crop_box(StitchPlan* sp, const std::string& code) {
    auto it = sp->boxes.find(code);
    if (it != sp->boxes.end()) return it->second;
    std::vector<int32_t> t((size_t)sp->n_pos * 4);
    for (int i = 0; i < sp->n_pos; ++i) {
        int b[4];
        if (!recipe_box(sp, code, i, b)) { b[0] = 1; b[1] = 0; b[2] = 1; b[3] = 0; }      // empty: everything reads as zero
        for (int k = 0; k < 4; ++k) t[4 * i + k] = b[k];
    }
    int32_t* dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dev), t.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(dev, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        if (dev) (void)hipFree(dev);
        dev = nullptr;
    }
    sp->boxes.emplace(code, dev);
    return dev;
}

const CropLut* get_crop_lut(StitchPlan* sp, const std::string& code, int rh = 16, int rw = 16) {
    const std::string key = code + ":" + std::to_string(rh) + "x" + std::to_string(rw);
    auto it = sp->luts.find(key);
    if (it != sp->luts.end()) return &it->second;
    CropLut cl;
    int size = 256;
    for (char c : code) if (c == 'h') size /= 2;
    cl.size = size;
    std::vector<int32_t> lut;
    const int rdim[2] = {rh, rw};
    if (size >= 16 && size % 16 == 0) {
        for (int i = 0; i < sp->n_pos; ++i) {
            int b[4];
            cl.start.push_back((int)lut.size());
            if (!recipe_box(sp, code, i, b)) continue;         // nothing of this window is ever read
            int o[2], nr[2];
            for (int a = 0; a < 2; ++a) {
                const int lo = b[2 * a], hi = b[2 * a + 1];
                const int R = rdim[a];
                o[a] = lo & ~3;                                // tile-aligned start
                nr[a] = (hi - o[a]) / R + 1;
                if (R * nr[a] >= size) { nr[a] = (size + R - 1) / R; o[a] = 0; }
                else if (o[a] + R * nr[a] > size) o[a] = size - R * nr[a];
            }
            for (int ry = 0; ry < nr[0]; ++ry)
                for (int rx = 0; rx < nr[1]; ++rx) lut.push_back((i << 16) | (((o[0] + rh * ry) / 4) << 8) | ((o[1] + rw * rx) / 4));
        }
        cl.start.push_back((int)lut.size());
        if (lut.size() >= (size_t)sp->n_pos * ((size + rh - 1) / rh) * ((size + rw - 1) / rw)) lut.clear();     // nothing to gain
    }
    if (!lut.empty()) {
        if (hipMalloc(reinterpret_cast<void**>(&cl.dev), lut.size() * sizeof(int32_t)) != hipSuccess ||
            hipMemcpy(cl.dev, lut.data(), lut.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            if (cl.dev) (void)hipFree(cl.dev);
            cl.dev = nullptr;
        } else cl.len = (int)lut.size();
    }
    return &sp->luts.emplace(key, cl).first->second;
}

// `crop`: the stitch that will read the model output (segment path), or null when every output pixel matters.
// Window lanes (round 4): `cnt` >= 0 runs the plan on windows [w0, w0 + cnt) of the `n_all` windows whose input has been
// written, on stream `lane_s` - several lanes of one small batch run beside each other on their own streams, so the
// half-empty last round of workgroups of one lane's deep layers (35 windows: 288 / 560 workgroups on 256 CUs) is filled by
// another lane's next layer.  Same kernels, same per-window arithmetic: results do not depend on the lanes.  A lane is
// either whole images or a part of ONE image (cropped launches then take a slice of the window-major region list).
struct LaneSpec { int w0, cnt; hipStream_t s; };

// Launch op `oi` of the plan (and the ops its kernel takes over: a following pool / head - `oi` is advanced past them) for one
// lane (null: all n_all windows on the main stream).
int run_plan_op(ecseg_ctx* h, size_t& oi, int n_all, StitchPlan* crop, const LaneSpec* ls) {
    hipStream_t s = ls ? ls->s : h->stream;
    const bool lane = ls != nullptr;
    const int n = lane ? ls->cnt : n_all;
    const int w0 = lane ? ls->w0 : 0, cnt = n;
    const bool part = lane && crop && (w0 % crop->n_pos != 0 || cnt % crop->n_pos != 0);
    const int wbase = part ? (w0 / crop->n_pos) * crop->n_pos : w0;     // cropped launches of a partial lane: views at the image's first window
    if (part && w0 + cnt > wbase + crop->n_pos) return fail(h, ECSEG_E_INVALID, "window lane crosses an image boundary");
    // A lane's windows of tensor t.  Buffers are shared by tensors of different sizes (liveness re-use), so lanes that run at
    // different points of the plan must not share ANY byte of a buffer: a lane owns the slice [w0, w0 + cnt) x (the buffer's
    // floats per window) of every buffer and packs its windows of whatever tensor lives there at the slice's start.  The
    // model input and output keep the plain window order (the tiling kernel / the stitch address them for all lanes at
    // once): their buffers hold nothing of another size (checked at load: lanes_ok).  `rebase`: views of a partial
    // lane's cropped launch - the kernel adds (window index within the image) x (window size) itself.
    auto at = [&](int t, bool rebase = false) {
        if (t < 0) return TView{};
        TView v = view_of(h, t);
        if (!lane) return v;
        const ecseg_tensor_desc& td = h->tensors[t];
        const ptrdiff_t hwc = (ptrdiff_t)v.h * v.w * v.cs;
        const bool io = t == h->input_tensor || t == h->output_tensor;
        ptrdiff_t off = io ? (ptrdiff_t)w0 * hwc : (ptrdiff_t)w0 * (ptrdiff_t)std::max<size_t>(h->buf_floats[td.buffer], 4);
        if (rebase) off -= (ptrdiff_t)(w0 - wbase) * hwc;
        v.p += off;
        return v;
    };
    // The network's first layer (Conv2D 3x3 'same', 1 -> 16 channels) in front of a 16 -> 16 convolution that conv_wino16_kernel
    // takes and that is its only reader: the second convolution's launch computes the first one into its own halo (FIRST); the
    // 16-channel tensor between them is never written.  `first` != null below: op oi - 1 rides on op oi's launch.
    const OpRt* first = nullptr;
    if (h->fuse_first && h->use_winograd && h->wino16 && oi + 1 < h->ops.size()) {
        const OpRt& a = h->ops[oi];
        const OpRt& b = h->ops[oi + 1];
        const ecseg_tensor_desc& ta = h->tensors[a.d.in0];
        const ecseg_tensor_desc& tm = h->tensors[a.d.out];
        const ecseg_tensor_desc& tb = h->tensors[b.d.out];
        auto core = [](const ecseg_op_desc& q) { return q.act <= ECSEG_ACT_ELU && q.act != ECSEG_ACT_SOFTMAX && !(q.act == ECSEG_ACT_ELU && q.alpha != 1.f); };
        if (a.d.op == ECSEG_OP_CONV && a.path == PATH_SMALL_CIN && a.d.kh == 3 && a.d.kw == 3 && a.d.stride == 1 && a.d.pad_top == 1 &&
            a.d.pad_left == 1 && a.d.dilation <= 1 && ta.c == 1 && ta.c_stride == 1 && (tm.c == 16 || tm.c == 32) && tm.h == ta.h && tm.w == ta.w && core(a.d) &&
            b.d.op == ECSEG_OP_CONV && b.path == PATH_MFMA && b.wt_wino16 != nullptr && b.d.in0 == a.d.out && h->consumers[a.d.out] == 1 &&
            a.d.out != h->output_tensor && b.d.kh == 3 && b.d.kw == 3 && b.d.stride == 1 && b.d.pad_top == 1 && b.d.pad_left == 1 &&
            tb.c == tm.c && tb.h == tm.h && tb.w == tm.w && tm.w % 4 == 0 && core(b.d) && !(crop && h->crop && b.crop_ok)) {
            first = &a;
            ++oi;
        }
    }
    {
        const OpRt& o = h->ops[oi];
        const ecseg_op_desc& d = o.d;
        const TView in = first ? at(first->d.in0) : at(d.in0), out = at(d.out);
        hipError_t e = hipSuccess;
        switch (d.op) {
            case ECSEG_OP_CONV:
            case ECSEG_OP_CONVT: {
                const bool softmax = d.act == ECSEG_ACT_SOFTMAX;
                const int act = (softmax && o.path != PATH_HEAD) ? ECSEG_ACT_LINEAR : d.act;
                if (o.path == PATH_MFMA) {
                    const size_t oi_first = oi;                 // (fusions below advance oi)
                    ConvParams p{};
                    p.in = in; p.out = out; p.wt = o.wt; p.bias = o.bias; p.n = n;
                    if (d.op == ECSEG_OP_CONV && d.kh == 1 && d.kw == 1 && in.h == 1 && in.w == 1 && out.h == 1 && out.w == 1) {
                        // Dense layer: the batch is the GEMM's M dimension - one "patch" whose pixels are the samples
                        p.in.w = n; p.out.w = n; p.n = 1;
                    }
                    p.act = act; p.alpha = d.alpha; p.cin_chunks = o.cin_chunks; p.coutp = o.coutp; p.zero = h->zero_page;
                    if (d.op == ECSEG_OP_CONV) {
                        p.R = d.kh; p.S = d.kw; p.pad_top = d.pad_top; p.pad_left = d.pad_left; p.convt = 0; p.stride = d.stride;
                    } else if (o.subpixel) {
                        // 2x2 taps over input rows / columns (i - 1, i); tiles walk one position past the input (the last output
                        // row / column of the full result comes from tap d = -1 alone)
                        p.R = 2; p.S = 2; p.pad_top = 1; p.pad_left = 1; p.convt = 1; p.kT = 2;
                        // (that position only matters when a kept output row / column lies at or beyond 2 x the input extent:
                        // 4x4 'same' and every 'valid' layer, not 3x3 'same' - whose 16 x 16 inputs then tile exactly)
                        p.convt_ext = (out.h + d.pad_top > 2 * in.h || out.w + d.pad_left > 2 * in.w) ? 1 : 0;
                        p.crop_top = d.pad_top; p.crop_left = d.pad_left;
                        // tap t = (d + 1) * 2 + (e + 1), phase (a, b): kernel index (a - 2 d, b - 2 e) >= k means a zero block (relayout_convt_subpixel)
                        for (int dd = -1; dd <= 0; ++dd)
                            for (int ee = -1; ee <= 0; ++ee)
                                for (int a = 0; a < 2; ++a)
                                    for (int b = 0; b < 2; ++b)
                                        if (a - 2 * dd >= d.kh || b - 2 * ee >= d.kw) p.tap_zero_mask |= 1 << (((dd + 1) * 2 + (ee + 1)) * 4 + a * 2 + b);
                    } else {
                        p.R = 1; p.S = 1; p.pad_top = 0; p.pad_left = 0; p.convt = 1; p.kT = d.kh;
                        p.crop_top = d.pad_top; p.crop_left = d.pad_left;
                    }
                    bool rebased = false;                      // views of this launch start at the image's first window
                    // region list of a cropped launch; a partial lane takes the slice of its windows (entries keep their window
                    // index within the image, so the views go back to the image's first window)
                    auto use_lut = [&](const CropLut* cl) {
                        p.lut = cl->dev; p.lut_len = cl->len; p.per_image = crop->n_pos;
                        if (part) {
                            const int a = cl->start[w0 - wbase], b = cl->start[w0 - wbase + cnt];
                            p.lut = cl->dev + a; p.lut_len = b - a; p.n = crop->n_pos;
                            rebased = true;
                            p.box_first = 0;
                            p.in = at(d.in0, true); p.out = at(d.out, true);
                        }
                    };
                    hipEvent_t* ev = h->profile_kernels ? prof_pair(h) : nullptr;
                    if (ev) (void)hipEventRecord(ev[0], s);
                    double computed = 1.0;                     // fraction of the layer a cropped launch really computes
                    // conv_wino4 / conv_wino16 implement activation codes 0..6 with ELU's alpha = 1 (device_util.h: apply_act_core)
                    const bool act_core_ok = act <= ECSEG_ACT_ELU && !(act == ECSEG_ACT_ELU && d.alpha != 1.f);
                    if (first) {
                        p.first_w = first->wt; p.first_b = first->bias; p.first_act = first->d.act; p.first_alpha = first->d.alpha;
                    }
                    const bool wino4 = !first && h->use_winograd >= 2 && o.wt_wino4 && act_core_ok && conv_wino4_supported(p);   // (wt_wino* exist only for stride-1 3x3 'same' layers)
                    const bool wino = !wino4 && h->use_winograd && o.wt_wino && out.h >= 4 && out.w >= 8;
                    bool w16 = false, split1 = false;
                    // a 3x3 convolution of the cropped chain on a Winograd kernel reads its input only inside the receptive field
                    // of the outputs somebody needs (ConvParams::in_box): results do not depend on what a cropped producer left
                    // outside it
                    const bool crop_on = crop && h->crop && o.crop_ok && (part || n % crop->n_pos == 0);
                    if (crop_on && h->crop_mask && d.op == ECSEG_OP_CONV && d.kh == 3 && d.kw == 3 && (wino4 || wino)) {
                        p.in_box = get_crop_box(crop, o.crop_code + "d");
                        p.per_image = crop->n_pos;
                        p.box_first = part ? w0 - wbase : 0;
                    }
                    {
                        const int npt = p.convt ? p.kT * p.kT * o.coutp : o.coutp;
                        p.wt_chunk_stride = wt_chunk_pitch(npt); p.wt_tap_stride = wt_tap_pitch(npt, o.cin_chunks);
                    }
                    // Winograd output stages can write the 2x2 max-pool of their result themselves: a MaxPooling2D(2x2, stride
                    // 2) that follows directly (even extents, its own buffer) is then done with the convolution
                    auto fuse_following_pool = [&]() {
                        if (oi + 1 >= h->ops.size()) return;
                        const ecseg_op_desc& nx = h->ops[oi + 1].d;
                        const TView po = nx.op == ECSEG_OP_MAXPOOL ? at(nx.out, rebased) : TView{};
                        if (nx.op == ECSEG_OP_MAXPOOL && nx.mode == 0 /* max, not average */ && nx.in0 == d.out && nx.kh == 2 && nx.kw == 2 && nx.stride == 2 &&
                            h->fuse_pool && !softmax && po.h * 2 == out.h && po.w * 2 == out.w && po.c == out.c && po.cs % 4 == 0 &&
                            reinterpret_cast<uintptr_t>(po.p) % 16 == 0 &&
                            h->tensors[nx.out].buffer != h->tensors[d.in0].buffer && h->tensors[nx.out].buffer != h->tensors[d.out].buffer) {
                            p.pool = po;
                            ++oi;                              // the pooling op is done
                        }
                    };
                    // a 1x1 head (<= 4 classes) that is the only reader of this convolution's output is computed by the same output
                    // stage (conv_wino4: 64 channels, conv_wino16: 16 / 32); the feature tensor is then never written
                    auto fuse_following_head = [&](int channels) {
                        if (p.pool.p != nullptr || oi + 1 >= h->ops.size() || !h->fuse_head || out.c != channels || softmax) return;
                        const OpRt& hx = h->ops[oi + 1];
                        const ecseg_tensor_desc& td = h->tensors[d.out];
                        if (hx.d.op == ECSEG_OP_CONV && hx.path == PATH_HEAD && hx.head_w4 && hx.d.in0 == d.out && hx.d.act <= ECSEG_ACT_TANH &&
                            hx.d.act != ECSEG_ACT_LEAKY /* (the fused stage has the convolution's alpha, not the head's) */ &&
                            h->consumers[d.out] == 1 && d.out != h->output_tensor && td.c_stride == td.c && td.c_offset == 0 &&
                            // workgroups write head pixels while others still read the convolution's input halo
                            h->tensors[hx.d.out].buffer != h->tensors[d.in0].buffer &&
                            h->tensors[hx.d.out].buffer != td.buffer) {
                            p.head_w = hx.head_w4; p.head_b = hx.head_b4; p.head_out = at(hx.d.out, rebased);
                            p.head_k = p.head_out.c; p.head_act = hx.d.act; p.head_only = 1;
                            ++oi;                              // the head op is done
                        }
                    };
                    const bool wino4s = wino4 && h->use_winograd >= 3 && o.wt_wino4s && conv_wino4s_supported(p);
                    if (wino4) {
                        p.wt = wino4s ? reinterpret_cast<const float*>(o.wt_wino4s) : o.wt_wino4; p.coutp = out.c; p.w4_split = h->wino4_split;
                        if (crop && h->crop && o.crop_ok && (part || n % crop->n_pos == 0)) {
                            const CropLut* cl = get_crop_lut(crop, o.crop_code);
                            if (cl->len > 0 && out.h == cl->size && out.w == cl->size && conv_wino4_span_ok(p, crop->n_pos)) {
                                use_lut(cl);
                                computed = (double)cl->len / ((double)crop->n_pos * (out.h / 16) * (out.w / 16));
                            }
                        }
                        // a MaxPooling2D(2x2, stride 2) that follows directly is written by the same output stage
                        fuse_following_pool();
                        fuse_following_head(64);
                        e = wino4s ? launch_conv_wino4s(p, s) : (h->wino4_rowpass && conv_wino4r_supported(p)) ? launch_conv_wino4r(p, s) : launch_conv_wino4(p, s);
                    } else if (wino && h->wino16 && o.wt_wino16 && act_core_ok && (first ? conv_wino16_first_supported(p) : conv_wino16_supported(p))) {
                        w16 = true;
                        p.wt = o.wt_wino16;
                        if (crop && h->crop && o.crop_ok && (part || n % crop->n_pos == 0)) {
                            // cropped launch: only the 16 x 32 blocks some later stage reads
                            const CropLut* cl = get_crop_lut(crop, o.crop_code, 16, 32);
                            if (cl->len > 0 && out.h == cl->size && out.w == cl->size) {
                                use_lut(cl);
                                computed = (double)cl->len / ((double)crop->n_pos * (out.h / 16) * (out.w / 32));
                            }
                        }
                        fuse_following_pool();
                        fuse_following_head(out.c);
                        e = launch_conv_wino16(p, s);
                    } else if (wino) {
                        p.wt = o.wt_wino; p.coutp = o.coutp_wino;
                        p.wt_chunk_stride = wt_chunk_pitch(o.coutp_wino); p.wt_tap_stride = wt_tap_pitch(o.coutp_wino, o.cin_chunks);
                        p.resident = h->wino_resident;
                        if (out.c % 4 == 0) fuse_following_pool();
                        e = launch_conv_wino(p, s);
                    } else {
                        if (crop && h->crop && o.crop_ok && p.convt && (part || n % crop->n_pos == 0) && in.h == in.w) {
                            // cropped up-convolution: only the input tiles whose outputs somebody reads; of the two tile
                            // shapes (4 x 32, 8 x 16) the one that needs fewer tiles
                            const CropLut* a = get_crop_lut(crop, o.crop_code, 4, 32);
                            const CropLut* b = get_crop_lut(crop, o.crop_code, 8, 16);
                            const CropLut* cl = nullptr; int tw = 0;
                            if (a->len > 0 && a->size == in.h && in.w >= 32 && (b->len == 0 || b->size != in.h || a->len <= b->len)) { cl = a; tw = 32; }
                            else if (b->len > 0 && b->size == in.h) { cl = b; tw = 16; }
                            if (cl) {
                                use_lut(cl); p.force_tw = tw;
                                const int th = 128 / tw;
                                computed = (double)cl->len / ((double)crop->n_pos * ((in.h + th - 1) / th) * ((in.w + tw - 1) / tw));
                            }
                        }
                        if (o.subpixel && o.ph_wt[0] != nullptr) {
                            // phase by phase (see OpRt::ph_wt): tiles walk the input positions j of the outputs 2 j + c that exist
                            const int crop_t = p.crop_top, crop_l = p.crop_left;
                            p.tap_zero_mask = 0;
                            p.crop_top = 0; p.crop_left = 0;
                            p.convt_ext = ((out.h + 1) / 2 > in.h || (out.w + 1) / 2 > in.w) ? 1 : 0;
                            (void)crop_t; (void)crop_l;
                            for (int ph = 0; ph < 4 && e == hipSuccess; ++ph) {
                                p.wt = o.ph_wt[ph]; p.R = o.ph_R[ph]; p.S = o.ph_S[ph]; p.pad_top = o.ph_pt[ph]; p.pad_left = o.ph_pl[ph];
                                p.phase_a = ph >> 1; p.phase_b = ph & 1;
                                p.convt = 2;                   // one output phase per launch: N = coutp
                                p.wt_chunk_stride = wt_chunk_pitch(o.coutp);
                                p.wt_tap_stride = wt_tap_pitch(o.coutp, o.cin_chunks);
                                e = launch_conv_mfma(p, s);
                            }
                        } else if (h->use_winograd >= 3 && o.wt_split1 != nullptr && p.convt == 1 && convs_supported(p)) {
                            p.wt = reinterpret_cast<const float*>(o.wt_split1);
                            split1 = true;
                            e = launch_convs(p, s);
                        } else {
                            e = launch_conv_mfma(p, s);
                        }
                    }
                    if (first && !w16 && e == hipSuccess) e = hipErrorInvalidValue;     // (the eligibility test above and the launcher's disagree)
                    if (ev) {
                        (void)hipEventRecord(ev[1], s);
                        if (first) { h->prof_flops += first->flops * n; h->prof_exec_flops += first->flops * n * 12.0 / 9.0; }   // (9 taps padded to 12 on the MFMA)
                        h->prof_flops += o.flops * n;
                        // multiplies actually issued (sub-pixel transposed convolution: 4 taps x 4 phases per input pixel minus the all-zero blocks the kernel skips)
                        const double ex = o.flops * n * computed * (wino4 ? 0.25 : wino ? 16.0 / 36.0 : (o.subpixel && o.ph_wt[0] == nullptr) ? (16.0 - __builtin_popcount((unsigned)p.tap_zero_mask)) / (d.kh * d.kw) : 1.0);
                        h->prof_exec_flops += ex;
                        const bool res = wino && p.resident && p.coutp == 32 && p.cin_chunks <= 4;
                        // kind: bits 0-7 the kernel, bit 8: the following 2x2 max-pool was written by this launch, bit 9: the following 1x1 head was
                        h->prof_recs.push_back({(int)oi_first, (split1 ? 6 : wino4s ? 5 : wino4 ? 2 : w16 ? 4 : res ? 3 : wino ? 1 : 0) | (p.pool.p != nullptr ? 0x100 : 0) |
                                                (p.head_w != nullptr ? 0x200 : 0) | (first ? 0x400 : 0), o.flops * n + (first ? first->flops * n : 0.0),
                                                ex + (first ? first->flops * n * 12.0 / 9.0 : 0.0), 0.f});
                    }
                } else if (o.path == PATH_TAP) {
                    ConvParams p{};
                    p.in = in; p.out = out; p.wt = o.wt; p.bias = o.bias; p.n = n; p.act = act; p.alpha = d.alpha;
                    p.cin_chunks = o.cin_chunks; p.coutp = o.coutp; p.zero = h->zero_page;
                    p.R = d.kh; p.S = d.kw; p.pad_top = d.pad_top; p.pad_left = d.pad_left; p.stride = d.stride;
                    p.wt_chunk_stride = wt_chunk_pitch(o.coutp); p.wt_tap_stride = wt_tap_pitch(o.coutp, o.cin_chunks);
                    hipEvent_t* ev = h->profile_kernels ? prof_pair(h) : nullptr;
                    if (ev) (void)hipEventRecord(ev[0], s);
                    e = launch_conv_mfma_tap(p, d.dilation, s);
                    if (ev) {
                        (void)hipEventRecord(ev[1], s);
                        h->prof_flops += o.flops * n; h->prof_exec_flops += o.flops * n;
                        h->prof_recs.push_back({(int)oi, 0, o.flops * n, o.flops * n, 0.f});
                    }
                } else if (o.path == PATH_SMALL_CIN) {
                    e = launch_conv_small_cin(in, out, o.wt, o.bias, n, d.kh, d.kw, d.pad_top, d.pad_left, act, d.alpha, s);
                } else if (o.path == PATH_HEAD) {
                    e = launch_conv_head(in, out, o.wt, o.bias, n, d.act, d.alpha, s);
                } else if (d.op == ECSEG_OP_CONV) {
                    if (d.dilation > 1 || (d.mode & 0xffff))
                        e = launch_conv_generic_dil(in, out, o.wt, o.bias, n, d.kh, d.kw, d.stride, (d.mode & 0xff) ? (d.mode & 0xff) : d.stride, d.dilation > 1 ? d.dilation : 1,
                                                    ((d.mode >> 8) & 0xff) ? ((d.mode >> 8) & 0xff) : (d.dilation > 1 ? d.dilation : 1), d.pad_top, d.pad_left, act, d.alpha, s);
                    else e = launch_conv_generic(in, out, o.wt, o.bias, n, d.kh, d.kw, d.stride, d.pad_top, d.pad_left, act, d.alpha, s);
                } else {
                    e = launch_convt_generic(in, out, o.wt, o.bias, n, d.kh, d.kw, d.stride, d.pad_top, d.pad_left, act, d.alpha, s);
                }
                if (e == hipSuccess && softmax && o.path != PATH_HEAD) e = launch_softmax(out, out, n, s);
                break;
            }
            case ECSEG_OP_MAXPOOL:
                if (d.pad_top || d.pad_left || (out.h - 1) * d.stride + d.kh > in.h || (out.w - 1) * d.stride + d.kw > in.w)
                    e = launch_pool_pad(in, out, n, d.kh, d.kw, d.stride, d.pad_top, d.pad_left, d.mode, s);     // padding = 'same'
                else e = launch_maxpool(in, out, n, d.kh, d.kw, d.stride, d.mode, s);
                break;
            case ECSEG_OP_DWCONV: {
                const bool softmax = d.act == ECSEG_ACT_SOFTMAX;
                e = launch_dwconv(in, out, o.wt, o.bias, n, d.kh, d.kw, d.stride, d.dilation, d.pad_top, d.pad_left, d.mode,
                                  softmax ? ECSEG_ACT_LINEAR : d.act, d.alpha, s);
                if (e == hipSuccess && softmax) e = launch_softmax(out, out, n, s);
                break;
            }
            case ECSEG_OP_PRELU: e = launch_prelu(in, out, o.wt, n, d.mode, s); break;
            case ECSEG_OP_LAYERNORM: e = launch_layernorm(in, out, o.scale, o.shift, n, d.alpha, s); break;
            case ECSEG_OP_GLOBALPOOL: e = launch_global_pool(in, out, n, d.mode, s); break;
            case ECSEG_OP_UPSAMPLE: e = launch_upsample(in, out, n, d.stride, d.mode, s); break;
            case ECSEG_OP_AFFINE:
                if (d.act == ECSEG_ACT_SOFTMAX) {
                    e = launch_affine(in, out, o.scale, o.shift, n, ECSEG_ACT_LINEAR, d.alpha, s);
                    if (e == hipSuccess) e = launch_softmax(out, out, n, s);
                } else {
                    e = launch_affine(in, out, o.scale, o.shift, n, d.act, d.alpha, s);
                }
                break;
            case ECSEG_OP_ACT:
                if (d.act == ECSEG_ACT_SOFTMAX) e = launch_softmax(in, out, n, s);
                else e = launch_affine(in, out, nullptr, nullptr, n, d.act, d.alpha, s);
                break;
            case ECSEG_OP_ADD: {
                const TView b = at(d.in1);
                const bool same = in.h == out.h && in.w == out.w && in.c == out.c && b.h == out.h && b.w == out.w && b.c == out.c;
                if (d.mode == ECSEG_BIN_ADD && same && d.act != ECSEG_ACT_SOFTMAX) e = launch_add(in, b, out, n, d.act, d.alpha, s);
                else {
                    e = launch_binary(in, b, out, n, d.mode, d.act == ECSEG_ACT_SOFTMAX ? ECSEG_ACT_LINEAR : d.act, d.alpha, s);
                    if (e == hipSuccess && d.act == ECSEG_ACT_SOFTMAX) e = launch_softmax(out, out, n, s);
                }
                break;
            }
            case ECSEG_OP_COPY: e = launch_copy(in, out, n, d.pad_top, d.pad_left, s); break;
            default: return fail(h, ECSEG_E_INVALID, "unknown op in plan");
        }
        if (e != hipSuccess) return fail_hip(h, e, "plan kernel launch");
    }
    return ECSEG_OK;
}

// The whole plan on n_all patches whose input tensor has been written; with lanes, op by op for every lane in turn (the lanes'
// kernels are enqueued interleaved, so the streams start together)
int run_plan(ecseg_ctx* h, int n_all, StitchPlan* crop = nullptr, const std::vector<LaneSpec>* lanes = nullptr) {
    for (size_t oi = 0; oi < h->ops.size(); ++oi) {
        int rc;
        if (!lanes || lanes->empty()) {
            if ((rc = run_plan_op(h, oi, n_all, crop, nullptr))) return rc;
        } else {
            size_t last = oi;
            for (const LaneSpec& l : *lanes) {
                size_t o2 = oi;
                if (l.cnt > 0 && (rc = run_plan_op(h, o2, n_all, crop, &l))) return rc;
                if (l.cnt > 0) last = o2;
            }
            oi = last;
        }
    }
    return ECSEG_OK;
}

void prof_begin(ecseg_ctx* h) { h->prof_used = 0; h->prof_flops = 0.0; h->prof_exec_flops = 0.0; h->prof_recs.clear(); }
void prof_end(ecseg_ctx* h) {   // stream must be idle
    double ms = 0.0;
    for (size_t k = 0; k + 1 < h->prof_used; k += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, h->prof_events[k], h->prof_events[k + 1]) == hipSuccess) ms += t;
        if (k / 2 < h->prof_recs.size()) h->prof_recs[k / 2].ms = t;
    }
    h->last_conv_ms = ms; h->last_conv_launches = (long long)(h->prof_used / 2); h->last_conv_flops = h->prof_flops;
    h->last_conv_exec_flops = h->prof_exec_flops;
}

// ---- tiling / stitch geometry (reference src/image_tools.py:148-252), computed once per image size ----
std::vector<int> window_starts(int dim) {
    const int cropped = dim - 50, spw = 206;
    std::vector<int> s;
    for (int e = 0; e < cropped / spw; ++e) s.push_back(spw * e);
    if (cropped % spw) s.push_back(cropped - spw);
    return s;
}

int get_stitch(ecseg_ctx* h, int H, int W, StitchPlan** out) {
    auto key = std::make_pair(H, W);
    auto it = h->stitch.find(key);
    if (it != h->stitch.end()) { *out = &it->second; return ECSEG_OK; }
    if (H < 256 || W < 256) return fail(h, ECSEG_E_INVALID, "image smaller than one 256x256 window");
    if ((long long)H * W >= (1ll << 31)) return fail(h, ECSEG_E_INVALID, "image too large");
    const std::vector<int> Lh = window_starts(H), Lw = window_starts(W);
    std::vector<int32_t> pos;
    for (int w : Lw) for (int hh : Lh) { pos.push_back(hh); pos.push_back(w); }   // meshgrid order: columns outer
    const int n = (int)pos.size() / 2;
    if (n >= 32768) return fail(h, ECSEG_E_INVALID, "too many patches per image");
    const int h_l = Lh.back(), w_l = Lw.back();
    const int Hc = h_l + 256, Wc = w_l + 256;    // == H, W
    std::vector<int32_t> map((size_t)Hc * Wc, -1);
    auto put = [&](int i, int dr0, int dr1, int dc0, int dc1, int sr0, int sc0) {
        for (int r = dr0; r < dr1; ++r)
            for (int c = dc0; c < dc1; ++c)
                map[(size_t)r * Wc + c] = (i << 16) | ((sr0 + r - dr0) << 8) | (sc0 + c - dc0);
    };
    const int o = 25, lo = 25, hi = 231;
    for (int i = 0; i < n; ++i) {
        const int ph = pos[2 * i], pw = pos[2 * i + 1];
        if (ph == 0) {
            if (pw == 0) { put(i, 0, o, 0, o, 0, 0); put(i, lo, hi, 0, o, lo, 0); put(i, 0, o, lo, hi, 0, lo); }
            else { if (pw == w_l) put(i, 0, o, Wc - o, Wc, 0, hi); put(i, 0, o, pw + lo, pw + hi, 0, lo); }
        }
        if (pw == 0 && ph != 0) put(i, ph + lo, ph + hi, 0, o, lo, 0);
        if (ph == h_l) {
            if (pw == w_l) {
                put(i, Hc - o, Hc, Wc - o, Wc, hi, hi);
                put(i, h_l + lo, Hc - o, Wc - o, Wc, lo, hi);
                put(i, Hc - o, Hc, w_l + lo, Wc - o, hi, lo);
            } else {
                if (pw == 0) put(i, Hc - o, Hc, 0, o, hi, 0);
                put(i, Hc - o, Hc, pw + lo, pw + hi, hi, lo);
            }
        }
        if (pw == w_l && pw != h_l) put(i, ph + lo, ph + hi, Wc - o, Wc, lo, hi);   // sic: column start vs h_l (:242)
    }
    for (int i = 0; i < n; ++i) put(i, pos[2 * i] + lo, pos[2 * i] + hi, pos[2 * i + 1] + lo, pos[2 * i + 1] + hi, lo, lo);
    StitchPlan sp;
    sp.n_pos = n;
    sp.box.assign((size_t)n * 4, 0);
    for (int i = 0; i < n; ++i) { sp.box[4 * i] = 256; sp.box[4 * i + 1] = -1; sp.box[4 * i + 2] = 256; sp.box[4 * i + 3] = -1; }
    for (int32_t v : map) {
        if (v < 0) continue;
        const int i = v >> 16, y = (v >> 8) & 255, x = v & 255;
        sp.box[4 * i] = std::min(sp.box[4 * i], y); sp.box[4 * i + 1] = std::max(sp.box[4 * i + 1], y);
        sp.box[4 * i + 2] = std::min(sp.box[4 * i + 2], x); sp.box[4 * i + 3] = std::max(sp.box[4 * i + 3], x);
    }
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&sp.pos_dev), pos.size() * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&sp.map_dev), map.size() * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(sp.pos_dev, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(sp.map_dev, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    auto ins = h->stitch.emplace(key, sp);
    *out = &ins.first->second;
    return ECSEG_OK;
}

void drop_post_graphs(ecseg_ctx* h) {
    for (auto& g : h->post_graphs) (void)hipGraphExecDestroy(g.exec);
    h->post_graphs.clear();
}

// run_meta_inference through a cached HIP graph (stream capture of the same launches).  Any failure of the graph path
// falls back to plain launches - the results are the same kernels either way.
hipError_t post_run(ecseg_ctx* h, uint8_t* img, int n, int H, int W, int32_t* nec, hipStream_t s) {
    if (!h->post_graph) return run_meta_inference(h->ws, img, n, H, W, nec, s);
    for (auto& g : h->post_graphs)
        if (g.img == img && g.nec == nec && g.n == n && g.H == H && g.W == W && g.s == s) {
            g.stamp = ++h->post_graph_clock;
            return hipGraphLaunch(g.exec, s);
        }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return run_meta_inference(h->ws, img, n, H, W, nec, s);
    }
    const hipError_t e1 = run_meta_inference(h->ws, img, n, H, W, nec, s);
    const hipError_t e2 = hipStreamEndCapture(s, &graph);
    if (e1 != hipSuccess || e2 != hipSuccess || graph == nullptr || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        h->post_graph = 0;                                   // do not try again on this handle
        return run_meta_inference(h->ws, img, n, H, W, nec, s);
    }
    (void)hipGraphDestroy(graph);
    if (h->post_graphs.size() >= 8) {                        // evict the least recently used entry
        size_t k = 0;
        for (size_t i = 1; i < h->post_graphs.size(); ++i) if (h->post_graphs[i].stamp < h->post_graphs[k].stamp) k = i;
        (void)hipGraphExecDestroy(h->post_graphs[k].exec);
        h->post_graphs.erase(h->post_graphs.begin() + (long)k);
    }
    h->post_graphs.push_back({img, nec, n, H, W, s, exec, ++h->post_graph_clock});
    return hipGraphLaunch(exec, s);
}

int ensure_post(ecseg_ctx* h, int n_img, size_t px) {
    PostWorkspace& w = h->ws;
    if (n_img <= w.cap_img && px <= w.cap_px && w.L) return ECSEG_OK;
    drop_post_graphs(h);                                     // the captured launches hold the old workspace pointers
    const int ni = std::max(n_img, w.cap_img);
    const size_t np = std::max(px, w.cap_px);
    void* ptrs[] = {w.L, w.area, w.sumy, w.sumx, w.flag, w.tmpA, w.tmpB, w.list, w.g, w.tile_any, w.own_bits, w.binned, w.binstart};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    w = PostWorkspace{};
    const size_t tot = (size_t)ni * np;
    // Root lists of the nucleus-in-metaphase test: run_meta_inference uses px/4 + (H+W)/2 + 4 entries per image
    // (>= ceil(H/2)*ceil(W/2), the most 8-connected components an image can hold); (H+W)/2 <= px/2 + 1, and every
    // entry costs 4 B (nucleus root) + 16 B (chromosome centroid).
    const size_t list_cap = np / 4 + np / 2 + 8;
    const size_t list_bytes = (size_t)ni * list_cap * 20 + 256;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
    A(reinterpret_cast<void**>(&w.L), tot * 4);
    A(reinterpret_cast<void**>(&w.area), tot * 4);
    A(reinterpret_cast<void**>(&w.sumy), tot * 8);
    A(reinterpret_cast<void**>(&w.sumx), tot * 8);
    A(reinterpret_cast<void**>(&w.flag), tot * 4);
    A(reinterpret_cast<void**>(&w.tmpA), tot);
    A(reinterpret_cast<void**>(&w.tmpB), tot);
    A(reinterpret_cast<void**>(&w.list), list_bytes);
    A(reinterpret_cast<void**>(&w.g), (size_t)G_SLOTS * ni * G_STRIDE * G_SHARDS * 4);
    A(reinterpret_cast<void**>(&w.tile_any), (size_t)ni * (np / 16 + 2));
    // owner bits: 256 B per 64 x 32 tile; ceil(W/64) ceil(H/32) <= px/2048 + W/64 + H/32 + 1 <= px/31 + 3 tiles for any H x W = px
    A(reinterpret_cast<void**>(&w.own_bits), (size_t)ni * (np / 31 + 4) * 256);
    const size_t binned_cap = std::min(list_cap, (size_t)1 << 20);
    A(reinterpret_cast<void**>(&w.binned), (size_t)ni * 2 * binned_cap * sizeof(double));
    A(reinterpret_cast<void**>(&w.binstart), (size_t)ni * 2 * (NUCLEUS_BIN_EXTENT + 2) * sizeof(int32_t));   // (W/64 + 1)(H/32 + 1) <= px/16 + 1 tiles per image
    if (e != hipSuccess) return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc(post workspace): ") + hipGetErrorString(e));
    w.cap_img = ni; w.cap_px = np; w.binned_cap = binned_cap;
    h->ws_list_bytes = list_bytes;
    return ECSEG_OK;
}

int check_model(ecseg_ctx* h) {
    if (!h) return ECSEG_E_INVALID;
    if (!h->has_model) return fail(h, ECSEG_E_NOMODEL, "no model loaded (call ecseg_model_load first)");
    return ECSEG_OK;
}

bool debug_calls() { static const bool on = getenv("ECSEG_DEBUG_CALLS") != nullptr; return on; }

double dbg_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

float stage_elapsed(hipEvent_t a, hipEvent_t b) {
    float t = 0.f;
    return hipEventElapsedTime(&t, a, b) == hipSuccess ? t : 0.f;
}

// Wait for everything enqueued on a stream (the long waits: a whole launch group).  "blocking_wait" 1 (default): record an
// event created with hipEventBlockingSync and sleep on it - beside the device's blocking-sync flag (ecseg_create) this also
// keeps the runtime's helper thread off the CPU (0.31 -> 0.13 cores busy per waiting call).
hipError_t wait_stream(ecseg_ctx* h, hipStream_t s) {
    if (!h->blocking_wait) return hipStreamSynchronize(s);
    hipError_t e = hipSuccess;
    if (!h->ev_block && (e = hipEventCreateWithFlags(&h->ev_block, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(h->ev_block, s)) != hipSuccess) return e;
    return hipEventSynchronize(h->ev_block);
}

// Device-resident pipeline: gray (n_img, H, W) -> raw labels, post labels, counts.  All pointers are device pointers.
// probs_host (optional): the stitched float32 probabilities of every image, copied out group by group.
int segment_dev(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W, uint8_t* raw, uint8_t* post, int32_t* n_ec,
                float* probs_host = nullptr) {
    const double tq00 = dbg_now();
    int rc = check_model(h);
    if (rc) return rc;
    if ((rc = ensure(h, h->d_tie, h->d_tie_cap, (size_t)n_img))) return rc;
    const ecseg_tensor_desc& ti = h->tensors[h->input_tensor];
    const ecseg_tensor_desc& to = h->tensors[h->output_tensor];
    if (ti.h != 256 || ti.w != 256 || ti.c != 1 || ti.c_stride != 1)
        return fail(h, ECSEG_E_INVALID, "segment: model input must be (256, 256, 1)");
    if (to.h != 256 || to.w != 256 || to.c != 4)
        return fail(h, ECSEG_E_INVALID, "segment: model output must be (256, 256, 4)");
    StitchPlan* sp = nullptr;
    if ((rc = get_stitch(h, H, W, &sp))) return rc;
    const size_t px = (size_t)H * W;
    hipStream_t s = h->stream, s2 = h->overlap_post ? h->stream2 : h->stream;
    // images per U-Net launch: images_per_group is calibrated for 35-window images (1040 x 1392); larger images have more
    // windows each, so the group shrinks to keep the activation memory (~82 MB per window for a base-64 U-Net) bounded
    const int wpg = windows_per_group(h);
    const int grp = std::max(1, std::min(wpg / 35, std::max(1, wpg / sp->n_pos)));
    if ((rc = ensure_patches(h, std::min(grp, n_img) * sp->n_pos))) return rc;
    if ((rc = ensure_post(h, std::min(n_img, grp), px))) return rc;
    if ((rc = ensure(h, h->d_tie_sh, h->d_tie_sh_cap, (size_t)std::min(grp, n_img) * G_SHARDS * G_STRIDE))) return rc;
    if (probs_host && (rc = ensure(h, h->d_sprobs, h->d_sprobs_cap, (size_t)std::min(grp, n_img) * px * 4))) return rc;
    for (float& v : h->stage_ms) v = 0.f;
    prof_begin(h);
    // Per group: tile -> U-Net -> stitch/argmax on the main stream; the group's clean-up + count then runs on the second
    // stream while the main stream already computes the next group's U-Net (MFMA-bound convs and latency-bound
    // integer kernels co-exist well).  6 events per group: tile start, unet start, tail start, tail end, post start/end.
    // events come from a pool owned by the handle (freed in ecseg_destroy): nothing to leak on an early return, and no
    // event creation inside the timed loop
    const size_t ngrp = ((size_t)n_img + grp - 1) / grp;
    while (h->grp_events.size() < 6 * ngrp) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        h->grp_events.push_back(e);
    }
    const std::vector<hipEvent_t>& evs = h->grp_events;
    size_t used = 0;
    const double tq0 = dbg_now();
    for (int i0 = 0; i0 < n_img; i0 += grp) {
        const int ni = std::min(grp, n_img - i0);
        const hipEvent_t* e6 = &evs[used];
        used += 6;
        HIP_TRY(h, hipEventRecord(e6[0], s));
        HIP_TRY(h, launch_tile_patches(gray + (size_t)i0 * px, ni, H, W, sp->pos_dev, sp->n_pos,
                                       view_of(h, h->input_tensor).p, s));
        HIP_TRY(h, hipEventRecord(e6[1], s));
        {
            // small batches: 2+ window lanes on their own streams (see run_plan); lane 0 stays on the main stream
            const int nw = ni * sp->n_pos;
            int lanes = h->unet_lanes > 0 ? h->unet_lanes : (nw <= h->lane_auto_windows ? 2 : 1);
            if (h->profile_kernels || !h->lanes_ok) lanes = 1;     // (per-launch events are taken on the main stream)
            if (ni > 1) lanes = std::min(lanes, ni);         // whole images per lane
            lanes = std::max(1, std::min(lanes, std::min(nw, 8)));
            while ((int)h->lane_streams.size() < lanes - 1) {
                hipStream_t ls;
                HIP_TRY(h, hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
                h->lane_streams.push_back(ls);
            }
            while ((int)h->lane_events.size() < lanes - 1) {     // (its own loop: a failed creation leaves the two lists consistent)
                hipEvent_t le;
                HIP_TRY(h, hipEventCreateWithFlags(&le, hipEventDisableTiming));
                h->lane_events.push_back(le);
            }
            if (lanes == 1) {
                if ((rc = run_plan(h, nw, sp))) return rc;
            } else {
                const int unit = ni > 1 ? sp->n_pos : 1, units = nw / unit;
                std::vector<LaneSpec> specs;
                int u0 = 0;
                for (int l = 0; l < lanes; ++l) {
                    const int u1 = (int)((long long)units * (l + 1) / lanes);
                    hipStream_t ls = l == 0 ? s : h->lane_streams[l - 1];
                    if (l > 0) HIP_TRY(h, hipStreamWaitEvent(ls, e6[1], 0));
                    specs.push_back({u0 * unit, (u1 - u0) * unit, ls});
                    u0 = u1;
                }
                if ((rc = run_plan(h, nw, sp, &specs))) return rc;
                for (int l = 1; l < lanes; ++l) {
                    HIP_TRY(h, hipEventRecord(h->lane_events[l - 1], h->lane_streams[l - 1]));
                    HIP_TRY(h, hipStreamWaitEvent(s, h->lane_events[l - 1], 0));
                }
            }
        }
        HIP_TRY(h, hipEventRecord(e6[2], s));
        const TView pv = view_of(h, h->output_tensor);
        HIP_TRY(h, launch_stitch_argmax(pv.p, pv.cs, sp->map_dev, ni, sp->n_pos, H, W, raw + (size_t)i0 * px, s, h->d_tie + i0, h->d_tie_sh));
        HIP_TRY(h, hipEventRecord(e6[3], s));
        if (probs_host) {                                  // (diagnostic output: outside the stage timers)
            HIP_TRY(h, launch_stitch_probs(pv.p, pv.cs, sp->map_dev, ni, sp->n_pos, H, W, h->d_sprobs, s));
            HIP_TRY(h, hipMemcpyAsync(probs_host + (size_t)i0 * px * 4, h->d_sprobs, (size_t)ni * px * 4 * sizeof(float), hipMemcpyDeviceToHost, s));
        }
        if (s2 != s) HIP_TRY(h, hipStreamWaitEvent(s2, e6[3], 0));
        HIP_TRY(h, hipEventRecord(e6[4], s2));
        if (post != raw)
            HIP_TRY(h, hipMemcpyAsync(post + (size_t)i0 * px, raw + (size_t)i0 * px, px * ni, hipMemcpyDeviceToDevice, s2));
        HIP_TRY(h, post_run(h, post + (size_t)i0 * px, ni, H, W, n_ec ? n_ec + i0 : nullptr, s2));
        HIP_TRY(h, hipEventRecord(e6[5], s2));
    }
    const double tq1 = dbg_now();
    HIP_TRY(h, wait_stream(h, s));
    if (s2 != s) HIP_TRY(h, wait_stream(h, s2));
    const double tq2 = dbg_now();
    // ECSEG_DEBUG_CALLS: host-side timeline of the call on stderr (a `make metaseg` whose device calls take longer than their
    // kernels: is the host late with the launches, or is the wait long - e.g. a throttled CPU quota - ?)
    if (debug_calls()) fprintf(stderr, "[segment_dev] setup %.2f enqueue %.2f wait %.2f ms\n", tq0 - tq00, tq1 - tq0, tq2 - tq1);
    for (size_t k = 0; k + 5 < used; k += 6) {
        h->stage_ms[ECSEG_T_TILE] += stage_elapsed(evs[k], evs[k + 1]);
        h->stage_ms[ECSEG_T_UNET] += stage_elapsed(evs[k + 1], evs[k + 2]);
        h->stage_ms[ECSEG_T_TAIL] += stage_elapsed(evs[k + 2], evs[k + 3]);
        h->stage_ms[ECSEG_T_POST] += stage_elapsed(evs[k + 4], evs[k + 5]);
    }
    prof_end(h);
    return ECSEG_OK;
}

}  // namespace

// =====================================================================================================================
extern "C" {

int ecseg_abi_version(void) { return ECSEG_ABI_VERSION; }

int ecseg_create(ecseg_ctx** out, int device_id) {
    if (!out) return fail(nullptr, ECSEG_E_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, ECSEG_E_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, ECSEG_E_INVALID, "device_id out of range");
    if ((e = hipSetDevice(device_id)) != hipSuccess) return fail_hip(nullptr, e, "hipSetDevice");
    // Waiting host threads sleep instead of spinning: with this runtime's default (hipDeviceScheduleAuto = spin when the host has
    // more cores than GPUs) a thread inside a segment call burns a whole core for the length of the call - measured 1.35 cores
    // busy per waiting call, 0.13 with this flag and the blocking event of wait_stream, at the same wall time
    // (tools/experiments/wait_cpu.py).  The flag belongs to the DEVICE, i.e. to every HIP user of it in this process (torch tensors
    // of an embedding application included): INTEGRATION.md says so next to the ABI notes, and ECSEG_SPIN_WAIT=1 leaves the runtime's
    // default alone.  Round 6 tried to make it opt-in and took that back: under the default scheduling mode hipFree
    // hung for ever in ecseg_destroy - device idle, hipDeviceSynchronize returning hipSuccess - once a process had created and
    // closed several handles (gpurun_out/r06_gputest_i.log: tests/test_gpu_unet.py then tests/test_gpu_configs.py; never with the flag).
    if (!getenv("ECSEG_SPIN_WAIT") || atoi(getenv("ECSEG_SPIN_WAIT")) == 0) {
        if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();    // (not fatal: the default stays)
    }
    ecseg_ctx* h = new ecseg_ctx();
    h->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        snprintf(h->devname, sizeof(h->devname), "%s (%s)", prop.name, prop.gcnArchName);
    }
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete h;
        return fail_hip(nullptr, e, "hipStreamCreate");
    }
    if ((e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking)) != hipSuccess) {
        (void)hipStreamDestroy(h->stream);
        delete h;
        return fail_hip(nullptr, e, "hipStreamCreate");
    }
    for (auto& ev : h->ev) (void)hipEventCreate(&ev);
    if (hipMalloc(reinterpret_cast<void**>(&h->zero_page), 1024) == hipSuccess) (void)hipMemset(h->zero_page, 0, 1024);
    else h->zero_page = nullptr;
    *out = h;
    return ECSEG_OK;
}

void ecseg_destroy(ecseg_ctx* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->stream2);
    free_model(h);
    drop_post_graphs(h);
    for (auto& kv : h->stitch) {
        (void)hipFree(kv.second.pos_dev); (void)hipFree(kv.second.map_dev);
        for (auto& lk : kv.second.luts) if (lk.second.dev) (void)hipFree(lk.second.dev);
        for (auto& bk : kv.second.boxes) if (bk.second) (void)hipFree(bk.second);
    }
    if (h->zero_page) (void)hipFree(h->zero_page);
    void* ptrs[] = {h->d_tie, h->d_tie_sh, h->d_sprobs, h->d_gray, h->d_raw, h->d_post, h->d_aux8, h->d_u8in, h->d_i32, h->d_i64, h->d_probs_in, h->d_hist,
                    h->ws.L, h->ws.area, h->ws.sumy, h->ws.sumx, h->ws.flag, h->ws.tmpA, h->ws.tmpB, h->ws.list, h->ws.g, h->ws.tile_any, h->ws.own_bits, h->ws.binned, h->ws.binstart};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->ev_block) (void)hipEventDestroy(h->ev_block);
    if (h->stream_in) { (void)hipStreamSynchronize(h->stream_in); (void)hipStreamDestroy(h->stream_in); }
    if (h->ev_pre) (void)hipEventDestroy(h->ev_pre);
    if (h->d_pre) (void)hipFree(h->d_pre);
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->grp_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->lane_events) (void)hipEventDestroy(e);
    for (hipStream_t ls : h->lane_streams) { (void)hipStreamSynchronize(ls); (void)hipStreamDestroy(ls); }
    (void)hipStreamDestroy(h->stream2);
    (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* ecseg_last_error(ecseg_ctx* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// Images named or sent ahead by ecseg_prefetch_input belong to the very next ecseg_meta_segment call: any other call on the handle
// in between withdraws them (the hit is keyed by host pointer + size, and a registration that outlived its call could
// meet a recycled page-locked buffer of the same shape holding OTHER pixels).
static inline void drop_sent_ahead(ecseg_ctx* h) {
    h->next_host = nullptr; h->next_bytes = 0;
    h->pre_host = nullptr; h->pre_bytes = 0;
}

int ecseg_device_name(ecseg_ctx* h, char* buf, int buflen) {
    if (!h || !buf || buflen <= 0) return ECSEG_E_INVALID;
    snprintf(buf, buflen, "%s", h->devname);
    return ECSEG_OK;
}

void* ecseg_stream(ecseg_ctx* h) { return h ? (void*)h->stream : nullptr; }

int ecseg_set_images_per_group(ecseg_ctx* h, int n) {
    if (!h || n < 0) return ECSEG_E_INVALID;               // 0: automatic (windows_per_group)
    h->images_per_group = n;
    return ECSEG_OK;
}

// "winograd" = 3: every layer that has an F(4x4) filter image and whole 64-channel output blocks gets the bf16x3 stage image of
// conv_wino4s_kernel, written by a device kernel from the fp32 image (U rounded to float32 as the fp32 kernel uses it, then split
// EXACTLY into three bf16 pieces).  Done when the option is set or a model is loaded under it - never inside a forward pass.
static int ensure_split_images(ecseg_ctx* h) {
    if (h->use_winograd < 3) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    bool any = false;
    for (OpRt& o : h->ops) {
        if (!o.wt_wino4 || o.wt_wino4s || o.w4_cout % 64 != 0) continue;
        void* d = nullptr;
        const hipError_t e = hipMalloc(&d, wino4s_image_bytes(o.w4_cin, o.w4_cout));
        if (e != hipSuccess) return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc(split filter image): ") + hipGetErrorString(e));
        h->dev_allocs.push_back(reinterpret_cast<float*>(d));
        HIP_TRY(h, launch_wino4s_filter(o.wt_wino4, d, o.w4_cin, o.w4_cout, h->stream));
        o.wt_wino4s = d;
        any = true;
    }
    for (OpRt& o : h->ops) {
        if (!o.s1_np || o.wt_split1 || !o.wt) continue;
        void* d = nullptr;
        const hipError_t e = hipMalloc(&d, convs_image_bytes(o.s1_cin, o.s1_np));
        if (e != hipSuccess) return fail(h, ECSEG_E_NOMEM, std::string("hipMalloc(split filter image): ") + hipGetErrorString(e));
        h->dev_allocs.push_back(reinterpret_cast<float*>(d));
        HIP_TRY(h, launch_convs_filter(o.wt, d, o.s1_cin, o.s1_np, h->stream));
        o.wt_split1 = d;
        any = true;
    }
    if (any) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ECSEG_OK;
}

int ecseg_set_option(ecseg_ctx* h, const char* key, int value) {
    if (!h || !key) return ECSEG_E_INVALID;
    const std::string k(key);
    if (k == "overlap_post") h->overlap_post = value != 0;
    else if (k == "blocking_wait") h->blocking_wait = value != 0;
    else if (k == "fuse_pool") h->fuse_pool = value != 0;
    else if (k == "fuse_head") h->fuse_head = value != 0;
    else if (k == "wino_resident") h->wino_resident = value != 0;
    else if (k == "wino16") h->wino16 = value != 0;
    else if (k == "fuse_first") h->fuse_first = value != 0;
    else if (k == "wino4_split") h->wino4_split = value != 0;
    else if (k == "wino4_rowpass") h->wino4_rowpass = value != 0;
    else if (k == "crop") h->crop = value != 0;
    else if (k == "winograd") {                            // 0 direct, 1 F(2x2), 2 F(4x4), 3 F(4x4) on the bf16 pipe with 3-way split operands
        h->use_winograd = value < 0 ? 0 : value > 3 ? 3 : (int)value;
        return ensure_split_images(h);
    }
    else if (k == "post_chunk" && value >= 1) h->post_chunk = value;
    else if (k == "post_graph") { h->post_graph = value != 0; if (!h->post_graph) drop_post_graphs(h); }
    else if (k == "images_per_group" && value >= 0) h->images_per_group = value;     // 0: automatic
    else if (k == "crop_mask") h->crop_mask = value != 0;
    else if (k == "unet_lanes" && value >= 0 && value <= 8) h->unet_lanes = value;   // 0: automatic
    else if (k == "lane_auto_windows" && value >= 0) h->lane_auto_windows = value;
    else return fail(h, ECSEG_E_INVALID, "unknown option or bad value: " + k);
    return ECSEG_OK;
}

int ecseg_model_load(ecseg_ctx* h, const ecseg_tensor_desc* tensors, int n_tensors, int n_buffers, const ecseg_op_desc* ops,
                     int n_ops, const float* const* weights, const int64_t* weight_len, int n_weights, int input_tensor,
                     int output_tensor) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (!tensors || !ops || n_tensors <= 0 || n_ops <= 0 || n_buffers <= 0) return fail(h, ECSEG_E_INVALID, "empty plan");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_model(h);
    h->tensors.assign(tensors, tensors + n_tensors);
    h->n_buffers = n_buffers;
    h->buf_floats.assign(n_buffers, 0);
    for (int t = 0; t < n_tensors; ++t) {
        const ecseg_tensor_desc& d = tensors[t];
        if (d.buffer < 0 || d.buffer >= n_buffers || d.h <= 0 || d.w <= 0 || d.c <= 0 || d.c_offset < 0 ||
            d.c_stride < d.c_offset + d.c)
            return fail(h, ECSEG_E_INVALID, "bad tensor descriptor " + std::to_string(t));
        h->buf_floats[d.buffer] = std::max(h->buf_floats[d.buffer], (size_t)d.h * d.w * d.c_stride);
    }
    if (input_tensor < 0 || input_tensor >= n_tensors || output_tensor < 0 || output_tensor >= n_tensors)
        return fail(h, ECSEG_E_INVALID, "bad input/output tensor index");
    h->input_tensor = input_tensor; h->output_tensor = output_tensor;
    h->flops_per_patch = 0.0; h->mfma_flops_per_patch = 0.0;

    auto W = [&](int idx, int64_t expect, const char* what, const float** out) -> int {
        *out = nullptr;
        if (idx < 0) return ECSEG_OK;
        if (idx >= n_weights || !weights || !weights[idx]) return fail(h, ECSEG_E_INVALID, std::string("missing weight for ") + what);
        if (weight_len[idx] != expect)
            return fail(h, ECSEG_E_INVALID, std::string("weight size mismatch for ") + what + ": got " +
                                                std::to_string(weight_len[idx]) + ", expected " + std::to_string(expect));
        *out = weights[idx];
        return ECSEG_OK;
    };

    for (int k = 0; k < n_ops; ++k) {
        OpRt o;
        o.d = ops[k];
        const ecseg_op_desc& d = o.d;
        if (d.in0 < 0 || d.in0 >= n_tensors || d.out < 0 || d.out >= n_tensors || (d.op == ECSEG_OP_ADD && (d.in1 < 0 || d.in1 >= n_tensors)))
            return fail(h, ECSEG_E_INVALID, "bad tensor index in op " + std::to_string(k));
        const ecseg_tensor_desc& ti = tensors[d.in0];
        const ecseg_tensor_desc& to = tensors[d.out];
        o.path = PATH_OTHER;
        int rc;
        if (d.op == ECSEG_OP_CONV || d.op == ECSEG_OP_CONVT) {
            if (d.kh <= 0 || d.kw <= 0 || d.stride <= 0) return fail(h, ECSEG_E_INVALID, "bad conv geometry in op " + std::to_string(k));
            const int cin = ti.c, cout = to.c;
            const float *kw = nullptr, *kb = nullptr;
            if ((rc = W(d.w0, (int64_t)d.kh * d.kw * cin * cout, "conv kernel", &kw))) return rc;
            if (!kw) return fail(h, ECSEG_E_INVALID, "conv without kernel in op " + std::to_string(k));
            if ((rc = W(d.w1, cout, "conv bias", &kb))) return rc;
            if (kb) { if ((rc = upload(h, std::vector<float>(kb, kb + cout), &o.bias))) return rc; }
            const bool in_al = (ti.c_stride % 4 == 0) && (ti.c_offset % 4 == 0) && (cin % 4 == 0);
            const bool out_al = (to.c_stride % 4 == 0) && (to.c_offset % 4 == 0);
            if (d.op == ECSEG_OP_CONV) {
                const int dil = d.dilation > 1 ? d.dilation : 1;
                // anisotropic strides / dilation rates (round 6): CONV's `mode` carries the HORIZONTAL stride (bits 0-7) and dilation rate (bits
                // 8-15) where they differ from the vertical ones in `stride` / `dilation` (0: the same) - such layers take the scalar kernel
                const int sx = (d.mode & 0xff) ? (d.mode & 0xff) : d.stride, dx = ((d.mode >> 8) & 0xff) ? ((d.mode >> 8) & 0xff) : dil;
                if ((to.h - 1) * d.stride + 1 > ti.h + (d.kh - 1) * dil || (to.w - 1) * sx + 1 > ti.w + (d.kw - 1) * dx)
                    return fail(h, ECSEG_E_INVALID, "conv output larger than its input in op " + std::to_string(k));
                o.flops = 2.0 * d.kh * d.kw * cin * cout * (double)to.h * to.w;
                if (sx != d.stride || dx != dil) {
                    o.path = PATH_GENERIC;
                    if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                    h->flops_per_patch += o.flops;
                    h->ops.push_back(o);
                    continue;
                }
                const bool taps_ok = d.kh == d.kw && (d.kh == 1 || d.kh == 2 || d.kh == 3);
                // the tap-by-tap MFMA kernel takes whatever the halo-staged kernels do not: dilated taps, taps other than 1x1 / 2x2 /
                // 3x3 (5x5, 7x7, 1x3 ...), strides above 2
                auto tap_path = [&]() -> int {
                    o.path = PATH_TAP;
                    const int bn = conv_mfma_ntile(cout);
                    o.coutp = (cout + bn - 1) / bn * bn;
                    o.cin_chunks = (cin + 7) / 8;
                    int rc2 = upload(h, relayout_conv(kw, d.kh, d.kw, cin, cout, o.cin_chunks, o.coutp), &o.wt);
                    if (!rc2) h->mfma_flops_per_patch += o.flops;
                    return rc2;
                };
                const bool tap_ok = in_al && cin >= 8 && cout >= 8;
                if (dil > 1 && !(d.kh == 1 && d.kw == 1)) {
                    if (tap_ok) { if ((rc = tap_path())) return rc; }
                    else {
                        o.path = PATH_GENERIC;
                        if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                    }
                } else if (d.stride != 1) {
                    // strided convolutions (classifier stems, down-sampling convolutions): the direct MFMA kernel gathers a
                    // strided halo (stride 2, 1x1 / 2x2 / 3x3 taps); anything else takes the generic kernel
                    if (d.stride == 2 && taps_ok && in_al && cin >= 8) {
                        o.path = PATH_MFMA;
                        const int bn = conv_mfma_ntile(cout);
                        o.coutp = (cout + bn - 1) / bn * bn;
                        o.cin_chunks = (cin + 7) / 8;
                        if ((rc = upload(h, relayout_conv(kw, d.kh, d.kw, cin, cout, o.cin_chunks, o.coutp), &o.wt))) return rc;
                        h->mfma_flops_per_patch += o.flops;
                    } else if (tap_ok) {
                        if ((rc = tap_path())) return rc;
                    } else {
                        o.path = PATH_GENERIC;
                        if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                    }
                } else if (cin <= 4 && cout % 4 == 0 && out_al) {
                    o.path = PATH_SMALL_CIN;
                    if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                } else if (d.kh == 1 && d.kw == 1 && cout <= 8 && in_al) {
                    o.path = PATH_HEAD;
                    if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)cin * cout), &o.wt))) return rc;
                    if (cout <= 4) {
                        std::vector<float> w4((size_t)cin * 4, 0.f), b4(4, 0.f);
                        for (int ci = 0; ci < cin; ++ci)
                            for (int co = 0; co < cout; ++co) w4[(size_t)ci * 4 + co] = kw[(size_t)ci * cout + co];
                        for (int co = 0; co < cout && kb; ++co) b4[co] = kb[co];
                        if ((rc = upload(h, w4, &o.head_w4))) return rc;
                        if ((rc = upload(h, b4, &o.head_b4))) return rc;
                    }
                } else if (taps_ok && in_al && cin >= 8 && (cout >= 16 || (d.kh >= 2 && cin >= 16))) {
                    // (a 2x2 / 3x3 convolution to a FEW channels - NuSeT's 3x3 'final' layer, src/model_layers/models.py:134 - still
                    // belongs on the matrix cores: a mostly empty 32-column tile beats the scalar kernel by an order of magnitude)
                    o.path = PATH_MFMA;
                    const int bn = conv_mfma_ntile(cout);
                    o.coutp = (cout + bn - 1) / bn * bn;
                    o.cin_chunks = (cin + 7) / 8;
                    if ((rc = upload(h, relayout_conv(kw, d.kh, d.kw, cin, cout, o.cin_chunks, o.coutp), &o.wt))) return rc;
                    h->mfma_flops_per_patch += o.flops;
                    if (d.kh == 3 && d.pad_top == 1 && d.pad_left == 1 && to.h == ti.h && to.w == ti.w && cout >= 16 && cout % 4 == 0 && out_al) {
                        const int bnw = conv_wino_ntile(cout);
                        o.coutp_wino = (cout + bnw - 1) / bnw * bnw;
                        const std::vector<float> u = winograd_filter(kw, cin, cout);
                        if ((rc = upload(h, relayout_conv(u.data(), 4, 4, cin, cout, o.cin_chunks, o.coutp_wino), &o.wt_wino))) return rc;
                        if ((cin == 16 || cin == 32) && (cout == 16 || cout == 32) && to.h >= 16 && to.w >= 32)
                            if ((rc = upload(h, relayout_wino16(u, cin, cout), &o.wt_wino16))) return rc;
                        // F(4x4): a lone 32-channel block wastes its second channel-half waves on zeros; measured on
                        // MI355X (profiles/r02_kernel_map.json) that still beats F(2x2) once the K loop is long enough
                        if (cin % 4 == 0 && cin >= 8 && cout % 32 == 0 && (cout != 32 || cin >= 64) && to.h % 16 == 0 && to.w % 16 == 0) {
                            if ((rc = upload(h, winograd4_filter(kw, cin, cout), &o.wt_wino4))) return rc;
                            o.w4_cin = cin; o.w4_cout = cout;
                        }
                    }
                } else if (tap_ok && !taps_ok) {
                    if ((rc = tap_path())) return rc;
                } else {
                    o.path = PATH_GENERIC;
                    if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                }
            } else {
                o.flops = 2.0 * d.kh * d.kw * cin * cout * (double)ti.h * ti.w;
                if (d.kh == d.kw && d.kh == d.stride && in_al && cin >= 8 && cout >= 16 && d.pad_top == 0 && d.pad_left == 0) {
                    o.path = PATH_MFMA;
                    const int bn = cout <= 16 && d.kh == 2 ? 16 : conv_mfma_ntile(cout);   // 2x2, <= 16 channels: all four phases in one 64-column tile
                    o.coutp = (cout + bn - 1) / bn * bn;
                    o.cin_chunks = (cin + 7) / 8;
                    if ((rc = upload(h, relayout_convt(kw, d.kh, cin, cout, o.cin_chunks, o.coutp), &o.wt))) return rc;
                    h->mfma_flops_per_patch += o.flops;
                    if (d.kh == 2 && o.coutp % 32 == 0 && cin >= 16 && cin % 4 == 0) { o.s1_cin = cin; o.s1_np = d.kh * d.kh * o.coutp; }
                } else if (d.kh == d.kw && (d.kh == 3 || d.kh == 4) && d.stride == 2 && in_al && cin >= 8 && d.pad_top >= 0 && d.pad_left >= 0) {
                    // k x k / stride 2, k != stride (a common Keras up-sampler; NuSeT's U-Net: src/model_layers/models.py:78-80):
                    // four sub-pixel convolutions as ONE 2x2-tap convolution over the input with N = 4 x Cout
                    o.path = PATH_MFMA;
                    o.subpixel = 1;
                    const int bn = cout <= 16 ? 16 : conv_mfma_ntile(cout);
                    o.coutp = (cout + bn - 1) / bn * bn;
                    o.cin_chunks = (cin + 7) / 8;
                    if ((rc = upload(h, relayout_convt_subpixel(kw, d.kh, cin, cout, o.cin_chunks, o.coutp), &o.wt))) return rc;
                    h->mfma_flops_per_patch += o.flops;
                    if (cout >= 32 && d.pad_top <= 1 && d.pad_left <= 1) {
                        const int k = d.kh;
                        // taps of output phase c along one axis, ascending input offset: kernel index kh = c + crop (mod 2), offset (c + crop - kh) / 2
                        auto taps = [&](int c, int crop, int idx[2], int& lead) {
                            int n = 0, dmin = 0, off[2] = {0, 0};
                            for (int kk = k - 1; kk >= 0; --kk)
                                if (((c + crop - kk) & 1) == 0) { off[n] = (c + crop - kk) / 2; idx[n] = kk; ++n; }      // kk descending = offset ascending
                            dmin = off[0];
                            lead = -dmin;
                            return n;
                        };
                        for (int cy = 0; cy < 2; ++cy)
                            for (int cx = 0; cx < 2; ++cx) {
                                int ky[2], kx[2], pt = 0, pl = 0;
                                const int R = taps(cy, d.pad_top, ky, pt), S = taps(cx, d.pad_left, kx, pl);
                                std::vector<float> hw((size_t)R * S * cin * cout);      // HWIO filter of this phase's forward convolution
                                for (int r = 0; r < R; ++r)
                                    for (int q = 0; q < S; ++q)
                                        for (int ci = 0; ci < cin; ++ci)
                                            for (int co = 0; co < cout; ++co)
                                                hw[(((size_t)r * S + q) * cin + ci) * cout + co] = kw[(((size_t)ky[r] * k + kx[q]) * cout + co) * cin + ci];
                                const int ph = cy * 2 + cx;
                                o.ph_R[ph] = R; o.ph_S[ph] = S; o.ph_pt[ph] = pt; o.ph_pl[ph] = pl;
                                if ((rc = upload(h, relayout_conv(hw.data(), R, S, cin, cout, o.cin_chunks, o.coutp), &o.ph_wt[ph]))) return rc;
                            }
                    }
                } else {
                    o.path = PATH_GENERIC;
                    if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * cin * cout), &o.wt))) return rc;
                }
            }
            h->flops_per_patch += o.flops;
        } else if (d.op == ECSEG_OP_AFFINE) {
            const float *sc = nullptr, *sh = nullptr;
            if ((rc = W(d.w0, to.c, "affine scale", &sc))) return rc;
            if ((rc = W(d.w1, to.c, "affine shift", &sh))) return rc;
            if (!sc || !sh) return fail(h, ECSEG_E_INVALID, "affine without scale/shift in op " + std::to_string(k));
            if ((rc = upload(h, std::vector<float>(sc, sc + to.c), &o.scale))) return rc;
            if ((rc = upload(h, std::vector<float>(sh, sh + to.c), &o.shift))) return rc;
        } else if (d.op == ECSEG_OP_DWCONV) {
            const int mult = d.mode;
            if (d.kh <= 0 || d.kw <= 0 || d.stride <= 0 || mult < 1 || to.c != ti.c * mult)
                return fail(h, ECSEG_E_INVALID, "bad depthwise-conv geometry in op " + std::to_string(k));
            const int dil = d.dilation > 1 ? d.dilation : 1;
            if ((to.h - 1) * d.stride + 1 > ti.h + (d.kh - 1) * dil || (to.w - 1) * d.stride + 1 > ti.w + (d.kw - 1) * dil || d.pad_top < 0 || d.pad_left < 0)
                return fail(h, ECSEG_E_INVALID, "depthwise-conv output larger than its input in op " + std::to_string(k));
            const float *kw = nullptr, *kb = nullptr;
            if ((rc = W(d.w0, (int64_t)d.kh * d.kw * to.c, "depthwise kernel", &kw))) return rc;
            if (!kw) return fail(h, ECSEG_E_INVALID, "depthwise conv without kernel in op " + std::to_string(k));
            if ((rc = W(d.w1, to.c, "depthwise bias", &kb))) return rc;
            if ((rc = upload(h, std::vector<float>(kw, kw + (size_t)d.kh * d.kw * to.c), &o.wt))) return rc;
            if (kb) { if ((rc = upload(h, std::vector<float>(kb, kb + to.c), &o.bias))) return rc; }
            o.flops = 2.0 * d.kh * d.kw * to.c * (double)to.h * to.w;
            h->flops_per_patch += o.flops;
        } else if (d.op == ECSEG_OP_PRELU) {
            const float* a = nullptr;
            const int64_t len = d.mode ? (int64_t)to.h * to.w * to.c : (int64_t)to.c;
            if ((rc = W(d.w0, len, "PReLU slopes", &a))) return rc;
            if (!a) return fail(h, ECSEG_E_INVALID, "PReLU without slopes in op " + std::to_string(k));
            if ((rc = upload(h, std::vector<float>(a, a + len), &o.wt))) return rc;
        } else if (d.op == ECSEG_OP_LAYERNORM) {
            const float *g = nullptr, *b = nullptr;
            if ((rc = W(d.w0, to.c, "LayerNormalization gamma", &g))) return rc;
            if ((rc = W(d.w1, to.c, "LayerNormalization beta", &b))) return rc;
            if (g) { if ((rc = upload(h, std::vector<float>(g, g + to.c), &o.scale))) return rc; }
            if (b) { if ((rc = upload(h, std::vector<float>(b, b + to.c), &o.shift))) return rc; }
        } else if (d.op == ECSEG_OP_MAXPOOL || d.op == ECSEG_OP_UPSAMPLE) {
            if (d.stride <= 0) return fail(h, ECSEG_E_INVALID, "bad stride in op " + std::to_string(k));
            // 'valid' pooling stays inside the input; 'same' (pad_top / pad_left given, or the last window overhanging) may not
            // start a window beyond it
            if (d.op == ECSEG_OP_MAXPOOL && (d.kh <= 0 || d.kw <= 0 || d.pad_top < 0 || d.pad_left < 0 || d.pad_top >= d.kh || d.pad_left >= d.kw ||
                                             (to.h - 1) * d.stride - d.pad_top >= ti.h || (to.w - 1) * d.stride - d.pad_left >= ti.w))
                return fail(h, ECSEG_E_INVALID, "max-pool window leaves the input in op " + std::to_string(k));
            if (d.op == ECSEG_OP_UPSAMPLE && (to.h != ti.h * d.stride || to.w != ti.w * d.stride))
                return fail(h, ECSEG_E_INVALID, "bad upsample shape in op " + std::to_string(k));
        } else if (d.op == ECSEG_OP_GLOBALPOOL) {
            if (to.h != 1 || to.w != 1 || to.c != ti.c) return fail(h, ECSEG_E_INVALID, "bad global-pool shape in op " + std::to_string(k));
        } else if (d.op == ECSEG_OP_ADD) {
            if (d.mode < ECSEG_BIN_ADD || d.mode > ECSEG_BIN_MIN) return fail(h, ECSEG_E_INVALID, "bad binary mode in op " + std::to_string(k));
            for (const ecseg_tensor_desc* tb : {&ti, &tensors[d.in1]})
                if ((tb->h != to.h && tb->h != 1) || (tb->w != to.w && tb->w != 1) || (tb->c != to.c && tb->c != 1))
                    return fail(h, ECSEG_E_INVALID, "shapes cannot be broadcast in op " + std::to_string(k));
        } else if (d.op == ECSEG_OP_ACT || d.op == ECSEG_OP_COPY) {
            // nothing to prepare
        } else {
            return fail(h, ECSEG_E_INVALID, "unknown op code in op " + std::to_string(k));
        }
        if (d.op != ECSEG_OP_CONV && d.op != ECSEG_OP_CONVT && d.op != ECSEG_OP_MAXPOOL && d.op != ECSEG_OP_UPSAMPLE && d.op != ECSEG_OP_DWCONV &&
            d.op != ECSEG_OP_ADD && d.op != ECSEG_OP_COPY && d.op != ECSEG_OP_GLOBALPOOL && (ti.h != to.h || ti.w != to.w || ti.c != to.c))
            return fail(h, ECSEG_E_INVALID, "shape mismatch in element-wise op " + std::to_string(k));
        h->ops.push_back(o);
    }
    // window lanes (run_plan) address the model input and output in plain window order: allowed when their buffers hold
    // only tensors of exactly the buffer's per-window size (keras_plan gives both a buffer of their own)
    h->lanes_ok = true;
    for (int io : {input_tensor, output_tensor}) {
        const int b = h->tensors[io].buffer;
        for (const ecseg_tensor_desc& t : h->tensors)
            if (t.buffer == b && (size_t)t.h * t.w * t.c_stride != std::max<size_t>(h->buf_floats[b], 4)) h->lanes_ok = false;
    }
    h->consumers.assign(n_tensors, 0);
    for (const OpRt& o : h->ops) {
        if (o.d.in0 >= 0) ++h->consumers[o.d.in0];
        if (o.d.op == ECSEG_OP_ADD && o.d.in1 >= 0) ++h->consumers[o.d.in1];
    }
    {   // crop recipes: walk back from the model output.  A 1x1 convolution passes its reader's need on, a 3x3 'same'
        // convolution needs its input one pixel further out ('d'); a concatenation is followed through the view written
        // by a 2x2 / stride-2 transposed convolution (which itself computes everything, from an input needed at half the
        // coordinates, 'h'); skip connections and anything with several readers keep their full extent and end the walk
        int t = output_tensor, reader = (int)h->ops.size();
        std::string code;
        for (int guard = 0; guard < 32 && code.size() < 16; ++guard) {
            int prod = -1, nprod = 0;
            for (int k = 0; k < reader; ++k) if (h->ops[k].d.out == t) { prod = k; ++nprod; }
            if (nprod == 0) {
                const ecseg_tensor_desc& tt = tensors[t];
                int up = -1;
                for (int k = 0; k < reader; ++k) {
                    const ecseg_op_desc& od = h->ops[k].d;
                    const ecseg_tensor_desc& tv = tensors[od.out];
                    if (od.op == ECSEG_OP_CONVT && tv.buffer == tt.buffer && tv.h == tt.h && tv.w == tt.w && tv.c_stride == tt.c_stride &&
                        tv.c < tt.c && h->consumers[od.out] == 0) up = k;                  // the last such writer before the reader
                }
                if (up < 0) break;
                const ecseg_op_desc& ud = h->ops[up].d;
                const ecseg_tensor_desc& ui = tensors[ud.in0];
                if (!(ud.kh == 2 && ud.kw == 2 && ud.stride == 2 && ud.pad_top == 0 && ud.pad_left == 0 && ui.h * 2 == tt.h && ui.w * 2 == tt.w)) break;
                if (h->consumers[ud.in0] != 1 || ui.c_stride != ui.c || ui.c_offset != 0) break;
                code += 'h';
                h->ops[up].crop_ok = true; h->ops[up].crop_code = code;      // the part of its INPUT that matters
                t = ud.in0; reader = up;
                continue;
            }
            if (nprod != 1) break;
            OpRt& o = h->ops[prod];
            const ecseg_tensor_desc& ti = tensors[o.d.in0];
            if (o.d.op != ECSEG_OP_CONV || o.d.dilation > 1 || o.d.stride != 1 || (o.d.mode & 0xffff)) break;
            o.crop_ok = true; o.crop_code = code;
            if (o.d.kh == 3 && o.d.kw == 3 && o.d.pad_top == 1 && o.d.pad_left == 1) code += 'd';
            else if (!(o.d.kh == 1 && o.d.kw == 1)) break;
            if (h->consumers[o.d.in0] != 1 || ti.c_stride != ti.c || ti.c_offset != 0) break;
            t = o.d.in0; reader = prod;
        }
    }
    h->has_model = true;
    return ensure_split_images(h);                         // ("winograd" = 3 set before the load)
}

int ecseg_model_flops_per_patch(ecseg_ctx* h, double* flops) {
    int rc = check_model(h);
    if (rc) return rc;
    if (flops) *flops = h->flops_per_patch;
    return ECSEG_OK;
}

static int forward_host(ecseg_ctx* h, const void* patches, bool is_f32, int n, float* out) {
    if (h) drop_sent_ahead(h);
    int rc = check_model(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!patches || !out))) return fail(h, ECSEG_E_INVALID, "forward_patches: bad arguments");
    if (n == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const ecseg_tensor_desc& ti = h->tensors[h->input_tensor];
    const ecseg_tensor_desc& to = h->tensors[h->output_tensor];
    if (ti.c_stride != ti.c || ti.c_offset != 0) return fail(h, ECSEG_E_INVALID, "input tensor must be compact");
    const size_t in_per = (size_t)ti.h * ti.w * ti.c, out_per = (size_t)to.h * to.w * to.c;
    const int chunk = std::max(1, windows_per_group(h));
    if ((rc = ensure_patches(h, std::min(n, chunk)))) return rc;
    if (!is_f32 && (rc = ensure(h, h->d_u8in, h->d_u8in_cap, in_per * std::min(n, chunk)))) return rc;
    hipStream_t s = h->stream;
    prof_begin(h);
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int ni = std::min(chunk, n - i0);
        if (is_f32) {
            HIP_TRY(h, hipMemcpyAsync(view_of(h, h->input_tensor).p, static_cast<const float*>(patches) + (size_t)i0 * in_per,
                                      in_per * ni * sizeof(float), hipMemcpyHostToDevice, s));
        } else {
            HIP_TRY(h, hipMemcpyAsync(h->d_u8in, static_cast<const uint8_t*>(patches) + (size_t)i0 * in_per, in_per * ni, hipMemcpyHostToDevice, s));
            HIP_TRY(h, launch_u8_to_f32(h->d_u8in, view_of(h, h->input_tensor).p, in_per * ni, s));
        }
        if ((rc = run_plan(h, ni))) return rc;
        const TView ov = view_of(h, h->output_tensor);
        HIP_TRY(h, hipMemcpy2DAsync(out + (size_t)i0 * out_per, (size_t)to.c * sizeof(float), ov.p, (size_t)ov.cs * sizeof(float),
                                    (size_t)to.c * sizeof(float), (size_t)to.h * to.w * ni, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    prof_end(h);
    return ECSEG_OK;
}

int ecseg_forward_patches(ecseg_ctx* h, const uint8_t* patches, int n, float* out) { return forward_host(h, patches, false, n, out); }
int ecseg_forward_patches_f32(ecseg_ctx* h, const float* patches, int n, float* out) { return forward_host(h, patches, true, n, out); }

int ecseg_read_tensor(ecseg_ctx* h, int tensor, int n, float* out) {
    int rc = check_model(h);
    if (rc) return rc;
    if (tensor < 0 || tensor >= (int)h->tensors.size() || n <= 0 || n > h->cap_patches || !out)
        return fail(h, ECSEG_E_INVALID, "read_tensor: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    const ecseg_tensor_desc& t = h->tensors[tensor];
    const TView v = view_of(h, tensor);
    HIP_TRY(h, hipMemcpy2DAsync(out, (size_t)t.c * sizeof(float), v.p, (size_t)v.cs * sizeof(float), (size_t)t.c * sizeof(float),
                                (size_t)t.h * t.w * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ECSEG_OK;
}

int ecseg_segment_images_dev(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W, uint8_t* raw, uint8_t* post, int32_t* n_ec) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || (n_img > 0 && (!gray || !post))) return fail(h, ECSEG_E_INVALID, "segment: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W;
    int rc;
    uint8_t* raw_buf = raw;
    if (!raw_buf) {
        if ((rc = ensure(h, h->d_raw, h->d_raw_cap, px * n_img))) return rc;
        raw_buf = h->d_raw;
    }
    return segment_dev(h, gray, n_img, H, W, raw_buf, post, n_ec);
}

int ecseg_segment_images(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W, uint8_t* raw, uint8_t* post, int32_t* n_ec) {
    return ecseg_segment_images_ex(h, gray, n_img, H, W, raw, post, n_ec, nullptr, nullptr);
}

int ecseg_segment_images_ex(ecseg_ctx* h, const uint8_t* gray, int n_img, int H, int W, uint8_t* raw, uint8_t* post, int32_t* n_ec,
                            int32_t* tie_risk, float* probs) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || (n_img > 0 && (!gray || !post))) return fail(h, ECSEG_E_INVALID, "segment: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    int rc = check_model(h);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W, tot = px * n_img;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_raw, h->d_raw_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_post, h->d_post_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_i32, h->d_i32_cap, (size_t)n_img))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_gray, gray, tot, hipMemcpyHostToDevice, h->stream));
    if ((rc = segment_dev(h, h->d_gray, n_img, H, W, h->d_raw, h->d_post, h->d_i32, probs))) return rc;
    if (raw) HIP_TRY(h, hipMemcpyAsync(raw, h->d_raw, tot, hipMemcpyDeviceToHost, h->stream));
    if (tie_risk) HIP_TRY(h, hipMemcpyAsync(tie_risk, h->d_tie, (size_t)n_img * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(post, h->d_post, tot, hipMemcpyDeviceToHost, h->stream));
    if (n_ec) HIP_TRY(h, hipMemcpyAsync(n_ec, h->d_i32, (size_t)n_img * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ECSEG_OK;
}

int ecseg_preprocess(ecseg_ctx* h, const void* img, int n_img, int H, int W, int C, int bps, uint8_t* gray_out, int32_t* inverted_out) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || H <= 0 || W <= 0 || (C != 1 && C != 3 && C != 4) || (bps != 1 && bps != 2) || (n_img > 0 && (!img || !gray_out)))
        return fail(h, ECSEG_E_INVALID, "preprocess: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W, tot = px * n_img, in_bytes = tot * C * bps;
    int rc;
    if ((rc = ensure(h, h->d_aux8, h->d_aux8_cap, in_bytes))) return rc;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_i32, h->d_i32_cap, (size_t)n_img))) return rc;
    if ((rc = ensure(h, h->d_hist, h->d_hist_cap, (size_t)n_img * 256))) return rc;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_aux8, img, in_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(h->ev[0], s));                 // the kernels alone (inputs resident): ecseg_get_timings()[ECSEG_T_COUNT]
    HIP_TRY(h, run_preprocess(h->d_aux8, n_img, H, W, C, bps, h->d_gray, h->d_i32, h->d_hist, s));
    HIP_TRY(h, hipEventRecord(h->ev[1], s));
    HIP_TRY(h, hipMemcpyAsync(gray_out, h->d_gray, tot, hipMemcpyDeviceToHost, s));
    if (inverted_out) HIP_TRY(h, hipMemcpyAsync(inverted_out, h->d_i32, (size_t)n_img * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (float& v : h->stage_ms) v = 0.f;
    h->stage_ms[ECSEG_T_COUNT] = stage_elapsed(h->ev[0], h->ev[1]);
    return ECSEG_OK;
}

// meta_segment of a batch in ONE call (src/utils.py:105-124 minus the file I/O, + src/metaseg.py:46): the raw images go up
// once, the pre-processed images never leave the device between meta_preprocess and the U-Net (the two-call sequence
// ecseg_preprocess + ecseg_segment_images_ex downloads them, synchronises and uploads them again), and their copy back
// to the host (dapi/<name> is written from it) travels on the second stream under the U-Net.
int ecseg_meta_segment(ecseg_ctx* h, const void* img, int n_img, int H, int W, int C, int bps, uint8_t* gray_out, uint8_t* post,
                       int32_t* n_ec, int32_t* tie_risk) {
    if (!h) return ECSEG_E_INVALID;
    if (n_img < 0 || H <= 0 || W <= 0 || (C != 1 && C != 3 && C != 4) || (bps != 1 && bps != 2) || (n_img > 0 && (!img || !post)))
        return fail(h, ECSEG_E_INVALID, "meta_segment: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    int rc = check_model(h);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W, tot = px * n_img, in_bytes = tot * C * bps;
    if ((rc = ensure(h, h->d_aux8, h->d_aux8_cap, in_bytes))) return rc;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_raw, h->d_raw_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_post, h->d_post_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_i32, h->d_i32_cap, (size_t)2 * n_img))) return rc;       // counts, then the inverted flags
    if ((rc = ensure(h, h->d_hist, h->d_hist_cap, (size_t)n_img * 256))) return rc;
    hipStream_t s = h->stream, sc = h->stream2;
    const double t0 = dbg_now();
    if (h->next_host == img) { h->next_host = nullptr; h->next_bytes = 0; }     // (registered for a call that never came: it names THIS call's images)
    if (h->pre_host == img && h->pre_bytes == in_bytes && h->d_pre) {
        // these images were sent ahead (ecseg_prefetch_input) while the call before this one computed: the two input buffers
        // change places (the one given up last held the images of the call before, whose pre-processing is long over)
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_pre, 0));
        std::swap(h->d_aux8, h->d_pre); std::swap(h->d_aux8_cap, h->d_pre_cap);
        h->pre_host = nullptr; h->pre_bytes = 0;
    } else {
        h->pre_host = nullptr; h->pre_bytes = 0;           // (images sent ahead are for the very next call or for nobody)
        HIP_TRY(h, hipMemcpyAsync(h->d_aux8, img, in_bytes, hipMemcpyHostToDevice, s));
    }
    const double t1 = dbg_now();
    HIP_TRY(h, run_preprocess(h->d_aux8, n_img, H, W, C, bps, h->d_gray, h->d_i32 + n_img, h->d_hist, s));
    if (gray_out) {
        HIP_TRY(h, hipEventRecord(h->ev[2], s));
        HIP_TRY(h, hipStreamWaitEvent(sc, h->ev[2], 0));
        HIP_TRY(h, hipMemcpyAsync(gray_out, h->d_gray, tot, hipMemcpyDeviceToHost, sc));
    }
    if (h->next_host) {                                    // the next call's images, registered by ecseg_prefetch_input
        const void* nx = h->next_host; const size_t nb = h->next_bytes;
        h->next_host = nullptr; h->next_bytes = 0;
        h->pre_host = nullptr; h->pre_bytes = 0;
        if (!h->stream_in) HIP_TRY(h, hipStreamCreateWithFlags(&h->stream_in, hipStreamNonBlocking));
        if (!h->ev_pre) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming));
        if ((rc = ensure(h, h->d_pre, h->d_pre_cap, nb))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_pre, nx, nb, hipMemcpyHostToDevice, h->stream_in));
        HIP_TRY(h, hipEventRecord(h->ev_pre, h->stream_in));
        h->pre_host = nx; h->pre_bytes = nb;
    }
    const double t2 = dbg_now();
    if ((rc = segment_dev(h, h->d_gray, n_img, H, W, h->d_raw, h->d_post, h->d_i32))) return rc;
    const double t3 = dbg_now();
    HIP_TRY(h, hipMemcpyAsync(post, h->d_post, tot, hipMemcpyDeviceToHost, s));
    if (tie_risk) HIP_TRY(h, hipMemcpyAsync(tie_risk, h->d_tie, (size_t)n_img * 4, hipMemcpyDeviceToHost, s));
    if (n_ec) HIP_TRY(h, hipMemcpyAsync(n_ec, h->d_i32, (size_t)n_img * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, wait_stream(h, s));
    const double t4 = dbg_now();
    HIP_TRY(h, wait_stream(h, sc));
    if (h->pre_host) HIP_TRY(h, wait_stream(h, h->stream_in));     // (long done: the caller's buffer is not read after this call)
    const double t5 = dbg_now();
    if (debug_calls()) fprintf(stderr, "[meta_segment n=%d] upload enqueue %.2f preprocess + gray copy enqueue %.2f segment_dev %.2f (stage timers %.2f) labels down %.2f gray wait %.2f total %.2f ms\n",
                     n_img, t1 - t0, t2 - t1, t3 - t2, h->stage_ms[0] + h->stage_ms[1] + h->stage_ms[2] + h->stage_ms[3], t4 - t3, t5 - t4, t5 - t0);
    return ECSEG_OK;
}

// Names the raw images of the call AFTER the coming ecseg_meta_segment call (same n_img x H x W x C x bytes_per_sample layout,
// `bytes` in total, page-locked memory: from pageable memory the copy would be staged by the calling thread inside the
// coming call and delay its kernels).  The coming call sends them ahead on a stream of their own, under its kernels (2.4 ms
// for 32 RGB images), into the spare input buffer; the call after it recognises its images by (pointer, size) and skips its
// own upload (so the images must not change in between).  A call with other images uploads as always and drops what was sent
// ahead.  The memory is read during the coming call only.
int ecseg_prefetch_input(ecseg_ctx* h, const void* img, size_t bytes) {
    if (!h) return ECSEG_E_INVALID;
    h->next_host = (img && bytes) ? img : nullptr;
    h->next_bytes = h->next_host ? bytes : 0;
    return ECSEG_OK;
}

// Page-locked host memory: hipMemcpyAsync from / to it is a DMA transfer the host thread does not wait for (pageable
// memory is staged through the runtime's own pinned chunks by the calling thread).  These two calls read nothing of the
// handle but its device number and never write its error string: they may run on another thread while the handle is
// inside a call (`make metaseg` page-locks batch buffers on a helper thread, ecseg_amd/metaseg.py: _PinnedPool).
int ecseg_host_alloc(ecseg_ctx* h, size_t bytes, void** out) {
    if (!h || !out) return ECSEG_E_INVALID;
    *out = nullptr;
    if (bytes == 0) return ECSEG_OK;
    if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return ECSEG_E_HIP; }
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return e == hipErrorOutOfMemory ? ECSEG_E_NOMEM : ECSEG_E_HIP; }
    return ECSEG_OK;
}

int ecseg_host_free(ecseg_ctx* h, void* p) {
    if (!h) return ECSEG_E_INVALID;
    if (!p) return ECSEG_OK;
    if (hipSetDevice(h->device) != hipSuccess || hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return ECSEG_E_HIP; }
    return ECSEG_OK;
}

int ecseg_u16_to_u8(ecseg_ctx* h, const uint16_t* in, long long count, uint8_t* out) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (count < 0 || (count > 0 && (!in || !out))) return fail(h, ECSEG_E_INVALID, "u16_to_u8: bad arguments");
    if (count == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = ensure(h, h->d_aux8, h->d_aux8_cap, (size_t)count * 2))) return rc;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, (size_t)count))) return rc;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_aux8, in, (size_t)count * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(h, launch_u16_to_u8(reinterpret_cast<const uint16_t*>(h->d_aux8), h->d_gray, (size_t)count, s));
    HIP_TRY(h, hipMemcpyAsync(out, h->d_gray, (size_t)count, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return ECSEG_OK;
}

int ecseg_stitch_argmax(ecseg_ctx* h, const float* probs, int n_img, int H, int W, uint8_t* labels_raw) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || (n_img > 0 && (!probs || !labels_raw))) return fail(h, ECSEG_E_INVALID, "stitch_argmax: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    StitchPlan* sp = nullptr;
    int rc;
    if ((rc = get_stitch(h, H, W, &sp))) return rc;
    const size_t px = (size_t)H * W, nfl = (size_t)n_img * sp->n_pos * 65536 * 4;
    if ((rc = ensure(h, h->d_probs_in, h->d_probs_cap, nfl))) return rc;
    if ((rc = ensure(h, h->d_raw, h->d_raw_cap, px * n_img))) return rc;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_probs_in, probs, nfl * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(h, launch_stitch_argmax(h->d_probs_in, 4, sp->map_dev, n_img, sp->n_pos, H, W, h->d_raw, s));
    HIP_TRY(h, hipMemcpyAsync(labels_raw, h->d_raw, px * n_img, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return ECSEG_OK;
}

int ecseg_meta_inference_dev(ecseg_ctx* h, const uint8_t* in, int n_img, int H, int W, uint8_t* out, int32_t* n_ec) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || H <= 0 || W <= 0 || (n_img > 0 && (!in || !out))) return fail(h, ECSEG_E_INVALID, "meta_inference: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    if ((long long)H * W >= (1ll << 31)) return fail(h, ECSEG_E_INVALID, "image too large");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W;
    int rc;
    if ((rc = ensure_post(h, std::min(n_img, h->post_chunk), px))) return rc;
    hipStream_t s = h->stream;
    if (out != in) HIP_TRY(h, hipMemcpyAsync(out, in, px * n_img, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipEventRecord(h->ev[0], s));
    for (int i0 = 0; i0 < n_img; i0 += h->post_chunk) {
        const int ni = std::min(h->post_chunk, n_img - i0);
        HIP_TRY(h, post_run(h, out + (size_t)i0 * px, ni, H, W, n_ec ? n_ec + i0 : nullptr, s));
    }
    HIP_TRY(h, hipEventRecord(h->ev[1], s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (float& v : h->stage_ms) v = 0.f;
    h->stage_ms[ECSEG_T_POST] = stage_elapsed(h->ev[0], h->ev[1]);
    return ECSEG_OK;
}

int ecseg_meta_inference(ecseg_ctx* h, const uint8_t* in, int n_img, int H, int W, uint8_t* out, int32_t* n_ec) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || H <= 0 || W <= 0 || (n_img > 0 && (!in || !out))) return fail(h, ECSEG_E_INVALID, "meta_inference: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t tot = (size_t)H * W * n_img;
    int rc;
    if ((rc = ensure(h, h->d_post, h->d_post_cap, tot))) return rc;
    if ((rc = ensure(h, h->d_i32, h->d_i32_cap, (size_t)n_img))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_post, in, tot, hipMemcpyHostToDevice, h->stream));
    if ((rc = ecseg_meta_inference_dev(h, h->d_post, n_img, H, W, h->d_post, h->d_i32))) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, h->d_post, tot, hipMemcpyDeviceToHost, h->stream));
    if (n_ec) HIP_TRY(h, hipMemcpyAsync(n_ec, h->d_i32, (size_t)n_img * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ECSEG_OK;
}

// shared driver of the mask-counting entry points: uploads one or two mask stacks, chunks over images
static int count_driver(ecseg_ctx* h, const uint8_t* a, const uint8_t* b, int n_img, int H, int W, int kind, int arg,
                        int32_t* n_out, int64_t* px_out, int32_t* labels_out) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || H <= 0 || W <= 0 || (n_img > 0 && !a)) return fail(h, ECSEG_E_INVALID, "count: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    if ((long long)H * W >= (1ll << 31)) return fail(h, ECSEG_E_INVALID, "image too large");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W;
    const int chunk = h->post_chunk;
    int rc;
    if ((rc = ensure_post(h, std::min(n_img, chunk), px))) return rc;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, px * std::min(n_img, chunk)))) return rc;
    if (b && (rc = ensure(h, h->d_aux8, h->d_aux8_cap, px * std::min(n_img, chunk)))) return rc;
    if ((rc = ensure(h, h->d_i32, h->d_i32_cap, labels_out ? px * std::min(n_img, chunk) : (size_t)chunk))) return rc;
    if ((rc = ensure(h, h->d_i64, h->d_i64_cap, (size_t)chunk))) return rc;
    hipStream_t s = h->stream;
    for (float& v : h->stage_ms) v = 0.f;
    for (int i0 = 0; i0 < n_img; i0 += chunk) {
        const int ni = std::min(chunk, n_img - i0);
        HIP_TRY(h, hipMemcpyAsync(h->d_gray, a + (size_t)i0 * px, px * ni, hipMemcpyHostToDevice, s));
        if (b) HIP_TRY(h, hipMemcpyAsync(h->d_aux8, b + (size_t)i0 * px, px * ni, hipMemcpyHostToDevice, s));
        hipError_t e = hipSuccess;
        HIP_TRY(h, hipEventRecord(h->ev[0], s));
        if (kind == 0) e = run_count_cc(h->ws, h->d_gray, ni, H, W, h->d_i32, h->d_i64, s);
        else if (kind == 1) e = run_count_coloc(h->ws, h->d_gray, h->d_aux8, ni, H, W, h->d_i32, s);
        else if (kind == 2) e = run_count_hsr(h->ws, h->d_gray, h->d_aux8, ni, H, W, arg, h->d_i32, s);
        else e = run_ccl_labels(h->ws, h->d_gray, ni, H, W, arg, h->d_i32, s);
        if (e != hipSuccess) return fail_hip(h, e, "count kernels");
        HIP_TRY(h, hipEventRecord(h->ev[1], s));
        if (labels_out) HIP_TRY(h, hipMemcpyAsync(labels_out + (size_t)i0 * px, h->d_i32, px * ni * 4, hipMemcpyDeviceToHost, s));
        else if (n_out) HIP_TRY(h, hipMemcpyAsync(n_out + i0, h->d_i32, (size_t)ni * 4, hipMemcpyDeviceToHost, s));
        if (px_out) HIP_TRY(h, hipMemcpyAsync(px_out + i0, h->d_i64, (size_t)ni * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        h->stage_ms[ECSEG_T_COUNT] += stage_elapsed(h->ev[0], h->ev[1]);
    }
    return ECSEG_OK;
}

int ecseg_count_cc(ecseg_ctx* h, const uint8_t* mask, int n_img, int H, int W, int32_t* n_out, int64_t* px_out) {
    return count_driver(h, mask, nullptr, n_img, H, W, 0, 0, n_out, px_out, nullptr);
}
int ecseg_ccl_labels(ecseg_ctx* h, const uint8_t* mask, int n_img, int H, int W, int connectivity, int32_t* labels_out) {
    if (h && connectivity != 4 && connectivity != 8) return fail(h, ECSEG_E_INVALID, "connectivity must be 4 or 8");
    if (h && n_img > 0 && !labels_out) return fail(h, ECSEG_E_INVALID, "labels_out is NULL");
    return count_driver(h, mask, nullptr, n_img, H, W, 3, connectivity, nullptr, nullptr, labels_out);
}
int ecseg_count_colocalization(ecseg_ctx* h, const uint8_t* ob1, const uint8_t* ob2, int n_img, int H, int W, int32_t* n_out) {
    if (h && n_img > 0 && !ob2) return fail(h, ECSEG_E_INVALID, "ob2 is NULL");
    return count_driver(h, ob1, ob2, n_img, H, W, 1, 0, n_out, nullptr, nullptr);
}
int ecseg_count_hsr(ecseg_ctx* h, const uint8_t* chrom, const uint8_t* fish, int n_img, int H, int W, int thr, int32_t* n_out) {
    if (h && n_img > 0 && !fish) return fail(h, ECSEG_E_INVALID, "fish is NULL");
    return count_driver(h, chrom, fish, n_img, H, W, 2, thr, n_out, nullptr, nullptr);
}

int ecseg_overlay(ecseg_ctx* h, const uint8_t* labels, const uint8_t* rgb, int n_img, int H, int W, int C, int sens, int hsr_thr,
                  int64_t* out) {
    if (!h) return ECSEG_E_INVALID;
    drop_sent_ahead(h);
    if (n_img < 0 || H <= 0 || W <= 0 || C < 2 || (n_img > 0 && (!labels || !rgb || !out)))
        return fail(h, ECSEG_E_INVALID, "overlay: bad arguments");
    if (n_img == 0) return ECSEG_OK;
    if ((long long)H * W >= (1ll << 31)) return fail(h, ECSEG_E_INVALID, "image too large");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t px = (size_t)H * W;
    const int chunk = h->post_chunk;
    int rc;
    if ((rc = ensure_post(h, std::min(n_img, chunk), px))) return rc;
    if ((rc = ensure(h, h->d_gray, h->d_gray_cap, px * std::min(n_img, chunk)))) return rc;
    if ((rc = ensure(h, h->d_aux8, h->d_aux8_cap, px * C * std::min(n_img, chunk)))) return rc;
    if ((rc = ensure(h, h->d_i64, h->d_i64_cap, (size_t)chunk * 12))) return rc;
    hipStream_t s = h->stream;
    for (float& v : h->stage_ms) v = 0.f;
    for (int i0 = 0; i0 < n_img; i0 += chunk) {
        const int ni = std::min(chunk, n_img - i0);
        HIP_TRY(h, hipMemcpyAsync(h->d_gray, labels + (size_t)i0 * px, px * ni, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(h->d_aux8, rgb + (size_t)i0 * px * C, px * C * ni, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipEventRecord(h->ev[0], s));             // the kernels alone (inputs resident): ecseg_get_timings()[ECSEG_T_COUNT]
        HIP_TRY(h, run_overlay(h->ws, h->d_gray, h->d_aux8, ni, H, W, C, sens, hsr_thr, h->d_i64, s));
        HIP_TRY(h, hipEventRecord(h->ev[1], s));
        HIP_TRY(h, hipMemcpyAsync(out + (size_t)i0 * 12, h->d_i64, (size_t)ni * 12 * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        h->stage_ms[ECSEG_T_COUNT] += stage_elapsed(h->ev[0], h->ev[1]);
    }
    return ECSEG_OK;
}

int ecseg_get_timings(ecseg_ctx* h, float* ms_out) {
    if (!h || !ms_out) return ECSEG_E_INVALID;
    for (int k = 0; k < ECSEG_T_N; ++k) ms_out[k] = h->stage_ms[k];
    return ECSEG_OK;
}

int ecseg_set_kernel_profiling(ecseg_ctx* h, int enabled) {
    if (!h) return ECSEG_E_INVALID;
    h->profile_kernels = enabled != 0;
    return ECSEG_OK;
}

int ecseg_get_conv_profile(ecseg_ctx* h, double* total_ms, int64_t* launches, double* flops) {
    if (!h) return ECSEG_E_INVALID;
    if (total_ms) *total_ms = h->last_conv_ms;
    if (launches) *launches = h->last_conv_launches;
    if (flops) *flops = h->last_conv_flops;
    return ECSEG_OK;
}

// Diagnostics: floats 16.. of the zero page (in-kernel cycle stamps); only in -DECSEG_DIAG builds (tools/build_variants.sh).
int ecseg_debug_peek(ecseg_ctx* h, float* out, int n) {
#ifdef ECSEG_DIAG
    if (!h || !out || n < 0 || n > 240 || !h->zero_page) return ECSEG_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(out, h->zero_page + 16, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return ECSEG_OK;
#else
    (void)out; (void)n;
    return fail(h, ECSEG_E_UNSUPPORTED, "ecseg_debug_peek: the shipped library carries no diagnostic kernels (build with -DECSEG_DIAG)");
#endif
}

int ecseg_get_conv_launch_profile(ecseg_ctx* h, int max_records, int32_t* op_index, int32_t* kind, float* ms, double* flops,
                                  double* executed_flops) {
    if (!h || max_records < 0) return ECSEG_E_INVALID;
    const int n = (int)std::min<size_t>(h->prof_recs.size(), (size_t)max_records);
    for (int k = 0; k < n; ++k) {
        const ecseg_ctx::ProfRec& r = h->prof_recs[k];
        if (op_index) op_index[k] = r.op;
        if (kind) kind[k] = r.kind;
        if (ms) ms[k] = r.ms;
        if (flops) flops[k] = r.flops;
        if (executed_flops) executed_flops[k] = r.exec_flops;
    }
    return n;
}

int ecseg_get_conv_executed_flops(ecseg_ctx* h, double* flops) {
    if (!h || !flops) return ECSEG_E_INVALID;
    *flops = h->last_conv_exec_flops;
    return ECSEG_OK;
}

}  // extern "C"
