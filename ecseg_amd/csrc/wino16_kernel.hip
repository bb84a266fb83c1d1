// Winograd F(2x2, 3x3) for NARROW layers (Cin, Cout in {16, 32}: the full-resolution levels of base-16 / base-32 U-Nets)
// on v_mfma_f32_16x16x4_f32 (gfx950).
//
// These layers are within 1.5x of their HBM bound (36-72 flop/B); the 32-wide kernels (conv_wino_kernel,
// conv_wino_res_kernel) run them at 0.16-0.45 of the MFMA peak because (a) a 32-column MFMA tile is half empty at 16
// output channels, (b) their four waves split the transform ROWS and have to meet through LDS after every tile
// (64 KB of LDS traffic and two barriers per 128 output pixels), (c) the streaming kernel re-reads a filter slab that is
// larger than its input tile.  This kernel is built the other way round:
//   * MFMA roles swapped: A = transformed filter (M = 16 output channels), B = transformed input (N = 16 Winograd tiles),
//     K = 4 channels.  The accumulator of a lane is then 4 CONSECUTIVE OUTPUT CHANNELS of ONE tile for each of the 16
//     transform points: the output transform Y = A^T M A, bias, activation, a fused 2x2 max-pool and the 16-byte stores
//     all happen in registers - no exchange through LDS, no barrier on the output path;
//   * a wave owns 16 tiles (2 x 8) and ALL 16 points; lane = (tile, channel quad kq): it reads its tile's 4 x 4 input
//     pixels x 4 channels with 16 conflict-free ds_read_b128, transforms them (32 adds per channel) and feeds 64 * NB
//     MFMAs per 16 input channels;
//   * workgroup = 8 waves = 16 x 32 output pixels; it walks `bpw` such blocks of a 16-row strip.  The whole transformed
//     filter (16 KB per 16 x 16 channels) is loaded into LDS once per workgroup; the input halo (18 x 34 pixels x 16
//     channels) arrives by LDS-DMA into a double buffer, one stage (block, 16-channel chunk) ahead; one s_barrier per stage.
//
// LDS halo image (16-byte slots = 4 channels of a pixel), round 3: slot(y, x, cq) = 138 y + 4 x + (cq ^ (((x >> 2) & 1) << 1)) -
// a pixel's four channel quads sit next to each other (in an order that flips with bit 2 of x) and a halo row is 34 pixels =
// 136 slots + 2 of padding.  A 64-slot LDS-DMA piece is then 16 CONSECUTIVE pixels of a halo row with all their channels: 1 KiB of
// contiguous global memory at 16 input channels (8 cache lines; 16 half lines at 32 channels) instead of 64 pieces of 16 bytes
// from 64 different lines (rounds 1-2: slot = 152 y + OFF[cq][x & 1] + (x >> 1), 370-540 issue cycles per piece against 55-100
// for a contiguous one), and a buffer needs 39 instead of 43 pieces.  Conflict-free all the same: the hardware serves a
// ds_read_b128 in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32), i.e. 8 tiles of quad kq and the other 8
// tiles of quad kq ^ 1; for every halo offset (i, j) their 16 slots are distinct modulo 16 (138 = 10 mod 16 separates the two
// tile rows by 4, the quad flip separates tiles 4 apart; verified exhaustively by tools/w16_layout_check.py).
#include <type_traits>

#include "common.h"
#include "device_util.h"

namespace ecseg {

namespace {

constexpr int W16_PITCH = 138;       // slots per halo row (136 used)
constexpr int W16_ROWS = 18;
constexpr int W16_PIECES = 39;       // 64-slot DMA pieces per halo buffer (18 * 138 = 2484 slots, padded to 2496)
constexpr int W16_HS = W16_PIECES * 64;
constexpr int W16_NP = 5;            // pieces per thread and stage: piece = wave + 8 k

typedef __attribute__((address_space(3))) void* w16_lptr_t;

// One LDS-DMA piece: 64 lanes x 16 bytes, global (per-lane address) -> LDS bytes [lds_dst + 16 * lane] (lds_dst is
// wave-uniform).  Inline asm for the reason given in wino4_kernel.hip: behind the builtin the compiler drains vmcnt
// before every later LDS read.  The waits in the kernel below are the only ordering; M0 is restored.
__device__ __forceinline__ void w16_dma16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

}  // namespace

// NBUF = 2: halo double buffer, the next stage's DMA goes out at the stage start, one barrier per stage.  NBUF = 1 (16 -> 16
// channels): one halo buffer - the transform is the only reader of the halo, so a barrier right behind it frees the buffer
// and the next stage's DMA lands under the MFMAs and the output stage: two barriers per stage, but 60 instead of 104 KB of
// LDS and <= 128 registers: two workgroups (four waves per SIMD) share a CU and fill each other's VALU / wait phases.
// HEAD: a following 1x1 convolution with <= 4 output channels (the U-Net's softmax head) is finished by the output stage:
// a lane holds 4 of a pixel's channels, the four lanes kq = 0..3 of a tile hold them all - partial logits per lane, a
// reduce-scatter over the two lane bits (12 cross-lane moves), lane kq finishes pixel kq of the tile's 2 x 2.
// FIRST (round 5): p.in is the network's 1-channel input and the halo of every stage is COMPUTED here - 16 channels (chunk kc) of
// the network's first layer (Conv2D 3x3 'same', 1 -> 16 KC channels + bias + activation) as a 16 x 16 x 12 GEMM per 16 halo pixels
// on the same matrix cores: A = the first layer's filter (16 output channels x 9 taps, zero padded to 12 = three k-steps), B = the nine
// neighbours of 16 halo pixels read from a raw 20 x 40 patch of the input in LDS (LDS-DMA, double buffered), D = 4 channels of
// one pixel per lane = exactly one 16-byte slot of the halo image.  612 halo pixels = 39 groups of 16 -> 117 MFMAs per block on
// top of the layer's own 512; in exchange the 16-channel tensor between the two layers (9.4 GB per 64 images, written by
// conv_first_kernel and read back here) never touches HBM.
constexpr int W16_RAW_PITCH = 40;    // raw patch: rows y0 - 2 .. y0 + 17, columns x0 - 4 .. x0 + 35 (whole 16-byte granules)
constexpr int W16_RAW_SLOTS = 256;   // 20 rows x 10 granules = 200, padded to 4 DMA pieces

template <int KC, int NB, int NBUF, bool HEAD, bool FIRST = false>
__global__ __launch_bounds__(512, (NBUF == 1 ? 4 : 2)) void conv_wino16_kernel(ConvParams p, int blocks_x, int strips_y, int segs_x, int bpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);              // [NBUF][W16_HS] halo buffer(s)
    f32x4* Fs = Hs + NBUF * W16_HS;                          // [16 points][KC][NB][64 lanes]: MFMA A fragments, 4 k-steps each
    f32x4* Hw = Fs + 16 * KC * NB * 64;                      // HEAD: [16 NB channels] x 4 classes
    f32x4* Rs = Hw + (HEAD ? 16 * NB : 0);                   // FIRST: [2][W16_RAW_SLOTS] raw input patches
    const unsigned lds_base = (unsigned)(size_t)(w16_lptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int H = p.in.h, W = p.in.w;                        // output extent == input extent
    // The blocks (16 x 32 output pixels) of this workgroup: `bpw` consecutive blocks of a 16-row strip - or, in a cropped
    // launch (p.lut: the regions some later stage reads, origins in 4-pixel units, the same list for every image), `bpw`
    // consecutive entries of the region list
    const bool use_lut = p.lut != nullptr;
    int w_img = 0, w_y0 = 0, w_bx0 = 0, w_first = 0, nblk;
    if (use_lut) {
        w_first = (int)bid * bpw;
        nblk = min(bpw, blocks_x - w_first);                 // (blocks_x = entries in all)
    } else {
        const int seg = bid % segs_x; bid /= segs_x;
        w_y0 = (int)(bid % strips_y) * 16; bid /= strips_y;
        w_img = (int)bid;
        w_bx0 = seg * bpw;
        nblk = min(bpw, blocks_x - w_bx0);
    }
    auto block_origin = [&](int blk, int& img, int& y0, int& x0) __attribute__((always_inline)) {      // all wave-uniform
        if (use_lut) {
            const int e = w_first + blk, i = e / p.lut_len, v = p.lut[e - i * p.lut_len];
            img = i * p.per_image + (v >> 16); y0 = ((v >> 8) & 255) * 4; x0 = (v & 255) * 4;
        } else {
            img = w_img; y0 = w_y0; x0 = (w_bx0 + blk) * 32;
        }
    };

    // ---- DMA descriptors of this thread: slot q = (wave + 8 k) * 64 + lane of a halo buffer ----
    int d_rel[W16_NP];                                       // float offset of the piece relative to halo pixel (0, 0), channel 0 of the chunk
    int d_meta[W16_NP];                                      // halo column (0..33) | halo row << 8 | 0x10000: padding slot
#pragma unroll
    for (int k = 0; k < W16_NP; ++k) {
        const int q = (wave + 8 * k) * 64 + lane;
        const int y = q / W16_PITCH, r = q - y * W16_PITCH;
        const int x = r >> 2, cq = (r & 3) ^ (((x >> 2) & 1) << 1);
        const bool ok = r < 136 && y < W16_ROWS;
        d_meta[k] = (ok ? x : 0) | (y << 8) | (ok ? 0 : 0x10000);
        d_rel[k] = ok ? (y * W + x) * p.in.cs + cq * 4 : 0;
    }
    // input rows / columns a block may read (ConvParams::in_box: the receptive field of the needed outputs in a cropped plan,
    // per window; the whole image otherwise); re-read when the window changes (all wave-uniform)
    int hb_img = -1, lo_y = 0, hi_y = H - 1, lo_x = 0, hi_x = W - 1;
    auto dma_halo = [&](int blk, int kc, int buf) __attribute__((always_inline)) {
        int img, y0, x0;
        block_origin(blk, img, y0, x0);
        if (p.in_box != nullptr && img != hb_img) {
            const int32_t* bx = p.in_box + 4 * ((img + p.box_first) % p.per_image);
            lo_y = bx[0]; hi_y = bx[1]; lo_x = bx[2]; hi_x = bx[3];
            hb_img = img;
        }
        const float* base = p.in.p + (((long)img * H + (y0 - 1)) * W + (x0 - 1)) * p.in.cs + kc * 16;      // halo pixel (0, 0); only dereferenced inside the image
#pragma unroll
        for (int k = 0; k < W16_NP; ++k) {
            const int piece = wave + 8 * k;                  // wave-uniform
            if (piece < W16_PIECES) {
                const int ix = x0 - 1 + (d_meta[k] & 0xff), iy = y0 - 1 + ((d_meta[k] >> 8) & 0xff);
                const float* src = p.zero;
                if (!(d_meta[k] & 0x10000) && ix >= lo_x && ix <= hi_x && iy >= lo_y && iy <= hi_y) src = base + d_rel[k];
                w16_dma16(src, lds_base + (unsigned)(buf * W16_HS + piece * 64) * 16u);
            }
        }
    };

    // ---- FIRST: raw-patch DMA (waves 0..3: one 64-granule piece each) and the halo fill on the matrix cores ----
    const unsigned raw_base = lds_base + (unsigned)((size_t)(reinterpret_cast<char*>(Rs) - smem));
    auto dma_raw = [&](int blk, int buf) __attribute__((always_inline)) {
        if (wave >= W16_RAW_SLOTS / 64) return;              // wave-uniform
        int img, y0, x0;
        block_origin(blk, img, y0, x0);
        const int q = wave * 64 + lane, row = q / 10, gc = q - row * 10;
        const int iy = y0 - 2 + row, ix = x0 - 4 + 4 * gc;
        const float* src = p.zero;
        if (q < 200 && iy >= 0 && iy < H && ix >= 0 && ix < W) src = p.in.p + ((long)img * H + iy) * W + ix;     // (W % 4 == 0: whole granules)
        w16_dma16(src, raw_base + (unsigned)(buf * W16_RAW_SLOTS + wave * 64) * 16u);
    };
    // chunk kc of the first layer into halo buffer hbuf, from raw patch rbuf.  This lane's A operand: the filter value of output
    // channel 16 kc + (lane & 15) at tap 4 ks + (lane >> 4) (re-read per fill: the wide variants have no registers to park it in)
    auto fill_halo = [&](int blk, int kc, int hbuf, int rbuf) __attribute__((always_inline)) {
        int img, y0, x0;
        block_origin(blk, img, y0, x0);
        const float* R = reinterpret_cast<const float*>(Rs + rbuf * W16_RAW_SLOTS);
        f32x4* Hd = Hs + hbuf * W16_HS;
        const int n = lane & 15, cq = lane >> 4;
        float fa[3];
        int ftap[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int k = 4 * ks + cq;
            fa[ks] = k < 9 ? p.first_w[k * (16 * KC) + 16 * kc + n] : 0.f;
            ftap[ks] = k < 9 ? (k / 3) * W16_RAW_PITCH + (k % 3) + 2 : 0;
        }
        f32x4 fbv = {0.f, 0.f, 0.f, 0.f};
        if (p.first_b != nullptr) {
            unsigned cqv = (unsigned)cq;
            asm volatile("" : "+v"(cqv));                     // (the address is formed here, per fill: hoisted out of the block loop it cost the HEAD instantiation two spilled registers)
            fbv = reinterpret_cast<const f32x4*>(p.first_b + 16 * kc)[cqv];
        }
#pragma unroll 1
        for (int grp = wave; grp < (W16_ROWS * 34 + 15) / 16; grp += 8) {      // 39 groups of 16 halo pixels
            const int pix = min(grp * 16 + n, W16_ROWS * 34 - 1);
            const int y = pix / 34, x = pix - y * 34;
            const float* rp = R + y * W16_RAW_PITCH + x;     // tap (dy, dx) of halo pixel (y, x): raw row y + dy, column x + dx + 2
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], rp[ftap[ks]], acc, 0, 0, 0);
            acc = apply_act4_core(acc + fbv, p.first_act, p.first_alpha);
            const int iy = y0 - 1 + y, ix = x0 - 1 + x;
            if (!(iy >= 0 && iy < H && ix >= 0 && ix < W)) acc = f32x4{0.f, 0.f, 0.f, 0.f};     // the second layer's zero padding
            if (grp * 16 + n < W16_ROWS * 34) Hd[y * W16_PITCH + 4 * x + (cq ^ (((x >> 2) & 1) << 1))] = acc;
        }
    };

    // ---- prologue: the filter image (linear copy) and the first halo ----
#pragma unroll
    for (int k = 0; k < 2 * KC * NB; ++k) {
        const int piece = wave + 8 * k;                      // 16 * KC * NB pieces in all
        w16_dma16(p.wt + ((size_t)piece * 64 + lane) * 4, lds_base + (unsigned)(NBUF * W16_HS + piece * 64) * 16u);
    }
    if (FIRST) dma_raw(0, 0);
    else dma_halo(0, 0, 0);

    // ---- lane geometry: tile (tr, tc) of the wave's 2 x 8 tiles, channel quad kq ----
    const int m = lane & 15, kq = lane >> 4;
    const int TR = 2 * (wave & 3) + (m >> 3), TC = 8 * (wave >> 2) + (m & 7);
    // halo pixel (2 TR + i, 2 TC + j), quad kq: columns j = 0, 1 share bit 2 of x (off_a), columns 2, 3 may have crossed it (off_b)
    const int off_a = (2 * TR) * W16_PITCH + 8 * TC + (kq ^ ((((2 * TC) >> 2) & 1) << 1));
    const int off_b = (2 * TR) * W16_PITCH + 8 * TC + 8 + (kq ^ ((((2 * TC + 2) >> 2) & 1) << 1));

    f32x4 bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        bv[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bv[nb] = *reinterpret_cast<const f32x4*>(p.bias + 16 * nb + 4 * kq);
    }

    if (HEAD && tid < 16 * NB) Hw[tid] = reinterpret_cast<const f32x4*>(p.head_w)[tid];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    const int nstage = nblk * KC;
    int st = 0;
    // in-kernel cycle stamps per phase (tools/w16_stamp_probe.py): diagnostic builds only (tools/build_variants.sh diag)
#ifdef ECSEG_DIAG
    unsigned long long stp[6] = {0, 0, 0, 0, 0, 0}, tk = __builtin_amdgcn_s_memtime();
    const unsigned long long tk0 = tk;
#define W16_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stp[i] += now - tk; tk = now; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define W16_STAMP(i) do { } while (0)
#endif
    for (int blk = 0; blk < nblk; ++blk) {
        f32x4 acc[16][NB];
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[pt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc, ++st) {
            // next stage's halo goes out first: it lands under this stage's MFMAs (its buffer was last read in stage st - 1)
            if (!FIRST && NBUF == 2 && st + 1 < nstage) {
                if (kc + 1 < KC) dma_halo(blk, kc + 1, (st + 1) & 1);
                else dma_halo(blk + 1, 0, (st + 1) & 1);
            }
            if (FIRST) {
                // the next block's raw patch goes out first (its buffer was last read two blocks ago), then this stage's halo is
                // computed from the patch that landed during the previous block (halo buffer st & 1 was last read two stages ago)
                if (kc == 0 && blk + 1 < nblk) dma_raw(blk + 1, (blk + 1) & 1);
                fill_halo(blk, kc, NBUF == 2 ? (st & 1) : 0, blk & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            W16_STAMP(0);                                    // [0] DMA issue
            const f32x4* Hb = Hs + (NBUF == 2 ? (st & 1) : 0) * W16_HS;
            f32x4 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = Hb[((j & 2) ? off_b : off_a) + i * W16_PITCH + 4 * (j & 1)];
            // V = B^T d B per channel (k-step) s: 32 adds
            float V[4][16];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float t[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = d[0][j][s] - d[2][j][s];
                    t[1][j] = d[1][j][s] + d[2][j][s];
                    t[2][j] = d[2][j][s] - d[1][j][s];
                    t[3][j] = d[1][j][s] - d[3][j][s];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    V[s][a * 4 + 0] = t[a][0] - t[a][2];
                    V[s][a * 4 + 1] = t[a][1] + t[a][2];
                    V[s][a * 4 + 2] = t[a][2] - t[a][1];
                    V[s][a * 4 + 3] = t[a][1] - t[a][3];
                }
            }
            if (NBUF == 1) {
                // every wave has its halo values in registers: the buffer is free for the next stage
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (!FIRST && st + 1 < nstage) {
                    if (kc + 1 < KC) dma_halo(blk, kc + 1, 0);
                    else dma_halo(blk + 1, 0, 0);
                }
            }
            W16_STAMP(1);                                    // [1] halo reads + transform
            // 64 * NB MFMAs; two points at a time so that consecutive MFMAs hit different accumulators
            const f32x4* Fk = Fs + (kc * NB) * 64 + lane;
#pragma unroll
            for (int p2 = 0; p2 < 8; ++p2) {
                f32x4 F[2][NB];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) F[pp][nb] = Fk[((2 * p2 + pp) * KC * NB + nb) * 64];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[2 * p2 + pp][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[pp][nb][s], V[s][2 * p2 + pp],
                                                                                        acc[2 * p2 + pp][nb], 0, 0, 0);
            }
            W16_STAMP(2);                                    // [2] filter reads + MFMA issue
            // own DMA pieces of the next stage have landed (and every LDS read of this stage has returned)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            W16_STAMP(3);                                    // [3] wait for the DMA
            if (kc == KC - 1) {
                // ---- output: Y = A^T M A + bias, activation, 16-byte stores (lane = 4 output channels of one tile) ----
                int img, by0, bx0;
                block_origin(blk, img, by0, bx0);
                const int oy = by0 + 2 * TR, ox = bx0 + 2 * TC;
                f32x4 hl[2][2];                                  // HEAD: partial logits [row][column] of the tile's pixels
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int x = 0; x < 2; ++x) hl[yy][x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int co = 16 * nb + 4 * kq;
                    f32x4 mx = {0.f, 0.f, 0.f, 0.f};
                    // one pixel COLUMN of the tile at a time (round 6): both columns at once held R[4][2] + Y[2][2] = 48 registers beside the
                    // 64 accumulators and the head's 16 partial logits - the HEAD instantiation (128-register cap) spilled 4 - 12 of
                    // them.  Every value is computed by the same operations in the same order as before.
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        f32x4 R[4];
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            R[a] = x == 0 ? acc[a * 4 + 0][nb] + acc[a * 4 + 1][nb] + acc[a * 4 + 2][nb]
                                          : acc[a * 4 + 1][nb] - acc[a * 4 + 2][nb] - acc[a * 4 + 3][nb];
                        f32x4 Y[2];
                        Y[0] = apply_act4_core(R[0] + R[1] + R[2] + bv[nb], p.act, p.alpha);
                        Y[1] = apply_act4_core(R[1] - R[2] - R[3] + bv[nb], p.act, p.alpha);
                        if (!HEAD) {             // (a fused head is always the only reader: api.hip sets head_only with head_w; launch_conv_wino16 checks)
#pragma unroll
                            for (int yy = 0; yy < 2; ++yy)
                                if (oy + yy < H && ox + x < W)
                                    __builtin_nontemporal_store(Y[yy], reinterpret_cast<f32x4*>(p.out.p + (((size_t)img * H + oy + yy) * W + ox + x) * p.out.cs + co));
                        }
                        if (HEAD) {                                  // this lane's 4 channels x 4 classes, for the column's 2 pixels
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const f32x4 w = Hw[co + r];
#pragma unroll
                                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                                    for (int c = 0; c < 4; ++c) hl[yy][x][c] = __builtin_fmaf(Y[yy][r], w[c], hl[yy][x][c]);
                            }
                        }
                        // fused MaxPooling2D(2x2, stride 2): a Winograd tile is one pooling window (even extents: checked by the caller)
#pragma unroll
                        for (int c = 0; c < 4; ++c) mx[c] = x == 0 ? fmaxf(Y[0][c], Y[1][c]) : fmaxf(mx[c], fmaxf(Y[0][c], Y[1][c]));
                        if (HEAD) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!HEAD && p.pool.p != nullptr && (oy >> 1) < p.pool.h && (ox >> 1) < p.pool.w)     // (never both: api.hip fuses a head only where no pool is)
                        *reinterpret_cast<f32x4*>(p.pool.p + (((size_t)img * p.pool.h + (oy >> 1)) * p.pool.w + (ox >> 1)) * p.pool.cs + co) = mx;
                }
                            if (HEAD) {
                    // reduce-scatter over the four lanes of the tile (lane bits 5 and 4): bit 5 keeps pixel row (kq >> 1), bit 4
                    // pixel column (kq & 1); lane kq ends up with the complete logits of pixel (kq >> 1, kq & 1)
                    const bool hi = (kq & 2) != 0, odd = (kq & 1) != 0;
                    f32x4 row[2], mine;
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float keep = hi ? hl[1][x][c] : hl[0][x][c], give = hi ? hl[0][x][c] : hl[1][x][c];
                            row[x][c] = keep + __shfl_xor(give, 32);
                        }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float keep = odd ? row[1][c] : row[0][c], give = odd ? row[0][c] : row[1][c];
                        mine[c] = keep + __shfl_xor(give, 16);
                    }
                    const f32x4 hb = *reinterpret_cast<const f32x4*>(p.head_b);
                    float l[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) l[c] = mine[c] + hb[c];
                    if (p.head_act == ECSEG_ACT_SOFTMAX) {
                        float mx = l[0];
#pragma unroll
                        for (int c = 1; c < 4; ++c) if (c < p.head_k) mx = fmaxf(mx, l[c]);
                        float sum = 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { l[c] = c < p.head_k ? expf(l[c] - mx) : 0.f; sum += l[c]; }
#pragma unroll
                        for (int c = 0; c < 4; ++c) l[c] = l[c] / sum;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) l[c] = apply_act_core(l[c], p.head_act, p.alpha);
                    }
                    const int py = oy + (kq >> 1), px = ox + (kq & 1);
                    if (py < H && px < W) {
                        float* ho = p.head_out.p + (((size_t)img * H + py) * W + px) * p.head_out.cs;
                        if (p.head_k == 4 && (p.head_out.cs & 3) == 0 && (reinterpret_cast<size_t>(p.head_out.p) & 15) == 0) *reinterpret_cast<f32x4*>(ho) = f32x4{l[0], l[1], l[2], l[3]};
                        else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (c < p.head_k) ho[c] = l[c];
                        }
                    }
                }
}
            W16_STAMP(4);                                    // [4] output stage
            asm volatile("s_barrier" ::: "memory");
            W16_STAMP(5);                                    // [5] barrier
        }
    }
#ifdef ECSEG_DIAG
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
        float* dbg = const_cast<float*>(p.zero) + 16 + wave * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = (float)stp[i];
        dbg[6] = (float)(__builtin_amdgcn_s_memtime() - tk0);
        dbg[7] = (float)nstage;
    }
#endif
#undef W16_STAMP
}

// Eligibility beyond "3x3, stride 1, pad 1, same size" (checked by the caller): 16 or 32 input and output channels, at
// least one 16 x 32 block, 16-byte aligned views.
bool conv_wino16_supported(const ConvParams& p) {
    return p.in.h == p.out.h && p.in.w == p.out.w && (p.in.c == 16 || p.in.c == 32) && (p.out.c == 16 || p.out.c == 32) &&
           p.out.h >= 16 && p.out.w >= 32 && p.in.cs % 4 == 0 && p.out.cs % 4 == 0 && p.zero != nullptr;
}

template <int KC, int NB, int NBUF, bool HEAD, bool FIRST = false>
static hipError_t launch_conv_wino16_tt(const ConvParams& p, hipStream_t s) {
    int blocks_x = (p.out.w + 31) / 32, strips_y = (p.out.h + 15) / 16;
    int bpw = 8;                                             // blocks per workgroup walk: the filter load is amortised over them
    int segs_x;
    size_t grid;
    if (p.lut != nullptr) {                                  // cropped launch: blocks_x = entries of the region list over all images
        const size_t total = (size_t)(p.n / p.per_image) * p.lut_len;
        if (total > 0x7fffffffull) return hipErrorInvalidValue;
        while (bpw > 1 && (total + bpw - 1) / bpw < 1024) bpw >>= 1;
        blocks_x = (int)total; segs_x = 1;
        grid = (total + bpw - 1) / bpw;
    } else {
        while (bpw > 1 && (size_t)p.n * strips_y * ((blocks_x + bpw - 1) / bpw) < 1024) bpw >>= 1;
        segs_x = (blocks_x + bpw - 1) / bpw;
        grid = (size_t)p.n * strips_y * segs_x;
    }
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = ((size_t)NBUF * W16_HS + (size_t)16 * KC * NB * 64 + (HEAD ? 16 * NB : 0) + (FIRST ? 2 * W16_RAW_SLOTS : 0)) * 16;
    static DeviceOnce attr_set;                              // the attribute is per device (and per template instance)
    const hipError_t ea = attr_set.run([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino16_kernel<KC, NB, NBUF, HEAD, FIRST>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((conv_wino16_kernel<KC, NB, NBUF, HEAD, FIRST>), dim3((unsigned)grid), dim3(512), lds, s, p, blocks_x, strips_y, segs_x, bpw);
    return hipGetLastError();
}

template <int KC, int NB, int NBUF>
static hipError_t launch_conv_wino16_t(const ConvParams& p, hipStream_t s) {
    return p.head_w != nullptr ? launch_conv_wino16_tt<KC, NB, NBUF, true>(p, s) : launch_conv_wino16_tt<KC, NB, NBUF, false>(p, s);
}

// The fused first layer (ConvParams::first_w): p.in is the 1-channel network input, 16 output channels, whole images (no region
// list, no need boxes), extents that are multiples of 4 pixels with 16-byte aligned rows.
bool conv_wino16_first_supported(const ConvParams& p) {
    return p.in.h == p.out.h && p.in.w == p.out.w && p.in.c == 1 && p.in.cs == 1 && (p.out.c == 16 || p.out.c == 32) && p.out.h >= 16 && p.out.w >= 32 &&
           p.out.w % 4 == 0 && p.out.cs % 4 == 0 && p.zero != nullptr && p.lut == nullptr && p.in_box == nullptr &&
           (reinterpret_cast<uintptr_t>(p.in.p) & 15) == 0;
}

hipError_t launch_conv_wino16(const ConvParams& p, hipStream_t s) {
    if (p.head_w != nullptr && (!p.head_only || p.pool.p != nullptr)) return hipErrorInvalidValue;     // (the HEAD kernels write neither the features nor a pool)
    if (p.first_w != nullptr) {
        // (the first layer has as many channels as this one reads: 16 -> 16 -> 16 or 32 -> 32 -> 32, the encoder's first pair)
        if (!conv_wino16_first_supported(p)) return hipErrorInvalidValue;
        if (p.out.c == 32) return p.head_w != nullptr ? launch_conv_wino16_tt<2, 2, 2, true, true>(p, s) : launch_conv_wino16_tt<2, 2, 2, false, true>(p, s);
        return p.head_w != nullptr ? launch_conv_wino16_tt<1, 1, 1, true, true>(p, s) : launch_conv_wino16_tt<1, 1, 1, false, true>(p, s);
    }
    // 16 -> 16: single halo buffer, two workgroups per CU (1.40 -> 1.27 / 1.27 -> 1.07 ms on the two such layers of the base-16
    // U-Net at 256 x 256, 560 windows); the other shapes need > 128 registers (a second workgroup would spill: 32 -> 16
    // measured 2.27 -> 3.71 ms) and keep the double buffer
    if (p.in.c == 16) return p.out.c == 16 ? launch_conv_wino16_t<1, 1, 1>(p, s) : launch_conv_wino16_t<1, 2, 2>(p, s);
    return p.out.c == 16 ? launch_conv_wino16_t<2, 1, 2>(p, s) : launch_conv_wino16_t<2, 2, 2>(p, s);
}

}  // namespace ecseg
