// Post-processing kernels for gfx950: stitch + quantised argmax, wavefront-level union-find connected-component
// labelling, the meta_inference clean-up schedule, component counts, the meta_overlay row and meta_preprocess.
//
// All of this is HBM/latency-bound integer work on uint8 label images (n_img, H, W) and int32 parent arrays; no MFMA.
//
// Connected components (reference: skimage.measure.label / scipy.ndimage.label, src/image_tools.py:26,42-50,66-67,115)
//   * a pixel's "key" is a small integer derived from its value through a 4-entry LUT (0 = background); two
//     neighbouring pixels are connected iff their keys are equal and non-zero, so the disjoint class masks
//     (img==1, img==2, img==3) are labelled in ONE pass;
//   * ccl_local: a workgroup labels one 64 x 32 tile in LDS.  One wavefront owns a 64-pixel row chunk; __ballot over
//     "run starts" gives every pixel the first pixel of its run without memory traffic or atomics; unions are issued
//     once per (run, upper-run) overlap instead of once per pixel pair - only the lane at the start of an overlap
//     calls unite(), the lock-free atomicMin union-find, so a root is the minimum raster index of its component;
//   * ccl_border: the same rule on the pixels along tile borders, on the global parent array; the root of a component
//     is its first pixel in raster order (the order scipy/skimage number components in, which merge_comp's skipped
//     last component depends on, src/image_tools.py:27);
//   * ccl_flatten: every pixel looks up its root; per-run (not per-pixel) atomics accumulate area / coordinate sums /
//     flag bits into the root's slot, per-wave popcounts accumulate component and pixel counts per key.
// blockIdx -> image mapping keeps all blocks of one image on one XCD (equal blockIdx % 8) so that the image's parent
// array stays in that XCD's L2.
#include <algorithm>

#include <cstdlib>

#include "common.h"
#include "device_util.h"

namespace ecseg {

typedef unsigned long long u64;

static constexpr uint32_t LUT_MULTI = 0x03020100u;   // key = value
static constexpr uint32_t LUT_NONZERO = 0xffffffffu; // key = (value != 0)
__host__ __device__ constexpr uint32_t lut_eq(int c) { return 1u << (8 * c); }                  // key = (value == c)
__host__ __device__ constexpr uint32_t lut_ne(int c) { return 0x01010101u & ~(0xffu << (8 * c)); }  // key = (value != c)
// key = (value != 0 && value != m)
__host__ __device__ constexpr uint32_t lut_nonzero_except(int m) { return 0x01010100u & ~(0xffu << (8 * m)); }

__device__ __forceinline__ int key_of(uint8_t v, uint32_t lut) {
    if (lut == LUT_NONZERO) return v != 0;
    return (lut >> ((v & 3) * 8)) & 0xff;
}

// per-image global counters.  ccl_flatten adds into one of G_SHARDS replicas (each on its own 128-B line: thousands of
// workgroups of one image adding to ONE line serialise at ~60 ns per atomic, which used to cost ~1 ms per labelling);
// reduce_g_kernel folds the replicas into replica 0, which is what every consumer reads.
enum { G_NCOMP = 0 /*[4]*/, G_NPX = 4 /*[4]*/, G_LAST_ROOT = 8, G_NLIST1 = 9, G_NLIST2 = 10,
       G_CNT0 = 12 /* [8] generic root counters */, G_OTSU_INV = 20 };
enum { NEED_NCOMP = 1, NEED_NPX = 2, NEED_LAST = 4, NEED_LISTS = 8 };
static constexpr int G_IMG = G_STRIDE * G_SHARDS;     // ints per image

// aux modes of ccl_flatten: which per-pixel bits are OR-ed into the root's flag word
enum { AUX_NONE = 0, AUX_BORDER = 1, AUX_VALUE_EQ = 2, AUX_IMAGE = 3 };
enum { STAT_AREA = 1, STAT_SUMS = 2 };

// ---------------------------------------------------------------------------------------------------------------
// geometry helpers: a block = 4 waves; wave w owns the 64-pixel chunk `cx` of rows y0 + w*ROWS .. + ROWS-1
// ---------------------------------------------------------------------------------------------------------------
static constexpr int CCL_ROWS = 8;                      // rows per wave
static constexpr int CCL_BLOCK_ROWS = 4 * CCL_ROWS;     // rows per block

struct CclGeom {
    int H, W, n_img;
    int chunks_x, strips_y, blocks_per_img;
    int img_groups;   // n_img / 8: complete groups of 8 images, one image per XCD (see decode_block)
    int affine_blocks;   // 8 * img_groups * blocks_per_img: the workgroups of those groups; the rest deal the remaining images' tiles over all XCDs
};

static CclGeom make_geom(int n_img, int H, int W) {
    CclGeom g;
    g.H = H; g.W = W; g.n_img = n_img;
    g.chunks_x = (W + 63) / 64;
    g.strips_y = (H + CCL_BLOCK_ROWS - 1) / CCL_BLOCK_ROWS;
    g.blocks_per_img = g.chunks_x * g.strips_y;
    g.img_groups = n_img / 8;
    g.affine_blocks = 8 * g.img_groups * g.blocks_per_img;
    return g;
}
static unsigned geom_grid(const CclGeom& g) { return (unsigned)(g.n_img * g.blocks_per_img); }

// block id -> (image, strip, chunk).  Complete groups of 8 images: images with equal (img % 8) share an XCD (consecutive
// workgroups go to consecutive XCDs), so the parents, statistics and owner bits of an image stay in ONE L2 (64 images: 0.135 against
// 0.146 ms per speckled image with the tiles dealt round-robin).  The images beyond the last complete group - all of them when there are
// fewer than 8 - would leave XCDs idle that way (a single image ran its 726 tiles on 32 of the 256 CUs: 67 - 86 us per ccl_local
// launch; 9 images took 2x the time of 8): their tiles are dealt round-robin over all XCDs (round 4; every cross-tile access is an
// agent-scope atomic anyway: 9 images 0.223 -> 0.174 ms per speckled image, 12 images 0.179 -> 0.165).
__device__ __forceinline__ unsigned block_in_image(const CclGeom& g) {
    const unsigned b = blockIdx.x;
    return (b < (unsigned)g.affine_blocks ? (b >> 3) : b - (unsigned)g.affine_blocks) % (unsigned)g.blocks_per_img;
}
__device__ __forceinline__ bool decode_block(const CclGeom& g, int& img, int& y0, int& cx) {
    const unsigned b = blockIdx.x;
    if (b < (unsigned)g.affine_blocks) img = (int)(((b >> 3) / g.blocks_per_img) * 8 + (b & 7u));
    else img = 8 * g.img_groups + (int)((b - (unsigned)g.affine_blocks) / (unsigned)g.blocks_per_img);
    if (img >= g.n_img) return false;
    const unsigned blk = block_in_image(g);
    cx = (int)(blk % g.chunks_x);
    y0 = (int)(blk / g.chunks_x) * CCL_BLOCK_ROWS + (int)(threadIdx.x >> 6) * CCL_ROWS;
    return true;
}

// index of this block's tile in PostWorkspace::tile_any (valid after decode_block returned true)
__device__ __forceinline__ size_t tile_index(const CclGeom& g, int img) {
    return (size_t)img * g.blocks_per_img + block_in_image(g);
}

// ---------------------------------------------------------------------------------------------------------------
// union-find
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_load(const int32_t* L, int i) {
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int32_t* L, int x) {
    int n;
    while ((n = uf_load(L, x)) != x) x = n;   // parents strictly decrease along a chain -> terminates
    return x;
}
__device__ __forceinline__ void uf_unite(int32_t* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);   // link the larger root under the smaller
        if (old == a) return;
        a = old;                                // a had just been linked elsewhere: keep merging from its old parent
    }
}

// ---- phase 1: label one 64 x 32 tile entirely in LDS ---------------------------------------------------------------
// Every workgroup labels its tile as if it were a stand-alone image: run heads by ballot, one LDS atomicMin union per
// (run, upper run) overlap, then every pixel's tile-local root is converted to a global pixel index and written out.
// Parent chains that phase 2 / ccl_flatten have to walk in global memory are thereby bounded by the number of tiles a
// component crosses, not by its height in pixels (a 1040-row background region used to cost ~1000 dependent loads).
__device__ __forceinline__ int lds_find(const volatile int* Ls, int x) {
    int n;
    while ((n = Ls[x]) != x) x = n;
    return x;
}
__device__ __forceinline__ void lds_unite(int* Ls, int a, int b) {
    for (;;) {
        a = lds_find(Ls, a);
        b = lds_find(Ls, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(Ls + a, b);
        if (old == a) return;
        a = old;
    }
}

// lane - 1 / lane + 1 of the whole wavefront in one VALU instruction (DPP wave shift; the end lane gets 0)
__device__ __forceinline__ int wave_from_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ int wave_from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }  // wave_shl:1

#ifndef ECSEG_CCL_WPE
#define ECSEG_CCL_WPE 8
#endif
template <int CONN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ECSEG_CCL_WPE, ECSEG_CCL_WPE))) void ccl_local_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint32_t lut,
                                                        int32_t* __restrict__ L_all, uint32_t* __restrict__ area_all,
                                                        u64* __restrict__ sumy_all, u64* __restrict__ sumx_all,
                                                        uint32_t* __restrict__ flag_all, int stat, int sparse,
                                                        uint8_t* __restrict__ tile_any, int allow_full,
                                                        int aux_mode, int aux_c, const uint8_t* __restrict__ aux_img, int need,
                                                        int32_t* __restrict__ G_all, uint32_t* __restrict__ own_bits) {
    __shared__ int Ls[CCL_BLOCK_ROWS * 64];
    __shared__ uint8_t Kl[4][64];                          // keys of every wave's last row (the next wave's "row above")
    // per-tile-component statistics (round 4: accumulated HERE, where a pixel's tile root is known from the LDS labelling, instead
    // of in a separate flatten pass that re-read every pixel's parent from global memory): bits 0-15 pixels of the slot
    // (<= 2048), bits 16-20 flag bits
    __shared__ uint32_t af_s[CCL_BLOCK_ROWS * 64];
    __shared__ int red[8];                                 // npx[1..3]
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    // [TP] only when STAT_SUMS: coordinate sums of the slot RELATIVE to the tile origin, packed (sum of rows << 32 | sum of
    // columns; each < 2^18 inside a 64 x 32 tile): one 64-bit LDS atomic per run
    u64* sum_s = reinterpret_cast<u64*>(dyn_smem);
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;            // block-uniform
    const size_t base = (size_t)img * g.H * g.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool want_slots = stat != 0 || aux_mode != AUX_NONE;
    if (want_slots) {
        for (int i = threadIdx.x; i < CCL_BLOCK_ROWS * 64; i += 256) {
            af_s[i] = 0u;
            if (stat & STAT_SUMS) sum_s[i] = 0ull;
        }
    }
    if (threadIdx.x < 8) red[threadIdx.x] = 0;
    const int x = cx * 64 + lane;
    const int yblk = y0 - wave * CCL_ROWS;                 // first row of the tile
    // Empty-tile pre-pass (round 4): in the class labellings of a realistic image most tiles hold no keyed pixel at all, and the
    // vote below used to come after the whole first pass (8 byte loads, DPP shifts, ballots and LDS writes per thread).  For an
    // interior tile of an 8-byte-aligned image each thread now looks at 8 pixels with ONE 8-byte load first (lane = row lane / 8,
    // columns 8 (lane % 8) ...): no keyed pixel in the workgroup -> the tile is voted empty and left after ~30 instructions.
    if (sparse && !allow_full && (g.W & 7) == 0 && cx * 64 + 64 <= g.W && yblk + CCL_BLOCK_ROWS <= g.H &&
        (reinterpret_cast<uintptr_t>(img_all) & 7) == 0) {
        const u64 w8 = *reinterpret_cast<const u64*>(img_all + base + (size_t)(y0 + (lane >> 3)) * g.W + cx * 64 + (lane & 7) * 8);
        int anyk = 0;
        if (lut == LUT_NONZERO || lut == LUT_MULTI) anyk = w8 != 0ull;       // (labels are 0..3: key != 0 <=> value != 0)
        else {
#pragma unroll
            for (int b = 0; b < 8; ++b) anyk |= key_of((uint8_t)(w8 >> (8 * b)), lut);
        }
        if (!__syncthreads_or(anyk)) {
            if (threadIdx.x == 0) tile_any[tile_index(g, img)] = 0;
            return;
        }
    }
    // Full-tile pre-pass (round 5), the mirror image for the background labellings of fill_holes (allow_full: binary keys,
    // 4-connectivity, no statistics): most tiles of a realistic image hold no pixel of the class being filled at all - every
    // pixel is keyed and the tile is ONE node.  Decided from one 8-byte load per thread; such a tile used to run the whole
    // first pass (8 byte loads, DPP shifts, ballots and LDS writes per thread) before it found out.
    if (allow_full && (g.W & 7) == 0 && cx * 64 + 64 <= g.W && yblk + CCL_BLOCK_ROWS <= g.H &&
        (reinterpret_cast<uintptr_t>(img_all) & 7) == 0) {
        const u64 w8 = *reinterpret_cast<const u64*>(img_all + base + (size_t)(y0 + (lane >> 3)) * g.W + cx * 64 + (lane & 7) * 8);
        int unkeyed = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) unkeyed |= key_of((uint8_t)(w8 >> (8 * b)), lut) == 0;
        if (!__syncthreads_or(unkeyed)) {
            if (threadIdx.x == 0) {
                tile_any[tile_index(g, img)] = 2;
                const size_t p0 = base + (size_t)yblk * g.W + cx * 64;
                L_all[p0] = yblk * g.W + cx * 64;
                const bool edge = yblk == 0 || yblk + CCL_BLOCK_ROWS >= g.H || cx == 0 || cx * 64 + 64 >= g.W;
                flag_all[p0] = (aux_mode == AUX_BORDER && edge) ? 1u : 0u;
            }
            return;
        }
    }
    const u64 upto = (2ull << lane) - 1ull;                // lanes <= this one
    // The keys of the wave's 8 rows stay in registers: the neighbours above come from the previous row's register through
    // DPP wave shifts (one VALU instruction each) instead of byte reads from an LDS copy of the tile.
    // one packed word per row keeps the kernel at 8 waves per SIMD (four arrays of 8 registers each cost three of them and 40 %
    // of the kernel's speed): bits 0-1 key, 2-7 lane of the pixel's run head, 8-15 pixel value, 16-27 tile root + 1
    uint32_t info[CCL_ROWS];
#define KEY_OF(r) ((int)(info[r] & 3u))
#define HPOS_OF(r) ((int)((info[r] >> 2) & 63u))
#define VAL_OF(r) ((int)((info[r] >> 8) & 255u))
#define TROOT_OF(r) ((int)(info[r] >> 16) - 1)
    int hole = 0;                                          // a valid pixel of this thread without a key
    int rowmask = 0;                                       // (wave-uniform) bit r: row r of this wave holds a keyed pixel
    // all eight rows' bytes first (one round trip to memory instead of eight: the wave-uniform `continue` below keeps the compiler
    // from hoisting the later rows' loads itself - the whole kernel is a chain of latencies when a single image leaves each SIMD
    // with two or three waves)
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r;
        info[r] = (y < g.H && x < g.W) ? (uint32_t)img_all[base + (size_t)y * g.W + x] << 8 : 0u;
    }
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r, li = (wave * CCL_ROWS + r) * 64 + lane;
        const bool valid = y < g.H && x < g.W;
        const uint8_t v = (uint8_t)(info[r] >> 8);
        const int key = valid ? key_of(v, lut) : 0;
        hole |= (valid && key == 0) ? 1 : 0;
        // rows without a keyed pixel (most rows of a realistic class labelling, even inside a busy tile) are skipped by every
        // later phase on a wave-uniform test: no DPP shifts, ballots or LDS traffic for them (their Ls slots are never read)
        if (__ballot(key != 0) == 0ull) { info[r] = (uint32_t)v << 8; continue; }
        rowmask |= 1 << r;
        const int kprev = wave_from_left(key);
        const bool start = key != 0 && kprev != key;       // (lane 0: kprev = 0)
        const u64 S = __ballot(start);
        const int hp = 63 - __clzll(S & upto);             // (meaningless where key == 0)
        info[r] = (uint32_t)key | ((uint32_t)(hp & 63) << 2) | ((uint32_t)v << 8);
        Ls[li] = key ? (li - lane) + hp : -1;
    }
    Kl[wave][lane] = (uint8_t)KEY_OF(CCL_ROWS - 1);
    int mine = 0;
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) mine |= KEY_OF(r);
    // sparse mode (every consumer of this labelling looks at the key before it reads a parent): a tile without a keyed
    // pixel writes nothing, the others write only their keyed pixels - the parents of background pixels stay stale
    // The vote is also left in tile_any for the later kernels of this labelling: ccl_border, ccl_flatten and count_roots
    // skip a tile without a keyed pixel before they load anything of it.
    const int any = __syncthreads_or(mine);
    // FULL tiles (allow_full: binary keys, 4-connectivity, no statistics - the background labellings of fill_holes, where
    // most tiles of a realistic image hold nothing but background): every valid pixel is keyed, so the tile is ONE
    // component whose root is its first pixel.  Only that pixel gets a parent; ccl_border / ccl_flatten / apply_fill_tile
    // stand in the tile root for any pixel of such a tile (tile_any == 2).
    const int holes = allow_full ? __syncthreads_or(hole) : 1;
    const bool full = allow_full && any && !holes;
    if (threadIdx.x == 0) tile_any[tile_index(g, img)] = (uint8_t)(full ? 2 : (any != 0));
    uint32_t* ob = own_bits + tile_index(g, img) * 64;     // owner bits of this tile: bit (li & 31) of word li >> 5
    if (!any && sparse) return;                            // (later kernels skip the tile on tile_any == 0: its owner bits are never read)
    if (full) {
        // the whole tile is one component and its first pixel the only owner; its "touches the image border" bit is the tile's
        if (threadIdx.x == 0) {
            const size_t p0 = base + (size_t)yblk * g.W + cx * 64;
            L_all[p0] = yblk * g.W + cx * 64;
            const bool edge = yblk == 0 || yblk + CCL_BLOCK_ROWS >= g.H || cx == 0 || cx * 64 + 64 >= g.W;
            flag_all[p0] = (aux_mode == AUX_BORDER && edge) ? 1u : 0u;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int ly = wave * CCL_ROWS + r, li = ly * 64 + lane;
        if (!((rowmask >> r) & 1)) continue;
        const int key = KEY_OF(r);
        const int above = r ? KEY_OF(r > 0 ? r - 1 : 0) : (wave ? (int)Kl[wave > 0 ? wave - 1 : 0][lane] : 0);
        const int al = wave_from_left(above), ar = wave_from_right(above), kl = wave_from_left(key);   // all lanes active here
        if (!key || ly == 0) continue;
        const bool left = kl == key, u0 = above == key, ul = al == key, ur = ar == key;
        if (CONN == 8) {
            if (ur && !u0) lds_unite(Ls, li, li - 63);
            if (!left) {
                if (u0) lds_unite(Ls, li, li - 64);
                else if (ul) lds_unite(Ls, li, li - 65);
            }
        } else {
            if (u0 && !(left && ul)) lds_unite(Ls, li, li - 64);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r, li = (wave * CCL_ROWS + r) * 64 + lane;
        if (!((rowmask >> r) & 1)) {
            if (!sparse && y < g.H && x < g.W) L_all[base + (size_t)y * g.W + x] = -1;
            continue;
        }
        const int key = KEY_OF(r);
        // run heads look the tile root up and leave it in their own slot (only finds run in this phase: a reader sees the
        // old parent or the root, both are on the chain); the rest of the run reads the head's slot - LDS operations of one
        // wave complete in order
        const bool head = key != 0 && HPOS_OF(r) == lane;
        if (head) {
            const int root = lds_find(Ls, li);
            Ls[li] = root;
        }
        const int rl = key ? Ls[(li - lane) + HPOS_OF(r)] : -1;
        info[r] |= (uint32_t)(rl + 1) << 16;                   // tile-local index of the pixel's tile root (+ 1; 0 = unkeyed)
        if (y < g.H && x < g.W) {
            const size_t p = base + (size_t)y * g.W + x;
            int lab = -1;
            if (key) lab = (yblk + (rl >> 6)) * g.W + cx * 64 + (rl & 63);
            if (key || !sparse) L_all[p] = lab;            // a pixel points at its TILE root for good; only tile roots are re-linked later
        }
    }
    // ---- per-run accumulation into the tile root's LDS slot (runs of equal key = runs of equal tile root) ----
    int npx[4] = {0, 0, 0, 0};
    if (want_slots || (need & NEED_NPX)) {
#pragma unroll
        for (int r = 0; r < CCL_ROWS; ++r) {
            const int y = y0 + r;
            const int key = KEY_OF(r), tr = TROOT_OF(r);
            if (need & NEED_NPX) {
#pragma unroll
                for (int k = 1; k < 4; ++k) npx[k] += __popcll(__ballot(key == k));
            }
            if (!want_slots || !((rowmask >> r) & 1)) continue;
            const u64 F = __ballot(key != 0);
            uint32_t bits = 0;
            if (aux_mode == AUX_BORDER) bits = (key && (y == 0 || y == g.H - 1 || x == 0 || x == g.W - 1)) ? 1u : 0u;
            else if (aux_mode == AUX_VALUE_EQ) bits = (key && VAL_OF(r) == aux_c) ? 1u : 0u;
            else if (aux_mode == AUX_IMAGE) bits = key ? aux_img[base + (size_t)y * g.W + x] : 0u;
            u64 B[5];
            const int nb = (aux_mode == AUX_IMAGE) ? 5 : (aux_mode == AUX_NONE ? 0 : 1);
#pragma unroll
            for (int b = 0; b < 5; ++b) B[b] = (b < nb) ? __ballot((bits >> b) & 1u) : 0ull;
            // a row without a flagged pixel has nothing to accumulate when only flags are wanted (fill_holes: rows off the image
            // border; merge_comp: rows without a class-c pixel) - wave-uniform
            if (!(stat & (STAT_AREA | STAT_SUMS)) && !(B[0] | B[1] | B[2] | B[3] | B[4])) continue;
            const bool start = key != 0 && HPOS_OF(r) == lane;
            const u64 S = __ballot(start);
            if (start) {
                const u64 above = ~((2ull << lane) - 1ull);
                const u64 stop = (S | ~F) & above;
                const int len = stop ? (__ffsll((long long)stop) - 1 - lane) : (64 - lane);
                const u64 run = (len == 64) ? ~0ull : (((1ull << len) - 1ull) << lane);
                if (stat & STAT_SUMS) {
                    const unsigned ry = (unsigned)(y - yblk), rx = (unsigned)lane, ulen = (unsigned)len;
                    atomicAdd(&sum_s[tr], ((u64)(ry * ulen) << 32) | (u64)(rx * ulen + ulen * (ulen - 1) / 2u));
                }
                uint32_t fb = 0;
#pragma unroll
                for (int b = 0; b < 5; ++b) if (B[b] & run) fb |= 1u << b;
                if (stat & (STAT_AREA | STAT_SUMS)) atomicAdd(&af_s[tr], (uint32_t)len);
                if (fb) atomicOr(&af_s[tr], fb << 16);
            }
        }
    }
    __syncthreads();
    // ---- owners (tile roots): partial statistics of the tile component into the owner's own global slots, owner bit for
    //      ccl_resolve (which walks only the owners and forwards their slots to the global roots) ----
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r, li = (wave * CCL_ROWS + r) * 64 + lane;
        if (!((rowmask >> r) & 1)) {
            if (lane == 0) { ob[2 * (wave * CCL_ROWS + r)] = 0u; ob[2 * (wave * CCL_ROWS + r) + 1] = 0u; }
            continue;
        }
        const bool own = KEY_OF(r) != 0 && TROOT_OF(r) == li;
        const u64 m = __ballot(own);
        if (lane == 0) { ob[2 * (wave * CCL_ROWS + r)] = (uint32_t)m; ob[2 * (wave * CCL_ROWS + r) + 1] = (uint32_t)(m >> 32); }
        if (own) {
            const size_t p = base + (size_t)y * g.W + x;
            const uint32_t af = want_slots ? af_s[li] : 0u;
            flag_all[p] = af >> 16;
            if (stat & STAT_AREA) area_all[p] = af & 0xffffu;
            if (stat & STAT_SUMS) {
                const u64 pk = sum_s[li], cnt = af & 0xffffu;
                sumy_all[p] = (pk >> 32) + (u64)yblk * cnt;
                sumx_all[p] = (pk & 0xffffffffull) + (u64)(cx * 64) * cnt;
            }
        }
    }
    if (need & NEED_NPX) {
        if (lane == 0) {
#pragma unroll
            for (int k = 1; k < 4; ++k) if (npx[k]) atomicAdd(&red[k], npx[k]);
        }
        __syncthreads();
        int32_t* G = G_all + (size_t)img * G_IMG + (size_t)(block_in_image(g) % G_SHARDS) * G_STRIDE;
        if (threadIdx.x >= 1 && threadIdx.x < 4 && red[threadIdx.x]) atomicAdd(G + G_NPX + threadIdx.x, red[threadIdx.x]);
    }
}
#undef KEY_OF
#undef HPOS_OF
#undef VAL_OF
#undef TROOT_OF

// ---- phase 2: unions across tile borders (global atomicMin union-find) ----------------------------------------------
// Workgroup = one tile, two waves: wave 0 takes the 64 pixels of the tile's first row (contacts with the tile above and,
// for its first pixel, with the tile to the left), wave 1 the 32 + 32 pixels of the first and last column of the other
// rows (lane = row | side << 5).  Every border pixel is one lane of ONE pass, so the dependent global loads of the
// union-find walks of a tile overlap instead of queueing row after row.
template <int CONN>
__global__ __launch_bounds__(128) void ccl_border_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint32_t lut,
                                                         int32_t* __restrict__ L_all, const uint8_t* __restrict__ tile_any) {
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const size_t ti = tile_index(g, img);
    const int ta = tile_any[ti];
    if (!ta) return;                                               // no keyed pixel in this tile: nothing to unite from here
    const size_t base = (size_t)img * g.H * g.W;
    const uint8_t* im = img_all + base;
    int32_t* L = L_all + base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ytile = y0 - wave * CCL_ROWS;                        // first row of the tile
    const int W = g.W;
    // full tiles (ccl_local, 4-connectivity only): only the tile's first pixel has a parent - it stands in for every pixel of
    // the tile; two full tiles are united once, by one lane
    const bool fullA = CONN == 4 && ta == 2;
    const int rootA = ytile * W + cx * 64;
    if (wave == 0) {
        // first row of a tile that has a tile above it
        const int y = ytile, x = cx * 64 + lane;
        if (y == 0 || x >= W) return;
        const int p = y * W + x;
        const int key = key_of(im[p], lut);
        if (!key) return;
        const bool left = x > 0 && key_of(im[p - 1], lut) == key;
        const bool fullL = CONN == 4 && lane == 0 && cx > 0 && tile_any[ti - 1] == 2;
        const bool fullU = CONN == 4 && tile_any[ti - g.chunks_x] == 2;
        const int P = fullA ? rootA : p;
        if (lane == 0 && left) uf_unite(L, P, fullL ? rootA - 64 : p - 1);
        const bool u0 = key_of(im[p - W], lut) == key;
        const bool ul = x > 0 && key_of(im[p - W - 1], lut) == key;
        const bool ur = x + 1 < W && key_of(im[p - W + 1], lut) == key;
        if (CONN == 8) {
            if (ur && !u0) uf_unite(L, p, p - W + 1);
            if (!left) {
                if (u0) uf_unite(L, p, p - W);
                else if (ul) uf_unite(L, p, p - W - 1);
            }
        } else {
            if (fullA && fullU) { if (lane == 0) uf_unite(L, rootA, rootA - CCL_BLOCK_ROWS * W); }
            else if (u0 && !(left && ul)) uf_unite(L, P, fullU ? rootA - CCL_BLOCK_ROWS * W : p - W);
        }
    } else {
        // first / last column of the rows that wave 0 does not cover (all rows of the image's first tile row)
        const int r = lane & 31, side = lane >> 5;
        const int y = ytile + r, x = cx * 64 + (side ? 63 : 0);
        if ((r == 0 && ytile > 0) || y >= g.H || x >= W) return;
        const int p = y * W + x;
        const int key = key_of(im[p], lut);
        if (!key) return;
        if (side == 0) {
            if (CONN == 4) {
                const bool fullL = cx > 0 && tile_any[ti - 1] == 2;
                if (x > 0 && key_of(im[p - 1], lut) == key) {
                    // two full tiles: one union is enough - the first row this wave handles (row 1, or row 0 in the top strip)
                    if (fullA && fullL) { if (r == (ytile > 0 ? 1 : 0)) uf_unite(L, rootA, rootA - 64); }
                    // one union per VERTICAL RUN of border contacts (round 4): when the pair above (p - W, p - W - 1) is a contact
                    // too, it is united by the lane above (or by wave 0 for the tile's first row) and both pixels hang on it
                    // vertically inside their own tiles - this union would walk two long chains of the background component
                    // to find them equal (32 such walks per tile edge were most of this kernel's time in the fill_holes labellings)
                    else if (r > 0 && key_of(im[p - W], lut) == key && key_of(im[p - W - 1], lut) == key) { }
                    else uf_unite(L, fullA ? rootA : p, fullL ? rootA - 64 : p - 1);
                }
                return;
            }
            if (x > 0 && key_of(im[p - 1], lut) == key) uf_unite(L, p, p - 1);
            if (CONN == 8 && y > 0 && x > 0 && key_of(im[p - W - 1], lut) == key) uf_unite(L, p, p - W - 1);
        } else if (CONN == 8) {                                    // diagonal contact across the vertical tile border
            if (y > 0 && x + 1 < W && key_of(im[p - W + 1], lut) == key) uf_unite(L, p, p - W + 1);
        }
    }
}
// ---- phase 3: resolve the tile roots ----------------------------------------------------------------------------------
// After ccl_local every pixel points at its tile root (the "owner" of a tile component) and the owner's global slots hold
// the component's statistics within the tile; ccl_border has linked owners across tiles.  Only the owners are touched
// here: each walks to its global root once, keeps it as its parent (a consumer reads a pixel's component as L[L[p]]: pixel
// -> owner -> global root) and forwards its slots to the root's with at most four global atomics.  No per-pixel pass: the
// int32 parent image is written once (ccl_local) and read once (the consumer).  (Rounds 1-3 re-read every pixel's parent
// here to rebuild the per-tile sums and rewrote it with the global root: ccl_flatten, 25 % of the post-processing time.)
// aux bits OR-ed into flag[root] (set per run by ccl_local): AUX_BORDER bit0 = touches the image border, AUX_VALUE_EQ
// bit0 = holds a pixel with value == aux_c, AUX_IMAGE bits of aux_img[p].
__global__ __launch_bounds__(256) void ccl_resolve_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint32_t lut,
                                                          int32_t* __restrict__ L_all, uint32_t* __restrict__ area_all,
                                                          u64* __restrict__ sumy_all, u64* __restrict__ sumx_all,
                                                          uint32_t* __restrict__ flag_all, int32_t* __restrict__ G_all,
                                                          int stat, int need,
                                                          int32_t* __restrict__ list1, int32_t* __restrict__ list2, size_t list_cap,
                                                          const uint8_t* __restrict__ tile_any, const uint32_t* __restrict__ own_bits) {
    constexpr int TP = CCL_BLOCK_ROWS * 64;                    // pixels per tile
    __shared__ int red[16];                                    // ncomp[1..3] at [0..2], last root [8], list counts [9..10], list bases [11..12], owners [13]
    __shared__ uint16_t own_s[TP];                             // compacted list of the owners
    __shared__ int16_t lsl_s[TP];                              // NEED_LISTS: (class << 12 | index inside the block's range) of owner k, -1 = none
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;                 // block-uniform
    const int tid = threadIdx.x;
    const size_t ti = tile_index(g, img);
    const int ta = tile_any[ti];
    if (!ta) return;                                           // ccl_local's vote: no keyed pixel, no owner
    const size_t base = (size_t)img * g.H * g.W;
    const uint8_t* im = img_all + base;
    int32_t* L = L_all + base;
    const int W = g.W;
    const int yblk = y0 - (tid >> 6) * CCL_ROWS;
    if (ta == 2) {
        // full tile (fill_holes background labelling: no statistics, no counters): its first pixel is the only owner
        if (tid == 0) {
            const int p0 = yblk * W + cx * 64;
            const int gr = uf_find(L, p0);
            if (gr != p0) {
                L[p0] = gr;
                const uint32_t f = flag_all[base + p0];
                if (f) atomicOr(flag_all + base + gr, f);
            }
        }
        return;
    }
    if (tid < 16) red[tid] = 0;
    __syncthreads();
    // owners -> compact list (thread = 8 consecutive pixels of the tile = one byte of the owner bits)
    {
        const uint32_t bits = (own_bits[ti * 64 + (tid >> 2)] >> ((tid & 3) * 8)) & 0xffu;
        if (bits) {
            int at = atomicAdd(&red[13], __popc(bits));
            uint32_t b = bits;
            while (b) {
                const int k = __ffs((int)b) - 1;
                b &= b - 1;
                own_s[at++] = (uint16_t)(tid * 8 + k);
            }
        }
    }
    __syncthreads();
    const int n_own = red[13];
    int32_t* G = G_all + (size_t)img * G_IMG + (size_t)(block_in_image(g) % G_SHARDS) * G_STRIDE;
    int ncomp[4] = {0, 0, 0, 0};
    int last_root = 0;
    // thread k takes owner k: the chain walks and global atomics of all owners of a tile are in flight together
    for (int k = tid; k < n_own; k += 256) {
        const int li = own_s[k];
        const int p = (yblk + (li >> 6)) * W + cx * 64 + (li & 63);
        const int gr = uf_find(L, p);
        int16_t ls = -1;
        if (gr != p) {
            L[p] = gr;                                         // (a concurrent walk through p sees the old parent or the root: both on the chain)
            if (stat & STAT_AREA) atomicAdd(area_all + base + gr, area_all[base + p]);
            if (stat & STAT_SUMS) {
                atomicAdd(sumy_all + base + gr, sumy_all[base + p]);
                atomicAdd(sumx_all + base + gr, sumx_all[base + p]);
            }
            const uint32_t f = flag_all[base + p];
            if (f) atomicOr(flag_all + base + gr, f);
        } else {                                               // a global root lives in this tile
            const uint8_t v = im[p];
            if (need & NEED_NCOMP) ncomp[key_of(v, lut) & 3] += 1;
            last_root = max(last_root, p + 1);
            // roots of class 1 / 2 go to the per-image lists (nucleus test)
            if ((need & NEED_LISTS) && (v == 1 || v == 2)) ls = (int16_t)((v << 12) | atomicAdd(&red[8 + v], 1));
        }
        if (need & NEED_LISTS) lsl_s[k] = ls;
    }
    if (!(need & (NEED_NCOMP | NEED_LAST | NEED_LISTS))) return;
    __syncthreads();
    if (need & NEED_LISTS) {
        // one global atomic per workgroup and class reserves the block's range in the image's list
        if (tid < 2 && red[9 + tid]) red[11 + tid] = atomicAdd(G_all + (size_t)img * G_IMG + (tid == 0 ? G_NLIST1 : G_NLIST2), red[9 + tid]);
        __syncthreads();
        for (int k = tid; k < n_own; k += 256) {
            const int ls = lsl_s[k];
            if (ls < 0) continue;
            const int v = ls >> 12, idx = ls & 0xfff, li = own_s[k];
            const int p = (yblk + (li >> 6)) * W + cx * 64 + (li & 63);
            if (v == 1) list1[(size_t)img * list_cap + red[11] + idx] = p;
            else list2[((size_t)img * list_cap + red[12] + idx) * 4] = p;      // first word of the chromosome's centroid slot
        }
    }
    // per-image counters: registers -> LDS -> one global atomic per workgroup and counter
    if (need & NEED_NCOMP) {
#pragma unroll
        for (int k = 1; k < 4; ++k) if (ncomp[k]) atomicAdd(&red[k - 1], ncomp[k]);
    }
    if ((need & NEED_LAST) && last_root) atomicMax(&red[8], last_root);
    __syncthreads();
    if (tid < 3) { if (red[tid]) atomicAdd(G + G_NCOMP + 1 + tid, red[tid]); }
    else if (tid == 8) { if (red[8]) atomicMax(G + G_LAST_ROOT, red[8]); }
}

// count_cc needs no resolve: after ccl_local + ccl_border a component's root is the one OWNER that still points at itself, so
// the components per key are counted from the owner bits (256 B per tile) and the owners' parents alone; the pixels per key were
// counted by ccl_local (NEED_NPX).  (Rounds 2-3 re-read image and parents of every pixel for this: 220 us per 64 speckled images.)
__global__ __launch_bounds__(256) void count_roots_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint32_t lut,
                                                          const int32_t* __restrict__ L_all, int32_t* __restrict__ G_all,
                                                          const uint8_t* __restrict__ tile_any, const uint32_t* __restrict__ own_bits) {
    __shared__ int red[4];
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const size_t ti = tile_index(g, img);
    const int tid = threadIdx.x;
    if (tile_any[ti]) {                                        // block-uniform
        if (tid < 4) red[tid] = 0;
        __syncthreads();
        const size_t base = (size_t)img * g.H * g.W;
        const int yblk = y0 - (tid >> 6) * CCL_ROWS;
        // thread = 8 consecutive pixels of the tile; a FULL tile (tile_any == 2: ccl_local wrote no owner bits for it) has one owner,
        // its first pixel - no pass combines allow_full with counting today, the case is handled all the same
        uint32_t bits = tile_any[ti] == 2 ? (tid == 0 ? 1u : 0u) : (own_bits[ti * 64 + (tid >> 2)] >> ((tid & 3) * 8)) & 0xffu;
        int nc[4] = {0, 0, 0, 0};
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int li = tid * 8 + k;
            const int p = (yblk + (li >> 6)) * g.W + cx * 64 + (li & 63);
            if (L_all[base + p] == p) nc[key_of(img_all[base + p], lut) & 3] += 1;
        }
#pragma unroll
        for (int k = 1; k < 4; ++k) if (nc[k]) atomicAdd(&red[k], nc[k]);
        __syncthreads();
        int32_t* G = G_all + (size_t)img * G_IMG + (size_t)(block_in_image(g) % G_SHARDS) * G_STRIDE;
        if (tid >= 1 && tid < 4 && red[tid]) atomicAdd(G + G_NCOMP + tid, red[tid]);
    }
}

// fold the G_SHARDS replicas of every image's counter block into replica 0
__global__ void reduce_g_kernel(int32_t* __restrict__ G_all, int n_img) {
    const int im = blockIdx.x, t = threadIdx.x;          // blockDim = G_STRIDE
    if (im >= n_img) return;
    int32_t* G = G_all + (size_t)im * G_IMG;
    int v = G[t];
    for (int sh = 1; sh < G_SHARDS; ++sh) {
        const int w = G[sh * G_STRIDE + t];
        v = (t == G_LAST_ROOT) ? max(v, w) : v + w;
        G[sh * G_STRIDE + t] = 0;                         // folded: a second fold (count_flagged_owner_roots adds later) adds nothing twice
    }
    G[t] = v;
}

// consumers of a pass with its own counter block fold the replicas themselves (one thread, once per workgroup)
__device__ __forceinline__ int g_fold_sum(const int32_t* G_img, int idx) {
    int v = 0;
    for (int sh = 0; sh < G_SHARDS; ++sh) v += G_img[sh * G_STRIDE + idx];
    return v;
}
__device__ __forceinline__ int g_fold_max(const int32_t* G_img, int idx) {
    int v = 0;
    for (int sh = 0; sh < G_SHARDS; ++sh) v = max(v, G_img[sh * G_STRIDE + idx]);
    return v;
}

__global__ void zero_g_kernel(int32_t* G, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) G[i] = 0;
}

struct CclPass {
    const uint8_t* key_img;
    uint32_t lut;
    int conn;
    int stat;
    int aux_mode, aux_c;
    const uint8_t* aux_img;
    int need;      // NEED_* counters this pass has to produce
    bool count_only = false;   // only NEED_NCOMP | NEED_NPX are wanted: count roots instead of flattening
    bool sparse = false;       // every consumer checks the key before it reads a parent: background parents are not written
    bool allow_full = false;   // binary keys, 4-connectivity, no statistics / counters: tiles without an unkeyed pixel are one node (tile_any == 2)
    int32_t* list1 = nullptr;  // NEED_LISTS: per-image lists of class-1 roots / class-2 roots (first word of 16-byte slots)
    int32_t* list2 = nullptr;
    size_t list_cap = 0;
    // counter block of this pass (null: ws.g, zeroed by a launch of its own and folded by reduce_g_kernel).  run_meta_inference
    // gives every labelling its own block, zeroes them all with ONE memset and lets the consumers fold the replicas themselves
    // (g_fold_*): two launches less per labelling
    int32_t* g = nullptr;
    // (round 5 tried to publish a count_only pass's result from the last workgroup of every image to finish - a ticket counter per
    // image: 726 workgroups x 64 images adding to ONE line per image made count_roots_kernel 25x slower (20 -> 546 us); the
    // per-image gather launch stays)
};

static hipError_t run_ccl_pass(PostWorkspace& ws, const CclGeom& g, const CclPass& c, hipStream_t s) {
    const int ng = g.n_img * G_IMG;
    int32_t* const G = c.g ? c.g : ws.g;
    // (a pass that produces no counters - the fill_holes labellings - touches none: no zeroing launch)
    if (c.need && !c.g) hipLaunchKernelGGL(zero_g_kernel, dim3((ng + 255) / 256), dim3(256), 0, s, G, ng);
    const unsigned grid = geom_grid(g);
    const bool slots = !c.count_only;                          // count_only: counts per key only - no statistics, no owners' slots
    const int stat = slots ? c.stat : 0, aux_mode = slots ? c.aux_mode : AUX_NONE;
    const int need_local = c.need & NEED_NPX;                  // pixels per key: counted where the keys are in registers
    const size_t dyn = (stat & STAT_SUMS) ? (size_t)CCL_BLOCK_ROWS * 64 * 8 : 0;
    if (c.conn == 8) {
        hipLaunchKernelGGL(ccl_local_kernel<8>, dim3(grid), dim3(256), dyn, s, g, c.key_img, c.lut, ws.L, ws.area, ws.sumy,
                           ws.sumx, ws.flag, stat, c.sparse ? 1 : 0, ws.tile_any, 0, aux_mode, c.aux_c, c.aux_img, need_local,
                           G, ws.own_bits);
        hipLaunchKernelGGL(ccl_border_kernel<8>, dim3(grid), dim3(128), 0, s, g, c.key_img, c.lut, ws.L, ws.tile_any);
    } else {
        hipLaunchKernelGGL(ccl_local_kernel<4>, dim3(grid), dim3(256), dyn, s, g, c.key_img, c.lut, ws.L, ws.area, ws.sumy,
                           ws.sumx, ws.flag, stat, c.sparse ? 1 : 0, ws.tile_any,
                           (c.allow_full && c.stat == 0 && c.need == 0 && !c.count_only) ? 1 : 0, aux_mode, c.aux_c, c.aux_img,
                           need_local, G, ws.own_bits);
        hipLaunchKernelGGL(ccl_border_kernel<4>, dim3(grid), dim3(128), 0, s, g, c.key_img, c.lut, ws.L, ws.tile_any);
    }
    if (c.count_only) {        // counts per key only: no per-pixel roots, no statistics
        hipLaunchKernelGGL(count_roots_kernel, dim3(grid), dim3(256), 0, s, g, c.key_img, c.lut, ws.L, G, ws.tile_any, ws.own_bits);
        if (!c.g) hipLaunchKernelGGL(reduce_g_kernel, dim3(g.n_img), dim3(G_STRIDE), 0, s, G, g.n_img);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ccl_resolve_kernel, dim3(grid), dim3(256), 0, s, g, c.key_img, c.lut, ws.L, ws.area, ws.sumy, ws.sumx,
                       ws.flag, G, c.stat, c.need, c.list1, c.list2, c.list_cap, ws.tile_any, ws.own_bits);
    if (c.need && !c.g) hipLaunchKernelGGL(reduce_g_kernel, dim3(g.n_img), dim3(G_STRIDE), 0, s, G, g.n_img);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// per-pixel "apply" kernels of the meta_inference schedule (one thread per pixel, grid-stride over the batch)
// ---------------------------------------------------------------------------------------------------------------
static unsigned px_grid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 256 * 64) b = 256 * 64;
    return (unsigned)(b ? b : 1);
}
#define PX_LOOP(total) for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (total); t += (size_t)gridDim.x * blockDim.x)
// The same over a batch with the image index in blockIdx.y (grid from img_grid): no 64-bit division per pixel to find the
// image a flat index belongs to.  Defines im (image), ib (its first flat index) and t (flat index) for the body.
#define IMG_PX_LOOP(n_img, px)                                                      \
    for (int im = blockIdx.y; im < (n_img); im += gridDim.y)                        \
        for (size_t ib = (size_t)im * (px), q_ = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = ib + q_; q_ < (px); \
             q_ += (size_t)gridDim.x * blockDim.x, t = ib + q_)
static dim3 img_grid(size_t px, int n_img) {
    size_t bx = (px + 255) / 256;
    const size_t cap = (size_t)(256 * 64) / (size_t)(n_img > 0 ? (n_img < 16384 ? n_img : 16384) : 1);   // ~16 k blocks in all, several pixels per thread
    if (bx > cap) bx = cap;
    return dim3((unsigned)(bx ? bx : 1), (unsigned)(n_img < 65535 ? (n_img > 0 ? n_img : 1) : 65535));
}

// fill_holes (src/image_tools.py:36-39): pixels != c whose 4-connected background component does not reach the border
__global__ __launch_bounds__(256) void apply_fill_kernel(uint8_t* __restrict__ img, const int32_t* __restrict__ L,
                                                         const uint32_t* __restrict__ flag, int n_img, size_t px, int c, uint32_t lut) {
    IMG_PX_LOOP(n_img, px) {
        if (!key_of(img[t], lut)) continue;                    // not part of the labelled background: its parent is stale
        const int r = L[ib + L[t]];                            // pixel -> owner (tile root) -> global root
        if (!(flag[ib + r] & 1u)) img[t] = (uint8_t)c;
    }
}

// The same per labelling tile (fill passes run with full-tile nodes): a full tile asks once - its root's component either
// reaches the image border (nothing to do) or the whole tile lies inside a hole; a tile without a keyed pixel has nothing to
// fill; the others go pixel by pixel.
__global__ __launch_bounds__(256) void apply_fill_tile_kernel(CclGeom g, uint8_t* __restrict__ img_all, const int32_t* __restrict__ L_all,
                                                              const uint32_t* __restrict__ flag_all, int c, uint32_t lut,
                                                              const uint8_t* __restrict__ tile_any, const uint32_t* __restrict__ own_bits) {
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const size_t ti = tile_index(g, img);
    const int ta = tile_any[ti];
    if (!ta) return;
    const size_t base = (size_t)img * g.H * g.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = cx * 64 + lane;
    if (ta == 1) {
        // Holes are rare: when every component of the tile (its owners, from the owner bits; their parents are the global roots
        // since ccl_resolve) reaches the image border there is nothing to fill here - decided from a handful of look-ups
        // instead of two gathers per pixel (round 4).
        const int yblk = y0 - wave * CCL_ROWS;
        uint32_t bits = (own_bits[ti * 64 + (threadIdx.x >> 2)] >> ((threadIdx.x & 3) * 8)) & 0xffu;
        int open_all = 1;
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int li = (int)threadIdx.x * 8 + k;
            const int p = (yblk + (li >> 6)) * g.W + cx * 64 + (li & 63);
            if (!(flag_all[base + L_all[base + p]] & 1u)) open_all = 0;
        }
        if (__syncthreads_and(open_all)) return;
    }
    bool fill_all = false;
    if (ta == 2) {
        const int p0 = (y0 - wave * CCL_ROWS) * g.W + cx * 64;
        const int r = L_all[base + L_all[base + p0]];          // (the tile's first pixel is its own owner: -> global root)
        if (flag_all[base + r] & 1u) return;                   // background connected to the image border
        fill_all = true;
    }
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= g.H || x >= g.W) continue;
        const size_t t = base + (size_t)y * g.W + x;
        if (fill_all) { img_all[t] = (uint8_t)c; continue; }
        if (!key_of(img_all[t], lut)) continue;
        const int root = L_all[base + L_all[t]];
        if (!(flag_all[base + root] & 1u)) img_all[t] = (uint8_t)c;
    }
}

// The class-2 root indices that ccl_flatten appended (first word of every 16-byte slot) become regionprops centroids
// (mean of the pixel coordinates, float64) in place.
__global__ __launch_bounds__(256) void centroids_kernel(const uint32_t* __restrict__ area, const u64* __restrict__ sumy,
                                                        const u64* __restrict__ sumx, const int32_t* __restrict__ G_all,
                                                        double2* __restrict__ list2, size_t px, size_t cap) {
    const int im = blockIdx.y;
    const int n2 = G_all[(size_t)im * G_IMG + G_NLIST2];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n2; k += gridDim.x * blockDim.x) {
        double2* slot = list2 + (size_t)im * cap + k;
        const int root = *reinterpret_cast<const int32_t*>(slot);
        const size_t t = (size_t)im * px + root;
        const double a = (double)area[t];
        *slot = make_double2((double)sumy[t] / a, (double)sumx[t] / a);
    }
}

// which of the two nucleus-test kernels takes an image (see bin_centroids_kernel below)
__device__ __forceinline__ bool nucleus_binned_ok(int n2, int H, int W, size_t binned_cap) {
    return H <= NUCLEUS_BIN_EXTENT && W <= NUCLEUS_BIN_EXTENT && (size_t)n2 <= binned_cap;
}

// Nucleus-in-metaphase test (src/image_tools.py:72-81): more than five chromosome centroids strictly inside each of
// the four 70-px half bands -> the nucleus is erased.  One wavefront per nucleus (grid-stride), lanes stride over the
// chromosome list; the scan stops as soon as all four counts exceed the threshold.
__global__ __launch_bounds__(256) void nucleus_test_kernel(const uint32_t* __restrict__ area, const u64* __restrict__ sumy,
                                                           const u64* __restrict__ sumx, const int32_t* __restrict__ G_all,
                                                           const int32_t* __restrict__ list1, const double2* __restrict__ list2,
                                                           uint32_t* __restrict__ flag, size_t px, size_t cap,
                                                           int blocks_per_img, double v, int min_count, int H, int W,
                                                           size_t binned_cap) {
    const int im = blockIdx.x / blocks_per_img, j = blockIdx.x % blocks_per_img;
    const int32_t* G = G_all + (size_t)im * G_IMG;
    const int n1 = G[G_NLIST1], n2 = G[G_NLIST2];
    if (nucleus_binned_ok(n2, H, W, binned_cap)) return;            // handled by nucleus_test_binned_kernel
    const int lane = threadIdx.x & 63;
    const int wave = j * 4 + (threadIdx.x >> 6), nwaves = blocks_per_img * 4;
    for (int k = wave; k < n1; k += nwaves) {                       // wave-uniform loop
        const int root = list1[(size_t)im * cap + k];
        const size_t slot = (size_t)im * px + root;
        const double a = (double)area[slot];
        const double ny = (double)sumy[slot] / a, nx = (double)sumx[slot] / a;
        int cl = 0, cr = 0, cb = 0, ct = 0;
        for (int c0 = 0; c0 < n2; c0 += 64) {
            const int c = c0 + lane;
            bool l = false, r = false, b = false, tp = false;
            if (c < n2) {
                const double2 cc = list2[(size_t)im * cap + c];   // (cy, cx)
                l = (cc.y > nx) && (cc.y < nx + v);
                r = (cc.y < nx) && (cc.y > nx - v);
                b = (cc.x < ny) && (cc.x > ny - v);
                tp = (cc.x > ny) && (cc.x < ny + v);
            }
            cl += __popcll(__ballot(l)); cr += __popcll(__ballot(r));
            cb += __popcll(__ballot(b)); ct += __popcll(__ballot(tp));
            if (cl > min_count && cr > min_count && cb > min_count && ct > min_count) break;
        }
        if (lane == 0 && cl > min_count && cr > min_count && cb > min_count && ct > min_count) flag[slot] |= 2u;
    }
}

// The four band tests only look at ONE coordinate each (left / right: the centroid's x, bottom / top: its y), so a count is
// "how many coordinates lie in an open interval".  bin_centroids_kernel groups the coordinates of an image by integer bin
// (counting sort: LDS histogram, scan, scatter; one workgroup per image and axis); a query then takes whole bins from the
// prefix array and compares only the entries of the two end bins: O(n1 + n2) per image for any reasonable spread instead of
// the n1 * n2 pair tests of nucleus_test_kernel (the speckled output of a random-weight base-16 model - 10^4..10^5 nuclei and
// chromosomes per image - spent 0.37 ms per image there).  The comparisons are the reference's own (strict, on the same
// doubles).  Images wider / taller than NUCLEUS_BIN_EXTENT or with more chromosomes than `binned` holds go to the pair test.
// `converted` == 0: the list still holds the roots (first word of every slot, as ccl_resolve left them) and the centroid is
// computed here from the root's area / coordinate sums - the same float64 quotients centroids_kernel would have stored.
__device__ __forceinline__ double list_centroid(const double2* __restrict__ src, int i, int axis, int converted,
                                                const uint32_t* __restrict__ area, const u64* __restrict__ sumy,
                                                const u64* __restrict__ sumx, size_t ib) {
    if (converted) return axis ? src[i].y : src[i].x;
    const size_t t = ib + (size_t)*reinterpret_cast<const int32_t*>(src + i);
    return (double)(axis ? sumx[t] : sumy[t]) / (double)area[t];
}

__global__ __launch_bounds__(1024) void bin_centroids_kernel(const int32_t* __restrict__ G_all, const double2* __restrict__ list2,
                                                             double* __restrict__ binned, int32_t* __restrict__ binstart,
                                                             size_t cap, size_t binned_cap, int H, int W,
                                                             const uint32_t* __restrict__ area, const u64* __restrict__ sumy,
                                                             const u64* __restrict__ sumx, size_t px, int converted) {
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    int* hist = reinterpret_cast<int*>(dyn_smem);                        // [E + 1]: counts, then bin starts, then cursors
    __shared__ int part[1024];
    const int im = blockIdx.y, axis = blockIdx.x;                       // axis 0: y (double2.x), 1: x (double2.y)
    const int n2 = G_all[(size_t)im * G_IMG + G_NLIST2];
    if (n2 <= 0 || !nucleus_binned_ok(n2, H, W, binned_cap)) return;
    const int E = axis ? W : H;
    const int tid = threadIdx.x;
    for (int i = tid; i <= E; i += 1024) hist[i] = 0;
    __syncthreads();
    const double2* src = list2 + (size_t)im * cap;
    const size_t ib = (size_t)im * px;
    for (int i = tid; i < n2; i += 1024) {
        const double c = list_centroid(src, i, axis, converted, area, sumy, sumx, ib);
        int b = (int)c;                                                  // centroids are means of pixel coordinates: 0 <= c <= E - 1
        b = b < 0 ? 0 : (b > E - 1 ? E - 1 : b);
        atomicAdd(&hist[b], 1);
    }
    __syncthreads();
    // exclusive scan of hist[0..E]: serial chunks per thread + a scan of the 1024 chunk sums
    const int chunk = (E + 1 + 1023) / 1024;
    const int lo = tid * chunk, hi = min(lo + chunk, E + 1);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;                                           // exclusive prefix of this thread's chunk
    int32_t* st = binstart + ((size_t)im * 2 + axis) * (NUCLEUS_BIN_EXTENT + 2);
    for (int i = lo; i < hi; ++i) { const int c = hist[i]; hist[i] = run; st[i] = run; run += c; }
    __syncthreads();
    double* out = binned + ((size_t)im * 2 + axis) * binned_cap;
    for (int i = tid; i < n2; i += 1024) {
        const double c = list_centroid(src, i, axis, converted, area, sumy, sumx, ib);
        int b = (int)c;
        b = b < 0 ? 0 : (b > E - 1 ? E - 1 : b);
        out[atomicAdd(&hist[b], 1)] = c;
    }
}

// #{c in the binned coordinates : lo < c < hi}
__device__ __forceinline__ int count_open_interval(const double* __restrict__ c, const int32_t* __restrict__ st, int E, double lo,
                                                   double hi) {
    if (!(lo < hi)) return 0;
    // floor() as integers, clamped to [-1, E]: bins strictly between the two end bins lie wholly inside the interval
    const double fl_d = floor(lo), fh_d = floor(hi);
    const int fl = fl_d < -1.0 ? -1 : (fl_d > (double)E ? E : (int)fl_d);
    const int fh = fh_d < -1.0 ? -1 : (fh_d > (double)E ? E : (int)fh_d);
    int n = 0;
    if (fh > fl + 1) n = st[fh > E ? E : fh] - st[fl + 1];
    if (fl >= 0 && fl < E)
        for (int i = st[fl]; i < st[fl + 1]; ++i) n += (c[i] > lo && c[i] < hi) ? 1 : 0;
    if (fh != fl && fh >= 0 && fh < E)
        for (int i = st[fh]; i < st[fh + 1]; ++i) n += (c[i] > lo && c[i] < hi) ? 1 : 0;
    return n;
}

// thread = nucleus: the four counts of nucleus_test_kernel from the binned coordinates
__global__ __launch_bounds__(256) void nucleus_test_binned_kernel(const uint32_t* __restrict__ area, const u64* __restrict__ sumy,
                                                                  const u64* __restrict__ sumx, const int32_t* __restrict__ G_all,
                                                                  const int32_t* __restrict__ list1, const double* __restrict__ binned,
                                                                  const int32_t* __restrict__ binstart, uint32_t* __restrict__ flag,
                                                                  size_t px, size_t cap, size_t binned_cap, int H, int W, double v,
                                                                  int min_count) {
    const int im = blockIdx.y;
    const int32_t* G = G_all + (size_t)im * G_IMG;
    const int n1 = G[G_NLIST1], n2 = G[G_NLIST2];
    if (n2 <= min_count || !nucleus_binned_ok(n2, H, W, binned_cap)) return;   // no band can hold more than min_count / pair test
    const double* cy = binned + (size_t)im * 2 * binned_cap;
    const double* cx = cy + binned_cap;
    const int32_t* sty = binstart + (size_t)im * 2 * (NUCLEUS_BIN_EXTENT + 2);
    const int32_t* stx = sty + (NUCLEUS_BIN_EXTENT + 2);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n1; k += gridDim.x * blockDim.x) {
        const int root = list1[(size_t)im * cap + k];
        const size_t slot = (size_t)im * px + root;
        const double a = (double)area[slot];
        const double ny = (double)sumy[slot] / a, nx = (double)sumx[slot] / a;
        if (count_open_interval(cx, stx, W, nx, nx + v) <= min_count) continue;     // left:   nx < cx < nx + v
        if (count_open_interval(cx, stx, W, nx - v, nx) <= min_count) continue;     // right:  nx - v < cx < nx
        if (count_open_interval(cy, sty, H, ny - v, ny) <= min_count) continue;     // bottom: ny - v < cy < ny
        if (count_open_interval(cy, sty, H, ny, ny + v) <= min_count) continue;     // top:    ny < cy < ny + v
        flag[slot] |= 2u;
    }
}

__global__ __launch_bounds__(256) void apply_nucleus_kill_kernel(uint8_t* __restrict__ img, const int32_t* __restrict__ L,
                                                                 const uint32_t* __restrict__ flag, int n_img, size_t px) {
    IMG_PX_LOOP(n_img, px) {
        if (img[t] != 1) continue;
        const int r = L[ib + L[t]];
        if (flag[ib + r] & 2u) img[t] = 0;
    }
}

// The nucleus kill, four pixels per thread (px % 4 == 0, 4-byte-aligned images): one 32-bit load decides for four pixels whether
// anything is a nucleus at all - on a realistic label map 94 % of the pixels are background.
__global__ __launch_bounds__(256) void nucleus_kill4_kernel(uint8_t* __restrict__ img, const int32_t* __restrict__ L,
                                                            const uint32_t* __restrict__ flag, int n_img, size_t px4) {
    IMG_PX_LOOP(n_img, px4) {                                  // t, ib: in 4-pixel groups
        const size_t p0 = t * 4, ibp = ib * 4;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(img + p0);
        uint32_t o = w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (((w >> (8 * k)) & 0xffu) != 1u) continue;
            const int r = L[ibp + L[p0 + k]];
            if (flag[ibp + r] & 2u) o &= ~(0xffu << (8 * k));
        }
        if (o != w) *reinterpret_cast<uint32_t*>(img + p0) = o;
    }
}

// fold != 0: the pass had a counter block of its own whose replicas nobody has folded yet
__global__ void gather_counts_kernel(const int32_t* __restrict__ G_all, int n_img, int key, int32_t* __restrict__ n_out,
                                     long long* __restrict__ px_out, long long full_px, int fold) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    const int32_t* G = G_all + (size_t)i * G_IMG;
    const int n = fold ? g_fold_sum(G, G_NCOMP + key) : G[G_NCOMP + key];
    const long long px = fold ? g_fold_sum(G, G_NPX + key) : G[G_NPX + key];
    if (n_out) n_out[i] = n;
    // count_cc's pixel total (src/image_tools.py:116-119) is float 0.0 when nothing is counted: no component, or no
    // background label for np.unique(...)[1:] to drop (mask without a single zero pixel)
    if (px_out) px_out[i] = (n == 0 || px == full_px) ? -1 : px;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused tile kernels of the clean-up schedule (round 5).  Rounds 1-4 ran every step of the schedule as a pass of its own over
// the whole batch - applier (gathers through the parent image), then each 3x3 stencil: 14 launches and as many round trips
// of the label images through HBM.  Here a workgroup owns one 64 x 32 labelling tile (same blockIdx -> tile map as the CCL
// kernels, so the parents / flags it gathers are in its XCD's L2), evaluates the applier on the tile plus the halo the
// stencils behind it need, keeps the intermediate images in LDS and writes only the final one:
//   thresh_band_kernel    size_thresh applier (src/image_tools.py:41-59) + ecDNA band removal (:64)                halo 1
//   merge_open_kernel     merge_comp applier (:19-30) + grey erosion + grey dilation / combine (:31-32)            halo 2
//                         FINAL: + the last ecDNA dilation (:83)                                                    halo 3
// Halo columns are staged in whole 4-pixel words (4 columns either side); W % 4 == 0 reads the images with 32-bit loads and
// skips the gathers of a word without a labelled pixel at once (94 % of a realistic label map is background).
// ---------------------------------------------------------------------------------------------------------------
constexpr int FT_HX = 4;                     // halo columns staged either side (a whole word)
constexpr int FT_HW = 64 + 2 * FT_HX;        // LDS row pitch in pixels (72 = 18 words)

template <bool W4>
__device__ __forceinline__ uint32_t ft_load_word(const uint8_t* __restrict__ im, size_t p, int x, int W) {
    if (W4) return *reinterpret_cast<const uint32_t*>(im + p);
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (x + k >= 0 && x + k < W) w |= (uint32_t)im[p + k] << (8 * k);
    return w;
}

template <bool W4>
__global__ __launch_bounds__(256) void thresh_band_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint8_t* __restrict__ out_all,
                                                          const int32_t* __restrict__ L_all, const uint32_t* __restrict__ area_all,
                                                          const int32_t* __restrict__ G_all, int ec_thresh) {
    constexpr int R = 1, HH = CCL_BLOCK_ROWS + 2 * R;
    __shared__ __attribute__((aligned(16))) uint8_t T[HH * FT_HW];
    __shared__ long long gs[4];
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int yblk = y0 - wave * CCL_ROWS, x0 = cx * 64;
    const int H = g.H, W = g.W;
    const size_t base = (size_t)img * H * W;
    if (tid < 4) gs[tid] = g_fold_sum(G_all + (size_t)img * G_IMG, tid == 0 ? G_NCOMP + 2 : tid == 1 ? G_NPX + 2 : tid == 2 ? G_NCOMP + 3 : G_NPX + 3);
    __syncthreads();
    // stage 1: thresholded labels of the tile + halo
    for (int i = tid; i < HH * (FT_HW / 4); i += 256) {
        const int hy = i / (FT_HW / 4), wx = i - hy * (FT_HW / 4);
        const int y = yblk - R + hy, x = x0 - FT_HX + wx * 4;
        uint32_t o = 0;
        if (y >= 0 && y < H && x + 3 >= 0 && x < W) {
            const size_t p0 = (size_t)y * W + x;             // (W4: x is a multiple of 4 inside [0, W))
            const uint32_t w = ft_load_word<W4>(img_all + base, p0, x, W);
            o = w;
            if (w) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v = (w >> (8 * k)) & 0xffu;
                    if (!v) continue;                          // LUT_MULTI: key = value
                    const int r = L_all[base + L_all[base + p0 + k]];
                    // area < mean(areas) = S / n, evaluated exactly in integers: area * n < S.  (The reference compares in float64;
                    // S / n is an integer or at least 1 / n away from one, far more than a rounding error, so both orders agree.
                    // n == 0: the reference's mean is NaN and every comparison false; here S == 0 gives the same.)
                    const long long a = (long long)area_all[base + r];
                    uint32_t nv = v;
                    if (v == 1) { if (a * gs[0] < gs[1]) nv = 0; }
                    else if (v == 2) { if (a * gs[2] < gs[3]) nv = 3; }
                    else if (v == 3) { if (a < (long long)ec_thresh) nv = 0; }
                    o = (o & ~(0xffu << (8 * k))) | (nv << (8 * k));
                }
            }
        }
        *reinterpret_cast<uint32_t*>(T + hy * FT_HW + wx * 4) = o;
    }
    __syncthreads();
    // stage 2: img[dilate(img == 3) ^ erode(img == 3)] = 0 (dilation pads with 0, erosion with 1)
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r, x = x0 + lane;
        if (y >= H || x >= W) continue;
        const uint8_t* c = T + (wave * CCL_ROWS + r + R) * FT_HW + lane + FT_HX;
        const uint8_t v = c[0];
        const bool cc = v == 3;
        const bool hn = y > 0, hs = y < H - 1, hw = x > 0, he = x < W - 1;
        const bool nn = hn && c[-FT_HW] == 3, ss = hs && c[FT_HW] == 3, ww = hw && c[-1] == 3, ee = he && c[1] == 3;
        const bool dil = cc | nn | ss | ww | ee;
        const bool ero = cc & (hn ? nn : true) & (hs ? ss : true) & (hw ? ww : true) & (he ? ee : true);
        out_all[base + (size_t)y * W + x] = (dil != ero) ? (uint8_t)0 : v;
    }
}

template <bool FINAL, bool W4>
__global__ __launch_bounds__(256) void merge_open_kernel(CclGeom g, const uint8_t* __restrict__ cur_all, uint8_t* __restrict__ out_all,
                                                         const int32_t* __restrict__ L_all, const uint32_t* __restrict__ flag_all,
                                                         const int32_t* __restrict__ G_all, int c, int m, uint32_t lut) {
    constexpr int R = FINAL ? 3 : 2, HH = CCL_BLOCK_ROWS + 2 * R;
    __shared__ __attribute__((aligned(16))) uint8_t Tm[HH * FT_HW];      // merge_comp's working image t (0xff outside the image)
    __shared__ __attribute__((aligned(16))) uint8_t Or[HH * FT_HW];      // the label image itself
    __shared__ __attribute__((aligned(16))) uint8_t Er[HH * FT_HW];      // grey erosion of t (0 outside the image)
    __shared__ __attribute__((aligned(16))) uint8_t Cm[FINAL ? HH * FT_HW : 16];   // FINAL: merged result before the last dilation
    __shared__ int s_last;
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int yblk = y0 - wave * CCL_ROWS, x0 = cx * 64;
    const int H = g.H, W = g.W;
    const size_t base = (size_t)img * H * W;
    if (tid == 0) s_last = g_fold_max(G_all + (size_t)img * G_IMG, G_LAST_ROOT) - 1;
    __syncthreads();
    const int last = s_last;
    // stage 1: t = the image with class m lifted out and every component (of the remaining non-zero pixels) that holds a class-c
    // pixel turned into c - except the component with the highest scipy label (src/image_tools.py:27: range(1, num))
    for (int i = tid; i < HH * (FT_HW / 4); i += 256) {
        const int hy = i / (FT_HW / 4), wx = i - hy * (FT_HW / 4);
        const int y = yblk - R + hy, x = x0 - FT_HX + wx * 4;
        uint32_t t = 0xffffffffu, orig = 0;
        if (y >= 0 && y < H && x + 3 >= 0 && x < W) {
            const size_t p0 = (size_t)y * W + x;
            const uint32_t w = ft_load_word<W4>(cur_all + base, p0, x, W);
            orig = w;
            t = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = (w >> (8 * k)) & 0xffu;
                uint32_t tv = (v == (uint32_t)m) ? 0u : v;
                if (!W4 && (x + k < 0 || x + k >= W)) tv = 0xffu;
                else if (v && key_of((uint8_t)v, lut)) {
                    const int r = L_all[base + L_all[base + p0 + k]];
                    if ((flag_all[base + r] & 1u) && r != last) tv = (uint32_t)c;
                }
                t |= tv << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t*>(Tm + hy * FT_HW + wx * 4) = t;
        *reinterpret_cast<uint32_t*>(Or + hy * FT_HW + wx * 4) = orig;
    }
    __syncthreads();
    // stage 2: grey erosion with the 3x3 cross, out-of-image neighbours ignored (0xff never wins a minimum of labels)
    for (int i = tid; i < (HH - 2) * (FT_HW - 2); i += 256) {
        const int hy = 1 + i / (FT_HW - 2), hx = 1 + i % (FT_HW - 2);
        const uint8_t* p = Tm + hy * FT_HW + hx;
        uint8_t v = p[0];
        if (v != 0xff) v = min(min(min(v, p[-1]), min(p[1], p[-FT_HW])), p[FT_HW]);
        Er[hy * FT_HW + hx] = v == 0xff ? (uint8_t)0 : v;
    }
    __syncthreads();
    // stage 3: grey dilation of the eroded image (= opening; the 0 of out-of-image pixels never wins a maximum), pixels whose
    // opened value equals c become c, lifted class m is restored
    auto combine = [&](int hy, int hx) -> uint8_t {
        const uint8_t* e = Er + hy * FT_HW + hx;
        const uint8_t mx = max(max(max(e[0], e[-1]), max(e[1], e[-FT_HW])), e[FT_HW]);
        const uint8_t orig = Or[hy * FT_HW + hx];
        return (orig == m) ? (uint8_t)m : (mx == c ? (uint8_t)c : Tm[hy * FT_HW + hx]);
    };
    if (!FINAL) {
#pragma unroll
        for (int r = 0; r < CCL_ROWS; ++r) {
            const int y = y0 + r, x = x0 + lane;
            if (y >= H || x >= W) continue;
            out_all[base + (size_t)y * W + x] = combine(wave * CCL_ROWS + r + R, lane + FT_HX);
        }
        return;
    }
    // FINAL: the merged image on the tile + 1 pixel, then img[dilate(img == 3)] = 3 (out-of-image neighbours ignored)
    for (int i = tid; i < (CCL_BLOCK_ROWS + 2) * 66; i += 256) {
        const int hy = R - 1 + i / 66, hx = FT_HX - 1 + i % 66;
        const int y = yblk - R + hy, x = x0 - FT_HX + hx;
        Cm[hy * FT_HW + hx] = (y >= 0 && y < H && x >= 0 && x < W) ? combine(hy, hx) : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CCL_ROWS; ++r) {
        const int y = y0 + r, x = x0 + lane;
        if (y >= H || x >= W) continue;
        const uint8_t* q = Cm + (wave * CCL_ROWS + r + R) * FT_HW + lane + FT_HX;
        const bool d = q[0] == 3 || q[-1] == 3 || q[1] == 3 || q[-FT_HW] == 3 || q[FT_HW] == 3;
        out_all[base + (size_t)y * W + x] = d ? (uint8_t)3 : q[0];
    }
}

hipError_t run_meta_inference(PostWorkspace& ws, uint8_t* img, int n_img, int H, int W, int32_t* n_ec_dev, hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    const size_t px = (size_t)H * W;
    const dim3 ig = img_grid(px, n_img);
    const unsigned tg = geom_grid(g);
    // W % 4 == 0 and 4-byte-aligned images: 32-bit loads in the fused tile kernels, four pixels per thread in the nucleus kill
    const bool v4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(ws.tmpA) |
                                      reinterpret_cast<uintptr_t>(ws.tmpB)) & 3) == 0;
    const dim3 ig4 = img_grid(px / 4, n_img);
    hipError_t e;
    // one counter block per labelling that produces counters, all zeroed here: no zeroing / folding launches in between
    // (a kernel, not hipMemsetAsync: replayed from a HIP graph - option post_graph - the memset node of ROCm 7.2 was not ordered
    // before the kernels behind it and zeroed list counters in mid-use: a memory fault on speckled images, round 5)
    const size_t gslot = (size_t)ws.cap_img * G_IMG;
    {
        const int nz = (int)((n_ec_dev ? 5 : 4) * gslot);
        hipLaunchKernelGGL(zero_g_kernel, dim3((nz + 255) / 256), dim3(256), 0, s, ws.g, nz);
    }
    // 1. fill_holes(1), fill_holes(2)
    for (int c = 1; c <= 2; ++c) {
        CclPass p{img, lut_ne(c), 4, 0, AUX_BORDER, 0, nullptr, 0};
        p.sparse = true;
        p.allow_full = true;
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(apply_fill_tile_kernel, dim3(tg), dim3(256), 0, s, g, img, ws.L, ws.flag, c, lut_ne(c), ws.tile_any, ws.own_bits);
    }
    // 2-5. size_thresh + ecDNA band removal -> ws.tmpA (the caller's buffer is free from here until the last step writes the
    // result into it)
    uint8_t* cur = ws.tmpA;
    uint8_t* nxt = ws.tmpB;
    {
        CclPass p{img, LUT_MULTI, 8, STAT_AREA, AUX_NONE, 0, nullptr, NEED_NCOMP | NEED_NPX};
        p.sparse = true;
        p.g = ws.g + 0 * gslot;
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        if (v4) hipLaunchKernelGGL(thresh_band_kernel<true>, dim3(tg), dim3(256), 0, s, g, img, cur, ws.L, ws.area, p.g, 15);
        else hipLaunchKernelGGL(thresh_band_kernel<false>, dim3(tg), dim3(256), 0, s, g, img, cur, ws.L, ws.area, p.g, 15);
    }
    // 6. nucleus-in-metaphase test
    {
        const size_t cap = px / 4 + (size_t)(H + W) / 2 + 4;
        int32_t* list1 = ws.list;
        double2* list2 = reinterpret_cast<double2*>(ws.list + (((size_t)n_img * cap + 3) & ~(size_t)3));
        CclPass p{cur, LUT_MULTI, 8, STAT_AREA | STAT_SUMS, AUX_NONE, 0, nullptr, NEED_LISTS};
        p.sparse = true;
        p.g = ws.g + 1 * gslot;
        p.list1 = list1; p.list2 = reinterpret_cast<int32_t*>(list2); p.list_cap = cap;
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        const bool binned = H <= NUCLEUS_BIN_EXTENT && W <= NUCLEUS_BIN_EXTENT;
        // the pair test is only ever needed for images the binned test cannot take: extents beyond the LDS histogram, or more
        // chromosomes than `binned` holds (n2 <= cap); it reads the centroids that centroids_kernel leaves in the list
        const bool pairs = !binned || cap > ws.binned_cap;
        if (pairs) hipLaunchKernelGGL(centroids_kernel, dim3(8, n_img), dim3(256), 0, s, ws.area, ws.sumy, ws.sumx, p.g, list2, px, cap);
        if (binned) {
            const size_t bin_lds = (size_t)(std::max(H, W) + 2) * sizeof(int);
            static DeviceOnce bin_attr;                            // up to 128 KB of dynamic LDS: per-device function attribute
            e = bin_attr.run([] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(bin_centroids_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((NUCLEUS_BIN_EXTENT + 2) * sizeof(int)));
            });
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(bin_centroids_kernel, dim3(2, n_img), dim3(1024), bin_lds, s, p.g, list2, ws.binned, ws.binstart,
                               cap, ws.binned_cap, H, W, ws.area, ws.sumy, ws.sumx, px, pairs ? 1 : 0);
            hipLaunchKernelGGL(nucleus_test_binned_kernel, dim3(32, n_img), dim3(256), 0, s, ws.area, ws.sumy, ws.sumx, p.g,
                               list1, ws.binned, ws.binstart, ws.flag, px, cap, ws.binned_cap, H, W, 70.0, 5);
        }
        if (pairs) {
            const int bpi = 32;
            hipLaunchKernelGGL(nucleus_test_kernel, dim3(n_img * bpi), dim3(256), 0, s, ws.area, ws.sumy, ws.sumx, p.g, list1,
                               list2, ws.flag, px, cap, bpi, 70.0, 5, H, W, ws.binned_cap);
        }
        if (v4) hipLaunchKernelGGL(nucleus_kill4_kernel, ig4, dim3(256), 0, s, cur, ws.L, ws.flag, n_img, px / 4);
        else hipLaunchKernelGGL(apply_nucleus_kill_kernel, ig, dim3(256), 0, s, cur, ws.L, ws.flag, n_img, px);
    }
    // 7-8. merge_comp(1), merge_comp(2); 9. the final ecDNA dilation rides on the second one, which writes the caller's buffer
    for (int c = 1; c <= 2; ++c) {
        const int m = (c == 1) ? 2 : 1;
        const uint32_t lut = lut_nonzero_except(m);
        CclPass p{cur, lut, 8, 0, AUX_VALUE_EQ, c, nullptr, NEED_LAST};
        p.sparse = true;
        p.g = ws.g + (size_t)(1 + c) * gslot;
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        if (c == 1) {
            if (v4) hipLaunchKernelGGL((merge_open_kernel<false, true>), dim3(tg), dim3(256), 0, s, g, cur, nxt, ws.L, ws.flag, p.g, c, m, lut);
            else hipLaunchKernelGGL((merge_open_kernel<false, false>), dim3(tg), dim3(256), 0, s, g, cur, nxt, ws.L, ws.flag, p.g, c, m, lut);
            std::swap(cur, nxt);
        } else {
            if (v4) hipLaunchKernelGGL((merge_open_kernel<true, true>), dim3(tg), dim3(256), 0, s, g, cur, img, ws.L, ws.flag, p.g, c, m, lut);
            else hipLaunchKernelGGL((merge_open_kernel<true, false>), dim3(tg), dim3(256), 0, s, g, cur, img, ws.L, ws.flag, p.g, c, m, lut);
        }
    }
    // 10. count_cc(img == 3)[0]
    if (n_ec_dev) {
        CclPass p{img, lut_eq(3), 8, 0, AUX_NONE, 0, nullptr, NEED_NCOMP | NEED_NPX, true};
        p.sparse = true;
        p.g = ws.g + 4 * gslot;
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(gather_counts_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, p.g, n_img, 1, n_ec_dev,
                           (long long*)nullptr, (long long)px, 1);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// counting entry points
// ---------------------------------------------------------------------------------------------------------------
hipError_t run_count_cc(PostWorkspace& ws, const uint8_t* mask, int n_img, int H, int W, int32_t* n_dev, long long* px_dev,
                        hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    CclPass p{mask, LUT_NONZERO, 8, 0, AUX_NONE, 0, nullptr, NEED_NCOMP | NEED_NPX, true};
    p.sparse = true;                                           // count_roots looks at the key first
    hipError_t e = run_ccl_pass(ws, g, p, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_counts_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, ws.g, n_img, 1, n_dev, px_dev,
                       (long long)H * W, 0);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void export_labels_kernel(const int32_t* __restrict__ L, int32_t* __restrict__ out, size_t total, size_t px) {
    PX_LOOP(total) {                    // background -1 -> 0, component -> 1 + raster index of its first pixel (pixel -> owner -> global root)
        const int o = L[t];
        out[t] = o < 0 ? 0 : L[(t / px) * px + o] + 1;
    }
}

hipError_t run_ccl_labels(PostWorkspace& ws, const uint8_t* mask, int n_img, int H, int W, int conn, int32_t* labels_dev,
                          hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    CclPass p{mask, LUT_NONZERO, conn, 0, AUX_NONE, 0, nullptr, 0};
    hipError_t e = run_ccl_pass(ws, g, p, s);
    if (e != hipSuccess) return e;
    const size_t total = (size_t)n_img * H * W;
    hipLaunchKernelGGL(export_labels_kernel, dim3(px_grid(total)), dim3(256), 0, s, ws.L, labels_dev, total, (size_t)H * W);
    return hipGetLastError();
}

// Up to 5 counts of roots by (key, flag bits) in ONE pass over the owner bits (round 4): roots are owners that still point at
// themselves, so only the 256 B of owner bits per tile and the owners' parents / flags are read - the per-pixel scan below
// (image + parent of every pixel, once per query) took 6 x 181 us of the 3.9 ms overlay row of 64 images.
// Query i: roots whose key == key[i] (0 = any) and whose flag word has all bits of need[i] -> G[G_CNT0 + slot[i]] (sharded:
// fold with reduce_g_kernel afterwards).
struct FlagQueries { int n; int key[5]; uint32_t need[5]; int slot[5]; };
__global__ __launch_bounds__(256) void count_flagged_owner_roots_kernel(CclGeom g, const uint8_t* __restrict__ img_all, uint32_t lut,
                                                                        const int32_t* __restrict__ L_all,
                                                                        const uint32_t* __restrict__ flag_all,
                                                                        int32_t* __restrict__ G_all,
                                                                        const uint8_t* __restrict__ tile_any,
                                                                        const uint32_t* __restrict__ own_bits, FlagQueries Q) {
    __shared__ int red[8];
    int img, y0, cx;
    if (!decode_block(g, img, y0, cx)) return;
    const size_t ti = tile_index(g, img);
    const int ta = tile_any[ti];
    if (!ta) return;
    const int tid = threadIdx.x;
    if (tid < 8) red[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)img * g.H * g.W;
    const int yblk = y0 - (tid >> 6) * CCL_ROWS;
    // thread = 8 consecutive pixels of the tile; a FULL tile (tile_any == 2) has one owner, its first pixel, and no owner bits
    uint32_t bits = ta == 2 ? (tid == 0 ? 1u : 0u) : (own_bits[ti * 64 + (tid >> 2)] >> ((tid & 3) * 8)) & 0xffu;
    int c[5] = {0, 0, 0, 0, 0};
    while (bits) {
        const int k = __ffs((int)bits) - 1;
        bits &= bits - 1;
        const int li = tid * 8 + k;
        const int p = (yblk + (li >> 6)) * g.W + cx * 64 + (li & 63);
        if (L_all[base + p] != p) continue;
        const int key = key_of(img_all[base + p], lut);
        const uint32_t f = flag_all[base + p];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (i < Q.n && (!Q.key[i] || key == Q.key[i]) && (f & Q.need[i]) == Q.need[i]) c[i] += 1;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) if (c[i]) atomicAdd(&red[i], c[i]);
    __syncthreads();
    int32_t* G = G_all + (size_t)img * G_IMG + (size_t)(block_in_image(g) % G_SHARDS) * G_STRIDE;
    if (tid < Q.n && red[tid]) atomicAdd(G + G_CNT0 + Q.slot[tid], red[tid]);
}

static void launch_flagged_counts(PostWorkspace& ws, const CclGeom& g, const uint8_t* key_img, uint32_t lut, const FlagQueries& Q,
                                  hipStream_t s) {
    hipLaunchKernelGGL(count_flagged_owner_roots_kernel, dim3(geom_grid(g)), dim3(256), 0, s, g, key_img, lut, ws.L, ws.flag, ws.g,
                       ws.tile_any, ws.own_bits, Q);
    hipLaunchKernelGGL(reduce_g_kernel, dim3(g.n_img), dim3(G_STRIDE), 0, s, ws.g, g.n_img);
}

__global__ void gather_slot_kernel(const int32_t* __restrict__ G_all, int n_img, int slot, int key_for_quirk,
                                   long long full_px, int32_t* __restrict__ out32, long long* __restrict__ out64,
                                   int out_stride, int out_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    const int32_t* G = G_all + (size_t)i * G_IMG;
    int v = G[G_CNT0 + slot];
    // np.unique(regs)[1:] (src/image_tools.py:107,129) drops the only component of a mask without background
    if (key_for_quirk && G[G_NPX + key_for_quirk] == full_px) v = 0;
    if (out32) out32[(size_t)i * out_stride + out_off] = v;
    if (out64) out64[(size_t)i * out_stride + out_off] = v;
}

__global__ __launch_bounds__(256) void mask_to_bits_kernel(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, size_t total) {
    PX_LOOP(total) out[t] = a[t] ? 1 : 0;
}

hipError_t run_count_coloc(PostWorkspace& ws, const uint8_t* ob1, const uint8_t* ob2, int n_img, int H, int W, int32_t* n_dev,
                           hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    const size_t px = (size_t)H * W, total = px * n_img;
    hipLaunchKernelGGL(mask_to_bits_kernel, dim3(px_grid(total)), dim3(256), 0, s, ob2, ws.tmpA, total);
    CclPass p{ob1, LUT_NONZERO, 8, 0, AUX_IMAGE, 0, ws.tmpA, NEED_NPX};
    hipError_t e = run_ccl_pass(ws, g, p, s);
    if (e != hipSuccess) return e;
    launch_flagged_counts(ws, g, ob1, LUT_NONZERO, FlagQueries{1, {0}, {1u}, {0}}, s);
    hipLaunchKernelGGL(gather_slot_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, ws.g, n_img, 0, 1, (long long)px, n_dev,
                       (long long*)nullptr, 1, 0);
    return hipGetLastError();
}

// remove_small_objects(fish, thr) on a bool image: 4-connected components with area < thr are dropped
__global__ __launch_bounds__(256) void keep_large_kernel(const int32_t* __restrict__ L, const uint32_t* __restrict__ area,
                                                         uint8_t* __restrict__ out, size_t total, size_t px, int thr, int bit) {
    PX_LOOP(total) {
        const int o = L[t];                                    // (non-sparse labelling: unkeyed pixels hold -1)
        uint8_t v = out[t];
        if (o >= 0) {
            const size_t ib = (t / px) * px;
            if (area[ib + L[ib + o]] >= (uint32_t)thr) v |= (uint8_t)(1u << bit);
        }
        out[t] = v;
    }
}

__global__ __launch_bounds__(256) void zero_u8_kernel(uint8_t* __restrict__ out, size_t total) {
    PX_LOOP(total) out[t] = 0;
}

hipError_t run_count_hsr(PostWorkspace& ws, const uint8_t* chrom, const uint8_t* fish, int n_img, int H, int W, int thr,
                         int32_t* n_dev, hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    const size_t px = (size_t)H * W, total = px * n_img;
    CclPass p1{fish, LUT_NONZERO, 4, STAT_AREA, AUX_NONE, 0, nullptr, 0};
    hipError_t e = run_ccl_pass(ws, g, p1, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(zero_u8_kernel, dim3(px_grid(total)), dim3(256), 0, s, ws.tmpA, total);
    hipLaunchKernelGGL(keep_large_kernel, dim3(px_grid(total)), dim3(256), 0, s, ws.L, ws.area, ws.tmpA, total, px, thr, 0);
    CclPass p2{chrom, LUT_NONZERO, 8, 0, AUX_IMAGE, 0, ws.tmpA, NEED_NPX};
    if ((e = run_ccl_pass(ws, g, p2, s)) != hipSuccess) return e;
    launch_flagged_counts(ws, g, chrom, LUT_NONZERO, FlagQueries{1, {0}, {1u}, {0}}, s);
    hipLaunchKernelGGL(gather_slot_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, ws.g, n_img, 0, 1, (long long)px, n_dev,
                       (long long*)nullptr, 1, 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// meta_overlay row (src/meta_overlay.py:59-95)
// ---------------------------------------------------------------------------------------------------------------
// which: 0 fish  = green & ~nuclei      1 fish2 = red & ~nuclei
//        2 A     = fish  & ~chrom       3 B     = fish2 & ~chrom
__global__ __launch_bounds__(256) void overlay_mask_kernel(const uint8_t* __restrict__ labels, const uint8_t* __restrict__ rgb,
                                                           int C, int sens, int which, uint8_t* __restrict__ out, size_t total) {
    PX_LOOP(total) {
        const uint8_t v = labels[t];
        const bool red = rgb[t * C + 0] > sens, green = rgb[t * C + 1] > sens;
        bool m = (which & 1) ? red : green;
        m = m && v != 1;
        if (which >= 2) m = m && v != 2;
        out[t] = m ? 1 : 0;
    }
}

// aux bits for the (chromosome, ecDNA) labelling: bit0 fish, bit1 fish2, bit2 fish & fish2 (ecDNA colocalisation);
// bits 3 / 4 (size-filtered fish2 / fish, for HSR) are OR-ed in afterwards by keep_large_kernel.
__global__ __launch_bounds__(256) void overlay_aux_kernel(const uint8_t* __restrict__ labels, const uint8_t* __restrict__ rgb,
                                                          int C, int sens, uint8_t* __restrict__ aux, size_t total) {
    PX_LOOP(total) {
        const uint8_t v = labels[t];
        const bool fish2 = rgb[t * C + 0] > sens && v != 1, fish = rgb[t * C + 1] > sens && v != 1;
        aux[t] = (uint8_t)((fish ? 1 : 0) | (fish2 ? 2 : 0) | ((fish && fish2) ? 4 : 0));
    }
}

__global__ void overlay_cc_gather_kernel(const int32_t* __restrict__ G_all, int n_img, int key, long long full_px,
                                         long long* __restrict__ out, int off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    const int n = G_all[(size_t)i * G_IMG + G_NCOMP + key];
    const long long px = G_all[(size_t)i * G_IMG + G_NPX + key];
    out[(size_t)i * 12 + off] = n;
    out[(size_t)i * 12 + off + 1] = (n == 0 || px == full_px) ? -1 : px;
}

hipError_t run_overlay(PostWorkspace& ws, const uint8_t* labels, const uint8_t* rgb, int n_img, int H, int W, int C, int sens,
                       int hsr_thr, long long* out, hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const CclGeom g = make_geom(n_img, H, W);
    const size_t px = (size_t)H * W, total = px * n_img;
    const unsigned pg = px_grid(total);
    const unsigned ig = (n_img + 63) / 64;
    uint8_t* aux = ws.tmpB;    // flag bits for the (chrom, ec) labelling
    uint8_t* msk = ws.tmpA;    // mask being labelled
    hipError_t e;
    hipLaunchKernelGGL(overlay_aux_kernel, dim3(pg), dim3(256), 0, s, labels, rgb, C, sens, aux, total);
    // HSR: size-filtered fish2 (red) -> bit3, fish (green) -> bit4   (4-connectivity, src/image_tools.py:104)
    for (int k = 0; k < 2; ++k) {
        const int which = (k == 0) ? 1 : 0;
        hipLaunchKernelGGL(overlay_mask_kernel, dim3(pg), dim3(256), 0, s, labels, rgb, C, sens, which, msk, total);
        CclPass p{msk, LUT_NONZERO, 4, STAT_AREA, AUX_NONE, 0, nullptr, 0};
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(keep_large_kernel, dim3(pg), dim3(256), 0, s, ws.L, ws.area, aux, total, px, hsr_thr, 3 + k);
    }
    // chromosomes (key 2) and ecDNA (key 3) in one labelling
    {
        const uint32_t lut = 0x03020000u;
        CclPass p{labels, lut, 8, 0, AUX_IMAGE, 0, aux, NEED_NCOMP | NEED_NPX};
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(overlay_cc_gather_kernel, dim3(ig), dim3(64), 0, s, ws.g, n_img, 3, (long long)px, out, 0);
        const struct { int key; uint32_t need; int slot; int off; } q[5] = {
            {3, 1u, 0, 6}, {3, 2u, 1, 7}, {3, 4u, 2, 9}, {2, 8u, 3, 10}, {2, 16u, 4, 11}};
        FlagQueries Q{5, {}, {}, {}};
        for (int k = 0; k < 5; ++k) { Q.key[k] = q[k].key; Q.need[k] = q[k].need; Q.slot[k] = q[k].slot; }
        launch_flagged_counts(ws, g, labels, lut, Q, s);
        for (int k = 0; k < 5; ++k) {
            hipLaunchKernelGGL(gather_slot_kernel, dim3(ig), dim3(64), 0, s, ws.g, n_img, q[k].slot, q[k].key, (long long)px,
                               (int32_t*)nullptr, out, 12, q[k].off);
        }
    }
    // A = fish & ~chrom: count_cc(A), coloc(A, B)
    {
        hipLaunchKernelGGL(overlay_mask_kernel, dim3(pg), dim3(256), 0, s, labels, rgb, C, sens, 2, msk, total);
        hipLaunchKernelGGL(overlay_mask_kernel, dim3(pg), dim3(256), 0, s, labels, rgb, C, sens, 3, aux, total);
        CclPass p{msk, LUT_NONZERO, 8, 0, AUX_IMAGE, 0, aux, NEED_NCOMP | NEED_NPX};
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(overlay_cc_gather_kernel, dim3(ig), dim3(64), 0, s, ws.g, n_img, 1, (long long)px, out, 2);
        launch_flagged_counts(ws, g, msk, LUT_NONZERO, FlagQueries{1, {0}, {1u}, {0}}, s);
        hipLaunchKernelGGL(gather_slot_kernel, dim3(ig), dim3(64), 0, s, ws.g, n_img, 0, 1, (long long)px, (int32_t*)nullptr,
                           out, 12, 8);
    }
    // B = fish2 & ~chrom: count_cc(B)
    {
        CclPass p{aux, LUT_NONZERO, 8, 0, AUX_NONE, 0, nullptr, NEED_NCOMP | NEED_NPX};
        if ((e = run_ccl_pass(ws, g, p, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(overlay_cc_gather_kernel, dim3(ig), dim3(64), 0, s, ws.g, n_img, 1, (long long)px, out, 4);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// stitch + img_as_ubyte + argmax (src/image_tools.py:188-252, src/utils.py:117-118)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stitch_argmax_kernel(const float* __restrict__ probs, int prob_cs,
                                                            const int32_t* __restrict__ src_map, int n_pos, size_t px,
                                                            uint8_t* __restrict__ labels, size_t total,
                                                            int32_t* __restrict__ tie_risk) {
    __shared__ int s_im, s_cnt;
    int cnt = 0, cur = -1;                                 // tie-risk pixels of image `cur` seen by this thread
    PX_LOOP(total) {
        const size_t im = t / px, q = t - im * px;
        const int src = src_map[q];
        uint8_t lab = 0;                       // never-written canvas pixels stay 0.0 in every channel -> argmax 0
        bool tie = false;
        if (src >= 0) {
            const size_t patch = im * n_pos + (size_t)(src >> 16);
            const float* pp = probs + ((patch << 16) + (size_t)(src & 0xffff)) * prob_cs;
            int best = -1, second = -1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // float64(p) * 255 is exact; rint = round half to even; clip to [0, 255]
                double v = rint((double)pp[c] * 255.0);
                v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
                const int qv = (int)v;
                if (qv > best) { second = best; best = qv; lab = (uint8_t)c; }   // strict '>' keeps the first maximum
                else if (qv > second) second = qv;
            }
            tie = best - second <= 1;           // a last-bit difference in one probability can change this pixel's label
        }
        labels[t] = lab;
        if (tie_risk) {
            if ((int)im != cur) {                              // (a thread's pixels cross an image boundary a few times per launch at most)
                if (cnt) atomicAdd(tie_risk + ((size_t)cur * G_SHARDS + blockIdx.x % G_SHARDS) * G_STRIDE, cnt);
                cur = (int)im; cnt = 0;
            }
            cnt += tie ? 1 : 0;
        }
    }
    if (tie_risk) {
        // one global atomic per workgroup, into one of G_SHARDS replicas of the image's counter (each on its own 128-B line):
        // device-scope atomics on ONE line serialise at ~200 ns each across the XCDs - 16 k workgroups adding to the 16
        // counters of a launch group, all in one line, took 3.2 ms (the whole stitch: 0.09 ms)
        if (threadIdx.x == 0) { s_im = cur; s_cnt = 0; }
        __syncthreads();
        if (cnt) {
            if (cur == s_im) atomicAdd(&s_cnt, cnt);
            else atomicAdd(tie_risk + ((size_t)cur * G_SHARDS + blockIdx.x % G_SHARDS) * G_STRIDE, cnt);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) atomicAdd(tie_risk + ((size_t)s_im * G_SHARDS + blockIdx.x % G_SHARDS) * G_STRIDE, s_cnt);
    }
}

__global__ __launch_bounds__(256) void stitch_probs_kernel(const float* __restrict__ probs, int prob_cs,
                                                           const int32_t* __restrict__ src_map, int n_pos, size_t px,
                                                           float* __restrict__ out, size_t total) {
    PX_LOOP(total) {
        const size_t im = t / px, q = t - im * px;
        const int src = src_map[q];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (src >= 0) {
            const size_t patch = im * n_pos + (size_t)(src >> 16);
            const float* pp = probs + ((patch << 16) + (size_t)(src & 0xffff)) * prob_cs;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = pp[c];
        }
        *reinterpret_cast<f32x4*>(out + t * 4) = v;
    }
}

__global__ void tie_reduce_kernel(const int32_t* __restrict__ shards, int32_t* __restrict__ out, int n_img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    int v = 0;
    for (int k = 0; k < G_SHARDS; ++k) v += shards[((size_t)i * G_SHARDS + k) * G_STRIDE];
    out[i] = v;
}

hipError_t launch_stitch_argmax(const float* probs, int prob_cs, const int32_t* src_map, int n_img, int n_pos, int H, int W,
                                uint8_t* labels, hipStream_t s, int32_t* tie_risk, int32_t* tie_shards) {
    const size_t px = (size_t)H * W, total = px * n_img;
    if (!total) return hipSuccess;
    const bool tie = tie_risk != nullptr && tie_shards != nullptr;
    if (tie) {
        hipError_t e = hipMemsetAsync(tie_shards, 0, (size_t)n_img * G_SHARDS * G_STRIDE * sizeof(int32_t), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stitch_argmax_kernel, dim3(px_grid(total)), dim3(256), 0, s, probs, prob_cs, src_map, n_pos, px,
                       labels, total, tie ? tie_shards : nullptr);
    if (tie) hipLaunchKernelGGL(tie_reduce_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, tie_shards, tie_risk, n_img);
    return hipGetLastError();
}

hipError_t launch_stitch_probs(const float* probs, int prob_cs, const int32_t* src_map, int n_img, int n_pos, int H, int W,
                               float* out, hipStream_t s) {
    const size_t px = (size_t)H * W, total = px * n_img;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(stitch_probs_kernel, dim3(px_grid(total)), dim3(256), 0, s, probs, prob_cs, src_map, n_pos, px, out, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// meta_preprocess (src/image_tools.py:86-101)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void extract_gray_kernel(const T* __restrict__ img, int C, uint8_t* __restrict__ gray,
                                                           size_t total) {
    const int ch = C > 1 ? 2 : 0;
    PX_LOOP(total) {
        const T v = img[t * C + ch];
        if (sizeof(T) == 2) {
            // cv2.convertScaleAbs(alpha = 255/65535): saturate_cast<uchar>(|float(v) * float(alpha)|), round half even
            const float f = fabsf((float)v * (float)(255.0 / 65535.0));
            int r = __float2int_rn(f);
            gray[t] = (uint8_t)(r > 255 ? 255 : r);
        } else {
            gray[t] = (uint8_t)v;
        }
    }
}

hipError_t launch_u16_to_u8(const uint16_t* in, uint8_t* out, size_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(extract_gray_kernel<uint16_t>, dim3(px_grid(count)), dim3(256), 0, s, in, 1, out, count);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void hist256_kernel(const uint8_t* __restrict__ gray, size_t px, uint32_t* __restrict__ hist,
                                                      int blocks_per_img) {
    __shared__ uint32_t h[256];
    const int im = blockIdx.x / blocks_per_img, j = blockIdx.x % blocks_per_img;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* g = gray + (size_t)im * px;
    for (size_t t = (size_t)j * 256 + threadIdx.x; t < px; t += (size_t)blocks_per_img * 256) atomicAdd(&h[g[t]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(hist + (size_t)im * 256 + threadIdx.x, h[threadIdx.x]);
}

// Otsu threshold (OpenCV getThreshVal_Otsu_8u restated) + the "> 50 % white" decision; one thread per image
__global__ void otsu_decide_kernel(const uint32_t* __restrict__ hist, int n_img, size_t px, int32_t* __restrict__ inverted) {
    const int im = blockIdx.x * blockDim.x + threadIdx.x;
    if (im >= n_img) return;
    const uint32_t* h = hist + (size_t)im * 256;
    const double scale = 1.0 / (double)px;
    double mu = 0.0;
    for (int i = 0; i < 256; ++i) mu += (double)i * (double)h[i];
    mu *= scale;
    double mu1 = 0.0, q1 = 0.0, max_sigma = 0.0;
    int max_val = 0;
    const double eps = 1.1920928955078125e-07;   // FLT_EPSILON
    for (int i = 0; i < 256; ++i) {
        const double p_i = (double)h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (fmin(q1, q2) < eps || fmax(q1, q2) > 1.0 - eps) continue;
        mu1 = (mu1 + (double)i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    unsigned long long white = 0;
    for (int i = max_val + 1; i < 256; ++i) white += h[i];
    inverted[im] = ((double)white > (double)px * 0.5) ? 1 : 0;
}

__global__ __launch_bounds__(256) void invert_kernel(uint8_t* __restrict__ gray, const int32_t* __restrict__ inverted,
                                                     size_t px, size_t total) {
    PX_LOOP(total) {
        if (inverted[t / px]) gray[t] = (uint8_t)~gray[t];
    }
}

hipError_t run_preprocess(const void* img, int n_img, int H, int W, int C, int bps, uint8_t* gray, int32_t* inverted,
                          uint32_t* hist_ws, hipStream_t s) {
    if (n_img <= 0) return hipSuccess;
    const size_t px = (size_t)H * W, total = px * n_img;
    const unsigned pg = px_grid(total);
    if (bps == 2) hipLaunchKernelGGL(extract_gray_kernel<uint16_t>, dim3(pg), dim3(256), 0, s, (const uint16_t*)img, C, gray, total);
    else hipLaunchKernelGGL(extract_gray_kernel<uint8_t>, dim3(pg), dim3(256), 0, s, (const uint8_t*)img, C, gray, total);
    hipError_t e = hipMemsetAsync(hist_ws, 0, (size_t)n_img * 256 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int bpi = 64;
    hipLaunchKernelGGL(hist256_kernel, dim3(n_img * bpi), dim3(256), 0, s, gray, px, hist_ws, bpi);
    hipLaunchKernelGGL(otsu_decide_kernel, dim3((n_img + 63) / 64), dim3(64), 0, s, hist_ws, n_img, px, inverted);
    hipLaunchKernelGGL(invert_kernel, dim3(pg), dim3(256), 0, s, gray, inverted, px, total);
    return hipGetLastError();
}

}  // namespace ecseg
