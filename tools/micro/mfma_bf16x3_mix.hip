// Micro-benchmark (round 6, the gate of VERDICT r05 item 1): what one SIMD sustains for the K loop of the Winograd F(4x4) kernel
// with its REAL instruction mix - LDS halo reads, row transform, column transform, MFMAs - and nothing else (no DMA, no barrier):
//
//   F   today's kernel: fp32 operands on v_mfma_f32_32x32x2_f32: per 8-channel group and wave 24 ds_read_b128 (halo) + 6 (filter),
//       72 + 48 transform fmas, 24 MFMAs of 64 pipe cycles
//   S1  3-way bf16 split of both operands on v_mfma_f32_32x32x16_bf16, same wave decomposition (transform row xi x 32 output
//       channels, 6 points): the lane's 4 channels and two split pieces fill the 8 K slots of a lane ("K-folded": [v1|v2] x [u1|u1],
//       [v1|v2] x [u2|u2], [v1|v3] x [u3|u1] = all six products above 2^-24), 18 MFMAs of 32 pipe cycles per group,
//       + 22 split instructions per point (v_and / v_sub / v_perm)
//   S2  the same arithmetic with wave = (transform row xi, HALF a row = 3 points) x 64 output channels: the column transform and the
//       split of a point serve two MFMA column blocks (18 MFMAs per group again)
//
// Prints cycles per 8-channel group per SIMD (3 waves) and the ratio to F.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float KA = 0.625f, KB = 1.5f, KA2 = KA * KA, KB2 = KB * KB, KP = KA2 * KB2, KS = -(KA2 + KB2);

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bfloat(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) { return __builtin_amdgcn_perm(fbits(hi), fbits(lo), 0x07060302u); }

// v (4 channels) -> three packed bf16 pieces (truncation: v1 + v2 + v3 == v exactly), 22 instructions
__device__ __forceinline__ void split3(const float v[4], u32x2& p1, u32x2& p2, u32x2& p3) {
    float r[4], s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = v[c] - bfloat(fbits(v[c]) & 0xffff0000u);
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = r[c] - bfloat(fbits(r[c]) & 0xffff0000u);
    p1 = u32x2{pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
    p2 = u32x2{pack_hi(r[0], r[1]), pack_hi(r[2], r[3])};
    p3 = u32x2{pack_hi(s[0], s[1]), pack_hi(s[2], s[3])};
}
__device__ __forceinline__ bf16x8 cat(u32x2 a, u32x2 b) { return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]}); }

// row transform of an inner row (18 reads beside the six direct ones), as the kernel does it
__device__ __forceinline__ void row_transform(const f32x4* A, f32x4 t[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x4 d3 = A[36 * 3 + j], d2 = A[36 * 2 + j], d1 = A[36 + j], d0 = A[j];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[j][c] = __builtin_fmaf(-KA2 * KB, d0[c], __builtin_fmaf(-KA2, d1[c], __builtin_fmaf(KB, d2[c], d3[c])));
    }
}
__device__ __forceinline__ void row_transform5(const f32x4* A, f32x4 t[6]) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const f32x4 d3 = A[36 * 3 + j], d2 = A[36 * 2 + j], d1 = A[36 + j], d0 = A[j];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[j][c] = __builtin_fmaf(-KA2 * KB, d0[c], __builtin_fmaf(-KA2, d1[c], __builtin_fmaf(KB, d2[c], d3[c])));
    }
}
__device__ __forceinline__ void col_points(const f32x4 t[6], int c, float V[6]) {
    const float u0 = t[0][c], u1 = t[1][c], u2 = t[2][c], u3 = t[3][c], u4 = t[4][c], u5 = t[5][c];
    const float ea = __builtin_fmaf(-KB2, u2, u4), oa = __builtin_fmaf(-KB2, u1, u3);
    const float eb = __builtin_fmaf(-KA2, u2, u4), ob = __builtin_fmaf(-KA2, u1, u3);
    V[0] = __builtin_fmaf(KP, u0, __builtin_fmaf(KS, u2, u4));
    V[1] = __builtin_fmaf(KA, oa, ea);
    V[2] = __builtin_fmaf(-KA, oa, ea);
    V[3] = __builtin_fmaf(KB, ob, eb);
    V[4] = __builtin_fmaf(-KB, ob, eb);
    V[5] = __builtin_fmaf(KP, u1, __builtin_fmaf(KS, u3, u5));
}

template <int MODE>
__global__ __launch_bounds__(768) void kmix(float* out, unsigned long long* cyc, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Hs = reinterpret_cast<f32x4*>(smem);                 // 3 x 1536 halo slots
    f32x4* Bs = Hs + 3 * 1536;                                  // per wave 2 x 192 filter slots
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3 * 1536 + 12 * 384; i += blockDim.x) Hs[i] = f32x4{1.f + i * 1e-4f, 0.5f, 0.25f, 2.f};
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    const int a_lane = ((li >> 4) * 18 + ((li >> 2) & 3)) * 36 + lh * 18 + (li & 3);
    f32x16 acc[6] = {};
    const f32x4* Bw = Bs + wave * 384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; ++g) {
        f32x4 t[6];
        if (MODE == 3) row_transform5(Hs + (g % 3) * 1536 + a_lane, t); else row_transform(Hs + (g % 3) * 1536 + a_lane, t);
        if (MODE == 0) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                float V[2][6];
                col_points(t, 2 * ss, V[0]);
                col_points(t, 2 * ss + 1, V[1]);
                f32x2 w2[6];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f32x4 w4 = Bw[(g & 1) * 192 + (ss * 3 + k) * 32 + lane];
                    w2[2 * k] = f32x2{w4[0], w4[1]};
                    w2[2 * k + 1] = f32x2{w4[2], w4[3]};
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int v = 0; v < 6; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[e][v], w2[v][e], acc[v], 0, 0, 0);
            }
        } else if (MODE == 1) {
            float V[4][6];
#pragma unroll
            for (int c = 0; c < 4; ++c) col_points(t, c, V[c]);
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                const float vv[4] = {V[0][v], V[1][v], V[2][v], V[3][v]};
                u32x2 p1, p2, p3;
                split3(vv, p1, p2, p3);
                // filter: compact image [u3 u1 | u2] per lane: one 16-byte + one 8-byte read
                const u32x4 f31 = __builtin_bit_cast(u32x4, Bw[(g & 1) * 32 + v * 48 + lane]);
                const u32x2 f2 = reinterpret_cast<const u32x2*>(Bw + (g & 1) * 16 + v * 48 + 24)[lane];
                const u32x2 u3 = {f31[0], f31[1]}, u1 = {f31[2], f31[3]};
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p3), cat(u3, u1), acc[v], 0, 0, 0);
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p2), cat(f2, f2), acc[v], 0, 0, 0);
                acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p2), cat(u1, u1), acc[v], 0, 0, 0);
            }
        } else if (MODE == 3) {
            // S3 = S2 with (a) the filter fragments [u2 u1 u3] (24 B per lane) read as two overlapping 16-byte windows [u2|u1] and
            // [u1|u3]: no duplicated piece on the B side; (b) the A side as ONE 8-register tuple [v3 v1 v2 v1] whose three overlapping
            // 4-register windows are the operands [v3|v1], [v1|v2], [v2|v1]: products v3u1 + v1u3, v1u2 + v2u1, v2u2 + v1u1;
            // (c) only the five columns a half row reads
            float V[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float u0 = t[0][c], u1 = t[1][c], u2 = t[2][c], u3 = t[3][c], u4 = t[4][c];
                const float ea = __builtin_fmaf(-KB2, u2, u4), oa = __builtin_fmaf(-KB2, u1, u3);
                V[c][0] = __builtin_fmaf(KP, u0, __builtin_fmaf(KS, u2, u4));
                V[c][1] = __builtin_fmaf(KA, oa, ea);
                V[c][2] = __builtin_fmaf(-KA, oa, ea);
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float vv[4] = {V[0][v], V[1][v], V[2][v], V[3][v]};
                u32x2 p1, p2, p3;
                split3(vv, p1, p2, p3);
                typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
                const u32x8 a8 = {p3[0], p3[1], p1[0], p1[1], p2[0], p2[1], p1[0], p1[1]};
                const bf16x8 A3 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3));
                const bf16x8 A2 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 2, 3, 4, 5));
                const bf16x8 A1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 4, 5, 6, 7));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
                    const char* fp = reinterpret_cast<const char*>(Bw) + (g & 1) * 3072 + (2 * (v & 1) + nb) * 1536 + lane * 24;
                    const u32x4 B12 = *reinterpret_cast<const u32x4_a8*>(fp);
                    const u32x4 B3 = *reinterpret_cast<const u32x4_a8*>(fp + 8);
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, __builtin_bit_cast(bf16x8, B3), acc[2 * v + nb], 0, 0, 0);
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, __builtin_bit_cast(bf16x8, B12), acc[2 * v + nb], 0, 0, 0);
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, __builtin_bit_cast(bf16x8, B12), acc[2 * v + nb], 0, 0, 0);
                }
            }
        } else {
            // half a row (the points 0, 1, 2 here; 3, 4, 5 cost the same), two column blocks of 32 output channels
            float V[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float u0 = t[0][c], u1 = t[1][c], u2 = t[2][c], u3 = t[3][c], u4 = t[4][c];
                const float ea = __builtin_fmaf(-KB2, u2, u4), oa = __builtin_fmaf(-KB2, u1, u3);
                V[c][0] = __builtin_fmaf(KP, u0, __builtin_fmaf(KS, u2, u4));
                V[c][1] = __builtin_fmaf(KA, oa, ea);
                V[c][2] = __builtin_fmaf(-KA, oa, ea);
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float vv[4] = {V[0][v], V[1][v], V[2][v], V[3][v]};
                u32x2 p1, p2, p3;
                split3(vv, p1, p2, p3);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const u32x4 f31 = __builtin_bit_cast(u32x4, Bw[(g & 1) * 32 + (2 * v + nb) * 48 + lane]);
                    const u32x2 f2 = reinterpret_cast<const u32x2*>(Bw + (g & 1) * 16 + (2 * v + nb) * 48 + 24)[lane];
                    const u32x2 u3 = {f31[0], f31[1]}, u1 = {f31[2], f31[3]};
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p3), cat(u3, u1), acc[2 * v + nb], 0, 0, 0);
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p2), cat(f2, f2), acc[2 * v + nb], 0, 0, 0);
                    acc[2 * v + nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(p1, p2), cat(u1, u1), acc[2 * v + nb], 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <typename K>
double run(K kern, float* out, unsigned long long* cyc, int groups) {
    const size_t lds = (size_t)(3 * 1536 + 12 * 384) * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(768), lds, 0, out, cyc, groups);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 0; }
    unsigned long long h[16];
    hipMemcpy(h, cyc + 16 * 100, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < 12; ++i) mx = h[i] > mx ? (double)h[i] : mx;
    return mx;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int groups = 256;
    const double f = run(kmix<0>, out, cyc, groups) / groups;
    const double s1 = run(kmix<1>, out, cyc, groups) / groups;
    const double s2 = run(kmix<2>, out, cyc, groups) / groups;
    const double s3 = run(kmix<3>, out, cyc, groups) / groups;
    // s_memtime ticks at 100 MHz on gfx950 (constant clock), the shader at ~2.4 GHz: report ratios and ticks
    printf("per 8-channel group and SIMD (3 waves), s_memtime ticks:\n");
    printf("  F  fp32 32x32x2, 72 MFMAs x 64 pipe cycles           : %8.2f  (pipe-bound floor 4608 cycles)\n", f);
    printf("  S1 bf16x3 K-folded, (row, 32 couts), 54 MFMAs x 32   : %8.2f  ratio F / S1 = %.2f  (floor 1728 cycles)\n", s1, f / s1);
    printf("  S2 bf16x3 K-folded, (row, half, 64 couts), 54 x 32   : %8.2f  ratio F / S2 = %.2f\n", s2, f / s2);
    printf("  S3 = S2, overlapping operand windows, 5 columns        : %8.2f  ratio F / S3 = %.2f\n", s3, f / s3);
    return 0;
}
