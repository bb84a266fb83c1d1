// Micro-benchmark (round 4): what the matrix pipe sustains for a given instruction MIX and occupancy, with nothing else in the
// way (no barriers, no DMA, operands in registers / LDS-resident).  Variants:
//   A  v_mfma_f32_32x32x2_f32, 3 waves per SIMD, 6 accumulators per wave (the F(4x4) kernel's shape): NV scalar VALU + NL ds_read_b64
//      per 12 MFMAs
//   B  v_mfma_f32_16x16x4_f32, 2 waves per SIMD, 36 accumulators per wave (144 registers: the 8-wave workgroup of DESIGN 7b): NV VALU +
//      NL ds_read_b64 per 72 MFMAs (= per 2304 pipe cycles, the same pipe time as 36 MFMAs of shape A)
// Prints SIMD-level pipe cycles per MFMA-equivalent of 64 cycles (64 = pipe-bound).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kA(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x2 lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x2{1.f, 2.f};
    __syncthreads();
    f32x16 a[6] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x2 w[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) w[j] = (j < NL) ? lds[(lane + 64 * j + it) & 4095] : f32x2{1.f, 1.f};
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(j + e) & 7], w[j][e], a[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int e = 0; e < 16; ++e) s += a[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// kA with the same operand bytes fetched by HALF as many, twice as wide reads (3 ds_read_b128 instead of 6 ds_read_b64)
template <int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kAw(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 1.f, 2.f};
    __syncthreads();
    f32x16 a[6] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 w[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) w[j] = lds[(lane + 64 * j + it) & 2047];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(j + e) & 7], w[j >> 1][(j & 1) * 2 + e], a[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int e = 0; e < 16; ++e) s += a[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// kA with the operand reads software-pipelined: w[j] of the NEXT iteration is read right behind the second MFMA that used w[j]
// (same registers), so no MFMA ever waits for LDS
template <int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kAp(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x2 lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x2{1.f, 2.f};
    __syncthreads();
    f32x16 a[6] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
    f32x2 w[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) w[j] = lds[(lane + 64 * j) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j & 7], w[j][0], a[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(j + 1) & 7], w[j][1], a[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            w[j] = lds[(lane + 64 * j + it + 1) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 6; ++j) for (int e = 0; e < 16; ++e) s += a[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// interleaved: one group of (NV / 72) VALU after each MFMA is not expressible at compile time for all NV; the VALU block is spread
// in 9 chunks between the 9 point blocks of 8 MFMAs
template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kB(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x2 lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x2{1.f, 2.f};
    __syncthreads();
    f32x4 a[36] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pt = 0; pt < 9; ++pt) {
            f32x2 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (pt * 4 + j < NL) ? lds[(lane + 64 * j + it + pt) & 4095] : f32x2{1.f, 1.f};
#pragma unroll
            for (int j = 0; j < NV / 9; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    a[pt * 4 + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[(pt + ks) & 7], w[nb][ks], a[pt * 4 + nb], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 36; ++j) for (int e = 0; e < 4; ++e) s += a[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <typename K>
double run(K kern, int waves, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc + 16 * 100, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? (double)h[i] : mx;
    return mx;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 400;
#define RA(NV, NL) { const double c = run(kA<NV, NL, 12>, 12, out, cyc, iters); \
    printf("A 32x32x2, 3 waves/SIMD: %3d VALU + %d ds_read_b64 per 12 MFMAs (%.1f VALU per 64 pipe cycles): %6.1f cycles per 64-cycle MFMA, pipe %.3f\n", NV, NL, NV / 12.0, c / (3.0 * iters * 12), 64.0 * 3 * iters * 12 / c); }
#define RB(NV, NL) { const double c = run(kB<NV, NL, 8>, 8, out, cyc, iters); \
    printf("B 16x16x4, 2 waves/SIMD: %3d VALU + %d ds_read_b64 per 72 MFMAs (%.1f VALU per 64 pipe cycles): %6.1f cycles per 64 pipe cycles, pipe %.3f\n", NV, NL, NV / 36.0, c / (2.0 * iters * 36), 64.0 * 2 * iters * 36 / c); }
    RA(0, 0) RA(0, 6) RA(24, 0) RA(24, 6) RA(48, 0) RA(48, 6) RA(72, 6) RA(96, 0) RA(96, 6)
#define RP(NV) { const double c = run(kAp<NV, 12>, 12, out, cyc, iters); \
    printf("A' 32x32x2, 3 waves/SIMD, operand reads pipelined: %3d VALU + 6 ds_read_b64 per 12 MFMAs: %6.1f cycles per 64-cycle MFMA, pipe %.3f\n", NV, c / (3.0 * iters * 12), 64.0 * 3 * iters * 12 / c); }
    RP(0) RP(24) RP(48) RP(96)
#define RW(NV) { const double c = run(kAw<NV, 12>, 12, out, cyc, iters); \
    printf("A'' 32x32x2, 3 waves/SIMD, 3 ds_read_b128 instead of 6 b64: %3d VALU per 12 MFMAs: %6.1f cycles per 64-cycle MFMA, pipe %.3f\n", NV, c / (3.0 * iters * 12), 64.0 * 3 * iters * 12 / c); }
    RW(0) RW(24) RW(96)
    RB(0, 0) RB(0, 36) RB(72, 0) RB(72, 36) RB(144, 36) RB(216, 36) RB(288, 36)
    return 0;
}
