// Micro-benchmark: three waves per SIMD (768-thread workgroup), each running the F(4x4) stage pattern: 12 fp32 MFMAs on 6
// accumulators + NV scalar VALU ops + NL ds_read_b64 per stage.  Prints the SIMD-level cycles per MFMA (64 = pipe-bound).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, unsigned long long* cyc, int iters) {
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

template <int NV, int NL, int WAVES>
void run(float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc + 16 * 100, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < WAVES; ++i) mx = h[i] > mx ? (double)h[i] : mx;
    const double per_simd = (WAVES / 4.0) * iters * 12.0;           // MFMAs per SIMD
    printf("%2d waves/CU, %2d VALU + %d ds_read_b64 per 12 MFMAs: %6.1f cycles per MFMA on the SIMD (64 = pipe-bound)\n", WAVES, NV, NL,
           mx / per_simd);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 1000;
    run<0, 0, 4>(out, cyc, iters);
    run<0, 0, 12>(out, cyc, iters);
    run<24, 6, 4>(out, cyc, iters);
    run<24, 6, 12>(out, cyc, iters);
    run<48, 6, 12>(out, cyc, iters);
    run<96, 6, 12>(out, cyc, iters);
    run<144, 6, 12>(out, cyc, iters);
    run<96, 6, 8>(out, cyc, iters);
    return 0;
}
