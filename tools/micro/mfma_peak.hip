// Micro-benchmark (round 4): the fp32 matrix rate the chip SUSTAINS over tens of milliseconds - a pure stream of
// v_mfma_f32_32x32x2_f32 (6 independent accumulators per wave, 3 waves per SIMD, every CU busy, nothing else) timed with HIP events,
// against the 157.3 TFLOP/s of MI355X_MICROARCH.md (256 CUs x 4 SIMDs x 2.4 GHz x 64 flops/cycle).  Tells how much of the distance
// between roofline.frac and 1 is clock / power and how much is the kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(768) void k(float* out, int iters) {
    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][e] = 0.f;
    const float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f + (float)(threadIdx.x & 3);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < 6; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[v], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[v][e];
    if (s == 12345.678f) out[0] = s;                     // (keeps the accumulators alive)
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {2000, 20000, 100000, 100000}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus), dim3(768), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 12 * iters * 24 * 4096.0;      // 12 waves x 24 MFMAs per iteration x 32*32*2*2
        printf("%d CUs, clock %d MHz (property), %6d iterations: %8.3f ms, %7.2f TFLOP/s fp32 MFMA = %.3f of 157.3\n", cus, pr.clockRate / 1000, iters,
               ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
    return 0;
}
