// Micro-benchmark: does a VALU / LDS stream on the second wave of a SIMD slow the first wave's fp32 MFMA stream?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_coexec.hip -o mfma_valu_coexec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    lds[threadIdx.x + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    if (wave < 4) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {};
        float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < (PRIO == 2 ? 0 : iters); ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a5, 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e] + a4[e] + a5[e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v0 = lane, v1 = lane + 1.f, v2 = lane + 2.f, v3 = lane + 3.f, v4 = 1.f, v5 = 2.f, v6 = 3.f, v7 = 4.f;
        f32x4 acc4 = {};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const int n = iters * 6;                      // one "unit" per MFMA of the partner wave
        if (MODE == 1) {                              // 8 scalar fmas per partner MFMA
            for (int i = 0; i < n; ++i) {
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
                v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
                v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f);
                v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            }
        } else if (MODE == 2) {                       // 8 integer ops per partner MFMA
            int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = 5, i5 = 6, i6 = 7, i7 = 8;
            for (int i = 0; i < n; ++i) {
                i0 = i0 * 3 + 1; i1 = i1 * 3 + 1; i2 = i2 * 3 + 1; i3 = i3 * 3 + 1;
                i4 = i4 * 3 + 1; i5 = i5 * 3 + 1; i6 = i6 * 3 + 1; i7 = i7 * 3 + 1;
                asm volatile("" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
            }
            v0 = (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
        } else if (MODE == 3) {                       // 2 ds_read_b128 per partner MFMA
            for (int i = 0; i < n; ++i) {
                f32x4 a = lds[(threadIdx.x + i) & 1023], b = lds[(threadIdx.x + 2 * i + 7) & 1023];
                acc4 += a + b;
            }
        } else if (MODE == 4) {                       // 4 packed fmas (same flops as MODE 1)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7};
            const f32x2 m = {1.0001f, 1.0001f}, c = {0.5f, 0.5f};
            for (int i = 0; i < n; ++i) {
                p0 = p0 * m + c; p1 = p1 * m + c; p2 = p2 * m + c; p3 = p3 * m + c;
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            }
            v0 = p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + acc4[0] + acc4[1] + acc4[2] + acc4[3];
        if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    }
}

template <int MODE, int PRIO = 0>
void run(const char* name, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s MFMA wave: %7.1f cyc/MFMA   partner wave: %7.1f cyc per MFMA-equivalent unit\n", name,
           (double)h[0] / (iters * 6.0), (double)h[4] / (iters * 6.0));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    run<0>("partner idle", out, cyc, iters);
    run<1>("partner 8 v_fma_f32 / MFMA", out, cyc, iters);
    run<2>("partner 8 int mad / MFMA", out, cyc, iters);
    run<3>("partner 2 ds_read_b128 / MFMA", out, cyc, iters);
    run<4>("partner 4 v_pk_fma_f32 / MFMA", out, cyc, iters);
    run<1, 2>("8 v_fma_f32, NO MFMAs", out, cyc, iters);
    run<3, 2>("2 ds_read_b128, NO MFMAs", out, cyc, iters);
    run<4, 2>("4 v_pk_fma_f32, NO MFMAs", out, cyc, iters);
    run<1, 1>("8 v_fma_f32, partner prio 3", out, cyc, iters);
    run<3, 1>("2 ds_read_b128, partner prio 3", out, cyc, iters);
    run<4, 1>("4 v_pk_fma_f32, partner prio 3", out, cyc, iters);
    return 0;
}
