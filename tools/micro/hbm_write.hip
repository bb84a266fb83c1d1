// Micro-benchmark (round 4): what a pure streaming WRITE (and a copy) sustains on this chip - the roof of conv_first / the output stages.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_write.hip -o /tmp/hbm_write && /tmp/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void wr(f32x4* out, size_t n) {
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void cp(const f32x4* in, f32x4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const f32x4 v = __builtin_nontemporal_load(in + i);
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    f32x4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t = [&](const char* name, auto f, double gb) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.3f ms per pass, %6.2f TB/s\n", name, ms / 5, gb * 5 / ms / 1e9 * 1e3 / 1e3);
    };
    for (int blocks : {2048, 8192, 32768}) {
        printf("grid %d x 256:\n", blocks);
        t("  write 8 GiB, plain stores", [&] { hipLaunchKernelGGL(wr<false>, dim3(blocks), dim3(256), 0, 0, b, n); }, (double)bytes);
        t("  write 8 GiB, non-temporal", [&] { hipLaunchKernelGGL(wr<true>, dim3(blocks), dim3(256), 0, 0, b, n); }, (double)bytes);
        t("  copy 8 + 8 GiB, nt load + plain", [&] { hipLaunchKernelGGL(cp<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        t("  copy 8 + 8 GiB, nt load + nt", [&] { hipLaunchKernelGGL(cp<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    }
    return 0;
}
