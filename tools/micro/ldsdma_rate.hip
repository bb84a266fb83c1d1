// Micro-benchmark (round 6): what the LDS-DMA path of one CU sustains from L2-resident data - the ceiling under the filter stream of
// conv_wino4s_kernel (110 KB per 8-channel group and workgroup).  W waves per workgroup (one workgroup per CU) stream 1-KB pieces
// (global_load_lds_dwordx4, 64 lanes x 16 contiguous bytes) out of a per-workgroup window of `win` bytes into private LDS rings,
// at most D pieces in flight per wave; no consumer.  Prints GB/s per CU and TB/s for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D>
__global__ __launch_bounds__(768) void stream(const char* src, long win, int pieces, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned long long base = (unsigned long long)(size_t)src + (unsigned long long)blockIdx.x * win;
    const unsigned l16 = lane * 16u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < pieces; ++k) {
        const unsigned long long g = base + ((unsigned long long)(k * nw + wave) * 1024) % win;
        const unsigned dst = lds_base + (unsigned)((wave * 8 + (k & 7)) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(l16), "s"(dst), "s"(g) : "memory");
        if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int D>
void run(int waves, long win, const char* src, unsigned long long* cyc) {
    const int pieces = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 8 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<D>, dim3(256), dim3(waves * 64), 12 * 8 * 1024, 0, src, win, pieces, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<D>, dim3(256), dim3(waves * 64), 12 * 8 * 1024, 0, src, win, pieces, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * waves * pieces * 1024;
    printf("%2d waves/CU, %d in flight per wave, window %5ld KB per CU: %6.1f GB/s per CU, %5.2f TB/s chip (%.3f ms)\n", waves, D, win >> 10,
           bytes / 256 / ms / 1e6, bytes / ms / 1e9, ms);
}

int main() {
    char* src; unsigned long long* cyc;
    const long total = 1l << 30;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&cyc, 256 * 16 * 8);
    for (long win : {36l << 10, 512l << 10, 4l << 20}) {
        run<2>(12, win, src, cyc); run<4>(12, win, src, cyc); run<8>(12, win, src, cyc);
        run<4>(4, win, src, cyc); run<8>(4, win, src, cyc);
    }
    return 0;
}
