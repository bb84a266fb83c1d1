// Small device-side helpers shared by the kernel sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ecseg_hip.h"

namespace ecseg {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
    switch (act) {
        case ECSEG_ACT_RELU: return v > 0.f ? v : 0.f;
        case ECSEG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ECSEG_ACT_LEAKY: return v > 0.f ? v : alpha * v;
        case ECSEG_ACT_TANH: return tanhf(v);
        case ECSEG_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
        default: return v;
    }
}

// Four values at once with ONE uniform branch on the activation code: the per-element switch of apply_act costs a scalar
// compare / branch chain per value, which dominated the output stages of the MFMA kernels (ReLU and linear are the
// activations of every convolution of a U-Net).
__device__ __forceinline__ f32x4 apply_act4(f32x4 v, int act, float alpha) {
    if (act == ECSEG_ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    } else if (act != ECSEG_ACT_LINEAR) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], act, alpha);
    }
    return v;
}

// T1: give every XCD (blocks with equal blockIdx % 8 share one L2) a contiguous range of logical block ids.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Host side: "do this once per device" latch for per-device function attributes (hipFuncSetAttribute applies to the
// current device only; one process may drive several handles on different GPUs).
struct DeviceOnce {
    unsigned long long done = 0;       // bit d: device d has been set up
    int dev = 0;
    bool first() {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done >> dev & 1ull) return false;
        done |= 1ull << dev;
        return true;
    }
    void reset() { if (dev >= 0 && dev < 64) done &= ~(1ull << dev); }
};

}  // namespace ecseg
