// Small device-side helpers shared by the kernel sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/ecseg_hip.h"

namespace ecseg {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Activations of the Winograd kernels' output stages (conv_wino4 / conv_wino16: register-critical - the F(4x4) head variant
// sits at its 168-VGPR ceiling): codes 0..6 with ELU's alpha = 1; api.hip sends anything else to the other kernels
// (act_core_ok).
__device__ __forceinline__ float apply_act_core(float v, int act, float alpha) {
    switch (act) {
        case ECSEG_ACT_RELU: return v > 0.f ? v : 0.f;
        case ECSEG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ECSEG_ACT_LEAKY: return v > 0.f ? v : alpha * v;
        case ECSEG_ACT_TANH: return tanhf(v);
        case ECSEG_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
        default: return v;
    }
}
__device__ __forceinline__ f32x4 apply_act4_core(f32x4 v, int act, float alpha) {
    if (act == ECSEG_ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    } else if (act != ECSEG_ACT_LINEAR) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = apply_act_core(v[c], act, alpha);
    }
    return v;
}

// The activations a convolution kernel's output stage may be asked for (codes 0..7; keras_plan never fuses the rarer ones
// into a convolution - their libm code would sit in every MFMA kernel's output stage and count against its registers).
__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
    switch (act) {
        case ECSEG_ACT_RELU: return v > 0.f ? v : 0.f;
        case ECSEG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ECSEG_ACT_LEAKY: return v > 0.f ? v : alpha * v;
        case ECSEG_ACT_TANH: return tanhf(v);
        case ECSEG_ACT_ELU: return v > 0.f ? v : alpha * (expf(v) - 1.f);
        case ECSEG_ACT_RELU_CLIP: return fminf(fmaxf(v, 0.f), alpha);
        default: return v;
    }
}

// Every activation code: the element-wise kernels (ACT / AFFINE / ADD ops, depthwise convolutions).
__device__ __forceinline__ float apply_act_ext(float v, int act, float alpha) {
    switch (act) {
        case ECSEG_ACT_SWISH: return v / (1.f + expf(-v));
        case ECSEG_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
        case ECSEG_ACT_SOFTPLUS: return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
        case ECSEG_ACT_SELU: return 1.05070098735548f * (v > 0.f ? v : 1.67326324235438f * (expf(v) - 1.f));
        case ECSEG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case ECSEG_ACT_EXP: return expf(v);
        case ECSEG_ACT_SOFTSIGN: return v / (1.f + fabsf(v));
        default: return apply_act(v, act, alpha);
    }
}
__device__ __forceinline__ f32x4 apply_act_ext4(f32x4 v, int act, float alpha) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = apply_act_ext(v[c], act, alpha);
    return v;
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
// current device only; one process may drive several handles on different GPUs, from several threads).  `run(setup)`
// holds the latch's mutex across check, setup and set, and marks the device only after `setup` has succeeded: a second
// thread can neither skip a setup that is still running (and launch a kernel whose dynamic-LDS limit has not been raised
// yet) nor see a half-written mask.
struct DeviceOnce {
    std::mutex mu;
    unsigned long long done = 0;       // bit d: device d has been set up
    template <typename F>
    hipError_t run(F&& setup) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return setup();
        std::lock_guard<std::mutex> lock(mu);
        if (done >> dev & 1ull) return hipSuccess;
        const hipError_t e = setup();
        if (e == hipSuccess) done |= 1ull << dev;
        return e;
    }
};

}  // namespace ecseg
