// Shared helpers for the gfx950 kernels behind include/xv2.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/xv2.h"

namespace xv2 {

void set_error(const char* fmt, ...);
// bench-time profiler (errors.cpp): no-ops unless xv2_prof_enable(1) was called
int prof_register(const char* name);
void prof_begin(int kid, double flops, double algorithmic_bytes, hipStream_t stream);
void prof_end(hipStream_t stream);

#define XV2_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            xv2::set_error(__VA_ARGS__);         \
            return XV2_EINVAL;                   \
        }                                        \
    } while (0)

#define XV2_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            xv2::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                    \
            return XV2_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define XV2_CHECK_LAUNCH() XV2_CHECK_HIP(hipGetLastError())

__host__ __device__ static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == XV2_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == XV2_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    if (act == XV2_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// derivative of the activation expressed through its OUTPUT z
__device__ __forceinline__ float act_grad_from_output(float z, int act) {
    if (act == XV2_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == XV2_ACT_LEAKY) return z > 0.f ? 1.f : 0.01f;
    if (act == XV2_ACT_SIGMOID) return z * (1.f - z);
    return 1.f;
}

// derivative of the activation expressed through its INPUT u (pre-activation)
__device__ __forceinline__ float act_grad_from_pre(float u, int act) {
    if (act == XV2_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (act == XV2_ACT_LEAKY) return u > 0.f ? 1.f : 0.01f;
    if (act == XV2_ACT_SIGMOID) {
        const float z = 1.f / (1.f + expf(-u));
        return z * (1.f - z);
    }
    return 1.f;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace xv2
