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

// ---- activation storage types --------------------------------------------------------------------------------
// Activations live in HBM either as fp32 or (XV2_BF16: the --precision 16 path) as bf16; every kernel computes in
// fp32 registers.  bf16_t is the raw 16-bit pattern; conversion to fp32 is a shift, back is round-to-nearest-even.
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// the value a bf16 store will hold (statistics are taken on what is actually stored)
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) {     // 4 channels = one 8-byte load
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st4(bf16_t* p, const float4& v) {
        uint2 u;
        u.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        u.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2*>(p) = u;
    }
    static __device__ __forceinline__ float round(float v) { return bf16_round(v); }
};
// vector accesses of the streaming kernels: NV float4 groups (4 * NV channels) per lane
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int NV = 1;
    static __device__ __forceinline__ void ld(const float* p, float4 (&v)[1]) { v[0] = *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st(float* p, const float4 (&v)[1]) { *reinterpret_cast<float4*>(p) = v[0]; }
};
// bf16: measured on MI355X, 8 channels (one 16-byte access) per lane LOSES against 4 channels (8 bytes) in the streaming
// BatchNorm kernels - the per-lane coefficient registers double (110-138 VGPRs, occupancy 3-4 instead of 7-8) and with
// them the bytes in flight per CU drop.  So the bf16 lane also owns 4 channels; the NV-generic kernel bodies stay.
template <> struct Vec16<bf16_t> {
    static constexpr int NV = 1;
    static __device__ __forceinline__ void ld(const bf16_t* p, float4 (&v)[1]) { v[0] = Elem<bf16_t>::ld4(p); }
    static __device__ __forceinline__ void st(bf16_t* p, const float4 (&v)[1]) { Elem<bf16_t>::st4(p, v[0]); }
};
template <typename T> __device__ __forceinline__ float4 ld4(const T* p) { return Elem<T>::ld4(p); }
template <typename T> __device__ __forceinline__ void st4(T* p, const float4& v) { Elem<T>::st4(p, v); }
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return Elem<T>::ld(p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { Elem<T>::st(p, v); }
// 4 packed bf16 (8 bytes, as loaded by a b64 buffer load) -> fp32
__device__ __forceinline__ float4 bf16x4_to_f32(unsigned lo, unsigned hi) {
    return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                       __uint_as_float(hi & 0xffff0000u));
}

// XV2_MATH_F32X3 operand split: x = h + m + l exactly, every term a bf16 (8 + 8 + 8 significant bits cover the 24 of an
// fp32; round-to-nearest at each level keeps the residuals zero-mean).  Two elements per instruction: v_cvt_pk_bf16_f32,
// two unpacks, one packed subtraction per level - 4.5 VALU per element.  Outputs are the packed bf16 planes.
__device__ __forceinline__ void split3x2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    const f2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f2 r1 = v - hf;                                   // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
    const f2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const f2 r2 = r1 - mf;                                  // exact, <= 8 significant bits left
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf2));
}
__device__ __forceinline__ void split3x4(const float4 v, uint2& h, uint2& m, uint2& l) {
    split3x2(v.x, v.y, h.x, m.x, l.x);
    split3x2(v.z, v.w, h.y, m.y, l.y);
}

// Scaled-fp16 pair ("F16X2", the two-plane form of the F32X3 kernels): x * s = h + m with h = fp16(x * s), m = fp16(x * s - h),
// s a power of two taken from the tensor's max |x| so that |x * s| < 2^15.  11 + 11 significant bits: every element within 2^18
// of the tensor's maximum is represented to 2^-22 relative, smaller ones to 2^-39 of the maximum; three fp16 MFMAs per product
// (h*h, h*m, m*h; the dropped m*m is <= 2^-22) instead of six bf16 ones.  The maximum comes from the PRODUCER of the tensor
// (64 slots of |x| bit patterns, atomicMax per block) - without it a launch stays on the three-plane bf16 form, which needs no
// range information.  A value above the recorded maximum (a stale slot) would overflow to Inf and is therefore loud.
__device__ __forceinline__ void split2hx2(float a, float b, unsigned& h, unsigned& m) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const h2 hh = __builtin_convertvector(v, h2);
    const f2 r = v - __builtin_convertvector(hh, f2);        // exact
    const h2 mm = __builtin_convertvector(r, h2);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
}
__device__ __forceinline__ void split2hx4(const float4 v, float s, uint2& h, uint2& m) {
    split2hx2(v.x * s, v.y * s, h.x, m.x);
    split2hx2(v.z * s, v.w * s, h.y, m.y);
}
constexpr int AMAX_SLOTS = 64;
constexpr int AMAX_STRIDE = 32;      // uint32 between two slots: one 128-byte line each (4096 blocks' atomics on TWO lines cost the
                                     // BatchNorm apply pass +9 us of 20), 8 KB per tensor
__device__ __forceinline__ float amax_acc(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
// biased exponent of the recorded maximum, clamped so that both the scale and its inverse are normal numbers
__device__ __forceinline__ int amax_exponent(const unsigned* slots, const unsigned* slots1 = nullptr) {
    unsigned v = slots[(threadIdx.x & 63) * AMAX_STRIDE];
    if (slots1) v = max(v, slots1[(threadIdx.x & 63) * AMAX_STRIDE]);
    v = wave_max_u(v);
    const int e = (int)((v >> 23) & 0xffu);
    return __builtin_amdgcn_readfirstlane(min(max(e, 16), 240));
}
__device__ __forceinline__ float amax_scale(int e) { return __uint_as_float((unsigned)(268 - e) << 23); }      // |x| * s < 2^15
__device__ __forceinline__ float amax_inv(int e) { return __uint_as_float((unsigned)(e - 14) << 23); }
// block-level record of max |x| (bits of |x| order like the values; NaN / Inf order above every finite number)
__device__ __forceinline__ void amax_record(unsigned* slots, float local_abs_max, float* smem4, unsigned slot_hint = 0xffffffffu) {
    unsigned v = wave_max_u(__float_as_uint(local_abs_max) & 0x7fffffffu);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned*>(smem4)[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) v = max(v, reinterpret_cast<unsigned*>(smem4)[w]);
        unsigned* d = slots + ((slot_hint == 0xffffffffu ? blockIdx.x : slot_hint) & (AMAX_SLOTS - 1)) * AMAX_STRIDE;
        // fire and forget (no returned value, no preceding load: a block must not wait ~2 us for a device-scope round trip -
        // with one the BatchNorm apply pass ran 20 -> 42 us); zero maxima (all-zero blocks) are not sent
        if (v) __hip_atomic_fetch_max(d, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wave-level record (kernels whose waves never meet: no LDS scratch, no barrier): one fire-and-forget atomic per wave
__device__ __forceinline__ void amax_record_wave(unsigned* slots, float local_abs_max, unsigned slot_hint) {
    const unsigned v = wave_max_u(__float_as_uint(local_abs_max) & 0x7fffffffu);
    if ((threadIdx.x & 63) == 0 && v)
        __hip_atomic_fetch_max(slots + (slot_hint & (AMAX_SLOTS - 1)) * AMAX_STRIDE, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef XV2_T0
#define XV2_T0 0   // emulation switch: 3 drops the three smallest product terms of the F32X3 kernels
#endif
#define XV2_CHECK_DTYPE(dt) XV2_CHECK_ARG((dt) == XV2_F32 || (dt) == XV2_BF16, "unknown activation dtype %d", (int)(dt))
// run `call` with T bound to the storage type named by dtype
#define XV2_DISPATCH_DTYPE(dt, ...)             \
    do {                                        \
        if ((dt) == XV2_BF16) {                 \
            typedef xv2::bf16_t T;              \
            __VA_ARGS__;                        \
        } else {                                \
            typedef float T;                    \
            __VA_ARGS__;                        \
        }                                       \
    } while (0)

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
