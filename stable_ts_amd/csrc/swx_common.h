// swx_common.h -- shared device helpers for the gfx950 kernels of libswx.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>

#define SWX_WAVE 64

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SWX_CHECK_LAUNCH() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return -100 - (int)_e; } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }

// 8 consecutive elements (16-byte aligned for f16, 32-byte span for f32) -> f32
template <typename T> __device__ __forceinline__ void load8(const T *p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<f16>(const f16 *p, float (&o)[8]) {
    const f16x8 v = *(const f16x8 *)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
template <> __device__ __forceinline__ void load8<float>(const float *p, float (&o)[8]) {
    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact GELU (erf form), f32 -- upstream nn.GELU()
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// monotone float <-> uint mapping for atomicMax on floats
__device__ __forceinline__ unsigned f32_to_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
