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

// ---- cross-lane exchange on the VALU (no ds_bpermute) -------------------------------------------------------------------------
// `__shfl_xor(x, o)` compiles to ds_bpermute_b32: an LDS-pipe round trip (~120 cycles) on whatever dependency chain it sits in --
// four per 16-query block and key tile of the flash kernels' online softmax, six per wave reduction (22 block reductions in the
// token-selection kernel), eight in a row of LayerNorm statistics.  lane_xor<O>(x) returns the SAME value (x of lane ^ O) from
// VALU instructions: O = 32 / 16 by gfx950's v_permlane32_swap / v_permlane16_swap (with both operands = x the swap leaves x of
// the lower half / even rows in one result and x of the upper half / odd rows in the other; a select by the lane's own half /
// row parity picks the partner's), O = 8 by DPP row_ror:8, O = 4 by two bank-masked DPP row shifts, O = 2 / 1 by DPP quad_perm.
// Callers keep their operand order (own op partner), so every reduction is bit-identical to its shuffle form by construction;
// the helpers themselves are checked lane by lane against __shfl_xor on hardware (tests/test_gpu_kernels.py::
// test_lane_xor_helpers_match_shuffles).
__device__ __forceinline__ int swx_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int O> __device__ __forceinline__ unsigned lane_xor_u32(unsigned x, int lane)
{
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "lane_xor: one butterfly step");
#ifdef SWX_LANE_XOR_BPERMUTE      // A/B build only (stable_ts_amd/build.py::build_variant): the ds_bpermute form of rounds 1-5
    return (unsigned)__shfl_xor((int)x, O, 64);
#endif
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // r[0] = x of lanes 0..31, r[1] = x of lanes 32..63
        return (lane & 32) ? r[0] : r[1];
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);      // r[0] = x of the even row, r[1] = of the odd row
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (O == 8) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);           // row_ror:8
    } else if constexpr (O == 4) {
        int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0x5, false);                     // banks 0, 2 <- lane + 4 (row_shl:4)
        t = __builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);                         // banks 1, 3 <- lane - 4 (row_shr:4)
        return (unsigned)t;
    } else if constexpr (O == 2) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);            // quad_perm [2, 3, 0, 1]
    } else {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);            // quad_perm [1, 0, 3, 2]
    }
}
template <int O> __device__ __forceinline__ float lane_xor(float x, int lane) { return __uint_as_float(lane_xor_u32<O>(__float_as_uint(x), lane)); }
template <int O> __device__ __forceinline__ int lane_xor(int x, int lane) { return (int)lane_xor_u32<O>((unsigned)x, lane); }
template <int O> __device__ __forceinline__ double lane_xor(double x, int lane)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = lane_xor_u32<O>((unsigned)u, lane), hi = lane_xor_u32<O>((unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the two cross-row steps of a COMMUTATIVE reduction without the select: v_max_f32 and the IEEE add commute bit for bit
__device__ __forceinline__ float lane_xor16_max(float x) {
#ifdef SWX_LANE_XOR_BPERMUTE
    return fmaxf(x, __shfl_xor(x, 16, 64));
#endif
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_xor32_max(float x) {
#ifdef SWX_LANE_XOR_BPERMUTE
    return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_xor16_add(float x) {
#ifdef SWX_LANE_XOR_BPERMUTE
    return x + __shfl_xor(x, 16, 64);
#endif
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_xor32_add(float x) {
#ifdef SWX_LANE_XOR_BPERMUTE
    return x + __shfl_xor(x, 32, 64);
#endif
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The same two steps through the LDS crossbar (ds_bpermute), for the HBM-streaming attention kernels: there the exchange is a few
// instructions beside a memory wait, the asynchronous LDS operation costs the wave no VALU issue slot, and the VALU form measured
// SLOWER (strict-f32 decode cross-attention 66.2 -> 78.5 us, fp16 25.9 -> 26.0 us: profiles/r06_c23_f32_lane_xor_by_kernel.txt).
// -DSWX_XATTN_VALU (A/B build) puts them back on the VALU.
__device__ __forceinline__ float lane_xor16_max_lds(float x) {
#ifdef SWX_XATTN_VALU
    return lane_xor16_max(x);
#else
    return fmaxf(x, __shfl_xor(x, 16, 64));
#endif
}
__device__ __forceinline__ float lane_xor32_max_lds(float x) {
#ifdef SWX_XATTN_VALU
    return lane_xor32_max(x);
#else
    return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
}
__device__ __forceinline__ float lane_xor16_add_lds(float x) {
#ifdef SWX_XATTN_VALU
    return lane_xor16_add(x);
#else
    return x + __shfl_xor(x, 16, 64);
#endif
}
__device__ __forceinline__ float lane_xor32_add_lds(float x) {
#ifdef SWX_XATTN_VALU
    return lane_xor32_add(x);
#else
    return x + __shfl_xor(x, 32, 64);
#endif
}

// wave reductions, butterfly from 32 down to 1 (the order every caller's bit-identity claims were made with)
__device__ __forceinline__ float wave_max(float v) {
    const int lane = swx_lane_id();
    v = fmaxf(v, lane_xor<32>(v, lane)); v = fmaxf(v, lane_xor<16>(v, lane)); v = fmaxf(v, lane_xor<8>(v, lane));
    v = fmaxf(v, lane_xor<4>(v, lane)); v = fmaxf(v, lane_xor<2>(v, lane)); v = fmaxf(v, lane_xor<1>(v, lane));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    const int lane = swx_lane_id();
    v += lane_xor<32>(v, lane); v += lane_xor<16>(v, lane); v += lane_xor<8>(v, lane);
    v += lane_xor<4>(v, lane); v += lane_xor<2>(v, lane); v += lane_xor<1>(v, lane);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    const int lane = swx_lane_id();
    v += lane_xor<32>(v, lane); v += lane_xor<16>(v, lane); v += lane_xor<8>(v, lane);
    v += lane_xor<4>(v, lane); v += lane_xor<2>(v, lane); v += lane_xor<1>(v, lane);
    return v;
}

// exact GELU (erf form), f32 -- upstream nn.GELU()
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// The same function for TWO values at once, bit for bit (round 6).  `erff` is the device library's: a divergent branch at |z| = 1
// (z = x / sqrt 2) between a degree-6 polynomial in z^2 and 1 - exp(-t(|z|)) with the library's extended-precision exp; with
// real activations both sides run in every wave (~38 VALU instructions per element; the MLP's first projection applies it to
// 4.9 G elements per headline pass with all eight waves of a CU in the epilogue at once).  Here the library's operation sequence --
// read off the compiled ISA: the same constants, the same fused multiply-adds in the same order -- is written once for a pair on
// packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and instruction), both sides
// evaluated and selected: 26 VALU instructions per element.  Equality with gelu_erf is checked EXHAUSTIVELY on hardware: all 2^32
// bit patterns (tests/test_gpu_kernels.py::test_gelu_pair_is_bit_identical_for_every_float).  -DSWX_GELU_SCALAR (A/B build) = gelu_erf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 swx_k2(unsigned u) { const float f = __uint_as_float(u); return (f32x2){f, f}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x)
{
#ifdef SWX_GELU_SCALAR
    return (f32x2){gelu_erf(x[0]), gelu_erf(x[1])};
#else
    const f32x2 z = x * swx_k2(0x3f3504f3u);
    const f32x2 az = __builtin_elementwise_abs(z);
    // |z| >= 1 (and NaN): t = |z| + |z| P(|z|), erf = 1 - exp(-t)
    f32x2 p = __builtin_elementwise_fma(az, swx_k2(0x378e98abu), swx_k2(0xb9c68948u));
    p = __builtin_elementwise_fma(az, p, swx_k2(0x3b7cd369u));
    p = __builtin_elementwise_fma(az, p, swx_k2(0xbcc618b2u));
    p = __builtin_elementwise_fma(az, p, swx_k2(0x3dda74e4u));
    p = __builtin_elementwise_fma(az, p, swx_k2(0x3f228afdu));
    p = __builtin_elementwise_fma(az, p, swx_k2(0x3e03c728u));
    const f32x2 t = __builtin_elementwise_fma(az, p, az);
    f32x2 a = t * swx_k2(0xbfb8aa3bu);                                      // -t log2(e), head
    f32x2 b = __builtin_elementwise_fma(t, swx_k2(0xbfb8aa3bu), -a);        // ... its rounding error
    f32x2 r;
    r[0] = __builtin_rintf(a[0]); r[1] = __builtin_rintf(a[1]);
    b = __builtin_elementwise_fma(t, swx_k2(0xb2a5705fu), b);               // ... and the tail of log2(e)
    a = a - r;
    a = a + b;
    f32x2 big;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float e = __builtin_amdgcn_exp2f(a[i]);
        e = __builtin_ldexpf(e, (int)r[i]);
        e = !(__uint_as_float(0x42ce8ed0u) < t[i]) ? e : 0.f;               // underflow of exp(-t)
        e = !(__uint_as_float(0xc2b17218u) > t[i]) ? e : __builtin_inff();
        big[i] = e;
    }
    big = swx_k2(0x3f800000u) - big;
    // |z| < 1: erf = |z| + |z| Q(z^2)
    const f32x2 t2 = z * z;
    f32x2 q = __builtin_elementwise_fma(swx_k2(0xba1345e1u), t2, swx_k2(0x3ba10414u));
    q = __builtin_elementwise_fma(t2, q, swx_k2(0xbcdac9b8u));
    q = __builtin_elementwise_fma(t2, q, swx_k2(0x3de703beu));
    q = __builtin_elementwise_fma(t2, q, swx_k2(0xbec09330u));
    q = __builtin_elementwise_fma(t2, q, swx_k2(0x3e0375d0u));
    const f32x2 small = __builtin_elementwise_fma(az, q, az);
    f32x2 erf;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float res = !(az[i] < 1.0f) ? big[i] : small[i];
        erf[i] = __uint_as_float((__float_as_uint(res) & 0x7fffffffu) | (__float_as_uint(z[i]) & 0x80000000u));    // copysign(res, z)
    }
    return (x * swx_k2(0x3f000000u)) * (erf + swx_k2(0x3f800000u));
#endif
}
// n (even) values in place
template <int N> __device__ __forceinline__ void gelu_erf_n(float (&v)[N])
{
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int e = 0; e < N; e += 2) { f32x2 t = {v[e], v[e + 1]}; t = gelu_erf2(t); v[e] = t[0]; v[e + 1] = t[1]; }
}

// monotone float <-> uint mapping for atomicMax on floats
__device__ __forceinline__ unsigned f32_to_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
