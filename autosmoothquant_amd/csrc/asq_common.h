// Shared device/host helpers for libasq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>

#include "../../include/asq_hip.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------
void asq_set_error(const char *fmt, ...);
int asq_debug_sync();  // env ASQ_DEBUG_SYNC=1

#define ASQ_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            asq_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

static inline int asq_after_launch(hipStream_t s, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && asq_debug_sync()) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        asq_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return ASQ_OK;
}

static inline size_t asq_dtype_size(int dt) { return dt == ASQ_F32 ? 4 : 2; }

// ---------------------------------------------------------------------------------
// device: logical-dtype conversions (one IEEE rounding, as ATen's CPU kernels do)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if (f != f) return 0x7FC0;  // canonical NaN, as c10::BFloat16
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t b)
{
    return __half2float(__ushort_as_half(b));
}

__device__ __forceinline__ uint16_t f32_to_f16_bits(float f)
{
    // Opaque barrier: without it LLVM folds `fptrunc(fmul a, b)` into v_fma_mixlo_f16 a, b, 0,
    // which rounds the EXACT product once to fp16 (and turns -0 into +0) -- not the
    // "round to fp32, then to fp16" the reference's separate ATen ops perform.  Caught by the
    // bit-exact parity tests (-144951 * 0.0040789647 = -591.25000126 -> -591.5 instead of -591.0).
    asm volatile("" : "+v"(f));
    return __half_as_ushort(__float2half_rn(f));
}

template <int DT> struct ElemT;
template <> struct ElemT<ASQ_F32> {
    using type = float;
    static constexpr int VEC = 4;  // elements per 16-byte vector
    __device__ static __forceinline__ float load(float v) { return v; }
    __device__ static __forceinline__ float round(float v) { return v; }
    __device__ static __forceinline__ float store(float v) { return v; }
};
template <> struct ElemT<ASQ_F16> {
    using type = uint16_t;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(uint16_t v) { return f16_bits_to_f32(v); }
    __device__ static __forceinline__ float round(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
    __device__ static __forceinline__ uint16_t store(float v) { return f32_to_f16_bits(v); }
};
template <> struct ElemT<ASQ_BF16> {
    using type = uint16_t;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(uint16_t v) { return bf16_bits_to_f32(v); }
    __device__ static __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    __device__ static __forceinline__ uint16_t store(float v) { return f32_to_bf16_bits(v); }
};

// .round().clamp(-128,127).to(int8): half-to-even; NaN -> 0 (reference CPU path), +-inf saturate
__device__ __forceinline__ int quant_i8(float v)
{
    float r = rintf(v);
    r = fminf(fmaxf(r, -128.0f), 127.0f);
    return (v != v) ? 0 : (int)r;
}
