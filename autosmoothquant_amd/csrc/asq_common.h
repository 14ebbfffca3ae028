// Shared device/host helpers for libasq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <utility>
#include <type_traits>

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

// ASQ_ROCTX=1: roctx ranges around the C-ABI entry points (quantise / GEMM / forward), so that `rocprofv3 --marker-trace` attributes kernels to the call that
// launched them.  The marker library is dlopen'ed on first use; without the variable (the default) a range is one predictable branch.
void asq_range_push(const char *name);
void asq_range_pop();
int asq_roctx_enabled();
struct AsqRange {
    bool on;
    explicit AsqRange(const char *name) : on(asq_roctx_enabled() != 0) { if (on) asq_range_push(name); }
    ~AsqRange() { if (on) asq_range_pop(); }
    AsqRange(const AsqRange &) = delete;
    AsqRange &operator=(const AsqRange &) = delete;
};

// the one-launch forward for decode-sized inputs (asq_gemm_fq.hip); *launched = 0: not this shape, run quantiser + GEMM
int asq_try_fused_forward(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int act_mode, float quant_scale, float s_scalar,
                          const float *s_col, const float *bias, void *stream, int *launched);

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

// two conversions + the pack in one instruction (gfx950 v_cvt_pk_bf16_f32).  The hardware keeps a NaN's sign and payload bits; c10::BFloat16
// (and f32_to_bf16_bits above) return the canonical 0x7FC0, so NaN lanes are patched.  Equal to the software rounding on all 2^32 inputs.
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi)
{
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b_ __attribute__((ext_vector_type(2)));
    uint32_t bits = __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f_){lo, hi}, v2b_));
    if (lo != lo) bits = (bits & 0xFFFF0000u) | 0x7FC0u;
    if (hi != hi) bits = (bits & 0x0000FFFFu) | 0x7FC00000u;
    return bits;
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
    asm("" : "+v"(f));  // (not volatile: the value dependency is all that is needed; volatile asms would also pin their mutual order)
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

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// .round().clamp(-128,127).to(int8): half-to-even; NaN -> 0 (reference CPU path), +-inf saturate
// Three instructions: v_rndne_f32, v_cvt_i32_f32 (the hardware conversion saturates out-of-range values and +-inf and turns NaN
// into 0 -- the C cast is undefined there, hence the asm), v_med3_i32.
__device__ __forceinline__ int quant_i8(float v)
{
    int q;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(q) : "v"(rintf(v)));
    return q < -128 ? -128 : (q > 127 ? 127 : q);
}

// ---------------------------------------------------------------------------------
// helpers shared by the int8 and fp8 activation quantisers
// ---------------------------------------------------------------------------------
// |x| maximum with torch.max's NaN propagation, carried as BIT PATTERNS: for non-negative IEEE values
// (fp32, fp16 and bf16 alike) the unsigned-integer order of the patterns IS the numeric order, and every
// NaN pattern lies above +inf.  One v_and + one integer max per element (a packed pair for the 16-bit
// types) instead of a compare/select chain.  (Measured: the chain made the fp16 per-token quantiser
// issue-bound, 238 us for 65536 x 4096 against 157 us for the per-tensor kernel.)
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t absbits(float x) { return __float_as_uint(x) & 0x7FFFFFFFu; }

// one 16-byte vector of DT elements as floats (4 x fp32 or 8 x fp16 / bf16)
template <int DT> __device__ __forceinline__ void vec_unpack(const v4i &v, float (&f)[ElemT<DT>::VEC])
{
    if constexpr (DT == ASQ_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __int_as_float(v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = (uint32_t)v[i];
            f[2 * i] = ElemT<DT>::load((uint16_t)(w & 0xFFFF));
            f[2 * i + 1] = ElemT<DT>::load((uint16_t)(w >> 16));
        }
    }
}

template <int DT> struct AbsMax {  // running |x| maximum of raw DT vectors
    uint32_t acc = 0;              // fp32: |bits|;  16-bit types: two packed 15-bit patterns
    __device__ __forceinline__ void add(const v4i &v)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (DT == ASQ_F32) acc = umax32(acc, (uint32_t)v[i] & 0x7FFFFFFFu);
            else acc = pk_max_u16(acc, (uint32_t)v[i] & 0x7FFF7FFFu);
        }
    }
    __device__ __forceinline__ uint32_t f32bits() const  // the maximum as an fp32 bit pattern (widening 16-bit -> fp32 is monotonic, NaN stays NaN)
    {
        if constexpr (DT == ASQ_F32) return acc;
        else return __float_as_uint(ElemT<DT>::load((uint16_t)umax32(acc & 0xFFFFu, acc >> 16)));
    }
};

// block-wide maximum of fp32 |x| bit patterns (256 threads) -> float (NaN if any NaN)
__device__ __forceinline__ float block_absmax_256(uint32_t m, float *red_f)
{
    uint32_t *red = (uint32_t *)red_f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = umax32(m, (uint32_t)__shfl_xor((int)m, off, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = m;
    __syncthreads();
    return __uint_as_float(umax32(umax32(red[0], red[1]), umax32(red[2], red[3])));
}

// The IEEE quotient RN(x / s) for a divisor shared by a whole row, in 5 VALU ops instead of the
// ~11 of the generic expansion (which re-derives 1/s per element; the per-token kernels were
// issue-bound on it).  Markstein's sequence: y = RN(1/s) by ONE true division per row, then
//     q0 = RN(x*y); r0 = x - s*q0 (exact, fma); q1 = RN(q0 + r0*y)   -> faithful
//     r1 = x - s*q1 (exact, fma);               q2 = RN(q1 + r1*y)   -> correctly rounded
// Valid when x is finite and nothing over-/underflows: `fast` (block-uniform) requires a finite row
// maximum and 2^-60 < s < 2^60 (then |x/s| <= 127.5 and every remainder is a normal number or exactly
// 0); other rows take the plain division.  A -0 numerator comes back as +0, which quantises to the same 0.
// Checked bit-for-bit against x / s on 2 x 10^11 operand pairs incl. all near-half-integer quotients
// (tools/ubench/exact_div_check.hip: 0 mismatches).
struct RowDivisor {
    float s, y;
    bool fast;
    __device__ __forceinline__ RowDivisor(float s_, float row_absmax)
        : s(s_), y(1.0f / s_), fast(row_absmax < 3.0e38f && s_ > 0x1p-60f && s_ < 0x1p60f) {}
};
struct QRowFast {
    float s, y;
    __device__ __forceinline__ float div(float x) const
    {
        const float q0 = __fmul_rn(x, y);
        const float q1 = __fmaf_rn(__fmaf_rn(-s, q0, x), y, q0);
        return __fmaf_rn(__fmaf_rn(-s, q1, x), y, q1);
    }
    __device__ __forceinline__ int operator()(float x) const { return quant_i8(div(x)); }
    // two quotients per instruction (v_pk_mul_f32 / v_pk_fma_f32): the same five IEEE operations per element
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ v2f_ div2(v2f_ x) const
    {
        const v2f_ ms = {-s, -s}, yy = {y, y};
        const v2f_ q0 = x * yy;
        const v2f_ q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(ms, q0, x), yy, q0);
        return __builtin_elementwise_fma(__builtin_elementwise_fma(ms, q1, x), yy, q1);
    }
};
