// SiLU(gate) * up in the activation dtype, as the reference's composition computes it -- a = dt(dt(silu(g)) * u), models/llama.py:206-211 / HF LlamaMLP with
// torch ops -- shared by the stand-alone quantiser silu_mul_quant (asq_quant.hip) and by the gate||up GEMM's epilogue (asq_gemm_gateup.h): ONE statement of the
// arithmetic, so the fused GEMM is bit-identical to "two GEMMs, then asq_silu_mul_quantize" by construction.
#pragma once
#include "asq_common.h"

namespace asq {

// exp(x) as a fixed sequence of IEEE fp32 operations (no FMA, no library call): Cody-Waite reduction by ln 2,
// degree-7 Taylor polynomial in Horner form, scaling by two exact powers of two.  <= ~2 ulp from the true value
// on [-87, 88]; +inf above 88.8, 0 below -104; NaN propagates.  oracle/n1.py::exp_det repeats it step by step, so
// every kernel that uses it is compared with the oracle bit for bit (libm / ocml exp differ in the last ulp between
// implementations, which is what made the r1 silu test a +-1 comparison).
__device__ __forceinline__ float pow2i(int k) { return __int_as_float((k + 127) << 23); }  // 2^k, -126 <= k <= 127
__device__ __forceinline__ float exp_det(float x)
{
    if (!(x == x)) return x;
    x = fminf(fmaxf(x, -104.0f), 89.0f);
    const float n = rintf(__fmul_rn(x, 1.44269502162933349609375f));          // round-half-even(x * log2 e)
    float r = __fadd_rn(x, -__fmul_rn(n, 0.693138122558593750f));             // ln2_hi (17 significant bits)
    r = __fadd_rn(r, -__fmul_rn(n, 9.05800061445916071534156799316e-06f));    // ln2_lo
    float p = 1.0f / 5040.0f;
    p = __fadd_rn(__fmul_rn(p, r), 1.0f / 720.0f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f / 120.0f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f / 24.0f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f / 6.0f);
    p = __fadd_rn(__fmul_rn(p, r), 0.5f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f);
    const int ni = (int)n, n1 = ni >> 1, n2 = ni - n1;                          // |n1|, |n2| <= 75
    return __fmul_rn(__fmul_rn(p, pow2i(n1)), pow2i(n2));
}

// The same function on a PAIR of values: every multiply / add is one v_pk_mul_f32 / v_pk_add_f32 (gfx950 issues two IEEE fp32
// operations per lane per instruction), operation for operation the sequence of exp_det -- the SiLU kernel is VALU-bound on exactly
// this polynomial.  Two shortcuts that cannot change a result the caller sees:
//  * the final scaling p * 2^n1 * 2^n2 is one v_ldexp_f32: identical whenever the result is a normal number or overflows (the
//    first product is exact, the second rounds once, as ldexp does); for subnormal results both round once from the same exact
//    value;
//  * NAN_SELECT = false skips the "NaN in, NaN out" select: the caller's g / (1 + e) is NaN through g anyway.
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool NAN_SELECT = true> __device__ __forceinline__ v2f exp_det2(v2f x0)
{
    const v2f x = {__builtin_amdgcn_fmed3f(x0[0], -104.0f, 89.0f), __builtin_amdgcn_fmed3f(x0[1], -104.0f, 89.0f)};  // (a NaN lane computes on -104)
    const v2f t = x * 1.44269502162933349609375f;
    const v2f n = {rintf(t[0]), rintf(t[1])};
    v2f r = x - n * 0.693138122558593750f;
    r = r - n * 9.05800061445916071534156799316e-06f;
    v2f p = {1.0f / 5040.0f, 1.0f / 5040.0f};
    p = p * r + 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const v2f e = {__builtin_ldexpf(p[0], (int)n[0]), __builtin_ldexpf(p[1], (int)n[1])};
    if constexpr (!NAN_SELECT) return e;
    else return (v2f){x0[0] == x0[0] ? e[0] : x0[0], x0[1] == x0[1] ? e[1] : x0[1]};
}

// silu of two values.  FAST: the hardware transcendentals, g * v_rcp_f32(1 + v_exp_f32(-g * log2 e)) (~1 ulp each; the default since round 5: within the same
// +-1-int8 bound of the torch path as the fixed-order form).  Otherwise the bit-reproducible exp_det + IEEE division oracle/n1.py repeats.
template <bool FAST> __device__ __forceinline__ v2f silu2(float g0, float g1)
{
    float q0, q1;
    if constexpr (FAST) {
        q0 = g0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g0 * -1.4426950408889634f));
        q1 = g1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g1 * -1.4426950408889634f));
        asm("" : "+v"(q0), "+v"(q1));   // (the conversion to the activation dtype that follows is its own rounding, in every caller: no v_fma_mix folding)
    } else {
        const v2f den = exp_det2<false>((v2f){-g0, -g1}) + 1.0f;
        q0 = __fdiv_rn(g0, den[0]);
        q1 = __fdiv_rn(g1, den[1]);
    }
    return (v2f){q0, q1};
}
// fp16: dt(silu) * u on packed halves -- the product of two fp16 values is exact in fp32, so the IEEE fp16 product (v_pk_mul_f16) IS dt(fp32(sl) * fp32(u))
typedef _Float16 v2h_silu __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t silu_times_up_h(v2f sl, uint32_t u_packed)
{
    const v2h_silu s = {(_Float16)sl[0], (_Float16)sl[1]};
    return __builtin_bit_cast(uint32_t, s * __builtin_bit_cast(v2h_silu, u_packed));
}
// fp32 / bf16: two separate roundings to the activation dtype
template <int DT> __device__ __forceinline__ v2f silu_times_up(v2f sl, float u0, float u1)
{
    const v2f pr = (v2f){ElemT<DT>::round(sl[0]), ElemT<DT>::round(sl[1])} * (v2f){u0, u1};
    return (v2f){ElemT<DT>::round(pr[0]), ElemT<DT>::round(pr[1])};
}

}  // namespace asq
