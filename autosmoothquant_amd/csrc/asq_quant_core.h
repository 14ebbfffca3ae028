// Per-element quantisation cores of the activation prologue (reference layers/nn/linear.py:88-96, :283-292): shared by the stand-alone quantisers
// (asq_quant.hip) and by the GEMM kernels that quantise their own activations (asq_gemm_skinny_fq.h) -- ONE statement of the arithmetic, so a fused launch is
// bit-identical to quantiser + GEMM by construction.
#pragma once
#include "asq_common.h"
#include <type_traits>

namespace asq {

// ---- per-element quantisation cores (bit-exact restatements; see oracle/w8a8.py) -------
template <int DT> struct QRound {  // x.round().clamp().to(int8)
    __device__ __forceinline__ int operator()(float x) const { return quant_i8(x); }
};
template <int DT> struct QDiv {  // (x / scalar) stays in x's dtype, then round/clamp
    float s;
    __device__ __forceinline__ int operator()(float x) const { return quant_i8(ElemT<DT>::round(x / s)); }
};
// QDiv without the per-element IEEE division (what made per-tensor-div 20 % slower than per-tensor-round): Markstein's sequence with
// y = RN(1/s), two quotients per instruction (QRowFast::div2), is the correctly rounded x / s whenever nothing over- or underflows.
// Host side: only for 2^-60 < s < 2^60.  Device side: a vector takes it when all of its elements are finite and below 2^60 (then
// |x / s| < 2^120; a quotient of magnitude >= 1/2 -- the only ones whose rounding matters -- has |x| > 2^-61, so every remainder is a
// normal number or exactly 0; smaller ones come out as some |q| < 2^-9 and quantise to 0 like the true quotient); any other vector
// (inf, NaN, huge bf16 / fp32 values) takes the plain division.  A -0 numerator may come back as +0: the same int8 0.
template <int DT> struct QDivFast {
    float s, y;
    static constexpr uint32_t kLim16 = DT == ASQ_F16 ? 0x7C00u : 0x5D80u;  // first 16-bit pattern that is not "finite and < 2^60" (bf16: 2^60 itself)
    __device__ __forceinline__ int operator()(float x) const
    {
        if (absbits(x) < 0x5D800000u) return quant_i8(ElemT<DT>::round(QRowFast{s, y}.div(x)));
        return quant_i8(ElemT<DT>::round(x / s));
    }
};
template <int DT> struct QDivF32 {  // x / f32 tensor promotes to fp32
    float s;
    __device__ __forceinline__ int operator()(float x) const { return quant_i8(x / s); }
};

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
    return (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
}

// quantise one 16-byte vector of DT -> VEC int8 packed in up to 2 dwords
template <int DT, class Q> __device__ __forceinline__ void quant_vec(const v4i &v, const Q &q, uint32_t (&o)[2])
{
    if constexpr (DT == ASQ_F32) {
        if constexpr (std::is_same<Q, QDivFast<DT>>::value) {
            const uint32_t m = umax32(umax32((uint32_t)v[0] & 0x7FFFFFFFu, (uint32_t)v[1] & 0x7FFFFFFFu), umax32((uint32_t)v[2] & 0x7FFFFFFFu, (uint32_t)v[3] & 0x7FFFFFFFu));
            if (m < 0x5D800000u) {  // all four finite and < 2^60
                const QRowFast qf{q.s, q.y};
                const auto a = qf.div2((QRowFast::v2f_){__int_as_float(v[0]), __int_as_float(v[1])}), b = qf.div2((QRowFast::v2f_){__int_as_float(v[2]), __int_as_float(v[3])});
                o[0] = pack4(quant_i8(a[0]), quant_i8(a[1]), quant_i8(b[0]), quant_i8(b[1]));
            } else {
                o[0] = pack4(quant_i8(__int_as_float(v[0]) / q.s), quant_i8(__int_as_float(v[1]) / q.s), quant_i8(__int_as_float(v[2]) / q.s), quant_i8(__int_as_float(v[3]) / q.s));
            }
        } else {
            o[0] = pack4(q(__int_as_float(v[0])), q(__int_as_float(v[1])), q(__int_as_float(v[2])), q(__int_as_float(v[3])));
        }
        o[1] = 0;
    } else {
        int r[8];
        if constexpr (std::is_same<Q, QDivFast<DT>>::value) {
            const uint32_t m = pk_max_u16(pk_max_u16((uint32_t)v[0] & 0x7FFF7FFFu, (uint32_t)v[1] & 0x7FFF7FFFu),
                                          pk_max_u16((uint32_t)v[2] & 0x7FFF7FFFu, (uint32_t)v[3] & 0x7FFF7FFFu));
            if ((m & 0xFFFFu) < Q::kLim16 && (m >> 16) < Q::kLim16) {  // every element finite and < 2^60: exact division without dividing
                const QRowFast qf{q.s, q.y};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t w = (uint32_t)v[i];
                    const auto t = qf.div2((QRowFast::v2f_){ElemT<DT>::load((uint16_t)(w & 0xFFFF)), ElemT<DT>::load((uint16_t)(w >> 16))});
                    r[2 * i] = quant_i8(ElemT<DT>::round(t[0]));
                    r[2 * i + 1] = quant_i8(ElemT<DT>::round(t[1]));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t w = (uint32_t)v[i];
                    r[2 * i] = quant_i8(ElemT<DT>::round(ElemT<DT>::load((uint16_t)(w & 0xFFFF)) / q.s));
                    r[2 * i + 1] = quant_i8(ElemT<DT>::round(ElemT<DT>::load((uint16_t)(w >> 16)) / q.s));
                }
            }
            o[0] = pack4(r[0], r[1], r[2], r[3]);
            o[1] = pack4(r[4], r[5], r[6], r[7]);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t w = (uint32_t)v[i];
            const float lo = ElemT<DT>::load((uint16_t)(w & 0xFFFF)), hi = ElemT<DT>::load((uint16_t)(w >> 16));
            if constexpr (std::is_same<Q, QRowFast>::value) {  // the row division on both halves at once (v_pk_mul_f32 / v_pk_fma_f32)
                const auto t = q.div2((QRowFast::v2f_){lo, hi});
                r[2 * i] = quant_i8(t[0]);
                r[2 * i + 1] = quant_i8(t[1]);
            } else {
                r[2 * i] = q(lo);
                r[2 * i + 1] = q(hi);
            }
        }
        o[0] = pack4(r[0], r[1], r[2], r[3]);
        o[1] = pack4(r[4], r[5], r[6], r[7]);
    }
}

}  // namespace asq
