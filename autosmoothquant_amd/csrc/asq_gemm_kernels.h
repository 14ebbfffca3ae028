// INT8 x INT8 -> INT32 GEMM on CDNA4 matrix cores with fused epilogues: kernels + launcher.
//
//   acc[m,n] = sum_k x[m,k] * w[n,k]          (both operands K-contiguous, "TN")
//
// replaces cublasLtMatmul behind I8CUGEMM::linear_a8_w8_o32_ (reference
// csrc/int8gemm/cublasINT8MMWrapper.cc:224-354) and, through the epilogue functors, the
// eager dequant/bias code of layers/nn/linear.py:104,197-206,300 and the int8-out flavours
// (cublasINT8MMWrapper.cc:360-672).
//
// MFMA mapping (v_mfma_i32_32x32x32_i8, 64-lane wave):
//   the W tile is the matrix-core "A" operand (rows = output channel n), the X tile is "B"
//   (cols = token m).  Lane l therefore owns token m = l&31 and, per accumulator register
//   group g = reg>>2, FOUR CONSECUTIVE output channels n = 8g + 4(l>>5) + (reg&3): the
//   epilogue stores 16 B (i32/f32) or 8 B (f16/bf16) per lane per group without any
//   cross-lane shuffle.  Both operands read the same 16 k-bytes per lane
//   (k = 32*ks + 16*(l>>5) ...), so the dot product is independent of the hardware's
//   internal k ordering.
//
// Epilogue protocol (one wave, NTN x NTM accumulator tiles of 32(n) x 32(m)):
//   1. every per-token / per-channel operand the lane will need (<= 4 row scales, <= 8 float4
//      of column scales, <= 8 float4 of bias) is loaded UP FRONT into registers,
//   2. then all tiles are converted and stored with no memory wait in between
//      (on gfx9 stores count in vmcnt: a load-wait inside the store loop would serialise the
//      stores on their write acknowledgements -- measured 2.7x on the whole epilogue).
#pragma once
#include "asq_common.h"
#include <string.h>
#include <type_traits>
#include <utility>
#include <unordered_map>

namespace asq {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Pin a wave-uniform pointer into an SGPR pair.  The LDS-DMA inline asm takes its base as an "s"
// operand; LLVM may otherwise decide to evaluate a uniform 64-bit address chain on the VALU (seen with
// one fp8 instantiation), which no constraint can repair afterwards.
__device__ __forceinline__ const int8_t *uniform_ptr(const int8_t *p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const int8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// d = sext24(a) * sext24(b) + c as ONE v_mad_i32_i24 (the offset-image correction in front of the epilogues: from `__mul24(a, b) + c` chains hipcc prefers
// v_mul_i32_i24 with an SDWA operand select + v_add3_u32, one VALU operation more per element)
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---------------------------------------------------------------------------------
// matrix-core policies: one "MMA step" consumes 16 k-bytes per lane of each operand (a v4i)
// ---------------------------------------------------------------------------------
typedef float v16f __attribute__((ext_vector_type(16)));
typedef long v2l __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

typedef int v8i __attribute__((ext_vector_type(8)));

// mma():  one step over 16 k-bytes per lane.   mma2(): two consecutive steps (32 k-bytes per lane) -- the
// form the tiled kernels use for fp8 so that it can run on the K = 64 block-scaled instruction.
struct MmaI8 {  // int8 x int8 -> int32, exact: one v_mfma_i32_32x32x32_i8
    using acc_t = v16i;
    using acc4_t = v4i;
    static constexpr bool kIsInt = true;
    static __device__ __forceinline__ acc_t mma2(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc_t &c)
    {
        return mma(a1, b1, mma(a0, b0, c));  // (the tiled kernels keep their own per-step issue order for int8)
    }
    // 16x16 tile, 32 k-bytes per lane (the weight-streaming kernel's unit): two v_mfma_i32_16x16x64_i8
    static __device__ __forceinline__ acc4_t mma16(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc4_t &c)
    {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, c, 0, 0, 0), 0, 0, 0);
    }
    static __device__ __forceinline__ acc_t mma(const v4i &a, const v4i &b, const acc_t &c)
    {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    }
};

// fp8 at full rate: the non-scaled v_mfma_f32_32x32x16_fp8_fp8 runs at the bf16 rate on gfx950; only the
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 reaches the fp8 peak.  With both E8M0 scales = 127 (2^0) it is
// the plain fp8 product sum; A and B fragments use the same (lane-half, byte) -> k assignment, so feeding two
// consecutive 16-byte fragments as the 32-byte operand pairs every k-byte with its partner.
template <int FMT> __device__ __forceinline__ v4f mma_fp8_16x16_k128(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const v4f &c)
{
    const v8i A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const v8i B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

template <int FMT>  // 0 = e4m3fn, 1 = e5m2 (cbsz / blgp encoding)
__device__ __forceinline__ v16f mma_fp8_k64(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const v16f &c)
{
    const v8i A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const v8i B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, FMT, FMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

struct MmaFp8 {  // OCP e4m3fn x e4m3fn -> fp32
    using acc_t = v16f;
    using acc4_t = v4f;
    static constexpr bool kIsInt = false;
    static __device__ __forceinline__ acc_t mma2(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc_t &c)
    {
        return mma_fp8_k64<0>(a0, a1, b0, b1, c);
    }
    static __device__ __forceinline__ acc4_t mma16(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc4_t &c)
    {
        return mma_fp8_16x16_k128<0>(a0, a1, b0, b1, c);
    }
    static __device__ __forceinline__ acc_t mma(const v4i &a, const v4i &b, const acc_t &c)
    {
        const v2l al = __builtin_bit_cast(v2l, a), bl = __builtin_bit_cast(v2l, b);
        acc_t r = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[0], bl[0], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al[1], bl[1], r, 0, 0, 0);
    }
};

struct MmaBf8 {  // OCP e5m2 x e5m2 -> fp32
    using acc_t = v16f;
    using acc4_t = v4f;
    static constexpr bool kIsInt = false;
    static __device__ __forceinline__ acc_t mma2(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc_t &c)
    {
        return mma_fp8_k64<1>(a0, a1, b0, b1, c);
    }
    static __device__ __forceinline__ acc4_t mma16(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const acc4_t &c)
    {
        return mma_fp8_16x16_k128<1>(a0, a1, b0, b1, c);
    }
    static __device__ __forceinline__ acc_t mma(const v4i &a, const v4i &b, const acc_t &c)
    {
        const v2l al = __builtin_bit_cast(v2l, a), bl = __builtin_bit_cast(v2l, b);
        acc_t r = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(al[0], bl[0], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(al[1], bl[1], r, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------
// epilogue functors
// ---------------------------------------------------------------------------------
struct EpiI32 {
    using Mma = MmaI8;
    static constexpr bool kHasRow = false, kHasCol = false, kHasBias = false;
    static constexpr int kOutBytes = 4;
    __device__ __forceinline__ v4i pack(const v4i &a, float, const v4f &, const v4f &) const { return a; }
    int32_t *out;
    int64_t N;
    bool vec_ok;
    // the functor a block works with: split-K slab `split` of [M, Ncols] int32, group `grp` of a grouped launch.
    // One by-value expression (no conditionally modified copy: that ended up in scratch memory).
    __device__ __forceinline__ EpiI32 rebased(int, int split, int64_t M, int64_t Ncols) const { return EpiI32{out + (int64_t)split * M * Ncols, N, vec_ok}; }
    // the same epilogue for the sub-problem that starts at output column n0 (host side; n0 % 256 == 0 keeps every alignment)
    static constexpr bool kColView = true;
    EpiI32 col_view(int64_t n0) const { return EpiI32{out + n0, N, vec_ok}; }
    __device__ __forceinline__ float row(int64_t) const { return 1.0f; }
    __device__ __forceinline__ void cols(int64_t, int64_t, v4f &, v4f &) const {}
    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4i &a, float, const v4f &, const v4f &, int64_t Ncols) const
    {
        int32_t *p = out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            *(v4i *)p = a;
        } else {
            if (n < Ncols) p[0] = a[0];
            if (n + 1 < Ncols) p[1] = a[1];
            if (n + 2 < Ncols) p[2] = a[2];
            if (n + 3 < Ncols) p[3] = a[3];
        }
    }
};

// out = dequant(acc) (+bias) -> DT.  HAS_* are compile-time so the hot instantiations carry no
// dead loads or branches; `order` is a wave-uniform runtime select.
template <int DT, bool HAS_ROW, bool HAS_COL, bool HAS_BIAS> struct EpiDequant {
    static constexpr bool kGroupable = true;  // has per-group scales: grouped launches instantiate gemm_i8_p8<Epi, 0, true>
    using Mma = MmaI8;
    static constexpr bool kHasRow = HAS_ROW, kHasCol = HAS_COL, kHasBias = HAS_BIAS;
    static constexpr int kOutBytes = (DT == ASQ_F32) ? 4 : 2;
    // (pointers first, 4-byte scalars together: a float followed by padding and a pointer keeps SROA from promoting the by-value
    // kernel argument, which costs the skinny kernel 32 B/lane of scratch)
    void *out;
    int64_t N;
    const float *s_row;    // [M]   (HAS_ROW)
    const float *s_col;    // [N]   (HAS_COL)
    const float *bias;     // [N]   (HAS_BIAS)
    const float *s_group;  // grouped launches: per-group scalar dequant scale [ngroups] (device) or null
    float s_scalar;
    int order;
    bool vec_ok;

    __device__ __forceinline__ EpiDequant rebased(int grp, int, int64_t, int64_t Ncols) const
    {
        return EpiDequant{out, N, s_row, HAS_COL ? s_col + (int64_t)grp * Ncols : s_col, HAS_BIAS ? bias + (int64_t)grp * Ncols : bias, s_group,
                          s_group ? s_group[grp] : s_scalar, order, vec_ok};
    }
    __device__ __forceinline__ float row(int64_t m) const { return HAS_ROW ? s_row[m] : 1.0f; }
    static constexpr bool kColView = true;
    EpiDequant col_view(int64_t n0) const  // host side: the sub-problem starting at output column n0 (n0 % 256 == 0)
    {
        return EpiDequant{(char *)out + n0 * kOutBytes, N, s_row, s_col ? s_col + n0 : s_col, bias ? bias + n0 : bias, s_group, s_scalar, order, vec_ok};
    }

    __device__ __forceinline__ void cols(int64_t n, int64_t Ncols, v4f &sc, v4f &b) const
    {
        sc = (v4f){s_scalar, s_scalar, s_scalar, s_scalar};
        b = (v4f){0.f, 0.f, 0.f, 0.f};
        if (vec_ok && n + 3 < Ncols) {
            if constexpr (HAS_COL) sc = *(const v4f *)(s_col + n);
            if constexpr (HAS_BIAS) b = *(const v4f *)(bias + n);
        } else {
            if constexpr (HAS_COL) {
                if (n < Ncols) sc[0] = s_col[n];
                if (n + 1 < Ncols) sc[1] = s_col[n + 1];
                if (n + 2 < Ncols) sc[2] = s_col[n + 2];
                if (n + 3 < Ncols) sc[3] = s_col[n + 3];
            }
            if constexpr (HAS_BIAS) {
                if (n < Ncols) b[0] = bias[n];
                if (n + 1 < Ncols) b[1] = bias[n + 1];
                if (n + 2 < Ncols) b[2] = bias[n + 2];
                if (n + 3 < Ncols) b[3] = bias[n + 3];
            }
        }
    }

    __device__ __forceinline__ float one(int acc, float sc, float sr, float b) const
    {
        const float a = (float)acc;  // v_cvt_f32_i32: round-to-nearest-even, as ATen
        float v;
        if (order == ASQ_EPI_SCALE_FIRST) {
            const float ds = HAS_ROW ? __fmul_rn(sc, sr) : sc;
            v = __fmul_rn(ds, a);
        } else {
            v = __fmul_rn(a, sc);
            if constexpr (HAS_ROW) v = __fmul_rn(v, sr);
        }
        if constexpr (HAS_BIAS) v = __fadd_rn(v, b);
        return v;
    }

    // 4 finished outputs in their storage encoding: v4i of fp32 bit patterns, or uint2 of 4 x 16-bit
    // two outputs per instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations as one(), pairwise)
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ v2f_ two(int acc0, int acc1, v2f_ sc, float sr, v2f_ b) const
    {
        const v2f_ a = {(float)acc0, (float)acc1};
        v2f_ v;
        if (order == ASQ_EPI_SCALE_FIRST) {
            v = (HAS_ROW ? sc * sr : sc) * a;
        } else {
            v = a * sc;
            if constexpr (HAS_ROW) v = v * sr;
        }
        if constexpr (HAS_BIAS) v = v + b;
        return v;
    }

    __device__ __forceinline__ auto pack(const v4i &a, float sr, const v4f &sc, const v4f &b) const
    {
        using E = ElemT<DT>;
        if constexpr (DT == ASQ_F16) {  // v_cvt_pk_f16_f32: two conversions and the pack in one instruction
            typedef _Float16 v2h_ __attribute__((ext_vector_type(2)));
            v2f_ lo = two(a[0], a[1], (v2f_){sc[0], sc[1]}, sr, (v2f_){b[0], b[1]}), hi = two(a[2], a[3], (v2f_){sc[2], sc[3]}, sr, (v2f_){b[2], b[3]});
            asm("" : "+v"(lo), "+v"(hi));  // opaque: keeps fptrunc(fmul) from becoming one v_fma_mix (a single rounding; see f32_to_f16_bits)
            return (v2u){__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, v2h_)), __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, v2h_))};
        }
        if constexpr (DT == ASQ_BF16) {  // v_cvt_pk_bf16_f32 (RNE, identical to the software rounding on every non-NaN input: tools/ubench/bf16_cvt_check) + canonical NaN
            v2f_ lo = two(a[0], a[1], (v2f_){sc[0], sc[1]}, sr, (v2f_){b[0], b[1]}), hi = two(a[2], a[3], (v2f_){sc[2], sc[3]}, sr, (v2f_){b[2], b[3]});
            asm("" : "+v"(lo), "+v"(hi));
            return (v2u){f32x2_to_bf16x2_bits(lo[0], lo[1]), f32x2_to_bf16x2_bits(hi[0], hi[1])};
        }
        const float v0 = one(a[0], sc[0], sr, b[0]), v1 = one(a[1], sc[1], sr, b[1]);
        const float v2 = one(a[2], sc[2], sr, b[2]), v3 = one(a[3], sc[3], sr, b[3]);
        if constexpr (DT == ASQ_F32) {
            return __builtin_bit_cast(v4i, (v4f){v0, v1, v2, v3});
        } else {
            return (v2u){(uint32_t)E::store(v0) | ((uint32_t)E::store(v1) << 16), (uint32_t)E::store(v2) | ((uint32_t)E::store(v3) << 16)};
        }
    }

    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4i &a, float sr, const v4f &sc, const v4f &b, int64_t Ncols) const
    {
        using E = ElemT<DT>;
        const float v0 = one(a[0], sc[0], sr, b[0]), v1 = one(a[1], sc[1], sr, b[1]);
        const float v2 = one(a[2], sc[2], sr, b[2]), v3 = one(a[3], sc[3], sr, b[3]);
        typename E::type *p = (typename E::type *)out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            if constexpr (DT == ASQ_F32) {
                *(v4f *)p = (v4f){v0, v1, v2, v3};
            } else {
                const uint32_t lo = (uint32_t)E::store(v0) | ((uint32_t)E::store(v1) << 16);
                const uint32_t hi = (uint32_t)E::store(v2) | ((uint32_t)E::store(v3) << 16);
                *(uint2 *)p = make_uint2(lo, hi);
            }
        } else {
            if (n < Ncols) p[0] = E::store(v0);
            if (n + 1 < Ncols) p[1] = E::store(v1);
            if (n + 2 < Ncols) p[2] = E::store(v2);
            if (n + 3 < Ncols) p[3] = E::store(v3);
        }
    }
};

// int8-OUT epilogue that hands the activation straight to the NEXT W8A8 linear (the K3-K5 idea of the
// reference's csrc/int8gemm/bindings.cpp:86-142, with the consumer's real quantiser instead of one alpha):
//     y  = DT(dequant(acc) (+ bias))            -- exactly what EpiDequant<DT> would have stored
//     a  = relu(y)                               (act = 1; OPT's fc1 -> ReLU -> fc2, reference models/opt.py:127-128)
//     xq = int8(clamp(rne(a)))                   (ASQ_ACT_ROUND: consumer is a per-tensor W8A8BFP32OFP32Linear, linear.py:95-96)
//        | int8(clamp(rne(DT(a / quant_scale)))) (ASQ_ACT_DIV:   consumer is a per-tensor ...WithQuantScale, linear.py:289-292)
// Every step is elementwise, so the result is bit-identical to "linear, activation, then the consumer's prologue".
// Row / column scales and bias are runtime-optional (one instantiation per DT keeps the build small).
template <int DT> struct EpiDequantQ {
    using Mma = MmaI8;
    static constexpr bool kHasRow = true, kHasCol = true, kHasBias = true;
    static constexpr int kOutBytes = 1;
    int8_t *out;
    int64_t N;
    const float *s_row, *s_col, *bias;
    float s_scalar, quant_scale;
    int order, act, qmode;
    bool vec_ok;
    __device__ __forceinline__ EpiDequantQ rebased(int, int, int64_t, int64_t) const { return *this; }
    static constexpr bool kColView = true;
    EpiDequantQ col_view(int64_t n0) const
    {
        return EpiDequantQ{out + n0, N, s_row, s_col ? s_col + n0 : s_col, bias ? bias + n0 : bias, s_scalar, quant_scale, order, act, qmode, vec_ok};
    }
    __device__ __forceinline__ float row(int64_t m) const { return s_row ? s_row[m] : 1.0f; }
    __device__ __forceinline__ void cols(int64_t n, int64_t Ncols, v4f &sc, v4f &b) const
    {
        sc = (v4f){s_scalar, s_scalar, s_scalar, s_scalar};
        b = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < Ncols) {
                if (s_col) sc[i] = s_col[n + i];
                if (bias) b[i] = bias[n + i];
            }
    }
    __device__ __forceinline__ int one(int acc, float sc, float sr, float b) const
    {
        const float a = (float)acc;
        float v;
        if (order == ASQ_EPI_SCALE_FIRST) {
            v = __fmul_rn(s_row ? __fmul_rn(sc, sr) : sc, a);
        } else {
            v = __fmul_rn(a, sc);
            if (s_row) v = __fmul_rn(v, sr);
        }
        if (bias) v = __fadd_rn(v, b);
        float y = ElemT<DT>::round(v);
        if (act == 1) y = (y < 0.0f) ? 0.0f : y;  // relu; NaN stays NaN
        return quant_i8(qmode == ASQ_ACT_DIV ? ElemT<DT>::round(y / quant_scale) : y);
    }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4i &a, float sr, const v4f &sc, const v4f &b, int64_t Ncols) const
    {
        int8_t *p = out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            *(uint32_t *)p = (uint32_t)(one(a[0], sc[0], sr, b[0]) & 0xFF) | ((uint32_t)(one(a[1], sc[1], sr, b[1]) & 0xFF) << 8) |
                             ((uint32_t)(one(a[2], sc[2], sr, b[2]) & 0xFF) << 16) | ((uint32_t)(one(a[3], sc[3], sr, b[3]) & 0xFF) << 24);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < Ncols) p[i] = (int8_t)one(a[i], sc[i], sr, b[i]);
        }
    }
};

struct EpiI8 {  // out = sat_i8(rne(alpha*acc + beta*c)), c = previous out
    using Mma = MmaI8;
    static constexpr bool kHasRow = false, kHasCol = false, kHasBias = false;
    static constexpr int kOutBytes = 1;  // read-modify-write of C: stays on the direct (unstaged) store path
    int8_t *out;
    int64_t N;
    float alpha, beta;
    bool vec_ok;
    __device__ __forceinline__ EpiI8 rebased(int, int, int64_t, int64_t) const { return *this; }
    __device__ __forceinline__ float row(int64_t) const { return 1.0f; }
    __device__ __forceinline__ void cols(int64_t, int64_t, v4f &, v4f &) const {}
    __device__ __forceinline__ int one(int acc, int c) const
    {
        float v = __fmul_rn(alpha, (float)acc);
        if (beta != 0.0f) v = __fadd_rn(v, __fmul_rn(beta, (float)c));
        return quant_i8(v);
    }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4i &a, float, const v4f &, const v4f &, int64_t Ncols) const
    {
        int8_t *p = out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            const uint32_t c = (beta != 0.0f) ? *(const uint32_t *)p : 0u;
            uint32_t r = 0;
            r |= (uint32_t)(one(a[0], (int)(int8_t)(c)) & 0xFF);
            r |= (uint32_t)(one(a[1], (int)(int8_t)(c >> 8)) & 0xFF) << 8;
            r |= (uint32_t)(one(a[2], (int)(int8_t)(c >> 16)) & 0xFF) << 16;
            r |= (uint32_t)(one(a[3], (int)(int8_t)(c >> 24)) & 0xFF) << 24;
            *(uint32_t *)p = r;
        } else {
            if (n < Ncols) p[0] = (int8_t)one(a[0], (beta != 0.0f) ? (int)p[0] : 0);
            if (n + 1 < Ncols) p[1] = (int8_t)one(a[1], (beta != 0.0f) ? (int)p[1] : 0);
            if (n + 2 < Ncols) p[2] = (int8_t)one(a[2], (beta != 0.0f) ? (int)p[2] : 0);
            if (n + 3 < Ncols) p[3] = (int8_t)one(a[3], (beta != 0.0f) ? (int)p[3] : 0);
        }
    }
};

// fp8 linear (reference easy_fp8_gemm, linear.py:336-369): out = acc_f32 * (a_scale * w_scale) (+ bias).
// a_scale: device pointer (per-token [M], or per-tensor [1] produced on the device by the dynamic
// quantiser) or, when null, the host scalar.  Tolerance-checked (the reference dequantises both
// operands and calls F.linear; its summation order is unspecified).
template <int DT, bool HAS_BIAS, class MMA_ = MmaFp8> struct EpiFp8 {
    static constexpr bool kGroupable = true;
    using Mma = MMA_;
    static constexpr bool kHasRow = true, kHasCol = false, kHasBias = HAS_BIAS;
    static constexpr int kOutBytes = (DT == ASQ_F32) ? 4 : 2;
    __device__ __forceinline__ auto pack(const v4f &a, float sr, const v4f &sc, const v4f &b) const
    {
        using E = ElemT<DT>;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = __fmul_rn(a[i], __fmul_rn(sr, sc[i]));
            if constexpr (HAS_BIAS) v[i] = __fadd_rn(v[i], b[i]);
        }
        if constexpr (DT == ASQ_F32) {
            return __builtin_bit_cast(v4i, (v4f){v[0], v[1], v[2], v[3]});
        } else if constexpr (DT == ASQ_F16) {  // two conversions + the pack per instruction, as EpiDequant::pack
            typedef float v2f_ __attribute__((ext_vector_type(2)));
            typedef _Float16 v2h_ __attribute__((ext_vector_type(2)));
            v2f_ lo = {v[0], v[1]}, hi = {v[2], v[3]};
            asm("" : "+v"(lo), "+v"(hi));
            return (v2u){__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, v2h_)), __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, v2h_))};
        } else {
            (void)sizeof(E);
            return (v2u){f32x2_to_bf16x2_bits(v[0], v[1]), f32x2_to_bf16x2_bits(v[2], v[3])};
        }
    }
    void *out;
    int64_t N;
    const float *a_scale_dev;
    const float *bias;
    const float *w_scale_group;  // grouped launches: per-group weight scale [ngroups] (device) or null
    float a_scale_host, w_scale;
    bool a_per_token, vec_ok;
    __device__ __forceinline__ EpiFp8 rebased(int grp, int, int64_t, int64_t Ncols) const
    {
        return EpiFp8{out, N, a_scale_dev, HAS_BIAS ? bias + (int64_t)grp * Ncols : bias, w_scale_group, a_scale_host,
                      w_scale_group ? w_scale_group[grp] : w_scale, a_per_token, vec_ok};
    }
    __device__ __forceinline__ float row(int64_t m) const { return a_scale_dev ? (a_per_token ? a_scale_dev[m] : a_scale_dev[0]) : a_scale_host; }
    __device__ __forceinline__ void cols(int64_t n, int64_t Ncols, v4f &sc, v4f &b) const
    {
        sc = (v4f){w_scale, w_scale, w_scale, w_scale};
        b = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) {
            if (vec_ok && n + 3 < Ncols) {
                b = *(const v4f *)(bias + n);
            } else {
                if (n < Ncols) b[0] = bias[n];
                if (n + 1 < Ncols) b[1] = bias[n + 1];
                if (n + 2 < Ncols) b[2] = bias[n + 2];
                if (n + 3 < Ncols) b[3] = bias[n + 3];
            }
        }
    }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4f &a, float sr, const v4f &sc, const v4f &b, int64_t Ncols) const
    {
        using E = ElemT<DT>;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = __fmul_rn(a[i], __fmul_rn(sr, sc[i]));
            if constexpr (HAS_BIAS) v[i] = __fadd_rn(v[i], b[i]);
        }
        typename E::type *p = (typename E::type *)out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            if constexpr (DT == ASQ_F32) {
                *(v4f *)p = (v4f){v[0], v[1], v[2], v[3]};
            } else {
                const uint32_t lo = (uint32_t)E::store(v[0]) | ((uint32_t)E::store(v[1]) << 16);
                const uint32_t hi = (uint32_t)E::store(v[2]) | ((uint32_t)E::store(v[3]) << 16);
                *(uint2 *)p = make_uint2(lo, hi);
            }
        } else {
            if (n < Ncols) p[0] = E::store(v[0]);
            if (n + 1 < Ncols) p[1] = E::store(v[1]);
            if (n + 2 < Ncols) p[2] = E::store(v[2]);
            if (n + 3 < Ncols) p[3] = E::store(v[3]);
        }
    }
};


// Wave-level epilogue over NTN x NTM accumulator tiles; tile (in, im) covers
// n in [nw0 + 32*in, +32), m in [mw0 + mstep(im), +32).  `get(in, im)` returns the v16i.
template <int NTN, int NTM, class Epi, class Get, class MOff>
__device__ __forceinline__ void epilogue_wave(const Epi &epi, Get get, MOff moff, int64_t mw0, int64_t nw0, int lane, int64_t M, int64_t N)
{
    float sr[NTM];
    int64_t mrow[NTM];
#pragma unroll
    for (int im = 0; im < NTM; ++im) {
        mrow[im] = mw0 + moff(im) + (lane & 31);
        sr[im] = 1.0f;
        if constexpr (Epi::kHasRow) {
            if (mrow[im] < M) sr[im] = epi.row(mrow[im]);
        }
    }
    v4f sc[NTN][4], bb[NTN][4];
#pragma unroll
    for (int in = 0; in < NTN; ++in)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t n = nw0 + in * 32 + 8 * g + 4 * (lane >> 5);
            sc[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
            bb[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (n < N) epi.cols(n, N, sc[in][g], bb[in][g]);
        }
    // compile-time (in, im): with a runtime loop the compiler sometimes keeps the tile loop rolled (large bodies: bf16 rounding +
    // int8 re-quantisation) and then indexes the accumulators through a 512-B/lane scratch copy
    static_for<NTM * NTN>([&](auto t) __attribute__((always_inline)) {
        constexpr int im = decltype(t)::value / NTN, in = decltype(t)::value % NTN;
        if (mrow[im] >= M) return;
        const typename Epi::Mma::acc_t a = get(in, im);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t n = nw0 + in * 32 + 8 * g + 4 * (lane >> 5);
            if (n < N)
                epi.store4(mrow[im], n, (typename Epi::Mma::acc4_t){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}, sr[im], sc[in][g], bb[in][g], N);
        }
    });
}

struct MmaI8x16 {  // one v_mfma_i32_16x16x64_i8: 16 (A rows) x 16 (B rows) x 64 k-bytes, exact
    static __device__ __forceinline__ v4i mma(const v4i &a, const v4i &b, const v4i &c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
};

// Direct-store epilogue of a (16 * NT16)(m) x (16 * NC16)(n) wave tile in the 16 x 16 layout (edge tiles / unaligned / 4-byte outputs): a store instruction
// touches 16 rows with 4 x 8 (16) bytes each -- slow, and only used where the row epilogue cannot be.
template <int NT16 = 8, int NC16 = 4, class Epi, class Get> __device__ __forceinline__ void epilogue_wave16(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, int64_t M, int64_t N)
{
    const int t = lane & 15, q = lane >> 4;
    static_for<NC16>([&](auto in_) __attribute__((always_inline)) {
        constexpr int in16 = decltype(in_)::value;
        const int64_t n = nw0 + in16 * 16 + 4 * q;
        v4f sc = (v4f){0.f, 0.f, 0.f, 0.f}, bb = (v4f){0.f, 0.f, 0.f, 0.f};
        if (n < N) epi.cols(n, N, sc, bb);
        static_for<NT16>([&](auto im_) __attribute__((always_inline)) {
            constexpr int im16 = decltype(im_)::value;
            const int64_t m = mw0 + im16 * 16 + t;
            if (m < M && n < N) {
                const float sr = Epi::kHasRow ? epi.row(m) : 1.0f;
                epi.store4(m, n, get(in16, im16), sr, sc, bb, N);
            }
        });
    });
}

// Coalescing epilogue of the 256x256 kernel (wave tile 128(m) x 64(n): NTN = 2, NTM = 4).
// In the matrix-core layout a lane owns ONE token and 4 consecutive channels, so a direct store
// instruction touches 32 different output rows with 16 B each: measured ~10-12 B/clk per CU, 12.9k
// cycles for the fp16 tile and 26k for an int32 split-K slab.  Here the finished (scaled, biased,
// converted) values go through a WAVE-PRIVATE 16 KiB LDS image first and leave as whole rows: 8
// lanes x 16 B = one 128-B line (2-byte outputs; 16 lanes x 16 B = 256 B for 4-byte outputs), 8 (4)
// full rows per store instruction.  16-B chunks are XOR-swizzled by the row (and, for 2-byte outputs, the
// 8-byte halves flipped on odd rows) so the transposing ds_write and the lane-linear
// ds_read_b128 are conflict-free.  No block barrier: each
// wave only re-reads what it wrote; LDS operations of one wave execute in order.
// Needs: out 16-B aligned and N * sizeof(out element) % 16 == 0 (then a 16-B chunk is never ragged).
// 16-byte global store with an explicit cache policy: 0 = default (write-back in the XCD's L2), 1 = nt (streaming),
// 2 = sc1, 3 = sc0 sc1 (write-through: the line leaves the L2 with the store, nothing stays dirty for the
// end-of-kernel release to write back).  tools/ubench/clock_probe measures them; see DESIGN.md section 4.
template <int POL> __device__ __forceinline__ void store16_policy(void *p, const v4i &v)
{
    if constexpr (POL == 0) *(v4i *)p = v;
    else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// 16 x 16 accumulator layout (v_mfma_i32_16x16x64_i8; gemm_i8_p8<.., L16 = true>): the write side of the staged epilogue.  Lane l owns token (l & 15) and
// channels 4 * (l >> 4) .. + 3 of every 16 x 16 tile; get(in16, im16) returns its 4 registers.  The images (and so the read / store side) are the ones of the
// 32 x 32 layout: 2-byte outputs [rows][128 B] with chunk ^= (row >> 1) & 7 and the 8-byte halves flipped on odd rows, 4-byte outputs [64 rows][256 B] with
// chunk ^= row & 15.  16 consecutive lanes write 16 consecutive rows of one chunk: conflict-free in both.
template <int NTM, class Epi, class Get>
__device__ __forceinline__ void staged_write16(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, int64_t M, int64_t N, unsigned stage, int pass)
{
    constexpr int EB = Epi::kOutBytes;
    typedef __attribute__((address_space(3))) v4i *lds_v4i;
    typedef __attribute__((address_space(3))) v2u *lds_u2;
    const int t = lane & 15, q = lane >> 4;
    v4f sc[4], bb[4];
#pragma unroll
    for (int in = 0; in < 4; ++in) {
        const int64_t n = nw0 + in * 16 + 4 * q;
        sc[in] = (v4f){0.f, 0.f, 0.f, 0.f};
        bb[in] = (v4f){0.f, 0.f, 0.f, 0.f};
        if (n < N) epi.cols(n, N, sc[in], bb[in]);
    }
    constexpr int NT = EB == 2 ? 2 * NTM : (NTM >= 2 ? 4 : 2);   // 16-token tiles per staging pass: the whole wave tile (2-byte) or 64 rows (4-byte)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int im16 = pass * NT + i, row = 16 * i + t;
        const int64_t m = mw0 + 16 * im16 + t;
        float sr = 1.0f;
        if constexpr (Epi::kHasRow) {
            if (m < M) sr = epi.row(m);
        }
#pragma unroll
        for (int in = 0; in < 4; ++in) {
            const auto v = epi.pack(get(in, im16), sr, sc[in], bb[in]);
            if constexpr (EB == 2) *(lds_u2)(uintptr_t)(stage + row * 128 + (((2 * in + (q >> 1)) ^ ((row >> 1) & 7)) << 4) + 8 * ((q ^ row) & 1)) = v;
            else *(lds_v4i)(uintptr_t)(stage + row * 256 + (((4 * in + q) ^ (row & 15)) << 4)) = v;
        }
    }
}

template <int NTM, int POL = 0, bool L16 = false, class Epi, class Get>  // NTM = 32-row token tiles of the wave tile: 4 (256-row block tile) or 2 (128-row)
__device__ __forceinline__ void epilogue_wave_staged(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, int64_t M, int64_t N, unsigned stage)
{
    constexpr int EB = Epi::kOutBytes;
    static_assert(NTM == 1 || NTM == 2 || NTM == 4, "wave tile of 32, 64 or 128 rows");
    static_assert(EB == 2 || EB == 4, "staged epilogue: 2- or 4-byte outputs");
    typedef __attribute__((address_space(3))) v4i *lds_v4i;
    typedef __attribute__((address_space(3))) v2u *lds_u2;
    const int ml = lane & 31, hi = lane >> 5;
    float sr[NTM];
    v4f sc[2][4], bb[2][4];
    if constexpr (!L16) {
#pragma unroll
        for (int im = 0; im < NTM; ++im) {
            const int64_t m = mw0 + im * 32 + ml;
            sr[im] = 1.0f;
            if constexpr (Epi::kHasRow) {
                if (m < M) sr[im] = epi.row(m);
            }
        }
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t n = nw0 + in * 32 + 8 * g + 4 * hi;
                sc[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
                bb[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (n < N) epi.cols(n, N, sc[in][g], bb[in][g]);
            }
    }
    char *const outb = (char *)epi.out;
    using acc4_t = typename Epi::Mma::acc4_t;
    if constexpr (EB == 2) {
        if constexpr (L16) {
            staged_write16<NTM>(epi, get, mw0, nw0, lane, M, N, stage, 0);
        } else {
#pragma unroll
            for (int im = 0; im < NTM; ++im) {
                const int row = 32 * im + ml;
#pragma unroll
                for (int in = 0; in < 2; ++in) {
                    const auto a = get(in, im);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v2u v = epi.pack((acc4_t){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}, sr[im], sc[in][g], bb[in][g]);
                        // ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks: rows 2j and 2j+1 share a chunk
                        // position, so odd rows take the other 8-byte half (without it: 2-way conflict, SQ_LDS_BANK_CONFLICT)
                        *(lds_u2)(uintptr_t)(stage + row * 128 + (((in * 4 + g) ^ ((row >> 1) & 7)) << 4) + 8 * (hi ^ (row & 1))) = v;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4 * NTM; ++i) {
            const int row = 8 * i + (lane >> 3);
            v4i v = *(lds_v4i)(uintptr_t)(stage + i * 1024 + lane * 16);
            if ((lane >> 3) & 1) v = (v4i){v[2], v[3], v[0], v[1]};  // odd rows were staged with their 8-byte halves flipped
            const int64_t m = mw0 + row, n = nw0 + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
            if (m < M && n < N) store16_policy<POL>(outb + (m * epi.N + n) * 2, v);  // (epi.N: the row stride; N bounds the columns of this launch)
        }
    } else {
        constexpr int TP = NTM >= 2 ? 2 : 1;  // 32-row tiles per staging pass (4-byte outputs: 64 rows fill the wave's 16 KiB)
#pragma unroll
        for (int h = 0; h < NTM / TP; ++h) {
            if constexpr (L16) {
                staged_write16<NTM>(epi, get, mw0, nw0, lane, M, N, stage, h);
            } else {
#pragma unroll
                for (int imh = 0; imh < TP; ++imh) {
                    const int im = TP * h + imh, row = 32 * imh + ml;
#pragma unroll
                    for (int in = 0; in < 2; ++in) {
                        const auto a = get(in, im);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const v4i v = epi.pack((acc4_t){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}, sr[im], sc[in][g], bb[in][g]);
                            *(lds_v4i)(uintptr_t)(stage + row * 256 + (((in * 8 + 2 * g + hi) ^ (row & 15)) << 4)) = v;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8 * TP; ++i) {
                const int row = 4 * i + (lane >> 4);
                const v4i v = *(lds_v4i)(uintptr_t)(stage + i * 1024 + lane * 16);
                const int64_t m = mw0 + 32 * TP * h + row, n = nw0 + (((lane & 15) ^ (row & 15)) << 2);
                if (m < M && n < N) store16_policy<POL>(outb + (m * epi.N + n) * 4, v);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
    }
}

// Interior fast path of the staged epilogue (2-byte outputs): the wave tile lies completely inside the output (no row /
// column bound anywhere), so every per-store compare-and-branch of epilogue_wave_staged disappears, and the three stages
// are software-pipelined over the NTM 32-row token tiles instead of running back to back:
//
//     pack(0);   for im = 0 .. NTM-1:   issue the LDS reads of tile im  |  pack(im + 1)  |  global stores of tile im
//
// so the LDS round trip of tile im is covered by the conversions of tile im + 1, and the stores leave the wave spaced by VALU
// work instead of in one burst.  What round 3 measured on the old form (tools/ubench/valu_rate + the ISA): the conversions
// themselves are ~2.7k cycles per SIMD for a 256 x 256 fp16 tile, but every store instruction was its own serial chain --
// bound compare + branch, ds_read_b128, s_waitcnt lgkmcnt(0) with the LDS latency exposed, four v_cndmask for the half flip,
// a 64-bit VALU address -- ~160 cycles per store with one wave per SIMD (gemm_i8_p4: 13.6k-cycle epilogue), half hidden
// with two (gemm_i8_p8: 7.7k).  Here a store costs two ds_read_b64 (the half flip is in the read ADDRESS), two SALU adds for
// the row base and one global_store_dwordx4 with an SGPR base and a per-lane 32-bit offset computed once.
// Image of one token tile: [32 rows][NTN x 64 B], NTM images side by side (no reuse: no wait for a previous tile's reads).
//   NTN = 2 (128-B rows): chunk position = c ^ ((row >> 1) & 7), 8-byte halves flipped on odd rows          (as the slow path)
//   NTN = 4 (256-B rows): chunk position p = c ^ (row & 15),     8-byte halves flipped for p >= 8            (as gemm_i8_p4's)
// Both make the transposing ds_write_b64 (16 consecutive rows, one chunk) and the lane-linear reads conflict-free, and in both
// a lane of the read side is flipped iff (lane >> 3) & 1.  Needs NTM * 32 * NTN * 64 bytes of wave-private LDS at `stage`.
// L16: the accumulators are in the 16 x 16 layout of v_mfma_i32_16x16x64_i8 (gemm_i8_p16): lane l owns token (l & 15) and channels 4 * (l >> 4) .. + 3 of
// 16 x 16 tiles, get(in16, im16) returns 4 registers.  The images, the read side and the stores are the same; a lane writes, for 16-channel tile in16 and
// 16-token tile im16 = 2 * im + h, the 8 bytes of chunk 2 * in16 + (q >> 1), half q & 1 of row 16 * h + t (16 consecutive lanes = 16 rows, one chunk:
// conflict-free under the same swizzle).
// wt (wave-uniform): the stores leave as write-through (sc0 sc1) buffer stores.  A launch whose whole output stays dirty in the XCDs' L2s pays for it at its end --
// the end-of-kernel release writes it back, 5.5 us between two 4096^3 launches (profiles/r3_clock_probe_p16.txt) -- so callers set it when the output is small enough
// for that to matter (rows_write_through: <= 128 MiB); large outputs overflow the L2s during the launch anyway and are better off write-back (r3_store_policy_ab.txt).
// (builtin buffer stores, offset in the VGPR, soffset 0: the hazard recogniser sees them -- an inline-asm store lets the next tile's ds_reads overwrite its data.)
#ifndef ASQ_WT_BYTES
#define ASQ_WT_BYTES (128ll << 20)   // outputs up to this size leave as write-through stores (`wt` below); 0: never
#endif
template <class Epi> __device__ __forceinline__ bool rows_write_through(int64_t M, const Epi &epi)
{
    return ASQ_WT_BYTES > 0 && M * epi.N * 2 <= (int64_t)ASQ_WT_BYTES && epi.N < (int64_t(1) << 23);
}
// IMGS: token-tile images at `stage` (default: one per tile, side by side; 2: two images used alternately -- half the staging space, for the persistent kernel whose
// other LDS stage already holds the next tile's operands; a wave's LDS operations execute in order, so rewriting an image behind its own reads is safe).
// DRAIN: `s_waitcnt vmcnt(0)` between the first tile's conversions and the first store (gemm_i8_p16p: retires the next tile's in-flight operand DMAs at a point
// where only they are outstanding -- behind ~500 cycles of VALU work, before any store joins the counter).
// IM0, IM1: only the token tiles [IM0, IM1) of the wave tile (gemm_i8_p16's tail: the wave tile's upper 64 rows leave while the lower 64 still accumulate).
template <int NTM, int NTN, bool L16 = false, int IMGS = NTM, bool DRAIN = false, int IM0 = 0, int IM1 = NTM, class Epi, class Get>
__device__ __forceinline__ void epilogue_wave_rows(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, unsigned stage, bool wt = false)
{
    static_assert(0 <= IM0 && IM0 < IM1 && IM1 <= NTM, "token-tile range");
    static_assert(Epi::kOutBytes == 2, "2-byte outputs");
    static_assert(NTN == 2 || NTN == 4, "wave tile of 64 or 128 channels");
    typedef __attribute__((address_space(3))) v2u *lds_u2;
    constexpr int ROWB = NTN * 64;     // bytes per image row
    constexpr int IMG = 32 * ROWB;     // one token tile: 4 / 8 KiB
    constexpr int CH = ROWB / 16;      // 16-byte chunks per row: 8 / 16
    constexpr int RPR = 1024 / ROWB;   // rows per 1-KiB read: 8 / 4
    constexpr int NRD = 32 / RPR;      // reads per token tile: 4 / 8
    constexpr int NV = NTN == 2 ? 2 : 4;  // distinct per-lane column offsets over the reads of a tile
    const int ml = L16 ? (lane & 15) : (lane & 31), hi = L16 ? (lane >> 4) : (lane >> 5);   // L16: token within the tile, channel quad 0..3
    constexpr int NSR = L16 ? 2 * NTM : NTM;   // row scales: one per 16- / 32-token tile
    constexpr int NSC = L16 ? 1 : 4;           // column vectors per 16- / 32-channel tile

    float sr[NSR];
#pragma unroll
    for (int i = 0; i < NSR; ++i) sr[i] = (Epi::kHasRow && i >= IM0 * (NSR / NTM) && i < IM1 * (NSR / NTM)) ? epi.row(mw0 + i * (L16 ? 16 : 32) + ml) : 1.0f;
    v4f sc[L16 ? 2 * NTN : NTN][NSC], bb[L16 ? 2 * NTN : NTN][NSC];
    if constexpr (L16) {
#pragma unroll
        for (int in = 0; in < 2 * NTN; ++in) epi.cols(nw0 + in * 16 + 4 * hi, nw0 + NTN * 32, sc[in][0], bb[in][0]);
    } else {
#pragma unroll
        for (int in = 0; in < NTN; ++in)
#pragma unroll
            for (int g = 0; g < 4; ++g) epi.cols(nw0 + in * 32 + 8 * g + 4 * hi, nw0 + NTN * 32, sc[in][g], bb[in][g]);
    }

    // write addresses: one per chunk c (L16: per 16-channel tile), reused by every token tile through the immediate offset
    unsigned wa[CH];
    {
        unsigned w0;
        if constexpr (L16 && NTN == 2) w0 = ml * ROWB + ((((hi >> 1) ^ (ml >> 1)) & 7) << 4) + 8 * ((hi ^ ml) & 1);
        else if constexpr (L16) w0 = ml * ROWB + (((hi >> 1) ^ ml) << 4) + 8 * ((hi & 1) ^ (ml >> 3));   // 256-B rows: position = chunk ^ row, halves flipped for positions >= 8
        else if constexpr (NTN == 2) w0 = ml * ROWB + (((ml >> 1) & 7) << 4) + 8 * (hi ^ (ml & 1));
        else w0 = ml * ROWB + ((ml & 15) << 4) + 8 * (hi ^ ((ml >> 3) & 1));
#pragma unroll
        for (int c = 0; c < CH; ++c)   // (L16: c = in16, chunk 2 * in16 + (q >> 1); only 2 * NTN of them used)
            wa[c] = stage + (w0 ^ (unsigned)(L16 ? ((c << 5) | (NTN == 4 ? (c >> 2) << 3 : 0)) : ((c << 4) | (NTN == 4 ? (c >> 3) << 3 : 0))));
    }
    // read addresses (lane-linear, the 8-byte halves swapped for flipped lanes) and the lane's byte offsets in the output
    // (one base register per read of a tile, opaque to the compiler: with a shared base it fuses the reads of two ROWS into one ds_read2st64_b64 and
    // then needs four v_mov per store to bring a row's halves together; this way a row's two ds_read_b64 land in the store's register quad)
    const unsigned flip = (lane >> 3) & 1;
    unsigned ra0[NRD], ra1[NRD];
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
        ra0[i] = stage + lane * 16 + 8 * flip + i * 1024;
        ra1[i] = ra0[i] ^ 8;
        asm volatile("" : "+v"(ra0[i]), "+v"(ra1[i]));
    }
    const unsigned ldb = __builtin_amdgcn_readfirstlane((unsigned)(epi.N * 2));   // row pitch in bytes (the fast path requires 8 * ldb < 2^31)
    const unsigned cb0 = (unsigned)(((lane & (CH - 1)) ^ (lane >> 4)) << 4);
    unsigned voff[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) voff[j] = (unsigned)(lane / CH) * ldb + (cb0 ^ (unsigned)(64 * j));
    typedef __attribute__((address_space(1))) v4i *glb_v4i;
    const uint64_t tile = (uint64_t)(uintptr_t)uniform_ptr((const int8_t *)epi.out + (mw0 * epi.N + nw0) * 2);

    const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)tile, 0, 0x7FFFFFFF, 0x00020000);   // (wt: offsets < 128 * ldb, the caller keeps ldb < 2^24)

    using acc4_t = typename Epi::Mma::acc4_t;
    auto pack_tile = [&](int im) {
        if constexpr (L16) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int in = 0; in < 2 * NTN; ++in)
                    *(lds_u2)(uintptr_t)(wa[in] + (im % IMGS) * IMG + h * 16 * ROWB) = epi.pack(get(in, 2 * im + h), sr[2 * im + h], sc[in][0], bb[in][0]);
        } else {
#pragma unroll
            for (int in = 0; in < NTN; ++in) {
                const auto a = get(in, im);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(lds_u2)(uintptr_t)(wa[in * 4 + g] + (im % IMGS) * IMG) = epi.pack((acc4_t){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}, sr[im], sc[in][g], bb[in][g]);
            }
        }
    };
    pack_tile(IM0);
    if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int im = IM0; im < IM1; ++im) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        v2u lo[NRD], up[NRD];
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            lo[i] = *(lds_u2)(uintptr_t)(ra0[i] + (im % IMGS) * IMG);
            up[i] = *(lds_u2)(uintptr_t)(ra1[i] + (im % IMGS) * IMG);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (im + 1 < IM1) pack_tile(im + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            const uint64_t rowbase = tile + (uint64_t)(im * 32 + i * RPR) * ldb;   // SGPR pair: two scalar adds per store
            const v4i v = {(int)lo[i][0], (int)lo[i][1], (int)up[i][0], (int)up[i][1]};
            if (wt) {
                typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, v), wt_rsrc, (unsigned)(im * 32 + i * RPR) * ldb + voff[i % NV], 0, 17 /* sc0 sc1 */);
            } else {
                *(glb_v4i)(uintptr_t)(rowbase + voff[i % NV]) = v;   // global_store_dwordx4 voff, data, s[base]
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same interior fast path for 4-BYTE outputs (int32 accumulators, fp32) of the 16 x 16 accumulator layout (gemm_i8_p16, round 4): a 32-row token tile is an
// 8 KiB image [32 rows][64 channels x 4 B], two of them alternate in the wave's 16 KiB, software-pipelined like epilogue_wave_rows:
//     pack(0);   for im = 0 .. 3:   issue the 8 ds_read_b128 of image im & 1  |  pack(im + 1) into the other image  |  8 global_store_dwordx4
// A lane (token t = l & 15, channel quad q = l >> 4) writes the 16 bytes of chunk 4 * in16 + q of row 16 * h + t at position chunk ^ (row & 15) (ds_write_b128,
// conflict-free: staged_write16's layout); the read side is lane-linear, lane l of read i holds row 4 * i + (l >> 4), chunk (l & 15) ^ (row & 15).
// Before: two unpipelined 64-row passes with a bound compare and a 64-bit address per store (epilogue_wave_staged) on gemm_i8_p8's L16 mode -- asq_gemm_i8_i32
// at 4096^3 62.8 us against 54.9 for the fp16 kernel (profiles/r3_vendor_compare.txt).
template <class Epi, class Get>
__device__ __forceinline__ void epilogue_wave_rows4(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, unsigned stage, bool wt = false)
{
    static_assert(Epi::kOutBytes == 4, "4-byte outputs");
    typedef __attribute__((address_space(3))) v4i *lds_v4i;
    const int t = lane & 15, q = lane >> 4;
    float sr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sr[i] = Epi::kHasRow ? epi.row(mw0 + i * 16 + t) : 1.0f;
    v4f sc[4], bb[4];
#pragma unroll
    for (int in = 0; in < 4; ++in) {
        sc[in] = bb[in] = (v4f){0.f, 0.f, 0.f, 0.f};
        epi.cols(nw0 + in * 16 + 4 * q, nw0 + 64, sc[in], bb[in]);
    }
    unsigned wa[4];   // write address of (in16, row t) in image 0; + 8192 for image 1, + 4096 for the second 16 rows
#pragma unroll
    for (int in = 0; in < 4; ++in) wa[in] = stage + t * 256 + (((4 * in + q) ^ t) << 4);
    unsigned ra[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ra[i] = stage + i * 1024 + lane * 16;
        asm volatile("" : "+v"(ra[i]));
    }
    const unsigned ldb = __builtin_amdgcn_readfirstlane((unsigned)(epi.N * 4));   // row pitch in bytes (callers keep 128 * ldb < 2^31)
    unsigned voff[4];   // per-lane byte offset of read i (rows 4 i + (l >> 4)): the column part has period 4 in i
#pragma unroll
    for (int j = 0; j < 4; ++j) voff[j] = (unsigned)q * ldb + (unsigned)((((lane & 15) ^ ((4 * j + q) & 15))) << 4);
    typedef __attribute__((address_space(1))) v4i *glb_v4i;
    const uint64_t tile = (uint64_t)(uintptr_t)uniform_ptr((const int8_t *)epi.out + (mw0 * epi.N + nw0) * 4);
    const __amdgpu_buffer_rsrc_t wt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)tile, 0, 0x7FFFFFFF, 0x00020000);
    auto pack_tile = [&](int im) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int in = 0; in < 4; ++in)
                *(lds_v4i)(uintptr_t)(wa[in] + (im & 1) * 8192 + h * 4096) = epi.pack(get(in, 2 * im + h), sr[2 * im + h], sc[in], bb[in]);
    };
    pack_tile(0);
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        // (the reads of the first 16 rows are in flight under the conversions of the next tile; the second 16 rows follow the first stores: 16 registers of
        // staged data instead of 32 -- the fp32 column-scale + bias epilogue on offset operands has none to spare)
        v4i v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(lds_v4i)(uintptr_t)(ra[i] + (im & 1) * 8192);
        __builtin_amdgcn_sched_barrier(0);
        if (im + 1 < 4) pack_tile(im + 1);
        __builtin_amdgcn_sched_barrier(0);
        auto store4 = [&](int i0) {
#pragma unroll
            for (int i = i0; i < i0 + 4; ++i) {
                const unsigned rowoff = (unsigned)(im * 32 + i * 4) * ldb;
                if (wt) {
                    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, v[i - i0]), wt_rsrc, rowoff + voff[i & 3], 0, 17 /* sc0 sc1 */);
                } else {
                    *(glb_v4i)(uintptr_t)(tile + (uint64_t)rowoff + voff[i & 3]) = v[i - i0];
                }
            }
        };
        store4(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(lds_v4i)(uintptr_t)(ra[4 + i] + (im & 1) * 8192);
        store4(4);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------
// "generic": any M, N, K, any alignment.  64x64x64 tile, 4 waves (2x2), single LDS buffer.
// Correctness net for odd shapes (K % 128 != 0, unaligned rows); not a tuned kernel.
// ---------------------------------------------------------------------------------
constexpr int GEN_T = 64, GEN_LD = 80;  // 64 k-bytes + 16 pad per row

__device__ __forceinline__ v4i load16_guarded(const int8_t *base, int64_t ld, int64_t row, int64_t nrows, int64_t k, int64_t K, bool fast)
{
    v4i v = {0, 0, 0, 0};
    if (row >= nrows || k >= K) return v;
    const int8_t *p = base + row * ld + k;
    if (fast && k + 16 <= K) return *(const v4i *)p;
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i)
        if (k + i < K) w[i >> 2] |= (uint32_t)(uint8_t)p[i] << (8 * (i & 3));
    return (v4i){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
}

template <class Epi>
__global__ void __launch_bounds__(256) gemm_i8_generic(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N,
                                                       int64_t K, bool fast, Epi epi)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * GEN_T * GEN_LD];
    char *xs = lds, *ws = lds + GEN_T * GEN_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GEN_T, n0 = (int64_t)blockIdx.x * GEN_T;
    const int lrow = tid >> 2, lchunk = tid & 3;
    using MMA = typename Epi::Mma;
    typename MMA::acc_t acc = {0};
    for (int64_t k0 = 0; k0 < K; k0 += GEN_T) {
        v4i vx = load16_guarded(x, K, m0 + lrow, M, k0 + lchunk * 16, K, fast);
        v4i vw = load16_guarded(w, K, n0 + lrow, N, k0 + lchunk * 16, K, fast);
        __syncthreads();
        *(v4i *)(xs + lrow * GEN_LD + lchunk * 16) = vx;
        *(v4i *)(ws + lrow * GEN_LD + lchunk * 16) = vw;
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i a = *(const v4i *)(ws + (wn * 32 + (lane & 31)) * GEN_LD + ks * 32 + (lane >> 5) * 16);
            v4i b = *(const v4i *)(xs + (wm * 32 + (lane & 31)) * GEN_LD + ks * 32 + (lane >> 5) * 16);
            acc = MMA::mma(a, b, acc);
        }
    }
    epilogue_wave<1, 1>(
        epi, [&](int, int) -> const typename MMA::acc_t & { return acc; }, [](int) { return 0; }, m0 + wm * 32, n0 + wn * 32, lane, M, N);
}

// bijective XCD-aware remap: consecutive logical ids land on the same XCD (its private L2
// then serves the operand panels shared by neighbouring tiles)
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// workspace header (asq_workspace_init): magic word + arrival tickets of the in-launch reductions (asq_gemm_wstream.h; grouped tail split of asq_gemm_p8.h)
constexpr int WS_HEADER_BYTES = 8192;
constexpr unsigned long long WS_MAGIC = 0x4153515753763031ull;  // "ASQWSv01"
constexpr int WS_MAX_GROUPS = (WS_HEADER_BYTES - 16) / 4;

// OFFSET operands (asq_linear_w8a8_off; include/asq_hip.h "offset operand images").  Under the socket power limit the 256 x 256 GEMM's time is its
// energy, and most of the matrix cores' data-dependent energy is two's-complement sign extension (profiles/r2_clock_power_evidence.md section 3,
// profiles/r4_operand_offsets.md): weights around 0 and activations around 0 flip the high bits of operands, products and partial sums all the time.
// The int8 MFMA is signed x signed only, but an offset per ROW is exact and never clamps when it is chosen from the row's own extremes:
//     x'[m,k] = x[m,k] + cx[m],  w'[n,k] = w[n,k] + cw[n]
//     sum_k x[m,k] w[n,k] = sum_k x'[m,k] w'[n,k]  -  cx[m] * sum_k w[n,k]  -  cw[n] * sum_k x'[m,k]
// The kernel multiplies the primed matrices and starts its accumulators at the two (rank-1) correction terms: K loop and epilogue are unchanged and the
// int32 result is bit-identical to the plain product (two's-complement wrap-around everywhere, also inside v_mfma_i32_*: tools/ubench/mfma_bias_power).
struct OffsetArgs {
    const int32_t *row = nullptr;  // [M][2]: {cx[m], sum_k x'[m,k]}       (written by the activation quantiser, asq_quantize_act_off)
    const int32_t *col = nullptr;  // [N][2]: {cw[n], sum_k w[n,k]}        (sum of the ORIGINAL weight row; asq_weight_offset_image)
};
constexpr int64_t OFFSET_MAX_K = 65536;   // the start values are formed with 24-bit multiplies: |sum_k| <= 128 K < 2^23 + 1

// epilogues of the gate || up GEMMs (asq_gemm_gateup.h) mark themselves with kGateUp; the kernels that carry them select their epilogue on this trait
template <class Epi, class = void> struct IsGateUp : std::false_type {};
template <class Epi> struct IsGateUp<Epi, std::enable_if_t<Epi::kGateUp>> : std::true_type {};

// ---- K splits reduced INSIDE the launch (round 5; gemm_i8_p8q2<Epi, true>; asq_gemm_p8q2.h has the story) -------------------
// Work item of block b = 8 * slot + xcd: XCD xcd owns tiles [base, base + cnt) (xcd_remap's shares); its blocks take (tile, split) = (base + slot / S, slot % S).
// false: an idle block of the rounded-up grid (8 x the largest share).
__device__ __forceinline__ bool splitk_fix_item(int ntiles, int ksplit, int &id, int &split)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int cnt = q + (xcd < r ? 1 : 0), base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    if (slot >= cnt * ksplit) return false;
    id = base + slot / ksplit;
    split = slot - (slot / ksplit) * ksplit;
    return true;
}
// The tail: every block stores its NV x 16 B accumulator registers per lane as a write-through register image behind the workspace header ((tile, split) -> image
// tile * S + split; the layout is the registers' own order), takes the tile's ticket in the header (one agent-scope atomic; the last arriver puts it back to zero, so the
// next launch and a captured graph's replay find it clean), and the last arriver adds the other images -- three images' loads in flight together where the registers
// allow -- and returns true: it runs the caller's epilogue.  The others return false.  No block waits for another; any placement of the blocks is correct.
// `lds`: 4 bytes of block-shared scratch (the operand ring is dead by now).  A header without the magic word or a ticket above S traps (asq_workspace_init contract).
template <int NV, class AccAt>
__device__ __forceinline__ bool splitk_fix_reduce(char *gws, unsigned long long ws_magic, int id, int split, int ksplit, int tid, char *lds, AccAt at)
{
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gws + WS_HEADER_BYTES, 0, 0x7FFFFFFF, 0x00020000);
    constexpr int PART = NV * 8192;   // one register image: NV x (512 lanes x 16 B)
    const int my = (id * ksplit + split) * PART + tid * 16;
    // (offset in the VGPR, soffset 0 and a wait state after the stores: see the note on buffer stores in asq_gemm_wstream.h)
#pragma unroll
    for (int i = 0; i < NV; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, at(i)), rsrc, my + i * 8192, 0, 16 /* sc1: write-through */);
    asm volatile("s_nop 1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's image has been acknowledged before the block takes its ticket
    unsigned *const flag = (unsigned *)lds;
    if (tid == 0) {
        unsigned *const tk = (unsigned *)gws + 4 + id;
        if (ws_magic != WS_MAGIC) __builtin_trap();   // workspace never went through asq_workspace_init
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)ksplit - 1) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch / replay
        if (old >= (unsigned)ksplit) __builtin_trap();  // poisoned tickets: header not initialised / workspace shared by concurrent launches
        *flag = old == (unsigned)ksplit - 1 ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();  // (the staged epilogue reuses this LDS)
    if (!last) return false;
    // the other S - 1 images, up to MAXB at a time (<= 24 x 16 B per lane in flight): image j of the sequence 0 .. S - 2 is split j + (j >= split)
    constexpr int MAXB = NV <= 8 ? 3 : 1;
    auto add_images = [&](auto cnt_tag, int j0) {
        constexpr int CNT = decltype(cnt_tag)::value;
        v4u_ v[CNT][NV];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int j = j0 + u, sp = j + (j >= split ? 1 : 0);
            const int src = (id * ksplit + sp) * PART + tid * 16;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, src + i * 8192, 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) at(i) += __builtin_bit_cast(v4i, v[u][i]);
    };
    int j0 = 0;
    if constexpr (MAXB == 3) {
        for (; j0 + 3 <= ksplit - 1; j0 += 3) add_images(std::integral_constant<int, 3>{}, j0);
        if (ksplit - 1 - j0 == 2) add_images(std::integral_constant<int, 2>{}, j0);
        else if (ksplit - 1 - j0 == 1) add_images(std::integral_constant<int, 1>{}, j0);
    } else {
        for (; j0 < ksplit - 1; ++j0) add_images(std::integral_constant<int, 1>{}, j0);
    }
    return true;
}

}  // namespace asq

#include "asq_gemm_p8.h"
#include "asq_gemm_p4.h"
#include "asq_gemm_p16.h"
#ifndef ASQ_P16_TAIL
#define ASQ_P16_TAIL 0   // 1: also build gemm_i8_p16t (asq_gemm_p16t.h: round 6, bit-identical to gemm_i8_p16 and no faster -- not shipped) and dispatch it
#endif
#if ASQ_P16_TAIL
#include "asq_gemm_p16t.h"
#endif
#include "asq_gemm_gateup.h"
#include "asq_gemm_p16p.h"
#include "asq_gemm_p4x16.h"
#include "asq_gemm_p8h.h"
#include "asq_gemm_p8q.h"
#include "asq_gemm_p8q2.h"
#include "asq_gemm_skinny.h"
#include "asq_gemm_wstream.h"

namespace asq {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (thread, device, kernel) instead of once per launch.  The cache is
// thread-local: launches take no lock (include/asq_hip.h promises a re-entrant, lock-free C-ABI); a second thread
// repeats the idempotent attribute call once.
static inline hipError_t ensure_dynamic_lds(const void *kfn, int bytes)
{
    static thread_local std::unordered_map<uintptr_t, int> done;  // key: kernel address ^ (device << 56): the attribute is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uintptr_t key = (uintptr_t)kfn ^ ((uintptr_t)(dev + 1) << 56);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}

// ---------------------------------------------------------------------------------
// split-K tail: sum S int32 slabs [S][M][N] (exact, order-free) and run the fused epilogue.
// One thread per 4 consecutive channels of one token; N % 4 == 0 (enforced by the launcher).
// ---------------------------------------------------------------------------------
template <class Epi>
__global__ void __launch_bounds__(256) splitk_reduce(const int32_t *__restrict__ slabs, int S, int64_t M, int64_t N, Epi epi)
{
    const int64_t nq = N / 4, total = M * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / nq, n = (i - m * nq) * 4;
        const int32_t *p = slabs + m * N + n;
        v4i a = *(const v4i *)p;
        for (int s = 1; s < S; ++s) a += *(const v4i *)(p + (int64_t)s * M * N);
        const float sr = Epi::kHasRow ? epi.row(m) : 1.0f;
        v4f sc = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
        epi.cols(n, N, sc, bb);
        epi.store4(m, n, a, sr, sc, bb, N);
    }
}

// ---------------------------------------------------------------------------------
// dispatch + launch
// ---------------------------------------------------------------------------------
enum GemmKernel { KERN_GENERIC = 0, KERN_SKINNY = 1, KERN_P8 = 2, KERN_P8H = 3, KERN_P4 = 4, KERN_P8Q = 5, KERN_P16 = 6, KERN_P4X16 = 7 };

// ASQ_P8Q2=0: the 128 x 128 int8 launches stay on gemm_i8_p8q (one barrier per K-tile) instead of gemm_i8_p8q2 (two staggered wave groups); A/B switch, read once
static inline int p8q2_enabled()   // 0: gemm_i8_p8q; 1: gemm_i8_p8q2
{
    static const int on = [] {
        const char *e = getenv("ASQ_P8Q2");
        return e ? atoi(e) : 1;
    }();
    return on;
}

int forced_kernel();  // env ASQ_GEMM_KERNEL=generic|skinny|p8|p8h (development / A-B aid), asq_gemm.hip

static inline GemmKernel pick_kernel(const void *x, const void *w, int64_t M, int64_t N, int64_t K)
{
    const bool aligned = ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
    const bool tiled_ok = aligned && K % 128 == 0 && K >= 128 && K <= (1 << 24);
    const int f = forced_kernel();
    if (f == KERN_GENERIC) return KERN_GENERIC;
    if (tiled_ok && (f == KERN_P8 || f == KERN_P8H || f == KERN_P4 || f == KERN_P8Q || f == KERN_P16 || f == KERN_P4X16)) return (GemmKernel)f;
    if (tiled_ok && M <= 1024 && M * K < (1ll << 32) && f == KERN_SKINNY) return KERN_SKINNY;  // (32-bit row offsets in the DMA address)
    if (tiled_ok && f < 0) {
        // measured crossover (tools/cold_grid.sh: 48..256 rows x 8 LLaMA/OPT/Mixtral weight shapes, weights rotated
        // through > 256 MiB so they come from HBM, not the Infinity Cache): the weight-streaming kernel re-reads X
        // from L2 once per 16 (or 32) channels, so it wins while the total work N*K*M stays small; wide-N
        // weights (11008x4096, 14336x4096, 20480x5120) stay ahead longer than square or long-K ones
        const double work = (double)N * (double)K * (double)M;
        const int64_t th = ((M + 127) / 128) * ((N + 255) / 256);  // tiles of 128 x 256
        // (against the 128 x 128 kernel the square-ish crossover sits lower once there are more than 64 rows: 96x5120x5120 17.1 -> 14.5 us,
        // 128x5120x5120 19.5 -> 15.5, while 128x4096x4096 stays with the stream, 11.7 vs 15.0)
        if (M <= 256 && work <= (N > 2 * K ? 5.8e9 : M > 64 ? 2.4e9 : 4.0e9) && !(M > 160 && th >= 16))
            return KERN_SKINNY;  // (measured up to 256 rows; with more than 160 rows the 128 x 128 kernel is 5-10 % ahead once it has >= 32 tiles)
        // 128-row tiles when the 256-row tiling cannot fill 256 CUs (or wastes half a tile row).  Measured
        // (tools/ksplit_sweep.sh): p8h is ~14 % slower per op on a full chip but wins up to 1.45x below ~144 tiles
        const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
        // two one-round corners (round 4, profiles/r4_dispatch_holes.txt, forced kernels on cold weights): 129..143 tiles of 256 x 256 whose 128 x 256 tiling no
        // longer fits ONE round run faster as one (partly filled) round of p16 than as p8h + a peeled remainder (768 x 11008 x 4096: 40.4 -> 37.3 us); and
        // >= 144 tiles whose last 256-row tile row is at most half full while the 128 x 256 tiling fits one round stay with p8h (640 x 12288 x 4096: 37.5 -> 28.6)
        const bool one_round_p16 = t256 >= 128 && th > 256;
        const bool odd_half_row = t256 >= 144 && th <= 256 && (((M + 127) / 128) & 1) != 0 && K < 16384;
        if ((t256 < 144 && !one_round_p16) || odd_half_row) {
            if (odd_half_row) return KERN_P8H;
            // 128 x 128 tiles (p8q) where the 128 x 256 tiling has at most 128 tiles, i.e. leaves half of the CUs without one: twice the
            // tiles at twice the L2->LDS bytes per MFMA.  Measured (tools/kbench.py, forced vs default, 48 shapes): -3 ... -22 % for 32..128
            // p8h tiles (512 x 4096 x 4096: 22.8 -> 17.7 us), +15 ... +35 % above 128.
            // (with its cost-model K split p8q also wins 5-13 % at 16..32 tiles and a long K -- 128x4096x11008 20.4 -> 19.0 us; between 40 and 80
            // tiles at K >= 8192 the two were within 4 % either way and p8h kept them -- round 5: with gemm_i8_p8q2 the 128 x 128 tiles are ahead there too,
            // 512 x 4096 x 11008 32.6 -> 30.6 us, 384 rows 27.3 -> 26.7 (profiles/r5_midsize_forced_kernels.txt), so the exception only remains for the
            // one-barrier kernel (ASQ_P8Q2=0); OPT's K = 20480 stays with p8h's deeper split)
            if (th >= 16 && th <= 128 && K < 16384 && !(!p8q2_enabled() && K >= 8192 && th >= 40 && th < 80)) return KERN_P8Q;
            return KERN_P8H;
        }
        // 256 x 256 tiles: gemm_i8_p16, p8's schedule on v_mfma_i32_16x16x64_i8.  Under the socket power limit the GEMM's time is its energy, and the
        // 16 x 16 x 64 instruction moves half the accumulator bytes per MAC: measured against p8 in one process (profiles/r3_p8_vs_p16_ab.txt, bit-identical
        // outputs) 4096^3 56.8 -> 51.1 us, 8192 x 4096 x 4096 -8 %, 16384 x 12288 x 4096 -8 %, 4096 x 4096 x 8192 102.3 -> 92.0 (p4, the round-2 choice for
        // K >= 8192 -- 128 x 128 per wave, a third fewer fragment bytes -- gained 1.7-3 % there).  launch_gemm sends what p16 does not carry (4-byte and
        // int8 outputs, fp8 operands, K splits, grouped launches) to p8; p8 and p4 stay reachable through ASQ_GEMM_KERNEL.
        return KERN_P16;
    }
    return KERN_GENERIC;
}

// shapes gemm_i8_p16 can run with offset operands (whether it SHOULD is offsets_profitable: the dispatcher's own choice of p16)
static inline bool offsets_shape_ok(const void *x, const void *w, int64_t M, int64_t N, int64_t K)
{
    const bool aligned = ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
    return aligned && K % 128 == 0 && K >= 128 && K <= OFFSET_MAX_K && N % 4 == 0 && N >= 4 && M >= 1;
}

static inline int forced_ksplit()  // env ASQ_KSPLIT=n forces a split count (development / tuning aid)
{
    static int forced = -2;
    if (forced == -2) {
        const char *e = getenv("ASQ_KSPLIT");
        forced = e ? atoi(e) : -1;
    }
    return forced;
}

// number of K splits for the tiled kernel: fill the 256 CUs when the M x N tile grid cannot
static inline int pick_ksplit(int64_t tiles, int64_t K, int64_t M, int64_t N, size_t ws_bytes)
{
    if (N % 4 != 0) return 1;
    const int64_t nt = K / 128;
    {
        const int forced = forced_ksplit();
        if (forced > 0) {
            int64_t f = forced > nt ? nt : forced;
            while (f > 1 && (size_t)f * (size_t)M * (size_t)N * 4 > ws_bytes) --f;
            return (int)f;
        }
    }
    // measured on MI355X (ASQ_KSPLIT sweep, 128..2048 rows x LLaMA/OPT widths): the slab write +
    // reduce pass costs ~2 x S x M x N x 4 B of traffic, so the optimum is ~130-200 blocks, not 256
    if (tiles >= 118) return 1;
    int64_t s = (176 + tiles / 2) / tiles;
    if (s > nt / 4) s = nt / 4;              // >= 4 K-tiles (512 k) per split: keep the pipeline efficient
    while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4 > ws_bytes) --s;
    return s < 1 ? 1 : (int)s;
}

constexpr double P8Q_FIX_TAIL_US = 4.4, P8Q_FIX_PER_SPLIT_US = 1.1;   // in-launch tail of a tile split in two; per further split (fitted: profiles/r5_splitk_fix_sweep.txt)
// K splits for the 128 x 128 kernel, from a small cost model fitted to measurements (us): a block costs 3 + (K-tiles) x (0.40 + 0.20 x the
// fraction of the 256 CUs that hold a block -- the L2->LDS path is shared), a split launch adds the reduce pass, 5 + S x M x N x 4 B at
// 3 TB/s.  512x4096x4096: S = 1 (18.0 us measured; S = 2: 21.3); 512x4096x11008: S = 2 (37.3; S = 1: 40.1); 128x4096x11008: S = 7 (21.9).
// ASQ_SPLITK_FIX=0: K splits of the 128 x 128 kernel go back to slab launch + reduce launch (A/B switch).  Default 1: reduced inside the launch (asq_gemm_p8q2.h).
static inline int splitk_fix_mode()
{
    static const int m = [] {
        const char *e = getenv("ASQ_SPLITK_FIX");
        return e ? atoi(e) : 1;
    }();
    return m;
}
// scratch bytes of the in-launch form: one 64 KiB register image per (tile, split)
static inline size_t p8q_fix_bytes(int64_t tiles, int64_t s) { return (size_t)tiles * (size_t)s * 65536; }

// `fix`: the caller can run the in-launch reduction (gemm_i8_p8q2<Epi, true>): a split then costs its tail (write-through image stores, one ticket, S - 1 image
// reads by the last arriver) instead of a second launch and a round trip of the slabs.
static inline int pick_ksplit_p8q(int64_t tiles, int64_t K, int64_t M, int64_t N, size_t ws_bytes, bool fix = false)
{
    if (N % 4 != 0) return 1;
    const int64_t nt = K / 128;
    const int forced = forced_ksplit();
    int64_t s = 1;
    if (forced > 0) {
        s = forced > nt ? nt : forced;
    } else {
        double best = 1e30;
        const int64_t smax = nt / 4 < 16 ? nt / 4 : 16;
        for (int64_t c = 1; c <= (smax < 1 ? 1 : smax); ++c) {
            const int64_t nblk = fix ? 8 * ((tiles + 7) / 8) * c : tiles * c;   // (the XCD-affine grid rounds every XCD's share up)
            const double blocks = (double)nblk, waves = (double)((nblk + 255) / 256), fill = blocks < 256.0 ? blocks / 256.0 : 1.0;
            double t = fix ? waves * (3.0 + (double)((nt + c - 1) / c) * (0.36 + 0.23 * (fill > 0.5 ? fill - 0.5 : 0.0)))   // (gemm_i8_p8q2's K-tile: flat up to half the chip)
                           : waves * (3.0 + (double)((nt + c - 1) / c) * (0.40 + 0.20 * fill));
            if (c > 1) t += fix ? P8Q_FIX_TAIL_US + P8Q_FIX_PER_SPLIT_US * (double)(c - 2) : 5.0 + (double)c * (double)M * (double)N * 4.0 / 3.0e6;
            if (t < best) { best = t; s = c; }
        }
    }
    if (fix) {
        while (s > 1 && p8q_fix_bytes(tiles, s) > ws_bytes) --s;
    } else {
        while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4 > ws_bytes) --s;
    }
    return s < 1 ? 1 : (int)s;
}

// K splits for the 128-row kernel (tiles = its block count at one split), from the same kind of fitted cost model as pick_ksplit_p8q (us): a block
// costs 3 + (K-tiles) x (0.45 + 0.33 x the fraction of the 256 CUs that hold a block), a split launch adds 5 + S x M x N x 4 B at 3 TB/s for the
// reduce pass.  OPT-13B fc2 at 256 rows (40 tiles, 160 K-tiles): S = 6 (39.5 us warm / 46.0 cold; the former "fill ~192 CUs" rule gave S = 4:
// 41.2 / 51.8); 384x4096x11008: S = 4 (31.4); 256x4096x11008: S = 6 (26.3); 2048x4096x4096 (128 tiles): S = 1.
static inline int pick_ksplit_p8h(int64_t tiles, int64_t K, int64_t M, int64_t N, size_t ws_bytes)
{
    if (N % 4 != 0) return 1;
    const int64_t nt = K / 128;
    const int forced = forced_ksplit();
    int64_t s = 1;
    if (forced > 0) {
        s = forced > nt ? nt : forced;
    } else {
        double best = 1e30;
        const int64_t smax = nt / 4 < 16 ? nt / 4 : 16;
        for (int64_t c = 1; c <= (smax < 1 ? 1 : smax); ++c) {
            const double blocks = (double)tiles * (double)c, waves = (double)((tiles * c + 255) / 256), fill = blocks < 256.0 ? blocks / 256.0 : 1.0;
            double t = waves * (3.0 + (double)((nt + c - 1) / c) * (0.45 + 0.33 * fill));
            if (c > 1) t += 5.0 + (double)c * (double)M * (double)N * 4.0 / 3.0e6;
            if (t < best) { best = t; s = c; }
        }
    }
    while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4 > ws_bytes) --s;
    return s < 1 ? 1 : (int)s;
}
// (K splits of this kernel reduced inside the launch were built on splitk_fix_reduce and measured: no gain at two splits, 3-7 us slower from three on -- 128 KiB images,
// 40-80 tiles, 4-6 splits: profiles/r5_p8h_splitk_in_launch_dropped.txt)

template <class Epi, int MT, int NT> int launch_skinny_mt(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, int mblocks, const Epi &epi, hipStream_t s)
{
    constexpr int64_t LDS_CU = 160 * 1024;
    const int64_t ntiles = (N + 16 * NT - 1) / (16 * NT);
    const int64_t nitems = ((ntiles + 7) / 8) * 8 * mblocks;  // tiles padded to groups of 8 so m-blocks of a tile share an XCD
    const int64_t perwave = SK_STAGES * (NT + MT) * 2048 + NT * MT * 1024;  // DMA ring + reduction slot
    // the most waves per block (K parallelism inside a work item) that still gives EVERY item a resident block
    int wpb = 8;
    while (wpb > 1 && (wpb * perwave > LDS_CU || 256 * (LDS_CU / (wpb * perwave)) < nitems)) wpb >>= 1;
    int64_t per_cu = LDS_CU / (wpb * perwave);
    if (per_cu > 16) per_cu = 16;
    if (per_cu * wpb > 32) per_cu = 32 / wpb;
    if (per_cu < 1) per_cu = 1;
    int64_t grid = 256 * per_cu;   // persistent beyond that: blocks walk the items grid-stride
    if (grid > nitems) grid = nitems;
    const size_t lds = (size_t)(wpb * perwave);
    auto kfn = gemm_i8_skinny<Epi, MT, NT>;
    hipError_t e = ensure_dynamic_lds((const void *)kfn, (int)lds);
    if (e != hipSuccess) {
        asq_set_error("skinny: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3((unsigned)(wpb * 64)), lds, s, x, w, M, N, K, wpb, mblocks, epi);
    return ASQ_OK;
}

// ---- second-generation weight stream (asq_gemm_wstream.h): grid size and workspace
struct WsPlan {
    int G = 0, KU = 0, T = 0, maxseg = 0, mt = 0;
    size_t bytes = 0;  // scratch behind the workspace header
};
static inline int ws_env(const char *name)  // development overrides, read once per process
{
    const char *e = getenv(name);
    return e ? atoi(e) : -1;
}
static inline WsPlan plan_wstream(int64_t M, int64_t N, int64_t K)
{
    WsPlan p;
    if (M < 1 || M > 128 || K % 128 != 0 || K < 128) return p;
    // Used where its in-launch reduction (~4 us) costs less than gemm_i8_skinny's LDS starvation: long-K weights at >= 16 rows.  Measured, cold
    // weights, us (tools/ubench/wstream_probe, profiles/r3_skinny_experiments.md): 5120x20480 (OPT-13B fc2) 16 rows 26.5 -> 24.0, 32 rows 31.3 -> 25.9;
    // 4096x14336 (Mixtral w2) 64 rows 22.5 -> 20.0; everything else +3 ... +90 % (4096x4096 at 32 rows: 6.5 -> 12.0): the region is narrow on purpose.
    static const int impl = ws_env("ASQ_SK_IMPL");  // development A/B: 0 = never, 1 = wherever it can run
    const bool region = (K >= 4 * N && M >= 16 && N * K >= (64ll << 20)) || (K >= 3 * N && M > 32 && N * K >= (48ll << 20));
    if (impl == 0 || !(region || impl == 1)) return p;
    const int64_t NG = (N + WS_CB - 1) / WS_CB, KU = K / 128, T = NG * KU;
    if (NG > WS_MAX_GROUPS || T >= (1ll << 22)) return p;  // (T * G < 2^31 with G <= 512)
    static const int forced_g = ws_env("ASQ_WS_GRID");
    int64_t G = forced_g > 0 ? forced_g : 256;
    if (G > 512) G = 512;
    if (forced_g <= 0) {
        // at least 4 units (64 KB of W) per block, and at most ~16 contributors per group for the last arriver to sum
        if (G > T / 4) G = T / 4 < 1 ? 1 : T / 4;
        if (G > 15 * NG) G = 15 * NG;
    }
    if (G > T) G = T;
    p.G = (int)G;
    p.KU = (int)KU;
    p.T = (int)T;
    const int64_t nun_max = (T + G - 1) / G;
    p.maxseg = (int)((nun_max + KU - 1) / KU + 1);
    p.mt = M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : 8;
    p.bytes = (size_t)G * (size_t)p.maxseg * (size_t)(p.mt * 8192);
    return p;
}

template <class Epi, int MT, bool WNT>
int launch_wstream_mt(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, const WsPlan &p, char *ws, const Epi &epi, hipStream_t s)
{
    auto kfn = gemm_i8_wstream<Epi, MT, WNT>;
    hipError_t e = ensure_dynamic_lds((const void *)kfn, WsCfg<MT>::LDS);
    if (e != hipSuccess) {
        asq_set_error("wstream: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.G), dim3(512), WsCfg<MT>::LDS, s, x, w, M, N, K, p.KU, p.T, p.maxseg, ws, epi);
    return ASQ_OK;
}

// ws_hdr = the caller's workspace (header first), usable scratch behind it = ws_bytes - WS_HEADER_BYTES
template <class Epi> int launch_skinny(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, const Epi &epi, hipStream_t s, void *ws_hdr = nullptr, size_t ws_bytes = 0)
{
    const int mblocks = (int)((M + 63) / 64);                       // m-blocks of <= 64 rows, balanced
    const int mt = (int)(((M + mblocks - 1) / mblocks + 15) / 16);  // 16-row tiles per m-block
    static int nt_forced = -1;
    if (nt_forced < 0) { const char *e = getenv("ASQ_SK_NT"); nt_forced = e ? atoi(e) : 0; }
    const int64_t items2 = ((N + 31) / 32) * mblocks;
    const bool wide = nt_forced ? nt_forced == 2 : (items2 >= 448 || (K >= 16384 && items2 >= 128));
    if constexpr (Epi::Mma::kIsInt) if (ws_hdr != nullptr && ws_bytes > (size_t)WS_HEADER_BYTES && M * K < (1ll << 32)) {
        const WsPlan p = plan_wstream(M, N, K);  // the stream-K kernel inside its measured region (plan_wstream)
        if (p.G > 0 && p.bytes <= ws_bytes - WS_HEADER_BYTES) {
            // streaming (nt) cache policy for the weight lines of the big streams: 5120x20480 (105 MB) 27.2 -> 25.9 us at 32 rows, 25.8 -> 24.0 at 16;
            // 4096x14336 (59 MB) 20.0 -> 21.9 at 64 rows, so only above 80 MB
            static const int nt_env = ws_env("ASQ_WS_NT");
            const int nt = nt_env >= 0 ? nt_env : (N * K >= (80ll << 20) ? 1 : 0);
#define ASQ_WS(MT_) (nt == 1 ? launch_wstream_mt<Epi, MT_, true>(x, w, M, N, K, p, (char *)ws_hdr, epi, s) : launch_wstream_mt<Epi, MT_, false>(x, w, M, N, K, p, (char *)ws_hdr, epi, s))
            switch (p.mt) {
            case 1: return ASQ_WS(1);
            case 2: return ASQ_WS(2);
            case 4: return ASQ_WS(4);
            default: return ASQ_WS(8);
            }
#undef ASQ_WS
        }
    }
    // 32 channels per item halve the X re-reads from L2.  Measured (tools/ubench/skinny_probe, ASQ_SK_NT=1|2, M = 32):
    // 14336x4096 16.8 -> 15.5 us, 20480x5120 29.5 -> 26.3, 5120x20480 40.9 -> 31.7; but 8192x8192 18.0 -> 20.6 and
    // 4096x11008 13.1 -> 15.5 (too few items), 11008x4096 unchanged: only with plenty of items, or a long K
#define ASQ_SK(MT_) (wide ? launch_skinny_mt<Epi, MT_, 2>(x, w, M, N, K, mblocks, epi, s) : launch_skinny_mt<Epi, MT_, 1>(x, w, M, N, K, mblocks, epi, s))
    switch (mt) {
    case 1: return ASQ_SK(1);
    case 2: return ASQ_SK(2);
    case 3: return ASQ_SK(3);
    default: return ASQ_SK(4);
    }
#undef ASQ_SK
}

// Tail peel (hybrid of data-parallel tiles and a finer-grained remainder).  A tile grid that is a few tiles over a multiple of 256 pays a whole
// extra wave for them (1536 x 11008: 258 tiles -> 90 us against 51 us for the 172 tiles of 1024 rows).  When the last wave would be < 3/8 full
// and <= 48 tiles cover it, the last `c` tile columns (all of the remainder and a little more) become their own launch of 128 x 128 tiles (p8q:
// four times as many blocks, with its usual K split when the caller's workspace allows one) and the main launch is left with <= 256 * waves tiles.
struct TailPeel {
    int64_t n_main = 0;  // columns [0, n_main) stay with the main launch; 0 = no peel
    size_t ws_bytes = 0; // workspace that lets the remainder split K (optional)
    bool rem_p8h = false; // the remainder runs on 128 x 256 tiles (gemm_i8_p8h) instead of 128 x 128 (gemm_i8_p8q)
};
static inline TailPeel plan_tail_peel(GemmKernel kern, int64_t M, int64_t N, int64_t K)
{
    TailPeel p;
    // (the remainder launch costs ~13 us at K = 4096; the extra wave it replaces ~20 us for the 128-row kernel, 35-45 us for the 256-row ones)
    if (kern != KERN_P8 && kern != KERN_P4 && kern != KERN_P8H && kern != KERN_P16 && kern != KERN_P4X16) return p;
    static const bool disabled = getenv("ASQ_NO_TAIL") != nullptr;  // development / A-B aid
    if (disabled || forced_kernel() >= 0 || forced_ksplit() > 0 || N % 4 != 0 || K < 4096) return p;  // (a short K loop makes the extra wave cheap)
    const int64_t rows = kern == KERN_P8H ? 128 : 256;
    const int64_t tm = (M + rows - 1) / rows, tn = (N + 255) / 256, tiles = tm * tn;
    const int64_t full = tiles / 256, r = tiles % 256;
    // Round 4 (profiles/r4_tail_sweep.txt, 2048 rows x (32 + c) tile columns, K = 4096, remainder = 8 c tiles of 256 x 256): as 128 x 128 tiles (p8q, four times
    // the blocks) the remainder gains 21 ... 14 % up to 64 tiles and nothing beyond; as 128 x 256 tiles (p8h, ONE round of twice the blocks) 11 ... 7 % from 88 to
    // 128 tiles; from 160 tiles on both lose to the plain second round.
    const int64_t r_max = rows == 256 ? 128 : 96;
    if (full < 1 || r == 0 || r > r_max) return p;
    const int64_t c = (r + tm - 1) / tm;
    const int64_t rem256 = ((M + 255) / 256) * c;   // the remainder in 256 x 256 tiles' worth
    if (c >= tn || rem256 > (rows == 256 ? 128 : 48)) return p;
    p.rem_p8h = rem256 > 64;
    p.n_main = (tn - c) * 256;
    const int64_t n_rem = N - p.n_main;
    const int ks = p.rem_p8h ? pick_ksplit_p8h(((M + 127) / 128) * ((n_rem + 255) / 256), K, M, n_rem, (size_t)-1)
                             : pick_ksplit_p8q(((M + 127) / 128) * ((n_rem + 127) / 128), K, M, n_rem, (size_t)-1);
    p.ws_bytes = ks > 1 ? (size_t)ks * (size_t)M * (size_t)n_rem * 4 : 0;
    if (!p.rem_p8h) {   // a 128 x 128 remainder reduces its K splits inside its launch when the caller's workspace has a header (register images of whole tiles, their own split count)
        const int64_t tiles = ((M + 127) / 128) * ((n_rem + 127) / 128);
        const int kf = tiles > WS_MAX_GROUPS ? 1 : pick_ksplit_p8q(tiles, K, M, n_rem, (size_t)-1, true);
        const size_t fb = kf > 1 ? p8q_fix_bytes(tiles, kf) : 0;
        p.ws_bytes = fb > p.ws_bytes ? fb : p.ws_bytes;
    }
    return p;
}

// ASQ_MMA=32: the tiled int8 kernels other than p16 / p8 / p4 themselves (grouped launches, p8h, p8q) keep v_mfma_i32_32x32x32_i8 instead of the 16 x 16 x 64 form
// (development / A-B aid; read once)
static inline bool mma32_forced()
{
    static const bool v = [] { const char *e = getenv("ASQ_MMA"); return e && atoi(e) == 32; }();
    return v;
}

// ASQ_GROUPED_SPLIT=0: grouped launches never split the K loop of their tail tiles (A/B switch; the default is on when a workspace is passed)
static inline bool grouped_tail_split_enabled()
{
    static const bool on = [] {
        const char *e = getenv("ASQ_GROUPED_SPLIT");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <class Epi, class = void> struct IsGroupable : std::false_type {};
template <class Epi> struct IsGroupable<Epi, std::enable_if_t<Epi::kGroupable>> : std::true_type {};
template <class Epi, class = void> struct HasColView : std::false_type {};
template <class Epi> struct HasColView<Epi, std::enable_if_t<Epi::kColView>> : std::true_type {};

// A caller's workspace is [ header WS_HEADER_BYTES (asq_workspace_init: magic + tickets of the weight-streaming kernel) | scratch ];
// `ws` below is the scratch part (split-K slabs), `ws_hdr` the whole thing (null when the caller's buffer is too small to hold a header).
template <class Epi>
int launch_gemm_impl(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, Epi epi, hipStream_t s, const char *what, void *ws_hdr, void *ws,
                     size_t ws_bytes, const int *goffs = nullptr, int ngroups = 0, int peel_role = 0 /* 0 top level, 1 main part, 2 / 3 remainder (128 x 128 / 128 x 256 tiles) */,
                     OffsetArgs off = OffsetArgs{})
{
    if (M == 0 || N == 0) return ASQ_OK;
    if constexpr (!IsGroupable<Epi>::value) {
        ASQ_REQUIRE(goffs == nullptr, ASQ_ERR_DIM, "%s: this epilogue has no grouped form", what);
    } else if (goffs != nullptr) {  // grouped: tiled kernel only; grid = host-side upper bound on the number of tiles
        const bool ok = (((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0) && K % 128 == 0 && K >= 128 && K <= (1 << 24);
        ASQ_REQUIRE(ok, ASQ_ERR_DIM, "%s: grouped launch needs K %% 128 == 0 and 16-B aligned operands", what);
        const int64_t tn = (N + 255) / 256;
        int64_t tiles = (M / 256 + ngroups) * tn;   // upper bound on sum ceil(m_g / 256) * tn
        char *gws = nullptr;
        if (ngroups <= P8_GROUPED_SCAN_MAX) {       // balanced scheduler: blocks b = 8 * slot + xcd, up to one extra round of K pieces per XCD
            tiles = 8 * ((tiles + 7) / 8 + 2 + P8_CUS_PER_XCD);   // (+2: full and half tiles are dealt to the XCDs separately)
            if constexpr (Epi::Mma::kIsInt)
                if (ws_hdr != nullptr && ws_bytes >= P8_GROUPED_WS_BYTES && grouped_tail_split_enabled()) gws = (char *)ws_hdr;
        }
        ASQ_REQUIRE(tiles < (1ll << 24), ASQ_ERR_DIM, "%s: too many tiles", what);
        // int8 groups run on v_mfma_i32_16x16x64_i8 (the L16 form of the kernel: asq_gemm_p16.h for why); ASQ_MMA=32 keeps the 32 x 32 x 32 form (A/B)
        auto kfn = gemm_i8_p8<Epi, 0, true>;
        if constexpr (Epi::Mma::kIsInt) {
            if (!mma32_forced()) kfn = gemm_i8_p8<Epi, 0, true, true>;
        }
        hipError_t e = ensure_dynamic_lds((const void *)kfn, P8_LDS_BYTES);
        if (e != hipSuccess) {
            asq_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
            return (int)e;
        }
        bool g_offs = false;
        if constexpr (Epi::Mma::kIsInt && Epi::kOutBytes == 2) g_offs = off.row != nullptr && !mma32_forced();
        ASQ_REQUIRE(off.row == nullptr || (g_offs && K <= OFFSET_MAX_K && N % 4 == 0), ASQ_ERR_DIM, "%s: offset operands need int8 groups, 2-byte outputs, K <= 65536, N %% 4 == 0", what);
        if (g_offs) {
            e = ensure_dynamic_lds((const void *)kfn, P16_LDS_BYTES);
            if (e != hipSuccess) {
                asq_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
                return (int)e;
            }
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)tiles), dim3(512), g_offs ? P16_LDS_BYTES : P8_LDS_BYTES, s, x, w, M, N, K, 0, (int)tn, 1, goffs, ngroups, gws, epi, g_offs ? off : OffsetArgs{});
        return asq_after_launch(s, what);
    }
    constexpr bool kInt = Epi::Mma::kIsInt;
    GemmKernel kern = peel_role == 2 ? KERN_P8Q : peel_role == 3 ? KERN_P8H : pick_kernel(x, w, M, N, K);   // (2 / 3: the column remainder of a tail peel)
    if (off.row != nullptr) {   // offset operands: gemm_i8_p16 only (the entry point has checked the shape: offsets_supported)
        ASQ_REQUIRE(kInt && (Epi::kOutBytes == 2 || Epi::kOutBytes == 4) && offsets_shape_ok(x, w, M, N, K), ASQ_ERR_DIM, "%s: offset operands need the 256 x 256 kernel (2- or 4-byte output, K %% 128 == 0, K <= 65536, N %% 4 == 0, aligned operands)", what);
        kern = KERN_P16;
    }
    if constexpr (kInt && HasColView<Epi>::value) {
        if (off.row == nullptr && peel_role == 0 && (N * Epi::kOutBytes) % 16 == 0) {
            const TailPeel tp = plan_tail_peel(kern, M, N, K);
            if (tp.n_main > 0) {
                const int rc = launch_gemm_impl(x, w, M, tp.n_main, K, epi, s, what, nullptr, nullptr, 0, nullptr, 0, 1);
                if (rc) return rc;
                return launch_gemm_impl(x, w + tp.n_main * K, M, N - tp.n_main, K, epi.col_view(tp.n_main), s, what, ws_hdr, ws, ws_bytes, nullptr, 0, tp.rem_p8h ? 3 : 2);
            }
        }
    }
    // ---- the tiled kernels: one launcher (dynamic-LDS attribute once per kernel, launch) and one split-K tail (exact int32 slabs, then reduce + the caller's epilogue)
    auto launch_tiled = [&](auto kfn, int lds_attr, int lds, int64_t grid, int block, auto... args) -> int {
        ASQ_REQUIRE(grid < (1ll << 24), ASQ_ERR_DIM, "%s: too many tiles", what);
        const hipError_t e = ensure_dynamic_lds((const void *)kfn, lds_attr);
        if (e != hipSuccess) {
            asq_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
            return (int)e;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3((unsigned)block), lds, s, x, w, M, N, K, args...);
        return ASQ_OK;
    };
    auto reduce_slabs = [&](int ksplit) {
        if constexpr (kInt) {
            int64_t blocks = (M * (N / 4) + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL((splitk_reduce<Epi>), dim3((unsigned)blocks), dim3(256), 0, s, (const int32_t *)ws, ksplit, M, N, epi);
        }
    };
    const bool ws_ok = kInt && ws != nullptr && (((uintptr_t)ws) & 15) == 0;
    const int64_t tm256 = (M + 255) / 256, tm128 = (M + 127) / 128, tn256 = (N + 255) / 256, tn128 = (N + 127) / 128;
    [[maybe_unused]] const EpiI32 slab{(int32_t *)ws, N, true};
    const int *const no_groups = nullptr;
    char *const no_gws = nullptr;
    const uint8_t *const no_mx = nullptr;
    int rc = ASQ_OK;

    constexpr bool kP4 = kInt && Epi::kOutBytes == 2;  // the 4-wave kernels: int8 operands, 2-byte outputs (their row epilogue; the int32 / int8-out
                                                       // epilogues next to 256 accumulator registers would spill)
    if (kern == KERN_P4 && !kP4) kern = KERN_P8;
    // four waves x 128 x 128 on the 16 x 16 x 64 instruction (asq_gemm_p4x16.h): scalar / per-token scale epilogues with 2-byte outputs; otherwise p16
    constexpr bool kP4X = kP4 && !Epi::kHasCol && !Epi::kHasBias;
    if (kern == KERN_P4X16 && !kP4X) kern = KERN_P16;
    // the 256 x 256 kernel on v_mfma_i32_16x16x64_i8 (asq_gemm_p16.h): plain launches with 2- and (round 4, epilogue_wave_rows4) 4-byte outputs; what it does not
    // carry -- int8 outputs, K splits -- runs on p8, in its L16 mode (the same instruction) unless the 32 x 32 x 32 form was asked for
    constexpr bool kP16 = kInt && (Epi::kOutBytes == 2 || Epi::kOutBytes == 4);
    bool p8_l16 = false;
    if (kern == KERN_P16 && !kP16) {
        kern = KERN_P8;
        p8_l16 = kInt && !mma32_forced();
    }
    ASQ_REQUIRE(off.row == nullptr || kern == KERN_P16, ASQ_ERR_DIM, "%s: offset operands: no kernel", what);

    if (kern == KERN_P4) {
        if constexpr (kP4) rc = launch_tiled(gemm_i8_p4<Epi>, P4_LDS_BYTES, P4_LDS_BYTES, tm256 * tn256, 256, (int)tm256, (int)tn256, epi);
    } else if (kern == KERN_P4X16) {
        if constexpr (kP4X) rc = launch_tiled(gemm_i8_p4x16<Epi>, P4_LDS_BYTES, P4_LDS_BYTES, tm256 * tn256, 256, (int)tm256, (int)tn256, epi);
    } else if (kern == KERN_P16) {
        bool done = false;
        if constexpr (kP16 && Epi::kOutBytes == 2) {
            // multi-round launches without edge tiles: the persistent form (asq_gemm_p16p.h) -- the next tile's first K-tile and epilogue operands arrive under
            // this tile's last K-tile and epilogue.  ASQ_P16_PERSIST=0 keeps gemm_i8_p16 (A/B); =2 also takes single-round launches (development).
            static const int persist = [] { const char *e = getenv("ASQ_P16_PERSIST"); return e ? atoi(e) : 1; }();
            const int64_t T = tm256 * tn256;
            if (persist && M % 256 == 0 && N % 256 == 0 && K % 256 == 0 && (T > 8 * P8_CUS_PER_XCD || persist == 2) && ((((uintptr_t)epi.out) & 15) == 0) && (epi.N * 2) % 16 == 0 &&
                epi.N * 2 < (int64_t(1) << 24)) {
                const int64_t grid = T < 8 * P8_CUS_PER_XCD ? T : 8 * P8_CUS_PER_XCD;
                rc = launch_tiled(gemm_i8_p16p<Epi>, P16P_LDS_BYTES, P16P_LDS_BYTES, grid, 512, (int)tm256, (int)tn256, epi, off);
                done = true;
            }
        }
#if ASQ_P16_TAIL
        if constexpr (kP16 && Epi::kOutBytes == 2) {   // (builds with -DASQ_P16_TAIL=1 only: asq_gemm_p16t.h, measured and NOT shipped, profiles/r6_tail_overlap_ab.txt)
            static const int tail = [] { const char *e = getenv("ASQ_P16_TAIL"); return e ? atoi(e) : 1; }();
            if (!done && tail && M % 256 == 0 && N % 256 == 0 && K % 256 == 0 && K >= 512 && ((((uintptr_t)epi.out) & 15) == 0) && (epi.N * 2) % 16 == 0 && epi.N * 2 < (int64_t(1) << 24)) {
                rc = launch_tiled(gemm_i8_p16t<Epi>, P16_LDS_BYTES, off.row ? P16_LDS_BYTES : P8_LDS_BYTES, tm256 * tn256, 512, (int)tm256, (int)tn256, epi, off);
                done = true;
            }
        }
#endif
        if constexpr (kP16) if (!done) rc = launch_tiled(gemm_i8_p16<Epi>, P16_LDS_BYTES, off.row ? P16_LDS_BYTES : P8_LDS_BYTES, tm256 * tn256, 512, (int)tm256, (int)tn256, epi, off);
    } else if (kern == KERN_P8) {
        const int ksplit = ws_ok ? pick_ksplit(tm256 * tn256, K, M, N, ws_bytes) : 1;
        if (ksplit > 1) {
            if constexpr (kInt) {
                rc = p8_l16 ? launch_tiled(gemm_i8_p8<EpiI32, 0, false, true>, P8_LDS_BYTES, P8_LDS_BYTES, tm256 * tn256 * ksplit, 512, (int)tm256, (int)tn256, ksplit, no_groups, 0, no_gws, slab, OffsetArgs{})
                            : launch_tiled(gemm_i8_p8<EpiI32>, P8_LDS_BYTES, P8_LDS_BYTES, tm256 * tn256 * ksplit, 512, (int)tm256, (int)tn256, ksplit, no_groups, 0, no_gws, slab, OffsetArgs{});
                if (rc == ASQ_OK) reduce_slabs(ksplit);
            }
        } else {
            bool done = false;
            if constexpr (kInt && !kP16) {   // (int8 outputs at p16's sizes)
                if (p8_l16) {
                    rc = launch_tiled(gemm_i8_p8<Epi, 0, false, true>, P8_LDS_BYTES, P8_LDS_BYTES, tm256 * tn256, 512, (int)tm256, (int)tn256, 1, no_groups, 0, no_gws, epi, OffsetArgs{});
                    done = true;
                }
            }
            if (!done) rc = launch_tiled(gemm_i8_p8<Epi>, P8_LDS_BYTES, P8_LDS_BYTES, tm256 * tn256, 512, (int)tm256, (int)tn256, 1, no_groups, 0, no_gws, epi, OffsetArgs{});
        }
    } else if (kern == KERN_P8H) {
        const int ksplit = ws_ok ? pick_ksplit_p8h(tm128 * tn256, K, M, N, ws_bytes) : 1;
        if (ksplit > 1) {
            if constexpr (kInt) {
                rc = mma32_forced() ? launch_tiled(gemm_i8_p8h<EpiI32>, P8H_LDS_BYTES, P8H_LDS_BYTES, tm128 * tn256 * ksplit, 512, (int)tm128, (int)tn256, ksplit, slab)
                                    : launch_tiled(gemm_i8_p8h<EpiI32, false, true>, P8H_LDS_BYTES, P8H_LDS_BYTES, tm128 * tn256 * ksplit, 512, (int)tm128, (int)tn256, ksplit, slab);
                if (rc == ASQ_OK) reduce_slabs(ksplit);
            }
        } else {
            bool done = false;
            if constexpr (kInt) {
                if (!mma32_forced()) {
                    rc = launch_tiled(gemm_i8_p8h<Epi, false, true>, P8H_LDS_BYTES, P8H_LDS_BYTES, tm128 * tn256, 512, (int)tm128, (int)tn256, 1, epi);
                    done = true;
                }
            }
            if (!done) rc = launch_tiled(gemm_i8_p8h<Epi>, P8H_LDS_BYTES, P8H_LDS_BYTES, tm128 * tn256, 512, (int)tm128, (int)tn256, 1, epi);
        }
    } else if (kern == KERN_P8Q) {
        bool fix = false;
        if constexpr (kInt && (Epi::kOutBytes == 2 || Epi::kOutBytes == 4)) fix = ws_ok && ws_hdr != nullptr && splitk_fix_mode() != 0 && p8q2_enabled() && !mma32_forced() && tm128 * tn128 <= WS_MAX_GROUPS;
        int ksplit = ws_ok ? pick_ksplit_p8q(tm128 * tn128, K, M, N, ws_bytes, fix) : 1;
        if (fix && (ksplit > 16 || p8q_fix_bytes(tm128 * tn128, ksplit) >= ((size_t)1 << 31))) {   // (32-bit image offsets; only a forced ASQ_KSPLIT gets here)
            fix = false;
            ksplit = pick_ksplit_p8q(tm128 * tn128, K, M, N, ws_bytes, false);   // (the slab form sizes its scratch differently)
        }
        if (ksplit > 1 && fix) {
            if constexpr (kInt && (Epi::kOutBytes == 2 || Epi::kOutBytes == 4))
                rc = launch_tiled(gemm_i8_p8q2<Epi, true>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, 8 * ((tm128 * tn128 + 7) / 8) * ksplit, 512, (int)tm128, (int)tn128, ksplit, epi, (char *)ws_hdr);
        } else if (ksplit > 1) {
            if constexpr (kInt) {
                rc = mma32_forced() ? launch_tiled(gemm_i8_p8q<EpiI32>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128 * ksplit, 512, (int)tm128, (int)tn128, ksplit, slab, no_mx, no_mx)
                     : p8q2_enabled() ? launch_tiled(gemm_i8_p8q2<EpiI32>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128 * ksplit, 512, (int)tm128, (int)tn128, ksplit, slab, (char *)nullptr)
                                      : launch_tiled(gemm_i8_p8q<EpiI32, false, true>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128 * ksplit, 512, (int)tm128, (int)tn128, ksplit, slab, no_mx, no_mx);
                if (rc == ASQ_OK) reduce_slabs(ksplit);
            }
        } else {
            bool done = false;
            if constexpr (kInt) {
                if (!mma32_forced()) {
                    rc = p8q2_enabled() ? launch_tiled(gemm_i8_p8q2<Epi>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128, 512, (int)tm128, (int)tn128, 1, epi, (char *)nullptr)
                                        : launch_tiled(gemm_i8_p8q<Epi, false, true>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128, 512, (int)tm128, (int)tn128, 1, epi, no_mx, no_mx);
                    done = true;
                }
            }
            if (!done) rc = launch_tiled(gemm_i8_p8q<Epi>, P8Q_LDS_BYTES, P8Q_LDS_BYTES, tm128 * tn128, 512, (int)tm128, (int)tn128, 1, epi, no_mx, no_mx);
        }
    } else if (kern == KERN_SKINNY) {
        const int rc = launch_skinny(x, w, M, N, K, epi, s, ws_hdr, ws_hdr ? ws_bytes + WS_HEADER_BYTES : 0);
        if (rc) return rc;
    } else {
        const bool fast = (K % 16 == 0) && (((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0);
        dim3 grid((unsigned)((N + GEN_T - 1) / GEN_T), (unsigned)((M + GEN_T - 1) / GEN_T));
        ASQ_REQUIRE(grid.y < 65536, ASQ_ERR_DIM, "%s: M too large for the generic kernel (K %% 128 != 0 or unaligned operands)", what);
        hipLaunchKernelGGL((gemm_i8_generic<Epi>), grid, dim3(256), 0, s, x, w, M, N, K, fast, epi);
    }
    if (rc != ASQ_OK) return rc;
    return asq_after_launch(s, what);
}

template <class Epi>
int launch_gemm(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, Epi epi, hipStream_t s, const char *what, void *ws = nullptr,
                size_t ws_bytes = 0, const int *goffs = nullptr, int ngroups = 0, OffsetArgs off = OffsetArgs{})
{
    const bool has = ws != nullptr && ws_bytes >= (size_t)WS_HEADER_BYTES && (((uintptr_t)ws) & 15) == 0;
    if (has && asq_debug_sync()) {   // ASQ_DEBUG_SYNC=1: an uninitialised workspace is an error code here instead of a device trap in the kernels that use its tickets
        unsigned long long magic = 0;
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpy(&magic, ws, sizeof(magic), hipMemcpyDeviceToHost);
        ASQ_REQUIRE(e == hipSuccess && magic == WS_MAGIC, ASQ_ERR_WORKSPACE, "%s: workspace header not initialised (asq_workspace_init)", what);
    }
    return launch_gemm_impl(x, w, M, N, K, epi, s, what, has ? ws : nullptr, has ? (char *)ws + WS_HEADER_BYTES : nullptr, has ? ws_bytes - WS_HEADER_BYTES : 0, goffs,
                            ngroups, 0, off);
}

// per-dtype instantiation units (asq_gemm_inst_*.hip)
struct DequantArgs {
    const int8_t *xq, *w;
    void *out;
    int64_t M, N, K;
    float s_scalar;
    const float *s_row, *s_col, *bias;
    int order;
    bool vec_ok;
    void *ws;
    size_t ws_bytes;
    const float *s_group = nullptr;  // grouped launch (asq_linear_w8a8_grouped)
    const int *goffs = nullptr;
    int ngroups = 0;
    OffsetArgs off;                  // offset operands (asq_linear_w8a8_off)
};
template <int DT> int launch_dequant(const DequantArgs &a, hipStream_t s);
struct DequantQArgs {
    DequantArgs d;  // d.out = int8 [M,N]
    int act, qmode;
    float quant_scale;
};
template <int DT> int launch_dequant_q(const DequantQArgs &a, hipStream_t s);
template <int DT> static inline int launch_dequant_q_impl(const DequantQArgs &q, hipStream_t s)
{
    const DequantArgs &a = q.d;
    return launch_gemm(a.xq, a.w, a.M, a.N, a.K,
                       EpiDequantQ<DT>{(int8_t *)a.out, a.N, a.s_row, a.s_col, a.bias, a.s_scalar, q.quant_scale, a.order, q.act, q.qmode, a.vec_ok}, s,
                       "asq_linear_w8a8_q8", a.ws, a.ws_bytes);
}

template <int DT, bool R, bool C, bool B> static inline int launch_dequant_one(const DequantArgs &a, hipStream_t s)
{
    return launch_gemm(a.xq, a.w, a.M, a.N, a.K, EpiDequant<DT, R, C, B>{a.out, a.N, a.s_row, a.s_col, a.bias, a.s_group, a.s_scalar, a.order, a.vec_ok},
                       s, "asq_linear_w8a8", a.ws, a.ws_bytes, a.goffs, a.ngroups, a.off);
}

// The 8 epilogue variants of one output dtype are compiled in TWO translation units (with / without per-token row scales: asq_gemm_inst_<dt>.hip,
// asq_gemm_inst_<dt>_row.hip) -- each instantiates every tiled kernel for its 4 variants, and the longest TU sets the wall time of a parallel build.
template <int DT, bool HAS_ROW> int launch_dequant_half(const DequantArgs &a, hipStream_t s);
template <> int launch_dequant_half<ASQ_F32, false>(const DequantArgs &a, hipStream_t s);   // (declared before their first use: the probes under tools/ubench compile all of the TUs as one)
template <> int launch_dequant_half<ASQ_F32, true>(const DequantArgs &a, hipStream_t s);
template <> int launch_dequant_half<ASQ_F16, false>(const DequantArgs &a, hipStream_t s);
template <> int launch_dequant_half<ASQ_F16, true>(const DequantArgs &a, hipStream_t s);
template <> int launch_dequant_half<ASQ_BF16, false>(const DequantArgs &a, hipStream_t s);
template <> int launch_dequant_half<ASQ_BF16, true>(const DequantArgs &a, hipStream_t s);
template <int DT, bool HAS_ROW> static inline int launch_dequant_half_impl(const DequantArgs &a, hipStream_t s)
{
    const int key = (a.s_col ? 2 : 0) | (a.bias ? 1 : 0);
    switch (key) {
    case 0: return launch_dequant_one<DT, HAS_ROW, false, false>(a, s);
    case 1: return launch_dequant_one<DT, HAS_ROW, false, true>(a, s);
    case 2: return launch_dequant_one<DT, HAS_ROW, true, false>(a, s);
    default: return launch_dequant_one<DT, HAS_ROW, true, true>(a, s);
    }
}
template <int DT> static inline int launch_dequant_impl(const DequantArgs &a, hipStream_t s)
{
    return a.s_row ? launch_dequant_half<DT, true>(a, s) : launch_dequant_half<DT, false>(a, s);
}

}  // namespace asq
