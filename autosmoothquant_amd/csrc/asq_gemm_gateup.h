// gate || up as ONE GEMM whose epilogue writes SiLU(gate) * up: the two projections of a gated MLP (reference models/llama.py:206-211,
// HF LlamaMLP: down_proj(act_fn(gate_proj(x)) * up_proj(x)); both W8A8BFP32OFP32Linear over the SAME quantised input) read one activation and their outputs
// meet elementwise, so the [M, F] gate and up tensors need never reach HBM: today the two GEMMs write 2 x M x F x 2 bytes and asq_silu_mul_quantize reads them
// back.  Here the weights are ROW-INTERLEAVED in blocks of 16 channels -- rows 32 j .. 32 j + 15 of w_gu are gate channels 16 j .. 16 j + 15, rows 32 j + 16 ..
// 32 j + 31 the same channels of up -- so that in the 16 x 16 accumulator layout of v_mfma_i32_16x16x64_i8 (lane = token l & 15, channels 4 (l >> 4) .. + 3 of a
// 16-channel tile) a lane holds gate AND up of the same (token, 4 channels) in adjacent accumulator tiles, and the epilogue is elementwise per lane:
//     y_g = dt(s_g [* s_row[m]] * acc_g)      y_u = dt(s_u [* s_row[m]] * acc_u)           exactly what EpiDequant<DT> would have stored for the two linears
//     a   = dt(dt(silu(y_g)) * y_u)                                                         asq_silu_core.h: the arithmetic of asq_silu_mul_quantize
// out[M, F] in the activation dtype (the consumer's quantiser -- per-token absmax needs the whole row -- follows as its own pass over HALF the bytes).
// Bit-identical to asq_linear_w8a8 (gate), asq_linear_w8a8 (up), then the SiLU * up of asq_silu_mul_quantize with the same `fast` flag.
// Runs on the persistent 256 x 256 kernel (gemm_i8_p16p) only: multi-round prefill launches, M % 256 == 0, F % 128 == 0, K % 256 == 0.
#pragma once
#include "asq_silu_core.h"
#include "asq_quant_core.h"

namespace asq {

// Q8 (round 5, last session): the consumer is a per-tensor linear (reference models/llama.py:228-235 with fc2 per-tensor: x.div(input_scale).round().clamp -- linear.py:289-292),
// so the epilogue goes one step further and stores int8(clamp(round(dt(a / quant_scale)))) -- the per-tensor-div quantiser's arithmetic (asq_quant_core.h) on the same dt(a):
// out int8 [M, F], a QUARTER of the two fp tensors' bytes, and the consumer's quantiser launch disappears.
template <int DT, bool HAS_ROW, bool Q8 = false> struct EpiGateUp {
    using Mma = MmaI8;
    static constexpr bool kHasRow = HAS_ROW, kHasCol = false, kHasBias = false, kGateUp = true, kQ8 = Q8;
    static constexpr int kOutBytes = 2;
    static_assert(DT == ASQ_F16 || DT == ASQ_BF16, "2-byte activations");
    void *out;           // [M, F]
    int64_t N;           // F: the output's row stride in elements (the GEMM itself runs over 2 F interleaved weight rows)
    const float *s_row;  // [M]  (HAS_ROW: per-token activation scales)
    float s_gate, s_up, s_scalar;   // dequant scales of the two projections (s_scalar: unused, keeps the functor interface)
    int fast;
    float qs = 1.0f, qy = 0.0f;   // Q8: the consumer's quant_scale and RN(1 / quant_scale) (0: outside the fast division's range, divide)
    const float *sg_group = nullptr, *su_group = nullptr;   // grouped launch (asq_linear_w8a8_grouped_gate_up): per-group dequant scales [ngroups]
    static constexpr bool kGroupable = true;
    __device__ __forceinline__ EpiGateUp rebased(int grp, int, int64_t, int64_t) const
    {
        EpiGateUp e = *this;
        if (sg_group) {
            e.s_gate = sg_group[grp];
            e.s_up = su_group[grp];
        }
        return e;
    }
    // 4 outputs (channels c .. c + 3 of one token) from the lane's 4 gate and 4 up accumulators
    __device__ __forceinline__ v2u pack_gate_up(const v4i &g, const v4i &u, float sr) const
    {
        const float dg = HAS_ROW ? __fmul_rn(s_gate, sr) : s_gate, du = HAS_ROW ? __fmul_rn(s_up, sr) : s_up;
        float yg[4], yu[4];
        uint32_t uh[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vg = __fmul_rn(dg, (float)g[i]), vu = __fmul_rn(du, (float)u[i]);
            asm("" : "+v"(vg), "+v"(vu));   // (the conversion below is its own rounding: EpiDequant::pack's barrier)
            if constexpr (DT == ASQ_F16) {
                const _Float16 hg = (_Float16)vg, hu = (_Float16)vu;
                yg[i] = (float)hg;
                uh[i >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, hu) << (16 * (i & 1));
            } else {
                yg[i] = ElemT<DT>::round(vg);
                yu[i] = ElemT<DT>::round(vu);
            }
        }
        uint32_t o[2];
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const v2f sl = fast ? silu2<true>(yg[i], yg[i + 1]) : silu2<false>(yg[i], yg[i + 1]);
            if constexpr (DT == ASQ_F16) {
                o[i >> 1] = silu_times_up_h(sl, uh[i >> 1]);
            } else {
                const v2f a = silu_times_up<DT>(sl, yu[i], yu[i + 1]);
                o[i >> 1] = (uint32_t)ElemT<DT>::store(a[0]) | ((uint32_t)ElemT<DT>::store(a[1]) << 16);
            }
        }
        return (v2u){o[0], o[1]};
    }
    // Q8: the four dt values of pack_gate_up quantised like asq_quantize_act(.., ASQ_ACT_DIV, quant_scale) would quantise them
    __device__ __forceinline__ uint32_t quantise4(const v2u &a) const
    {
        int r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = ElemT<DT>::load((uint16_t)(a[i >> 1] >> (16 * (i & 1))));
            r[i] = qy != 0.0f ? QDivFast<DT>{qs, qy}(v) : QDiv<DT>{qs}(v);
        }
        return pack4(r[0], r[1], r[2], r[3]);
    }
};

// (IsGateUp<Epi>: asq_gemm_kernels.h, in front of the kernels)

// Epilogue of one wave tile (128 tokens x 64 interleaved channels = 32 output channels) of gemm_i8_p16p: interior tiles only, per-token scales from the tile's
// LDS operand area (EL = EpiTileLds<EpiGateUp>), direct 8-byte stores (a store instruction covers 16 rows x 32 bytes; the output is half a plain tile's bytes).
// The vmcnt(0) that retires the next tile's in-flight operand DMAs sits between the first conversions and the first store, as in epilogue_wave_rows<.., DRAIN>.
template <class EL, class Get> __device__ __forceinline__ void epilogue_gate_up(const EL &el, Get get, int64_t mw0, int64_t nw0, int lane)
{
    const auto &e = el.e;
    const int t = lane & 15, q = lane >> 4;
    constexpr int OB = std::remove_reference_t<decltype(e)>::kQ8 ? 1 : 2;   // bytes per output element
    const unsigned ldb = __builtin_amdgcn_readfirstlane((unsigned)(e.N * OB));   // output row pitch in bytes (128 * ldb < 2^31: the launcher checks)
    const uint64_t tile = (uint64_t)(uintptr_t)uniform_ptr((const int8_t *)e.out + (mw0 * e.N + (nw0 >> 1)) * OB);
    const unsigned voff = (unsigned)t * ldb + (unsigned)q * (4 * OB);
    typedef __attribute__((address_space(1))) v2u *glb_v2u;
    typedef __attribute__((address_space(1))) uint32_t *glb_u32;
    float sr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sr[i] = el.row(mw0 + i * 16 + t);
    v2u first = e.pack_gate_up(get(0, 0), get(1, 0), sr[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int im16 = 0; im16 < 8; ++im16)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const v2u o = (im16 == 0 && p == 0) ? first : e.pack_gate_up(get(2 * p, im16), get(2 * p + 1, im16), sr[im16]);
            if constexpr (OB == 2) *(glb_v2u)(uintptr_t)(tile + (uint64_t)((unsigned)(im16 * 16) * ldb + (unsigned)(p * 32)) + voff) = o;
            else *(glb_u32)(uintptr_t)(tile + (uint64_t)((unsigned)(im16 * 16) * ldb + (unsigned)(p * 16)) + voff) = e.quantise4(o);
        }
}

// The same epilogue for a wave tile of the GROUPED 256 x 256 kernel (gemm_i8_p8<Epi, 0, true, true>; Mixtral's w1 || w3): rows are bounded by the group's end / the
// half tile (Mw), per-token scales come from memory (the grouped launches of the models run per-tensor activations: HAS_ROW = false costs nothing), no operand area.
template <class Epi, class Get> __device__ __forceinline__ void epilogue_gate_up_rows(const Epi &e, Get get, int64_t mw0, int64_t nw0, int lane, int64_t Mw)
{
    const int t = lane & 15, q = lane >> 4;
    typedef __attribute__((address_space(1))) v2u *glb_v2u;
#pragma unroll
    for (int im16 = 0; im16 < 8; ++im16) {
        const int64_t m = mw0 + im16 * 16 + t;
        if (m >= Mw) continue;
        const float sr = Epi::kHasRow ? e.s_row[m] : 1.0f;
        char *const row = (char *)e.out + (m * e.N + (nw0 >> 1)) * 2 + q * 8;
#pragma unroll
        for (int p = 0; p < 2; ++p) *(glb_v2u)(uintptr_t)(row + p * 32) = e.pack_gate_up(get(2 * p, im16), get(2 * p + 1, im16), sr);
    }
}

// ---- fp8 (round 6): the same fusion for FP8LinearDynamic experts (reference models/mixtral.py:99-101 on FP8LinearDynamic modules, layers/nn/linear.py:413-427,
// easy_fp8_gemm :336-369): w1 || w3 of every expert ROW-INTERLEAVED in blocks of 32 channels, so that in the 32 x 32 accumulator layout of
// v_mfma_scale_f32_32x32x64_f8f6f4 (lane = token l & 31, channels 8 g + 4 (l >> 5) .. + 3 of a 32-channel tile) a wave's two n-halves ARE gate and up of the same 32
// output channels and a lane holds both for its (token, 4 channels):
//     y_g = dt(acc_g * (a_scale[m] * w_scale_gate))      y_u = dt(acc_u * (a_scale[m] * w_scale_up))      exactly EpiFp8<DT>'s values for the two linears
//     a   = dt(dt(silu(y_g)) * y_u)                                                                         asq_silu_core.h
// Bit-identical to asq_linear_fp8_grouped (w1), (w3) and the SiLU * up of asq_silu_mul_quantize_fp8 with the same flag.  Runs on the grouped 256 x 256 kernel
// (gemm_i8_p8<Epi, 0, true, false>); per-token activation scales.
template <int DT> struct EpiGateUpFp8 {
    using Mma = MmaFp8;
    static constexpr bool kHasRow = true, kHasCol = false, kHasBias = false, kGateUp = true, kQ8 = false, kGroupable = true;
    static constexpr int kOutBytes = 2;
    static_assert(DT == ASQ_F16 || DT == ASQ_BF16, "2-byte activations");
    void *out;               // [M, F]
    int64_t N;               // F
    const float *a_scale;    // [M] per-token activation scales
    const float *sg_group, *su_group;   // per-group weight scales of w1 / w3 [ngroups]
    float s_gate, s_up;
    int fast;
    __device__ __forceinline__ EpiGateUpFp8 rebased(int grp, int, int64_t, int64_t) const
    {
        EpiGateUpFp8 e = *this;
        e.s_gate = sg_group[grp];
        e.s_up = su_group[grp];
        return e;
    }
    __device__ __forceinline__ v2u pack_gate_up(const v4f &g, const v4f &u, float sr) const
    {
        const float dg = __fmul_rn(sr, s_gate), du = __fmul_rn(sr, s_up);   // EpiFp8: a * (sr * sc)
        float yg[4], yu[4];
        uint32_t uh[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vg = __fmul_rn(g[i], dg), vu = __fmul_rn(u[i], du);
            asm("" : "+v"(vg), "+v"(vu));   // (the conversion below is its own rounding)
            if constexpr (DT == ASQ_F16) {
                const _Float16 hg = (_Float16)vg, hu = (_Float16)vu;
                yg[i] = (float)hg;
                uh[i >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, hu) << (16 * (i & 1));
            } else {
                yg[i] = bf16_bits_to_f32(f32x2_to_bf16x2_bits(vg, vg) & 0xFFFFu);   // EpiFp8's own bf16 convert
                yu[i] = bf16_bits_to_f32(f32x2_to_bf16x2_bits(vu, vu) & 0xFFFFu);
            }
        }
        uint32_t o[2];
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const v2f sl = fast ? silu2<true>(yg[i], yg[i + 1]) : silu2<false>(yg[i], yg[i + 1]);
            if constexpr (DT == ASQ_F16) {
                o[i >> 1] = silu_times_up_h(sl, uh[i >> 1]);
            } else {
                const v2f a = silu_times_up<DT>(sl, yu[i], yu[i + 1]);
                o[i >> 1] = (uint32_t)ElemT<DT>::store(a[0]) | ((uint32_t)ElemT<DT>::store(a[1]) << 16);
            }
        }
        return (v2u){o[0], o[1]};
    }
};

// wave tile of the grouped 256 x 256 kernel in the 32 x 32 accumulator layout: get(in, im) = n-half in (0 = gate, 1 = up), 32-token tile im; rows bounded by the group's end
template <class Epi, class Get> __device__ __forceinline__ void epilogue_gate_up_rows32(const Epi &e, Get get, int64_t mw0, int64_t nw0, int lane, int64_t Mw)
{
    const int t = lane & 31, h = lane >> 5;
    typedef __attribute__((address_space(1))) v2u *glb_v2u;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int64_t m = mw0 + im * 32 + t;
        if (m >= Mw) continue;
        const float sr = e.a_scale[m];
        const auto &G = get(0, im);
        const auto &U = get(1, im);
        char *const row = (char *)e.out + (m * e.N + (nw0 >> 1)) * 2 + h * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(glb_v2u)(uintptr_t)(row + g * 16) = e.pack_gate_up((v4f){G[4 * g], G[4 * g + 1], G[4 * g + 2], G[4 * g + 3]}, (v4f){U[4 * g], U[4 * g + 1], U[4 * g + 2], U[4 * g + 3]}, sr);
    }
}

}  // namespace asq
