// Rotary position embedding (HF "rotate_half" / NeoX convention) as ONE pass over a projection's output -- caller-side glue of the W8A8 path ("the callers
// either side"): the reference's LLaMA / Mixtral / Baichuan wrappers run HF's apply_rotary_pos_emb between their q / k W8A8 linears and the attention product
// (models/llama.py:111 -> transformers' LlamaAttention.forward; models/mixtral.py:75; models/baichuan.py "ROPE" branch).  As torch ops that is four strided
// elementwise kernels per tensor (mul, addcmul, mul, addcmul over the two halves of every head: 8.5 % of the GPU time of BASELINE configs[2]'s forward,
// profiles/r5_cfg3_fused_kernel_stats.txt); here a thread reads 16 bytes of each half once and writes both.
//   x [B, S, H, D] with H * D contiguous and a row pitch (a projection's own output, or a slice of a fused q || k || v GEMM's), out [B, S, H, D] dense (the [B, H, S, D]
//   view attention wants is a transpose of it), cos / sin [S, D / 2] of the same dtype
//   out[.., :D/2] = dt(f32(dt(x1 * cos)) - x2 * sin)        out[.., D/2:] = dt(f32(dt(x2 * cos)) + x1 * sin)        positions 0 .. S - 1
// -- what torch computes for  addcmul(x1 * cos, x2, sin, value=-1)  /  addcmul(x2 * cos, x1, sin): fp16 bit for bit (the second rounding is the mixed-precision fma's
// single one, see fma_mix_f16x2); bf16 / fp32: the same operations at fp32 width (tests/test_hip_harness.py compares with the torch composition).
#include "asq_common.h"

namespace asq {

// d = f16(c +- a * b) with ONE rounding, per half of packed fp16 words: v_fma_mixlo_f16 / v_fma_mixhi_f16 evaluate the fma on the fp16 sources at fp32 width and round
// the exact result once to fp16 -- the instruction torch's own addcmul kernel compiles to on this platform (fptrunc(a + alpha * (b * c)) folded into it), so the two
// agree on every bit, also where the fp32 sum lands on an fp16 tie (19 of 1.3 M elements differ from a round-to-fp32-then-fp16 form).
template <bool NEG> __device__ __forceinline__ uint32_t fma_mix_f16x2(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d = 0;
    if constexpr (NEG) {
        asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b), "v"(c));
        asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b), "v"(c));
    } else {
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b), "v"(c));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b), "v"(c));
    }
    return d;
}

// ld: elements between two consecutive (b, s) rows of x (H * D for a projection's own output, the fused width for a slice of a q ‖ k ‖ v GEMM); out is dense [B, S, H, D]
template <int DT> __global__ void __launch_bounds__(256) rope_kernel(const void *__restrict__ xv, void *__restrict__ ov, const void *__restrict__ cosv,
                                                                      const void *__restrict__ sinv, int S, int H, int D, int64_t ld, int64_t nwork)
{
    constexpr int VEC = ElemT<DT>::VEC;
    constexpr int ES = 16 / VEC;   // bytes per element
    const int hv = D / 2 / VEC;    // 16-byte vectors per half head
    const int64_t half_bytes = (int64_t)(D / 2) * ES;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nwork; idx += (int64_t)gridDim.x * 256) {
        const int64_t hrow = idx / hv;                // (b * S + s) * H + h
        const int c = (int)(idx - hrow * hv);
        const int64_t bs = hrow / H;
        const int h = (int)(hrow - bs * H), s = (int)(bs % S);
        const char *xr = (const char *)xv + (bs * ld + (int64_t)h * D + (int64_t)c * VEC) * ES;
        char *orow = (char *)ov + (hrow * D + (int64_t)c * VEC) * ES;
        const int64_t toff = ((int64_t)s * (D / 2) + (int64_t)c * VEC) * ES;
        const v4i a1 = *(const v4i *)xr, a2 = *(const v4i *)(xr + half_bytes);
        const v4i cw = *(const v4i *)((const char *)cosv + toff), sw = *(const v4i *)((const char *)sinv + toff);
        v4i w1, w2;
        if constexpr (DT == ASQ_F16) {
            typedef _Float16 v2h __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (element copies first: __builtin_bit_cast applied to a vector-element expression `a1[i]` reads element 0 for every i with this hipcc)
                const int x1w = a1[i], x2w = a2[i], cwi = cw[i], swi = sw[i];
                const uint32_t t1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, x1w) * __builtin_bit_cast(v2h, cwi));   // dt(x1 * cos): v_pk_mul_f16
                const uint32_t t2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2h, x2w) * __builtin_bit_cast(v2h, cwi));
                w1[i] = (int)fma_mix_f16x2<true>((uint32_t)x2w, (uint32_t)swi, t1);    // dt(t1 - x2 * sin)
                w2[i] = (int)fma_mix_f16x2<false>((uint32_t)x1w, (uint32_t)swi, t2);   // dt(t2 + x1 * sin)
            }
        } else {
            float x1[VEC], x2[VEC], cs[VEC], sn[VEC], o1[VEC], o2[VEC];
            vec_unpack<DT>(a1, x1);
            vec_unpack<DT>(a2, x2);
            vec_unpack<DT>(cw, cs);
            vec_unpack<DT>(sw, sn);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float t1 = ElemT<DT>::round(__fmul_rn(x1[j], cs[j])), t2 = ElemT<DT>::round(__fmul_rn(x2[j], cs[j]));
                o1[j] = __fadd_rn(t1, __fmul_rn(-x2[j], sn[j]));   // self + (value * tensor1) * tensor2, value = -1
                o2[j] = __fadd_rn(t2, __fmul_rn(x1[j], sn[j]));
            }
            if constexpr (DT == ASQ_F32) {
                w1 = (v4i){__float_as_int(o1[0]), __float_as_int(o1[1]), __float_as_int(o1[2]), __float_as_int(o1[3])};
                w2 = (v4i){__float_as_int(o2[0]), __float_as_int(o2[1]), __float_as_int(o2[2]), __float_as_int(o2[3])};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w1[i] = (int)((uint32_t)ElemT<DT>::store(o1[2 * i]) | ((uint32_t)ElemT<DT>::store(o1[2 * i + 1]) << 16));
                    w2[i] = (int)((uint32_t)ElemT<DT>::store(o2[2 * i]) | ((uint32_t)ElemT<DT>::store(o2[2 * i + 1]) << 16));
                }
            }
        }
        *(v4i *)orow = w1;
        *(v4i *)(orow + half_bytes) = w2;
    }
}

}  // namespace asq
using namespace asq;

extern "C" int asq_rope(const void *x, int64_t x_row_pitch, void *out, int x_dtype, const void *cos_tab, const void *sin_tab, int64_t B, int64_t S, int64_t H, int64_t D, void *stream)
{
    const AsqRange range_("asq_rope");
    ASQ_REQUIRE(B >= 0 && S >= 0 && H >= 0 && D > 0 && S < (1ll << 31) && H < (1ll << 31) && D < (1ll << 20), ASQ_ERR_DIM, "asq_rope: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_rope: bad x_dtype %d", x_dtype);
    if (B == 0 || S == 0 || H == 0) return ASQ_OK;
    ASQ_REQUIRE(x && out && cos_tab && sin_tab, ASQ_ERR_NULL, "asq_rope: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(D % (2 * vec) == 0, ASQ_ERR_DIM, "asq_rope: head_dim must be a multiple of %d", 2 * vec);
    const int64_t ld = x_row_pitch == 0 ? H * D : x_row_pitch;
    ASQ_REQUIRE(ld >= H * D && ld % vec == 0, ASQ_ERR_DIM, "asq_rope: x_row_pitch must be 0 (dense) or >= H * D and a multiple of %d elements", vec);
    ASQ_REQUIRE(x != out || ld == H * D, ASQ_ERR_DIM, "asq_rope: in place only on a dense x");
    ASQ_REQUIRE(((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)cos_tab) | ((uintptr_t)sin_tab)) & 15) == 0, ASQ_ERR_ALIGN, "asq_rope: pointers must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nwork = B * S * H * (D / 2 / vec);
    int64_t blocks = (nwork + 255) / 256;
    blocks = blocks > 256 * 64 ? 256 * 64 : blocks;   // grid-stride beyond 64 blocks per CU
    switch (x_dtype) {
    case ASQ_F32: hipLaunchKernelGGL((rope_kernel<ASQ_F32>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, cos_tab, sin_tab, (int)S, (int)H, (int)D, ld, nwork); break;
    case ASQ_F16: hipLaunchKernelGGL((rope_kernel<ASQ_F16>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, cos_tab, sin_tab, (int)S, (int)H, (int)D, ld, nwork); break;
    default: hipLaunchKernelGGL((rope_kernel<ASQ_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, cos_tab, sin_tab, (int)S, (int)H, (int)D, ld, nwork); break;
    }
    return asq_after_launch(s, "asq_rope");
}
