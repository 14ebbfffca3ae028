// C-ABI of the gate || up GEMM with the SiLU(gate) * up epilogue (asq_gemm_gateup.h) on the persistent 256 x 256 kernel.
#include "asq_gemm_kernels.h"

using namespace asq;

static bool gate_up_shape(int64_t M, int64_t F, int64_t K, int out_dtype)
{
    static const bool on = [] { const char *e = getenv("ASQ_GATE_UP"); return !(e && e[0] == '0'); }();   // ASQ_GATE_UP=0: callers fall back to two GEMMs + asq_silu_mul_quantize (A/B)
    if (!on || !(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16)) return false;
    if (M < 256 || M % 256 != 0 || F < 128 || F % 128 != 0 || K < 256 || K % 256 != 0 || K > OFFSET_MAX_K) return false;
    const int64_t T = (M / 256) * (2 * F / 256);
    return T > 8 * P8_CUS_PER_XCD && F * 2 * 128 < (int64_t(1) << 31) && M * K < (int64_t(1) << 40);
}

extern "C" int asq_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype) { return gate_up_shape(M, F, K, out_dtype) ? 1 : 0; }

template <int DT, bool ROW>
static int launch_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, int64_t M, int64_t F, int64_t K, float s_gate, float s_up, const float *s_row, int fast,
                          OffsetArgs off, hipStream_t s)
{
    using Epi = EpiGateUp<DT, ROW>;
    auto kfn = gemm_i8_p16p<Epi>;
    const hipError_t e = ensure_dynamic_lds((const void *)kfn, P16P_LDS_BYTES);
    if (e != hipSuccess) {
        asq_set_error("asq_linear_w8a8_gate_up: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int64_t N = 2 * F, tm = M / 256, tn = N / 256, T = tm * tn;
    const int64_t grid = T < 8 * P8_CUS_PER_XCD ? T : 8 * P8_CUS_PER_XCD;
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), P16P_LDS_BYTES, s, xq, w_gu, M, N, K, (int)tm, (int)tn, Epi{out, F, s_row, s_gate, s_up, 1.0f, fast}, off);
    return asq_after_launch(s, "asq_linear_w8a8_gate_up");
}

template <int DT, bool ROW>
static int launch_gate_up_q8(const int8_t *xq, const int8_t *w_gu, int8_t *out, int64_t M, int64_t F, int64_t K, float s_gate, float s_up, const float *s_row, int fast, float qs,
                             OffsetArgs off, hipStream_t s)
{
    using Epi = EpiGateUp<DT, ROW, true>;
    auto kfn = gemm_i8_p16p<Epi>;
    const hipError_t e = ensure_dynamic_lds((const void *)kfn, P16P_LDS_BYTES);
    if (e != hipSuccess) {
        asq_set_error("asq_linear_w8a8_gate_up_q8: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int64_t N = 2 * F, tm = M / 256, tn = N / 256, T = tm * tn;
    const int64_t grid = T < 8 * P8_CUS_PER_XCD ? T : 8 * P8_CUS_PER_XCD;
    Epi epi{out, F, s_row, s_gate, s_up, 1.0f, fast};
    epi.qs = qs;
    epi.qy = (qs > 0x1p-60f && qs < 0x1p60f) ? 1.0f / qs : 0.0f;   // (asq_quantize_act's own choice between the two divisions: the same int8 either way)
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), P16P_LDS_BYTES, s, xq, w_gu, M, N, K, (int)tm, (int)tn, epi, off);
    return asq_after_launch(s, "asq_linear_w8a8_gate_up_q8");
}

extern "C" int asq_linear_w8a8_gate_up_q8(const int8_t *xq, const int8_t *w_gu, int8_t *out_q, int act_dtype, int64_t M, int64_t F, int64_t K, float s_gate, float s_up,
                                          const float *s_row, int flags, float quant_scale, const int32_t *row_off, const int32_t *col_off, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_gate_up_q8");
    ASQ_REQUIRE(M >= 0 && F >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_gate_up_q8: bad dims");
    if (M == 0 || F == 0) return ASQ_OK;
    ASQ_REQUIRE(xq != nullptr && w_gu != nullptr && out_q != nullptr, ASQ_ERR_NULL, "asq_linear_w8a8_gate_up_q8: NULL xq / w_gu / out_q");
    ASQ_REQUIRE(act_dtype == ASQ_F16 || act_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_gate_up_q8: act_dtype must be ASQ_F16 or ASQ_BF16 (got %d)", act_dtype);
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_linear_w8a8_gate_up_q8: flags is a bit field (ASQ_SILU_FAST), got %d", flags);
    ASQ_REQUIRE(quant_scale > 0.0f && quant_scale == quant_scale, ASQ_ERR_DIM, "asq_linear_w8a8_gate_up_q8: quant_scale must be positive");
    ASQ_REQUIRE(gate_up_shape(M, F, K, act_dtype), ASQ_ERR_DIM,
                "asq_linear_w8a8_gate_up_q8: needs M %% 256 == 0, F %% 128 == 0, K %% 256 == 0, K <= 65536 and more than 256 tiles of 256 x 256 over [M, 2 F] (asq_gate_up_supported)");
    ASQ_REQUIRE((((uintptr_t)xq | (uintptr_t)w_gu | (uintptr_t)out_q) & 15) == 0 && (((uintptr_t)s_row) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_gate_up_q8: xq / w_gu / out_q must be 16-byte aligned");
    ASQ_REQUIRE((row_off == nullptr) == (col_off == nullptr) && ((((uintptr_t)row_off | (uintptr_t)col_off) & 7) == 0), ASQ_ERR_ALIGN,
                "asq_linear_w8a8_gate_up_q8: row_off and col_off come together (offset operand images), 8-byte aligned");
    const OffsetArgs off{row_off, col_off};
    const int fast = (flags & ASQ_SILU_FAST) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == ASQ_F16) return s_row ? launch_gate_up_q8<ASQ_F16, true>(xq, w_gu, out_q, M, F, K, s_gate, s_up, s_row, fast, quant_scale, off, s) : launch_gate_up_q8<ASQ_F16, false>(xq, w_gu, out_q, M, F, K, s_gate, s_up, s_row, fast, quant_scale, off, s);
    return s_row ? launch_gate_up_q8<ASQ_BF16, true>(xq, w_gu, out_q, M, F, K, s_gate, s_up, s_row, fast, quant_scale, off, s) : launch_gate_up_q8<ASQ_BF16, false>(xq, w_gu, out_q, M, F, K, s_gate, s_up, s_row, fast, quant_scale, off, s);
}

extern "C" int asq_linear_w8a8_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, int out_dtype, int64_t M, int64_t F, int64_t K, float s_gate, float s_up,
                                       const float *s_row, int flags, const int32_t *row_off, const int32_t *col_off, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_gate_up");
    ASQ_REQUIRE(M >= 0 && F >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_gate_up: bad dims");
    if (M == 0 || F == 0) return ASQ_OK;
    ASQ_REQUIRE(xq != nullptr && w_gu != nullptr && out != nullptr, ASQ_ERR_NULL, "asq_linear_w8a8_gate_up: NULL xq / w_gu / out");
    ASQ_REQUIRE(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_gate_up: out_dtype must be ASQ_F16 or ASQ_BF16 (got %d)", out_dtype);
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_linear_w8a8_gate_up: flags is a bit field (ASQ_SILU_FAST), got %d", flags);
    ASQ_REQUIRE(gate_up_shape(M, F, K, out_dtype), ASQ_ERR_DIM,
                "asq_linear_w8a8_gate_up: needs M %% 256 == 0, F %% 128 == 0, K %% 256 == 0, K <= 65536 and more than 256 tiles of 256 x 256 over [M, 2 F] (asq_gate_up_supported)");
    ASQ_REQUIRE((((uintptr_t)xq | (uintptr_t)w_gu | (uintptr_t)out) & 15) == 0 && (((uintptr_t)s_row) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_gate_up: xq / w_gu / out must be 16-byte aligned");
    ASQ_REQUIRE((row_off == nullptr) == (col_off == nullptr) && ((((uintptr_t)row_off | (uintptr_t)col_off) & 7) == 0), ASQ_ERR_ALIGN,
                "asq_linear_w8a8_gate_up: row_off and col_off come together (offset operand images), 8-byte aligned");
    const OffsetArgs off{row_off, col_off};
    const int fast = (flags & ASQ_SILU_FAST) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == ASQ_F16) return s_row ? launch_gate_up<ASQ_F16, true>(xq, w_gu, out, M, F, K, s_gate, s_up, s_row, fast, off, s) : launch_gate_up<ASQ_F16, false>(xq, w_gu, out, M, F, K, s_gate, s_up, s_row, fast, off, s);
    return s_row ? launch_gate_up<ASQ_BF16, true>(xq, w_gu, out, M, F, K, s_gate, s_up, s_row, fast, off, s) : launch_gate_up<ASQ_BF16, false>(xq, w_gu, out, M, F, K, s_gate, s_up, s_row, fast, off, s);
}

// ---- grouped form (round 5, last session): Mixtral's w1 || w3 (reference models/mixtral.py:99-101,142-145: w2(act(w1 x) * w3 x) per expert) as ONE grouped launch of the
// 256 x 256 kernel over the experts' interleaved [2 F, K] stacks; the [rows, F] SiLU(w1 x) * (w3 x) tensor is all that reaches HBM.  Scheduler, half tiles, in-launch K split
// of the tail round and offset images are asq_linear_w8a8_grouped[_off]'s (asq_gemm_p8.h); the epilogue is epilogue_gate_up_rows.
static bool grouped_gate_up_shape(int64_t M, int64_t F, int64_t K, int out_dtype)
{
    static const bool on = [] { const char *e = getenv("ASQ_GATE_UP"); return !(e && e[0] == '0'); }();
    if (!on || !(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16)) return false;
    return M >= 1 && F >= 128 && F % 128 == 0 && K >= 128 && K % 128 == 0 && K <= OFFSET_MAX_K && M * F < (int64_t(1) << 40);
}

extern "C" int asq_grouped_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype) { return grouped_gate_up_shape(M, F, K, out_dtype) ? 1 : 0; }

template <int DT>
static int launch_grouped_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, const int32_t *goffs, int ngroups, int64_t M, int64_t F, int64_t K, const float *s_gate,
                                  const float *s_up, int fast, OffsetArgs off, void *ws, size_t ws_bytes, hipStream_t s)
{
    using Epi = EpiGateUp<DT, false>;
    const int64_t N = 2 * F, tn = N / 256;
    int64_t tiles = (M / 256 + ngroups) * tn;   // upper bound on sum ceil(m_g / 256) * tn (launch_gemm_impl's grouped branch)
    char *gws = nullptr;
    if (ngroups <= P8_GROUPED_SCAN_MAX) {
        tiles = 8 * ((tiles + 7) / 8 + 2 + P8_CUS_PER_XCD);
        if (ws != nullptr && ws_bytes >= (size_t)WS_HEADER_BYTES + P8_GROUPED_WS_BYTES && grouped_tail_split_enabled()) gws = (char *)ws;
    }
    ASQ_REQUIRE(tiles < (1ll << 24), ASQ_ERR_DIM, "asq_linear_w8a8_grouped_gate_up: too many tiles");
    auto kfn = gemm_i8_p8<Epi, 0, true, true>;
    const bool g_offs = off.row != nullptr;
    const int lds = g_offs ? P16_LDS_BYTES : P8_LDS_BYTES;
    const hipError_t e = ensure_dynamic_lds((const void *)kfn, lds);
    if (e != hipSuccess) {
        asq_set_error("asq_linear_w8a8_grouped_gate_up: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    Epi epi{out, F, nullptr, 1.0f, 1.0f, 1.0f, fast};
    epi.sg_group = s_gate;
    epi.su_group = s_up;
    hipLaunchKernelGGL(kfn, dim3((unsigned)tiles), dim3(512), lds, s, xq, w_gu, M, N, K, 0, (int)tn, 1, goffs, ngroups, gws, epi, g_offs ? off : OffsetArgs{});
    return asq_after_launch(s, "asq_linear_w8a8_grouped_gate_up");
}

extern "C" int asq_linear_w8a8_grouped_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M, int64_t F,
                                               int64_t K, const float *s_gate, const float *s_up, int flags, const int32_t *row_off, const int32_t *col_off, void *workspace,
                                               size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_grouped_gate_up");
    ASQ_REQUIRE(M >= 0 && F >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_grouped_gate_up: bad dims");
    if (M == 0 || F == 0) return ASQ_OK;
    ASQ_REQUIRE(xq != nullptr && w_gu != nullptr && out != nullptr && group_offsets != nullptr && s_gate != nullptr && s_up != nullptr, ASQ_ERR_NULL,
                "asq_linear_w8a8_grouped_gate_up: NULL xq / w_gu / out / group_offsets / s_gate / s_up");
    ASQ_REQUIRE(ngroups > 0 && ngroups <= 4096, ASQ_ERR_DIM, "asq_linear_w8a8_grouped_gate_up: need 1 <= ngroups <= 4096");
    ASQ_REQUIRE(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_grouped_gate_up: out_dtype must be ASQ_F16 or ASQ_BF16 (got %d)", out_dtype);
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_linear_w8a8_grouped_gate_up: flags is a bit field (ASQ_SILU_FAST), got %d", flags);
    ASQ_REQUIRE(grouped_gate_up_shape(M, F, K, out_dtype), ASQ_ERR_DIM, "asq_linear_w8a8_grouped_gate_up: needs F %% 128 == 0, K %% 128 == 0, K <= 65536 (asq_grouped_gate_up_supported)");
    ASQ_REQUIRE((((uintptr_t)xq | (uintptr_t)w_gu | (uintptr_t)out) & 15) == 0 && ((((uintptr_t)s_gate | (uintptr_t)s_up | (uintptr_t)group_offsets) & 3) == 0), ASQ_ERR_ALIGN,
                "asq_linear_w8a8_grouped_gate_up: xq / w_gu / out must be 16-byte aligned");
    ASQ_REQUIRE((row_off == nullptr) == (col_off == nullptr) && ((((uintptr_t)row_off | (uintptr_t)col_off) & 7) == 0), ASQ_ERR_ALIGN,
                "asq_linear_w8a8_grouped_gate_up: row_off and col_off come together (offset operand images), 8-byte aligned");
    ASQ_REQUIRE(workspace == nullptr || (((uintptr_t)workspace) & 255) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_grouped_gate_up: workspace must be 256-B aligned");
    const OffsetArgs off{row_off, col_off};
    const int fast = (flags & ASQ_SILU_FAST) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    return out_dtype == ASQ_F16 ? launch_grouped_gate_up<ASQ_F16>(xq, w_gu, out, group_offsets, ngroups, M, F, K, s_gate, s_up, fast, off, workspace, workspace_bytes, s)
                                : launch_grouped_gate_up<ASQ_BF16>(xq, w_gu, out, group_offsets, ngroups, M, F, K, s_gate, s_up, fast, off, workspace, workspace_bytes, s);
}

// ---- fp8 grouped form (round 6): FP8LinearDynamic experts' w1 || w3 (interleaved in blocks of 32 channels) as one grouped launch with the SiLU * up epilogue
// (EpiGateUpFp8, asq_gemm_gateup.h).  Per-token activation scales, per-group weight scales.
static bool fp8_grouped_gate_up_shape(int64_t M, int64_t F, int64_t K, int out_dtype)
{
    static const bool on = [] { const char *e = getenv("ASQ_GATE_UP"); return !(e && e[0] == '0'); }();
    if (!on || !(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16)) return false;
    return M >= 1 && F >= 128 && F % 128 == 0 && K >= 128 && K % 128 == 0 && K < (int64_t(1) << 31) && M * F < (int64_t(1) << 40);
}

extern "C" int asq_fp8_grouped_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype) { return fp8_grouped_gate_up_shape(M, F, K, out_dtype) ? 1 : 0; }

template <int DT>
static int launch_fp8_grouped_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, const int32_t *goffs, int ngroups, int64_t M, int64_t F, int64_t K, const float *a_scale,
                                      const float *s_gate, const float *s_up, int fast, hipStream_t s)
{
    using Epi = EpiGateUpFp8<DT>;
    const int64_t N = 2 * F, tn = N / 256;
    int64_t tiles = (M / 256 + ngroups) * tn;   // upper bound on sum ceil(m_g / 256) * tn
    if (ngroups <= P8_GROUPED_SCAN_MAX) tiles = 8 * ((tiles + 7) / 8 + 2 + P8_CUS_PER_XCD);
    ASQ_REQUIRE(tiles < (1ll << 24), ASQ_ERR_DIM, "asq_linear_fp8_grouped_gate_up: too many tiles");
    auto kfn = gemm_i8_p8<Epi, 0, true, false>;
    const hipError_t e = ensure_dynamic_lds((const void *)kfn, P8_LDS_BYTES);
    if (e != hipSuccess) {
        asq_set_error("asq_linear_fp8_grouped_gate_up: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    const Epi epi{out, F, a_scale, s_gate, s_up, 1.0f, 1.0f, fast};
    hipLaunchKernelGGL(kfn, dim3((unsigned)tiles), dim3(512), P8_LDS_BYTES, s, xq, w_gu, M, N, K, 0, (int)tn, 1, goffs, ngroups, (char *)nullptr, epi, OffsetArgs{});
    return asq_after_launch(s, "asq_linear_fp8_grouped_gate_up");
}

extern "C" int asq_linear_fp8_grouped_gate_up(const uint8_t *xq, const uint8_t *w_gu, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M, int64_t F,
                                              int64_t K, const float *a_scale, const float *s_gate, const float *s_up, int flags, void *stream)
{
    const AsqRange range_("asq_linear_fp8_grouped_gate_up");
    ASQ_REQUIRE(M >= 0 && F >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_fp8_grouped_gate_up: bad dims");
    if (M == 0 || F == 0) return ASQ_OK;
    ASQ_REQUIRE(xq != nullptr && w_gu != nullptr && out != nullptr && group_offsets != nullptr && a_scale != nullptr && s_gate != nullptr && s_up != nullptr, ASQ_ERR_NULL,
                "asq_linear_fp8_grouped_gate_up: NULL xq / w_gu / out / group_offsets / a_scale / s_gate / s_up");
    ASQ_REQUIRE(ngroups > 0 && ngroups <= 4096, ASQ_ERR_DIM, "asq_linear_fp8_grouped_gate_up: need 1 <= ngroups <= 4096");
    ASQ_REQUIRE(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_fp8_grouped_gate_up: out_dtype must be ASQ_F16 or ASQ_BF16 (got %d)", out_dtype);
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_linear_fp8_grouped_gate_up: flags is 0 or ASQ_SILU_FAST, got %d", flags);
    ASQ_REQUIRE(fp8_grouped_gate_up_shape(M, F, K, out_dtype), ASQ_ERR_DIM, "asq_linear_fp8_grouped_gate_up: needs F %% 128 == 0, K %% 128 == 0 (asq_fp8_grouped_gate_up_supported)");
    ASQ_REQUIRE((((uintptr_t)xq | (uintptr_t)w_gu | (uintptr_t)out) & 15) == 0 && ((((uintptr_t)a_scale | (uintptr_t)s_gate | (uintptr_t)s_up | (uintptr_t)group_offsets) & 3) == 0),
                ASQ_ERR_ALIGN, "asq_linear_fp8_grouped_gate_up: xq / w_gu / out must be 16-byte aligned");
    const int fast = (flags & ASQ_SILU_FAST) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    return out_dtype == ASQ_F16
               ? launch_fp8_grouped_gate_up<ASQ_F16>((const int8_t *)xq, (const int8_t *)w_gu, out, group_offsets, ngroups, M, F, K, a_scale, s_gate, s_up, fast, s)
               : launch_fp8_grouped_gate_up<ASQ_BF16>((const int8_t *)xq, (const int8_t *)w_gu, out, group_offsets, ngroups, M, F, K, a_scale, s_gate, s_up, fast, s);
}
