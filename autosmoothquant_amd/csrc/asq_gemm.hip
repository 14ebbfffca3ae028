// C-ABI entry points of the INT8 GEMM family (kernels: asq_gemm_kernels.h, asq_gemm_p8.h;
// fp epilogue instantiations: asq_gemm_inst_{f32,f16,bf16}.hip).
#include "asq_gemm_kernels.h"

namespace asq {

int forced_kernel()
{
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("ASQ_GEMM_KERNEL");
        v = -1;
        if (e) {
            if (!strcmp(e, "generic")) v = KERN_GENERIC;
            else if (!strcmp(e, "p8")) v = KERN_P8;
            else if (!strcmp(e, "p8h")) v = KERN_P8H;
            else if (!strcmp(e, "p4")) v = KERN_P4;
            else if (!strcmp(e, "p16")) v = KERN_P16;
            else if (!strcmp(e, "p4x16")) v = KERN_P4X16;
            else if (!strcmp(e, "p8q")) v = KERN_P8Q;
            else if (!strcmp(e, "skinny")) v = KERN_SKINNY;
        }
    }
    return v;
}

static int check_gemm_args(const char *what, const void *x, const void *w, const void *out, int64_t M, int64_t N, int64_t K)
{
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), ASQ_ERR_DIM,
                "%s: bad dims M=%lld N=%lld K=%lld", what, (long long)M, (long long)N, (long long)K);
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(out != nullptr, ASQ_ERR_NULL, "%s: NULL out", what);
    ASQ_REQUIRE(K == 0 || (x != nullptr && w != nullptr), ASQ_ERR_NULL, "%s: NULL x / w", what);
    return ASQ_OK;
}

}  // namespace asq

using namespace asq;

extern "C" const char *asq_gemm_kernel_name(int64_t M, int64_t N, int64_t K)
{
    const GemmKernel kern = pick_kernel(nullptr, nullptr, M, N, K);
    if (plan_tail_peel(kern, M, N, K).n_main > 0) return kern == KERN_P8 ? "p8+tail" : kern == KERN_P8H ? "p8h+tail" : kern == KERN_P16 ? "p16+tail" : "p4+tail";
    switch (kern) {
    case KERN_P8: return "p8";
    case KERN_P8H: return "p8h";
    case KERN_P4: return "p4";
    case KERN_P16: return "p16";
    case KERN_P4X16: return "p4x16";
    case KERN_P8Q: return "p8q";
    case KERN_SKINNY: return "skinny";
    default: return "generic";
    }
}

extern "C" size_t asq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const GemmKernel kern = pick_kernel(nullptr, nullptr, M, N, K);
    size_t scratch = 0;
    const TailPeel tp = plan_tail_peel(kern, M, N, K);
    if (tp.n_main > 0) {
        scratch = tp.ws_bytes;  // (a launch with >= 256 tiles never splits K as a whole; its peeled remainder may)
    } else if (kern == KERN_SKINNY) {
        scratch = plan_wstream(M, N, K).bytes;  // partial slabs of the in-launch reduction (asq_gemm_wstream.h)
    } else if (kern == KERN_P8 || kern == KERN_P8H || kern == KERN_P8Q) {
        const int s = kern == KERN_P8    ? pick_ksplit(((M + 255) / 256) * ((N + 255) / 256), K, M, N, (size_t)-1)
                      : kern == KERN_P8H ? pick_ksplit_p8h(((M + 127) / 128) * ((N + 255) / 256), K, M, N, (size_t)-1)
                                         : pick_ksplit_p8q(((M + 127) / 128) * ((N + 127) / 128), K, M, N, (size_t)-1);
        scratch = s > 1 ? (size_t)s * (size_t)M * (size_t)N * 4 : 0;
        if (kern == KERN_P8Q) {   // the in-launch reduction of gemm_i8_p8q2<Epi, true> keeps one 64 KiB register image per (tile, split): whole tiles, and its own split count
            const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
            const int sf = tiles <= WS_MAX_GROUPS ? pick_ksplit_p8q(tiles, K, M, N, (size_t)-1, true) : 1;
            const size_t fb = sf > 1 ? p8q_fix_bytes(tiles, sf) : 0;
            scratch = fb > scratch ? fb : scratch;
        }
    }
    return scratch ? scratch + WS_HEADER_BYTES : 0;
}

extern "C" int asq_workspace_init(void *workspace, size_t workspace_bytes, void *stream)
{
    ASQ_REQUIRE(workspace != nullptr, ASQ_ERR_NULL, "asq_workspace_init: NULL workspace");
    ASQ_REQUIRE(workspace_bytes >= (size_t)WS_HEADER_BYTES, ASQ_ERR_WORKSPACE, "asq_workspace_init: workspace %zu B < header %d B (asq_workspace_header_bytes)", workspace_bytes, WS_HEADER_BYTES);
    ASQ_REQUIRE((((uintptr_t)workspace) & 255) == 0, ASQ_ERR_ALIGN, "asq_workspace_init: workspace must be 256-B aligned");
    hipLaunchKernelGGL(ws_init_header, dim3(1), dim3(256), 0, (hipStream_t)stream, (unsigned *)workspace);
    return asq_after_launch((hipStream_t)stream, "asq_workspace_init");
}

extern "C" size_t asq_workspace_header_bytes(void) { return (size_t)WS_HEADER_BYTES; }

extern "C" int asq_gemm_i8_i32(const int8_t *x, const int8_t *w, int32_t *out, int64_t M, int64_t N, int64_t K, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_gemm_i8_i32");
    int rc = check_gemm_args("asq_gemm_i8_i32", x, w, out, M, N, K);
    if (rc) return rc;
    ASQ_REQUIRE(((uintptr_t)out & 3) == 0, ASQ_ERR_ALIGN, "asq_gemm_i8_i32: out misaligned");
    EpiI32 epi{out, N, (N % 4 == 0) && (((uintptr_t)out & 15) == 0)};
    return launch_gemm(x, w, M, N, K, epi, (hipStream_t)stream, "asq_gemm_i8_i32", workspace, workspace_bytes);
}

extern "C" int asq_gemm_i8_i8(const int8_t *x, const int8_t *w, int8_t *out, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_gemm_i8_i8");
    int rc = check_gemm_args("asq_gemm_i8_i8", x, w, out, M, N, K);
    if (rc) return rc;
    EpiI8 epi{out, N, alpha, beta, (N % 4 == 0) && (((uintptr_t)out & 3) == 0)};
    return launch_gemm(x, w, M, N, K, epi, (hipStream_t)stream, "asq_gemm_i8_i8", workspace, workspace_bytes);
}

extern "C" int asq_linear_w8a8(const int8_t *xq, const int8_t *w, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                               float s_scalar, const float *s_row, const float *s_col, const float *bias, int epi_order, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8");
    int rc = check_gemm_args("asq_linear_w8a8", xq, w, out, M, N, K);
    if (rc) return rc;
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(epi_order == ASQ_EPI_SCALE_FIRST || epi_order == ASQ_EPI_ACC_FIRST, ASQ_ERR_DTYPE, "asq_linear_w8a8: bad epi_order %d", epi_order);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8: out misaligned");
    ASQ_REQUIRE((((uintptr_t)s_row | (uintptr_t)s_col | (uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8: scale/bias misaligned");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)s_col | (uintptr_t)bias) & 15) == 0);
    DequantArgs a{xq, w, out, M, N, K, s_scalar, s_row, s_col, bias, epi_order, vec_ok, workspace, workspace_bytes};
    hipStream_t s = (hipStream_t)stream;
    switch (out_dtype) {
    case ASQ_F32: return launch_dequant<ASQ_F32>(a, s);
    case ASQ_F16: return launch_dequant<ASQ_F16>(a, s);
    default: return launch_dequant<ASQ_BF16>(a, s);
    }
}

// ASQ_OFFSETS=0: the module-level paths never use offset operand images (A/B switch; read once)
static bool offsets_enabled()
{
    static const bool on = [] {
        const char *e = getenv("ASQ_OFFSETS");
        return !(e && e[0] == '0');
    }();
    return on;
}

extern "C" int asq_offsets_supported(int64_t M, int64_t N, int64_t K, int out_dtype)
{
    if (!offsets_enabled() || (out_dtype != ASQ_F16 && out_dtype != ASQ_BF16 && out_dtype != ASQ_F32) || M <= 0 || N <= 0 || K <= 0) return 0;
    if (!offsets_shape_ok(nullptr, nullptr, M, N, K) || K / (out_dtype == ASQ_F32 ? 4 : 8) > 256 * 20) return 0;   // (second: asq_quantize_act_off keeps the row in registers)
    const GemmKernel kern = pick_kernel(nullptr, nullptr, M, N, K);
    if (kern != KERN_P16) return 0;
    // a launch with a peeled column remainder runs on plain operands (the remainder kernels take no images) -- unless it has three or more full rounds: then the
    // images are worth more than the peel (8192 x 11008 x 4096: -5.7 % against -3.2 %; 2048 x 11008 x 4096: -2.7 % against -9.5 %; profiles/r4_tail_rule_ab.txt)
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    return plan_tail_peel(kern, M, N, K).n_main == 0 || tiles >= 3 * 256 ? 1 : 0;
}

extern "C" int asq_linear_w8a8_off(const int8_t *xq_off, const int8_t *w_off, void *out, int out_dtype, int64_t M, int64_t N, int64_t K, float s_scalar,
                                   const float *s_row, const float *s_col, const float *bias, int epi_order, const int32_t *row_off, const int32_t *col_off,
                                   void *stream)
{
    const AsqRange range_("asq_linear_w8a8_off");
    int rc = check_gemm_args("asq_linear_w8a8_off", xq_off, w_off, out, M, N, K);
    if (rc) return rc;
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16 || out_dtype == ASQ_F32, ASQ_ERR_DTYPE, "asq_linear_w8a8_off: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(epi_order == ASQ_EPI_SCALE_FIRST || epi_order == ASQ_EPI_ACC_FIRST, ASQ_ERR_DTYPE, "asq_linear_w8a8_off: bad epi_order %d", epi_order);
    ASQ_REQUIRE(row_off != nullptr && col_off != nullptr, ASQ_ERR_NULL, "asq_linear_w8a8_off: NULL row_off / col_off");
    ASQ_REQUIRE(offsets_shape_ok(xq_off, w_off, M, N, K), ASQ_ERR_DIM,
                "asq_linear_w8a8_off: needs K %% 128 == 0, 128 <= K <= 65536, N %% 4 == 0 and 16-B aligned operands (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0 && (((uintptr_t)row_off & 7) == 0) && (((uintptr_t)col_off & 15) == 0), ASQ_ERR_ALIGN, "asq_linear_w8a8_off: out / row_off / col_off misaligned");
    ASQ_REQUIRE((((uintptr_t)s_row | (uintptr_t)s_col | (uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_off: scale/bias misaligned");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)s_col | (uintptr_t)bias) & 15) == 0);
    DequantArgs a{xq_off, w_off, out, M, N, K, s_scalar, s_row, s_col, bias, epi_order, vec_ok, nullptr, 0};
    a.off = OffsetArgs{row_off, col_off};
    hipStream_t s = (hipStream_t)stream;
    return out_dtype == ASQ_F16 ? launch_dequant<ASQ_F16>(a, s) : out_dtype == ASQ_BF16 ? launch_dequant<ASQ_BF16>(a, s) : launch_dequant<ASQ_F32>(a, s);
}

extern "C" int asq_linear_w8a8_q8(const int8_t *xq, const int8_t *w, int8_t *out_q, int mid_dtype, int64_t M, int64_t N, int64_t K, float s_scalar,
                                  const float *s_row, const float *s_col, const float *bias, int epi_order, int act, int qmode, float quant_scale,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_q8");
    int rc = check_gemm_args("asq_linear_w8a8_q8", xq, w, out_q, M, N, K);
    if (rc) return rc;
    ASQ_REQUIRE(mid_dtype == ASQ_F32 || mid_dtype == ASQ_F16 || mid_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_q8: bad mid_dtype %d", mid_dtype);
    ASQ_REQUIRE(epi_order == ASQ_EPI_SCALE_FIRST || epi_order == ASQ_EPI_ACC_FIRST, ASQ_ERR_DTYPE, "asq_linear_w8a8_q8: bad epi_order %d", epi_order);
    ASQ_REQUIRE((act == 0 || act == 1) && (qmode == ASQ_ACT_ROUND || qmode == ASQ_ACT_DIV), ASQ_ERR_DTYPE, "asq_linear_w8a8_q8: bad act %d / qmode %d", act, qmode);
    ASQ_REQUIRE((((uintptr_t)s_row | (uintptr_t)s_col | (uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_q8: scale/bias misaligned");
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out_q & 3) == 0);
    DequantQArgs a{{xq, w, out_q, M, N, K, s_scalar, s_row, s_col, bias, epi_order, vec_ok, workspace, workspace_bytes}, act, qmode, quant_scale};
    hipStream_t s = (hipStream_t)stream;
    switch (mid_dtype) {
    case ASQ_F32: return launch_dequant_q<ASQ_F32>(a, s);
    case ASQ_F16: return launch_dequant_q<ASQ_F16>(a, s);
    default: return launch_dequant_q<ASQ_BF16>(a, s);
    }
}

extern "C" size_t asq_grouped_workspace_bytes(int64_t M, int64_t N, int64_t K, int ngroups)
{
    if (M <= 0 || N <= 0 || K < 128 * 2 * P8_TAIL_MIN_KTILES || ngroups <= 0 || ngroups > P8_GROUPED_SCAN_MAX) return 0;
    return (size_t)WS_HEADER_BYTES + P8_GROUPED_WS_BYTES;
}

extern "C" int asq_linear_w8a8_grouped(const int8_t *xq, const int8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups,
                                       int64_t M, int64_t N, int64_t K, const float *s_group, const float *s_row, const float *bias, void *stream)
{
    return asq_linear_w8a8_grouped_ws(xq, w, out, out_dtype, group_offsets, ngroups, M, N, K, s_group, s_row, bias, nullptr, 0, stream);
}

static int grouped_impl(const int8_t *xq, const int8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M, int64_t N, int64_t K,
                        const float *s_group, const float *s_row, const float *bias, const int32_t *row_off, const int32_t *col_off, void *workspace,
                        size_t workspace_bytes, void *stream);

extern "C" int asq_linear_w8a8_grouped_ws(const int8_t *xq, const int8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups,
                                          int64_t M, int64_t N, int64_t K, const float *s_group, const float *s_row, const float *bias,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_grouped_ws");
    return grouped_impl(xq, w, out, out_dtype, group_offsets, ngroups, M, N, K, s_group, s_row, bias, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int asq_linear_w8a8_grouped_off(const int8_t *xq_off, const int8_t *w_off, void *out, int out_dtype, const int32_t *group_offsets, int ngroups,
                                           int64_t M, int64_t N, int64_t K, const float *s_group, const float *s_row, const float *bias,
                                           const int32_t *row_off, const int32_t *col_off, void *workspace, size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_grouped_off");
    ASQ_REQUIRE(M == 0 || N == 0 || (row_off != nullptr && col_off != nullptr), ASQ_ERR_NULL, "asq_linear_w8a8_grouped_off: NULL row_off / col_off");
    ASQ_REQUIRE(out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_grouped_off: out_dtype must be ASQ_F16 or ASQ_BF16 (got %d)", out_dtype);
    ASQ_REQUIRE((((uintptr_t)row_off | (uintptr_t)col_off) & 7) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_grouped_off: row_off / col_off must be 8-B aligned");
    return grouped_impl(xq_off, w_off, out, out_dtype, group_offsets, ngroups, M, N, K, s_group, s_row, bias, row_off, col_off, workspace, workspace_bytes, stream);
}

static int grouped_impl(const int8_t *xq, const int8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M, int64_t N, int64_t K,
                        const float *s_group, const float *s_row, const float *bias, const int32_t *row_off, const int32_t *col_off, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    int rc = check_gemm_args("asq_linear_w8a8_grouped", xq, w, out, M, N, K);
    if (rc) return rc;
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(group_offsets != nullptr && s_group != nullptr && ngroups > 0 && ngroups <= 4096, ASQ_ERR_NULL,
                "asq_linear_w8a8_grouped: need group_offsets, s_group and 1 <= ngroups <= 4096");
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_grouped: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0 && ((((uintptr_t)s_row | (uintptr_t)s_group | (uintptr_t)bias | (uintptr_t)group_offsets) & 3) == 0),
                ASQ_ERR_ALIGN, "asq_linear_w8a8_grouped: misaligned pointer");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)bias) & 15) == 0);
    ASQ_REQUIRE(workspace == nullptr || (((uintptr_t)workspace) & 255) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_grouped: workspace must be 256-B aligned");
    DequantArgs a{xq, w, out, M, N, K, 1.0f, s_row, nullptr, bias, ASQ_EPI_SCALE_FIRST, vec_ok, workspace, workspace ? workspace_bytes : 0};
    a.s_group = s_group;
    a.goffs = group_offsets;
    a.ngroups = ngroups;
    a.off = OffsetArgs{row_off, col_off};
    hipStream_t s = (hipStream_t)stream;
    switch (out_dtype) {
    case ASQ_F32: return launch_dequant<ASQ_F32>(a, s);
    case ASQ_F16: return launch_dequant<ASQ_F16>(a, s);
    default: return launch_dequant<ASQ_BF16>(a, s);
    }
}
