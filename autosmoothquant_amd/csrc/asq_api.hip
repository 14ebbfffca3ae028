// C-ABI plumbing + the whole-module forward (quantise -> GEMM+epilogue on one stream).
#include "asq_common.h"
#include <string.h>
#include <dlfcn.h>

static thread_local char g_err[512] = "";

void asq_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int asq_debug_sync()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("ASQ_DEBUG_SYNC");
        v = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return v;
}

// ---- roctx ranges (ASQ_ROCTX=1)
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
int asq_roctx_enabled()
{
    static const int v = [] {
        const char *e = getenv("ASQ_ROCTX");
        if (!(e && e[0] && e[0] != '0')) return 0;
        // rocprofv3's marker library first, the legacy roctracer one second
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
            g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) return 1;
        }
        fprintf(stderr, "libasq_hip: ASQ_ROCTX=1 but no roctx library could be loaded; ranges are off\n");
        return 0;
    }();
    return v;
}
void asq_range_push(const char *name) { if (g_roctx_push) g_roctx_push(name); }
void asq_range_pop() { if (g_roctx_pop) g_roctx_pop(); }

extern "C" int asq_version(void) { return ASQ_VERSION; }
extern "C" const char *asq_last_error(void) { return g_err; }

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// workspace layout: [ GEMM workspace: header (asq_workspace_init) + scratch, padded to 256 | xq int8 M*K | pad to 256 | s_row f32 M | pad to 256 ]
// The header sits at offset 0 whatever the shape -- also for shapes whose GEMM needs no scratch -- so one initialised buffer serves every call
// that fits into it and no call writes activations over the header another call's kernel will read.
static inline size_t forward_gemm_part(int64_t M, int64_t N, int64_t K)
{
    const size_t g = asq_gemm_workspace_bytes(M, N, K);
    return round_up(g > asq_workspace_header_bytes() ? g : asq_workspace_header_bytes(), 256);
}
extern "C" size_t asq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M < 0 || K < 0 || N < 0) return 0;
    return forward_gemm_part(M, N, K) + round_up((size_t)M * (size_t)K, 256) + round_up((size_t)M * 4, 256) + round_up((size_t)M * 8, 256);   // (last: row offsets of asq_linear_w8a8_forward_off)
}

extern "C" int asq_linear_w8a8_forward(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K,
                                       int act_mode, float quant_scale, float s_scalar, const float *s_col, const float *bias,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_forward");
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_forward: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;
    const size_t need = round_up((size_t)M * (size_t)K, 256) + round_up((size_t)M * 4, 256);  // the GEMM part is optional
    ASQ_REQUIRE(workspace != nullptr && workspace_bytes >= need, ASQ_ERR_WORKSPACE,
                "asq_linear_w8a8_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    ASQ_REQUIRE(((uintptr_t)workspace & 255) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_forward: workspace must be 256-B aligned");
    {   // <= 16 rows on a weight-streaming shape: ONE launch, the quantiser is the GEMM's prologue (asq_gemm_skinny_fq.h; the workspace is not touched)
        int launched = 0;
        const int rc = asq_try_fused_forward(x, x_dtype, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, stream, &launched);
        if (rc || launched) return rc;
    }
    // Layout by size (never put activations over a header other calls rely on):
    //   >= gemm part + need : [ header + GEMM scratch | xq | s_row ]   (the full layout)
    //   >= header + need    : [ header | xq | s_row ], the GEMM runs without scratch -- the header of a shared, initialised buffer stays intact
    //   >= need             : [ xq | s_row ] from offset 0: a private scratch buffer without a header (first-generation kernels only)
    const size_t gbytes = forward_gemm_part(M, N, K), hdr = round_up(asq_workspace_header_bytes(), 256);
    const bool full = workspace_bytes >= gbytes + need;
    const bool with_gemm_ws = full && asq_gemm_workspace_bytes(M, N, K) > 0;
    char *base = (char *)workspace + (full ? gbytes : (workspace_bytes >= hdr + need ? hdr : 0));
    int8_t *xq = (int8_t *)base;
    float *s_row = (float *)(base + round_up((size_t)M * (size_t)K, 256));
    int rc = asq_quantize_act(x, x_dtype, act_mode, quant_scale, xq, s_row, M, K, stream);
    if (rc) return rc;
    return asq_linear_w8a8(xq, w, out, x_dtype, M, N, K, s_scalar, act_mode == ASQ_ACT_PER_TOKEN ? s_row : nullptr, s_col, bias,
                           ASQ_EPI_SCALE_FIRST, with_gemm_ws ? workspace : nullptr, with_gemm_ws ? gbytes : 0, stream);
}

// The same forward on OFFSET operand images (include/asq_hip.h): when the caller passes the weight's image and the shape is one the dispatcher gives to the
// 256 x 256 kernel, the quantiser emits x + cx[m] (+ the row's {cx, sum}) and the GEMM subtracts the two rank-1 correction terms in front of its epilogue
// (asq_gemm_p16.h); every other call is the plain forward.
extern "C" int asq_linear_w8a8_forward_off(const void *x, int x_dtype, const int8_t *w, const int8_t *w_off, const int32_t *col_off, void *out, int64_t M, int64_t N,
                                           int64_t K, int act_mode, float quant_scale, float s_scalar, const float *s_col, const float *bias, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_forward_off");
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_forward_off: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;
    const size_t plain = round_up((size_t)M * (size_t)K, 256) + round_up((size_t)M * 4, 256), need = plain + round_up((size_t)M * 8, 256);
    const size_t gbytes = forward_gemm_part(M, N, K);
    const bool use = w_off != nullptr && col_off != nullptr && workspace != nullptr && workspace_bytes >= gbytes + need && (((uintptr_t)x | (uintptr_t)w_off) & 15) == 0 &&
                     (((uintptr_t)col_off) & 15) == 0 && asq_offsets_supported(M, N, K, x_dtype);
    if (!use) return asq_linear_w8a8_forward(x, x_dtype, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, workspace, workspace_bytes, stream);
    ASQ_REQUIRE(((uintptr_t)workspace & 255) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_forward_off: workspace must be 256-B aligned");
    char *base = (char *)workspace + gbytes;
    int8_t *xq = (int8_t *)base;
    float *s_row = (float *)(base + round_up((size_t)M * (size_t)K, 256));
    int32_t *row_off = (int32_t *)(base + plain);
    int rc = asq_quantize_act_off(x, x_dtype, act_mode, quant_scale, xq, s_row, row_off, M, K, stream);
    if (rc) return rc;
    return asq_linear_w8a8_off(xq, w_off, out, x_dtype, M, N, K, s_scalar, act_mode == ASQ_ACT_PER_TOKEN ? s_row : nullptr, s_col, bias, ASQ_EPI_SCALE_FIRST, row_off,
                               col_off, stream);
}
