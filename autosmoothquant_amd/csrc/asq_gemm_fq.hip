// The one-launch module forward for decode-sized inputs: C-ABI glue of gemm_i8_skinny_fq (asq_gemm_skinny_fq.h).
#include "asq_gemm_kernels.h"
#include "asq_gemm_skinny_fq.h"

using namespace asq;

// ASQ_FUSED_FORWARD=0: asq_linear_w8a8_forward always runs quantiser + GEMM as two launches (A/B switch; read once)
static bool fused_enabled()
{
    static const bool on = [] {
        const char *e = getenv("ASQ_FUSED_FORWARD");
        return !(e && e[0] == '0');
    }();
    return on;
}

static bool fused_shape(const void *x, const void *w, int64_t M, int64_t N, int64_t K, int x_dtype, int act_mode)
{
    if (!fused_enabled() || forced_kernel() >= 0) return false;
    if (!skfq_supported(x, w, M, N, K, x_dtype, act_mode)) return false;
    // only where the dispatcher streams the weight anyway (gemm_i8_skinny's region, outside the stream-K kernel's: those are long-K shapes whose X image would not fit)
    return pick_kernel(x, w, M, N, K) == KERN_SKINNY && plan_wstream(M, N, K).G == 0;
}

extern "C" int asq_forward_fused_supported(int64_t M, int64_t N, int64_t K, int x_dtype, int act_mode)
{
    if (!(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16)) return 0;
    if (!(act_mode == ASQ_ACT_ROUND || act_mode == ASQ_ACT_DIV || act_mode == ASQ_ACT_PER_TOKEN)) return 0;
    return fused_shape(nullptr, nullptr, M, N, K, x_dtype, act_mode) ? 1 : 0;
}

static int fused_launch(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int act_mode, float quant_scale, float s_scalar,
                        const float *s_col, const float *bias, void *stream, int *launched);

// called by asq_linear_w8a8_forward (asq_api.hip) after its own argument checks; *launched = 0 means "not this shape": the caller runs the two-launch forward
int asq_try_fused_forward(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int act_mode, float quant_scale, float s_scalar,
                          const float *s_col, const float *bias, void *stream, int *launched)
{
    *launched = 0;
    if (!fused_shape(x, w, M, N, K, x_dtype, act_mode)) return ASQ_OK;
    return fused_launch(x, x_dtype, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, stream, launched);
}

// The one-launch forward on request: any shape the kernel can run (<= 16 rows, K % 128 == 0, the int8 activation image fits 64 KiB of LDS, 16-byte aligned x / w),
// whatever asq_linear_w8a8_forward would choose by itself.  ASQ_ERR_DIM when the kernel cannot run the shape.
extern "C" int asq_linear_w8a8_forward_fused(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int act_mode, float quant_scale,
                                             float s_scalar, const float *s_col, const float *bias, void *stream)
{
    const AsqRange range_("asq_linear_w8a8_forward_fused");
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_w8a8_forward_fused: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(skfq_kernel_ok(x, w, M, N, K), ASQ_ERR_DIM,
                "asq_linear_w8a8_forward_fused: needs 1 <= M <= 16, K %% 128 == 0, rows(M) x K <= 65536 bytes (rows = 4 / 8 / 16), N x K < 2^32, 16-byte aligned x and w");
    int launched = 0;
    return fused_launch(x, x_dtype, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, stream, &launched);
}

static int fused_launch(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int act_mode, float quant_scale, float s_scalar,
                        const float *s_col, const float *bias, void *stream, int *launched)
{
    // the argument checks of the two entry points this launch replaces (asq_quantize_act, asq_linear_w8a8)
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8_forward: bad x_dtype %d", x_dtype);
    ASQ_REQUIRE(act_mode == ASQ_ACT_ROUND || act_mode == ASQ_ACT_DIV || act_mode == ASQ_ACT_PER_TOKEN, ASQ_ERR_DTYPE, "asq_linear_w8a8_forward: bad mode %d", act_mode);
    ASQ_REQUIRE(x != nullptr && w != nullptr && out != nullptr, ASQ_ERR_NULL, "asq_linear_w8a8_forward: NULL x / w / out");
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(x_dtype)) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_forward: out misaligned");
    ASQ_REQUIRE((((uintptr_t)s_col | (uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8_forward: scale/bias misaligned");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (x_dtype) {
    case ASQ_F32: rc = launch_skinny_fq<ASQ_F32>(x, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, s); break;
    case ASQ_F16: rc = launch_skinny_fq<ASQ_F16>(x, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, s); break;
    default: rc = launch_skinny_fq<ASQ_BF16>(x, w, out, M, N, K, act_mode, quant_scale, s_scalar, s_col, bias, s); break;
    }
    if (rc) return rc;
    *launched = 1;
    return asq_after_launch(s, "asq_linear_w8a8_forward(fused)");
}
