// FP8 (OCP e4m3fn) linear path: activation quantisers + GEMM entry point.
// Reference: layers/functional/quantization.py:144-211 (per_tensor / per_token / static
// quantisers), layers/nn/linear.py:336-369 (easy_fp8_gemm), :373-452 (FP8LinearDynamic),
// :503-580 (FP8LinearStatic).  The reference has native_fp8_support = False hard-coded and
// runs a full-precision F.linear on dequantised operands; here the product runs on the fp8
// matrix cores (v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate) and the two scales are applied once
// in the epilogue.
#include "asq_gemm_kernels.h"

namespace asq {

// fp32 -> e4m3fn bits, round-to-nearest-even, unsaturated (values that round past 448 give NaN),
// bit-identical to tensor.to(torch.float8_e4m3fn) (checked exhaustively in oracle/fp8.py's tests).
__device__ __forceinline__ uint32_t f32_to_e4m3fn_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    const uint32_t sign = (u >> 24) & 0x80u;
    const uint32_t a = u & 0x7FFFFFFFu;
    uint32_t r;
    if (a >= 0x43F00000u) {  // |f| >= 480 (or inf / NaN): above the rounding boundary of the max finite 448
        r = 0x7Fu;
    } else if (a < 0x3C800000u) {  // |f| < 2^-6: result is subnormal (or zero): align to 2^-9 steps with an fp32 add (RNE)
        const float t = __uint_as_float(a) + 16384.0f;  // 2^14: ulp = 2^-9
        r = __float_as_uint(t) - 0x46800000u;
    } else {  // normal: rebias exponent (127 -> 7), round mantissa 23 -> 3 bits to nearest even
        uint32_t v = a - (120u << 23);
        v += 0x7FFFFu + ((v >> 20) & 1u);
        r = v >> 20;
    }
    return r | sign;
}

// fp32 -> e5m2 bits, round-to-nearest-even, IEEE-like (overflow -> inf), = tensor.to(torch.float8_e5m2)
__device__ __forceinline__ uint32_t f32_to_e5m2_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    const uint32_t sign = (u >> 24) & 0x80u;
    const uint32_t a = u & 0x7FFFFFFFu;
    uint32_t r;
    if (a > 0x7F800000u) {  // NaN
        r = 0x7Fu;
    } else if (a >= 0x47800000u) {  // |f| >= 65536 = 2^16 (incl. inf): beyond the rounding boundary of 57344
        r = 0x7Cu;
    } else if (a < 0x38800000u) {  // |f| < 2^-14: subnormal, ulp 2^-16
        const float t = __uint_as_float(a) + 128.0f;  // 2^7: ulp = 2^-16
        r = __float_as_uint(t) - 0x43000000u;
    } else {
        uint32_t v = a - (112u << 23);  // rebias 127 -> 15
        v += 0xFFFFFu + ((v >> 21) & 1u);
        r = v >> 21;  // may round up to 0x7C = inf, as IEEE
    }
    return r | sign;
}

struct F8CastE5M2 {
    __device__ __forceinline__ uint32_t operator()(float x) const { return f32_to_e5m2_bits(x); }
};

__device__ __forceinline__ float clamp448(float v) { return (v != v) ? v : fminf(fmaxf(v, -448.0f), 448.0f); }

template <int DT> struct F8Tok {  // per-token: x / f32 scale promotes to fp32
    float s;
    __device__ __forceinline__ uint32_t operator()(float x) const { return f32_to_e4m3fn_bits(clamp448(x / s)); }
};
template <int DT> struct F8Div {  // per-tensor (dynamic or static): quotient stays in x's dtype
    float s;
    __device__ __forceinline__ uint32_t operator()(float x) const { return f32_to_e4m3fn_bits(clamp448(ElemT<DT>::round(x / s))); }
};

template <int DT, class Q> __device__ __forceinline__ void f8_quant_vec(const v4i &v, const Q &q, uint32_t (&o)[2])
{
    if constexpr (DT == ASQ_F32) {
        o[0] = q(__int_as_float(v[0])) | (q(__int_as_float(v[1])) << 8) | (q(__int_as_float(v[2])) << 16) | (q(__int_as_float(v[3])) << 24);
        o[1] = 0;
    } else {
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = (uint32_t)v[i];
            r[2 * i] = q(ElemT<DT>::load((uint16_t)(w & 0xFFFF)));
            r[2 * i + 1] = q(ElemT<DT>::load((uint16_t)(w >> 16)));
        }
        o[0] = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        o[1] = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
    }
}

__device__ __forceinline__ float f8_nanmax(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }

template <int DT> __device__ __forceinline__ float f8_vec_absmax(const v4i &v)
{
    float m = 0.0f;
    if constexpr (DT == ASQ_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m = f8_nanmax(m, fabsf(__int_as_float(v[i])));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = (uint32_t)v[i];
            m = f8_nanmax(m, fabsf(ElemT<DT>::load((uint16_t)(w & 0xFFFF))));
            m = f8_nanmax(m, fabsf(ElemT<DT>::load((uint16_t)(w >> 16))));
        }
    }
    return m;
}

__device__ __forceinline__ float f8_block_max(float m, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = f8_nanmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return f8_nanmax(f8_nanmax(red[0], red[1]), f8_nanmax(red[2], red[3]));
}

// per-token: one block per row, two passes over the row (second pass hits L1/L2); any K
template <int DT>
__global__ void __launch_bounds__(256) fp8_quant_per_token(const void *__restrict__ xv, uint8_t *__restrict__ xq, float *__restrict__ scale, int64_t K,
                                                           bool vec)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T *xrow = (const T *)xv + row * K;
    uint8_t *orow = xq + row * K;
    float m = 0.0f;
    if (vec) {
        for (int64_t i = threadIdx.x; i < K / VEC; i += 256) m = f8_nanmax(m, f8_vec_absmax<DT>(*(const v4i *)((const char *)xrow + i * 16)));
    } else {
        for (int64_t k = threadIdx.x; k < K; k += 256) m = f8_nanmax(m, fabsf(ElemT<DT>::load(xrow[k])));
    }
    m = f8_block_max(m, red);
    const float s = ElemT<DT>::round(m / 448.0f);  // rowabsmax.div(finfo.max) in x's dtype, then .to(float32)
    if (threadIdx.x == 0) scale[row] = s;
    F8Tok<DT> q{s};
    if (vec) {
        for (int64_t i = threadIdx.x; i < K / VEC; i += 256) {
            uint32_t o[2];
            f8_quant_vec<DT>(*(const v4i *)((const char *)xrow + i * 16), q, o);
            if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + i * 4) = o[0];
            else *(uint2 *)(orow + i * 8) = make_uint2(o[0], o[1]);
        }
    } else {
        for (int64_t k = threadIdx.x; k < K; k += 256) orow[k] = (uint8_t)q(ElemT<DT>::load(xrow[k]));
    }
}

// dynamic per-tensor, pass 1: absmax into a device word (non-negative floats order like their bit patterns)
template <int DT> __global__ void __launch_bounds__(256) fp8_absmax(const void *__restrict__ xv, int64_t n, bool vec, unsigned *__restrict__ amax_bits)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    float m = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        for (int64_t i = t0; i < n / VEC; i += stride) m = f8_nanmax(m, f8_vec_absmax<DT>(*((const v4i *)xv + i)));
        for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) m = f8_nanmax(m, fabsf(ElemT<DT>::load(((const T *)xv)[k])));
    } else {
        for (int64_t k = t0; k < n; k += stride) m = f8_nanmax(m, fabsf(ElemT<DT>::load(((const T *)xv)[k])));
    }
    m = f8_block_max(m, red);
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(m));  // NaN (0x7fc00000) also wins, as torch's aminmax propagates it
}

// per-tensor pass 2 (and the static mode): scale from the device word or the host value
template <int DT>
__global__ void __launch_bounds__(256) fp8_quant_per_tensor(const void *__restrict__ xv, uint8_t *__restrict__ xq, int64_t n, bool vec,
                                                            const unsigned *__restrict__ amax_bits, float host_scale, float *__restrict__ scale_out)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    float s = host_scale;
    if (amax_bits) s = ElemT<DT>::round(__uint_as_float(*amax_bits) / 448.0f);  // amax / finfo.max in x's dtype
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    F8Div<DT> q{s};
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        for (int64_t i = t0; i < n / VEC; i += stride) {
            uint32_t o[2];
            f8_quant_vec<DT>(*((const v4i *)xv + i), q, o);
            if constexpr (DT == ASQ_F32) *((uint32_t *)xq + i) = o[0];
            else *((uint2 *)xq + i) = make_uint2(o[0], o[1]);
        }
        for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) xq[k] = (uint8_t)q(ElemT<DT>::load(((const T *)xv)[k]));
    } else {
        for (int64_t k = t0; k < n; k += stride) xq[k] = (uint8_t)q(ElemT<DT>::load(((const T *)xv)[k]));
    }
}

template <int DT>
int fp8_quantize_dt(const void *x, int mode, float static_scale, uint8_t *xq, float *scale_out, int64_t M, int64_t K, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t n = M * K;
    if (mode == ASQ_FP8_PER_TOKEN) {
        const bool vec = (K % VEC == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)xq) & 7) == 0);
        hipLaunchKernelGGL((fp8_quant_per_token<DT>), dim3((unsigned)M), dim3(256), 0, s, x, xq, scale_out, K, vec);
        return asq_after_launch(s, "asq_quantize_act_fp8(per-token)");
    }
    const bool vec = ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)xq) & 7) == 0);
    int64_t blocks = (n / 8 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    if (mode == ASQ_FP8_PER_TENSOR) {
        unsigned *amax = (unsigned *)(scale_out + 1);  // scale_out[0] = scale, scale_out[1] = scratch word for the absmax
        hipError_t e = hipMemsetAsync(amax, 0, 4, s);
        if (e != hipSuccess) {
            asq_set_error("asq_quantize_act_fp8: hipMemsetAsync: %s", hipGetErrorString(e));
            return (int)e;
        }
        hipLaunchKernelGGL((fp8_absmax<DT>), dim3((unsigned)blocks), dim3(256), 0, s, x, n, vec, amax);
        hipLaunchKernelGGL((fp8_quant_per_tensor<DT>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, n, vec, (const unsigned *)amax, 0.0f, scale_out);
    } else {
        hipLaunchKernelGGL((fp8_quant_per_tensor<DT>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, n, vec, (const unsigned *)nullptr, static_scale,
                           scale_out);
    }
    return asq_after_launch(s, "asq_quantize_act_fp8(per-tensor)");
}

template <int DT, class MMA>
int launch_fp8_linear(const int8_t *xq, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, const float *a_scale_dev, bool a_per_token,
                      float a_scale_host, float w_scale, const float *bias, bool vec_ok, hipStream_t s)
{
    if (bias)
        return launch_gemm(xq, w, M, N, K, EpiFp8<DT, true, MMA>{out, N, a_scale_dev, a_per_token, a_scale_host, w_scale, bias, vec_ok}, s, "asq_linear_fp8");
    return launch_gemm(xq, w, M, N, K, EpiFp8<DT, false, MMA>{out, N, a_scale_dev, a_per_token, a_scale_host, w_scale, bias, vec_ok}, s, "asq_linear_fp8");
}

// unscaled elementwise cast to e5m2 (flat)
template <int DT> __global__ void __launch_bounds__(256) cast_e5m2_kernel(const void *__restrict__ xv, uint8_t *__restrict__ xq, int64_t n, bool vec)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    F8CastE5M2 q;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        for (int64_t i = t0; i < n / VEC; i += stride) {
            uint32_t o[2];
            f8_quant_vec<DT>(*((const v4i *)xv + i), q, o);
            if constexpr (DT == ASQ_F32) *((uint32_t *)xq + i) = o[0];
            else *((uint2 *)xq + i) = make_uint2(o[0], o[1]);
        }
        for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) xq[k] = (uint8_t)q(ElemT<DT>::load(((const T *)xv)[k]));
    } else {
        for (int64_t k = t0; k < n; k += stride) xq[k] = (uint8_t)q(ElemT<DT>::load(((const T *)xv)[k]));
    }
}

}  // namespace asq
using namespace asq;

extern "C" int asq_quantize_act_fp8(const void *x, int x_dtype, int mode, float static_scale, uint8_t *xq, float *scale_out, int64_t M,
                                    int64_t K, void *stream)
{
    ASQ_REQUIRE(M >= 0 && K >= 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_quantize_act_fp8: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_quantize_act_fp8: bad x_dtype %d", x_dtype);
    ASQ_REQUIRE(mode == ASQ_FP8_PER_TOKEN || mode == ASQ_FP8_PER_TENSOR || mode == ASQ_FP8_STATIC, ASQ_ERR_DTYPE, "asq_quantize_act_fp8: bad mode %d", mode);
    ASQ_REQUIRE(mode == ASQ_FP8_STATIC || scale_out != nullptr, ASQ_ERR_NULL, "asq_quantize_act_fp8: dynamic modes need scale_out");
    ASQ_REQUIRE((((uintptr_t)scale_out) & 3) == 0, ASQ_ERR_ALIGN, "asq_quantize_act_fp8: scale_out misaligned");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0 || K == 0) {
        // empty tensors (empty MoE experts): the reference's dynamic per-tensor scale is 16 / 448 (quantization.py:153-158)
        if (mode == ASQ_FP8_PER_TENSOR && scale_out) {
            const float v = x_dtype == ASQ_F32 ? 16.0f / 448.0f : (x_dtype == ASQ_F16 ? 0.03570556640625f : 0.0357666015625f);
            hipError_t e = hipMemcpyAsync(scale_out, &v, 4, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);  // v lives on this stack frame
            if (e != hipSuccess) {
                asq_set_error("asq_quantize_act_fp8: %s", hipGetErrorString(e));
                return (int)e;
            }
        }
        return ASQ_OK;
    }
    ASQ_REQUIRE(x != nullptr && xq != nullptr, ASQ_ERR_NULL, "asq_quantize_act_fp8: NULL x / xq");
    ASQ_REQUIRE(((uintptr_t)x % asq_dtype_size(x_dtype)) == 0, ASQ_ERR_ALIGN, "asq_quantize_act_fp8: x misaligned");
    switch (x_dtype) {
    case ASQ_F32: return fp8_quantize_dt<ASQ_F32>(x, mode, static_scale, xq, scale_out, M, K, s);
    case ASQ_F16: return fp8_quantize_dt<ASQ_F16>(x, mode, static_scale, xq, scale_out, M, K, s);
    default: return fp8_quantize_dt<ASQ_BF16>(x, mode, static_scale, xq, scale_out, M, K, s);
    }
}

extern "C" int asq_cast_e5m2(const void *x, int x_dtype, uint8_t *xq, int64_t n, void *stream)
{
    ASQ_REQUIRE(n >= 0, ASQ_ERR_DIM, "asq_cast_e5m2: bad size");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_cast_e5m2: bad x_dtype %d", x_dtype);
    if (n == 0) return ASQ_OK;
    ASQ_REQUIRE(x != nullptr && xq != nullptr, ASQ_ERR_NULL, "asq_cast_e5m2: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)xq) & 7) == 0);
    int64_t blocks = (n / 8 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    switch (x_dtype) {
    case ASQ_F32: hipLaunchKernelGGL((cast_e5m2_kernel<ASQ_F32>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, n, vec); break;
    case ASQ_F16: hipLaunchKernelGGL((cast_e5m2_kernel<ASQ_F16>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, n, vec); break;
    default: hipLaunchKernelGGL((cast_e5m2_kernel<ASQ_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, n, vec); break;
    }
    return asq_after_launch(s, "asq_cast_e5m2");
}

extern "C" int asq_linear_fp8(const uint8_t *xq, const uint8_t *w, int fp8_format, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                              const float *a_scale_dev, int a_per_token, float a_scale_host, float w_scale, const float *bias, void *stream)
{
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), ASQ_ERR_DIM, "asq_linear_fp8: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;  // linear.py:337-339: empty A -> empty output
    ASQ_REQUIRE(out != nullptr && (K == 0 || (xq != nullptr && w != nullptr)), ASQ_ERR_NULL, "asq_linear_fp8: NULL pointer");
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_fp8: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(fp8_format == ASQ_FP8_E4M3 || fp8_format == ASQ_FP8_E5M2, ASQ_ERR_DTYPE, "asq_linear_fp8: bad fp8_format %d", fp8_format);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0 && ((((uintptr_t)a_scale_dev | (uintptr_t)bias) & 3) == 0), ASQ_ERR_ALIGN,
                "asq_linear_fp8: misaligned pointer");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)bias) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int8_t *xa = (const int8_t *)xq, *wa = (const int8_t *)w;
#define ASQ_F8L(DT_, MMA_) launch_fp8_linear<DT_, MMA_>(xa, wa, out, M, N, K, a_scale_dev, a_per_token != 0, a_scale_host, w_scale, bias, vec_ok, s)
    if (fp8_format == ASQ_FP8_E4M3) {
        switch (out_dtype) {
        case ASQ_F32: return ASQ_F8L(ASQ_F32, MmaFp8);
        case ASQ_F16: return ASQ_F8L(ASQ_F16, MmaFp8);
        default: return ASQ_F8L(ASQ_BF16, MmaFp8);
        }
    }
    switch (out_dtype) {
    case ASQ_F32: return ASQ_F8L(ASQ_F32, MmaBf8);
    case ASQ_F16: return ASQ_F8L(ASQ_F16, MmaBf8);
    default: return ASQ_F8L(ASQ_BF16, MmaBf8);
    }
#undef ASQ_F8L
}
