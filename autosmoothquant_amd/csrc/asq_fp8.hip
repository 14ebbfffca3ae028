// FP8 (OCP e4m3fn) linear path: activation quantisers + GEMM entry point.
// Reference: layers/functional/quantization.py:144-211 (per_tensor / per_token / static
// quantisers), layers/nn/linear.py:336-369 (easy_fp8_gemm), :373-452 (FP8LinearDynamic),
// :503-580 (FP8LinearStatic).  The reference has native_fp8_support = False hard-coded and
// runs a full-precision F.linear on dequantised operands; here the product runs on the fp8
// matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales, fp32 accumulate) and the two
// scales are applied once in the epilogue.
#include "asq_gemm_kernels.h"
#include "asq_silu_core.h"

namespace asq {

// fp32 -> fp8 bytes: the hardware converts (v_cvt_pk_fp8_f32 = OCP e4m3fn, v_cvt_pk_bf8_f32 = e5m2 on gfx950;
// round-to-nearest-even; e4m3 overflow -> NaN, e5m2 overflow -> inf with the default MODE.FP16_OVFL = 0).
// tools/ubench/fp8_cvt_check.hip sweeps ALL 2^32 fp32 inputs against the software encoders that the oracle
// pins to tensor.to(torch.float8_*): 0 mismatches in every class (in range, overflow boundary, inf, NaN).
template <bool E5M2> __device__ __forceinline__ uint32_t f8_pack4(float a, float b, float c, float d)
{
    if constexpr (E5M2) return (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(c, d, __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false), true);
    else return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false), true);
}
template <bool E5M2> __device__ __forceinline__ uint32_t f8_one(float a) { return f8_pack4<E5M2>(a, a, a, a) & 0xFFu; }

__device__ __forceinline__ float clamp448(float v) { return (v != v) ? v : fminf(fmaxf(v, -448.0f), 448.0f); }

// value functors: the fp32 number that is then cast to fp8
struct F8CastE5M2 {
    static constexpr bool kE5M2 = true;
    __device__ __forceinline__ float operator()(float x) const { return x; }
};
template <int DT> struct F8Tok {  // per-token: x / f32 scale promotes to fp32
    static constexpr bool kE5M2 = false;
    float s;
    __device__ __forceinline__ float operator()(float x) const { return clamp448(x / s); }
};
// The 5-op division returns +0 for a -0 numerator; fp8 keeps the sign of zero (0x80), and the divisor is
// positive on the fast path, so the quotient always carries x's sign: copysign restores exactly that case.
struct F8TokFast {  // same quotient by the exact 5-op row division (asq_common.h RowDivisor)
    static constexpr bool kE5M2 = false;
    QRowFast d;
    __device__ __forceinline__ float operator()(float x) const { return clamp448(__builtin_copysignf(d.div(x), x)); }
};
template <int DT> struct F8Div {  // per-tensor (dynamic or static): quotient stays in x's dtype
    static constexpr bool kE5M2 = false;
    float s;
    __device__ __forceinline__ float operator()(float x) const { return clamp448(ElemT<DT>::round(x / s)); }
};

template <int DT> struct F8DivFast {  // F8Div's quotient by the exact 5-op division (all x known finite: dynamic mode with a finite absmax)
    static constexpr bool kE5M2 = false;
    QRowFast d;
    __device__ __forceinline__ float operator()(float x) const { return clamp448(ElemT<DT>::round(__builtin_copysignf(d.div(x), x))); }
};

template <int DT, class Q> __device__ __forceinline__ void f8_quant_vec(const v4i &v, const Q &q, uint32_t (&o)[2])
{
    if constexpr (DT == ASQ_F32) {
        o[0] = f8_pack4<Q::kE5M2>(q(__int_as_float(v[0])), q(__int_as_float(v[1])), q(__int_as_float(v[2])), q(__int_as_float(v[3])));
        o[1] = 0;
    } else {
        float r[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = (uint32_t)v[i];
            r[2 * i] = q(ElemT<DT>::load((uint16_t)(w & 0xFFFF)));
            r[2 * i + 1] = q(ElemT<DT>::load((uint16_t)(w >> 16)));
        }
        o[0] = f8_pack4<Q::kE5M2>(r[0], r[1], r[2], r[3]);
        o[1] = f8_pack4<Q::kE5M2>(r[4], r[5], r[6], r[7]);
    }
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
    uint32_t mb;  // |x| maximum as an fp32 bit pattern (asq_common.h AbsMax: NaN-propagating integer max)
    if (vec) {
        AbsMax<DT> am;
        for (int64_t i = threadIdx.x; i < K / VEC; i += 256) am.add(*(const v4i *)((const char *)xrow + i * 16));
        mb = am.f32bits();
    } else {
        mb = 0;
        for (int64_t k = threadIdx.x; k < K; k += 256) mb = umax32(mb, absbits(ElemT<DT>::load(xrow[k])));
    }
    const float m = block_absmax_256(mb, red);
    const float s = ElemT<DT>::round(m / 448.0f);  // rowabsmax.div(finfo.max) in x's dtype, then .to(float32)
    if (threadIdx.x == 0) scale[row] = s;
    auto emit = [&](auto q) {
        if (vec) {
            for (int64_t i = threadIdx.x; i < K / VEC; i += 256) {
                uint32_t o[2];
                f8_quant_vec<DT>(*(const v4i *)((const char *)xrow + i * 16), q, o);
                if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + i * 4) = o[0];
                else *(uint2 *)(orow + i * 8) = make_uint2(o[0], o[1]);
            }
        } else {
            for (int64_t k = threadIdx.x; k < K; k += 256) orow[k] = (uint8_t)f8_one<false>(q(ElemT<DT>::load(xrow[k])));
        }
    };
    const RowDivisor d(s, m);
    if (d.fast) emit(F8TokFast{QRowFast{d.s, d.y}});
    else emit(F8Tok<DT>{s});
}

// ---- one WAVE per row (round 6; the int8 side's quant_rows_wave, asq_quant.hip): the block kernel above reads every row twice behind two block-wide reductions -- at
// Mixtral's w2 input (8192 x 14336 fp16) that is the slowest non-GEMM launch of the fp8 expert step.  Here a wave owns a row: NV non-temporal 16-byte loads per lane in
// flight (inline asm, waits counted by hand: hipcc does not count asm memory operations), the maximum a wave butterfly, the row quantised from registers -- one HBM read,
// no LDS, no barrier, 4 rows per block.  Same arithmetic, same functors as fp8_quant_per_token: bit-identical outputs.  Rows of up to 64 * 28 vectors (14336 fp16 elements).
__device__ __forceinline__ void f8_load16_nt_async(v4i &dst, const void *p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory"); }
template <int DT, int NV>
__global__ void __launch_bounds__(256) fp8_quant_rows_wave(const void *__restrict__ xv, uint8_t *__restrict__ xq, float *__restrict__ scale, int M, int K)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= M) return;   // (wave-uniform; nothing below synchronises across waves)
    const int nvec = K / VEC;
    const char *xrow = (const char *)xv + row * (int64_t)K * (16 / VEC);
    v4i v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {   // a lane past the end of the row re-reads the row's last vector (a duplicate changes no maximum; its store is skipped)
        const int idx = i * 64 + lane;
        f8_load16_nt_async(v[i], xrow + (int64_t)(idx < nvec ? idx : nvec - 1) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AbsMax<DT> am;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        asm volatile("" : "+v"(v[i]));   // (uses of v[i] stay below the wait)
        am.add(v[i]);
    }
    uint32_t mb = am.f32bits();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mb = umax32(mb, (uint32_t)__shfl_xor((int)mb, off, 64));
    const float m = __uint_as_float(mb);
    const float s = ElemT<DT>::round(m / 448.0f);  // rowabsmax.div(finfo.max) in x's dtype, then .to(float32)
    if (lane == 0) scale[row] = s;
    uint8_t *orow = xq + row * (int64_t)K;
    auto emit = [&](auto q) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            uint32_t o[2];
            f8_quant_vec<DT>(v[i], q, o);
            if (idx < nvec) {
                if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + (int64_t)idx * 4) = o[0];
                else *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(o[0], o[1]);
            }
        }
    };
    const RowDivisor d(s, m);
    if (d.fast) emit(F8TokFast{QRowFast{d.s, d.y}});
    else emit(F8Tok<DT>{s});
}

template <int DT> static bool launch_fp8_rows_wave(const void *x, uint8_t *xq, float *scale, int64_t M, int64_t K, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    static const bool on = [] { const char *e = getenv("ASQ_FP8_ROWS_WAVE"); return !(e && e[0] == '0'); }();   // 0: the block kernel everywhere (A/B)
    const int64_t nvec = K / VEC;
    if (!on || nvec > 64 * 28 || nvec < 1 || M >= (1ll << 31)) return false;
    const int nv = (int)((nvec + 63) / 64);
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
#define ASQ_F8RW(NV) hipLaunchKernelGGL((fp8_quant_rows_wave<DT, NV>), grid, block, 0, s, x, xq, scale, (int)M, (int)K)
    if (nv <= 1) ASQ_F8RW(1);
    else if (nv <= 2) ASQ_F8RW(2);
    else if (nv <= 4) ASQ_F8RW(4);
    else if (nv <= 8) ASQ_F8RW(8);
    else if (nv <= 12) ASQ_F8RW(12);
    else if (nv <= 16) ASQ_F8RW(16);
    else if (nv <= 22) ASQ_F8RW(22);
    else ASQ_F8RW(28);
#undef ASQ_F8RW
    return true;
}

// dynamic per-tensor, pass 1: absmax into a device word (non-negative floats order like their bit patterns,
// NaN patterns above +inf: the integer atomicMax propagates NaN as torch's aminmax does)
template <int DT> __global__ void __launch_bounds__(256) fp8_absmax(const void *__restrict__ xv, int64_t n, bool vec, unsigned *__restrict__ amax_bits)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    uint32_t mb = 0;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        AbsMax<DT> am;
        for (int64_t i = t0; i < n / VEC; i += stride) am.add(*((const v4i *)xv + i));
        mb = am.f32bits();
        for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) mb = umax32(mb, absbits(ElemT<DT>::load(((const T *)xv)[k])));
    } else {
        for (int64_t k = t0; k < n; k += stride) mb = umax32(mb, absbits(ElemT<DT>::load(((const T *)xv)[k])));
    }
    const float m = block_absmax_256(mb, red);
    // 4096 blocks hammering one word serialise in L2 (measured: ~45 us of a 65 us call).  Only record-breakers
    // need the atomic: a possibly stale (lower) peek merely costs a redundant atomicMax
    if (threadIdx.x == 0) {
        const unsigned mine = __float_as_uint(m);
        if (mine > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, mine);
    }
}

// per-tensor pass 2 (and the static mode): scale from the device word or the host value
template <int DT>
__global__ void __launch_bounds__(256) fp8_quant_per_tensor(const void *__restrict__ xv, uint8_t *__restrict__ xq, int64_t n, bool vec,
                                                            const unsigned *__restrict__ amax_bits, float host_scale, float *__restrict__ scale_out)
{
    using T = typename ElemT<DT>::type;
    constexpr int VEC = ElemT<DT>::VEC;
    float s = host_scale, amax = __builtin_inff();  // static mode: nothing is known about x -> plain division
    if (amax_bits) {
        amax = __uint_as_float(*amax_bits);
        s = ElemT<DT>::round(amax / 448.0f);  // amax / finfo.max in x's dtype
    }
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto emit = [&](auto q) {
        if (vec) {
            for (int64_t i = t0; i < n / VEC; i += stride) {
                uint32_t o[2];
                f8_quant_vec<DT>(*((const v4i *)xv + i), q, o);
                if constexpr (DT == ASQ_F32) *((uint32_t *)xq + i) = o[0];
                else *((uint2 *)xq + i) = make_uint2(o[0], o[1]);
            }
            for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) xq[k] = (uint8_t)f8_one<false>(q(ElemT<DT>::load(((const T *)xv)[k])));
        } else {
            for (int64_t k = t0; k < n; k += stride) xq[k] = (uint8_t)f8_one<false>(q(ElemT<DT>::load(((const T *)xv)[k])));
        }
    };
    const RowDivisor d(s, amax);
    if (d.fast) emit(F8DivFast<DT>{QRowFast{d.s, d.y}});
    else emit(F8Div<DT>{s});
}

template <int DT>
int fp8_quantize_dt(const void *x, int mode, float static_scale, uint8_t *xq, float *scale_out, int64_t M, int64_t K, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t n = M * K;
    if (mode == ASQ_FP8_PER_TOKEN) {
        const bool vec = (K % VEC == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)xq) & 7) == 0);
        if (vec && launch_fp8_rows_wave<DT>(x, xq, scale_out, M, K, s)) return asq_after_launch(s, "asq_quantize_act_fp8(per-token)");
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
        return launch_gemm(xq, w, M, N, K, EpiFp8<DT, true, MMA>{out, N, a_scale_dev, bias, nullptr, a_scale_host, w_scale, a_per_token, vec_ok}, s, "asq_linear_fp8");
    return launch_gemm(xq, w, M, N, K, EpiFp8<DT, false, MMA>{out, N, a_scale_dev, bias, nullptr, a_scale_host, w_scale, a_per_token, vec_ok}, s, "asq_linear_fp8");
}

template <int DT>
int launch_fp8_grouped(const int8_t *xq, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, const float *a_scale, const float *w_scale_group,
                       const float *bias, bool vec_ok, const int *goffs, int ngroups, hipStream_t s)
{
    if (bias)
        return launch_gemm(xq, w, M, N, K, EpiFp8<DT, true, MmaFp8>{out, N, a_scale, bias, w_scale_group, 1.0f, 1.0f, true, vec_ok}, s, "asq_linear_fp8_grouped",
                           nullptr, 0, goffs, ngroups);
    return launch_gemm(xq, w, M, N, K, EpiFp8<DT, false, MmaFp8>{out, N, a_scale, bias, w_scale_group, 1.0f, 1.0f, true, vec_ok}, s, "asq_linear_fp8_grouped",
                       nullptr, 0, goffs, ngroups);
}

// ---------------------------------------------------------------------------------
// SiLU(gate) * up -> e4m3 per-token (round 6): the activation between w1 / w3 and w2 of a gated MLP (reference models/mixtral.py:99-101: w2(act_fn(w1(x)) * w3(x)),
// every expert linear an FP8LinearDynamic, layers/nn/linear.py:413-427) fused with w2's own prologue per_token_quantize_fp8 (layers/functional/quantization.py:173-191):
//     a = dt(dt(silu(g)) * u)   (the two ATen ops' roundings, asq_silu_core.h: the statement the int8 kernels share)
//     scale[m] = f32(dt(max_k |a[m,k]| / 448)),   q = e4m3(clamp(f32(a) / scale, +-448))
// One block per row, the row's activation kept in registers between the maximum and the cast: reads 2 x s bytes and writes 1 per element, where the composition
// (silu, mul, quantiser: three passes) reads 5 x s and writes 2 x s + 1 -- at Mixtral's 8192 x 14336 fp16: 0.59 GB instead of 1.53 GB.
// Exact form: bit-identical to oracle/n1.py::silu_mul_quant_fp8_kernel_order; FAST: the hardware transcendentals (the e4m3 code moves by at most one step where a lies
// within an fp16 / bf16 ulp of a rounding boundary).
// ---------------------------------------------------------------------------------
template <int DT, int NV, bool FAST>
__global__ void __launch_bounds__(256) silu_mul_quant_fp8_cached(const void *__restrict__ gv, const void *__restrict__ uv, uint8_t *__restrict__ xq, float *__restrict__ scale, int K)
{
    constexpr int VEC = ElemT<DT>::VEC;
    constexpr bool H = DT == ASQ_F16;   // fp16: the product of two fp16 values is exact in fp32 -> one v_pk_mul_f16 on the raw `up` words, the activation kept as packed halves
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const char *grow = (const char *)gv + row * (int64_t)K * (16 / VEC);
    const char *urow = (const char *)uv + row * (int64_t)K * (16 / VEC);
    const int nvec = K / VEC;
    float a[H ? 1 : NV][H ? 1 : VEC];
    uint32_t ah[H ? NV : 1][4];
    uint32_t amax = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            float g[VEC];
            vec_unpack<DT>(*(const v4i *)(grow + (int64_t)idx * 16), g);
            const v4i uw = *(const v4i *)(urow + (int64_t)idx * 16);
            [[maybe_unused]] float u[VEC];
            if constexpr (!H) vec_unpack<DT>(uw, u);
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                const v2f sl = silu2<FAST>(g[j], g[j + 1]);
                if constexpr (H) {
                    const uint32_t pr = silu_times_up_h(sl, (uint32_t)uw[j / 2]);
                    ah[i][j / 2] = pr;
                    amax = pk_max_u16(amax, pr & 0x7FFF7FFFu);
                } else {
                    const v2f pr = silu_times_up<DT>(sl, u[j], u[j + 1]);
                    a[i][j] = pr[0];
                    a[i][j + 1] = pr[1];
                    amax = umax32(amax, umax32(absbits(a[i][j]), absbits(a[i][j + 1])));
                }
            }
        }
    }
    if constexpr (H) amax = __float_as_uint(ElemT<DT>::load((uint16_t)umax32(amax & 0xFFFFu, amax >> 16)));  // widening keeps the order, NaN stays NaN
    const float m = block_absmax_256(amax, red);
    const float s = ElemT<DT>::round(m / 448.0f);  // rowabsmax.div(finfo.max) in the activation dtype, then .to(float32)
    if (threadIdx.x == 0) scale[row] = s;
    uint8_t *orow = xq + row * (int64_t)K;
    auto emit = [&](auto q) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                float r[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j += 2) {
                    if constexpr (H) {
                        const v2h hh = __builtin_bit_cast(v2h, ah[i][j / 2]);
                        r[j] = q((float)hh[0]);
                        r[j + 1] = q((float)hh[1]);
                    } else {
                        r[j] = q(a[i][j]);
                        r[j + 1] = q(a[i][j + 1]);
                    }
                }
                if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + (int64_t)idx * 4) = f8_pack4<false>(r[0], r[1], r[2], r[3]);
                else *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(f8_pack4<false>(r[0], r[1], r[2], r[3]), f8_pack4<false>(r[4], r[5], r[6], r[7]));
            }
        }
    };
    const RowDivisor d(s, m);
    if (d.fast) emit(F8TokFast{QRowFast{d.s, d.y}});
    else emit(F8Tok<DT>{s});
}

template <int DT, bool FAST> int launch_silu_mul_quant_fp8(const void *g, const void *u, uint8_t *xq, float *scale, int64_t M, int64_t K, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t nvec = K / VEC;
    dim3 grid((unsigned)M), block(256);
#define ASQ_SM8(NV) hipLaunchKernelGGL((silu_mul_quant_fp8_cached<DT, NV, FAST>), grid, block, 0, s, g, u, xq, scale, (int)K)
    if (nvec <= 256 * 2) ASQ_SM8(2);
    else if (nvec <= 256 * 4) ASQ_SM8(4);
    else if (nvec <= 256 * 6) ASQ_SM8(6);
    else ASQ_SM8(8);
#undef ASQ_SM8
    return asq_after_launch(s, "asq_silu_mul_quantize_fp8");
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
        for (int64_t k = (n / VEC) * VEC + t0; k < n; k += stride) xq[k] = (uint8_t)f8_one<true>(q(ElemT<DT>::load(((const T *)xv)[k])));
    } else {
        for (int64_t k = t0; k < n; k += stride) xq[k] = (uint8_t)f8_one<true>(q(ElemT<DT>::load(((const T *)xv)[k])));
    }
}

// ---------------------------------------------------------------------------------
// MX (OCP Microscaling) e4m3 with REAL block scales -- an opt-in beyond the reference: 32 consecutive k share one E8M0 scale
// 2^(floor(log2 amax) - 8), elements are e4m3 of x / scale (saturating), and the product runs on v_mfma_scale_f32_32x32x64_f8f6f4 with the two
// scale bytes as operands.  Accuracy-checked against oracle/mx.py; on bench-like activations it is LESS accurate than the reference's per-token
// e4m3, so nothing dispatches to it by default and the kernel below is the plain one-wave-per-32x32-tile form, not a tuned one.
// ---------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256) mx_quant_e4m3(const void *__restrict__ xv, uint8_t *__restrict__ xq, uint8_t *__restrict__ xs, int64_t nblocks)
{
    constexpr int VEC = ElemT<DT>::VEC, NL = 32 / VEC;  // 16-byte loads per block of 32 elements
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * 256) {
        float f[32];
        const v4i *src = (const v4i *)xv + b * NL;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const v4i v = src[i];
            if constexpr (DT == ASQ_F32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[4 * i + j] = __int_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[8 * i + 2 * j] = ElemT<DT>::load((uint16_t)((uint32_t)v[j] & 0xFFFF));
                    f[8 * i + 2 * j + 1] = ElemT<DT>::load((uint16_t)((uint32_t)v[j] >> 16));
                }
            }
        }
        uint32_t am = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) am = umax32(am, absbits(f[i]));
        // shared exponent: floor(log2 amax) - emax(e4m3 = 8), as an E8M0 byte (bias 127); zero / subnormal maxima -> 2^-127 .. handled by the clamp;
        // a NaN / inf in the block makes the scale NaN (0xFF), as the OCP spec asks
        int e = (int)(am >> 23) - 8;
        e = e < 0 ? 0 : (e > 254 ? 254 : e);
        const uint8_t sb = (am >= 0x7F800000u) ? 0xFF : (uint8_t)e;
        const float inv = __uint_as_float((uint32_t)(254 - e) << 23);  // 2^-(e-127), exact (e = 254 -> 2^-127 as the subnormal 0x00400000)
        const float invs = e == 254 ? __uint_as_float(0x00400000u) : inv;
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = f8_pack4<false>(clamp448(f[4 * i] * invs), clamp448(f[4 * i + 1] * invs), clamp448(f[4 * i + 2] * invs), clamp448(f[4 * i + 3] * invs));
        uint4 *dst = (uint4 *)(xq + b * 32);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        xs[b] = sb;
    }
}

// out[M,N] = sum over blocks of (e4m3 dot) * 2^(sx + sw - 254) (+ bias); x [M,K] e4m3 + xs [M,K/32], w [N,K] e4m3 + ws [N,K/32].  One wave per
// 32(m) x 32(n) tile, 4 waves per block (64 x 64); a lane (r = lane % 32, h = lane / 32) supplies 16 + 16 bytes of row r of both operands and the scale
// bytes of block h.
template <int DT>
__global__ void __launch_bounds__(256) mx_gemm_e4m3(const uint8_t *__restrict__ x, const uint8_t *__restrict__ xs, const uint8_t *__restrict__ w,
                                                    const uint8_t *__restrict__ ws, void *__restrict__ out, const float *__restrict__ bias, int64_t M, int64_t N,
                                                    int64_t K)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * 64 + (wave >> 1) * 32, n0 = (int64_t)blockIdx.x * 64 + (wave & 1) * 32;
    if (m0 >= M || n0 >= N) return;  // wave-uniform
    const int64_t mr = (m0 + r) < M ? (m0 + r) : (M - 1), nr = (n0 + r) < N ? (n0 + r) : (N - 1);
    const int64_t kb = K / 32;
    // operand layout of the K = 64 instruction: a lane's first 16 bytes are k = 16h .. 16h+15 (MX block 0 of the pair), its second 16 bytes
    // k = 32 + 16h .. (block 1); the scale byte of lane (r, h) is the one of block h of row r
    const uint8_t *xp = x + mr * K + 16 * h, *wp = w + nr * K + 16 * h, *xsp = xs + mr * kb + h, *wsp = ws + nr * kb + h;
    v16f acc = {0};
    for (int64_t k0 = 0; k0 < K; k0 += 64) {
        const v4i a0 = *(const v4i *)(wp + k0), a1 = *(const v4i *)(wp + k0 + 32), b0 = *(const v4i *)(xp + k0), b1 = *(const v4i *)(xp + k0 + 32);
        const int sa = wsp[k0 / 32], sb = xsp[k0 / 32];
        const v8i A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const v8i B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
    }
    // W is the A operand: the lane owns token m0 + r and, per group g, the 4 channels n0 + 8g + 4h .. +3
    const int64_t m = m0 + r;
    if (m >= M) return;
    using E = ElemT<DT>;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t n = n0 + 8 * g + 4 * h + i;
            if (n < N) {
                float v = acc[4 * g + i];
                if (bias) v = __fadd_rn(v, bias[n]);
                if constexpr (DT == ASQ_F32) ((float *)out)[m * N + n] = v;
                else ((uint16_t *)out)[m * N + n] = E::store(v);
            }
        }
}

template <int DT, bool HAS_BIAS>
static int launch_mx_tiled_one(const int8_t *xq, const uint8_t *xs, const int8_t *wq, const uint8_t *ws, void *out, int64_t M, int64_t N, int64_t K, const float *bias,
                               hipStream_t s)
{
    using Epi = EpiFp8<DT, HAS_BIAS, MmaFp8>;
    const size_t vbytes = DT == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)bias) & 15) == 0);
    Epi epi{out, N, nullptr, bias, nullptr, 1.0f, 1.0f, false, vec_ok};  // unit epilogue scales: the block scales are inside the product
    auto kfn = gemm_i8_p8q<Epi, true>;
    hipError_t e = ensure_dynamic_lds((const void *)kfn, P8Q_MX_LDS_BYTES);
    if (e != hipSuccess) {
        asq_set_error("asq_linear_mxfp8: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128;
    ASQ_REQUIRE(tm * tn < (1ll << 24), ASQ_ERR_DIM, "asq_linear_mxfp8: too many tiles");
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tm * tn)), dim3(512), P8Q_MX_LDS_BYTES, s, xq, wq, M, N, K, (int)tm, (int)tn, 1, epi, xs, ws);
    return asq_after_launch(s, "asq_linear_mxfp8");
}
static int launch_mx_tiled(int out_dtype, const int8_t *xq, const uint8_t *xs, const int8_t *wq, const uint8_t *ws, void *out, int64_t M, int64_t N, int64_t K,
                           const float *bias, hipStream_t s)
{
    if (((((uintptr_t)xs) | ((uintptr_t)ws)) & 15) != 0) return ASQ_ERR_ALIGN;
#define ASQ_MX_CASE(DT) (bias ? launch_mx_tiled_one<DT, true>(xq, xs, wq, ws, out, M, N, K, bias, s) : launch_mx_tiled_one<DT, false>(xq, xs, wq, ws, out, M, N, K, bias, s))
    switch (out_dtype) {
    case ASQ_F32: return ASQ_MX_CASE(ASQ_F32);
    case ASQ_F16: return ASQ_MX_CASE(ASQ_F16);
    default: return ASQ_MX_CASE(ASQ_BF16);
    }
#undef ASQ_MX_CASE
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

extern "C" int asq_silu_mul_quantize_fp8(const void *gate, const void *up, int x_dtype, int flags, uint8_t *xq, float *scale, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_silu_mul_quantize_fp8");
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_silu_mul_quantize_fp8: flags is 0 or ASQ_SILU_FAST, got %d", flags);
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_silu_mul_quantize_fp8: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_silu_mul_quantize_fp8: bad x_dtype %d", x_dtype);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(gate && up && xq && scale, ASQ_ERR_NULL, "asq_silu_mul_quantize_fp8: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 8, ASQ_ERR_DIM, "asq_silu_mul_quantize_fp8: K must be a multiple of %d and <= %d", vec, 256 * 8 * vec);
    ASQ_REQUIRE(((((uintptr_t)gate) | ((uintptr_t)up)) & 15) == 0 && (((uintptr_t)xq) & 7) == 0 && (((uintptr_t)scale) & 3) == 0, ASQ_ERR_ALIGN,
                "asq_silu_mul_quantize_fp8: gate / up must be 16-B aligned, xq 8-B, scale 4-B");
    hipStream_t s = (hipStream_t)stream;
    const bool fast = (flags & ASQ_SILU_FAST) != 0;
#define ASQ_SM8_DT(DT_) (fast ? launch_silu_mul_quant_fp8<DT_, true>(gate, up, xq, scale, M, K, s) : launch_silu_mul_quant_fp8<DT_, false>(gate, up, xq, scale, M, K, s))
    switch (x_dtype) {
    case ASQ_F32: return ASQ_SM8_DT(ASQ_F32);
    case ASQ_F16: return ASQ_SM8_DT(ASQ_F16);
    default: return ASQ_SM8_DT(ASQ_BF16);
    }
#undef ASQ_SM8_DT
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

extern "C" int asq_linear_fp8_grouped(const uint8_t *xq, const uint8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M,
                                      int64_t N, int64_t K, const float *a_scale, const float *w_scale_group, const float *bias, void *stream)
{
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), ASQ_ERR_DIM, "asq_linear_fp8_grouped: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(out != nullptr && xq != nullptr && w != nullptr && group_offsets != nullptr && a_scale != nullptr && w_scale_group != nullptr &&
                    ngroups > 0 && ngroups <= 4096,
                ASQ_ERR_NULL, "asq_linear_fp8_grouped: need xq, w, out, group_offsets, a_scale, w_scale_group and 1 <= ngroups <= 4096");
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_fp8_grouped: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0 &&
                    ((((uintptr_t)a_scale | (uintptr_t)w_scale_group | (uintptr_t)bias | (uintptr_t)group_offsets) & 3) == 0),
                ASQ_ERR_ALIGN, "asq_linear_fp8_grouped: misaligned pointer");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)bias) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int8_t *xa = (const int8_t *)xq, *wa = (const int8_t *)w;
    switch (out_dtype) {
    case ASQ_F32: return launch_fp8_grouped<ASQ_F32>(xa, wa, out, M, N, K, a_scale, w_scale_group, bias, vec_ok, group_offsets, ngroups, s);
    case ASQ_F16: return launch_fp8_grouped<ASQ_F16>(xa, wa, out, M, N, K, a_scale, w_scale_group, bias, vec_ok, group_offsets, ngroups, s);
    default: return launch_fp8_grouped<ASQ_BF16>(xa, wa, out, M, N, K, a_scale, w_scale_group, bias, vec_ok, group_offsets, ngroups, s);
    }
}

extern "C" int asq_quantize_mxfp8(const void *x, int x_dtype, uint8_t *xq, uint8_t *scales, int64_t M, int64_t K, void *stream)
{
    ASQ_REQUIRE(M >= 0 && K >= 0, ASQ_ERR_DIM, "asq_quantize_mxfp8: bad dims");
    if (M == 0 || K == 0) return ASQ_OK;
    ASQ_REQUIRE(x && xq && scales, ASQ_ERR_NULL, "asq_quantize_mxfp8: null pointer");
    ASQ_REQUIRE(K % 32 == 0, ASQ_ERR_DIM, "asq_quantize_mxfp8: K = %lld is not a multiple of the MX block (32)", (long long)K);
    ASQ_REQUIRE(((((uintptr_t)x) | ((uintptr_t)xq)) & 15) == 0, ASQ_ERR_ALIGN, "asq_quantize_mxfp8: x / xq must be 16-byte aligned");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_quantize_mxfp8: bad dtype %d", x_dtype);
    const int64_t nb = M * (K / 32);
    int64_t blocks = (nb + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipStream_t s = (hipStream_t)stream;
    switch (x_dtype) {
    case ASQ_F32: hipLaunchKernelGGL((mx_quant_e4m3<ASQ_F32>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, scales, nb); break;
    case ASQ_F16: hipLaunchKernelGGL((mx_quant_e4m3<ASQ_F16>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, scales, nb); break;
    default: hipLaunchKernelGGL((mx_quant_e4m3<ASQ_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, scales, nb); break;
    }
    return asq_after_launch(s, "asq_quantize_mxfp8");
}

extern "C" int asq_linear_mxfp8(const uint8_t *xq, const uint8_t *x_scales, const uint8_t *wq, const uint8_t *w_scales, void *out, int out_dtype, int64_t M,
                                int64_t N, int64_t K, const float *bias, void *stream)
{
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, ASQ_ERR_DIM, "asq_linear_mxfp8: bad dims");
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(xq && x_scales && wq && w_scales && out, ASQ_ERR_NULL, "asq_linear_mxfp8: null pointer");
    ASQ_REQUIRE(K > 0 && K % 64 == 0, ASQ_ERR_DIM, "asq_linear_mxfp8: K = %lld must be a positive multiple of 64", (long long)K);
    ASQ_REQUIRE(((((uintptr_t)xq) | ((uintptr_t)wq)) & 15) == 0 && (((uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_mxfp8: misaligned operand");
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_mxfp8: bad out_dtype %d", out_dtype);
    hipStream_t s = (hipStream_t)stream;
    if (K % 512 == 0 && getenv("ASQ_MX_SIMPLE") == nullptr) {  // the tiled kernel (gemm_i8_p8q with MX = true): scale rows must be 16-byte multiples
        const int rc = launch_mx_tiled(out_dtype, (const int8_t *)xq, x_scales, (const int8_t *)wq, w_scales, out, M, N, K, bias, s);
        if (rc != ASQ_ERR_ALIGN) return rc;  // (misaligned scales / output: the plain kernel below)
    }
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    ASQ_REQUIRE(grid.y < 65536, ASQ_ERR_DIM, "asq_linear_mxfp8: M too large");
    switch (out_dtype) {
    case ASQ_F32: hipLaunchKernelGGL((mx_gemm_e4m3<ASQ_F32>), grid, dim3(256), 0, s, xq, x_scales, wq, w_scales, out, bias, M, N, K); break;
    case ASQ_F16: hipLaunchKernelGGL((mx_gemm_e4m3<ASQ_F16>), grid, dim3(256), 0, s, xq, x_scales, wq, w_scales, out, bias, M, N, K); break;
    default: hipLaunchKernelGGL((mx_gemm_e4m3<ASQ_BF16>), grid, dim3(256), 0, s, xq, x_scales, wq, w_scales, out, bias, M, N, K); break;
    }
    return asq_after_launch(s, "asq_linear_mxfp8");
}
