// Activation quantisers: the GEMM prologue of autosmoothquant/layers/nn/linear.py
// (reference :88-96, :164-174, :283-292), one memory-bound pass each.
//
// HBM traffic per activation element: sizeof(x) read + 1 B written (+4 B per row for the
// per-token scale).  The reference spends >= 37 B/element on the same work (abs, max, div,
// cast, div, round, clamp, cast as separate ATen launches).
#include "asq_common.h"
#include "asq_quant_core.h"
#include "asq_silu_core.h"
#include <type_traits>
#include <algorithm>
#include <cstdlib>

namespace asq {

// ---- per-token: one 256-thread block per row, row cached in registers ---------------
// K % VEC == 0, K <= 256*VEC*NV, x and xq rows 16-/VEC-byte aligned.
template <int DT, int NV>
__global__ void __launch_bounds__(256) quant_per_token_cached(const void *__restrict__ xv, int8_t *__restrict__ xq,
                                                              float *__restrict__ s_row, int K)
{
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const char *xrow = (const char *)xv + row * (int64_t)K * (16 / VEC);
    const int nvec = K / VEC;
    v4i v[NV];
    AbsMax<DT> am;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            v[i] = *(const v4i *)(xrow + (int64_t)idx * 16);
            am.add(v[i]);
        }
    }
    const float m = block_absmax_256(am.f32bits(), red);
    // quant_scale = absmax.div(127.0) in x's dtype, widened to fp32 (linear.py:89-91)
    const float qs = ElemT<DT>::round(m / 127.0f);
    if (threadIdx.x == 0) s_row[row] = qs;
    int8_t *orow = xq + row * (int64_t)K;
    auto emit = [&](auto q) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                uint32_t o[2];
                quant_vec<DT>(v[i], q, o);
                if constexpr (DT == ASQ_F32) {
                    *(uint32_t *)(orow + (int64_t)idx * 4) = o[0];
                } else {
                    *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(o[0], o[1]);
                }
            }
        }
    };
    const RowDivisor d(qs, m);
    if (d.fast) emit(QRowFast{d.s, d.y});
    else emit(QDivF32<DT>{qs});
}

// ---- per-token, any K / alignment: two passes over the row (second pass hits L1/L2) ----
template <int DT>
__global__ void __launch_bounds__(256) quant_per_token_generic(const void *__restrict__ xv, int8_t *__restrict__ xq,
                                                               float *__restrict__ s_row, int64_t K)
{
    using T = typename ElemT<DT>::type;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T *xrow = (const T *)xv + row * K;
    uint32_t mb = 0;
    for (int64_t k = threadIdx.x; k < K; k += 256) mb = umax32(mb, absbits(ElemT<DT>::load(xrow[k])));
    const float m = block_absmax_256(mb, red);
    const float qs = ElemT<DT>::round(m / 127.0f);
    if (threadIdx.x == 0) s_row[row] = qs;
    int8_t *orow = xq + row * K;
    for (int64_t k = threadIdx.x; k < K; k += 256) orow[k] = (int8_t)quant_i8(ElemT<DT>::load(xrow[k]) / qs);
}

// ---- per-tensor (round / div): flat elementwise, 16 elements per thread ------------
template <int DT, class Q>
__global__ void __launch_bounds__(256) quant_flat_vec(const void *__restrict__ xv, int8_t *__restrict__ xq, int64_t nchunk, Q q)
{
    constexpr int VEC = ElemT<DT>::VEC;
    constexpr int NL = 16 / VEC;  // 16-byte loads per 16 elements
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * 256) {
        const v4i *src = (const v4i *)xv + c * NL;
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            v4i v = src[i];
            uint32_t t[2];
            quant_vec<DT>(v, q, t);
            if constexpr (DT == ASQ_F32) {
                o[i] = t[0];
            } else {
                o[2 * i] = t[0];
                o[2 * i + 1] = t[1];
            }
        }
        *((uint4 *)xq + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <int DT, class Q>
__global__ void __launch_bounds__(256) quant_flat_scalar(const void *__restrict__ xv, int8_t *__restrict__ xq, int64_t begin, int64_t n, Q q)
{
    using T = typename ElemT<DT>::type;
    const T *x = (const T *)xv;
    for (int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        xq[i] = (int8_t)q(ElemT<DT>::load(x[i]));
}

template <int DT, class Q> int launch_flat(const void *x, int8_t *xq, int64_t n, Q q, hipStream_t s)
{
    const bool aligned = (((uintptr_t)x | (uintptr_t)xq) & 15) == 0;
    int64_t done = 0;
    if (aligned && n >= 16) {
        const int64_t nchunk = n / 16;
        int64_t blocks = (nchunk + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((quant_flat_vec<DT, Q>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, nchunk, q);
        done = nchunk * 16;
    }
    if (done < n) {
        int64_t rem = n - done;
        int64_t blocks = (rem + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((quant_flat_scalar<DT, Q>), dim3((unsigned)blocks), dim3(256), 0, s, x, xq, done, n, q);
    }
    return asq_after_launch(s, "asq_quantize_act(per-tensor)");
}

template <int DT> bool launch_per_token_wave(const void *x, int8_t *xq, float *s_row, int64_t M, int64_t K, hipStream_t s);   // (one wave per row: defined below)
template <int DT> int launch_per_token(const void *x, int8_t *xq, float *s_row, int64_t M, int64_t K, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const bool vec_ok = (K % VEC == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)xq) & (VEC - 1)) == 0);
    if (vec_ok && K > 0 && launch_per_token_wave<DT>(x, xq, s_row, M, K, s)) return asq_after_launch(s, "asq_quantize_act(per-token)");
    const int64_t nvec = K / VEC;
    dim3 grid((unsigned)M), block(256);
#define ASQ_PT(NV)                                                                                          \
    hipLaunchKernelGGL((quant_per_token_cached<DT, NV>), grid, block, 0, s, x, xq, s_row, (int)K)
    if (vec_ok && nvec <= 256 * 1) ASQ_PT(1);
    else if (vec_ok && nvec <= 256 * 2) ASQ_PT(2);
    else if (vec_ok && nvec <= 256 * 4) ASQ_PT(4);
    else if (vec_ok && nvec <= 256 * 6) ASQ_PT(6);
    else if (vec_ok && nvec <= 256 * 8) ASQ_PT(8);
    else if (vec_ok && nvec <= 256 * 12) ASQ_PT(12);
    else if (vec_ok && nvec <= 256 * 20) ASQ_PT(20);
    else hipLaunchKernelGGL((quant_per_token_generic<DT>), grid, block, 0, s, x, xq, s_row, K);
#undef ASQ_PT
    return asq_after_launch(s, "asq_quantize_act(per-token)");
}

template <int DT> int quantize_dt(const void *x, int mode, float quant_scale, int8_t *xq, float *s_row, int64_t M, int64_t K,
                                  hipStream_t s)
{
    switch (mode) {
    case ASQ_ACT_ROUND: return launch_flat<DT>(x, xq, M * K, QRound<DT>{}, s);
    case ASQ_ACT_DIV:
        if (quant_scale > 0x1p-60f && quant_scale < 0x1p60f) return launch_flat<DT>(x, xq, M * K, QDivFast<DT>{quant_scale, 1.0f / quant_scale}, s);
        return launch_flat<DT>(x, xq, M * K, QDiv<DT>{quant_scale}, s);
    default: return launch_per_token<DT>(x, xq, s_row, M, K, s);
    }
}

// ---------------------------------------------------------------------------------
// OFFSET operand images (asq_gemm_kernels.h: OffsetArgs; include/asq_hip.h).  The quantisers below emit x'[m,k] = xq[m,k] + cx[m] and the row vector
// {cx[m], sum_k x'[m,k]}; asq_weight_offset_image does the same once per weight.  The offsets come from the row's own extremes, so nothing clamps:
//     activations  cx = +C if max_k xq <= 127 - C, else -C if min_k xq >= -128 + C, else 0         (C = 3: the bulk of a SmoothQuant-style row is within +-3)
//     weights      cw = min(CW, 127 - max_k w)                                                     (CW = 64)
// (profiles/r4_operand_offsets.md: what the matrix cores sustain for which offsets; oracle/offsets.py restates both rules.)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}
// running max / min / sum of packed int8 words, carried on the bytes biased by +128 (0..255) as pairs of 16-bit lanes: no unpacking
struct RowStats {
    uint32_t mx = 0, mn = 0x00FF00FFu, sum = 0;   // (sum: two 16-bit partial sums of biased bytes -- at most 2 * 160 words * 255 per thread)
    int n = 0;                                    // bytes added
    __device__ __forceinline__ void add(uint32_t packed)
    {
        const uint32_t u = packed ^ 0x80808080u, a = u & 0x00FF00FFu, b = (u >> 8) & 0x00FF00FFu;
        mx = pk_max_u16(mx, pk_max_u16(a, b));
        mn = pk_min_u16(mn, pk_min_u16(a, b));
        sum += a + b;
        n += 4;
    }
    // branch-free form for a lane whose vector may be a DUPLICATE of a valid one (max / min do not care; sum and count must skip it)
    __device__ __forceinline__ void add(uint32_t packed, bool valid)
    {
        const uint32_t u = packed ^ 0x80808080u, a = u & 0x00FF00FFu, b = (u >> 8) & 0x00FF00FFu;
        mx = pk_max_u16(mx, pk_max_u16(a, b));
        mn = pk_min_u16(mn, pk_min_u16(a, b));
        sum += valid ? a + b : 0u;
        n += valid ? 4 : 0;
    }
};
// block-wide (256 threads) max / min / sum of the quantised row; red: 12 ints of shared memory
__device__ __forceinline__ void block_row_stats(const RowStats &st, int *red, int &rmax, int &rmin, int &rsum)
{
    int mx = (int)umax32(st.mx & 0xFFFFu, st.mx >> 16) - 128, mn = (int)(((st.mn & 0xFFFFu) < (st.mn >> 16)) ? (st.mn & 0xFFFFu) : (st.mn >> 16)) - 128;
    int sm = (int)((st.sum & 0xFFFFu) + (st.sum >> 16)) - 128 * st.n;
    if (st.n == 0) { mx = -128; mn = 127; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int omx = __shfl_xor(mx, off, 64), omn = __shfl_xor(mn, off, 64);
        mx = mx > omx ? mx : omx;
        mn = mn < omn ? mn : omn;
        sm += __shfl_xor(sm, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // (red may still be read by a previous reduction)
    if ((threadIdx.x & 63) == 0) { red[wave] = mx; red[4 + wave] = mn; red[8 + wave] = sm; }
    __syncthreads();
    const int a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    const int c = red[4] < red[5] ? red[4] : red[5], d = red[6] < red[7] ? red[6] : red[7];
    rmax = a > b ? a : b;
    rmin = c < d ? c : d;
    rsum = (red[8] + red[9]) + (red[10] + red[11]);
}
__device__ __forceinline__ int pick_row_offset(int rmax, int rmin, int C)
{
    return rmax <= 127 - C ? C : (rmin >= -128 + C ? -C : 0);
}
// four packed int8 + the same small constant each, no carries between the bytes (the caller guarantees every sum stays in [-128, 127])
__device__ __forceinline__ uint32_t pk_add_i8(uint32_t a, uint32_t c4)
{
    return ((a & 0x7F7F7F7Fu) + (c4 & 0x7F7F7F7Fu)) ^ ((a ^ c4) & 0x80808080u);
}

// any of the three activation quantisers with row offsets: one 256-thread block per row, the row in registers (as quant_per_token_cached), the
// quantised row kept PACKED in registers between the statistics and the store.  K % VEC == 0, K <= 256 * VEC * NV, 16-byte aligned rows.
template <int DT, int NV, class Q, bool PER_TOKEN>
__global__ void __launch_bounds__(256) quant_rows_off(const void *__restrict__ xv, int8_t *__restrict__ xq, float *__restrict__ s_row,
                                                      int32_t *__restrict__ row_off, int K, Q q_in, int C)
{
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    __shared__ int redi[12];
    const int64_t row = blockIdx.x;
    const char *xrow = (const char *)xv + row * (int64_t)K * (16 / VEC);
    const int nvec = K / VEC;
    v4i v[NV];
    [[maybe_unused]] AbsMax<DT> am;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            v[i] = *(const v4i *)(xrow + (int64_t)idx * 16);
            if constexpr (PER_TOKEN) am.add(v[i]);
        }
    }
    uint32_t o[NV][2];
    RowStats st;
    auto emit = [&](auto q) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                quant_vec<DT>(v[i], q, o[i]);
                st.add(o[i][0]);
                if constexpr (DT != ASQ_F32) st.add(o[i][1]);
            }
        }
    };
    if constexpr (PER_TOKEN) {
        const float m = block_absmax_256(am.f32bits(), red);
        const float qs = ElemT<DT>::round(m / 127.0f);
        if (threadIdx.x == 0) s_row[row] = qs;
        const RowDivisor d(qs, m);
        if (d.fast) emit(QRowFast{d.s, d.y});
        else emit(QDivF32<DT>{qs});
    } else {
        emit(q_in);
    }
    int rmax, rmin, rsum;
    block_row_stats(st, redi, rmax, rmin, rsum);
    const int cx = pick_row_offset(rmax, rmin, C);
    if (threadIdx.x == 0) *(v2i *)(row_off + 2 * row) = (v2i){cx, rsum + cx * K};
    const uint32_t c4 = (uint32_t)(cx & 0xFF) * 0x01010101u;
    int8_t *orow = xq + row * (int64_t)K;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            if constexpr (DT == ASQ_F32) {
                *(uint32_t *)(orow + (int64_t)idx * 4) = pk_add_i8(o[i][0], c4);
            } else {
                *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(pk_add_i8(o[i][0], c4), pk_add_i8(o[i][1], c4));
            }
        }
    }
}

// ---- one WAVE per row (round 4): the row-wise quantisers above spend their time in two block-wide reductions (LDS + __syncthreads) on 4096 tiny blocks --
// 15.1 / 17.4 us for 4096 x 4096 fp16 (3.3 / 2.9 TB/s) against 12.5 us for the flat per-tensor kernel.  Here a wave owns a row: NV 16-byte loads per lane in flight
// (non-temporal: the activation is read once), every reduction a wave butterfly, no LDS, no barrier; 4 rows per 256-thread block.  K <= 64 * VEC * NV.
// OFF = false is the plain per-token quantiser (same arithmetic as quant_per_token_cached), OFF = true emits the offset image + row_off.
// The loads are inline asm and the waits are counted BY HAND (hipcc does not count asm memory operations): with C loads under `if (idx < nvec)` hipcc waited
// for each load at its branch's join -- ONE load in flight per wave instead of NV (4096 x 4096 fp16 per-token: 13.5 -> 10.7 us, its image 15.4 -> 12.7 us).
__device__ __forceinline__ void load16_nt_async(v4i &dst, const void *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void pin_vgprs(v4i &x) { asm volatile("" : "+v"(x)); }   // (uses of x stay below this point)
template <int N> __device__ __forceinline__ void wait_vm()   // at most N vector memory operations still in flight (they retire in issue order)
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// (Tried and dropped: two rows per wave with the second row's loads in flight behind the first row's arithmetic -- 4 .. 10 % SLOWER than one row per wave at
// every shape, profiles/r4_quantiser_rows.txt: the chip-wide bytes in flight matter more than overlapping a wave's own arithmetic.)
template <int DT, int NV, class Q, bool PER_TOKEN, bool OFF>
__global__ void __launch_bounds__(256) quant_rows_wave(const void *__restrict__ xv, int8_t *__restrict__ xq, float *__restrict__ s_row,
                                                       int32_t *__restrict__ row_off, int M, int K, Q q_in, int C)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= M) return;   // (wave-uniform; nothing below synchronises across waves)
    const int nvec = K / VEC;
    // straight-line: a lane past the end of the row re-reads the row's last vector (a duplicate changes no maximum or minimum; sums and stores skip it)
    auto load = [&](v4i (&v)[NV], int64_t r) {
        const char *xrow = (const char *)xv + r * (int64_t)K * (16 / VEC);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            load16_nt_async(v[i], xrow + (int64_t)(idx < nvec ? idx : nvec - 1) * 16);
        }
    };
    auto finish = [&](v4i (&v)[NV], int64_t r) {
        [[maybe_unused]] AbsMax<DT> am;
        uint32_t o[NV][2];
        [[maybe_unused]] RowStats st;
        auto arrive = [&](auto ic) {   // v[i] is in its registers from here on
            constexpr int i = decltype(ic)::value;
            wait_vm<NV - 1 - i>();   // (operations retire in issue order)
            pin_vgprs(v[i]);
        };
        auto emit = [&](auto q, auto waits) {
            static_for<NV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (decltype(waits)::value) arrive(ic);
                quant_vec<DT>(v[i], q, o[i]);
                if constexpr (OFF) {
                    const bool valid = i * 64 + lane < nvec;
                    st.add(o[i][0], valid);
                    if constexpr (DT != ASQ_F32) st.add(o[i][1], valid);
                }
            });
        };
        if constexpr (PER_TOKEN) {
            static_for<NV>([&](auto ic) { arrive(ic); am.add(v[decltype(ic)::value]); });
            uint32_t mb = am.f32bits();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mb = umax32(mb, (uint32_t)__shfl_xor((int)mb, off, 64));
            const float m = __uint_as_float(mb);
            const float qs = ElemT<DT>::round(m / 127.0f);
            if (lane == 0) s_row[r] = qs;
            const RowDivisor d(qs, m);
            if (d.fast) emit(QRowFast{d.s, d.y}, std::false_type{});
            else emit(QDivF32<DT>{qs}, std::false_type{});
        } else {
            emit(q_in, std::true_type{});
        }
        uint32_t c4 = 0;
        if constexpr (OFF) {
            int mx = (int)umax32(st.mx & 0xFFFFu, st.mx >> 16) - 128, mn = (int)(((st.mn & 0xFFFFu) < (st.mn >> 16)) ? (st.mn & 0xFFFFu) : (st.mn >> 16)) - 128;
            int sm = (int)((st.sum & 0xFFFFu) + (st.sum >> 16)) - 128 * st.n;
            if (st.n == 0) { mx = -128; mn = 127; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int omx = __shfl_xor(mx, off, 64), omn = __shfl_xor(mn, off, 64);
                mx = mx > omx ? mx : omx;
                mn = mn < omn ? mn : omn;
                sm += __shfl_xor(sm, off, 64);
            }
            const int cx = pick_row_offset(mx, mn, C);
            if (lane == 0) *(v2i *)(row_off + 2 * r) = (v2i){cx, sm + cx * K};
            c4 = (uint32_t)(cx & 0xFF) * 0x01010101u;
        }
        int8_t *orow = xq + r * (int64_t)K;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            if (idx < nvec) {
                if constexpr (DT == ASQ_F32) {
                    *(uint32_t *)(orow + (int64_t)idx * 4) = OFF ? pk_add_i8(o[i][0], c4) : o[i][0];
                } else {
                    *(uint2 *)(orow + (int64_t)idx * 8) = OFF ? make_uint2(pk_add_i8(o[i][0], c4), pk_add_i8(o[i][1], c4)) : make_uint2(o[i][0], o[i][1]);
                }
            }
        }
    };
    v4i a[NV];
    load(a, row);
    finish(a, row);
}

// true when a launch was made (rows of up to 64 * VEC * 28 elements)
template <int DT, class Q, bool PT, bool OFF>
bool launch_rows_wave(const void *x, int8_t *xq, float *s_row, int32_t *row_off, int64_t M, int64_t K, Q q, int C, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t nvec = K / VEC;
    // (28 vectors per lane for per-token rows only: Mixtral's F = 14336 in fp16, the quantiser in front of w2 -- 64.1 instead of 74.5 us at 8192 rows, round 5; the per-tensor
    // forms of that length run out of registers -- 262 VGPRs, 3.5 TB/s measured -- and stay on the block-per-row kernels)
    if (nvec > 64 * (PT ? 28 : 24) || M >= (1ll << 31)) return false;
    const int nv = (int)((nvec + 63) / 64);
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
#define ASQ_RW(NV) hipLaunchKernelGGL((quant_rows_wave<DT, NV, Q, PT, OFF>), grid, block, 0, s, x, xq, s_row, row_off, (int)M, (int)K, q, C)
    if (nv <= 1) ASQ_RW(1);
    else if (nv <= 2) ASQ_RW(2);
    else if (nv <= 4) ASQ_RW(4);
    else if (nv <= 6) ASQ_RW(6);
    else if (nv <= 8) ASQ_RW(8);
    else if (nv <= 10) ASQ_RW(10);
    else if (nv <= 12) ASQ_RW(12);
    else if (nv <= 16) ASQ_RW(16);
    else if (nv <= 22) ASQ_RW(22);
    else if (nv <= 24) ASQ_RW(24);
    else {
        if constexpr (PT) ASQ_RW(28);
    }
#undef ASQ_RW
    return true;
}

template <int DT> bool launch_per_token_wave(const void *x, int8_t *xq, float *s_row, int64_t M, int64_t K, hipStream_t s)
{
    return launch_rows_wave<DT, QRound<DT>, true, false>(x, xq, s_row, nullptr, M, K, QRound<DT>{}, 0, s);
}

template <int DT, class Q, bool PT> int launch_rows_off(const void *x, int8_t *xq, float *s_row, int32_t *row_off, int64_t M, int64_t K, Q q, int C, hipStream_t s)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t nvec = K / VEC;
    if (launch_rows_wave<DT, Q, PT, true>(x, xq, s_row, row_off, M, K, q, C, s)) return asq_after_launch(s, "asq_quantize_act_off");
    dim3 grid((unsigned)M), block(256);
#define ASQ_RO(NV) hipLaunchKernelGGL((quant_rows_off<DT, NV, Q, PT>), grid, block, 0, s, x, xq, s_row, row_off, (int)K, q, C)
    if (nvec <= 256 * 1) ASQ_RO(1);
    else if (nvec <= 256 * 2) ASQ_RO(2);
    else if (nvec <= 256 * 4) ASQ_RO(4);
    else if (nvec <= 256 * 8) ASQ_RO(8);
    else ASQ_RO(20);
#undef ASQ_RO
    return asq_after_launch(s, "asq_quantize_act_off");
}

template <int DT> int quantize_off_dt(const void *x, int mode, float quant_scale, int8_t *xq, float *s_row, int32_t *row_off, int64_t M, int64_t K, int C, hipStream_t s)
{
    switch (mode) {
    case ASQ_ACT_ROUND: return launch_rows_off<DT, QRound<DT>, false>(x, xq, s_row, row_off, M, K, QRound<DT>{}, C, s);
    case ASQ_ACT_DIV:
        if (quant_scale > 0x1p-60f && quant_scale < 0x1p60f)
            return launch_rows_off<DT, QDivFast<DT>, false>(x, xq, s_row, row_off, M, K, QDivFast<DT>{quant_scale, 1.0f / quant_scale}, C, s);
        return launch_rows_off<DT, QDiv<DT>, false>(x, xq, s_row, row_off, M, K, QDiv<DT>{quant_scale}, C, s);
    default: return launch_rows_off<DT, QRound<DT>, true>(x, xq, s_row, row_off, M, K, QRound<DT>{}, C, s);
    }
}

// weights, once per module: w'[n,k] = w[n,k] + cw[n]; col_off[n] = {cw[n], sum_k w[n,k]} (the sum of the ORIGINAL row).  One block per row, two passes
// (the second one over L2); K % 16 == 0, 16-byte aligned rows.
__global__ void __launch_bounds__(256) weight_offset_image(const int8_t *__restrict__ w, int8_t *__restrict__ w_off, int32_t *__restrict__ col_off, int64_t N,
                                                           int K, int CW)
{
    __shared__ int redi[12];
    const int64_t row = blockIdx.x;
    const uint4 *src = (const uint4 *)(w + row * (int64_t)K);
    uint4 *dst = (uint4 *)(w_off + row * (int64_t)K);
    const int nvec = K / 16;
    int mx = -128, mn = 127, sm = 0;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        const uint4 v = src[i];
        RowStats st;   // (per vector: the packed 16-bit partial sums stay far from overflow)
        st.add(v.x), st.add(v.y), st.add(v.z), st.add(v.w);
        const int vmx = (int)umax32(st.mx & 0xFFFFu, st.mx >> 16) - 128;
        const int vmn = (int)(((st.mn & 0xFFFFu) < (st.mn >> 16)) ? (st.mn & 0xFFFFu) : (st.mn >> 16)) - 128;
        mx = mx > vmx ? mx : vmx;
        mn = mn < vmn ? mn : vmn;
        sm += (int)((st.sum & 0xFFFFu) + (st.sum >> 16)) - 128 * 16;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int omx = __shfl_xor(mx, off, 64);
        mx = mx > omx ? mx : omx;
        sm += __shfl_xor(sm, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { redi[wave] = mx; redi[8 + wave] = sm; }
    __syncthreads();
    const int a = redi[0] > redi[1] ? redi[0] : redi[1], b = redi[2] > redi[3] ? redi[2] : redi[3];
    const int rmax = a > b ? a : b, rsum = (redi[8] + redi[9]) + (redi[10] + redi[11]);
    (void)mn;
    int cw = 127 - rmax;
    cw = cw < CW ? cw : CW;
    if (threadIdx.x == 0) *(v2i *)(col_off + 2 * row) = (v2i){cw, rsum};
    const uint32_t c4 = (uint32_t)(cw & 0xFF) * 0x01010101u;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        const uint4 v = src[i];
        dst[i] = make_uint4(pk_add_i8(v.x, c4), pk_add_i8(v.y, c4), pk_add_i8(v.z, c4), pk_add_i8(v.w, c4));
    }
}

// development overrides of the two offset constants (read once): ASQ_OFF_CX (default 3, 0 = no activation offset), ASQ_OFF_CW (default 64, 0 = no weight offset)
static int offset_const(const char *name, int dflt, int hi)
{
    const char *e = getenv(name);
    if (!e || !e[0]) return dflt;
    const int v = atoi(e);
    return v < 0 ? 0 : (v > hi ? hi : v);
}
static int offset_cx() { static const int v = offset_const("ASQ_OFF_CX", 3, 64); return v; }
static int offset_cw() { static const int v = offset_const("ASQ_OFF_CW", 64, 127); return v; }


// ---------------------------------------------------------------------------------
// N1: RMSNorm (or LayerNorm) with the SmoothQuant scale folded into its weight, emitting int8
// directly -- the fused form of reference models/llama.py:27-37 (QuantizedLlamaRMSNorm: weight /
// input_scale) followed by the per-tensor prologue of the next linears (linear.py:95-96), i.e. the
// reference's own (dead) LayerNormQ idea, layers/nn/fused.py:10-15.  Arithmetic follows HF's
// LlamaRMSNorm: var = mean(f32(x)^2); n = dt(f32(x) * rsqrt(var + eps)); y = dt(w * n);
// xq = int8(clamp(rne(y))) -- or, per-token: s_row = dt(absmax(y)/127), xq = int8(clamp(rne(f32(y)/s_row))).
// One block per row, the row lives in registers (one HBM read, K bytes written).  The fp32 sum order
// differs from ATen's reduction, so y can differ from the unfused path in its last place: the int8
// result equals the two-kernel path except at rounding boundaries (tests: <= 1e-3 of entries, +-1).
// LAYERNORM additionally subtracts the mean and adds a bias (OPT, reference models/opt.py:20-29).
// ---------------------------------------------------------------------------------
// 1 / sqrt(v) from one correctly rounded square root and one correctly rounded division (hipcc keeps fp32 sqrt and
// division IEEE by default): what torch.rsqrt computes on the host, and reproducible operation by operation in
// oracle/n1.py -- v_rsq_f32 (1 ulp, implementation-defined) is not.
// (sqrtf, not __fsqrt_rn: the HIP header maps the latter to the NATIVE square root unless OCML_BASIC_ROUNDED_OPERATIONS is set)
__device__ __forceinline__ float rsqrt_exact(float v) { return __fdiv_rn(1.0f, sqrtf(v)); }

__device__ __forceinline__ float block_sum_256(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();  // red may still be read by the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int DT> __device__ __forceinline__ v4i vec_pack(const float (&f)[ElemT<DT>::VEC])
{
    if constexpr (DT == ASQ_F32) {
        return (v4i){__float_as_int(f[0]), __float_as_int(f[1]), __float_as_int(f[2]), __float_as_int(f[3])};
    } else {
        v4i v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (int)((uint32_t)ElemT<DT>::store(f[2 * i]) | ((uint32_t)ElemT<DT>::store(f[2 * i + 1]) << 16));
        return v;
    }
}

// ADD: the residual-add form (the reference's dq_add_layernorm_q, csrc/kernels/fused.cu:5-25, with a floating `x`
// -- here the previous GEMM's epilogue has already dequantised): h = dt(x + residual) is written to `hout` (the
// new residual stream) and normalised + quantised in the same pass.
template <int DT, int NV, bool LAYERNORM, bool PER_TOKEN, bool ADD>
// (xv, resv and hout carry NO __restrict__: the API allows h_out to alias residual or x -- the residual stream updated in place -- and an aliased
// restrict pointer is undefined behaviour even though every thread loads a vector before it stores the same one)
__global__ void __launch_bounds__(256) norm_quant_cached(const void *xv, const void *resv, void *hout,
                                                         const void *__restrict__ wv, const void *__restrict__ bv, float eps,
                                                         int8_t *__restrict__ xq, float *__restrict__ s_row, int K, int32_t *__restrict__ row_off, int C)
{
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int64_t rowoff = row * (int64_t)K * (16 / VEC);
    const char *xrow = (const char *)xv + rowoff;
    const int nvec = K / VEC;
    float f[NV][VEC];
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            vec_unpack<DT>(*(const v4i *)(xrow + (int64_t)idx * 16), f[i]);
            if constexpr (ADD) {
                float r[VEC];
                vec_unpack<DT>(*(const v4i *)((const char *)resv + rowoff + (int64_t)idx * 16), r);
#pragma unroll
                for (int j = 0; j < VEC; ++j) f[i][j] = ElemT<DT>::round(__fadd_rn(r[j], f[i][j]));  // torch.add(residual, x) in dt
                *(v4i *)((char *)hout + rowoff + (int64_t)idx * 16) = vec_pack<DT>(f[i]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                sum = __fadd_rn(sum, f[i][j]);
                sq = __fadd_rn(sq, __fmul_rn(f[i][j], f[i][j]));
            }
        }
    }
    float mean = 0.f;
    if constexpr (LAYERNORM) {
        mean = __fdiv_rn(block_sum_256(sum, red), (float)K);
        sq = 0.f;  // two-pass variance, as ATen's layer_norm
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * 256 + threadIdx.x < nvec)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float d = __fadd_rn(f[i][j], -mean);
                    sq = __fadd_rn(sq, __fmul_rn(d, d));
                }
    }
    const float var = __fdiv_rn(block_sum_256(sq, red), (float)K);
    const float rs = rsqrt_exact(__fadd_rn(var, eps));
    uint32_t amax = 0;  // |y| maximum as an fp32 bit pattern (see AbsMax)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            float wf[VEC], bf[VEC];
            vec_unpack<DT>(*(const v4i *)((const char *)wv + (int64_t)idx * 16), wf);
            if constexpr (LAYERNORM) vec_unpack<DT>(*(const v4i *)((const char *)bv + (int64_t)idx * 16), bf);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float y;
                if constexpr (LAYERNORM) {
                    y = ElemT<DT>::round(__fadd_rn(__fmul_rn(__fmul_rn(__fadd_rn(f[i][j], -mean), rs), wf[j]), bf[j]));
                } else {
                    const float n = ElemT<DT>::round(__fmul_rn(f[i][j], rs));  // hidden_states.to(input_dtype)
                    y = ElemT<DT>::round(__fmul_rn(wf[j], n));                 // self.weight * ...
                }
                f[i][j] = y;
                if constexpr (PER_TOKEN) amax = umax32(amax, absbits(y));
            }
        }
    }
    float qs = 1.0f, rowmax = 0.0f;
    if constexpr (PER_TOKEN) {
        __syncthreads();
        rowmax = block_absmax_256(amax, red);
        qs = ElemT<DT>::round(rowmax / 127.0f);
        if (threadIdx.x == 0) s_row[row] = qs;
    }
    const RowDivisor d(qs, rowmax);
    const QRowFast qf{d.s, d.y};
    int8_t *orow = xq + row * (int64_t)K;
    if (row_off != nullptr) {   // (block-uniform) offset image: the quantised row stays packed in registers between its statistics and the store (quant_rows_off)
        __shared__ int redi[12];
        uint32_t o[NV][2];
        RowStats st;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                int q[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) q[j] = PER_TOKEN ? quant_i8(d.fast ? qf.div(f[i][j]) : f[i][j] / qs) : quant_i8(f[i][j]);
                o[i][0] = pack4(q[0], q[1], q[2], q[3]);
                st.add(o[i][0]);
                if constexpr (DT != ASQ_F32) {
                    o[i][1] = pack4(q[4], q[5], q[6], q[7]);
                    st.add(o[i][1]);
                }
            }
        }
        int rmax, rmin, rsum;
        block_row_stats(st, redi, rmax, rmin, rsum);
        const int cx = pick_row_offset(rmax, rmin, C);
        if (threadIdx.x == 0) *(v2i *)(row_off + 2 * row) = (v2i){cx, rsum + cx * K};
        const uint32_t c4 = (uint32_t)(cx & 0xFF) * 0x01010101u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + (int64_t)idx * 4) = pk_add_i8(o[i][0], c4);
                else *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(pk_add_i8(o[i][0], c4), pk_add_i8(o[i][1], c4));
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            int q[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) q[j] = PER_TOKEN ? quant_i8(d.fast ? qf.div(f[i][j]) : f[i][j] / qs) : quant_i8(f[i][j]);
            if constexpr (DT == ASQ_F32) {
                *(uint32_t *)(orow + (int64_t)idx * 4) = pack4(q[0], q[1], q[2], q[3]);
            } else {
                *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(pack4(q[0], q[1], q[2], q[3]), pack4(q[4], q[5], q[6], q[7]));
            }
        }
    }
}

// ---- the same two operations WITHOUT the quantiser (round 6): caller-side glue for the reference's module composition, where the norm and the activation are modules of
// their own between the W8A8 linears (models/llama.py:27-37 QuantizedLlamaRMSNorm with the folded weight -> a floating tensor each linear quantises itself; :206-211 /
// HF LlamaMLP act_fn(gate) * up).  As torch ops they are 7 + 2 elementwise / reduction kernels per use (14.6 % + 6.3 % of the GPU time of BASELINE configs[2]'s forward
// in that composition, profiles/r6_cfg3_kernel_stats.txt); here one pass each.  Arithmetic: norm_quant_cached's y / silu_mul_quant_cached's a, i.e. HF's roundings
// (n = dt(f32(x) * rsqrt(var + eps)), y = dt(w * n); a = dt(dt(silu(g)) * u)) in this file's fixed operation order (oracle/n1.py repeats it bit for bit).
template <int DT, int NV> __global__ void __launch_bounds__(256) rmsnorm_rows(const void *__restrict__ xv, const void *__restrict__ wv, float eps, void *__restrict__ yv, int K)
{
    constexpr int VEC = ElemT<DT>::VEC;
    __shared__ float red[4];
    const int64_t rowoff = (int64_t)blockIdx.x * (int64_t)K * (16 / VEC);
    const int nvec = K / VEC;
    float f[NV][VEC];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            vec_unpack<DT>(*(const v4i *)((const char *)xv + rowoff + (int64_t)idx * 16), f[i]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) sq = __fadd_rn(sq, __fmul_rn(f[i][j], f[i][j]));
        }
    }
    const float var = __fdiv_rn(block_sum_256(sq, red), (float)K);
    const float rs = rsqrt_exact(__fadd_rn(var, eps));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            float wf[VEC];
            vec_unpack<DT>(*(const v4i *)((const char *)wv + (int64_t)idx * 16), wf);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float n = ElemT<DT>::round(__fmul_rn(f[i][j], rs));  // hidden_states.to(input_dtype)
                f[i][j] = ElemT<DT>::round(__fmul_rn(wf[j], n));           // self.weight * ...
            }
            *(v4i *)((char *)yv + rowoff + (int64_t)idx * 16) = vec_pack<DT>(f[i]);
        }
    }
}

template <int DT, bool FAST> __global__ void __launch_bounds__(256) silu_mul_flat(const void *__restrict__ gv, const void *__restrict__ uv, void *__restrict__ ov, int64_t nvec)
{
    constexpr int VEC = ElemT<DT>::VEC;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (int64_t)gridDim.x * 256) {
        float g[VEC];
        vec_unpack<DT>(*((const v4i *)gv + idx), g);
        const v4i uw = *((const v4i *)uv + idx);
        v4i o;
        if constexpr (DT == ASQ_F16) {
#pragma unroll
            for (int j = 0; j < VEC; j += 2) o[j / 2] = (int)silu_times_up_h(silu2<FAST>(g[j], g[j + 1]), (uint32_t)uw[j / 2]);
        } else {
            float u[VEC], a[VEC];
            vec_unpack<DT>(uw, u);
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                const v2f pr = silu_times_up<DT>(silu2<FAST>(g[j], g[j + 1]), u[j], u[j + 1]);
                a[j] = pr[0];
                a[j + 1] = pr[1];
            }
            o = vec_pack<DT>(a);
        }
        *((v4i *)ov + idx) = o;
    }
}

template <int DT, bool LN, bool PT, bool ADD = false>
int launch_norm_quant(const void *x, const void *w, const void *b, float eps, int8_t *xq, float *s_row, int64_t M, int64_t K, hipStream_t s,
                      const void *res = nullptr, void *hout = nullptr, int32_t *row_off = nullptr, int C = 0)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t nvec = K / VEC;
    dim3 grid((unsigned)M), block(256);
#define ASQ_NQ(NV) hipLaunchKernelGGL((norm_quant_cached<DT, NV, LN, PT, ADD>), grid, block, 0, s, x, res, hout, w, b, eps, xq, s_row, (int)K, row_off, C)
    if (nvec <= 256 * 1) ASQ_NQ(1);
    else if (nvec <= 256 * 2) ASQ_NQ(2);
    else if (nvec <= 256 * 4) ASQ_NQ(4);
    else ASQ_NQ(8);
#undef ASQ_NQ
    return asq_after_launch(s, "asq_norm_quantize");
}

// ---------------------------------------------------------------------------------
// SiLU(gate) * up -> int8: the activation between gate/up and down_proj (LLaMA / Mixtral MLP) fused with
// down_proj's quantiser (W8A8BFP32OFP32LinearWithQuantScale: per-token, or per-tensor x / quant_scale,
// reference linear.py:283-292).  a = dt(dt(g / (1 + exp(-g))) * u) as the two ATen ops compute it; the
// [M, K] fp16 product never reaches HBM (reads 2 x s_in, writes 1 B per element instead of
// reading 3 x and writing s_in + 1).  exp() is exp_det above: bit-identical to oracle/n1.py; against
// torch's silu (a different exp) the int8 result can differ by +-1 at rounding boundaries.
// ---------------------------------------------------------------------------------
// FAST (opt-in, ASQ_SILU_FAST flag): silu from the hardware transcendentals -- g * v_rcp_f32(1 + v_exp_f32(-g * log2 e)), ~1 ulp each -- instead of the
// bit-reproducible exp_det + IEEE division (30 of the 42 VALU instructions per element).  The int8 result then differs from the exact kernel / oracle/n1.py by
// at most +-1 (bf16: +-2 on < 1e-4 of the elements), where silu(g) * up lies within an fp16 / bf16 ulp of a rounding boundary (tests/test_hip_n1.py).
template <int DT, int NV, bool PER_TOKEN, bool FAST = false>
__global__ void __launch_bounds__(256) silu_mul_quant_cached(const void *__restrict__ gv, const void *__restrict__ uv, float quant_scale,
                                                             int8_t *__restrict__ xq, float *__restrict__ s_row, int K, int32_t *__restrict__ row_off, int C)
{
    constexpr int VEC = ElemT<DT>::VEC;
    // fp16: the product of two fp16 values is exact in fp32, so dt(fp32(sl) * fp32(u)) IS the IEEE fp16 product: one v_pk_mul_f16 per
    // pair on the raw `up` words (no unpack, no round trip), the activation kept as packed halves (half the registers)
    constexpr bool H = DT == ASQ_F16;
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const char *grow = (const char *)gv + row * (int64_t)K * (16 / VEC);
    const char *urow = (const char *)uv + row * (int64_t)K * (16 / VEC);
    const int nvec = K / VEC;
    float a[H ? 1 : NV][H ? 1 : VEC];
    uint32_t ah[H ? NV : 1][4];
    uint32_t amax = 0;  // |a| maximum as an fp32 bit pattern (see AbsMax); fp16: two packed 15-bit patterns
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
            for (int j = 0; j < VEC; j += 2) {  // two elements per packed instruction
                const v2f sl = silu2<FAST>(g[j], g[j + 1]);   // (asq_silu_core.h: the arithmetic the gate||up GEMM's epilogue shares)
                if constexpr (H) {
                    const uint32_t pr = silu_times_up_h(sl, (uint32_t)uw[j / 2]);
                    ah[i][j / 2] = pr;
                    if constexpr (PER_TOKEN) amax = pk_max_u16(amax, pr & 0x7FFF7FFFu);
                } else {
                    const v2f pr = silu_times_up<DT>(sl, u[j], u[j + 1]);
                    a[i][j] = pr[0];
                    a[i][j + 1] = pr[1];
                    if constexpr (PER_TOKEN) amax = umax32(amax, umax32(absbits(a[i][j]), absbits(a[i][j + 1])));
                }
            }
        }
    }
    float qs = quant_scale, rowmax = __builtin_inff();
    if constexpr (PER_TOKEN) {
        if constexpr (H) amax = __float_as_uint(ElemT<DT>::load((uint16_t)umax32(amax & 0xFFFFu, amax >> 16)));  // widening keeps the order, NaN stays NaN
        rowmax = block_absmax_256(amax, red);
        qs = ElemT<DT>::round(rowmax / 127.0f);
        if (threadIdx.x == 0) s_row[row] = qs;
    }
    const RowDivisor d(qs, rowmax);  // per-tensor rows have no known maximum: plain division
    const QRowFast qf{d.s, d.y};
    int8_t *orow = xq + row * (int64_t)K;
    auto quant_one = [&](int i, int (&q)[VEC]) {
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            v2f av;
            if constexpr (H) {
                const v2h hh = __builtin_bit_cast(v2h, ah[i][j / 2]);
                av = (v2f){(float)hh[0], (float)hh[1]};
            } else {
                av = (v2f){a[i][j], a[i][j + 1]};
            }
            if constexpr (PER_TOKEN) {
                if (d.fast) {
                    const v2f t = qf.div2(av);
                    q[j] = quant_i8(t[0]);
                    q[j + 1] = quant_i8(t[1]);
                } else {
                    q[j] = quant_i8(av[0] / qs);
                    q[j + 1] = quant_i8(av[1] / qs);
                }
            } else {
                q[j] = quant_i8(ElemT<DT>::round(av[0] / qs));
                q[j + 1] = quant_i8(ElemT<DT>::round(av[1] / qs));
            }
        }
    };
    if (row_off != nullptr) {   // (block-uniform) offset image, as quant_rows_off
        __shared__ int redi[12];
        uint32_t o[NV][2];
        RowStats st;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                int q[VEC];
                quant_one(i, q);
                o[i][0] = pack4(q[0], q[1], q[2], q[3]);
                st.add(o[i][0]);
                if constexpr (DT != ASQ_F32) {
                    o[i][1] = pack4(q[4], q[5], q[6], q[7]);
                    st.add(o[i][1]);
                }
            }
        }
        int rmax, rmin, rsum;
        block_row_stats(st, redi, rmax, rmin, rsum);
        const int cx = pick_row_offset(rmax, rmin, C);
        if (threadIdx.x == 0) *(v2i *)(row_off + 2 * row) = (v2i){cx, rsum + cx * K};
        const uint32_t c4 = (uint32_t)(cx & 0xFF) * 0x01010101u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 256 + threadIdx.x;
            if (idx < nvec) {
                if constexpr (DT == ASQ_F32) *(uint32_t *)(orow + (int64_t)idx * 4) = pk_add_i8(o[i][0], c4);
                else *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(pk_add_i8(o[i][0], c4), pk_add_i8(o[i][1], c4));
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < nvec) {
            int q[VEC];
            quant_one(i, q);
            if constexpr (DT == ASQ_F32) {
                *(uint32_t *)(orow + (int64_t)idx * 4) = pack4(q[0], q[1], q[2], q[3]);
            } else {
                *(uint2 *)(orow + (int64_t)idx * 8) = make_uint2(pack4(q[0], q[1], q[2], q[3]), pack4(q[4], q[5], q[6], q[7]));
            }
        }
    }
}

template <int DT, bool PT, bool FAST = false>
int launch_silu_mul_quant(const void *g, const void *u, float qs, int8_t *xq, float *s_row, int64_t M, int64_t K, hipStream_t s, int32_t *row_off = nullptr, int C = 0)
{
    constexpr int VEC = ElemT<DT>::VEC;
    const int64_t nvec = K / VEC;
    dim3 grid((unsigned)M), block(256);
#define ASQ_SM(NV) hipLaunchKernelGGL((silu_mul_quant_cached<DT, NV, PT, FAST>), grid, block, 0, s, g, u, qs, xq, s_row, (int)K, row_off, C)
    if (nvec <= 256 * 2) ASQ_SM(2);
    else if (nvec <= 256 * 4) ASQ_SM(4);
    else if (nvec <= 256 * 6) ASQ_SM(6);
    else ASQ_SM(8);
#undef ASQ_SM
    return asq_after_launch(s, "asq_silu_mul_quantize");
}
}  // namespace asq
using namespace asq;

extern "C" int asq_quantize_act(const void *x, int x_dtype, int mode, float quant_scale, int8_t *xq, float *s_row,
                                int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_quantize_act");
    ASQ_REQUIRE(M >= 0 && K >= 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_quantize_act: bad dims M=%lld K=%lld", (long long)M, (long long)K);
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_quantize_act: bad x_dtype %d", x_dtype);
    ASQ_REQUIRE(mode == ASQ_ACT_ROUND || mode == ASQ_ACT_DIV || mode == ASQ_ACT_PER_TOKEN, ASQ_ERR_DTYPE, "asq_quantize_act: bad mode %d", mode);
    if (M == 0 || (K == 0 && mode != ASQ_ACT_PER_TOKEN)) return ASQ_OK;
    ASQ_REQUIRE(xq != nullptr && (x != nullptr || K == 0), ASQ_ERR_NULL, "asq_quantize_act: NULL x / xq");
    ASQ_REQUIRE(mode != ASQ_ACT_PER_TOKEN || s_row != nullptr, ASQ_ERR_NULL, "asq_quantize_act: per-token needs s_row");
    ASQ_REQUIRE(((uintptr_t)x % asq_dtype_size(x_dtype)) == 0, ASQ_ERR_ALIGN, "asq_quantize_act: x misaligned for its dtype");
    ASQ_REQUIRE(mode != ASQ_ACT_PER_TOKEN || ((uintptr_t)s_row % 4) == 0, ASQ_ERR_ALIGN, "asq_quantize_act: s_row misaligned");
    hipStream_t s = (hipStream_t)stream;
    switch (x_dtype) {
    case ASQ_F32: return quantize_dt<ASQ_F32>(x, mode, quant_scale, xq, s_row, M, K, s);
    case ASQ_F16: return quantize_dt<ASQ_F16>(x, mode, quant_scale, xq, s_row, M, K, s);
    default: return quantize_dt<ASQ_BF16>(x, mode, quant_scale, xq, s_row, M, K, s);
    }
}

extern "C" int asq_quantize_act_off(const void *x, int x_dtype, int mode, float quant_scale, int8_t *xq, float *s_row, int32_t *row_off,
                                    int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_quantize_act_off");
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_quantize_act_off: bad dims M=%lld K=%lld", (long long)M, (long long)K);
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_quantize_act_off: bad x_dtype %d", x_dtype);
    ASQ_REQUIRE(mode == ASQ_ACT_ROUND || mode == ASQ_ACT_DIV || mode == ASQ_ACT_PER_TOKEN, ASQ_ERR_DTYPE, "asq_quantize_act_off: bad mode %d", mode);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(x != nullptr && xq != nullptr && row_off != nullptr, ASQ_ERR_NULL, "asq_quantize_act_off: NULL x / xq / row_off");
    ASQ_REQUIRE(mode != ASQ_ACT_PER_TOKEN || s_row != nullptr, ASQ_ERR_NULL, "asq_quantize_act_off: per-token needs s_row");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 20 && K <= 65536, ASQ_ERR_DIM, "asq_quantize_act_off: K must be a multiple of %d and <= %d", vec, 256 * 20 * vec);
    ASQ_REQUIRE((((uintptr_t)x & 15) == 0) && (((uintptr_t)xq & (vec - 1)) == 0) && (((uintptr_t)row_off & 7) == 0) && (((uintptr_t)s_row & 3) == 0), ASQ_ERR_ALIGN,
                "asq_quantize_act_off: x must be 16-B aligned, row_off 8-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int C = offset_cx();
    switch (x_dtype) {
    case ASQ_F32: return quantize_off_dt<ASQ_F32>(x, mode, quant_scale, xq, s_row, row_off, M, K, C, s);
    case ASQ_F16: return quantize_off_dt<ASQ_F16>(x, mode, quant_scale, xq, s_row, row_off, M, K, C, s);
    default: return quantize_off_dt<ASQ_BF16>(x, mode, quant_scale, xq, s_row, row_off, M, K, C, s);
    }
}

extern "C" int asq_weight_offset_image(const int8_t *w, int64_t N, int64_t K, int8_t *w_off, int32_t *col_off, void *stream)
{
    const AsqRange range_("asq_weight_offset_image");
    ASQ_REQUIRE(N >= 0 && K > 0 && N < (1ll << 31) && K <= 65536, ASQ_ERR_DIM, "asq_weight_offset_image: bad dims N=%lld K=%lld (K <= 65536)", (long long)N, (long long)K);
    if (N == 0) return ASQ_OK;
    ASQ_REQUIRE(w != nullptr && w_off != nullptr && col_off != nullptr, ASQ_ERR_NULL, "asq_weight_offset_image: NULL pointer");
    ASQ_REQUIRE(K % 16 == 0 && ((((uintptr_t)w | (uintptr_t)w_off) & 15) == 0) && (((uintptr_t)col_off & 15) == 0) && N % 4 == 0, ASQ_ERR_ALIGN,
                "asq_weight_offset_image: K %% 16 == 0, N %% 4 == 0 and 16-B aligned pointers required");
    hipLaunchKernelGGL(weight_offset_image, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, w, w_off, col_off, N, (int)K, offset_cw());
    return asq_after_launch((hipStream_t)stream, "asq_weight_offset_image");
}

static int norm_quantize_impl(const void *x, int x_dtype, const void *weight, const void *bias, float eps, int per_token, int8_t *xq,
                              float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream)
{
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_norm_quantize: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_norm_quantize: bad x_dtype %d", x_dtype);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(x && weight && xq && (!per_token || s_row), ASQ_ERR_NULL, "asq_norm_quantize: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 8, ASQ_ERR_DIM, "asq_norm_quantize: K must be a multiple of %d and <= %d", vec, 256 * 8 * vec);
    ASQ_REQUIRE(((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0) && (((uintptr_t)xq & (vec - 1)) == 0), ASQ_ERR_ALIGN,
                "asq_norm_quantize: x / weight / bias must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int C = offset_cx();
#define ASQ_NQD(DT_)                                                                                                           \
    (bias ? (per_token ? launch_norm_quant<DT_, true, true>(x, weight, bias, eps, xq, s_row, M, K, s, nullptr, nullptr, row_off, C)                          \
                       : launch_norm_quant<DT_, true, false>(x, weight, bias, eps, xq, s_row, M, K, s, nullptr, nullptr, row_off, C))                         \
          : (per_token ? launch_norm_quant<DT_, false, true>(x, weight, bias, eps, xq, s_row, M, K, s, nullptr, nullptr, row_off, C)                         \
                       : launch_norm_quant<DT_, false, false>(x, weight, bias, eps, xq, s_row, M, K, s, nullptr, nullptr, row_off, C)))
    switch (x_dtype) {
    case ASQ_F32: return ASQ_NQD(ASQ_F32);
    case ASQ_F16: return ASQ_NQD(ASQ_F16);
    default: return ASQ_NQD(ASQ_BF16);
    }
#undef ASQ_NQD
}
extern "C" int asq_norm_quantize(const void *x, int x_dtype, const void *weight, const void *bias, float eps, int per_token, int8_t *xq,
                                 float *s_row, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_norm_quantize");
    return norm_quantize_impl(x, x_dtype, weight, bias, eps, per_token, xq, s_row, nullptr, M, K, stream);
}
extern "C" int asq_norm_quantize_off(const void *x, int x_dtype, const void *weight, const void *bias, float eps, int per_token, int8_t *xq,
                                     float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_norm_quantize_off");
    ASQ_REQUIRE(M == 0 || (row_off != nullptr && ((uintptr_t)row_off & 7) == 0), ASQ_ERR_NULL, "asq_norm_quantize_off: row_off NULL or not 8-B aligned");
    ASQ_REQUIRE(K <= 65536, ASQ_ERR_DIM, "asq_norm_quantize_off: K <= 65536");
    return norm_quantize_impl(x, x_dtype, weight, bias, eps, per_token, xq, s_row, row_off, M, K, stream);
}

static int add_norm_quantize_impl(const void *x, const void *residual, void *h_out, int x_dtype, const void *weight, const void *bias, float eps,
                                  int per_token, int8_t *xq, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream)
{
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_add_norm_quantize: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_add_norm_quantize: bad x_dtype %d", x_dtype);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(x && residual && h_out && weight && xq && (!per_token || s_row), ASQ_ERR_NULL, "asq_add_norm_quantize: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 8, ASQ_ERR_DIM, "asq_add_norm_quantize: K must be a multiple of %d and <= %d", vec, 256 * 8 * vec);
    ASQ_REQUIRE(((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)h_out | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0) && (((uintptr_t)xq & (vec - 1)) == 0),
                ASQ_ERR_ALIGN, "asq_add_norm_quantize: x / residual / h_out / weight / bias must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int C = offset_cx();
#define ASQ_ANQ(DT_)                                                                                                                       \
    (bias ? (per_token ? launch_norm_quant<DT_, true, true, true>(x, weight, bias, eps, xq, s_row, M, K, s, residual, h_out, row_off, C)               \
                       : launch_norm_quant<DT_, true, false, true>(x, weight, bias, eps, xq, s_row, M, K, s, residual, h_out, row_off, C))             \
          : (per_token ? launch_norm_quant<DT_, false, true, true>(x, weight, bias, eps, xq, s_row, M, K, s, residual, h_out, row_off, C)              \
                       : launch_norm_quant<DT_, false, false, true>(x, weight, bias, eps, xq, s_row, M, K, s, residual, h_out, row_off, C)))
    switch (x_dtype) {
    case ASQ_F32: return ASQ_ANQ(ASQ_F32);
    case ASQ_F16: return ASQ_ANQ(ASQ_F16);
    default: return ASQ_ANQ(ASQ_BF16);
    }
#undef ASQ_ANQ
}
extern "C" int asq_add_norm_quantize(const void *x, const void *residual, void *h_out, int x_dtype, const void *weight, const void *bias, float eps,
                                     int per_token, int8_t *xq, float *s_row, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_add_norm_quantize");
    return add_norm_quantize_impl(x, residual, h_out, x_dtype, weight, bias, eps, per_token, xq, s_row, nullptr, M, K, stream);
}
extern "C" int asq_add_norm_quantize_off(const void *x, const void *residual, void *h_out, int x_dtype, const void *weight, const void *bias, float eps,
                                         int per_token, int8_t *xq, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_add_norm_quantize_off");
    ASQ_REQUIRE(M == 0 || (row_off != nullptr && ((uintptr_t)row_off & 7) == 0), ASQ_ERR_NULL, "asq_add_norm_quantize_off: row_off NULL or not 8-B aligned");
    ASQ_REQUIRE(K <= 65536, ASQ_ERR_DIM, "asq_add_norm_quantize_off: K <= 65536");
    return add_norm_quantize_impl(x, residual, h_out, x_dtype, weight, bias, eps, per_token, xq, s_row, row_off, M, K, stream);
}

static int silu_mul_quantize_impl(const void *gate, const void *up, int x_dtype, int per_token, float quant_scale, int8_t *xq, float *s_row, int32_t *row_off,
                                  int64_t M, int64_t K, void *stream)
{
    ASQ_REQUIRE((per_token & ~3) == 0, ASQ_ERR_DTYPE, "asq_silu_mul_quantize: per_token is a bit field (bit 0 per-token, bit 1 ASQ_SILU_FAST), got %d", per_token);
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_silu_mul_quantize: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_silu_mul_quantize: bad x_dtype %d", x_dtype);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(gate && up && xq && (!(per_token & 1) || s_row), ASQ_ERR_NULL, "asq_silu_mul_quantize: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 8, ASQ_ERR_DIM, "asq_silu_mul_quantize: K must be a multiple of %d and <= %d", vec, 256 * 8 * vec);
    ASQ_REQUIRE(((((uintptr_t)gate | (uintptr_t)up) & 15) == 0) && (((uintptr_t)xq & (vec - 1)) == 0), ASQ_ERR_ALIGN,
                "asq_silu_mul_quantize: gate / up must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool fast = (per_token & ASQ_SILU_FAST) != 0;
    per_token &= 1;
    const int C = offset_cx();
#define ASQ_SMD(DT_)                                                                                                                              \
    (fast ? (per_token ? launch_silu_mul_quant<DT_, true, true>(gate, up, quant_scale, xq, s_row, M, K, s, row_off, C)                                        \
                       : launch_silu_mul_quant<DT_, false, true>(gate, up, quant_scale, xq, s_row, M, K, s, row_off, C))                                     \
          : (per_token ? launch_silu_mul_quant<DT_, true>(gate, up, quant_scale, xq, s_row, M, K, s, row_off, C)                                              \
                       : launch_silu_mul_quant<DT_, false>(gate, up, quant_scale, xq, s_row, M, K, s, row_off, C)))
    switch (x_dtype) {
    case ASQ_F32: return ASQ_SMD(ASQ_F32);
    case ASQ_F16: return ASQ_SMD(ASQ_F16);
    default: return ASQ_SMD(ASQ_BF16);
    }
#undef ASQ_SMD
}
extern "C" int asq_silu_mul_quantize(const void *gate, const void *up, int x_dtype, int per_token, float quant_scale, int8_t *xq, float *s_row,
                                     int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_silu_mul_quantize");
    return silu_mul_quantize_impl(gate, up, x_dtype, per_token, quant_scale, xq, s_row, nullptr, M, K, stream);
}
extern "C" int asq_silu_mul_quantize_off(const void *gate, const void *up, int x_dtype, int per_token, float quant_scale, int8_t *xq, float *s_row,
                                         int32_t *row_off, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_silu_mul_quantize_off");
    ASQ_REQUIRE(M == 0 || (row_off != nullptr && ((uintptr_t)row_off & 7) == 0), ASQ_ERR_NULL, "asq_silu_mul_quantize_off: row_off NULL or not 8-B aligned");
    ASQ_REQUIRE(K <= 65536, ASQ_ERR_DIM, "asq_silu_mul_quantize_off: K <= 65536");
    return silu_mul_quantize_impl(gate, up, x_dtype, per_token, quant_scale, xq, s_row, row_off, M, K, stream);
}

extern "C" int asq_rmsnorm(const void *x, int x_dtype, const void *weight, float eps, void *y, int64_t M, int64_t K, void *stream)
{
    const AsqRange range_("asq_rmsnorm");
    ASQ_REQUIRE(M >= 0 && K > 0 && M < (1ll << 31), ASQ_ERR_DIM, "asq_rmsnorm: bad dims");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_rmsnorm: bad x_dtype %d", x_dtype);
    if (M == 0) return ASQ_OK;
    ASQ_REQUIRE(x && weight && y, ASQ_ERR_NULL, "asq_rmsnorm: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(K % vec == 0 && K / vec <= 256 * 8, ASQ_ERR_DIM, "asq_rmsnorm: K must be a multiple of %d and <= %d", vec, 256 * 8 * vec);
    ASQ_REQUIRE(((((uintptr_t)x) | ((uintptr_t)weight) | ((uintptr_t)y)) & 15) == 0, ASQ_ERR_ALIGN, "asq_rmsnorm: x / weight / y must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvec = K / vec;
    dim3 grid((unsigned)M), block(256);
#define ASQ_RN(DT_, NV) hipLaunchKernelGGL((rmsnorm_rows<DT_, NV>), grid, block, 0, s, x, weight, eps, y, (int)K)
#define ASQ_RN_DT(DT_) do { if (nvec <= 256) ASQ_RN(DT_, 1); else if (nvec <= 512) ASQ_RN(DT_, 2); else if (nvec <= 1024) ASQ_RN(DT_, 4); else ASQ_RN(DT_, 8); } while (0)
    switch (x_dtype) {
    case ASQ_F32: ASQ_RN_DT(ASQ_F32); break;
    case ASQ_F16: ASQ_RN_DT(ASQ_F16); break;
    default: ASQ_RN_DT(ASQ_BF16); break;
    }
#undef ASQ_RN_DT
#undef ASQ_RN
    return asq_after_launch(s, "asq_rmsnorm");
}

extern "C" int asq_silu_mul(const void *gate, const void *up, int x_dtype, int flags, void *out, int64_t n, void *stream)
{
    const AsqRange range_("asq_silu_mul");
    ASQ_REQUIRE(n >= 0, ASQ_ERR_DIM, "asq_silu_mul: bad size");
    ASQ_REQUIRE(x_dtype == ASQ_F32 || x_dtype == ASQ_F16 || x_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_silu_mul: bad x_dtype %d", x_dtype);
    ASQ_REQUIRE((flags & ~ASQ_SILU_FAST) == 0, ASQ_ERR_DTYPE, "asq_silu_mul: flags is 0 or ASQ_SILU_FAST, got %d", flags);
    if (n == 0) return ASQ_OK;
    ASQ_REQUIRE(gate && up && out, ASQ_ERR_NULL, "asq_silu_mul: NULL pointer");
    const int vec = x_dtype == ASQ_F32 ? 4 : 8;
    ASQ_REQUIRE(n % vec == 0, ASQ_ERR_DIM, "asq_silu_mul: the element count must be a multiple of %d", vec);
    ASQ_REQUIRE(((((uintptr_t)gate) | ((uintptr_t)up) | ((uintptr_t)out)) & 15) == 0, ASQ_ERR_ALIGN, "asq_silu_mul: gate / up / out must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvec = n / vec;
    int64_t blocks = (nvec + 255) / 256;
    blocks = blocks > 256 * 64 ? 256 * 64 : blocks;
    const bool fast = (flags & ASQ_SILU_FAST) != 0;
#define ASQ_SM_DT(DT_) do { if (fast) hipLaunchKernelGGL((silu_mul_flat<DT_, true>), dim3((unsigned)blocks), dim3(256), 0, s, gate, up, out, nvec); \
                            else hipLaunchKernelGGL((silu_mul_flat<DT_, false>), dim3((unsigned)blocks), dim3(256), 0, s, gate, up, out, nvec); } while (0)
    switch (x_dtype) {
    case ASQ_F32: ASQ_SM_DT(ASQ_F32); break;
    case ASQ_F16: ASQ_SM_DT(ASQ_F16); break;
    default: ASQ_SM_DT(ASQ_BF16); break;
    }
#undef ASQ_SM_DT
    return asq_after_launch(s, "asq_silu_mul");
}
