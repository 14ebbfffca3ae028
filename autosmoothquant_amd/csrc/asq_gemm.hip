// INT8 x INT8 -> INT32 GEMM on CDNA4 matrix cores with fused epilogues.
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
#include "asq_common.h"

namespace {

// ---------------------------------------------------------------------------------
// epilogue functors: consume 4 consecutive-n accumulators of one token row
// ---------------------------------------------------------------------------------
struct EpiI32 {
    int32_t *out;
    int64_t N;
    bool vec_ok;
    __device__ __forceinline__ void store4(int64_t m, int64_t n, int a0, int a1, int a2, int a3, int64_t Ncols) const
    {
        int32_t *p = out + m * N + n;
        if (vec_ok && n + 3 < Ncols) {
            *(v4i *)p = (v4i){a0, a1, a2, a3};
        } else {
            if (n < Ncols) p[0] = a0;
            if (n + 1 < Ncols) p[1] = a1;
            if (n + 2 < Ncols) p[2] = a2;
            if (n + 3 < Ncols) p[3] = a3;
        }
    }
};

template <int DT> struct EpiDequant {
    void *out;
    int64_t N;
    float s_scalar;
    const float *s_row;  // [M] or null
    const float *s_col;  // [N] or null
    const float *bias;   // [N] or null
    int order;
    bool vec_ok;

    __device__ __forceinline__ float one(int acc, float sc, float sr, bool has_row, float b, bool has_bias) const
    {
        const float a = (float)acc;  // v_cvt_f32_i32: round-to-nearest-even, as ATen
        float v;
        if (order == ASQ_EPI_SCALE_FIRST) {
            const float ds = has_row ? __fmul_rn(sc, sr) : sc;
            v = __fmul_rn(ds, a);
        } else {
            v = __fmul_rn(a, sc);
            if (has_row) v = __fmul_rn(v, sr);
        }
        if (has_bias) v = __fadd_rn(v, b);
        return v;
    }

    __device__ __forceinline__ void store4(int64_t m, int64_t n, int a0, int a1, int a2, int a3, int64_t Ncols) const
    {
        using E = ElemT<DT>;
        const bool has_row = s_row != nullptr, has_bias = bias != nullptr;
        const float sr = has_row ? s_row[m] : 1.0f;
        float sc[4] = {s_scalar, s_scalar, s_scalar, s_scalar}, b[4] = {0.f, 0.f, 0.f, 0.f};
        const int acc[4] = {a0, a1, a2, a3};
        const bool full = n + 3 < Ncols;
        if (full && vec_ok) {
            if (s_col) {
                v4f t = *(const v4f *)(s_col + n);
                sc[0] = t[0]; sc[1] = t[1]; sc[2] = t[2]; sc[3] = t[3];
            }
            if (has_bias) {
                v4f t = *(const v4f *)(bias + n);
                b[0] = t[0]; b[1] = t[1]; b[2] = t[2]; b[3] = t[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < Ncols) {
                    if (s_col) sc[i] = s_col[n + i];
                    if (has_bias) b[i] = bias[n + i];
                }
        }
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = one(acc[i], sc[i], sr, has_row, b[i], has_bias);
        typename E::type *p = (typename E::type *)out + m * N + n;
        if (full && vec_ok) {
            if constexpr (DT == ASQ_F32) {
                *(v4f *)p = (v4f){v[0], v[1], v[2], v[3]};
            } else {
                uint32_t lo = (uint32_t)E::store(v[0]) | ((uint32_t)E::store(v[1]) << 16);
                uint32_t hi = (uint32_t)E::store(v[2]) | ((uint32_t)E::store(v[3]) << 16);
                *(uint2 *)p = make_uint2(lo, hi);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < Ncols) p[i] = E::store(v[i]);
        }
    }
};

struct EpiI8 {  // out = sat_i8(rne(alpha*acc + beta*c)), c = previous out
    int8_t *out;
    int64_t N;
    float alpha, beta;
    bool vec_ok;
    __device__ __forceinline__ int one(int acc, int c) const
    {
        float v = __fmul_rn(alpha, (float)acc);
        if (beta != 0.0f) v = __fadd_rn(v, __fmul_rn(beta, (float)c));
        return quant_i8(v);
    }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, int a0, int a1, int a2, int a3, int64_t Ncols) const
    {
        int8_t *p = out + m * N + n;
        const int acc[4] = {a0, a1, a2, a3};
        if (vec_ok && n + 3 < Ncols) {
            uint32_t c = (beta != 0.0f) ? *(const uint32_t *)p : 0u;
            uint32_t r = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r |= (uint32_t)(one(acc[i], (int)(int8_t)(c >> (8 * i))) & 0xFF) << (8 * i);
            *(uint32_t *)p = r;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < Ncols) p[i] = (int8_t)one(acc[i], (beta != 0.0f) ? (int)p[i] : 0);
        }
    }
};

// write one 32(n) x 32(m) accumulator tile: lane owns token m0+(l&31), channels n0 + 8g + 4(l>>5) + 0..3
template <class Epi>
__device__ __forceinline__ void store_acc_tile(const Epi &epi, const v16i &acc, int64_t m0, int64_t n0, int lane, int64_t M, int64_t N)
{
    const int64_t m = m0 + (lane & 31);
    if (m >= M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t n = n0 + 8 * g + 4 * (lane >> 5);
        if (n < N) epi.store4(m, n, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], N);
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
    v16i acc = {0};
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
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        }
    }
    store_acc_tile(epi, acc, m0 + wm * 32, n0 + wn * 32, lane, M, N);
}

// ---------------------------------------------------------------------------------
// "t256": 256(m) x 256(n) block tile, BK = 128, 8 waves (2 along m x 4 along n), each wave
// 128(m) x 64(n) = 4 x 2 MFMA tiles of 32x32.  Operand tiles go HBM/L2 -> LDS with
// global_load_lds (16 B/lane, no VGPR round trip) into two 64 KiB stages.
//
// LDS image of one operand tile: [256 rows][8 chunks of 16 B], row-major, 128 B rows, with
// chunk index XOR-swizzled by ((row>>1)&7).  A ds_read_b128 fragment read touches, per
// 16-lane service group, 16 distinct rows at one logical chunk -> 16 distinct 16-B slots of
// the 256-B bank row: conflict-free.  global_load_lds writes lane-linearly (1 KiB = 8 rows
// per wave-instruction), so the swizzle is applied to the per-lane GLOBAL source address.
//
// Requirements (checked by the dispatcher): K % 128 == 0, x and w 16-B aligned.
// M, N arbitrary: out-of-range rows are clamped for loading and masked at the store.
// ---------------------------------------------------------------------------------
constexpr int T_BM = 256, T_BN = 256, T_BK = 128;
constexpr int T_TILE_BYTES = 256 * T_BK;          // 32 KiB per operand tile
constexpr int T_STAGE_BYTES = 2 * T_TILE_BYTES;   // X tile + W tile
constexpr int T_LDS_BYTES = 2 * T_STAGE_BYTES;    // double buffered: 128 KiB

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// issue the 4 global_load_lds of this wave for one operand tile
__device__ __forceinline__ void stage_tile(const int8_t *__restrict__ g, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                           char *lds_tile, int wave, int lane)
{
    const int r8 = lane >> 3, cphys = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rg = wave * 4 + i;           // 8-row group 0..31
        const int rl = rg * 8 + r8;            // row within the tile
        const int clog = cphys ^ ((rl >> 1) & 7);
        int64_t rglob = row0 + rl;
        rglob = rglob < nrows ? rglob : nrows - 1;
        const int8_t *src = g + rglob * ld + k0 + clog * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + rg * 1024), 16, 0, 0);
    }
}

// bijective XCD-aware remap: consecutive logical ids land on the same XCD (its private L2
// then serves the operand panels shared by neighbouring tiles)
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <class Epi>
__global__ void __launch_bounds__(512) gemm_i8_t256(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N,
                                                    int64_t K, int tiles_m, int tiles_n, Epi epi)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // tile assignment: XCD remap, then groups of GM tile-rows walked column-major
    constexpr int GM = 4;
    const int nwg = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = (int64_t)tile_m * T_BM, n0 = (int64_t)tile_n * T_BN;

    // per-lane fragment read offsets inside an operand tile (swizzled), one per k-substep
    const int frow = lane & 31;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * T_BK + (((ks * 2 + (lane >> 5)) ^ ((frow >> 1) & 7)) << 4);

    v16i acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v16i){0};

    const int nt = (int)(K / T_BK);
    stage_tile(x, K, m0, M, 0, lds, wave, lane);
    stage_tile(w, K, n0, N, 0, lds + T_TILE_BYTES, wave, lane);

    for (int t = 0; t < nt; ++t) {
        char *cur = lds + (t & 1) * T_STAGE_BYTES;
        if (t + 1 < nt) {
            char *nxt = lds + ((t + 1) & 1) * T_STAGE_BYTES;
            stage_tile(x, K, m0, M, (int64_t)(t + 1) * T_BK, nxt, wave, lane);
            stage_tile(w, K, n0, N, (int64_t)(t + 1) * T_BK, nxt + T_TILE_BYTES, wave, lane);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile t landed; tile t+1 (8 DMAs) stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave's share of tile t is in LDS
        const char *xs = cur + (wm * 128) * T_BK;
        const char *ws = cur + T_TILE_BYTES + (wn * 64) * T_BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            v4i a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *(const v4i *)(ws + i * 32 * T_BK + foff[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const v4i *)(xs + j * 32 * T_BK + foff[ks]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // all fragment reads of `cur` retired before it is restaged
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            store_acc_tile(epi, acc[i][j], m0 + wm * 128 + j * 32, n0 + wn * 64 + i * 32, lane, M, N);
}

// ---------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------
enum GemmKernel { KERN_GENERIC = 0, KERN_T256 = 1 };

GemmKernel pick_kernel(const void *x, const void *w, int64_t M, int64_t N, int64_t K)
{
    const bool aligned = ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
    if (aligned && K % T_BK == 0 && K >= T_BK && M >= 128 && N >= 128) return KERN_T256;
    return KERN_GENERIC;
}

template <class Epi> int launch_gemm(const int8_t *x, const int8_t *w, int64_t M, int64_t N, int64_t K, Epi epi, hipStream_t s, const char *what)
{
    if (M == 0 || N == 0) return ASQ_OK;
    const GemmKernel kern = pick_kernel(x, w, M, N, K);
    if (kern == KERN_T256) {
        const int64_t tm = (M + T_BM - 1) / T_BM, tn = (N + T_BN - 1) / T_BN;
        ASQ_REQUIRE(tm * tn < (1ll << 31), ASQ_ERR_DIM, "%s: too many tiles", what);
        static bool attr_set[1] = {false};
        auto kfn = gemm_i8_t256<Epi>;
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)attr_set;
        if (e != hipSuccess) {
            asq_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
            return (int)e;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)(tm * tn)), dim3(512), T_LDS_BYTES, s, x, w, M, N, K, (int)tm, (int)tn, epi);
    } else {
        const bool fast = (K % 16 == 0) && (((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0);
        dim3 grid((unsigned)((N + GEN_T - 1) / GEN_T), (unsigned)((M + GEN_T - 1) / GEN_T));
        ASQ_REQUIRE(grid.y < 65536, ASQ_ERR_DIM, "%s: M too large for the generic kernel (K %% 128 != 0 or unaligned operands)", what);
        hipLaunchKernelGGL((gemm_i8_generic<Epi>), grid, dim3(256), 0, s, x, w, M, N, K, fast, epi);
    }
    return asq_after_launch(s, what);
}

int check_gemm_args(const char *what, const void *x, const void *w, const void *out, int64_t M, int64_t N, int64_t K)
{
    ASQ_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), ASQ_ERR_DIM,
                "%s: bad dims M=%lld N=%lld K=%lld", what, (long long)M, (long long)N, (long long)K);
    if (M == 0 || N == 0) return ASQ_OK;
    ASQ_REQUIRE(out != nullptr, ASQ_ERR_NULL, "%s: NULL out", what);
    ASQ_REQUIRE(K == 0 || (x != nullptr && w != nullptr), ASQ_ERR_NULL, "%s: NULL x / w", what);
    return ASQ_OK;
}

}  // namespace

extern "C" const char *asq_gemm_kernel_name(int64_t M, int64_t N, int64_t K)
{
    switch (pick_kernel(nullptr, nullptr, M, N, K)) {
    case KERN_T256: return "t256";
    default: return "generic";
    }
}

extern "C" int asq_gemm_i8_i32(const int8_t *x, const int8_t *w, int32_t *out, int64_t M, int64_t N, int64_t K, void *stream)
{
    int rc = check_gemm_args("asq_gemm_i8_i32", x, w, out, M, N, K);
    if (rc) return rc;
    ASQ_REQUIRE(((uintptr_t)out & 3) == 0, ASQ_ERR_ALIGN, "asq_gemm_i8_i32: out misaligned");
    EpiI32 epi{out, N, (N % 4 == 0) && (((uintptr_t)out & 15) == 0)};
    return launch_gemm(x, w, M, N, K, epi, (hipStream_t)stream, "asq_gemm_i8_i32");
}

extern "C" int asq_gemm_i8_i8(const int8_t *x, const int8_t *w, int8_t *out, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                              void *stream)
{
    int rc = check_gemm_args("asq_gemm_i8_i8", x, w, out, M, N, K);
    if (rc) return rc;
    EpiI8 epi{out, N, alpha, beta, (N % 4 == 0) && (((uintptr_t)out & 3) == 0)};
    return launch_gemm(x, w, M, N, K, epi, (hipStream_t)stream, "asq_gemm_i8_i8");
}

extern "C" int asq_linear_w8a8(const int8_t *xq, const int8_t *w, void *out, int out_dtype, int64_t M, int64_t N, int64_t K,
                               float s_scalar, const float *s_row, const float *s_col, const float *bias, int epi_order, void *stream)
{
    int rc = check_gemm_args("asq_linear_w8a8", xq, w, out, M, N, K);
    if (rc) return rc;
    ASQ_REQUIRE(out_dtype == ASQ_F32 || out_dtype == ASQ_F16 || out_dtype == ASQ_BF16, ASQ_ERR_DTYPE, "asq_linear_w8a8: bad out_dtype %d", out_dtype);
    ASQ_REQUIRE(epi_order == ASQ_EPI_SCALE_FIRST || epi_order == ASQ_EPI_ACC_FIRST, ASQ_ERR_DTYPE, "asq_linear_w8a8: bad epi_order %d", epi_order);
    ASQ_REQUIRE(((uintptr_t)out % asq_dtype_size(out_dtype)) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8: out misaligned");
    ASQ_REQUIRE((((uintptr_t)s_row | (uintptr_t)s_col | (uintptr_t)bias) & 3) == 0, ASQ_ERR_ALIGN, "asq_linear_w8a8: scale/bias misaligned");
    const size_t vbytes = out_dtype == ASQ_F32 ? 16 : 8;
    const bool vec_ok = (N % 4 == 0) && (((uintptr_t)out & (vbytes - 1)) == 0) && ((((uintptr_t)s_col | (uintptr_t)bias) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    switch (out_dtype) {
    case ASQ_F32: return launch_gemm(xq, w, M, N, K, EpiDequant<ASQ_F32>{out, N, s_scalar, s_row, s_col, bias, epi_order, vec_ok}, s, "asq_linear_w8a8");
    case ASQ_F16: return launch_gemm(xq, w, M, N, K, EpiDequant<ASQ_F16>{out, N, s_scalar, s_row, s_col, bias, epi_order, vec_ok}, s, "asq_linear_w8a8");
    default: return launch_gemm(xq, w, M, N, K, EpiDequant<ASQ_BF16>{out, N, s_scalar, s_row, s_col, bias, epi_order, vec_ok}, s, "asq_linear_w8a8");
    }
}
