// "skinny_fq": the whole W8A8 module forward for <= 16 rows in ONE launch -- the activation quantiser is the GEMM's prologue (north star: "quant / dequant
// fused into prologue / epilogue"; reference layers/nn/linear.py:88-96 per-token absmax / 127 and round / clamp, :283-292 the quant_scale division, :93-104 the
// dequant + bias epilogue).  Decode steps and BASELINE configs[0] (4 x 4096 x 4096) are launch-bound: the two-launch forward costs 5.6-8.3 us of host time and a
// ~2 us gap on the device for an ~8 us weight stream.
//
// Why the prologue can live in every block.  gemm_i8_skinny re-reads its int8 X rows from L2 once per work item anyway; for <= 16 rows the floating-point X is
// small (4 x 4096 fp32 = 64 KB), so each block
//   1. starts its weight stream (two 128-byte K units of W per wave by LDS-DMA: the stream begins at t = 0, the prologue runs under its first-data latency),
//   2. reads the rows of X once, quantises them with the stand-alone quantisers' own cores (asq_quant_core.h: the same IEEE operations in the same order ->
//      the same int8 bits; per-token: block-wide row maxima through LDS atomics on the bit patterns, then the exact row division) and leaves them in LDS as
//      a RESIDENT int8 image [K / 128 units][MR rows][8 swizzled 16-byte chunks] -- the unit image gemm_i8_skinny's fragment reads are conflict-free on,
//   3. runs gemm_i8_skinny's wave-private W ring / k-split / in-block reduction with the X fragments read from the resident image: no X traffic at all in the
//      main loop, and a block that walks several channel tiles (grid-stride) quantises once.
// The redundant work (every block reads and quantises the same rows) grows with M x the number of blocks: measured, the fused launch wins up to 4 rows and loses
// from 8 (SKFQ_MAX_ROWS below), so that is where asq_forward_fused_supported() stops; the kernel itself runs any M <= 16 (tests, ASQ_FQ_MAXROWS).
//
// Requirements: K % 128 == 0, x rows 16-byte aligned (K * sizeof(x) % 16 == 0 follows), MR x K <= SKFQ_XIMG_MAX bytes of LDS, w 16-byte aligned.
#pragma once
#include "asq_quant_core.h"

namespace asq {

constexpr int SKFQ_STAGES = 3;       // gemm_i8_skinny's depth.  Measured (profiles/r5_fused_forward_sweep.txt): a 6-stage ring is no faster at one channel tile per block
                                     // (4 x 4096 x 4096: 8.9 vs 9.4 us) and slower as soon as a block walks several tiles (8 x 11008 x 4096: 29.4 vs 20.4 us)
constexpr int SKFQ_MAX_ROWS = 4;     // Rows the fused launch takes by default IN EVERY MODE (5 .. 16 rows: skfq_supported's second rule).  Measured against quantiser + GEMM on the device (same file): <= 4 rows win everywhere up to ~700
                                     // channel tiles (4 x 4096 x 4096 8.9 vs 9.6 us, 4 x 11008 x 4096 14.0 vs 14.6, 2 x 4096 x 11008 12.7 vs 14.2) and save one launch of host time
                                     // (module call 10.3 vs 13.2 us); 8 and 16 rows LOSE (8 x 4096 x 4096 10.8 vs 10.3, 8 x 11008 x 4096 20.4 vs 15.3, 16 x 4096 x 4096 13.7 vs 10.2):
                                     // every block repeats the quantiser's work.  ASQ_FQ_MAXROWS (1..16) overrides it for A/B runs.
constexpr int SKFQ_ONE_TILE_BLOCKS = 256;   // channel tiles of 16 that get a block each (one per CU) at 5 .. 16 rows
constexpr int SKFQ_MAX_TILES = 1024; // ... and 4 x 32000 x 4096 (1000 tiles of 32 channels) loses too (31.7 vs 25.8 us): beyond this many 16-channel tiles the two-launch forward stays
constexpr int SKFQ_XIMG_MAX = 64 * 1024;   // resident int8 X image (MR rows x K): leaves >= 96 KB for the W rings of 1-2 blocks per CU

// epilogue of the fused forward: EpiDequant<DT>::one with ASQ_EPI_SCALE_FIRST, the optional operands as wave-uniform runtime flags (a lane stores 4 outputs per
// channel tile: the branches cost nothing here, and 6 kernels replace 48 instantiations)
template <int DT> struct FqEpi {
    void *out;
    const float *s_col, *bias;
    float s_scalar;
    __device__ __forceinline__ void store4(int64_t m, int64_t n, const v4i &a, bool has_row, float sr, int64_t N) const
    {
        using E = ElemT<DT>;
        typename E::type *p = (typename E::type *)out + m * N + n;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = n + i < N;
            const float sc = (s_col && in) ? s_col[n + i] : s_scalar;
            const float ds = has_row ? __fmul_rn(sc, sr) : sc;
            v[i] = __fmul_rn(ds, (float)a[i]);
            if (bias && in) v[i] = __fadd_rn(v[i], bias[n + i]);
        }
        if (n + 3 < N && ((((uintptr_t)p) & (4 * sizeof(typename E::type) - 1)) == 0)) {
            if constexpr (DT == ASQ_F32) {
                *(v4f *)p = (v4f){v[0], v[1], v[2], v[3]};
            } else {
                const uint32_t lo = (uint32_t)E::store(v[0]) | ((uint32_t)E::store(v[1]) << 16);
                const uint32_t hi = (uint32_t)E::store(v[2]) | ((uint32_t)E::store(v[3]) << 16);
                *(uint2 *)p = make_uint2(lo, hi);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < N) p[i] = E::store(v[i]);
        }
    }
};

// 16 consecutive elements of one row: raw vectors (fp32: 4, 16-bit types: 2)
template <int DT> struct FqGroup {
    static constexpr int NV = 16 / ElemT<DT>::VEC;
    v4i v[NV];
    __device__ __forceinline__ void load(const char *p)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *(const v4i *)(p + 16 * i);
    }
    template <class Q> __device__ __forceinline__ v4i quant(const Q &q) const
    {
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            uint32_t t[2];
            quant_vec<DT>(v[i], q, t);
            if constexpr (DT == ASQ_F32) o[i] = t[0];
            else o[2 * i] = t[0], o[2 * i + 1] = t[1];
        }
        return (v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    }
};

// STG: stages of the wave-private W ring (STG - 1 units of 128 k-bytes in flight per wave).  gemm_i8_skinny has 3 (its ring also holds X); here the X stages are
// gone and the ring is deep enough that a block requests ALL the weight bytes of its first channel tile before the prologue touches X: the stream runs under it.
template <int DT, int NT, int STG>
__global__ void __launch_bounds__(512) gemm_i8_skinny_fq(const void *__restrict__ xv, const int8_t *__restrict__ w, int M, int64_t N, int64_t K, int wpb, int mr,
                                                         int mode, float quant_scale, int div_fast, FqEpi<DT> epi)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int UNIT = NT * 2048;   // one 128-byte K unit of W: NT x 16 rows
    constexpr int D = 2 * NT;         // DMA instructions per unit per wave
    constexpr int ES = DT == ASQ_F32 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nunits = (int)(K / 128);
    // LDS: [ X image mr * K | row info: 16 maxima (bit patterns) + 16 scales | W rings wpb x 3 x UNIT | reduction slots wpb x NT x 1 KiB ]
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const int ximg_bytes = mr * (int)K;
    unsigned *const rowmax = (unsigned *)(lds + ximg_bytes);
    float *const rowscale = (float *)(lds + ximg_bytes + 64);
    const unsigned ring0 = lds0 + ximg_bytes + 128;
    const unsigned ring = ring0 + wave * STG * UNIT;
    v4i *const red = (v4i *)(lds + ximg_bytes + 128 + wpb * STG * UNIT);   // [wpb][NT][64]

    const int upt = nunits > wave ? (nunits - wave + wpb - 1) / wpb : 0;   // this wave's units per channel tile
    const int64_t ntiles = (N + 16 * NT - 1) / (16 * NT);
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total = my_tiles * upt;

    // ---- W stream: gemm_i8_skinny's DMA lane mapping (8 rows x 128 B per instruction, chunk swizzled by (row >> 1) & 7)
    const int rr = lane >> 3, cp = lane & 7;
    int it_tile = 0, it_u = 0, issued = 0, cur_item = -1;
    unsigned woff[NT][2];
    int64_t it_n0 = 0;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        if (it_tile != cur_item) {
            cur_item = it_tile;
            it_n0 = ((int64_t)blockIdx.x + (int64_t)it_tile * gridDim.x) * (16 * NT);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t r = nt * 16 + 8 * i + rr;
                    r = (it_n0 + r) < N ? r : (N - 1 - it_n0);
                    woff[nt][i] = (unsigned)(r * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
                }
        }
        const int u = wave + it_u * wpb;
        const int8_t *wb = uniform_ptr(w + it_n0 * K + (int64_t)u * 128);
        const unsigned dst = ring + stage * UNIT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) sk_dma16(wb, woff[nt][i], dst + nt * 2048 + i * 1024);
        ++issued;
        if (++it_u == upt) {
            it_u = 0;
            ++it_tile;
        }
    };
    if (upt > 0) {   // the weight stream starts before the prologue touches X
#pragma unroll
        for (int s = 0; s < STG - 1; ++s)
            if (s < total) issue(s);
    }

    // ---- prologue: X -> int8 -> the resident unit image.  Group j (16 elements) of row m lands in unit j / 8, chunk j % 8.
    const int G = (int)(K / 16);            // 16-element groups per row
    const int nthr = wpb * 64;              // (the block has wpb waves)
    const int ngroups = (div_fast & 0x100) ? 0 : M * G;   // (0x100: timing ablation ASQ_FQ_ABL=1 -- no activation loads, the image stays unwritten; results invalid)
    const char *const xb = (const char *)xv;
    auto image_addr = [&](int m, int j) { return lds + (j >> 3) * (mr * 128) + m * 128 + ((((j & 7) ^ ((m >> 1) & 7))) << 4); };
    if (tid < 16) rowmax[tid] = 0;
    for (int idx = tid; idx < (mr - M) * G; idx += nthr) {   // rows M .. mr-1 of the image: zeros (their products are never stored)
        const int m = M + idx / G, j = idx - (idx / G) * G;
        *(v4i *)image_addr(m, j) = (v4i){0, 0, 0, 0};
    }
    // Loads are issued in batches of KEEP groups per thread (64 VGPRs of raw data) BEFORE any of them is used: a rolled `load, quantise, store` loop makes every
    // iteration a dependent round trip to L2 / HBM (measured: 4 iterations = +14 us at 8 x 11008 x 4096, all 256 blocks waiting on the same lines).
    constexpr int KEEP = 32 / (4 * FqGroup<DT>::NV);   // (4 groups of a 16-bit type, 2 of fp32: 4 rows x 4096 in one batch; the unrolled quantiser code stays small)
    const int batch = KEEP * nthr;
    auto src = [&](int idx, int &m, int &j) -> const char * {
        m = idx / G;
        j = idx - m * G;
        return xb + ((int64_t)m * K + (int64_t)j * 16) * ES;
    };
    if (mode == ASQ_ACT_PER_TOKEN) {
        __syncthreads();   // (rowmax is zeroed)
        const bool one_pass = ngroups <= batch;   // the rows stay in registers across the row-maximum reduction
        const bool wave_rows = G % 64 == 0 && ngroups % 64 == 0;   // (block-uniform; then `idx < ngroups` is wave-uniform too)
        FqGroup<DT> g[KEEP];
        for (int base = 0; base < ngroups; base += batch) {   // row maxima as bit patterns (torch.max's NaN propagation: AbsMax)
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int idx = base + tid + i * nthr;
                int m, j;
                if (idx < ngroups) g[i].load(src(idx, m, j));
            }
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int idx = base + tid + i * nthr;
                if (idx < ngroups) {
                    AbsMax<DT> am;
#pragma unroll
                    for (int v = 0; v < FqGroup<DT>::NV; ++v) am.add(g[i].v[v]);
                    uint32_t mb = am.f32bits();
                    if (wave_rows) {   // the 64 lanes of a wave hold 64 consecutive groups of ONE row (G % 64 == 0, whole batches): one atomic per wave, not per lane
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) mb = umax32(mb, (uint32_t)__shfl_xor((int)mb, off, 64));
                        if (lane == 0) atomicMax(&rowmax[idx / G], mb);
                    } else {
                        atomicMax(&rowmax[idx / G], mb);
                    }
                }
            }
        }
        __syncthreads();
        for (int base = 0; base < ngroups; base += batch) {   // quant_scale = absmax.div(127.0) in x's dtype; x / quant_scale in fp32 (linear.py:89-92)
            if (!one_pass) {
#pragma unroll
                for (int i = 0; i < KEEP; ++i) {   // (the rows come back from L2)
                    const int idx = base + tid + i * nthr;
                    int m, j;
                    if (idx < ngroups) g[i].load(src(idx, m, j));
                }
            }
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int idx = base + tid + i * nthr;
                if (idx < ngroups) {
                    const int m = idx / G, j = idx - m * G;
                    const float mx = __uint_as_float(rowmax[m]);
                    const float qs = ElemT<DT>::round(mx / 127.0f);
                    const RowDivisor d(qs, mx);
                    *(v4i *)image_addr(m, j) = d.fast ? g[i].quant(QRowFast{d.s, d.y}) : g[i].quant(QDivF32<DT>{qs});
                    if (j == 0) rowscale[m] = qs;
                }
            }
        }
    } else {
        for (int base = 0; base < ngroups; base += batch) {
            FqGroup<DT> g[KEEP];
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int idx = base + tid + i * nthr;
                int m, j;
                if (idx < ngroups) g[i].load(src(idx, m, j));
            }
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int idx = base + tid + i * nthr;
                if (idx < ngroups) {
                    const int m = idx / G, j = idx - m * G;
                    v4i q;
                    if (mode == ASQ_ACT_ROUND) q = g[i].quant(QRound<DT>{});
                    else if (div_fast & 1) q = g[i].quant(QDivFast<DT>{quant_scale, 1.0f / quant_scale});
                    else q = g[i].quant(QDiv<DT>{quant_scale});
                    *(v4i *)image_addr(m, j) = q;
                }
            }
        }
    }
    __syncthreads();   // the image (and the row scales) are visible to every wave; every compiler-issued load has been waited for: only the W DMAs are in flight
    if (div_fast & 0x200) {   // (timing ablation ASQ_FQ_ABL=2: prologue only)
        sk_wait_vm<0>();
        return;
    }

    // ---- fragment read addresses: lane (r, g) reads row r (mod mr), logical chunk 4h + g of a unit
    const int fr = lane & 15, fg = lane >> 4, xr = fr & (mr - 1);
    unsigned faddr[STG][2], xaddr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int s = 0; s < STG; ++s) {
            faddr[s][h] = ring + s * UNIT + fr * 128 + (((4 * h + fg) ^ ((fr >> 1) & 7)) << 4);
            asm volatile("" : "+v"(faddr[s][h]));
        }
        xaddr[h] = lds0 + xr * 128 + (((4 * h + fg) ^ ((xr >> 1) & 7)) << 4);
    }
    const unsigned xunit = (unsigned)(mr * 128);

    v4i acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (v4i){0, 0, 0, 0};
    int done = 0, c_u = 0, c_tile = 0;
    const bool has_row = mode == ASQ_ACT_PER_TOKEN;
    auto tile_end = [&]() __attribute__((always_inline)) {   // block-wide: sum the wpb partial tiles (exact, order-free), fused epilogue
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            red[(wave * NT + nt) * 64 + lane] = acc[nt];
            acc[nt] = (v4i){0, 0, 0, 0};
        }
        __syncthreads();
        const int64_t n0 = ((int64_t)blockIdx.x + (int64_t)c_tile * gridDim.x) * (16 * NT);
        for (int p = wave; p < NT; p += wpb) {
            v4i s = red[p * 64 + lane];
            for (int v = 1; v < wpb; ++v) s += red[(v * NT + p) * 64 + lane];
            const int64_t n = n0 + p * 16 + 4 * fg;
            if (fr < M && n < N) epi.store4(fr, n, s, has_row, has_row ? rowscale[fr] : 1.0f, N);
        }
        __syncthreads();
        ++c_tile;
    };
    auto step = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(stage_tag)::value;
        if (issued < total) issue((S + STG - 1) % STG);
        const int newer = issued - done - 1;   // units issued after the one consumed now: 0 .. STG - 1
        static_assert(STG >= 3 && STG <= 6, "ring depth");
        if (newer >= STG - 1) sk_wait_vm<(STG - 1) * D>();
        else if (STG > 5 && newer == 4) sk_wait_vm<(STG > 5 ? 4 : 0) * D>();
        else if (STG > 4 && newer == 3) sk_wait_vm<(STG > 4 ? 3 : 0) * D>();
        else if (STG > 3 && newer == 2) sk_wait_vm<(STG > 3 ? 2 : 0) * D>();
        else if (newer == 1) sk_wait_vm<D>();
        else sk_wait_vm<0>();
        const unsigned xu = (unsigned)(wave + c_u * wpb) * xunit;
        v4i wf[NT][2], xf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt][h] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(faddr[S][h] + nt * 2048);
            xf[h] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(xaddr[h] + xu);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt][h], xf[h], acc[nt], 0, 0, 0);
        ++done;
        if (++c_u == upt) {
            c_u = 0;
            tile_end();
        }
    };

    if (upt == 0) {   // more waves than K units: this wave only takes part in the reductions
        for (int t = 0; t < my_tiles; ++t) tile_end();
        return;
    }
    bool more = true;
    while (more) {
        static_for<STG>([&](auto s_) __attribute__((always_inline)) {
            if (more) {
                step(s_);
                more = done < total;
            }
        });
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------
// shapes the fused launch takes: decode-sized row counts whose int8 image fits beside the W rings, on the dispatcher's weight-streaming shapes
static inline int skfq_rows(int64_t M) { return M <= 4 ? 4 : M <= 8 ? 8 : 16; }
// what the KERNEL can run (asq_linear_w8a8_forward_fused) ...
static inline bool skfq_kernel_ok(const void *x, const void *w, int64_t M, int64_t N, int64_t K)
{
    if (M < 1 || M > 16 || N < 1 || K < 128 || K % 128 != 0 || K > (1 << 24)) return false;
    if ((((uintptr_t)x | (uintptr_t)w) & 15) != 0) return false;
    if (N * K >= (1ll << 32)) return false;   // (32-bit row offsets in the DMA address)
    return (int64_t)skfq_rows(M) * K <= SKFQ_XIMG_MAX;
}
// ... and what asq_linear_w8a8_forward hands to it by itself (where it was measured to win)
static inline bool skfq_supported(const void *x, const void *w, int64_t M, int64_t N, int64_t K, int x_dtype, int act_mode)
{
    static const int max_rows = [] { const char *e = getenv("ASQ_FQ_MAXROWS"); const int v = e ? atoi(e) : SKFQ_MAX_ROWS; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    if (!skfq_kernel_ok(x, w, M, N, K)) return false;
    if (M <= max_rows) return (N + 15) / 16 <= SKFQ_MAX_TILES;
    // 5 .. 16 rows: only the per-tensor modes (no row-maximum pass in the prologue) at ONE channel tile per block -- there the module call, which is host-bound at
    // these sizes, gains a launch (8 x 4096 x 4096 10.6 vs 12.7 us, 16 rows 13.3 vs 14.0-16.9) while the device period is within 0.5 us; per-token rows
    // (13.8 vs 12.9) and several tiles per block (8 x 11008 x 4096 16.5 vs 14.4) lose: profiles/r5_fused_forward_module_call.txt
    return max_rows >= 4 && act_mode != ASQ_ACT_PER_TOKEN && (N + 15) / 16 <= SKFQ_ONE_TILE_BLOCKS;
}

template <int DT, int NT, int STG>
int launch_skinny_fq_nt(const void *x, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int mode, float quant_scale, float s_scalar, const float *s_col,
                        const float *bias, hipStream_t s)
{
    constexpr int64_t LDS_CU = 160 * 1024;
    const int mr = skfq_rows(M);
    const int64_t ximg = (int64_t)mr * K + 128;
    const int64_t ntiles = (N + 16 * NT - 1) / (16 * NT);
    const int64_t perwave = STG * NT * 2048 + NT * 1024;   // W ring + reduction slot
    // Waves per block.  Up to 4 rows (a 16-44 KiB image, one load batch per thread): gemm_i8_skinny's rule -- the most waves that still give every channel tile a
    // resident block (4 x 11008 x 4096: 688 blocks of 4 waves 14.0 us, 512 blocks of 8 waves 15.1).  More rows (the explicit entry point): eight waves wherever
    // they fit -- every block carries the image and its prologue, so few fat blocks that walk their tiles beat many thin ones (8 x 11008 x 4096: 688 blocks of 2 waves
    // needed four dependent load batches per thread, prologue alone 13 us, forward 21.1 us; with 8 waves 16.2; two launches 14.8: profiles/r5_fused_forward_ablation.txt).
    int wpb = 8;
    while (wpb > 1 && (ximg + wpb * perwave > LDS_CU || K / 128 < 2 * wpb || (mr <= 4 && 256 * (LDS_CU / (ximg + wpb * perwave)) < ntiles))) wpb >>= 1;
    int64_t per_cu = LDS_CU / (ximg + wpb * perwave);
    if (per_cu * wpb > 16) per_cu = 16 / wpb;   // (<= 4 waves per SIMD: the kernel holds ~100 VGPRs)
    if (per_cu < 1) per_cu = 1;
    int64_t grid = 256 * per_cu;   // persistent beyond that: a block walks its channel tiles grid-stride and quantises once
    if (grid > ntiles) grid = ntiles;
    const size_t lds = (size_t)(ximg + wpb * perwave);
    auto kfn = gemm_i8_skinny_fq<DT, NT, STG>;
    hipError_t e = ensure_dynamic_lds((const void *)kfn, (int)lds);
    if (e != hipSuccess) {
        asq_set_error("skinny_fq: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return (int)e;
    }
    static const int abl = [] { const char *e = getenv("ASQ_FQ_ABL"); return e ? atoi(e) : 0; }();   // development: 1 = no activation loads, 2 = prologue only (timing only, results invalid)
    const int div_fast = ((quant_scale > 0x1p-60f && quant_scale < 0x1p60f) ? 1 : 0) | (abl << 8);   // (bit 0 as quantize_dt: the exact division without dividing, inside its validity range)
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3((unsigned)(wpb * 64)), lds, s, x, w, (int)M, N, K, wpb, mr, mode, quant_scale, div_fast,
                       FqEpi<DT>{out, s_col, bias, s_scalar});
    return ASQ_OK;
}

template <int DT>
int launch_skinny_fq(const void *x, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K, int mode, float quant_scale, float s_scalar, const float *s_col,
                     const float *bias, hipStream_t s)
{
    // 32 channels per block halve the blocks that each quantise X; only with plenty of tiles (as gemm_i8_skinny's `wide`)
    const bool wide = (N + 31) / 32 >= 448 || (K >= 16384 && (N + 31) / 32 >= 128);
    static const int stg = [] { const char *e = getenv("ASQ_FQ_STAGES"); return e ? atoi(e) : SKFQ_STAGES; }();   // development A/B: 3 (default) or 6
    if (stg != 6) return wide ? launch_skinny_fq_nt<DT, 2, 3>(x, w, out, M, N, K, mode, quant_scale, s_scalar, s_col, bias, s)
                              : launch_skinny_fq_nt<DT, 1, 3>(x, w, out, M, N, K, mode, quant_scale, s_scalar, s_col, bias, s);
    return wide ? launch_skinny_fq_nt<DT, 2, 6>(x, w, out, M, N, K, mode, quant_scale, s_scalar, s_col, bias, s)   // (ASQ_FQ_STAGES=6: the measured-and-dropped deep ring)
                : launch_skinny_fq_nt<DT, 1, 6>(x, w, out, M, N, K, mode, quant_scale, s_scalar, s_col, bias, s);
}

}  // namespace asq
