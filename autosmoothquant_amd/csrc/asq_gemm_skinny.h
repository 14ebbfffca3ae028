// "skinny": few rows (decode / cfg1 / short prefill).  The problem is a weight stream: W [N,K]
// int8 is read once (per block of <= 64 rows) from HBM (4096x4096 at M=32: 16.8 MB of W vs 0.13 MB of X
// and 0.26 MB of output) -> the roofline is HBM / L2 bandwidth, not MFMA.
//
// Decomposition: one block of `wpb` waves per 16 output channels (grid-stride over channel tiles);
// the waves split K in 128-byte units (wave v takes units v, v+wpb, ...).  Each unit is
// 2 x v_mfma_i32_16x16x64_i8 per 16-token tile (W rows = matrix-core A operand, X rows = B operand);
// fp8 operands run the same schedule with one v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) per unit.
//
// Both operands travel HBM/L2 -> LDS by LDS-DMA in FULL 128-byte lines (8 rows x 128 B per
// wave-instruction) into a WAVE-PRIVATE 3-stage ring, and are read back as MFMA fragments with
// conflict-free swizzled ds_read_b128.  Measured reasons (tools/ubench, profiles/):
//   * loading fragments straight to VGPRs makes every wave-load touch 16 rows x 64 B: half of each
//     128-B line per instruction.  The W stream then tops out at 3.2-3.5 TB/s (full lines: 5.5-5.9),
//     and the X operand -- which EVERY block re-reads from L2 -- became the bottleneck: X and W
//     times added up instead of overlapping (OPT fc2, M=32: 33 us W-only + 37 us X-only = 65 us);
//   * wave-private rings need no barrier: a wave reads only what it DMA'd itself, ordered by its
//     own counted s_waitcnt vmcnt; two units stay in flight per wave (>= 100 KB per CU).
// The wpb partial accumulators of a tile are summed through LDS (integers: exact, order-free) and
// the first waves run the fused epilogue.  No split-K across blocks, no workspace, no atomics.
// wpb (8/4/2/1) is chosen by the launcher so that every channel tile has a resident block.
//
// Rows are processed in "m-blocks" of MT*16 <= 64 rows: a work item is (channel tile, m-block); the
// m-blocks of one channel tile are mapped to the same XCD so the later ones find W in that L2.
// (Measured: 128-row m-blocks (MT = 8) leave room for only 2 waves per CU and lose to 2 x 64.)
// Each item re-reads its X rows from L2, so this kernel is L2-bandwidth bound (~10-13 TB/s
// measured) once M x N grows: the dispatcher (pick_kernel) hands larger problems to the tiled kernel.
//
// Requirements: K % 128 == 0, K <= 2^24, x / w 16-B aligned.  Ragged N, M: rows are clamped for loading
// and masked at the store.
#pragma once
#include <type_traits>

namespace asq {

constexpr int SK_STAGES = 3;

#ifndef SK_W_NT
#define SK_W_NT 0   // 1: weight rows fetched with the streaming (nt) cache policy.  Measured (skinny_probe, M = 32): 20480x5120 -4 %, 8192x8192 -3.5 %,
                    // 5120x20480 -2.5 %, but 4096x4096 +5 %, 4096x11008 +7 %, M = 64 +9 %: not adopted
#endif
template <bool NT = false> __device__ __forceinline__ void sk_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}

#define SK_VM_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N> __device__ __forceinline__ void sk_wait_vm()
{
    static_assert(N >= 0 && N <= 63 && N % 2 == 0, "vmcnt is a 6-bit counter; unit sizes are even");
    static_assert(N == 0 || N == 2 || (N >= 4 && N <= 20) || N == 24 || N == 28 || N == 32 || N == 36, "no s_waitcnt case for this count (a missing case would wait for nothing)");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SK_VM_CASE(2); SK_VM_CASE(4); SK_VM_CASE(6); SK_VM_CASE(8); SK_VM_CASE(10); SK_VM_CASE(12); SK_VM_CASE(14); SK_VM_CASE(16); SK_VM_CASE(18);
    SK_VM_CASE(20); SK_VM_CASE(24); SK_VM_CASE(28); SK_VM_CASE(32); SK_VM_CASE(36);
}
#undef SK_VM_CASE

// MT = 16-row token tiles per m-block (1..4); NT = 16-channel tiles per work item (1 or 2).  NT = 2 halves the
// number of items that each re-read the X rows from L2 (the kernel's bound once N is large): the launcher
// picks it when there are still >= 256 items.
template <class Epi, int MT, int NT>
__global__ void __launch_bounds__(512) gemm_i8_skinny(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                      int wpb, int mblocks, Epi epi)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int UNIT = (NT + MT) * 2048;  // one 128-byte K unit: NT x 16 W rows + MT x 16 X rows
    constexpr int D = 2 * (NT + MT);        // DMA instructions per unit per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned ring = lds0 + wave * SK_STAGES * UNIT;
    using MMA = typename Epi::Mma;
    using acc4_t = typename MMA::acc4_t;  // v4i (int8: exact, order-free) or v4f (fp8: waves summed in a fixed order)
    acc4_t *const red = (acc4_t *)(lds + wpb * SK_STAGES * UNIT);  // [wpb][NT][MT][64]

    const int nunits = (int)(K / 128);
    const int upt = nunits > wave ? (nunits - wave + wpb - 1) / wpb : 0;  // this wave's units per tile
    // work item i of this block = global item (blockIdx.x + i * gridDim.x) -> (channel tile, m-block);
    // items g and g+8 (same XCD under round-robin dispatch) are the m-blocks of one tile
    const int64_t nitems = (((N + 16 * NT - 1) / (16 * NT) + 7) / 8) * 8 * mblocks;  // tiles padded to groups of 8 (padding items store nothing)
    const int my_tiles = (int)((nitems - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total = my_tiles * upt;  // (item, unit) pairs of this wave, item-major
    auto decode = [&](int i, int64_t &n0, int &mb) __attribute__((always_inline)) {
        const int64_t gidx = (int64_t)blockIdx.x + (int64_t)i * gridDim.x;
        const int64_t grp = gidx / (8 * mblocks), rem = gidx - grp * (8 * mblocks);
        mb = (int)(rem >> 3);
        n0 = (grp * 8 + (rem & 7)) * (16 * NT);
    };

    // ---- DMA lane mapping: instruction i of a 16-row tile covers rows 8i .. 8i+7, 128 B each;
    // lane = 8*row + physical 16-B chunk; the logical chunk it fetches is swizzled by (row>>1)&7
    const int rr = lane >> 3, cp = lane & 7;
    unsigned xoff[MT][2];  // refreshed per work item (depends on the m-block)
    // ---- fragment read addresses (per stage, per k-step): lane (r, g) reads row r, logical chunk 4h+g
    const int fr = lane & 15, fg = lane >> 4;
    unsigned faddr[SK_STAGES][2];
#pragma unroll
    for (int s = 0; s < SK_STAGES; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            faddr[s][h] = ring + s * UNIT + fr * 128 + (((4 * h + fg) ^ ((fr >> 1) & 7)) << 4);
            asm volatile("" : "+v"(faddr[s][h]));  // keep as loop-invariant VGPRs
        }

    // issue cursor (runs two items ahead of the consume cursor)
    int it_tile = 0, it_u = 0, issued = 0;
    unsigned woff[NT][2];
    int cur_item = -1;
    int64_t it_n0 = 0;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        if (it_tile != cur_item) {  // new work item: per-lane W / X offsets (clamped at the ragged edges)
            cur_item = it_tile;
            int mb;
            decode(it_tile, it_n0, mb);
            if (it_n0 >= N) it_n0 = ((N - 1) / (16 * NT)) * (16 * NT);  // padding item of the last 8-tile group: harmless reload
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t r = nt * 16 + 8 * i + rr;
                    r = (it_n0 + r) < N ? r : (N - 1 - it_n0);
                    woff[nt][i] = (unsigned)(r * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t m = (int64_t)mb * (MT * 16) + mt * 16 + 8 * i + rr;
                    m = m < M ? m : M - 1;
                    xoff[mt][i] = (unsigned)(m * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
                }
        }
        const int64_t n0 = it_n0;
        const int u = wave + it_u * wpb;
        const int8_t *wb = uniform_ptr(w + n0 * K + (int64_t)u * 128);
        const int8_t *xb = uniform_ptr(x + (int64_t)u * 128);
        const unsigned dst = ring + stage * UNIT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) sk_dma16<SK_W_NT != 0>(wb, woff[nt][i], dst + nt * 2048 + i * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 2; ++i) sk_dma16(xb, xoff[mt][i], dst + (NT + mt) * 2048 + i * 1024);
        ++issued;
        if (++it_u == upt) {
            it_u = 0;
            ++it_tile;
        }
    };

    acc4_t acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (acc4_t){0, 0, 0, 0};

    int done = 0, c_u = 0, c_tile = 0;
    auto tile_end = [&]() __attribute__((always_inline)) {  // block-wide: sum the wpb partials of this channel tile, fused epilogue
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                red[((wave * NT + nt) * MT + mt) * 64 + lane] = acc[nt][mt];
                acc[nt][mt] = (acc4_t){0, 0, 0, 0};
            }
        __syncthreads();
        int64_t n0;
        int mb;
        decode(c_tile, n0, mb);
        for (int p = wave; p < NT * MT; p += wpb) {
            const int nt = p / MT, mt = p - nt * MT;
            acc4_t s = red[p * 64 + lane];
            for (int v = 1; v < wpb; ++v) s += red[(v * NT * MT + p) * 64 + lane];
            const int64_t m = (int64_t)mb * (MT * 16) + mt * 16 + fr, n = n0 + nt * 16 + 4 * fg;
            if (m < M && n < N) {
                const float sr = Epi::kHasRow ? epi.row(m) : 1.0f;
                v4f sc = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
                epi.cols(n, N, sc, bb);
                epi.store4(m, n, s, sr, sc, bb, N);
            }
        }
        __syncthreads();
        ++c_tile;
    };

    auto step = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(stage_tag)::value;
        if (issued < total) issue((S + 2) % SK_STAGES);
        const int newer = issued - done - 1;  // units issued after the one consumed now (0..2)
        if (newer >= 2) sk_wait_vm<2 * D>();
        else if (newer == 1) sk_wait_vm<D>();
        else sk_wait_vm<0>();
        v4i wf[NT][2], xf[MT][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt][h] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(faddr[S][h] + nt * 2048);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt][h] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(faddr[S][h] + (NT + mt) * 2048);
        }
        if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt][h], xf[mt][h], acc[nt][mt], 0, 0, 0);
        } else {  // fp8: one K = 128 block-scaled instruction (unit scales) over both halves of the unit
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = MMA::mma16(wf[nt][0], wf[nt][1], xf[mt][0], xf[mt][1], acc[nt][mt]);
        }
        ++done;
        if (++c_u == upt) {
            c_u = 0;
            tile_end();
        }
    };

    if (upt == 0) {  // more waves than K units: this wave only takes part in the reductions
        for (int t = 0; t < my_tiles; ++t) tile_end();
        return;
    }
    issue(0);
    if (total > 1) issue(1);
    while (true) {
        step(std::integral_constant<int, 0>{});
        if (done >= total) break;
        step(std::integral_constant<int, 1>{});
        if (done >= total) break;
        step(std::integral_constant<int, 2>{});
        if (done >= total) break;
    }
}

}  // namespace asq
