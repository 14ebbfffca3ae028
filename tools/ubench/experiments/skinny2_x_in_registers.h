// "skinny2": the weight-streaming kernel's decomposition of gemm_i8_skinny (asq_gemm_skinny.h) with the X operand moved OUT of LDS.
//
// Measured on gemm_i8_skinny / gemm_i8_wstream (profiles/r3_skinny_*): a CU's share of the weight stream is bounded by
// (weight bytes it keeps in flight) / (HBM latency under load, 2.5-3.5 us), and all of that in-flight storage is LDS.  In
// gemm_i8_skinny every wave's ring stage carries its X rows next to its W rows: at 32 rows X takes 50-67 % of the 144 KB ring, so a
// CU holds 48-64 KB of weight lines in flight -> 20-25 GB/s per CU, 3.3-4.1 TB/s on OPT-13B fc2.  The alternative that shares X between
// waves (gemm_i8_wstream: 8 waves x 16 channels, K split ACROSS blocks) streams at the HBM rate but pays >= 4 us of in-launch
// reduction (write-through drain + ticket + a dependent sc1 read), more than it saves below ~100 MB of weights.
//
// Here the decomposition stays reduction-free across blocks -- a work item is NT x 16 channels x all of K, the block's waves split K
// in 128-byte units and sum their partials through LDS -- but
//   * W alone goes through LDS: wave-private ring of D stages x NT x 16 rows x 128 B, LDS-DMA in full 128-B lines (8 rows per
//     instruction, XOR-swizzled on the global address, conflict-free ds_read_b128 fragments) -> D - 1 stages = up to 112 KB of weight
//     lines in flight per CU whatever M is;
//   * X goes straight into a REGISTER ring in MFMA-fragment shape (lane (r, g) loads row r, 16 B at k = 64h + 16g: one
//     global_load_dwordx4 per 16-row tile and k-half, both halves of a 128-B line back to back so the second hits the line the first
//     brought into L1).  X is L2-resident; it now costs TA bandwidth (M/16/NT of the weight bytes) but no LDS.
//   Both kinds of load are issued from inline asm and waited for with ONE counted s_waitcnt vmcnt per unit (in-order return); the
//   wait statement names the stage's X registers as read-write operands so no consumer is scheduled above it.
// The (item, unit) pairs of a wave form one continuous stream across work items, D - 1 units ahead; the last D - 1 units of a block
// are waited for with vmcnt(0) (nothing is left to overlap with).
//
// Requirements: K % 128 == 0, K <= 2^24, x / w 16-B aligned.  Ragged N, M: rows are clamped for loading and masked at the store.
#pragma once
#include <type_traits>

namespace asq {

template <int MT, int NT> struct Sk2Cfg {
    static constexpr int D = (MT <= 2 && NT == 1) ? 8 : 4;   // ring depth (units); VGPRs: D * MT * 8 for the X ring
    static constexpr int WUNIT = NT * 2048;                   // W bytes per unit per wave
    static constexpr int PERWAVE = D * WUNIT;                 // ring bytes per wave
    static constexpr int LDSWAVE = PERWAVE + NT * MT * 1024;  // + its reduction slot
    static constexpr int DCNT = 2 * MT + 2 * NT;              // vmcnt events per unit per wave
};

template <bool NT_> __device__ __forceinline__ void sk2_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    if constexpr (NT_)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}

// both k-halves of one 16-row X tile of one unit: 2 x global_load_dwordx4 (the compiler neither counts nor waits for them)
__device__ __forceinline__ void sk2_ldx(v4i &lo, v4i &hi, const int8_t *sbase, unsigned voff)
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:64"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(voff), "s"(sbase)
                 : "memory");
}

// s_waitcnt vmcnt(N) that names the stage's X fragment registers as read-write operands: nothing that consumes them is scheduled
// above the wait, and the compiler cannot treat them as available before it (cdna_hip_programming.md 5.7 form (ii)).
template <int N, int MT> __device__ __forceinline__ void sk2_wait(v4i *xf)
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    static_assert(MT >= 1 && MT <= 4, "1..4 row tiles");
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xf[0]), "+v"(xf[1]) : "n"(N) : "memory");
    if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]) : "n"(N) : "memory");
    if constexpr (MT == 3)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]) : "n"(N) : "memory");
    if constexpr (MT == 4)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7])
                     : "n"(N)
                     : "memory");
}

template <class F, int... S> __device__ __forceinline__ void sk2_for_stages(F &&f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

template <class Epi, int MT, int NT, bool WNT = false>
__global__ void __launch_bounds__(512) gemm_i8_skinny2(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K, int wpb,
                                                       int mblocks, Epi epi)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using C = Sk2Cfg<MT, NT>;
    constexpr int D = C::D, WUNIT = C::WUNIT, DC = C::DCNT;
    static_assert((D - 1) * DC <= 63, "counted wait exceeds the vmcnt field");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned ring = lds0 + wave * C::PERWAVE;
    using MMA = typename Epi::Mma;
    using acc4_t = typename MMA::acc4_t;
    acc4_t *const red = (acc4_t *)(lds + wpb * C::PERWAVE);  // [wpb][NT][MT][64], behind the rings
    typedef const __attribute__((address_space(3))) v4i *lds_v4i;

    const int nunits = (int)(K / 128);
    const int upt = nunits > wave ? (nunits - wave + wpb - 1) / wpb : 0;  // this wave's units per item
    const int64_t nitems = (((N + 16 * NT - 1) / (16 * NT) + 7) / 8) * 8 * mblocks;  // channel tiles padded to groups of 8 (m-blocks of a tile share an XCD)
    const int my_items = (int)((nitems - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total = my_items * upt;
    auto decode = [&](int i, int64_t &n0, int &mb) __attribute__((always_inline)) {
        const int64_t gidx = (int64_t)blockIdx.x + (int64_t)i * gridDim.x;
        const int64_t grp = gidx / (8 * mblocks), rem = gidx - grp * (8 * mblocks);
        mb = (int)(rem >> 3);
        n0 = (grp * 8 + (rem & 7)) * (16 * NT);
    };

    const int rr = lane >> 3, cp = lane & 7;   // DMA: lane = 8 * row + physical chunk
    const int fr = lane & 15, fg = lane >> 4;  // fragments: lane (row, 16-B k-group)
    unsigned faddr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) faddr[h] = ring + fr * 128 + (((4 * h + fg) ^ ((fr >> 1) & 7)) << 4);

    // ---- issue cursor
    int it_item = 0, it_u = 0, issued = 0, cur_item = -1;
    int64_t it_n0 = 0;
    unsigned woff[NT][2], xoff[MT];
    v4i xf[D][2 * MT];  // register ring: [stage][mt * 2 + h]
    auto issue = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(stage_tag)::value;
        if (it_item != cur_item) {  // new work item: per-lane offsets (rows clamped at the ragged edges)
            cur_item = it_item;
            int mb;
            decode(it_item, it_n0, mb);
            if (it_n0 >= N) it_n0 = ((N - 1) / (16 * NT)) * (16 * NT);  // padding item of the last 8-tile group: harmless reload, nothing stored
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t r = nt * 16 + 8 * i + rr;
                    r = (it_n0 + r) < N ? r : (N - 1 - it_n0);
                    woff[nt][i] = (unsigned)(r * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int64_t m = (int64_t)mb * (MT * 16) + mt * 16 + fr;
                m = m < M ? m : M - 1;
                xoff[mt] = (unsigned)(m * K) + (unsigned)(16 * fg);
            }
        }
        const int u = wave + it_u * wpb;
        const int8_t *xb = uniform_ptr(x + (int64_t)u * 128);
        const int8_t *wb = uniform_ptr(w + it_n0 * K + (int64_t)u * 128);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sk2_ldx(xf[S][2 * mt], xf[S][2 * mt + 1], xb, xoff[mt]);
        const unsigned dst = ring + S * WUNIT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) sk2_dma16<WNT>(wb, woff[nt][i], dst + nt * 2048 + i * 1024);
        ++issued;
        if (++it_u == upt) {
            it_u = 0;
            ++it_item;
        }
    };

    acc4_t acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (acc4_t){0, 0, 0, 0};

    int done = 0, c_u = 0, c_item = 0;
    auto item_end = [&]() __attribute__((always_inline)) {  // block-wide: sum the wpb partials of this item, fused epilogue
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                red[((wave * NT + nt) * MT + mt) * 64 + lane] = acc[nt][mt];
                acc[nt][mt] = (acc4_t){0, 0, 0, 0};
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the partials are in LDS (a plain __syncthreads() would also drain the DMA queue)
        __builtin_amdgcn_s_barrier();
        int64_t n0;
        int mb;
        decode(c_item, n0, mb);
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
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // the slots are free again
        ++c_item;
    };

    auto step = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(stage_tag)::value;
        if (issued < total) issue(std::integral_constant<int, (S + D - 1) % D>{});
        if (issued - done - 1 == D - 1) sk2_wait<(D - 1) * DC, MT>(xf[S]);  // steady state: D - 1 younger units stay in flight
        else sk2_wait<0, MT>(xf[S]);                                          // the block's last D - 1 units: nothing left to overlap with
        v4i wf[NT][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt][h] = *(lds_v4i)(uintptr_t)(faddr[h] + S * WUNIT + nt * 2048);
        if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt][h], xf[S][2 * mt + h], acc[nt][mt], 0, 0, 0);
        } else {  // fp8: one K = 128 block-scaled instruction (unit scales) over both halves of the unit
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = MMA::mma16(wf[nt][0], wf[nt][1], xf[S][2 * mt], xf[S][2 * mt + 1], acc[nt][mt]);
        }
        ++done;
        if (++c_u == upt) {
            c_u = 0;
            item_end();
        }
    };

    if (upt == 0) {  // more waves than K units: this wave only takes part in the reductions
        for (int t = 0; t < my_items; ++t) item_end();
        return;
    }
    const auto stages = std::make_integer_sequence<int, D>{};
    sk2_for_stages([&](auto tag) __attribute__((always_inline)) {  // prologue: D - 1 units in flight
        constexpr int S = decltype(tag)::value;
        if (S < D - 1 && S < total) issue(tag);
    }, stages);
    while (done < total)
        sk2_for_stages([&](auto tag) __attribute__((always_inline)) {
            if (done < total) step(tag);
        }, stages);
}

}  // namespace asq
