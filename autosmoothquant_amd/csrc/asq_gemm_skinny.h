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
//
// Fused prologue (XDT >= 0; reference layers/nn/linear.py:95-96 round / :289-292 divide, i.e. the per-tensor activation
// quantisers): X arrives as the module's FLOATING input (f32 / f16 / bf16) instead of pre-quantised int8.  The X part of a
// K unit is then 16 rows x 128 elements of XDT (256 or 512 B per row), DMA'd in full rows into the same wave-private ring
// with its own swizzle, read back as 2 (16-bit) or 4 (f32) ds_read_b128 per fragment and converted in registers with exactly
// the arithmetic of asq_quantize_act (QRound / QDiv of asq_quant.hip: same bits), so a decode-sized module forward is ONE
// launch: no int8 round trip through HBM, no second kernel.  The work item re-reads 2-4x more X bytes from L2, so the launcher
// uses it only while M x K x sizeof(XDT) is small (asq_api.hip).
#pragma once
#include <type_traits>

namespace asq {

constexpr int SK_STAGES = 3;

__device__ __forceinline__ void sk_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// x -> int8 with the per-tensor prologues' arithmetic (asq_quant.hip QRound / QDiv; bit-identical by construction):
//   mode ASQ_ACT_ROUND: clamp(rne(x));   ASQ_ACT_DIV: clamp(rne(XDT(x / s)))  -- the IEEE quotient via the 5-op sequence of
//   RowDivisor/QRowFast (y = RN(1/s) from the host) when that is provably exact, else the division itself.
struct SkXQuant {
    int mode;
    float s, y;
    bool fast;  // 2^-60 < s < 2^60
};
template <int XDT> __device__ __forceinline__ int sk_quant1(float x, const SkXQuant &q)
{
    if (q.mode == ASQ_ACT_ROUND) return quant_i8(x);
    float v;
    if (q.fast && absbits(x) < 0x5D800000u) {  // |x| < 2^60: finite, no overflow in the remainders
        const float q0 = __fmul_rn(x, q.y);
        const float q1 = __fmaf_rn(__fmaf_rn(-q.s, q0, x), q.y, q0);
        v = __fmaf_rn(__fmaf_rn(-q.s, q1, x), q.y, q1);
    } else {
        v = x / q.s;
    }
    return quant_i8(ElemT<XDT>::round(v));
}
// 16 consecutive k-elements of XDT (2 or 4 v4i) -> one int8 MFMA fragment
template <int XDT> __device__ __forceinline__ v4i sk_convert16(const v4i *src, const SkXQuant &q)
{
    v4i out;
    if constexpr (XDT == ASQ_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = sk_quant1<XDT>(__int_as_float(src[j][0]), q), b = sk_quant1<XDT>(__int_as_float(src[j][1]), q);
            const int c = sk_quant1<XDT>(__int_as_float(src[j][2]), q), d = sk_quant1<XDT>(__int_as_float(src[j][3]), q);
            out[j] = (int)((uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // output dword j = elements 4j .. 4j+3 = dwords 2j, 2j+1 of the 32 source bytes
            const uint32_t w0 = (uint32_t)src[j >> 1][(2 * j) & 3], w1 = (uint32_t)src[j >> 1][(2 * j + 1) & 3];
            const int a = sk_quant1<XDT>(ElemT<XDT>::load((uint16_t)(w0 & 0xFFFF)), q), b = sk_quant1<XDT>(ElemT<XDT>::load((uint16_t)(w0 >> 16)), q);
            const int c = sk_quant1<XDT>(ElemT<XDT>::load((uint16_t)(w1 & 0xFFFF)), q), d = sk_quant1<XDT>(ElemT<XDT>::load((uint16_t)(w1 >> 16)), q);
            out[j] = (int)((uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24));
        }
    }
    return out;
}
// LDS swizzle of the floating X rows (16-byte chunk index ^= sk_xsw(row)): conflict-free for the 16-lane service groups of
// ds_read_b128 ({0-3,12-15,20-27}, ...), which mix two k-groups whose logical chunks differ by 2 (16-bit rows, 16 chunks) or
// 4 (f32 rows, 32 chunks): row r itself for the former, its two 2-bit halves swapped for the latter.
template <int XDT> __device__ __forceinline__ int sk_xsw(int r)
{
    if constexpr (XDT == ASQ_F32) return ((r & 3) << 2) | ((r >> 2) & 3);
    else return r & 15;
}

#define SK_VM_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N> __device__ __forceinline__ void sk_wait_vm()
{
    static_assert(N >= 0 && N <= 63 && N % 2 == 0, "vmcnt is a 6-bit counter; unit sizes are even");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SK_VM_CASE(4); SK_VM_CASE(6); SK_VM_CASE(8); SK_VM_CASE(10); SK_VM_CASE(12); SK_VM_CASE(14); SK_VM_CASE(16); SK_VM_CASE(18);
    SK_VM_CASE(20); SK_VM_CASE(22); SK_VM_CASE(24); SK_VM_CASE(28); SK_VM_CASE(32); SK_VM_CASE(36); SK_VM_CASE(40); SK_VM_CASE(44); SK_VM_CASE(48);
}
#undef SK_VM_CASE

// MT = 16-row token tiles per m-block (1..4); NT = 16-channel tiles per work item (1 or 2).  NT = 2 halves the
// number of items that each re-read the X rows from L2 (the kernel's bound once N is large): the launcher
// picks it when there are still >= 256 items.
template <int XDT> struct SkX {  // geometry of the X part of a K unit
    static constexpr int ESZ = XDT < 0 ? 1 : (XDT == ASQ_F32 ? 4 : 2);  // bytes per element (int8 when pre-quantised)
    static constexpr int TILE = 2048 * ESZ;                             // 16 rows x 128 elements
    static constexpr int NI = 2 * ESZ;                                  // 1-KiB DMA instructions per 16-row tile
    static constexpr int C = 8 * ESZ;                                   // 16-byte chunks per row
};

template <class Epi, int MT, int NT, int XDT = -1>
__global__ void __launch_bounds__(512) gemm_i8_skinny(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                      int wpb, int mblocks, Epi epi, SkXQuant xqa)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using XG = SkX<XDT>;
    constexpr int UNIT = NT * 2048 + MT * XG::TILE;  // one 128-element K unit: NT x 16 W rows + MT x 16 X rows
    constexpr int D = 2 * NT + MT * XG::NI;          // DMA instructions per unit per wave
    static_assert(2 * D <= 48, "two units in flight must fit the counted vmcnt cases");
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
    unsigned xoff[MT][XG::NI];  // refreshed per work item (depends on the m-block)
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

    unsigned xfa[2][XG::ESZ];  // floating X: byte offset of (row fr, logical chunk of half h, piece j) inside an X tile
    if constexpr (XDT >= 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < XG::ESZ; ++j) {
                const int c = ((64 * h + 16 * fg) * XG::ESZ) / 16 + j;
                xfa[h][j] = (unsigned)(fr * (128 * XG::ESZ) + ((c ^ sk_xsw<XDT>(fr)) << 4));
                asm volatile("" : "+v"(xfa[h][j]));
            }
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
                for (int i = 0; i < XG::NI; ++i) {
                    if constexpr (XDT < 0) {
                        int64_t m = (int64_t)mb * (MT * 16) + mt * 16 + 8 * i + rr;
                        m = m < M ? m : M - 1;
                        xoff[mt][i] = (unsigned)(m * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
                    } else {  // instruction i: rows i*(64/C) + lane/C of the tile, physical chunk lane % C
                        const int r = i * (64 / XG::C) + lane / XG::C, pch = lane % XG::C;
                        int64_t m = (int64_t)mb * (MT * 16) + mt * 16 + r;
                        m = m < M ? m : M - 1;
                        xoff[mt][i] = (unsigned)(m * K * XG::ESZ) + (unsigned)((pch ^ sk_xsw<XDT>(r)) << 4);
                    }
                }
        }
        const int64_t n0 = it_n0;
        const int u = wave + it_u * wpb;
        const int8_t *wb = uniform_ptr(w + n0 * K + (int64_t)u * 128);
        const int8_t *xb = uniform_ptr(x + (int64_t)u * 128 * XG::ESZ);
        const unsigned dst = ring + stage * UNIT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 2; ++i) sk_dma16(wb, woff[nt][i], dst + nt * 2048 + i * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < XG::NI; ++i) sk_dma16(xb, xoff[mt][i], dst + NT * 2048 + mt * XG::TILE + i * 1024);
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
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (XDT < 0) {
                    xf[mt][h] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(faddr[S][h] + (NT + mt) * 2048);
                } else {
                    v4i raw[XG::ESZ];
#pragma unroll
                    for (int j = 0; j < XG::ESZ; ++j)
                        raw[j] = *(const __attribute__((address_space(3))) v4i *)(uintptr_t)(ring + S * UNIT + NT * 2048 + mt * XG::TILE + xfa[h][j]);
                    xf[mt][h] = sk_convert16<XDT>(raw, xqa);
                }
            }
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
