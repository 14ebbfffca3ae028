// "p8": 256(m) x 256(n) x 128(k) INT8 MFMA GEMM with a phase-staggered, deep-prefetch
// schedule.  Included by asq_gemm.hip (needs its epilogue functors and helpers).
//
// Work split: 8 waves = 2 (m) x 4 (n); wave (wm, wn) owns out[wm*128 .. +128][wn*64 .. +64]
// = 2x2 "quadrants" of 64(m) x 32(n), each 2 MFMA tiles of 32x32 (v_mfma_i32_32x32x32_i8,
// W = matrix-core A operand, X = B operand; see asq_gemm.hip header).
//
// One K-tile (128 k-bytes) is FOUR phases, one quadrant each (8 MFMAs = 256 MFMA cycles):
//     P1 (m-half 0, n-half 0)   reads X-even (8 x ds_read_b128) + W-even (4)
//     P2 (m-half 0, n-half 1)   reads W-odd (4)
//     P3 (m-half 1, n-half 1)   reads X-odd (8)
//     P4 (m-half 1, n-half 0)   reads nothing (W-even fragments stay in VGPRs)
// Each phase is   { issue 2 LDS-DMAs ; ds_reads ; counted vmcnt ; s_barrier ;
//                   lgkmcnt(0) ; 8 MFMAs ; s_barrier }.
// The wm=1 waves run ONE BARRIER behind the wm=0 waves, so on every SIMD (one wave of each
// group) one wave is in its MFMA segment while the other issues loads: the matrix pipe
// never waits for LDS or the barrier.
//
// The load segment contains NO VALU instruction (measured: a VALU op issued beside the
// partner wave's prio-1 MFMA stream costs 100-300 cycles): DMA addresses are
// SGPR base (+k, advanced with SALU) + a loop-invariant 32-bit VGPR offset, the LDS
// destination goes through M0, and every ds_read address is a loop-invariant VGPR plus an
// immediate (the K loop is unrolled by 2 so the LDS stage is a compile-time constant).
//
// LDS = ring of 8 "units" of 16 KiB (2 K-tiles x {X-even, W-even, W-odd, X-odd}); a unit is
// the 128 operand rows all 8 waves need for one phase:
//     X-even: tile rows [0,64) u [128,192)      X-odd: [64,128) u [192,256)
//     W-even: rows 64*wn + [0,32)               W-odd: rows 64*wn + [32,64)
// Units are filled by global_load_lds (16 B/lane, lane-linear 1 KiB per wave-instruction,
// 2 per wave per unit) in consumption order, FOUR phases ahead of their first read:
// phase g issues unit g+4.  Waits are counted, never 0 in the loop: after a phase's issue,
// `s_waitcnt vmcnt(4)` retires exactly the units the NEXT phase reads (2 units = 4 DMAs may
// stay in flight); the barrier after it makes every wave's share visible (RAW), and a unit's
// slot is re-filled >= 2 phases after its last ds_read (WAR).
// Unit image: [128 rows][8 x 16-B chunks], chunk ^= (row>>1)&7 -> conflict-free ds_read_b128
// (SQ_LDS_BANK_CONFLICT = 0 measured); the swizzle is applied to the DMA's per-lane global
// source address.
//
// Past the last K-tile the prefetcher re-reads the last tile (clamped k) into dead slots,
// keeping every vmcnt count uniform; all DMAs are drained before the epilogue.
#pragma once
#include <type_traits>

namespace asq {

constexpr int P8_UNIT = 128 * 128;        // 16 KiB
constexpr int P8_STAGE = 4 * P8_UNIT;     // one K-tile: 64 KiB
constexpr int P8_LDS_BYTES = 2 * P8_STAGE;
constexpr int P16_LDS_BYTES = P8_LDS_BYTES + 4096;   // gemm_i8_p16 on offset operands: + the tile's 256 row and 256 column pairs behind the ring
// grouped launches (the scheduler at the top of gemm_i8_p8)
constexpr int P8_GROUPED_SCAN_MAX = 64;     // groups a block may scan twice (tile total, then its own tile)
constexpr int P8_CUS_PER_XCD = 32;          // MI355X: 256 CUs in 8 XCDs; one 128-KiB-LDS block per CU
constexpr int P8_TAIL_SPLIT_MAX = 8;        // K pieces of a tail tile
#ifndef P8_FIX_BATCH
#define P8_FIX_BATCH 8                      // 16-B loads per lane in flight while the last piece sums the others' images
#endif
constexpr int P8_TAIL_MIN_KTILES = 8;       // ... each at least this many 128-byte K tiles
constexpr size_t P8_GROUPED_WS_BYTES = (size_t)8 * P8_CUS_PER_XCD * 256 * 256 * 4;  // 256 register images of 256 KiB

#define P8_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- probe support.  Production TUs never define ASQ_P8_PROBE: the ABL template parameter is then pinned to 0 and every
// probe macro below is empty (no device globals, no s_memtime).  tools/ubench/{p8_probe,clock_probe}.hip define it
// before including the sources and own the stamp arrays.
//   ABL 1 = no LDS-DMA, 2 = no fragment ds_reads, 4 = no MFMA, 8 = no barriers, 16 = every fragment read issued twice,
//       32 = timing stamps of block 0 waves 0/4, 64 = no s_setprio, 128 = per-block stamps,
//       256 / 512 / 768 = epilogue stores nt / sc1 / sc0 sc1 instead of P8_STORE_DEFAULT.
//       2048 = the round-2 staged epilogue on interior tiles too (A/B against epilogue_wave_rows).
#ifdef ASQ_P8_PROBE   // a probe TU under tools/ubench: the stamping hooks live there, not in the product
#include "../../tools/ubench/probe_hooks.h"
#else
#define P8_ABL_OK(ABL) ((ABL) == 0)
#define P8_BLK(i) do { } while (0)
#define P8_BLK_RT(i) do { } while (0)
#define P8_STAMP(i) do { } while (0)
#define P8_ACCUM(ph) do { } while (0)
#define P8_PROBE_TIMERS() do { } while (0)
#define P8_PROBE_DUMP_TIMERS() do { } while (0)
#define P8_PROBE_END() do { } while (0)
#define P8_PROBE_END_WHERE(tile_m, tile_n) do { } while (0)
#define P8_PROBE_PIN_BASE(p) do { } while (0)
#define P8H_BLK(i) do { } while (0)
#define P8H_PROBE_END(tile_m, tile_n) do { } while (0)
#define P4_BLK(i) do { } while (0)
#define P4_BLK_RT(i) do { } while (0)
#define P4_PROBE_END() do { } while (0)
#endif
#ifndef P8_STORE_DEFAULT
#define P8_STORE_DEFAULT 0   // cache policy of the staged epilogue's global stores (store16_policy)
#endif
#define P8_STORE_POLICY(ABL) ((((ABL) >> 8) & 3) ? (((ABL) >> 8) & 3) : P8_STORE_DEFAULT)
#define P8_BAR() do { if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier(); } while (0)
#define P8_PRIO(v) do { if constexpr (!(ABL & 64)) __builtin_amdgcn_s_setprio(v); } while (0)

template <int ABL, class MMA> __device__ __forceinline__ typename MMA::acc_t p8_mfma(const v4i &a, const v4i &b, const typename MMA::acc_t &c)
{
    if constexpr (ABL & 4) {
        asm volatile("" ::"v"(a), "v"(b));  // keep the fragment loads alive
        return c;
    } else {
        return MMA::mma(a, b, c);
    }
}
template <int ABL, class MMA>
__device__ __forceinline__ typename MMA::acc_t p8_mfma2(const v4i &a0, const v4i &a1, const v4i &b0, const v4i &b1, const typename MMA::acc_t &c)
{
    if constexpr (ABL & 4) {
        asm volatile("" ::"v"(a0), "v"(a1), "v"(b0), "v"(b1));  // keep the fragment loads alive
        return c;
    } else {
        return MMA::mma2(a0, a1, b0, b1, c);
    }
}
typedef const __attribute__((address_space(3))) v4i *p8_lds_v4i;
template <int ABL> __device__ __forceinline__ v4i p8_ldfrag(unsigned lds_addr, int lane)
{
    if constexpr (ABL & 2) return (v4i){lane, 1, 2, 3};
    else {
        if constexpr (ABL & 16) {  // probe: every fragment read issued twice (sensitivity of the kernel to LDS read traffic)
            v4i dup;
            asm volatile("ds_read_b128 %0, %1" : "=v"(dup) : "v"(lds_addr) : "memory");
            asm volatile("" ::"v"(dup));
        }
        return *(p8_lds_v4i)(uintptr_t)lds_addr;
    }
}

#ifndef P8_DMA_MOD
#define P8_DMA_MOD ""   // cache policy of the operand DMAs (" nt", " sc1", " sc0 sc1" measured in round 3: profiles/r3_dma_policy_ab.txt)
#endif
// LDS-DMA, 16 B per lane: global address = SGPR-pair base + 32-bit VGPR offset (no VALU address
// math), LDS destination = M0 (wave-uniform) + lane*16.  hipcc does not count asm memory
// operations: every consumer is ordered by this kernel's own counted s_waitcnt vmcnt + barrier.
__device__ __forceinline__ void p8_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    P8_PROBE_PIN_BASE(sbase);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" P8_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// L16: the matrix work on v_mfma_i32_16x16x64_i8 instead of 32x32x32 (see asq_gemm_p16.h: less energy per MAC, +8-10 % under the socket power limit; int8 only).
// Same units, DMAs, phases and barriers; a quadrant is 2 k-steps x {4 token tiles x 2 channel tiles} of 16 x 16, the accumulators are acc16[m-half][n-half][4][2].
template <class Epi, int ABL = 0, bool GRP = false, bool L16 = false>   // GRP: the grouped launch (goffs != null); the plain instantiation carries none of its code
__global__ void __launch_bounds__(512, 2) gemm_i8_p8(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N,
                                                     int64_t K, int tiles_m, int tiles_n, int ksplit, const int *__restrict__ goffs, int ngroups,
                                                     char *gws, Epi epi_in, OffsetArgs off)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    static_assert(P8_ABL_OK(ABL), "probe variants need -DASQ_P8_PROBE (tools/ubench)");

    P8_BLK_RT(6);
    P8_BLK(0);
    // split-K (ksplit > 1, int32 slabs only): logical id = split * ntiles + tile, so the blocks an XCD
    // receives share one K range and neighbouring tiles (operand panels stay L2-local)
    // Grouped mode (goffs != null; Mixtral-style experts): x rows are sorted by group, goffs[0..ngroups]
    // are the row offsets (device int32), w is [ngroups][N][K].  The grid is a host-side upper bound; each block finds its
    // (group, tile) by a scalar scan, surplus blocks exit.
    //   ngroups <= P8_GROUPED_SCAN_MAX: the block first counts the launch's tiles and the remap runs over that count, not over the
    //   grid: every XCD (blocks b = 8 * slot + xcd) gets an equal share, instead of the last XCDs holding the grid's surplus blocks
    //   and idling.  Tiles come in two kinds: FULL (256 rows of a group) and HALF -- a group's last tile when it has <= 128 rows;
    //   its rows [0,64) / [64,128) go to the two wave groups' first row halves and the K loop skips phases P3 / P4 (half the matrix
    //   work instead of a full tile of padding).  An XCD runs its full tiles first, then its half tiles (long jobs first).
    //   With a workspace (gws) the tiles an XCD would run in its last, less than half full round (32 CUs, one block each) are split
    //   along K into S = 32 / tail pieces that fill the round; the pieces leave their accumulators in the workspace as register
    //   images and the last one to arrive (ticket in the workspace header, as in asq_gemm_wstream.h) adds them and runs the
    //   epilogue: exact for int32.
    constexpr int GM = 4;
    constexpr bool MMA_kIsInt_ = Epi::Mma::kIsInt;
    int split = 0, id, grp = 0, ksp = ksplit, moff = 0;
    bool half_rt = false;
    int tsplit = 0, tsplits = 1, tpart0 = 0, ttail = 0, tticket = 0;  // K split of a tail tile inside a grouped launch
    int64_t m_base = 0;
    const int64_t M_all = M;
    if constexpr (GRP) {
        int gid, e = 0, r0 = 0, r1 = 0, tm_e = 0;
        const bool sched = ngroups <= P8_GROUPED_SCAN_MAX;
        if (sched) {
            int F = 0, Hh = 0;  // tiles of the two kinds
            for (int g = 0; g < ngroups; ++g) {
                const int m = goffs[g + 1] - goffs[g];
                const int h = (MMA_kIsInt_ && m > 0 && ((m - 1) & 255) < 128) ? 1 : 0;  // (int8 only: the guarded phases cost the fp8 instantiations spills)
                F += (((m + 255) >> 8) - h) * tiles_n;
                Hh += h * tiles_n;
            }
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            const int qf = F >> 3, rf = F & 7, qh = Hh >> 3, rh = Hh & 7, xr = 7 - xcd;  // (the odd full tiles go to the low XCDs, the odd half tiles to the high ones)
            const int cF = qf + (xcd < rf ? 1 : 0), baseF = xcd < rf ? xcd * (qf + 1) : rf * (qf + 1) + (xcd - rf) * qf;
            const int cH = qh + (xr < rh ? 1 : 0), baseH = xr < rh ? xr * (qh + 1) : rh * (qh + 1) + (xr - rh) * qh;
            // The XCD's last, incomplete round: tf full tiles left over by the complete rounds of 32 and the half tiles behind them
            // (or, with no full tile left over, the half tiles left over by THEIR complete rounds); in half-tile units of work:
            const int tf = cF & (P8_CUS_PER_XCD - 1), th = tf > 0 ? cH : (cH & (P8_CUS_PER_XCD - 1));
            const int Wt = 2 * tf + th;
            int SF = 1, SH = 1;  // K pieces of a tail full / half tile: equal durations, at most 32 pieces in all
            if (MMA_kIsInt_ && gws != nullptr && Wt > 0 && Wt <= P8_CUS_PER_XCD) {
                SH = Wt <= P8_CUS_PER_XCD / 4 ? 4 : Wt <= P8_CUS_PER_XCD / 2 ? 2 : 1;
                SF = 2 * SH;
                const int by_k = (int)(K / 128) / P8_TAIL_MIN_KTILES;
                SF = SF < by_k ? SF : by_k;
                SH = SH < by_k ? SH : by_k;
                if (SF < 2) SF = 1;
                if (SH < 2) SH = 1;
            }
            const bool cut = SF > 1 || SH > 1;
            const int body = cF + cH - (cut ? tf + th : 0);
            int seq;  // position in this XCD's sequence [full tiles | half tiles]
            if (slot < body) {
                seq = slot;
            } else {
                int t = slot - body, S, ti, kind0, cnt;
                if (!cut) return;  // surplus block (uniform for the whole block, before any barrier)
                if (t < tf * SF) {
                    S = SF, cnt = tf, kind0 = 0, seq = cF - tf;
                } else {
                    t -= tf * SF;
                    if (t >= th * SH) return;
                    S = SH, cnt = th, kind0 = tf * SF, seq = cF + cH - th;
                }
                ti = t % cnt;
                seq += ti;
                if (S > 1) {
                    tsplit = t / cnt;
                    tsplits = S;
                    ksp = S;
                    ttail = cnt;
                    tpart0 = xcd * P8_CUS_PER_XCD + kind0 + ti;          // image of (tile ti, piece s) at tpart0 + s * cnt
                    tticket = xcd * P8_CUS_PER_XCD + (kind0 ? tf : 0) + ti;
                }
            }
            half_rt = seq >= cF;
            gid = half_rt ? baseH + (seq - cF) : baseF + seq;
        } else {
            gid = xcd_remap(blockIdx.x, gridDim.x);
        }
        for (; e < ngroups; ++e) {
            r0 = goffs[e];
            r1 = goffs[e + 1];
            const int m = r1 - r0;
            const int h = (MMA_kIsInt_ && sched && m > 0 && ((m - 1) & 255) < 128) ? 1 : 0;
            tm_e = ((m + 255) >> 8) - h;                       // full tile rows of this group
            const int cnt = (half_rt ? h : tm_e) * tiles_n;
            if (gid < cnt) break;
            gid -= cnt;
        }
        if (e == ngroups) return;  // uniform for the whole block, before any barrier
        id = gid;
        tiles_m = half_rt ? 1 : tm_e;
        moff = half_rt ? tm_e : 0;    // the half tile is the group's last tile row
        m_base = r0;
        M = r1;  // rows >= r1 belong to the next group: clamp loads, mask stores
        w += (int64_t)e * N * K;
        grp = e;
        if (tsplits > 1) split = tsplit;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int lid = xcd_remap(blockIdx.x, nwg * ksplit);
        split = lid / nwg;
        id = lid - split * nwg;
    }
    const bool half = GRP && MMA_kIsInt_ ? half_rt : false;
    unsigned long long ws_magic = 0;
    if (tsplits > 1) ws_magic = *(const volatile unsigned long long *)gws;  // checked at the ticket (asq_workspace_init wrote it)
    const Epi epi = epi_in.rebased(grp, tsplits > 1 ? 0 : split, M_all, N);
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = m_base + (int64_t)(tile_m + moff) * 256, n0 = (int64_t)tile_n * 256;

    // OFFSET operands (OffsetArgs; grouped int8 launches on the 16 x 16 x 64 instruction only -- asq_linear_w8a8_grouped_off): as in gemm_i8_p16 the K loop runs
    // on x + cx[m] / w + cw[n] unchanged and the exact product is recovered in front of the epilogue; the tile's {cx, xsum'} / {cw, wsum} pairs are fetched by one
    // 8-byte load per thread now and parked behind the ring.  Rows of other groups (>= M = this group's end) are clamped: their outputs are masked anyway.
    constexpr bool kOffs = GRP && L16 && MMA_kIsInt_ && Epi::kOutBytes == 2 && !Epi::kHasCol;   // (grouped launches carry per-group scalars, never a column-scale vector: those instantiations are not launched)
    const bool offs = kOffs && off.row != nullptr;   // (block-uniform)
    v2i opair = {0, 0};
    if constexpr (kOffs) if (offs) {
        const int i = tid & 255;
        const int64_t idx = tid < 256 ? (m0 + i < M ? m0 + i : M - 1) : (int64_t)grp * N + (n0 + i < N ? n0 + i : N - 1);
        const int32_t *src = (tid < 256 ? off.row : off.col) + 2 * idx;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(opair) : "v"(src) : "memory");
    }

    // ---- DMA sources: uniform tile base (SGPR pair, + k advanced per K-tile) + 32-bit lane offset.
    // This wave fills row-groups 2*wave, 2*wave+1 (8 rows each) of every unit.
    const int nt_all = (int)(K / 128);
    const int kt0 = (int)((int64_t)nt_all * split / ksp), kt1 = (int)((int64_t)nt_all * (split + 1) / ksp);
    const int8_t *const xbase = uniform_ptr(x + m0 * K + (int64_t)kt0 * 128);  // wave-uniform, SGPR pair
    const int8_t *const wbase = uniform_ptr(w + n0 * K + (int64_t)kt0 * 128);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;  // last valid local row
    unsigned voff[4][2];  // [kind][i]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ru = (wave * 2 + i) * 8 + (lane >> 3);          // row within the unit, 0..127
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);  // swizzled source chunk
        int64_t rxe = half ? ru : (ru >> 6) * 128 + (ru & 63), rxo = rxe + 64;  // local tile rows (half tile: rows [0,128) feed the first row halves)
        int64_t rwe = (ru >> 5) * 64 + (ru & 31), rwo = rwe + 32;
        rxe = rxe < mrem ? rxe : mrem;
        rxo = rxo < mrem ? rxo : mrem;
        rwe = rwe < nrem ? rwe : nrem;
        rwo = rwo < nrem ? rwo : nrem;
        voff[0][i] = (unsigned)(rxe * K) + cb;
        voff[1][i] = (unsigned)(rwe * K) + cb;
        voff[2][i] = (unsigned)(rwo * K) + cb;
        voff[3][i] = (unsigned)(rxo * K) + cb;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned dma_dst = lds0 + wave * 2048;  // + stage*P8_STAGE + kind*P8_UNIT + i*1024

    // ---- fragment read addresses: one VGPR per (stage, operand, k-substep); the rest is an immediate
    const int frow = lane & 31, sw = (frow >> 1) & 7, hi = lane >> 5;
    unsigned xb[2][4], wbp[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = lds0 + frow * 128 + ((((ks * 2 + hi) ^ sw)) << 4) + s * P8_STAGE;
            xb[s][ks] = off + wm * 64 * 128;   // X units: this wave's 64 rows
            wbp[s][ks] = off + wn * 32 * 128;  // W units: this wave's 32 rows
            // opaque to the optimiser: keep these 16 addresses in VGPRs instead of re-deriving them
            // with VALU adds inside the load segments
            asm volatile("" : "+v"(xb[s][ks]), "+v"(wbp[s][ks]));
        }

    unsigned xb16[2][2], wbp16[2][2];   // L16: one VGPR per (stage, operand, k-step of 64); + 2048 * tile (16 rows) + the unit as immediates
    if constexpr (L16) {
        const int t16 = lane & 15, q16 = lane >> 4, sw16 = (t16 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned off = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw16)) << 4) + s * P8_STAGE;
                xb16[s][kk] = off + wm * 64 * 128;
                wbp16[s][kk] = off + wn * 32 * 128;
                asm volatile("" : "+v"(xb16[s][kk]), "+v"(wbp16[s][kk]));
            }
    }

    using MMA = typename Epi::Mma;
    using acc_t = typename MMA::acc_t;
    static_assert(!L16 || MMA::kIsInt, "the 16 x 16 x 64 form is the int8 instruction");
    acc_t acc[2][2][2];  // [m-half][n-half][j]
    v4i acc16[2][2][4][2];  // L16: [m-half][n-half][token tile][channel tile]
    if constexpr (L16) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc16[a][b][c][d] = (v4i){0, 0, 0, 0};
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][b][c] = (acc_t){0};

    const int nt = kt1 - kt0;  // K-tiles of this block (>= 1)
    const int klast = (nt - 1) * 128;

    auto issue = [&](int kind, int stage, int k0) {
        if constexpr (ABL & 1) return;
        const int8_t *b = ((kind == 0 || kind == 3) ? xbase : wbase) + k0;  // SALU
#pragma unroll
        for (int i = 0; i < 2; ++i) p8_dma16(b, voff[kind][i], dma_dst + stage * P8_STAGE + kind * P8_UNIT + i * 1024);
    };

    // ---- prologue: K-tile 0 entirely (units 0..3), wait for the two units P1 reads
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) issue(kind, 0, 0);
    P8_WAIT_VM(4);
    typedef __attribute__((address_space(3))) v2i *lds_v2i;
    typedef __attribute__((address_space(3))) v4i *lds_v4i_;
    typedef __attribute__((address_space(3))) int *lds_i32;
    const unsigned obase = lds0 + P8_LDS_BYTES;   // offset launches: [256 row words: (-cx) << 24 | (-xsum') & 0xFFFFFF][256 column pairs {cw, wsum}]
    if constexpr (kOffs) if (offs) {
        asm volatile("" : "+v"(opair));   // (keeps every use of the loaded pair behind the wait above: in-order return, the pair is older than the DMAs)
        if (tid < 256) *(lds_i32)(uintptr_t)(obase + tid * 4) = (int)(((unsigned)(-opair[0]) << 24) | ((unsigned)(-opair[1]) & 0xFFFFFFu));
        else *(lds_v2i)(uintptr_t)(obase + 1024 + (tid - 256) * 8) = opair;
    }
    P8_BAR();
    if (wm == 1) P8_BAR();  // stagger: the wm=1 group runs one barrier behind
    P8_BLK(1);

    v4i xf[2][4], wa[4], wb[4];
    v4i xf16[4][2], wa16[2][2], wb16[2][2];   // L16: [tile][k-step]
    // L16: the 16 matrix instructions of a quadrant; the activation fragment stays for two instructions, the weight fragments alternate (asq_gemm_p16.h)
    auto quadrant16 = [&](v4i (&A)[4][2], const v4i (&wf)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it) A[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[it][kk], xf16[jt][kk], A[jt][it], 0, 0, 0);
    };
    auto ld16 = [&](unsigned a) { return *(p8_lds_v4i)(uintptr_t)a; };
    P8_PROBE_TIMERS();

    // one K-tile at LDS stage S (compile-time), prefetching K-tile (t+1) into stage S^1
    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = S ^ 1;
        int kn = (t + 1) * 128;  // SALU: s_min_i32
        kn = kn < klast ? kn : klast;

        // ---------------- P1: (m-half 0, n-half 0)
        P8_STAMP(0);
        issue(0, NS, kn);
        P8_STAMP(1);
        if constexpr (L16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 2; ++it) wa16[it][kk] = ld16(wbp16[S][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) xf16[jt][kk] = ld16(xb16[S][kk] + 0 * P8_UNIT + jt * 2048);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wa[ks] = p8_ldfrag<ABL>(wbp[S][ks] + 1 * P8_UNIT, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xf[j][ks] = p8_ldfrag<ABL>(xb[S][ks] + 0 * P8_UNIT + j * 4096, lane);
        }
        P8_STAMP(2);
        P8_WAIT_VM(4);
        P8_STAMP(3);
        P8_BAR();
        P8_STAMP(4);
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(5);
        P8_PRIO(1);
if constexpr (L16) {
            quadrant16(acc16[0][0], wa16);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][0][j] = p8_mfma<ABL, MMA>(wa[ks], xf[j][ks], acc[0][0][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][0][j] = p8_mfma2<ABL, MMA>(wa[kp], wa[kp + 1], xf[j][kp], xf[j][kp + 1], acc[0][0][j]);
        }
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(6);
        P8_BAR();
        P8_STAMP(7);
        P8_ACCUM(0);

        // ---------------- P2: (m-half 0, n-half 1)
        P8_STAMP(0);
        issue(1, NS, kn);
        P8_STAMP(1);
        if constexpr (L16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 2; ++it) wb16[it][kk] = ld16(wbp16[S][kk] + 2 * P8_UNIT + it * 2048);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wb[ks] = p8_ldfrag<ABL>(wbp[S][ks] + 2 * P8_UNIT, lane);
        }
        P8_STAMP(2);
        P8_WAIT_VM(4);
        P8_STAMP(3);
        P8_BAR();
        P8_STAMP(4);
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(5);
        P8_PRIO(1);
if constexpr (L16) {
            quadrant16(acc16[0][1], wb16);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][1][j] = p8_mfma<ABL, MMA>(wb[ks], xf[j][ks], acc[0][1][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][1][j] = p8_mfma2<ABL, MMA>(wb[kp], wb[kp + 1], xf[j][kp], xf[j][kp + 1], acc[0][1][j]);
        }
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(6);
        P8_BAR();
        P8_STAMP(7);
        P8_ACCUM(1);

        // ---------------- P3: (m-half 1, n-half 1)
        P8_STAMP(0);
        issue(2, NS, kn);
        P8_STAMP(1);
        if (!half) {
            if constexpr (L16) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) xf16[jt][kk] = ld16(xb16[S][kk] + 3 * P8_UNIT + jt * 2048);
            } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xf[j][ks] = p8_ldfrag<ABL>(xb[S][ks] + 3 * P8_UNIT + j * 4096, lane);
            }
        }
        P8_STAMP(2);
        P8_STAMP(3);
        P8_BAR();
        P8_STAMP(4);
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(5);
        P8_PRIO(1);
        if (!half) {  // (a half tile has no second row half: phases P3 / P4 only keep the DMA / barrier cadence)
if constexpr (L16) {
            quadrant16(acc16[1][1], wb16);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][1][j] = p8_mfma<ABL, MMA>(wb[ks], xf[j][ks], acc[1][1][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][1][j] = p8_mfma2<ABL, MMA>(wb[kp], wb[kp + 1], xf[j][kp], xf[j][kp + 1], acc[1][1][j]);
        }
        }
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(6);
        P8_BAR();
        P8_STAMP(7);
        P8_ACCUM(2);

        // ---------------- P4: (m-half 1, n-half 0) -- W-even fragments are still in wa[]
        P8_STAMP(0);
        issue(3, NS, kn);
        P8_STAMP(1);
        P8_STAMP(2);
        P8_WAIT_VM(4);
        P8_STAMP(3);
        P8_BAR();
        P8_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(5);
        P8_PRIO(1);
        if (!half) {
if constexpr (L16) {
            quadrant16(acc16[1][0], wa16);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][0][j] = p8_mfma<ABL, MMA>(wa[ks], xf[j][ks], acc[1][0][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][0][j] = p8_mfma2<ABL, MMA>(wa[kp], wa[kp + 1], xf[j][kp], xf[j][kp + 1], acc[1][0][j]);
        }
        }
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(6);
        P8_BAR();
        P8_STAMP(7);
        P8_ACCUM(3);
    };

    int t = 0;
    for (; t + 1 < nt; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nt) ktile(std::integral_constant<int, 0>{}, t);

    P8_PROBE_DUMP_TIMERS();
    P8_BLK(2);
    P8_WAIT_VM(0);          // drain the dead prefetches before LDS is released
    if (wm == 0) P8_BAR();  // balance the stagger barrier

    if constexpr (GRP && MMA::kIsInt) if (tsplits > 1) {  // block-uniform: a K piece of a tail tile of a grouped launch (see the top of the kernel; int32 only)
        typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
        typedef decltype(acc[0][0][0][0]) elem_ref;
        typedef std::remove_cv_t<std::remove_reference_t<elem_ref>> elem_t;
        typedef elem_t v4e_ __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gws + WS_HEADER_BYTES, 0, 0x7FFFFFFF, 0x00020000);
        constexpr int PART = 256 * 256 * 4;   // one register image: 32 x (512 lanes x 16 B)
        const int my = (tpart0 + tsplit * ttail) * PART + tid * 16;
        // (offset in the VGPR, soffset 0 and a wait state after the stores: see the note on buffer stores in asq_gemm_wstream.h)
#pragma unroll
        for (int i = 0; i < 32; ++i) {   // (the image is the accumulator registers in their own order: the layout does not matter)
            v4e_ v;
            if constexpr (L16) {
                v = __builtin_bit_cast(v4e_, acc16[i >> 4][(i >> 3) & 1][(i >> 1) & 3][i & 1]);
            } else {
                const acc_t &A = acc[i >> 4][(i >> 3) & 1][(i >> 2) & 1];
                const int j = (i & 3) * 4;
                v = (v4e_){A[j], A[j + 1], A[j + 2], A[j + 3]};
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, v), rsrc, my + i * 8192, 0, 16 /* sc1: write-through */);
        }
        asm volatile("s_nop 1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's image is in memory before the block takes its ticket
        unsigned *const flag = (unsigned *)lds;  // (the ring is dead)
        if (tid == 0) {
            unsigned *const tk = (unsigned *)gws + 4 + tticket;
            const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old >= (unsigned)tsplits || ws_magic != WS_MAGIC) __builtin_trap();  // header not initialised / workspace shared by concurrent launches
            if (old == (unsigned)tsplits - 1) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch / replay
            *flag = old == (unsigned)tsplits - 1 ? 1u : 0u;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads();  // (the staged epilogue reuses this LDS)
        if (!last) return;
        for (int sp = 0; sp < tsplits; ++sp) {
            if (sp == tsplit) continue;
            const int src = (tpart0 + sp * ttail) * PART + tid * 16;
#pragma unroll
            for (int i0 = 0; i0 < 32; i0 += P8_FIX_BATCH) {   // (32 VGPRs of loads in flight beside the 128 accumulators; 16 spills)
                v4u_ v[P8_FIX_BATCH];
#pragma unroll
                for (int u = 0; u < P8_FIX_BATCH; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, src + (i0 + u) * 8192, 0, 16 /* sc1 */);
#pragma unroll
                for (int u = 0; u < P8_FIX_BATCH; ++u) {
                    const int i = i0 + u;
                    const v4e_ a = __builtin_bit_cast(v4e_, v[u]);
                    if constexpr (L16) {
                        v4i &A = acc16[i >> 4][(i >> 3) & 1][(i >> 1) & 3][i & 1];
                        A += __builtin_bit_cast(v4i, a);
                    } else {
                        acc_t &A = acc[i >> 4][(i >> 3) & 1][(i >> 2) & 1];
                        const int j = (i & 3) * 4;
                        A[j] += a[0];
                        A[j + 1] += a[1];
                        A[j + 2] += a[2];
                        A[j + 3] += a[3];
                    }
                }
            }
        }
    }

    if constexpr (L16) {
        // accumulator tile (in16 = 16-channel tile 0..3, im16 = 16-token tile 0..7) -> rows mw0 + 16*im16, cols n0 + wn*64 + 16*in16
        auto get16 = [&](int in16, int im16) -> const v4i & { return acc16[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1]; };
        bool staged16 = false;
        if constexpr (Epi::kOutBytes >= 2) staged16 = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);
        const int64_t mw0 = m0 + (half ? wm * 64 : wm * 128), nw0 = n0 + wn * 64;   // (half tile: see below)
        const int64_t Mw = half && mw0 + 64 < M ? mw0 + 64 : M;
        auto run = [&](auto getter) {
            if constexpr (IsGateUp<Epi>::value) {   // gate || up groups: SiLU(gate) * up of the lane's own accumulator pairs, half the output bytes (asq_gemm_gateup.h)
                epilogue_gate_up_rows(epi, getter, mw0, nw0, lane, Mw);
            } else if (staged16) {
                if constexpr (Epi::kOutBytes >= 2) {
                    P8_BAR();
                    bool rows_path = false;
                    if constexpr (Epi::kOutBytes == 2) rows_path = !half && mw0 + 128 <= M && nw0 + 64 <= N && epi.N < (int64_t(1) << 27);
                    if (rows_path) {
                        if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<4, 2, true>(epi, getter, mw0, nw0, lane, lds0 + wave * 16384, rows_write_through(M_all, epi));
                    } else
                        epilogue_wave_staged<4, 0, true>(epi, getter, mw0, nw0, lane, Mw, N, lds0 + wave * 16384);
                }
            } else {
                epilogue_wave16(epi, getter, mw0, nw0, lane, Mw, N);
            }
        };
        if constexpr (kOffs) if (offs) {
            // (gemm_i8_p16's correction, with the column pairs PACKED too -- cw << 24 | wsum & 0xFFFFFF, 16 registers instead of 32: this kernel's scheduler and K-split
            // fix-up leave fewer registers than p16 has; one more shift per accumulator tile.  v_mul_i32_i24 takes the low 24 bits of either factor.)
            const int t16i = lane & 15, q16i = lane >> 4;
            int cp[4][4], rw[8];
#pragma unroll
            for (int in16 = 0; in16 < 4; ++in16) {
                const unsigned ca = obase + 1024 + (wn * 64 + in16 * 16 + 4 * q16i) * 8;
                const v4i p0 = *(lds_v4i_)(uintptr_t)ca, p1 = *(lds_v4i_)(uintptr_t)(ca + 16);
                cp[in16][0] = (int)(((unsigned)p0[0] << 24) | ((unsigned)p0[1] & 0xFFFFFFu));
                cp[in16][1] = (int)(((unsigned)p0[2] << 24) | ((unsigned)p0[3] & 0xFFFFFFu));
                cp[in16][2] = (int)(((unsigned)p1[0] << 24) | ((unsigned)p1[1] & 0xFFFFFFu));
                cp[in16][3] = (int)(((unsigned)p1[2] << 24) | ((unsigned)p1[3] & 0xFFFFFFu));
            }
            const int row0 = (int)(mw0 - m0) + t16i;
            constexpr bool kRowsInRegs = !(Epi::kHasRow && Epi::kHasBias);   // (the per-token + bias epilogue has no 8 registers to spare: it re-reads the word per tile)
            if constexpr (kRowsInRegs) {
#pragma unroll
                for (int im16 = 0; im16 < 8; ++im16) rw[im16] = *(lds_i32)(uintptr_t)(obase + ((row0 + im16 * 16) & 255) * 4);
            }
            auto getc16 = [&](int in16, int im16) -> v4i {
                const v4i &a = acc16[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1];
                int r;
                if constexpr (kRowsInRegs) r = rw[im16];
                else r = *(lds_i32)(uintptr_t)(obase + ((row0 + im16 * 16) & 255) * 4);
                const int ncx = r >> 24;
                v4i o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __mul24(cp[in16][e] >> 24, r) + (__mul24(ncx, cp[in16][e]) + a[e]);
                return o;
            };
            run(getc16);
            return;
        }
        run(get16);
        return;
    }
    if constexpr (IsGateUp<Epi>::value) {   // int8 gate || up runs on the 16 x 16 x 64 form above; here: fp8 groups on the 32 x 32 block-scaled instruction (EpiGateUpFp8)
        if constexpr (GRP && !MMA::kIsInt) {
            auto getf = [&](int in, int im) -> const acc_t & { return acc[im >> 1][in][im & 1]; };
            epilogue_gate_up_rows32(epi, getf, m0 + wm * 128, n0 + wn * 64, lane, M);
        }
        return;
    } else {
    // accumulator tile (in = n-half, im = 2*m-half + j) -> rows m0 + wm*128 + 32*im, cols n0 + wn*64 + 32*in
    auto get = [&](int in, int im) -> const acc_t & { return acc[im >> 1][in][im & 1]; };
    bool staged = false;
    if constexpr (Epi::kOutBytes >= 2) staged = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);  // N: this launch's column bound, epi.N: the row stride
    // (half tile: wave group wm holds rows [64 wm, 64 wm + 64) in its first row half; the second half is masked by the row bound)
    const int64_t mw0 = m0 + (half ? wm * 64 : wm * 128);
    const int64_t Mw = half && mw0 + 64 < M ? mw0 + 64 : M;
    if (staged) {
        if constexpr (Epi::kOutBytes >= 2) {
            P8_BAR();  // every wave's ring reads are done and every wave's (dead) DMAs have landed: the ring becomes staging space
            bool rows_path = false;
            if constexpr (Epi::kOutBytes == 2 && P8_STORE_POLICY(ABL) == 0 && !(ABL & 2048))   // interior wave tile: no bound checks, pipelined (epilogue_wave_rows)
                rows_path = !half && mw0 + 128 <= M && n0 + wn * 64 + 64 <= N && epi.N < (int64_t(1) << 27);
            if (rows_path) {
                if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<4, 2>(epi, get, mw0, n0 + wn * 64, lane, lds0 + wave * 16384);
            } else
                epilogue_wave_staged<4, P8_STORE_POLICY(ABL)>(epi, get, mw0, n0 + wn * 64, lane, Mw, N, lds0 + wave * 16384);
        }
    } else {
        epilogue_wave<2, 4>(epi, get, [](int im) { return im * 32; }, mw0, n0 + wn * 64, lane, Mw, N);
    }
    P8_PROBE_END_WHERE(tile_m, tile_n);
    }
}

}  // namespace asq
