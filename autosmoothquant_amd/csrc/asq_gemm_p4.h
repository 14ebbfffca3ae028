// "p4": 256(m) x 256(n) x 128(k) INT8 MFMA GEMM, FOUR waves (one per SIMD), 128 x 128 outputs per wave.
// Included by asq_gemm_kernels.h after asq_gemm_p8.h (shares its LDS-DMA helper, unit image and epilogues).
//
// Why a second 256 x 256 kernel.  On real operands the MI355X runs this GEMM against its 1.4 kW socket limit, not against
// the matrix-core issue rate (profiles/r2_clock_power_evidence.md): wall time ~ E_dyn / (P_limit - P_base), and of the
// 46.5 mJ a 4096^3 launch of gemm_i8_p8 spends, 6.8 mJ are the fragment ds_reads (1.57 GB per launch, 94 B/clk/CU).  A
// wave tile of 128(m) x 64(n) reads 24 KiB of fragments per K-tile for 32 MFMAs; 128 x 128 reads 32 KiB for 64 -- one third
// fewer LDS bytes per MFMA -- and lets 4 consecutive MFMAs keep the same W fragment while the X fragments snake
// (..., x3 | x3, x2, ...), so fewer operand registers toggle between matrix instructions (MFMA-only: "snake" order
// 3861 vs "distinct" 3757 TOPS).  The price is the p8 schedule's slack: with one wave per SIMD nothing covers a wave that
// waits at the barrier, so the loop is organised to never wait on a short lead:
//
//   LDS ring = 2 K-tiles x {X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255} x 16 KiB (unit image as p8:
//   [128 rows][8 x 16 B], chunk ^= (row >> 1) & 7, swizzle applied to the DMA's global source).
//   K-tile t (stage S = t & 1), four k-steps of 32, 16 MFMAs (512 matrix-pipe cycles) each:
//     step 0   issue all 16 LDS-DMAs of K-tile t+1 -> stage S^1 | ds_read fragments of step 1 | MFMAs of step 0
//     step 1                                                     | ds_read fragments of step 2 | MFMAs of step 1
//     step 2                                                     | ds_read fragments of step 3 | MFMAs of step 2
//              s_waitcnt vmcnt(0); s_barrier     (every wave's share of K-tile t+1 has landed >= 2 steps after its issue;
//                                                 every wave's reads of stage S are done: K-tile t+2 may overwrite it)
//     step 3   ds_read fragments of (t+1, step 0) from stage S^1                               | MFMAs of step 3
//   Fragments are double-buffered in registers (8 x v4i each); the accumulators take 256 VGPRs; ~380 of the 512 in total.
//   One barrier per K-tile; no fragment read is ever exposed; the DMA lead is 2-3 steps (1000-1500 matrix cycles).
//
// Same operand roles as p8 (W = matrix-core A operand, X = B: a lane owns 4 consecutive output channels of one token),
// same staged, coalescing epilogue (the 128 x 128 wave tile leaves as two 128 x 64 halves through a 16 KiB wave-private
// LDS image), same XCD-aware tile map.  ksplit == 1, no groups: split-K and grouped launches stay on p8.
#pragma once

namespace asq {

constexpr int P4_UNIT = 128 * 128;      // 16 KiB
constexpr int P4_STAGE = 4 * P4_UNIT;   // 64 KiB: X-lo, X-hi, W-lo, W-hi
constexpr int P4_LDS_BYTES = 2 * P4_STAGE;

// LDS-DMA without saving M0 (nothing else in this kernel uses it)
__device__ __forceinline__ void p4_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}


// Staged, coalescing epilogue of one 128(m) x 128(n) wave tile with 2-byte outputs, pipelined over the four 32-row token
// tiles: tile im is packed and written to a wave-private 8 KiB LDS image [32 rows][16 x 16 B] (two images, used alternately),
// read back lane-linearly and stored as whole 256-byte rows (16 lanes x 16 B) while the VALU already converts tile im + 1.
// With ONE wave per SIMD nothing else overlaps the conversion (~4.5 VALU ops per output incl. the AGPR read) with the stores,
// and the first store of the simpler "two 128 x 64 halves" form left only after half of the tile had been converted
// (epilogue 13.6k cycles at 4096^3; the memory system needs ~8k for the 33.5 MB).
// Image swizzle: 16-byte chunk c of row r sits at chunk position p = c ^ (r & 15); inside a chunk the two 8-byte halves are
// flipped when p >= 8, so that the 16 consecutive lanes (rows) served together by ds_write_b64 -- same c, 32 banks -- hit 32
// different banks; the lane-linear ds_read_b128 is conflict-free by construction.
template <class Epi, class Get>
__device__ __forceinline__ void p4_epilogue_rows(const Epi &epi, Get get, int64_t mw0, int64_t nw0, int lane, int64_t M, int64_t N, unsigned stage)
{
    static_assert(Epi::kOutBytes == 2, "2-byte outputs");
    typedef __attribute__((address_space(3))) v4i *lds_v4i;
    typedef __attribute__((address_space(3))) v2u *lds_u2;
    const int ml = lane & 31, hi = lane >> 5;
    float sr[4];
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int64_t m = mw0 + im * 32 + ml;
        sr[im] = 1.0f;
        if constexpr (Epi::kHasRow) {
            if (m < M) sr[im] = epi.row(m);
        }
    }
    v4f sc[4][4], bb[4][4];
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t n = nw0 + in * 32 + 8 * g + 4 * hi;
            sc[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
            bb[in][g] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (n < N) epi.cols(n, N, sc[in][g], bb[in][g]);
        }
    char *const outb = (char *)epi.out;
    using acc4_t = typename Epi::Mma::acc4_t;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const unsigned img = stage + (im & 1) * 8192;
#pragma unroll
        for (int in = 0; in < 4; ++in) {
            const typename Epi::Mma::acc_t a = get(in, im);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v2u v = epi.pack((acc4_t){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}, sr[im], sc[in][g], bb[in][g]);
                const int p = (in * 4 + g) ^ (ml & 15);
                *(lds_u2)(uintptr_t)(img + ml * 256 + (p << 4) + 8 * (hi ^ (p >> 3))) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4), p = lane & 15;
            v4i v = *(lds_v4i)(uintptr_t)(img + i * 1024 + lane * 16);
            if (p >> 3) v = (v4i){v[2], v[3], v[0], v[1]};
            const int64_t m = mw0 + im * 32 + row, n = nw0 + ((p ^ (row & 15)) << 3);
            if (m < M && n < N) *(v4i *)(outb + (m * epi.N + n) * 2) = v;
        }
    }
}

// PROBE (tools/ubench/clock_probe, needs ASQ_P8_PROBE): 1 = stamps; +2 = no LDS-DMA in the loop, +4 = no vmcnt wait / barrier in the
// loop, +8 = no fragment ds_reads (ablations: results invalid), +16 = the round-2 epilogue (A/B against epilogue_wave_rows).  Production: 0.
template <class Epi, int PROBE = 0>
__global__ void __launch_bounds__(256, 1) gemm_i8_p4(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                     int tiles_m, int tiles_n, Epi epi_in)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    P4_BLK_RT(6);
    P4_BLK(0);
    constexpr int GM = 4;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const Epi epi = epi_in.rebased(0, 0, M, N);
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;

    // ---- DMA sources.  Instruction q = 0..15 of this wave fills piece (q & 3) * 4 + wave (8 rows x 128 B) of unit q >> 2.
    const int nt = (int)(K / 128);
    const int8_t *const xbase = uniform_ptr(x + m0 * K);
    const int8_t *const wbase = uniform_ptr(w + n0 * K);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;  // last valid local row
    unsigned voff[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int unit = q >> 2, piece = (q & 3) * 4 + wave;
        const int ru = piece * 8 + (lane >> 3);                                    // row within the unit, 0..127
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);       // swizzled source chunk
        int64_t r = (unit & 1) * 128 + ru;                                        // local tile row
        const int64_t lim = unit < 2 ? mrem : nrem;
        r = r < lim ? r : lim;
        voff[q] = (unsigned)(r * K) + cb;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;

    // ---- fragment read addresses: one VGPR per (stage, operand, k-step); + 4096 * i (32-row tile i) as an immediate
    const int frow = lane & 31, sw = (frow >> 1) & 7, hi = lane >> 5;
    unsigned xa[2][4], wa[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = lds0 + s * P4_STAGE + frow * 128 + ((((ks * 2 + hi) ^ sw)) << 4);
            xa[s][ks] = off + wm * P4_UNIT;
            wa[s][ks] = off + (2 + wn) * P4_UNIT;
            asm volatile("" : "+v"(xa[s][ks]), "+v"(wa[s][ks]));
        }

    v16i acc[4][4];  // [n-tile][m-tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v16i){0};

    auto issue_tile = [&](int stage, int k0) {  // all 16 DMAs of one K-tile (k0 = its first k byte)
        const int8_t *xb = xbase + k0, *wb = wbase + k0;
        const unsigned dst = lds0 + stage * P4_STAGE;
#pragma unroll
        for (int q = 0; q < 16; ++q) p4_dma16((q >> 2) < 2 ? xb : wb, voff[q], dst + (q >> 2) * P4_UNIT + ((q & 3) * 4 + wave) * 1024);
    };
    // fragment i of the next k-step (i = 0..3: W tiles, 4..7: X tiles); issued one per matrix instruction so that the LDS
    // issue slots sit inside the MFMA shadow instead of in front of it
    auto ldfrag1 = [&](v4i (&fw)[4], v4i (&fx)[4], int stage, int ks, int i) {
        if constexpr (PROBE & 8) return;
        if (i < 4) fw[i] = *(p8_lds_v4i)(uintptr_t)(wa[stage][ks] + i * 4096);
        else fx[i - 4] = *(p8_lds_v4i)(uintptr_t)(xa[stage][ks] + (i - 4) * 4096);
    };
    // One k-step: 16 MFMAs on (fw, fx), snake over the 4 x 4 (W fragment, X fragment) grid -- the W fragment stays for 4
    // instructions, the X fragment at each turn for 2 -- with, after MFMA number idx, (a) the load of fragment idx of the NEXT
    // k-step into (nw, nx) for idx < 8 and (b) whatever `dma(idx)` wants to issue.
    auto kstep = [&](const v4i (&fw)[4], const v4i (&fx)[4], v4i (&nw)[4], v4i (&nx)[4], int nstage, int nks, auto dma) {
#pragma unroll
        for (int in = 0; in < 4; ++in)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int im = (in & 1) ? 3 - j : j, idx = in * 4 + j;
                acc[in][im] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[in], fx[im], acc[in][im], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (idx < 8) ldfrag1(nw, nx, nstage, nks, idx);
                dma(idx);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto nodma = [](int) {};
    auto dma_q = [&](int q, int stage, int k0) {
        if constexpr (PROBE & 2) return;
        p4_dma16(((q >> 2) < 2 ? xbase : wbase) + k0, voff[q], lds0 + stage * P4_STAGE + (q >> 2) * P4_UNIT + ((q & 3) * 4 + wave) * 1024);
    };
    const int klast = (nt - 1) * 128;
    auto kclamp = [&](int tt) { const int k = tt * 128; return k < klast ? k : klast; };  // past the end: the last tile again, into a dead stage

    // ---- prologue: K-tile 0 entirely, the first 5 DMAs of K-tile 1, fragments of (0, step 0)
    issue_tile(0, 0);
    P8_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    v4i fw0[4], fx0[4], fw1[4], fx1[4];
    if constexpr (PROBE & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fw0[i] = fx0[i] = fw1[i] = fx1[i] = (v4i){lane, i, 2, 3};
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) dma_q(q, 1, kclamp(1));
#pragma unroll
    for (int i = 0; i < 8; ++i) ldfrag1(fw0, fx0, 0, 0, i);
    P4_BLK(1);

    // K-tile t at stage S.  LDS-DMA schedule for K-tile t+1 (stage S^1), one issue every third MFMA:
    //   q = 0..4   in step 3 of K-tile t-1 (right after the barrier that freed stage S^1)
    //   q = 5..10  in step 0,  q = 11..15 in step 1;  step 2 issues nothing: every DMA has >= 512 matrix cycles of lead before
    //   the s_waitcnt vmcnt(0) + s_barrier at its end.
    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = S ^ 1;
        const int kn = kclamp(t + 1), kn2 = kclamp(t + 2);
        kstep(fw0, fx0, fw1, fx1, S, 1, [&](int idx) { if (idx % 3 == 0) dma_q(5 + idx / 3, NS, kn); });              // q = 5..10
        kstep(fw1, fx1, fw0, fx0, S, 2, [&](int idx) { if (idx % 3 == 0 && idx / 3 < 5) dma_q(11 + idx / 3, NS, kn); });  // q = 11..15
        kstep(fw0, fx0, fw1, fx1, S, 3, nodma);
        if constexpr (!(PROBE & 4)) {
            P8_WAIT_VM(0);      // my share of K-tile t+1 has landed
            P8_WAIT_LGKM0();    // my reads of stage S have returned (issued >= 256 cycles ago) -- K-tile t+2 may overwrite it
            __builtin_amdgcn_s_barrier();
        }
        kstep(fw1, fx1, fw0, fx0, NS, 0, [&](int idx) { if (idx % 3 == 0 && idx / 3 < 5) dma_q(idx / 3, S, kn2); });       // q = 0..4 of K-tile t+2
    };

    // The loop is a dense hand-ordered MFMA stream: its speed depends on where it starts in the instruction cache (measured: the same
    // loop ran 86.7k and 71.3k cycles at 4096^3 in two builds that differed only in code placed before it), so its start is pinned.
    asm volatile(".p2align 8");
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nt) ktile(std::integral_constant<int, 0>{}, t);
    asm volatile("" ::"v"(fw0[0]), "v"(fx0[0]));  // (the last prefetch is dead)
    P8_WAIT_VM(0);                                // (so are the 5 DMAs issued in the last step)

    P4_BLK(2);
    // ---- epilogue: the ring becomes staging space; the 128 x 128 wave tile leaves as two 128 x 64 halves
    P8_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    const int64_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
    static_assert(Epi::kOutBytes == 2, "gemm_i8_p4: 2-byte outputs (launch_gemm sends the other epilogues to p8)");
    const bool staged = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);  // N: this launch's column bound, epi.N: the row stride
    // interior wave tile: pipelined, no bound checks (epilogue_wave_rows).  Only for the scalar-scale epilogues: with per-channel scales and a bias the
    // 128 x 128 wave tile holds 128 registers of column vectors, and the pipelined form's extra address registers push it into scratch.
    constexpr bool kRows = !Epi::kHasCol && !Epi::kHasBias;
    if (kRows && !(PROBE & 16) && staged && mw0 + 128 <= M && nw0 + 128 <= N && epi.N < (int64_t(1) << 27)) {
        if constexpr (kRows) epilogue_wave_rows<4, 4>(epi, [&](int in, int im) -> const v16i & { return acc[in][im]; }, mw0, nw0, lane, lds0 + wave * 32768);
    } else if (staged) {
        p4_epilogue_rows(epi, [&](int in, int im) -> const v16i & { return acc[in][im]; }, mw0, nw0, lane, M, N, lds0 + wave * 16384);
    } else {  // unaligned output / ragged row pitch: direct stores in the matrix-core layout, two 128 x 64 halves
        epilogue_wave<2, 4>(epi, [&](int in, int im) -> const v16i & { return acc[in][im]; }, [](int im) { return im * 32; }, mw0, nw0, lane, M, N);
        epilogue_wave<2, 4>(epi, [&](int in, int im) -> const v16i & { return acc[2 + in][im]; }, [](int im) { return im * 32; }, mw0, nw0 + 64, lane, M, N);
    }
    P4_PROBE_END();
}

}  // namespace asq