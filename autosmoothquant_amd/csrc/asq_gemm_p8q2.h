// "p8q2": the 128(m) x 128(n) x 128(k) mid-size kernel (gemm_i8_p8q, int8 on v_mfma_i32_16x16x64_i8) rebuilt around its measured bound (VERDICT r4 item 2,
// profiles/r4_p8q_bound.txt): gemm_i8_p8q spends ~950 cycles per K-tile on 512 cycles of matrix work without being LDS-, DMA- or MFMA-bound -- its ONE barrier
// per K-tile releases all eight waves together, so on every SIMD both waves issue their loads at the same time and then share the matrix pipe, and nothing
// overlaps one wave's load segment with the other's MFMAs.
//
// Here the K-tile is TWO segments separated by barriers, and the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER apart, as
// gemm_i8_p8 / p16 do per phase:
//     L (load)     s_waitcnt lgkmcnt(0)      the fragments of tile t (read one K-tile ago) are in registers
//                  4 LDS-DMAs                tile t+3 into the slot of tile t-1
//                  12 ds_read_b128           fragments of tile t+1 into the OTHER register set
//                  s_waitcnt vmcnt(4)        this wave's part of tile t+2 has landed
//                  s_barrier
//     C (compute)  16 MFMAs on tile t        (no wait: its fragments were waited for at the top of L)
//                  s_barrier
// so while one group's matrix instructions run, the other group issues its loads: a K-tile costs 2 x max(L, C) instead of L + 2 C.  Everything the
// schedule relies on is ordered by a counted wait plus a barrier (global barrier index g; group A = waves 0-3, group B one barrier behind):
//   * tile t+2 is visible to the reads of L(t+1): A waits for its part before g = 2t, B before g = 2t+1; A reads after g = 2t+1, B after g = 2t+2;
//   * the slot of tile t-1 is free when tile t+3 is issued into it: every wave waited for its fragments of t-1 at the top of its L(t-1), i.e. before
//     g = 2t-2 (A) / g = 2t-1 (B); A issues after g = 2t-1, B after g = 2t.
// Same operands, same exact integer sums, same epilogues as gemm_i8_p8q<Epi, false, true>: results are bit-identical.
#pragma once

namespace asq {

// Measured (profiles/r5_midsize_p8q2.txt, cold weights): 512 x 4096 x 4096 16.5 -> 15.0 us, 256 rows 16.2 -> 14.8, 1024 rows 20.4 -> 19.9, 1024 x 4096 x 11008 43.9 -> 42.4:
// ~790 cycles per K-tile instead of ~950.  What is left is the load segment itself: 4 LDS-DMA instructions + 12 ds_read_b128 per wave take longer to ISSUE than the
// 256 cycles of the partner's 16 MFMAs (the 128 x 128 tile moves twice p16's DMA bytes per MFMA: 32 one-KiB pieces per 512 cycles of matrix work per CU).  Two variants
// changed nothing: warm instead of cold weights (15.3 vs 15.0: not the prefetch lead), and group A waiting for tile t+2 at the end of its compute segment
// (half a K-tile more lead: 15.3 vs 15.0).
//
// FIX = true (round 5): the K splits of a tile are reduced INSIDE the launch (round 4's dropped experiment, profiles/r4_splitk_in_launch_dropped.txt, rebuilt on this
// kernel and kept where it measures ahead): every block stores its 32 accumulator registers per lane as a 64 KiB write-through image, takes the tile's ticket in the
// workspace header (one agent-scope atomic; tickets return to zero, so a captured graph replays), and the last arriver adds the other images -- the loads of up to
// three images in flight together -- and runs the CALLER's epilogue: no reduce launch, no int32 slab round trip.  Correct on any placement of the blocks, no block
// ever waits for another.  blockIdx b = 8 * slot + xcd takes item `slot` of XCD xcd's share of the tiles, (tile, split) = (base + slot / S, slot % S).
// Measured and NOT kept (profiles/r5_splitk_fix_sweep_v1.txt, r5_xcd_local_probe.txt): an XCD-local protocol -- a tile's splits on one XCD, plain image stores
// that stay dirty in that L2, a workgroup-scope ticket performed there, L2-hit reads (0.36 us per image instead of 0.6 in tools/ubench/xcd_local) behind a
// placement probe in the workspace header -- was 3-5 % SLOWER than this form on every shape: the dirty images are written back at the end of the kernel, in
// front of the next launch, instead of under the other blocks' K loops.
template <class Epi, bool FIX = false>
__global__ void __launch_bounds__(512, 2) gemm_i8_p8q2(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                       int tiles_m, int tiles_n, int ksplit, Epi epi_in, char *__restrict__ gws)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(Epi::Mma::kIsInt, "int8 operands");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the wave's tile: 4 (m) x 2 (n) as gemm_i8_p8q; waves w and w + 4 share a SIMD and belong to different groups
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2;   // 0: group A, 1: group B (one barrier behind)

    constexpr int GM = 8;
    const int nwg = tiles_m * tiles_n;
    int split, id;
    [[maybe_unused]] unsigned long long ws_magic = 0;
    if constexpr (FIX) {
        if (!splitk_fix_item(nwg, ksplit, id, split)) return;   // (block-uniform; the grid is 8 x the largest share)
        ws_magic = *(const volatile unsigned long long *)gws;   // checked at the ticket
    } else {
        const int lid = xcd_remap(blockIdx.x, nwg * ksplit);
        split = lid / nwg;
        id = lid - split * nwg;
    }
    const Epi epi = epi_in.rebased(0, FIX ? 0 : split, M, N);   // (FIX: the last arriver writes the caller's output itself; slabs otherwise)
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = (int64_t)tile_m * 128, n0 = (int64_t)tile_n * 128;

    const int nt_all = (int)(K / 128);
    const int kt0 = (int)((int64_t)nt_all * split / ksplit), kt1 = (int)((int64_t)nt_all * (split + 1) / ksplit);
    const int8_t *const xbase = uniform_ptr(x + m0 * K + (int64_t)kt0 * 128);
    const int8_t *const wbase = uniform_ptr(w + n0 * K + (int64_t)kt0 * 128);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;
    unsigned voff[2][2];  // [kind: X, W][i]; this wave fills row-groups 2 * wave, 2 * wave + 1 (8 rows each) of both units
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ru = (wave * 2 + i) * 8 + (lane >> 3);
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);
        const int64_t rx = ru < mrem ? ru : mrem, rw = ru < nrem ? ru : nrem;
        voff[0][i] = (unsigned)(rx * K) + cb;
        voff[1][i] = (unsigned)(rw * K) + cb;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned dma_dst = lds0 + wave * 2048;

    const int t16 = lane & 15, q16 = lane >> 4, sw16 = (t16 >> 1) & 7;
    unsigned xb16[4][2], wbp16[4][2];   // one VGPR per (stage, k-step of 64) and operand
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned off = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw16)) << 4) + s * P8Q_STAGE;
            xb16[s][kk] = off + wm * 32 * 128;
            wbp16[s][kk] = off + P8_UNIT + wn * 64 * 128;
            asm volatile("" : "+v"(xb16[s][kk]), "+v"(wbp16[s][kk]));
        }

    v4i acc16[2][4];  // [token tile][channel tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc16[a][b] = (v4i){0, 0, 0, 0};

    const int nt = kt1 - kt0;  // K-tiles of this block (>= 1)
    const int klast = (nt - 1) * 128;
    auto issue = [&](int stage, int k0) {
        p8_dma16(xbase + k0, voff[0][0], dma_dst + stage * P8Q_STAGE);
        p8_dma16(xbase + k0, voff[0][1], dma_dst + stage * P8Q_STAGE + 1024);
        p8_dma16(wbase + k0, voff[1][0], dma_dst + stage * P8Q_STAGE + P8_UNIT);
        p8_dma16(wbase + k0, voff[1][1], dma_dst + stage * P8Q_STAGE + P8_UNIT + 1024);
    };

    v4i xf16[2][2][2], wf16[2][4][2];  // [register set][tile][k-step]
    auto read_frags = [&](auto stage_tag, auto set_tag) {
        constexpr int S = decltype(stage_tag)::value, R = decltype(set_tag)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) xf16[R][jt][kk] = *(p8_lds_v4i)(uintptr_t)(xb16[S][kk] + jt * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 4; ++it) wf16[R][it][kk] = *(p8_lds_v4i)(uintptr_t)(wbp16[S][kk] + it * 2048);
    };

    // ---- prologue: K-tiles 0, 1, 2 (clamped); tiles 0 and 1 landed and visible; fragments of tile 0 on their way
    issue(0, 0);
    issue(1, 128 < klast ? 128 : klast);
    issue(2, 256 < klast ? 256 : klast);
    P8_WAIT_VM(4);
    __builtin_amdgcn_s_barrier();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: group B runs one barrier behind

    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = (S + 3) % 4, S1 = (S + 1) % 4, R = S & 1;
        int kn = (t + 3) * 128;  // SALU
        kn = kn < klast ? kn : klast;
        // ---------------- L
        P8_WAIT_LGKM0();   // fragments of tile t (issued one K-tile ago)
        issue(NS, kn);
        read_frags(std::integral_constant<int, S1>{}, std::integral_constant<int, R ^ 1>{});
        P8_WAIT_VM(4);     // this wave's part of tile t+2 has landed; tile t+3 stays in flight
        __builtin_amdgcn_s_barrier();
        // ---------------- C
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc16[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf16[R][it][kk], xf16[R][jt][kk], acc16[jt][it], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    int t = 0;
    for (; t + 3 < nt; t += 4) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
        ktile(std::integral_constant<int, 2>{}, t + 2);
        ktile(std::integral_constant<int, 3>{}, t + 3);
    }
    if (t < nt) ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) ktile(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < nt) ktile(std::integral_constant<int, 2>{}, t + 2);

    P8_WAIT_VM(0);    // drain the dead prefetches
    P8_WAIT_LGKM0();  // ... and the fragment reads past the last tile, before LDS becomes staging space
    if (grp == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier

    if constexpr (FIX)
        if (ksplit > 1 && !splitk_fix_reduce<8>(gws, ws_magic, id, split, ksplit, tid, lds, [&](int i) -> v4i & { return acc16[i >> 2][i & 3]; })) return;   // not the last arriver

    // accumulator tile (in16 = 16-channel tile 0..3, im16 = 16-token tile 0..1) -> rows m0 + wm*32 + 16*im16, cols n0 + wn*64 + 16*in16
    auto get16 = [&](int in16, int im16) -> const v4i & { return acc16[im16][in16]; };
    bool staged16 = false;
    if constexpr (Epi::kOutBytes >= 2) staged16 = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);
    const int64_t mw0 = m0 + wm * 32, nw0 = n0 + wn * 64;
    if (staged16) {
        if constexpr (Epi::kOutBytes >= 2) {
            __builtin_amdgcn_s_barrier();  // all ring reads done, all (dead) DMAs landed: the ring becomes staging space
            bool rows_path = false;
            if constexpr (Epi::kOutBytes == 2) rows_path = mw0 + 32 <= M && nw0 + 64 <= N && epi.N < (int64_t(1) << 27);
            if (rows_path) {
                if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<1, 2, true>(epi, get16, mw0, nw0, lane, lds0 + wave * 16384, rows_write_through(M, epi));
            } else
                epilogue_wave_staged<1, 0, true>(epi, get16, mw0, nw0, lane, M, N, lds0 + wave * 16384);
        }
    } else {
        epilogue_wave16<2>(epi, get16, mw0, nw0, lane, M, N);
    }
}

}  // namespace asq
