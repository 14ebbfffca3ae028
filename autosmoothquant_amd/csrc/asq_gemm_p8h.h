// "p8h": 128(m) x 256(n) x 128(k) sibling of the p8 kernel (asq_gemm_p8.h) for row counts that leave
// the 256-row tiling with too few tiles for 256 CUs (M = 2048 x N = 4096: 128 tiles) or half-empty
// tiles (M = 384).  Same machinery -- LDS-DMA in full 128-B rows, XOR-swizzled 16 KiB units,
// counted vmcnt, two wave groups staggered by one barrier, no VALU in the load segments, staged
// coalescing epilogue -- with a different work split:
//
//   8 waves = 2 (m) x 4 (n); wave (wm, wn) owns out[wm*64 .. +64][wn*64 .. +64] = 2 n-halves of
//   64(m) x 32(n), each 2 MFMA tiles (v_mfma_i32_32x32x32_i8, W = A operand, X = B operand).
//   One K-tile (128 k-bytes) is TWO phases of 8 MFMAs:
//       P1 (n-half 0)  reads X (8 x ds_read_b128) + W-even (4)
//       P2 (n-half 1)  reads W-odd (4), X fragments stay in VGPRs
//   K-tile image = 3 units {X: tile rows 0..127, W-even: rows 64*wn + [0,32), W-odd: + 32} = 48 KiB;
//   ring of THREE K-tiles (144 KiB): tile t+2 is fetched while tile t is consumed, so a unit has two
//   K-tiles (4 phases, ~2000 cycles) to arrive -- the same depth as p8's 4-phase lead.
//   DMA issue order (6 per wave per K-tile, split 2 + 4 to even out the two load segments):
//       P1(t): X[0] X[1]  of tile t+2              P2(t): We[0] We[1] Wo[0] Wo[1]  of tile t+2
//   Counted waits (after the phase's own issue; a wave retires its DMAs in order):
//       P1(t): W-odd(t) must have landed for P2(t): 8 younger DMAs may stay in flight -> vmcnt(8)
//       P2(t): X(t+1), W-even(t+1) for P1(t+1): 8 younger                            -> vmcnt(8)
//   A slot is re-filled >= 2 phases (4 barriers) after its last ds_read; the stagger is 1 barrier.
//
//   Measured (tools/ubench/p8_probe, s_memtime per block, 4096-deep K): prologue 4.3k + K-loop 37.6k (MFMA-issue
//   floor 32.8k = 87 %) + epilogue 4.2k cycles per 128x256 tile, i.e. 92.6k per 256x256 of output against the
//   81.4k of p8.  3+3 and 2+4 DMA splits measure the same.  Per K-tile the half tile moves 48 KiB through the
//   64 B/clk/CU vector-memory path in ~1024 cycles (73 % of that peak; p8: 64 KiB in 2048 = 50 %) and its 64x64
//   wave tiles read one 1-KiB fragment per MFMA (125 B/clk of LDS reads, peak 256; p8's 128x64 wave tile: 94).
#pragma once
#include <type_traits>

namespace asq {

constexpr int P8H_STAGE = 3 * P8_UNIT;      // 48 KiB
constexpr int P8H_LDS_BYTES = 3 * P8H_STAGE;  // 144 KiB


// L16 (int8 only): the matrix work on v_mfma_i32_16x16x64_i8 (asq_gemm_p16.h for why): a phase is 2 k-steps x {4 token tiles x 2 channel tiles} = 16
// instructions of 16 cycles on acc16[n-half][token tile][channel tile]; fragments 16 rows x 64 k-bytes from the same unit images.
template <class Epi, bool PROBE = false, bool L16 = false>  // PROBE: per-block s_memtime stamps for tools/ubench/p8_probe (production: false)
__global__ void __launch_bounds__(512, 2) gemm_i8_p8h(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                      int tiles_m, int tiles_n, int ksplit, Epi epi_in)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    P8H_BLK(0);

    // logical id = split * ntiles + tile (see p8); groups of GM tile rows share their X panels in L2
    constexpr int GM = 8;
    const int nwg = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg * ksplit);
    const int split = lid / nwg, id = lid - split * nwg;
    const Epi epi = epi_in.rebased(0, split, M, N);
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = (int64_t)tile_m * 128, n0 = (int64_t)tile_n * 256;

    const int nt_all = (int)(K / 128);
    const int kt0 = (int)((int64_t)nt_all * split / ksplit), kt1 = (int)((int64_t)nt_all * (split + 1) / ksplit);
    const int8_t *const xbase = uniform_ptr(x + m0 * K + (int64_t)kt0 * 128);
    const int8_t *const wbase = uniform_ptr(w + n0 * K + (int64_t)kt0 * 128);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;  // last valid local row
    unsigned voff[3][2];  // [kind: X, W-even, W-odd][i]; this wave fills row-groups 2*wave, 2*wave+1 (8 rows each) of every unit
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ru = (wave * 2 + i) * 8 + (lane >> 3);                       // row within the unit, 0..127
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);  // swizzled source chunk
        int64_t rx = ru, rwe = (ru >> 5) * 64 + (ru & 31), rwo = rwe + 32;
        rx = rx < mrem ? rx : mrem;
        rwe = rwe < nrem ? rwe : nrem;
        rwo = rwo < nrem ? rwo : nrem;
        voff[0][i] = (unsigned)(rx * K) + cb;
        voff[1][i] = (unsigned)(rwe * K) + cb;
        voff[2][i] = (unsigned)(rwo * K) + cb;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned dma_dst = lds0 + wave * 2048;  // + stage*P8H_STAGE + kind*P8_UNIT + i*1024

    // fragment read addresses: one VGPR per (stage, operand, k-substep); the rest is an immediate
    const int frow = lane & 31, sw = (frow >> 1) & 7, hi = lane >> 5;
    unsigned xb[3][4], wbp[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = lds0 + frow * 128 + ((((ks * 2 + hi) ^ sw)) << 4) + s * P8H_STAGE;
            xb[s][ks] = off + wm * 64 * 128;   // X unit: this wave's 64 rows (+ j * 4096)
            wbp[s][ks] = off + wn * 32 * 128;  // W units: this wave's 32 rows (+ kind * P8_UNIT)
            asm volatile("" : "+v"(xb[s][ks]), "+v"(wbp[s][ks]));
        }

    unsigned xb16[3][2], wbp16[3][2];   // L16: one VGPR per (stage, operand, k-step of 64)
    if constexpr (L16) {
        const int t16 = lane & 15, q16 = lane >> 4, sw16 = (t16 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned off = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw16)) << 4) + s * P8H_STAGE;
                xb16[s][kk] = off + wm * 64 * 128;
                wbp16[s][kk] = off + wn * 32 * 128;
                asm volatile("" : "+v"(xb16[s][kk]), "+v"(wbp16[s][kk]));
            }
    }

    using MMA = typename Epi::Mma;
    using acc_t = typename MMA::acc_t;
    static_assert(!L16 || MMA::kIsInt, "the 16 x 16 x 64 form is the int8 instruction");
    v4i acc16[2][4][2];  // L16: [n-half][token tile][channel tile]
    if constexpr (L16) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc16[a][b][c] = (v4i){0, 0, 0, 0};
    }
    acc_t acc[2][2];  // [n-half][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (acc_t){0};

    const int nt = kt1 - kt0;  // K-tiles of this block (>= 1)
    const int klast = (nt - 1) * 128;

    auto dma = [&](int kind, int i, int stage, int k0) {
        const int8_t *b = (kind == 0 ? xbase : wbase) + k0;  // SALU
        p8_dma16(b, voff[kind][i], dma_dst + stage * P8H_STAGE + kind * P8_UNIT + i * 1024);
    };
    auto issue_a = [&](int stage, int k0) { dma(0, 0, stage, k0); dma(0, 1, stage, k0); };
    auto issue_b = [&](int stage, int k0) { dma(1, 0, stage, k0); dma(1, 1, stage, k0); dma(2, 0, stage, k0); dma(2, 1, stage, k0); };

    // ---- prologue: K-tiles 0 and 1 (clamped), wait for X(0), W-even(0)
    {
        const int k1 = 128 < klast ? 128 : klast;
        issue_a(0, 0);
        issue_b(0, 0);
        issue_a(1, k1);
        issue_b(1, k1);
    }
    P8_WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger: the wm=1 group runs one barrier behind
    P8H_BLK(1);

    v4i xf[2][4], wf[4];
    v4i xf16[4][2], wf16[2][2];   // L16: [tile][k-step]
    auto quadrant16 = [&](v4i (&A)[4][2]) {   // the activation fragment stays for two instructions, the weight fragments alternate
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it) A[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf16[it][kk], xf16[jt][kk], A[jt][it], 0, 0, 0);
    };
    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = (S + 2) % 3;
        int kn = (t + 2) * 128;  // SALU
        kn = kn < klast ? kn : klast;

        // ---------------- P1: n-half 0
        issue_a(NS, kn);
        if constexpr (L16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 2; ++it) wf16[it][kk] = *(p8_lds_v4i)(uintptr_t)(wbp16[S][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) xf16[jt][kk] = *(p8_lds_v4i)(uintptr_t)(xb16[S][kk] + jt * 2048);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *(p8_lds_v4i)(uintptr_t)(wbp[S][ks] + 1 * P8_UNIT);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xf[j][ks] = *(p8_lds_v4i)(uintptr_t)(xb[S][ks] + j * 4096);
        }
        P8_WAIT_VM(8);
        __builtin_amdgcn_s_barrier();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
if constexpr (L16) {
            quadrant16(acc16[0]);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j] = MMA::mma(wf[ks], xf[j][ks], acc[0][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j] = MMA::mma2(wf[kp], wf[kp + 1], xf[j][kp], xf[j][kp + 1], acc[0][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();

        // ---------------- P2: n-half 1
        issue_b(NS, kn);
        if constexpr (L16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 2; ++it) wf16[it][kk] = *(p8_lds_v4i)(uintptr_t)(wbp16[S][kk] + 2 * P8_UNIT + it * 2048);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *(p8_lds_v4i)(uintptr_t)(wbp[S][ks] + 2 * P8_UNIT);
        }
        P8_WAIT_VM(8);
        __builtin_amdgcn_s_barrier();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
if constexpr (L16) {
            quadrant16(acc16[1]);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][j] = MMA::mma(wf[ks], xf[j][ks], acc[1][j]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][j] = MMA::mma2(wf[kp], wf[kp + 1], xf[j][kp], xf[j][kp + 1], acc[1][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    int t = 0;
    for (; t + 2 < nt; t += 3) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
        ktile(std::integral_constant<int, 2>{}, t + 2);
    }
    if (t < nt) ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) ktile(std::integral_constant<int, 1>{}, t + 1);

    P8H_BLK(2);
    P8_WAIT_VM(0);                                // drain the dead prefetches before LDS is released
    if (wm == 0) __builtin_amdgcn_s_barrier();    // balance the stagger barrier

    if constexpr (L16) {
        // accumulator tile (in16 = 16-channel tile 0..3, im16 = 16-token tile 0..3) -> rows m0 + wm*64 + 16*im16, cols n0 + wn*64 + 16*in16
        auto get16 = [&](int in16, int im16) -> const v4i & { return acc16[in16 >> 1][im16][in16 & 1]; };
        bool staged16 = false;
        if constexpr (Epi::kOutBytes >= 2) staged16 = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);
        const int64_t mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;
        if (staged16) {
            if constexpr (Epi::kOutBytes >= 2) {
                __builtin_amdgcn_s_barrier();
                bool rows_path = false;
                if constexpr (Epi::kOutBytes == 2) rows_path = mw0 + 64 <= M && nw0 + 64 <= N && epi.N < (int64_t(1) << 27);
                if (rows_path) {
                    if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<2, 2, true>(epi, get16, mw0, nw0, lane, lds0 + wave * 16384, rows_write_through(M, epi));
                } else
                    epilogue_wave_staged<2, 0, true>(epi, get16, mw0, nw0, lane, M, N, lds0 + wave * 16384);
            }
        } else {
            epilogue_wave16<4>(epi, get16, mw0, nw0, lane, M, N);
        }
        return;
    }
    // accumulator tile (in = n-half, im = j) -> rows m0 + wm*64 + 32*im, cols n0 + wn*64 + 32*in
    auto get = [&](int in, int im) -> const acc_t & { return acc[in][im]; };
    bool staged = false;
    if constexpr (Epi::kOutBytes >= 2) staged = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);  // N: this launch's column bound, epi.N: the row stride
    if (staged) {
        if constexpr (Epi::kOutBytes >= 2) {
            __builtin_amdgcn_s_barrier();  // all ring reads done, all (dead) DMAs landed: the ring becomes staging space
            bool rows_path = false;
            if constexpr (Epi::kOutBytes == 2) rows_path = m0 + wm * 64 + 64 <= M && n0 + wn * 64 + 64 <= N && epi.N < (int64_t(1) << 27);   // interior wave tile
            if (rows_path) {
                if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<2, 2>(epi, get, m0 + wm * 64, n0 + wn * 64, lane, lds0 + wave * 16384);
            } else
                epilogue_wave_staged<2>(epi, get, m0 + wm * 64, n0 + wn * 64, lane, M, N, lds0 + wave * 16384);
        }
    } else {
        epilogue_wave<2, 2>(epi, get, [](int im) { return im * 32; }, m0 + wm * 64, n0 + wn * 64, lane, M, N);
    }
    P8H_PROBE_END(tile_m, tile_n);
}

}  // namespace asq
