// "p4x16": 256(m) x 256(n) x 128(k) INT8 GEMM with FOUR waves (one per SIMD, 128 x 128 outputs per wave) on v_mfma_i32_16x16x64_i8.
// Included by asq_gemm_kernels.h after asq_gemm_p4.h (shares its ring, DMA map and helpers).
//
// The two energy levers of the power-limited 4096^3 GEMM in one kernel: gemm_i8_p4's wave tile -- 32 KiB of fragment
// reads per K-tile for 128 x 128 outputs instead of 24 KiB for 128 x 64: one third fewer LDS bytes per MAC -- and gemm_i8_p16's matrix
// instruction (half the accumulator traffic per MAC).
//
//   LDS ring = 2 K-tiles x {X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255} x 16 KiB (p4's: unit image [128 rows][8 x 16 B],
//   chunk ^= (row >> 1) & 7, swizzle applied to the DMA's global source).
//   K-tile t (stage S) = two k-steps of 64 k-bytes, 64 matrix instructions (8 token tiles x 8 channel tiles, 16 cycles each) per k-step;
//   the activation fragment stays for 8 instructions, the weight fragments cycle.
//     k-step 0   MFMAs on register set A | 16 fragment reads of (t, k-step 1) -> set B, one per 4 instructions | LDS-DMAs 5..15 of K-tile t+1 -> stage S^1
//     k-step 1   MFMAs on set B; after 32 of them: s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier (K-tile t+1 has landed, stage S is free), then
//                16 fragment reads of (t+1, k-step 0) from stage S^1 -> set A, one per 2 instructions | LDS-DMAs 0..4 of K-tile t+2 -> stage S
//   Registers: 256 accumulators + 2 x 16 fragments (128) + addresses: ~430 of 512.
//
// Plain launches, int8 operands, 2-byte outputs, scalar / per-token scales (no per-channel vector, no bias: with them the 128 x 128 wave tile's
// column vectors do not fit beside the accumulators); interior wave tiles leave through epilogue_wave_rows<4, 4, L16>, edge tiles through direct stores.
#pragma once

namespace asq {

template <class Epi, int PROBE = 0>
__global__ void __launch_bounds__(256, 1) gemm_i8_p4x16(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                        int tiles_m, int tiles_n, Epi epi_in)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    static_assert(Epi::Mma::kIsInt && Epi::kOutBytes == 2 && !Epi::kHasCol && !Epi::kHasBias, "int8 operands, 2-byte outputs, scalar / per-token scales");

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

    // ---- DMA sources (as p4).  Instruction q = 0..15 of this wave fills piece (q & 3) * 4 + wave (8 rows x 128 B) of unit q >> 2.
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

    // ---- fragment read addresses: one VGPR per (stage, operand, k-step of 64); + 2048 * i (16-row tile i = 0..7) as an immediate
    const int t16 = lane & 15, q16 = lane >> 4, sw = (t16 >> 1) & 7;
    unsigned xa[2][2], wa[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned off = lds0 + s * P4_STAGE + t16 * 128 + ((((kk * 4 + q16) ^ sw)) << 4);
            xa[s][kk] = off + wm * P4_UNIT;
            wa[s][kk] = off + (2 + wn) * P4_UNIT;
            asm volatile("" : "+v"(xa[s][kk]), "+v"(wa[s][kk]));
        }

    v4i acc[8][8];  // [token tile][channel tile]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (v4i){0, 0, 0, 0};

    auto issue_tile = [&](int stage, int k0) {  // all 16 DMAs of one K-tile (k0 = its first k byte)
        const int8_t *xb = xbase + k0, *wb = wbase + k0;
        const unsigned dst = lds0 + stage * P4_STAGE;
#pragma unroll
        for (int q = 0; q < 16; ++q) p4_dma16((q >> 2) < 2 ? xb : wb, voff[q], dst + (q >> 2) * P4_UNIT + ((q & 3) * 4 + wave) * 1024);
    };
    auto dma_q = [&](int q, int stage, int k0) {
        p4_dma16(((q >> 2) < 2 ? xbase : wbase) + k0, voff[q], lds0 + stage * P4_STAGE + (q >> 2) * P4_UNIT + ((q & 3) * 4 + wave) * 1024);
    };
    // fragment i of a k-step (i = 0..7: weight tiles, 8..15: activation tiles)
    auto ldfrag1 = [&](v4i (&fw)[8], v4i (&fx)[8], int stage, int kk, int i) {
        if (i < 8) fw[i] = *(p8_lds_v4i)(uintptr_t)(wa[stage][kk] + i * 2048);
        else fx[i - 8] = *(p8_lds_v4i)(uintptr_t)(xa[stage][kk] + (i - 8) * 2048);
    };
    // One k-step: 64 matrix instructions on (fw, fx); `hook(idx)` runs after instruction number idx
    auto kstep = [&](const v4i (&fw)[8], const v4i (&fx)[8], auto hook) {
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                // (inline asm with the accumulator tied in an AGPR quad: through the builtin the allocator -- all 256 AGPRs are accumulators -- picks the
                // untied form, rotates every accumulator through its neighbour's registers and repairs the permutation with ~200 v_accvgpr moves per trip)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[jt][it]) : "v"(fw[it]), "v"(fx[jt]));
                __builtin_amdgcn_sched_barrier(0);
                hook(jt * 8 + it);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    const int klast = (nt - 1) * 128;
    auto kclamp = [&](int tt) { const int k = tt * 128; return k < klast ? k : klast; };  // past the end: the last tile again, into a dead stage

    // ---- prologue: K-tile 0 entirely, the first 5 DMAs of K-tile 1, fragments of (0, k-step 0)
    issue_tile(0, 0);
    P8_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    v4i fwA[8], fxA[8], fwB[8], fxB[8];
#pragma unroll
    for (int q = 0; q < 5; ++q) dma_q(q, 1, kclamp(1));
#pragma unroll
    for (int i = 0; i < 16; ++i) ldfrag1(fwA, fxA, 0, 0, i);
    P4_BLK(1);

    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = S ^ 1;
        const int kn = kclamp(t + 1), kn2 = kclamp(t + 2);
        kstep(fwA, fxA, [&](int idx) {
            if (idx % 4 == 0) ldfrag1(fwB, fxB, S, 1, idx / 4);                       // 16 fragments of k-step 1
            if (idx % 5 == 2 && idx / 5 < 11) dma_q(5 + idx / 5, NS, kn);             // q = 5..15 of K-tile t+1 (idx 2, 7, ..., 52)
        });
        kstep(fwB, fxB, [&](int idx) {
            if (idx == 31) {
                P8_WAIT_VM(0);      // my share of K-tile t+1 has landed (its last DMA was issued >= 40 instructions ago)
                P8_WAIT_LGKM0();    // my reads of stage S have returned -- K-tile t+2 may overwrite it
                __builtin_amdgcn_s_barrier();
            }
            if (idx >= 32 && idx % 2 == 0) ldfrag1(fwA, fxA, NS, 0, (idx - 32) / 2);   // 16 fragments of (t+1, k-step 0)
            if (idx >= 33 && (idx - 33) % 6 == 0 && (idx - 33) / 6 < 5) dma_q((idx - 33) / 6, S, kn2);   // q = 0..4 of K-tile t+2
        });
    };

    asm volatile(".p2align 8");
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nt) ktile(std::integral_constant<int, 0>{}, t);
    asm volatile("" ::"v"(fwA[0]), "v"(fxA[0]));  // (the last prefetch is dead)
    P8_WAIT_VM(0);                                // (so are the 5 DMAs issued in the last k-step)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");   // the last matrix instructions (asm: invisible to the hazard recogniser) retire before the epilogue reads their registers

    P4_BLK(2);
    // ---- epilogue: the ring becomes staging space (32 KiB per wave: four 8 KiB token-tile images)
    P8_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    const int64_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
    auto get = [&](int in16, int im16) -> const v4i & { return acc[im16][in16]; };
    const bool rows_path = ((((uintptr_t)epi.out) & 15) == 0) && ((epi.N * 2) % 16 == 0) && mw0 + 128 <= M && nw0 + 128 <= N && epi.N < (int64_t(1) << 27);
    if (rows_path) epilogue_wave_rows<4, 4, true>(epi, get, mw0, nw0, lane, lds0 + wave * 32768);
    else epilogue_wave16<8, 8>(epi, get, mw0, nw0, lane, M, N);
    P4_PROBE_END();
}

}  // namespace asq
