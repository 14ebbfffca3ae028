// "p8q": 128(m) x 128(n) x 128(k) member of the p8 family for the mid-size row counts whose 128 x 256 tiling (p8h) leaves half the
// chip idle -- 1024 x 4096: 128 tiles of p8h on 256 CUs; here 256.  Same machinery as p8h (LDS-DMA in full 128-B rows, XOR-swizzled
// 16 KiB units, counted vmcnt, two wave groups staggered by one barrier, staged coalescing epilogue, split-K slabs) with ONE phase per
// K-tile:
//
//   8 waves = 4 (m) x 2 (n); wave (wm, wn) owns out[wm*32 .. +32][wn*64 .. +64] = 2 MFMA tiles (n-halves) of 32 x 32.
//   K-tile image = 2 units {X: tile rows 0..127, W: tile columns 0..127} = 32 KiB; ring of FOUR K-tiles (128 KiB): tile t+3 is fetched
//   while tile t is consumed -- three K-tiles (~2000 cycles) of lead, what p8 / p8h have with their longer K-tiles.
//   Per K-tile and wave: 4 DMAs (X[0] X[1] W[0] W[1] of tile t+3), 12 fragment reads (4 X + 8 W), 8 MFMAs.
//   Counted wait: tile t+1 must have landed before its fragments are read; tiles t+2, t+3 (8 DMAs) may stay in flight -> vmcnt(8).
//   Unlike p8 / p8h (two wave groups that alternate load and MFMA segments between barriers) the fragments are double-buffered in
//   registers: tile t+1 is read from LDS while the matrix cores work on tile t, ONE barrier per K-tile.  (With the p8h structure the load
//   segment -- 4 DMAs + 12 ds_reads + their latency -- is as long as two MFMA segments of this small tile: 1100 cycles per K-tile measured.)
//
// The small tile pays for its parallelism: 32 KiB of DMA and 12 KiB of fragment reads per wave for 8 MFMAs per K-tile are twice p8's
// bytes per MFMA, so its K loop runs at the vector-memory path's pace, not the matrix cores' (see DESIGN 4 for the measured cycles).
// The dispatcher uses it only where p8h would leave CUs idle.
#pragma once
#include <type_traits>

namespace asq {

constexpr int P8Q_STAGE = 2 * P8_UNIT;        // 32 KiB
constexpr int P8Q_LDS_BYTES = 4 * P8Q_STAGE;  // 128 KiB
constexpr int P8Q_MX_LDS_BYTES = P8Q_LDS_BYTES + 2 * 4096;  // + two buffers of MX scale bytes (256 rows x 16 B = four K-tiles each)

// MX = true (fp8 only, asq_linear_mxfp8): xs / ws are the operands' E8M0 block scales [rows][K/32]; K % 512 == 0.  Every fourth K-tile waves 0..3
// fetch the next 16 scale bytes of the tile's 128 + 128 rows with one LDS-DMA each; a lane reads the dword of its row and K-tile next to the
// fragments and hands byte `hi` (k 0..63 of the tile) / byte 2 + `hi` (k 64..127) to v_mfma_scale_f32_32x32x64_f8f6f4 -- the lane (r, h) of that
// instruction supplies the scale of the h-th 32-k block of row r.
// L16 (int8 only): the matrix work on v_mfma_i32_16x16x64_i8 (asq_gemm_p16.h for why): per K-tile 2 k-steps x {2 token tiles x 4 channel tiles}
// = 16 instructions of 16 cycles on acc16[token tile][channel tile]; fragments 16 rows x 64 k-bytes from the same unit images.
template <class Epi, bool MX = false, bool L16 = false>
__global__ void __launch_bounds__(512, 2) gemm_i8_p8q(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                      int tiles_m, int tiles_n, int ksplit, Epi epi_in, const uint8_t *__restrict__ xs,
                                                      const uint8_t *__restrict__ ws)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

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
    const int64_t m0 = (int64_t)tile_m * 128, n0 = (int64_t)tile_n * 128;

    const int nt_all = (int)(K / 128);
    const int kt0 = (int)((int64_t)nt_all * split / ksplit), kt1 = (int)((int64_t)nt_all * (split + 1) / ksplit);
    const int8_t *const xbase = uniform_ptr(x + m0 * K + (int64_t)kt0 * 128);
    const int8_t *const wbase = uniform_ptr(w + n0 * K + (int64_t)kt0 * 128);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;  // last valid local row
    unsigned voff[2][2];  // [kind: X, W][i]; this wave fills row-groups 2*wave, 2*wave+1 (8 rows each) of both units
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ru = (wave * 2 + i) * 8 + (lane >> 3);                       // row within the unit, 0..127
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);  // swizzled source chunk
        const int64_t rx = ru < mrem ? ru : mrem, rw = ru < nrem ? ru : nrem;
        voff[0][i] = (unsigned)(rx * K) + cb;
        voff[1][i] = (unsigned)(rw * K) + cb;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned dma_dst = lds0 + wave * 2048;  // + stage*P8Q_STAGE + kind*P8_UNIT + i*1024

    // fragment read addresses: one VGPR per (stage, k-substep) and operand; the rest is an immediate
    const int frow = lane & 31, sw = (frow >> 1) & 7, hi = lane >> 5;
    unsigned xb[4][4], wbp[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = lds0 + frow * 128 + ((((ks * 2 + hi) ^ sw)) << 4) + s * P8Q_STAGE;
            xb[s][ks] = off + wm * 32 * 128;             // X unit: this wave's 32 rows
            wbp[s][ks] = off + P8_UNIT + wn * 64 * 128;  // W unit: this wave's 64 rows (+ in * 4096)
            asm volatile("" : "+v"(xb[s][ks]), "+v"(wbp[s][ks]));
        }

    unsigned xb16[4][2], wbp16[4][2];   // L16: one VGPR per (stage, k-step of 64) and operand
    if constexpr (L16) {
        const int t16 = lane & 15, q16 = lane >> 4, sw16 = (t16 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned off = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw16)) << 4) + s * P8Q_STAGE;
                xb16[s][kk] = off + wm * 32 * 128;
                wbp16[s][kk] = off + P8_UNIT + wn * 64 * 128;
                asm volatile("" : "+v"(xb16[s][kk]), "+v"(wbp16[s][kk]));
            }
    }

    using MMA = typename Epi::Mma;
    using acc_t = typename MMA::acc_t;
    static_assert(!L16 || (MMA::kIsInt && !MX), "the 16 x 16 x 64 form is the int8 instruction");
    acc_t acc[2];  // [n-half]
    acc[0] = (acc_t){0};
    acc[1] = (acc_t){0};
    v4i acc16[2][4];  // L16: [token tile][channel tile]
    if constexpr (L16) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc16[a][b] = (v4i){0, 0, 0, 0};
    }

    const int nt = kt1 - kt0;  // K-tiles of this block (>= 1)
    const int klast = (nt - 1) * 128;

    auto issue = [&](int stage, int k0) {
        p8_dma16(xbase + k0, voff[0][0], dma_dst + stage * P8Q_STAGE);
        p8_dma16(xbase + k0, voff[0][1], dma_dst + stage * P8Q_STAGE + 1024);
        p8_dma16(wbase + k0, voff[1][0], dma_dst + stage * P8Q_STAGE + P8_UNIT);
        p8_dma16(wbase + k0, voff[1][1], dma_dst + stage * P8Q_STAGE + P8_UNIT + 1024);
    };

    // ---- MX: scale bytes of K-tile group g (tiles 4g .. 4g+3) -> scale buffer g & 1; waves 0,1: the 128 X rows, waves 2,3: the 128 W rows
    const unsigned sc0 = lds0 + P8Q_LDS_BYTES;
    [[maybe_unused]] unsigned sc_voff = 0, sx_addr = 0, sw_addr = 0;
    [[maybe_unused]] const int ngrp = nt_all / 4;
    if constexpr (MX) {
        const int kind = wave >> 1, rl = (wave & 1) * 64 + lane;
        int64_t row = (kind ? n0 : m0) + rl;
        const int64_t lim = (kind ? N : M) - 1;
        row = row < lim ? row : lim;
        sc_voff = (unsigned)(row * (K / 32));
        sx_addr = sc0 + (wm * 32 + frow) * 16;
        sw_addr = sc0 + 2048 + (wn * 64 + frow) * 16;  // (+ in * 512)
    }
    auto issue_scales = [&](int g) {
        if constexpr (MX) {
            if (wave < 4) {
                const int gc = g < ngrp ? g : ngrp - 1;  // dead prefetch past the end: stay inside the row
                p8_dma16((const int8_t *)((wave >> 1) ? ws : xs) + 16 * gc, sc_voff, sc0 + (g & 1) * 4096 + (wave >> 1) * 2048 + (wave & 1) * 1024);
            }
        }
    };

    // ---- prologue: K-tiles 0, 1, 2 (clamped); fragments of tile 0
    issue_scales(0);  // (oldest: landed with tile 0)
    issue(0, 0);
    issue(1, 128 < klast ? 128 : klast);
    issue(2, 256 < klast ? 256 : klast);
    P8_WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    v4i xf[2][4], wf[2][2][4];  // [register set = K-tile parity]
    v4i xf16[2][2][2], wf16[2][4][2];  // L16: [register set][tile][k-step]
    [[maybe_unused]] unsigned sxr[2] = {0, 0}, swr[2][2] = {{0, 0}, {0, 0}};  // MX: the row's 4 scale bytes of the K-tile, shifted so that byte 0 / 2 are this lane's
    typedef const __attribute__((address_space(3))) unsigned *p8q_lds_u32;
    auto read_frags = [&](auto stage_tag, auto set_tag, int tile) {
        constexpr int S = decltype(stage_tag)::value, R = decltype(set_tag)::value;
        if constexpr (L16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) xf16[R][jt][kk] = *(p8_lds_v4i)(uintptr_t)(xb16[S][kk] + jt * 2048);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int it = 0; it < 4; ++it) wf16[R][it][kk] = *(p8_lds_v4i)(uintptr_t)(wbp16[S][kk] + it * 2048);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xf[R][ks] = *(p8_lds_v4i)(uintptr_t)(xb[S][ks]);
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[R][in][ks] = *(p8_lds_v4i)(uintptr_t)(wbp[S][ks] + in * 4096);
        }
        if constexpr (MX) {
            const unsigned off = ((tile >> 2) & 1) * 4096 + S * 4;  // scale buffer of the tile's group, dword of the tile (S = tile % 4)
            sxr[R] = *(p8q_lds_u32)(uintptr_t)(sx_addr + off);
            swr[R][0] = *(p8q_lds_u32)(uintptr_t)(sw_addr + off);
            swr[R][1] = *(p8q_lds_u32)(uintptr_t)(sw_addr + 512 + off);
        }
    };
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);

    // One barrier per K-tile.  Iteration t: re-fill the slot of tile t-1 with tile t+3 (its fragments were read during iteration t-2 and
    // waited for in iteration t-1, before the barrier every wave has passed to get here); make sure tile t+1 has landed and the
    // fragments of tile t are in registers; barrier; read the fragments of tile t+1 into the other register set WHILE the matrix
    // cores work on tile t.
    auto ktile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = (S + 3) % 4, S1 = (S + 1) % 4, R = S & 1;
        int kn = (t + 3) * 128;  // SALU
        kn = kn < klast ? kn : klast;
        issue(NS, kn);
        if constexpr (MX && S == 0) issue_scales((t >> 2) + 1);  // after the tile's DMAs: three more K-tiles of DMAs follow before it is needed
        P8_WAIT_VM(8);   // tile t+1 has landed (this wave's part); tiles t+2, t+3 stay in flight
        P8_WAIT_LGKM0();  // fragments of tile t
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(std::integral_constant<int, S1>{}, std::integral_constant<int, R ^ 1>{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MX) {
            const int sx = (int)(sxr[R] >> (8 * hi));
#pragma unroll
            for (int in = 0; in < 2; ++in) {
                const int swv = (int)(swr[R][in] >> (8 * hi));
                const v8i A0 = {wf[R][in][0][0], wf[R][in][0][1], wf[R][in][0][2], wf[R][in][0][3], wf[R][in][1][0], wf[R][in][1][1], wf[R][in][1][2], wf[R][in][1][3]};
                const v8i B0 = {xf[R][0][0], xf[R][0][1], xf[R][0][2], xf[R][0][3], xf[R][1][0], xf[R][1][1], xf[R][1][2], xf[R][1][3]};
                acc[in] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B0, acc[in], 0, 0, 0, swv, 0, sx);
                const v8i A1 = {wf[R][in][2][0], wf[R][in][2][1], wf[R][in][2][2], wf[R][in][2][3], wf[R][in][3][0], wf[R][in][3][1], wf[R][in][3][2], wf[R][in][3][3]};
                const v8i B1 = {xf[R][2][0], xf[R][2][1], xf[R][2][2], xf[R][2][3], xf[R][3][0], xf[R][3][1], xf[R][3][2], xf[R][3][3]};
                acc[in] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B1, acc[in], 0, 0, 2, swv, 2, sx);
            }
        } else if constexpr (L16) {   // the activation fragment stays, the weight fragments cycle (asq_gemm_p16.h)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int it = 0; it < 4; ++it) acc16[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf16[R][it][kk], xf16[R][jt][kk], acc16[jt][it], 0, 0, 0);
        } else if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int in = 0; in < 2; ++in) acc[in] = MMA::mma(wf[R][in][ks], xf[R][ks], acc[in]);
        } else {  // fp8: K = 64 block-scaled instruction over two consecutive fragments
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
#pragma unroll
                for (int in = 0; in < 2; ++in) acc[in] = MMA::mma2(wf[R][in][kp], wf[R][in][kp + 1], xf[R][kp], xf[R][kp + 1], acc[in]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
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

    if constexpr (L16) {
        // accumulator tile (in16 = 16-channel tile 0..3, im16 = 16-token tile 0..1) -> rows m0 + wm*32 + 16*im16, cols n0 + wn*64 + 16*in16
        auto get16 = [&](int in16, int im16) -> const v4i & { return acc16[im16][in16]; };
        bool staged16 = false;
        if constexpr (Epi::kOutBytes >= 2) staged16 = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);
        const int64_t mw0 = m0 + wm * 32, nw0 = n0 + wn * 64;
        if (staged16) {
            if constexpr (Epi::kOutBytes >= 2) {
                __builtin_amdgcn_s_barrier();
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
        return;
    }
    // accumulator tile (in = n-half, im = 0) -> rows m0 + wm*32, cols n0 + wn*64 + 32*in
    auto get = [&](int in, int) -> const acc_t & { return acc[in]; };
    bool staged = false;
    if constexpr (Epi::kOutBytes >= 2) staged = ((((uintptr_t)epi.out) & 15) == 0) && (((N | epi.N) * Epi::kOutBytes) % 16 == 0);
    if (staged) {
        if constexpr (Epi::kOutBytes >= 2) {
            __builtin_amdgcn_s_barrier();  // all ring reads done, all (dead) DMAs landed: the ring becomes staging space
            bool rows_path = false;
            if constexpr (Epi::kOutBytes == 2) rows_path = m0 + wm * 32 + 32 <= M && n0 + wn * 64 + 64 <= N && epi.N < (int64_t(1) << 27);   // interior wave tile
            if (rows_path) {
                if constexpr (Epi::kOutBytes == 2) epilogue_wave_rows<1, 2>(epi, get, m0 + wm * 32, n0 + wn * 64, lane, lds0 + wave * 16384);
            } else
                epilogue_wave_staged<1>(epi, get, m0 + wm * 32, n0 + wn * 64, lane, M, N, lds0 + wave * 16384);
        }
    } else {
        epilogue_wave<2, 1>(epi, get, [](int im) { return im * 32; }, m0 + wm * 32, n0 + wn * 64, lane, M, N);
    }
}

}  // namespace asq
