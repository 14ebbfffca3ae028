// "skinny": M <= 64 rows (decode / cfg1).  The problem is a weight stream: W [N,K] int8 is read
// exactly once from HBM and nothing else matters (4096x4096 at M=32: 16.8 MB of W vs 0.13 MB
// of X and 0.26 MB of output) -> the roofline is HBM bandwidth, not MFMA.
//
// Decomposition: one block of WPB (4..16) waves per 16 output channels; the waves split
// K in 128-byte units (wave v takes units v, v+WPB, ...: a block's waves sweep whole rows
// contiguously and each 128-B line is consumed by exactly one wave).  Every unit is two
// v_mfma_i32_16x16x64_i8 steps: W rows are the matrix-core A operand, X rows the B operand,
// loaded STRAIGHT from global memory into VGPRs (no LDS round trip: the tile is never reused
// inside the block), up to 4 units (8 x 16 B of W + 8*MT x 16 B of X per lane) in flight per
// wave before the first MFMA -- at K = 4096 the whole problem is in flight at once.
// The 16 partial accumulators are summed through LDS (integers: exact, order-free) and the
// first MT waves run the fused epilogue.  No split-K across blocks, no workspace, no atomics.
//
// Requirements: K % 128 == 0, x / w 16-B aligned, M <= 64.  N, M arbitrary otherwise (rows are
// clamped for loading and masked at the store).
#pragma once

namespace asq {

// waves per block: 16 when there are few blocks (N small: put the whole problem in flight at once),
// 4 when N/16 blocks already oversubscribe the chip (4 independent blocks per CU overlap each
// other's load latency, reduction and epilogue).
template <class Epi, int MT, int WPB>  // MT = number of 16-row token tiles (1..4)
__global__ void __launch_bounds__(WPB * 64) gemm_i8_skinny(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N,
                                                            int64_t K, Epi epi)
{
    __shared__ __attribute__((aligned(16))) v4i red[WPB][MT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;

    int64_t nrow = n0 + r;
    nrow = nrow < N ? nrow : N - 1;
    const int8_t *wp = w + nrow * K + 16 * g;
    const int8_t *xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int64_t mrow = mt * 16 + r;
        mrow = mrow < M ? mrow : M - 1;
        xp[mt] = x + mrow * K + 16 * g;
    }

    v4i acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4i){0, 0, 0, 0};

    const int nunits = (int)(K / 128);
    constexpr int U = MT <= 2 ? 2 : 1;  // 128-B units per pipeline stage (2 stages live: VGPR budget 128 at 16 waves/CU)
    struct Stage {
        v4i wf[U][2], xf[U][2][MT];
    };
    auto load = [&](Stage &st, int u0) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int u = u0 + i * WPB;
            if (u < nunits) {
                const int64_t kb = (int64_t)u * 128;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#ifdef SK_ABL_NOW   // probe-only ablations (tools/ubench/skinny_abl.hip): results invalid
                    st.wf[i][h] = (v4i){lane, u, h, 1};
#else
                    st.wf[i][h] = __builtin_nontemporal_load((const v4i *)(wp + kb + 64 * h));  // streamed once
#endif
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#ifdef SK_ABL_NOX
                        st.xf[i][h][mt] = (v4i){lane, mt, u, h};
#else
                        st.xf[i][h][mt] = *(const v4i *)(xp[mt] + kb + 64 * h);
#endif
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    st.wf[i][h] = (v4i){0, 0, 0, 0};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) st.xf[i][h][mt] = (v4i){0, 0, 0, 0};
                }
            }
        }
    };
    auto mma = [&](const Stage &st) {
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wf[i][h], st.xf[i][h][mt], acc[mt], 0, 0, 0);
    };

    // two register stages: the loads of stage B are in flight while stage A feeds the matrix core
    constexpr int STEP = WPB * U;
    Stage sa, sb;
    load(sa, wave);
    for (int u0 = wave; u0 < nunits; u0 += 2 * STEP) {
        load(sb, u0 + STEP);
        mma(sa);
        load(sa, u0 + 2 * STEP);
        mma(sb);
    }

    // D layout (16x16): lane owns token m = lane&15 and channels n = 4*(lane>>4) + reg
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave][mt][lane] = acc[mt];
    __syncthreads();
    if (wave < MT) {
        const int mt = wave;
        v4i s = red[0][mt][lane];
#pragma unroll
        for (int v = 1; v < WPB; ++v) s += red[v][mt][lane];
        const int64_t m = mt * 16 + r, n = n0 + 4 * g;
        if (m < M && n < N) {
            const float sr = Epi::kHasRow ? epi.row(m) : 1.0f;
            v4f sc = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
            epi.cols(n, N, sc, bb);
            epi.store4(m, n, s, sr, sc, bb, N);
        }
    }
}

}  // namespace asq
