// "p16t": gemm_i8_p16 with THE TAIL -- the single-round 256 x 256 x 128 kernel whose last two K-tiles run m-half first, so that half of every block's epilogue is
// issued under the partner wave's last 64 MFMAs instead of behind the K loop.  EXPERIMENTAL, NOT BUILT BY DEFAULT: bit-identical to gemm_i8_p16
// (tests/test_hip_tail.py on a -DASQ_P16_TAIL=1 build) and no faster -- on the bench operands the launch is energy-bound, on zeros (cycle-bound) the E0 wave does not
// advance beside its SIMD partner's prio-1 MFMA stream: profiles/r6_tail_overlap_ab.txt.  Kept as the record of what was built; `make FLAGS+=-DASQ_P16_TAIL=1`
// builds it and launch_gemm_impl dispatches it for launches of interior tiles only (M % 256 == 0, N % 256 == 0, K % 256 == 0, K >= 512, aligned 2-byte output rows);
// ASQ_P16_TAIL=0/1 in the environment then switches inside that build.  Everything up to the last three K-tiles is gemm_i8_p16's code.
#pragma once

namespace asq {

template <class Epi, int ABL = 0>   // launches of interior block tiles only, aligned 2-byte output rows, an even number >= 4 of K-tiles (launch_gemm_impl checks)
__global__ void __launch_bounds__(512, 2) gemm_i8_p16t(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                      int tiles_m, int tiles_n, Epi epi_in, OffsetArgs off)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    static_assert(P8_ABL_OK(ABL), "probe variants need -DASQ_P8_PROBE (tools/ubench)");
    static_assert(Epi::Mma::kIsInt, "int8 operands");

    P8_BLK_RT(6);
    P8_BLK(0);
    constexpr int GM = 4;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const Epi epi = epi_in.rebased(0, 0, M, N);
    const int per_group = GM * tiles_n;
    const int group = id / per_group, in_group = id - group * per_group;
    const int first_m = group * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;

    // OFFSET operands (OffsetArgs, asq_gemm_kernels.h): x and w hold X + cx[m] and W + cw[n]; the K loop is the plain one on them and the exact product is
    // recovered in front of the epilogue: acc -= cx[m] * wsum[n] + cw[n] * xsum'[m] (int32 arithmetic wraps, in the matrix cores too), two v_mad_i32_i24 per
    // element.  The tile's 256 {cx, xsum'} and 256 {cw, wsum} pairs are fetched by ONE 8-byte load per thread, first thing in the kernel (inline asm, like the
    // DMAs: hipcc counts neither; the prologue's vmcnt(4) covers it -- in-order return), and parked in the 3 KiB of LDS behind the ring until the K loop is done.
    // (Round-4 measurements, tools/ubench/clock_probe off.  The same correction as the accumulators' START values costs 1.75-2.0 k cycles of prologue whatever the
    // form -- 16 per-lane loads per wave or one load per thread staged through LDS, 256 or 144 VALU operations: the first data of a launch arrives ~2.5 k cycles
    // after its start whatever is asked for, so nothing that waits for global data is covered by the first K-tile's latency -- and the large start values cost the
    // K loop ~1.5 % of clock.  As plain C loads in the prologue the compiler's own vmcnt(7..0), counted without the 8 younger DMAs, held the start values
    // back until the whole first K-tile had landed; split over two conditional regions they even leaked vmcnt(0) into the K loop: 48 -> 55 us.)
    const bool offs = off.row != nullptr;   // (block-uniform)
    v2i opair = {0, 0};
    if (offs) {
        const int i = tid & 255;
        const int64_t idx = tid < 256 ? (m0 + i < M ? m0 + i : M - 1) : (n0 + i < N ? n0 + i : N - 1);
        const int32_t *src = (tid < 256 ? off.row : off.col) + 2 * idx;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(opair) : "v"(src) : "memory");
    }

    // ---- DMA sources (as p8): this wave fills row-groups 2*wave, 2*wave+1 (8 rows each) of every unit
    const int nt = (int)(K / 128);
    const int8_t *xbase = uniform_ptr(x + m0 * K);
    const int8_t *wbase = uniform_ptr(w + n0 * K);
    const int64_t mrem = M - m0 - 1, nrem = N - n0 - 1;  // last valid local row
    unsigned voff[4][2];  // [kind][i]: X-even, W-even, W-odd, X-odd
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ru = (wave * 2 + i) * 8 + (lane >> 3);                      // row within the unit, 0..127
        const unsigned cb = (unsigned)(((lane & 7) ^ ((ru >> 1) & 7)) * 16);  // swizzled source chunk
        int64_t rxe = (ru >> 6) * 128 + (ru & 63), rxo = rxe + 64;            // local tile rows
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
    unsigned dma_dst = lds0 + wave * 2048;  // + stage*P8_STAGE + kind*P8_UNIT + i*1024

    // ---- fragment read addresses: one VGPR per (stage, operand, k-step of 64); + 2048 * tile (16 rows) + the unit as immediates
    const int t16 = lane & 15, q16 = lane >> 4, sw = (t16 >> 1) & 7;
    unsigned xb[2][2], wbp[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned off = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw)) << 4) + s * P8_STAGE;
            xb[s][kk] = off + wm * 64 * 128;   // X units: this wave's 64 rows
            wbp[s][kk] = off + wn * 32 * 128;  // W units: this wave's 32 rows
            asm volatile("" : "+v"(xb[s][kk]), "+v"(wbp[s][kk]));
        }

    const int klast = (nt - 1) * 128;
    auto issue = [&](int kind, int stage, int k0) {
        const int8_t *b = ((kind == 0 || kind == 3) ? xbase : wbase) + k0;  // SALU
#pragma unroll
        for (int i = 0; i < 2; ++i) p8_dma16(b, voff[kind][i], dma_dst + stage * P8_STAGE + kind * P8_UNIT + i * 1024);
    };

    // Accumulators start at 0, on plain operands and on offset images alike (the images' correction waits in LDS for the epilogue, above).
    // ---- prologue: K-tile 0 entirely (units 0..3), wait for the two units P1 reads
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) issue(kind, 0, 0);
    v4i acc[2][2][4][2];  // [m-half][n-half][token tile][channel tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (v4i){0, 0, 0, 0};
    P8_WAIT_VM(4);
    typedef __attribute__((address_space(3))) v2i *lds_v2i;
    typedef __attribute__((address_space(3))) v4i *lds_v4i_;
    typedef __attribute__((address_space(3))) int *lds_i32;
    const unsigned obase = lds0 + P8_LDS_BYTES;   // [256 row words: (-cx) << 24 | (-xsum') & 0xFFFFFF][256 column pairs {cw, wsum}]
    if (offs) {
        asm volatile("" : "+v"(opair));   // (keeps every use of the loaded pair behind the wait above)
        if (tid < 256) *(lds_i32)(uintptr_t)(obase + tid * 4) = (int)(((unsigned)(-opair[0]) << 24) | ((unsigned)(-opair[1]) & 0xFFFFFFu));   // |cx| <= 64, |xsum'| < 2^23
        else *(lds_v2i)(uintptr_t)(obase + 1024 + (tid - 256) * 8) = opair;
    }
    P8_BAR();
    if (wm == 1) P8_BAR();  // stagger: the wm=1 group runs one barrier behind
    P8_BLK(1);

    v4i xf[4][2], wa[2][2], wb[2][2];  // [tile][k-step]
    auto ld = [&](unsigned a) { return *(p8_lds_v4i)(uintptr_t)a; };
    // 16 MFMAs of one quadrant: k-step outermost; the activation fragment stays for two instructions, the weight fragments alternate (measured against the
    // weight-stationary, snaking and token-tile-major orders in rounds 3 / 4: profiles/r3_p16_order_ab.txt, r4_p16_order_images.txt -- within noise, this one kept)
    auto quadrant = [&](v4i (&A)[4][2], const v4i (&wf)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it) A[jt][it] = MmaI8x16::mma(wf[it][kk], xf[jt][kk], A[jt][it]);
        }
    };

    // one K-tile at LDS stage S (compile-time), prefetching K-tile (t+1) into stage S^1: p8's four phases
    // PRE (the tail below): this is K-tile nt - 3; besides K-tile nt - 2 (into NS) it also fetches the X-even and W-even units of K-tile nt - 1 into its OWN stage,
    // whose two units were last read in P1 (>= 2 phases before: WAR as in the steady state), and P4 waits with two more units in flight.
    auto ktile = [&](auto stage_tag, auto pre_tag, int t) {
        constexpr int S = decltype(stage_tag)::value, NS = S ^ 1;
        constexpr bool PRE = decltype(pre_tag)::value;
        int kn = (t + 1) * 128;
        kn = kn < klast ? kn : klast;

        // ---------------- P1: (m-half 0, n-half 0): reads W-even (4) + X-even (8)
        issue(0, NS, kn);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wa[it][kk] = ld(wbp[S][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[S][kk] + 0 * P8_UNIT + jt * 2048);
        P8_WAIT_VM(4);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][0], wa);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();

        // ---------------- P2: (m-half 0, n-half 1): reads W-odd (4)
        issue(1, NS, kn);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wb[it][kk] = ld(wbp[S][kk] + 2 * P8_UNIT + it * 2048);
        P8_WAIT_VM(4);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][1], wb);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();

        // ---------------- P3: (m-half 1, n-half 1): reads X-odd (8)
        issue(2, NS, kn);
        if constexpr (PRE) issue(0, S, klast);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[S][kk] + 3 * P8_UNIT + jt * 2048);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[1][1], wb);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();

        // ---------------- P4: (m-half 1, n-half 0): W-even fragments are still in wa[]
        issue(3, NS, kn);
        if constexpr (PRE) {
            issue(1, S, klast);
            P8_WAIT_VM(8);   // in flight: W-odd, X-odd of K-tile nt - 2; X-even, W-even of K-tile nt - 1
        } else {
            P8_WAIT_VM(4);
        }
        P8_BAR();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[1][0], wa);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
    };

    using std::integral_constant;
    using std::false_type;
    using std::true_type;
    const int64_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
    const bool wt = M * epi.N * Epi::kOutBytes <= (int64_t)ASQ_WT_BYTES && ASQ_WT_BYTES > 0;   // small outputs leave as write-through stores (rows_write_through)
    // accumulator tile (in16 = 16-channel tile 0..3, im16 = 16-token tile 0..7) -> rows mw0 + 16*im16, cols nw0 + 16*in16
    auto get = [&](int in16, int im16) -> const v4i & { return acc[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1]; };
    // offset images: this lane's 16 channels {cw, wsum} (32 registers) and its 8 tokens' packed words (8): two v_mad_i32_i24 per element (v_mul_i32_i24 semantics:
    // the low 24 bits of either factor, sign-extended -- the packed word IS -xsum' there) + one shift per accumulator tile for -cx
    // (pairs packed as cw << 24 | wsum: tried for the fp32 column-scale + bias epilogue, the compiler hoists the unpacking and spills more)
    int cw[4][4], ws[4][4], rw[8];
    auto load_pairs = [&]() {
        const int t16i = lane & 15, q16i = lane >> 4;
#pragma unroll
        for (int in16 = 0; in16 < 4; ++in16) {
            const unsigned ca = obase + 1024 + (wn * 64 + in16 * 16 + 4 * q16i) * 8;
            const v4i p0 = *(lds_v4i_)(uintptr_t)ca, p1 = *(lds_v4i_)(uintptr_t)(ca + 16);   // channels n, n+1 | n+2, n+3
            const int c4[4] = {p0[0], p0[2], p1[0], p1[2]}, s4[4] = {p0[1], p0[3], p1[1], p1[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) cw[in16][e] = c4[e], ws[in16][e] = s4[e];
        }
#pragma unroll
        for (int im16 = 0; im16 < 8; ++im16) rw[im16] = *(lds_i32)(uintptr_t)(obase + (wm * 128 + im16 * 16 + t16i) * 4);
    };
    auto getc = [&](int in16, int im16) -> v4i {
        const v4i &a = acc[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1];
        const int ncx = rw[im16] >> 24;
        v4i o;
#pragma unroll
        // two v_mad_i32_i24, spelled out: from the C expression hipcc builds v_mul_i32_i24 x 2 (the >> 24 folded into an SDWA byte select) + v_add3_u32,
        // three VALU operations per element where two do (128 elements per lane, two waves per SIMD: ~1 k cycles of a 85 k-cycle block)
        for (int e = 0; e < 4; ++e) o[e] = mad24(cw[in16][e], rw[im16], mad24(ncx, ws[in16][e], a[e]));
        return o;
    };

    // ---- THE TAIL (round 6, TAIL launches: 2-byte outputs, interior block tiles only, an even number >= 4 of K-tiles -- the host checks, p16_tail_ok): the block's last two K-tiles, n2 = nt - 2 (stage 0) and
    // n1 = nt - 1 (stage 1), are BOTH resident when the last eight phases start, and these run m-half first instead of K-tile first:
    //     s0 Q(0,0) n2 | s1 Q(0,1) n2 | s2 Q(0,0) n1 | s3 Q(0,1) n1 || free-running:  M1 = Q(1,1) n2, Q(1,0) n2, Q(1,1) n1, Q(1,0) n1
    // After s3 the wave tile's upper 64 rows are final, every DMA has landed and the X-even units of both stages are dead: no wave needs another barrier.  Each
    // wave stages its epilogue in a PRIVATE 4 KiB image inside them (epilogue_wave_rows<.., IMGS = 1, IM0, IM1>), and the two waves of a SIMD take the remaining
    // work in opposite orders -- the wm = 0 wave writes its upper half (E0) while the wm = 1 wave runs its last 64 MFMAs (M1), then they swap -- so half of the
    // tile's conversions, LDS round trips and output bytes leave under matrix work instead of behind it, and the chip's 256 simultaneous output bursts start
    // ~2 k cycles earlier.  The unit fetches move with it (vmcnt counts = DMAs that may stay in flight, two per unit):
    //     K-tile nt - 3 (PRE):  P3 + X-even n1, P4 + W-even n1, wait 8  |  s0: W-odd n1, wait 8  |  s1: X-odd n1, wait 4  |  s2: wait 2  |  s3: wait 0
    // Results are bit-identical (the same integer sums; the same epilogue arithmetic per element).
    static_assert(Epi::kOutBytes == 2, "the tail stages 2-byte outputs");
    int t = 0;
    const int nplain = nt - 4;
    for (; t + 1 < nplain; t += 2) {
        ktile(integral_constant<int, 0>{}, false_type{}, t);
        ktile(integral_constant<int, 1>{}, false_type{}, t + 1);
    }
    {
        // The DMA bases again, through v_readfirstlane: hipcc hoists the loop's `dma_dst + constant` sums, merges them with the copies of the code below in PHIs of
        // the (possibly skipped) loop's exit and has been seen to keep those in VGPRs -- which the DMAs' "s" operands then receive unrepaired.
        dma_dst = __builtin_amdgcn_readfirstlane(dma_dst);
        xbase = uniform_ptr(xbase);
        wbase = uniform_ptr(wbase);
        ktile(integral_constant<int, 0>{}, false_type{}, nt - 4);
        ktile(integral_constant<int, 1>{}, true_type{}, nt - 3);
        // ---------------- s0: Q(0,0) of n2
        issue(2, 1, klast);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wa[it][kk] = ld(wbp[0][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[0][kk] + 0 * P8_UNIT + jt * 2048);
        P8_WAIT_VM(8);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][0], wa);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
        // ---------------- s1: Q(0,1) of n2
        issue(3, 1, klast);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wb[it][kk] = ld(wbp[0][kk] + 2 * P8_UNIT + it * 2048);
        P8_WAIT_VM(4);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][1], wb);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
        // ---------------- s2: Q(0,0) of n1
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wa[it][kk] = ld(wbp[1][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[1][kk] + 0 * P8_UNIT + jt * 2048);
        P8_WAIT_VM(2);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][0], wa);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
        // ---------------- s3: Q(0,1) of n1; the last wait: everything has landed
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wb[it][kk] = ld(wbp[1][kk] + 2 * P8_UNIT + it * 2048);
        P8_WAIT_VM(0);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][1], wb);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 0) P8_BAR();   // the wm = 0 group's closing barrier pairs with the wm = 1 group's barrier above: the stagger is balanced, nobody waits again
        P8_BLK(2);

        // ---------------- free-running: M1 (the lower 64 rows' last two K-tiles), E0 / E1 (the upper / lower 64 rows' epilogue)
        auto m1 = [&]() {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[s][kk] + 3 * P8_UNIT + jt * 2048);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        wb[it][kk] = ld(wbp[s][kk] + 2 * P8_UNIT + it * 2048);
                        wa[it][kk] = ld(wbp[s][kk] + 1 * P8_UNIT + it * 2048);
                    }
                P8_WAIT_LGKM0();
                __builtin_amdgcn_sched_barrier(0);
                P8_PRIO(1);
                quadrant(acc[1][1], wb);
                quadrant(acc[1][0], wa);
                P8_PRIO(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const unsigned img = lds0 + (wave >> 2) * P8_STAGE + (wave & 3) * 4096;   // this wave's image: a quarter of the (dead) X-even unit of stage wave / 4
        if (wm == 1) m1();
        if (offs) {
            load_pairs();
            epilogue_wave_rows<4, 2, true, 1, false, 0, 2>(epi, getc, mw0, nw0, lane, img, wt);
        } else {
            epilogue_wave_rows<4, 2, true, 1, false, 0, 2>(epi, get, mw0, nw0, lane, img, wt);
        }
        if (wm == 0) m1();
        if (offs) epilogue_wave_rows<4, 2, true, 1, false, 2, 4>(epi, getc, mw0, nw0, lane, img, wt);
        else epilogue_wave_rows<4, 2, true, 1, false, 2, 4>(epi, get, mw0, nw0, lane, img, wt);
    }
    P8_PROBE_END();
}

}  // namespace asq
