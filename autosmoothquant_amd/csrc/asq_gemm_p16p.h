// "p16p": gemm_i8_p16 as a PERSISTENT kernel for multi-round launches (more than one 256 x 256 tile per CU).
//
// What it is for.  time(4096 x 4096 x K) = a + b K with a = 7.3-7.6 us that no operand statistics change (profiles/r4_power_ceiling.md): the first data of a
// block arrives 1.2-1.6 us after its start, its epilogue takes 3-4 us, ~2 us pass between two blocks on a CU -- and a CU holds ONE block (the ring is 128 of its
// 160 KB of LDS), so none of it overlaps another block's K loop: 17 % of every tile of every prefill GEMM.  Here a block keeps its CU and walks its tiles
// (L = blockIdx.x, + gridDim.x, ...; the same XCD-aware tile map as gemm_i8_p16 per round):
//   * the LAST K-tile of tile i prefetches K-tile 0 of tile i + 1 into the LDS stage it would otherwise fill with a dead, clamped prefetch -- the next tile's
//     first-data wait runs under this tile's last 512 MFMAs and its epilogue;
//   * everything the epilogue reads from memory -- per-token scales, column scales, bias, the offset images' {cx, xsum'} / {cw, wsum} pairs -- comes by four
//     4-byte LDS-DMAs per wave, issued at the start of that last K-tile for the NEXT tile into one of two 8 KiB operand areas behind the ring: the epilogue
//     has no global load, so nothing in it waits behind the prefetched K-tile (gfx950 has ONE vmcnt for loads and stores; loads return in order);
//   * the epilogue stages in the OTHER LDS stage only (8 KiB per wave: two 4 KiB images used alternately, epilogue_wave_rows<.., IMGS = 2>) and, between its
//     first conversions and its first store, retires the in-flight operand DMAs with one vmcnt(0) (DRAIN) -- the only point where nothing but they is
//     outstanding; its stores are write-back (acknowledged by the L2), never write-through;
//   * the first K-tile of the next tile therefore needs no counted wait in its first two phases (its operands landed before the stores were issued), and the
//     first counted wait (phase 4) comes ~800 cycles after the last store: it finds them acknowledged.
// The K loop between the first and the last K-tile is gemm_i8_p16's, instruction for instruction.
//
// Launch conditions (launch_gemm_impl): 2-byte outputs, M % 256 == 0, N % 256 == 0 (no edge tiles: the DMA lane offsets are tile-invariant), K % 256 == 0 (an even
// number of K-tiles: the stage parity is the same for every tile), 16-byte aligned output rows, more tiles than CUs.  Everything else stays on gemm_i8_p16.
// Results are bit-identical (the same integer sums, the same epilogue arithmetic).
#pragma once

namespace asq {

constexpr int P16P_EAREA = 8192;                                  // one epilogue-operand area: [row pairs 2K | column pairs 2K | s_row 1K | s_col 1K | bias 1K | unused 1K]
constexpr int P16P_LDS_BYTES = P8_LDS_BYTES + 2 * P16P_EAREA;     // 144 KiB

// LDS-DMA of one dword per lane: 64 lanes x 4 B = 256 contiguous bytes per wave-instruction (global = SGPR base + lane offset, LDS = M0 + lane * 4)
__device__ __forceinline__ void p16p_dma4(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// the tile's epilogue operands, read from the LDS area the DMAs above filled (row(), cols()); conversions and packing are the wrapped functor's
template <class Epi> struct EpiTileLds {
    using Mma = typename Epi::Mma;
    static constexpr bool kHasRow = Epi::kHasRow, kHasCol = Epi::kHasCol, kHasBias = Epi::kHasBias;
    static constexpr int kOutBytes = Epi::kOutBytes;
    const Epi &e;
    unsigned ea;        // LDS byte address of the operand area
    int64_t m0, n0;     // the tile's origin
    void *out;
    int64_t N;
    typedef __attribute__((address_space(3))) float *lds_f32;
    typedef __attribute__((address_space(3))) v4f *lds_v4f;
    __device__ __forceinline__ float row(int64_t m) const { return kHasRow ? *(lds_f32)(uintptr_t)(ea + 4096 + (unsigned)(m - m0) * 4) : 1.0f; }
    __device__ __forceinline__ void cols(int64_t n, int64_t, v4f &sc, v4f &b) const
    {
        sc = (v4f){e.s_scalar, e.s_scalar, e.s_scalar, e.s_scalar};
        b = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (kHasCol) sc = *(lds_v4f)(uintptr_t)(ea + 5120 + (unsigned)(n - n0) * 4);
        if constexpr (kHasBias) b = *(lds_v4f)(uintptr_t)(ea + 6144 + (unsigned)(n - n0) * 4);
    }
    template <class A> __device__ __forceinline__ auto pack(const A &a, float sr, const v4f &sc, const v4f &b) const { return e.pack(a, sr, sc, b); }
};

template <class Epi>
__global__ void __launch_bounds__(512, 2) gemm_i8_p16p(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K,
                                                       int tiles_m, int tiles_n, Epi epi_in, OffsetArgs off)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ABL = 0;   // (the P8_* macros' ablation parameter: none here)
    // (Measured and dropped, profiles/r5_p16p_skew_ab.txt: starting the blocks of an XCD up to 7 x 0.25 ... 1 us apart, so that the 256 epilogue bursts of a round do
    // not coincide -- 16384 x 4096 x 4096 +1.6 % / -0.1 % / +1.6 % for 8 / 16 / 32 sleep units per step, 65536 x 4096 x 4096 within 0.4 %: nothing.)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    static_assert(Epi::Mma::kIsInt && Epi::kOutBytes == 2, "int8 operands, 2-byte outputs");
    constexpr int GM = 4;
    const int T = tiles_m * tiles_n;
    const Epi epi = epi_in.rebased(0, 0, M, N);
    const bool offs = off.row != nullptr;   // (block-uniform)
    const int nt = (int)(K / 128);          // even, >= 2
    const int klast = (nt - 1) * 128;

    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned dma_dst = lds0 + wave * 2048;
    const unsigned ebase = lds0 + P8_LDS_BYTES;
    // Per-lane address registers of the K loop: tile-invariant (interior tiles only), but (re)computed at the top of every tile from an opaque copy of the lane id,
    // so that they are DEAD during the epilogue (kept live across it they cost the epilogue ~20 VGPRs: spills in the per-token / bias variants).
    unsigned voff[4][2];  // [kind][i]: X-even, W-even, W-odd, X-odd -- this wave fills row-groups 2 * wave, 2 * wave + 1 (8 rows each) of every unit
    unsigned xb[2][2], wbp[2][2], lane4;
    auto lane_registers = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ru = (wave * 2 + i) * 8 + (ln >> 3);
            const unsigned cb = (unsigned)(((ln & 7) ^ ((ru >> 1) & 7)) * 16);
            const int64_t rxe = (ru >> 6) * 128 + (ru & 63), rwe = (ru >> 5) * 64 + (ru & 31);
            voff[0][i] = (unsigned)(rxe * K) + cb;
            voff[1][i] = (unsigned)(rwe * K) + cb;
            voff[2][i] = (unsigned)((rwe + 32) * K) + cb;
            voff[3][i] = (unsigned)((rxe + 64) * K) + cb;
        }
        const int t16 = ln & 15, q16 = ln >> 4, sw = (t16 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned o = lds0 + t16 * 128 + ((((kk * 4 + q16) ^ sw)) << 4) + s * P8_STAGE;
                xb[s][kk] = o + wm * 64 * 128;
                wbp[s][kk] = o + wn * 32 * 128;
                asm volatile("" : "+v"(xb[s][kk]), "+v"(wbp[s][kk]));
            }
        lane4 = (unsigned)ln * 4;
    };
    lane_registers();

    // tile L of this launch -> its origin (the XCD-aware map of gemm_i8_p16: consecutive ids share an XCD, 4 x tiles_n groups)
    auto tile_of = [&](int L, int64_t &m0, int64_t &n0) {
        const int id = xcd_remap(L, T);
        const int per_group = GM * tiles_n;
        const int group = id / per_group, in_group = id - group * per_group;
        const int first_m = group * GM;
        const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        m0 = (int64_t)(first_m + in_group % gm) * 256;
        n0 = (int64_t)(in_group / gm) * 256;
    };
    auto issue = [&](int kind, int stage, const int8_t *xs, const int8_t *ws, int k0) {
        const int8_t *b = ((kind == 0 || kind == 3) ? xs : ws) + k0;  // SALU
#pragma unroll
        for (int i = 0; i < 2; ++i) p8_dma16(b, voff[kind][i], dma_dst + stage * P8_STAGE + kind * P8_UNIT + i * 1024);
    };
    // the four operand DMAs of one tile (per wave; all eight waves: 2 KiB per round) into area `ea`; absent operands read a valid dummy (the weight)
    auto issue_operands = [&](int64_t m0, int64_t n0, unsigned ea) {
        const char *dummy = (const char *)w;
        const char *r0 = offs ? (const char *)(off.row + 2 * m0) : dummy;
        const char *r1 = offs ? (const char *)(off.col + 2 * n0) : dummy;
        const char *r2a = dummy, *r2b = dummy, *r3a = dummy;
        if constexpr (Epi::kHasRow) r2a = (const char *)(epi.s_row + m0);
        if constexpr (Epi::kHasCol) r2b = (const char *)(epi.s_col + n0);
        if constexpr (Epi::kHasBias) r3a = (const char *)(epi.bias + n0);
        const int wv = wave & 3;
        const char *r2 = wave < 4 ? r2a : r2b, *r3 = wave < 4 ? r3a : dummy;
        p16p_dma4(uniform_ptr((const int8_t *)r0 + wave * 256), lane4, ea + wave * 256);
        p16p_dma4(uniform_ptr((const int8_t *)r1 + wave * 256), lane4, ea + 2048 + wave * 256);
        p16p_dma4(uniform_ptr((const int8_t *)r2 + wv * 256), lane4, ea + 4096 + wave * 256);
        p16p_dma4(uniform_ptr((const int8_t *)r3 + wv * 256), lane4, ea + 6144 + wave * 256);
    };

    v4i acc[2][2][4][2];  // [m-half][n-half][token tile][channel tile]
    v4i xf[4][2], wa[2][2], wb[2][2];  // [tile][k-step]
    auto ld = [&](unsigned a) { return *(p8_lds_v4i)(uintptr_t)a; };
    auto quadrant = [&](v4i (&A)[4][2], const v4i (&wf)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it) A[jt][it] = MmaI8x16::mma(wf[it][kk], xf[jt][kk], A[jt][it]);
    };

    // ---- this block's first tile: operands + K-tile 0 entirely; everything has landed and is visible before the tile loop starts
    int L = blockIdx.x;
    int64_t m0, n0;
    tile_of(L, m0, n0);
    const int8_t *xbase = uniform_ptr(x + m0 * K), *wbase = uniform_ptr(w + n0 * K);
    issue_operands(m0, n0, ebase);
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) issue(kind, 0, xbase, wbase, 0);
    P8_WAIT_VM(0);
    P8_BAR();

    // one K-tile at LDS stage S, prefetching into stage S ^ 1: p8's four phases.  FIRST: the tile's first K-tile (its operands landed before the tile loop /
    // before the previous epilogue's stores: no counted wait until phase 4).  LAST: the tile's last K-tile (prefetches (xs, ws) = the next tile's K-tile 0 -- or a
    // dead re-read -- and first issues the next tile's epilogue operands: phase 1 waits with four more DMAs in flight).
    auto ktile = [&](auto stage_tag, auto first_tag, auto last_tag, const int8_t *xs, const int8_t *ws, int kn, int64_t m0n, int64_t n0n, unsigned ean) {
        constexpr int S = decltype(stage_tag)::value, NS = S ^ 1;
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        // ---------------- P1: (m-half 0, n-half 0): reads W-even (4) + X-even (8)
        if constexpr (LAST) issue_operands(m0n, n0n, ean);
        issue(0, NS, xs, ws, kn);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wa[it][kk] = ld(wbp[S][kk] + 1 * P8_UNIT + it * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) xf[jt][kk] = ld(xb[S][kk] + 0 * P8_UNIT + jt * 2048);
        if constexpr (LAST) P8_WAIT_VM(8);
        else if constexpr (!FIRST) P8_WAIT_VM(4);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][0], wa);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
        // ---------------- P2: (m-half 0, n-half 1): reads W-odd (4)
        issue(1, NS, xs, ws, kn);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int it = 0; it < 2; ++it) wb[it][kk] = ld(wbp[S][kk] + 2 * P8_UNIT + it * 2048);
        if constexpr (!FIRST) P8_WAIT_VM(4);
        P8_BAR();
        P8_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        P8_PRIO(1);
        quadrant(acc[0][1], wb);
        P8_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        P8_BAR();
        // ---------------- P3: (m-half 1, n-half 1): reads X-odd (8)
        issue(2, NS, xs, ws, kn);
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
        issue(3, NS, xs, ws, kn);
        P8_WAIT_VM(4);
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

    typedef __attribute__((address_space(3))) v2i *lds_v2i;
    typedef __attribute__((address_space(3))) v4i *lds_v4i_;
    int it = 0;
    for (;; ++it) {
        const unsigned ea = ebase + (it & 1) * P16P_EAREA, ean = ebase + ((it + 1) & 1) * P16P_EAREA;
        const int Ln = L + (int)gridDim.x;
        const bool has_next = Ln < T;
        int64_t m0n = m0, n0n = n0;
        if (has_next) tile_of(Ln, m0n, n0n);
        const int8_t *xnext = uniform_ptr(x + m0n * K), *wnext = uniform_ptr(w + n0n * K);
        // what the last K-tile prefetches: the next tile's K-tile 0, or (no next tile) this tile's last K-tile again (dead, keeps every count uniform)
        const int8_t *xpre = has_next ? xnext : xbase + klast, *wpre = has_next ? wnext : wbase + klast;

#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (v4i){0, 0, 0, 0};
        lane_registers();   // (unconditionally: a value carried around the back edge would stay live through the epilogue)
        if (wm == 1) P8_BAR();  // stagger: the wm = 1 group runs one barrier behind

        if (nt == 2) {
            ktile(integral_constant<int, 0>{}, true_type{}, false_type{}, xbase, wbase, 128, m0n, n0n, ean);
        } else {
            ktile(integral_constant<int, 0>{}, true_type{}, false_type{}, xbase, wbase, 128, m0n, n0n, ean);
            for (int t = 1; t + 2 < nt; t += 2) {
                ktile(integral_constant<int, 1>{}, false_type{}, false_type{}, xbase, wbase, (t + 1) * 128, m0n, n0n, ean);
                ktile(integral_constant<int, 0>{}, false_type{}, false_type{}, xbase, wbase, (t + 2) * 128, m0n, n0n, ean);
            }
        }
        ktile(integral_constant<int, 1>{}, false_type{}, true_type{}, xpre, wpre, 0, m0n, n0n, ean);

        if (wm == 0) P8_BAR();  // balance the stagger barrier: every wave has finished its last phase; stage 1 is quiescent (the DMAs in flight target stage 0)

        // ---- epilogue: staging in stage 1 only (8 KiB per wave), operands from the LDS area
        const int64_t mw0 = m0 + wm * 128, nw0 = n0 + wn * 64;
        const EpiTileLds<Epi> el{epi, ea, m0, n0, epi.out, epi.N};
        const unsigned stage = lds0 + P8_STAGE + wave * 8192;
        int le = lane;   // an opaque copy per tile: every lane-derived address of the epilogue is tile-invariant, and hoisted out of the tile loop it would live through the K loop
        asm volatile("" : "+v"(le));
        auto emit = [&](auto getter) {
            if constexpr (IsGateUp<Epi>::value) epilogue_gate_up(el, getter, mw0, nw0, le);   // gate || up GEMM: SiLU(gate) * up, half the output bytes (asq_gemm_gateup.h)
            else epilogue_wave_rows<4, 2, true, 2, true>(el, getter, mw0, nw0, le, stage, false);
        };
        if (offs) {
            const int t16i = le & 15, q16i = le >> 4;
            int cp[4][4], rw[8];   // column pairs packed cw << 24 | wsum & 0xFFFFFF (0 <= cw <= 64, |wsum| <= 2^23: v_mad_i32_i24 takes the low 24 bits of a factor, sign-extended)
#pragma unroll
            for (int in16 = 0; in16 < 4; ++in16) {
                const unsigned ca = ea + 2048 + (wn * 64 + in16 * 16 + 4 * q16i) * 8;
                const v4i p0 = *(lds_v4i_)(uintptr_t)ca, p1 = *(lds_v4i_)(uintptr_t)(ca + 16);   // channels n, n+1 | n+2, n+3: {cw, wsum} pairs
                cp[in16][0] = (int)(((unsigned)p0[0] << 24) | ((unsigned)p0[1] & 0xFFFFFFu));
                cp[in16][1] = (int)(((unsigned)p0[2] << 24) | ((unsigned)p0[3] & 0xFFFFFFu));
                cp[in16][2] = (int)(((unsigned)p1[0] << 24) | ((unsigned)p1[1] & 0xFFFFFFu));
                cp[in16][3] = (int)(((unsigned)p1[2] << 24) | ((unsigned)p1[3] & 0xFFFFFFu));
            }
            constexpr bool kRowsInRegs = !(Epi::kHasRow && Epi::kHasCol && Epi::kHasBias);   // (the per-token + column-scale + bias epilogue has no 8 registers to spare: it re-reads the pair per tile)
            auto row_word = [&](int im16) {
                const v2i p = *(lds_v2i)(uintptr_t)(ea + (wm * 128 + im16 * 16 + t16i) * 8);     // {cx, xsum'} of this lane's token
                return (int)(((unsigned)(-p[0]) << 24) | ((unsigned)(-p[1]) & 0xFFFFFFu));       // |cx| <= 64, |xsum'| < 2^23: the packed word of gemm_i8_p16
            };
            if constexpr (kRowsInRegs) {
#pragma unroll
                for (int im16 = 0; im16 < 8; ++im16) rw[im16] = row_word(im16);
            }
            auto getc = [&](int in16, int im16) -> v4i {
                const v4i &a = acc[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1];
                int r;
                if constexpr (kRowsInRegs) r = rw[im16];
                else r = row_word(im16);
                const int ncx = r >> 24;
                v4i o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = mad24(cp[in16][e] >> 24, r, mad24(ncx, cp[in16][e], a[e]));
                return o;
            };
            emit(getc);
        } else {
            auto get = [&](int in16, int im16) -> const v4i & { return acc[im16 >> 2][in16 >> 1][im16 & 3][in16 & 1]; };
            emit(get);
        }
        if (!has_next) break;
        P8_BAR();   // every wave's staging reads are done: the next tile's K-tile 1 may be prefetched into stage 1
        L = Ln;
        m0 = m0n;
        n0 = n0n;
        xbase = xnext;
        wbase = wnext;
    }
}

}  // namespace asq
