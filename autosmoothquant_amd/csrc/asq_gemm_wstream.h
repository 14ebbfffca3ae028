// "wstream": the weight-streaming kernel for few rows (decode / cfg1 / cfg4's per-GPU shape), second generation.
// The problem is ONE pass over W [N,K] int8 (4096x4096 at 32 rows: 16.8 MB of W against 0.13 MB of X and 0.26 MB of output): the
// roofline is HBM bandwidth, and the kernel's job is to keep >= 64 KB of weight lines in flight per CU while moving as few other
// bytes as possible through the L2->LDS path (which tops out at ~30-40 B/clk per CU, 13-15 TB/s chip-wide).
//
// What changed against gemm_i8_skinny (asq_gemm_skinny.h): there a work item was 16 (or 32) channels x all of K and every item
// re-read its X rows from L2 -- at 32 rows 2x (1x) the weight bytes again through the same DMA path.  Here
//
//   * a block owns a GROUP of 128 channels: its 8 waves split the CHANNELS (16 each) and share ONE X stage per K unit, so X goes
//     through L2->LDS once per 128 channels: rows/128 of the weight bytes (25 % at 32 rows, was 100-200 %);
//   * the work is the linear sequence of (group, 128-byte K unit) pairs, cut into gridDim.x equal contiguous ranges (stream-K):
//     every CU streams the same number of weight bytes whatever N and K are (11008 x 4096 = 86 groups x 32 units = 10.75 units per
//     block on 256 CUs), and a range that ends inside a group leaves an exact int32 (fp8: fp32) partial tile;
//   * partial tiles are combined INSIDE the launch: write-through (sc1) slab stores, one relaxed agent-scope ticket per group, the
//     last arriver sums the slabs (sc1 loads; fixed contributor order, so fp8 sums are reproducible) and runs the fused epilogue.
//     No dispatch-order or placement assumption, no spinning: a block never waits for another block.
//     Mid-stream segment ends only STORE (a compiler-visible load inside the loop would make hipcc drain the DMA queue); every
//     ticket, slab read and epilogue happens after the block's last unit.
//
// LDS = ring of R stages, stage = [8 waves x 16 W rows x 128 B | MT x 16 X rows x 128 B], filled by LDS-DMA in full 128-B lines
// (8 rows per wave-instruction, XOR-swizzled on the global source address; fragment reads conflict-free, as in gemm_i8_skinny).
// R - 1 stages stay in flight: 96-112 KB of weight lines per CU.  One s_barrier per unit: a unit is 16 KB of W per CU = ~1800
// cycles at the HBM rate, the barrier costs nothing next to it and makes the shared X stage (and slot reuse) trivially safe.
//
// Workspace (caller-owned, asq_gemm_workspace_bytes): [ header 8 KB: magic + one ticket per group | partial slabs ].  The header
// must have been initialised ONCE by asq_workspace_init(); tickets are left at zero by every launch (also under hipGraph replay,
// which is why there is no per-launch nonce: kernel arguments are frozen in a captured graph).  A launch on a workspace that was
// never initialised traps instead of producing wrong sums.
//
// Requirements: K % 128 == 0, K <= 2^24, x / w 16-B aligned, M <= 16 * MT <= 128.  Ragged N, M: rows are clamped for loading and
// masked at the store.
#pragma once
#include <type_traits>

namespace asq {

// (WS_HEADER_BYTES, WS_MAGIC, WS_MAX_GROUPS: asq_gemm_kernels.h, shared with the grouped launch of asq_gemm_p8.h)
constexpr int WS_CB = 128;  // channels per group = 8 waves x 16

template <int MT> struct WsCfg {
    static constexpr int XI = 2 * MT;                      // X DMA instructions per unit (8 rows each)
    static constexpr int NXW = (XI + 7) / 8;               // ... per wave (at most)
    static constexpr int STAGE = 16384 + MT * 2048;
    static constexpr int R = (160 * 1024 / STAGE) > 8 ? 8 : (160 * 1024 / STAGE);
    static constexpr int LDS = R * STAGE;
    static constexpr int SLAB = MT * 8192;                 // [8 waves][MT][64 lanes] x 16 B
};

// 5 wait states between a VALU write of an SGPR (v_readfirstlane of the base) and the VMEM instruction that reads it: the two
// s_mov and the s_nop 2 provide them inside the statement (hipcc pads nothing inside an asm string).
template <bool NT> __device__ __forceinline__ void ws_dma16(const int8_t *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the immediate must be a constant: a switch over the values that occur)
__device__ __forceinline__ void ws_wait_vm(int n)
{
#define ASQ_WS_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        ASQ_WS_VM(0) ASQ_WS_VM(1) ASQ_WS_VM(2) ASQ_WS_VM(3) ASQ_WS_VM(4) ASQ_WS_VM(5) ASQ_WS_VM(6) ASQ_WS_VM(7) ASQ_WS_VM(8) ASQ_WS_VM(9)
        ASQ_WS_VM(10) ASQ_WS_VM(11) ASQ_WS_VM(12) ASQ_WS_VM(13) ASQ_WS_VM(14) ASQ_WS_VM(15) ASQ_WS_VM(16) ASQ_WS_VM(17) ASQ_WS_VM(18)
        ASQ_WS_VM(19) ASQ_WS_VM(20) ASQ_WS_VM(21) ASQ_WS_VM(22) ASQ_WS_VM(23) ASQ_WS_VM(24) ASQ_WS_VM(25) ASQ_WS_VM(26) ASQ_WS_VM(27)
        ASQ_WS_VM(28)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef ASQ_WS_VM
}

// the block whose range [T*b/G, T*(b+1)/G) contains unit t
// (32-bit: the launcher requires T * G < 2^31)
__host__ __device__ __forceinline__ int ws_block_of(unsigned t, int T, int G) { return (int)(((t + 1u) * (unsigned)G - 1u) / (unsigned)T); }

#ifdef ASQ_WS_STAMPS  // tools/ubench/wstream_tl only: per-block wall-clock stamps (s_memrealtime, 100 MHz) of the kernel's phases
__device__ unsigned long long ws_stamps[1024][8];
#define WS_STAMP(i) do { if (threadIdx.x == 0) ws_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

template <class Epi, int MT, bool WNT>
__global__ void __launch_bounds__(512) gemm_i8_wstream(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int64_t M, int64_t N, int64_t K, int KU,
                                                       int T, int maxseg, char *ws, Epi epi)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using C = WsCfg<MT>;
    constexpr int R = C::R, STAGE = C::STAGE, XI = C::XI, NXW = C::NXW;
    using MMA = typename Epi::Mma;
    using acc4_t = typename MMA::acc4_t;
    typedef const __attribute__((address_space(3))) v4i *lds_v4i;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int t0 = (int)((unsigned)T * (unsigned)b / (unsigned)G), t1 = (int)((unsigned)T * (unsigned)(b + 1) / (unsigned)G);  // (T * G < 2^31)
    if (t0 >= t1) return;  // (the launcher keeps G <= T; block-uniform anyway)
    WS_STAMP(0);
    const unsigned long long magic = *(const volatile unsigned long long *)ws;  // (checked after the stream: the load is off the critical path)
    const int nun = t1 - t0;
    const int g_first = t0 / KU;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;

    // ---- DMA lane mapping: one instruction = 8 rows x 128 B; lane = 8*row + physical 16-B chunk, logical chunk swizzled by (row>>1)&7
    const int rr = lane >> 3, cp = lane & 7;
    int nx = 0;
    unsigned xoff[NXW];
#pragma unroll
    for (int q = 0; q < NXW; ++q) {
        const int j = wave + 8 * q;  // X instruction j covers rows 8j .. 8j+7 (tile j>>1, half j&1)
        if (j < XI) ++nx;
        int64_t m = 8 * j + rr;
        m = m < M ? m : M - 1;
        xoff[q] = (unsigned)(m * K) + (unsigned)((cp ^ (((8 * (j & 1) + rr) >> 1) & 7)) << 4);
    }
    const int d = 2 + nx;  // this wave's DMA instructions per unit
    unsigned woff[2];
    int wg = -1;
    // ---- fragment read offsets: lane (fr, fg) reads row fr, logical chunk 4h + fg
    const int fr = lane & 15, fg = lane >> 4;
    unsigned foff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) foff[h] = (unsigned)(fr * 128 + (((4 * h + fg) ^ ((fr >> 1) & 7)) << 4));

    // ---- issue cursor
    int ig = g_first, iu = t0 - g_first * KU, ni = 0;
    auto issue = [&](int slot) __attribute__((always_inline)) {
        if (ig != wg) {  // new group: this wave's 16 channels (clamped at the ragged edge; offsets relative to the group's first row)
            wg = ig;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int64_t r = 16 * wave + 8 * i + rr;
                if ((int64_t)ig * WS_CB + r >= N) r = N - 1 - (int64_t)ig * WS_CB;
                woff[i] = (unsigned)(r * K) + (unsigned)((cp ^ (((8 * i + rr) >> 1) & 7)) << 4);
            }
        }
        const int8_t *wb = uniform_ptr(w + (int64_t)ig * WS_CB * K + (int64_t)iu * 128);
        const int8_t *xb = uniform_ptr(x + (int64_t)iu * 128);
        const unsigned dst = lds0 + slot * STAGE;
        ws_dma16<WNT>(wb, woff[0], dst + wave * 2048);
        ws_dma16<WNT>(wb, woff[1], dst + wave * 2048 + 1024);
#pragma unroll
        for (int q = 0; q < NXW; ++q)
            if (wave + 8 * q < XI) ws_dma16<false>(xb, xoff[q], dst + 16384 + (wave + 8 * q) * 1024);
        if (++iu == KU) {
            iu = 0;
            ++ig;
        }
        ++ni;
    };

    // ---- partial slabs: [block][segment ordinal][wave][mt][lane] x 16 B, write-through stores / sc1 loads through a buffer descriptor
    char *const slabs = ws + WS_HEADER_BYTES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, 0x7FFFFFFF, 0x00020000);
    const int lane_off = (wave * MT * 64 + lane) * 16;
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    acc4_t acc[MT];
    // (the slab offset travels in the VGPR offset, soffset = 0: with a REGISTER soffset hipcc pads no wait state between a 128-bit buffer
    //  store and a VALU write of its data registers -- LLVM's hazard table says none is needed -- and on gfx950 the zeroing v_mov that
    //  follows the stash then reaches the store's last data register first: component 3 of lanes 12..15 of every 16 stored as 0.
    //  Found by the exact-integer comparison with the first-generation kernel.)
    auto stash = [&](int ord) __attribute__((always_inline)) {
        const int off = (b * maxseg + ord) * C::SLAB + lane_off;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, acc[mt]), rsrc, off + mt * 1024, 0, 16 /* sc1 */);
        asm volatile("s_nop 1" ::: "memory");
    };

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (acc4_t){0, 0, 0, 0};

    // ---- prologue: R - 1 units in flight
    int islot = 0;
    {
        const int pre = nun < R - 1 ? nun : R - 1;
        for (int p = 0; p < pre; ++p) {
            issue(islot);
            islot = islot + 1 == R ? 0 : islot + 1;
        }
    }

    int cg = g_first, cu = t0 - g_first * KU, slot = 0;
    for (int c = 0; c < nun; ++c) {
        ws_wait_vm((ni - c - 1) * d);     // this wave's part of unit c has landed; younger units stay in flight
        if (c == 0) WS_STAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's fragment reads of unit c - 1 have returned: its slot is refilled below)
        __builtin_amdgcn_s_barrier();     // ... every wave's part has; and every wave is done with unit c - 1
        if (ni < nun) {
            issue(islot);                 // into the slot of unit c - 1
            islot = islot + 1 == R ? 0 : islot + 1;
        }
        const unsigned base = lds0 + slot * STAGE;
        v4i wf[2], xf[MT][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            wf[h] = *(lds_v4i)(uintptr_t)(base + wave * 2048 + foff[h]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt][h] = *(lds_v4i)(uintptr_t)(base + 16384 + mt * 2048 + foff[h]);
        }
        if constexpr (MMA::kIsInt) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[h], xf[mt][h], acc[mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = MMA::mma16(wf[0], wf[1], xf[mt][0], xf[mt][1], acc[mt]);
        }
        slot = slot + 1 == R ? 0 : slot + 1;
        if (c + 1 < nun) {
            if (cu == KU - 1) {  // a segment ends mid-stream: stores only (see the file comment)
                stash(cg - g_first);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = (acc4_t){0, 0, 0, 0};
                ++cg;
                cu = 0;
            } else {
                ++cu;
            }
        }
    }
    WS_STAMP(2);
    // the block's last segment (group cg) is in registers; it ends at unit cu of its group
    const int nseg = cg - g_first + 1;
    const unsigned g_last_lo = (unsigned)cg * (unsigned)KU;
    const bool last_shared = ws_block_of(g_last_lo, T, G) != ws_block_of(g_last_lo + KU - 1, T, G);
    if (last_shared) stash(nseg - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __builtin_amdgcn_s_barrier();                      // ... before the block takes its tickets
    WS_STAMP(3);

    // ---- tickets: lane `ord` of wave 0 arrives for segment `ord`; bit `ord` of the mask = "this block finishes that group"
    unsigned *const hdr = (unsigned *)ws;
    unsigned long long *const mask_lds = (unsigned long long *)lds;  // (ring is dead: every DMA has landed and been read)
    if (wave == 0) {
        bool fin = false;
        for (int o0 = 0; o0 < nseg; o0 += 64) {  // (more than 64 segments per block only for absurd N / K ratios; handled anyway)
            const int ord = o0 + lane;
            fin = false;
            if (ord < nseg) {
                const unsigned lo = (unsigned)(g_first + ord) * (unsigned)KU;
                const int bf = ws_block_of(lo, T, G), bl = ws_block_of(lo + KU - 1, T, G);
                if (bf == bl) {
                    fin = true;  // whole group inside this block's range
                } else {
                    const unsigned S = (unsigned)(bl - bf + 1);
                    const unsigned old = __hip_atomic_fetch_add(hdr + 4 + g_first + ord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old >= S) __builtin_trap();  // poisoned tickets: the workspace header was not initialised / is shared by concurrent launches
                    if (old == S - 1) {
                        fin = true;
                        __hip_atomic_store(hdr + 4 + g_first + ord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // left at zero for the next launch / replay
                    }
                }
            }
            const unsigned long long m = __ballot(fin);
            if (lane == 0) mask_lds[o0 >> 6] = m;
        }
        if (magic != WS_MAGIC) __builtin_trap();  // workspace never went through asq_workspace_init
    }
    __syncthreads();
    WS_STAMP(4);

    // ---- finish the groups this block arrived last at: sum the contributors' slabs in block order, fused epilogue
    for (int ord = 0; ord < nseg; ++ord) {
        if (!((mask_lds[ord >> 6] >> (ord & 63)) & 1ull)) continue;  // block-uniform
        const int g = g_first + ord;
        const unsigned lo = (unsigned)g * (unsigned)KU;
        const int bf = ws_block_of(lo, T, G), bl = ws_block_of(lo + KU - 1, T, G);
        // the epilogue's own operands first (round 5): per-token scales, channel scales and bias used to be fetched behind the slab sums -- a second dependent
        // round trip (~1 us of the ~3 us between "ticket known" and "group done", profiles/r3_wstream_timelines.txt); issued here they travel with the slab loads
        float srv[MT];
        v4f scv = {0.f, 0.f, 0.f, 0.f}, bbv = {0.f, 0.f, 0.f, 0.f};
        const int64_t ncol = (int64_t)g * WS_CB + 16 * wave + 4 * fg;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = mt * 16 + fr;
            srv[mt] = Epi::kHasRow ? epi.row(m < M ? m : M - 1) : 1.0f;
        }
        if (ncol < N) epi.cols(ncol, N, scv, bbv);
        acc4_t sum[MT];
        if (bf == bl && ord == nseg - 1) {  // the whole group ran in this block and ended with it: still in registers
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) sum[mt] = acc[mt];
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) sum[mt] = (acc4_t){0, 0, 0, 0};
            constexpr int FB = MT <= 2 ? 8 : MT == 4 ? 4 : 2;  // contributors whose loads are in flight together (64 VGPRs)
            for (int cb = bf; cb <= bl; cb += FB) {  // a clamped index + mask instead of a conditional load
                v4u_ v[FB][MT];
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    const int cc = cb + i <= bl ? cb + i : bl;
                    const int cfirst = (int)(((unsigned)T * (unsigned)cc / (unsigned)G) / (unsigned)KU);  // its first group -> its segment ordinal for g
                    const int off = (cc * maxseg + (g - cfirst)) * C::SLAB + lane_off;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) v[i][mt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + mt * 1024, 0, 16 /* sc1 */);
                }
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    const unsigned msk = cb + i <= bl ? 0xFFFFFFFFu : 0u;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) sum[mt] += __builtin_bit_cast(acc4_t, v[i][mt] & msk);  // (fp8: +0.0 for the masked slots)
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = mt * 16 + fr;
            if (m < M && ncol < N) epi.store4(m, ncol, sum[mt], srv[mt], scv, bbv, N);
        }
        WS_STAMP(5);
    }
    WS_STAMP(6);
}

// one-time initialisation of a workspace header (magic + zero tickets); a launch, so it is ordered on `stream` like everything else
static __global__ void __launch_bounds__(256) ws_init_header(unsigned *hdr)
{
    for (int i = threadIdx.x; i < WS_HEADER_BYTES / 4; i += 256) hdr[i] = i == 0 ? (unsigned)(WS_MAGIC & 0xFFFFFFFFu) : i == 1 ? (unsigned)(WS_MAGIC >> 32) : 0u;
}

}  // namespace asq
