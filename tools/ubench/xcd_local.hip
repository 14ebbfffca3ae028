// What does a block on the SAME XCD see of another block's freshly stored partial tile, and how fast?  (round 5: the in-launch split-K / stream-K tails pay
// ~3 us of their ~4.3 us for the last arriver's sc1 reads of write-through slabs; round 4's XCD-local form found that a write-through store leaves no line in
// the writer's L2.)  Here a writer block stores a 64 KiB image under one of four policies, publishes a nonce with an agent-scope atomic, and reader blocks --
// one on the writer's XCD, one on another -- wait for the nonce (bounded spin: a probe, not product code), then time their loads of the image under two load
// policies and count wrong words.  XCC_ID comes from the hardware register; block b is expected on XCD b % 8.
//   store policy: 0 plain   1 write-through (sc0 sc1)   2 write-through, then plain   3 plain, then write-through
//   load policy : 0 plain   1 sc1
// Launches repeat with a fresh nonce so that stale lines of earlier launches are in the L2s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
template <int POL> __device__ __forceinline__ void st16(v4u *p, v4u v)
{
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ v4u ld16(const v4u *p)
{
    v4u v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Rec { unsigned xcc, us100, bad, spins; };   // us100: s_memrealtime ticks (100 MHz)

// image: 64 KiB = 4096 x 16 B; 512 threads x 8 pieces
template <int SP, int LP>
__global__ void __launch_bounds__(512) k(v4u *img, unsigned *flag, Rec *rec, unsigned nonce, int reader_same, int reader_other)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned bad_s;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    if (b == 0) {   // writer
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 512 + tid;
            const v4u v = {nonce, (unsigned)idx, nonce ^ (unsigned)idx, ~nonce};
            if constexpr (SP == 0) st16<0>(img + idx, v);
            if constexpr (SP == 1) st16<1>(img + idx, v);
            if constexpr (SP == 2) { st16<1>(img + idx, v); }
            if constexpr (SP == 3) { st16<0>(img + idx, v); }
        }
        if constexpr (SP == 2 || SP == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 512 + tid;
                const v4u v = {nonce, (unsigned)idx, nonce ^ (unsigned)idx, ~nonce};
                if constexpr (SP == 2) st16<0>(img + idx, v); else st16<1>(img + idx, v);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            __hip_atomic_store(flag, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rec[0] = Rec{xcc_id(), (unsigned)(t1 - t0), 0, 0};
        }
        return;
    }
    if (b != reader_same && b != reader_other) return;
    __shared__ unsigned spins_s;
    if (tid == 0) {
        unsigned s = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nonce && s < 2000000u) ++s;
        spins_s = s;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    v4u v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ld16<LP>(img + i * 512 + tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned idx = i * 512 + tid;
        bad += (v[i][0] != nonce) + (v[i][1] != idx) + (v[i][2] != (nonce ^ idx)) + (v[i][3] != ~nonce);
    }
    if (bad) atomicAdd(&bad_s, bad);
    __syncthreads();
    if (tid == 0) rec[b == reader_same ? 1 : 2] = Rec{xcc_id(), (unsigned)(t1 - t0), bad_s, spins_s};
}

template <int SP, int LP> void run(v4u *img, unsigned *flag, Rec *rec, unsigned &nonce, int other)
{
    const char *sp[4] = {"plain", "write-through", "write-through then plain", "plain then write-through"};
    const char *lp[2] = {"plain", "sc1"};
    std::vector<Rec> h(3);
    unsigned t_same = 0, t_other = 0, bad_same = 0, bad_other = 0, t_w = 0, n = 0, xs = 0, xo = 0, xw = 0, tmin_same = ~0u, tmin_other = ~0u;
    for (int r = 0; r < 50; ++r) {
        ++nonce;
        hipLaunchKernelGGL((k<SP, LP>), dim3(16), dim3(512), 0, 0, img, flag, rec, nonce, 8, other);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), rec, 3 * sizeof(Rec), hipMemcpyDeviceToHost));
        if (r < 5) continue;
        t_w += h[0].us100; t_same += h[1].us100; t_other += h[2].us100; bad_same += h[1].bad; bad_other += h[2].bad; ++n;
        tmin_same = h[1].us100 < tmin_same ? h[1].us100 : tmin_same; tmin_other = h[2].us100 < tmin_other ? h[2].us100 : tmin_other;
        xw = h[0].xcc; xs = h[1].xcc; xo = h[2].xcc;
    }
    printf("store %-26s load %-5s | writer xcc %u store+drain %5.2f us | same-XCD reader (xcc %u) %5.2f us (min %4.2f) wrong words %u | other-XCD reader (xcc %u) %5.2f us (min %4.2f) wrong words %u\n",
           sp[SP], lp[LP], xw, t_w * 0.01 / n, xs, t_same * 0.01 / n, tmin_same * 0.01, bad_same, xo, t_other * 0.01 / n, tmin_other * 0.01, bad_other);
}

int main()
{
    v4u *img; unsigned *flag; Rec *rec;
    CK(hipMalloc(&img, 65536)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&rec, 3 * sizeof(Rec)));
    CK(hipMemset(img, 0, 65536)); CK(hipMemset(flag, 0, 4));
    unsigned nonce = 1000;
    for (int other : {1, 4}) {
        printf("# writer = block 0, same-XCD reader = block 8, other reader = block %d; 64 KiB image, 512 threads x 8 x 16 B; 45 launches each\n", other);
        run<0, 0>(img, flag, rec, nonce, other); run<0, 1>(img, flag, rec, nonce, other);
        run<1, 0>(img, flag, rec, nonce, other); run<1, 1>(img, flag, rec, nonce, other);
        run<2, 0>(img, flag, rec, nonce, other); run<2, 1>(img, flag, rec, nonce, other);
        run<3, 0>(img, flag, rec, nonce, other); run<3, 1>(img, flag, rec, nonce, other);
    }
    return 0;
}
