// Can a cross-CU split-K reduction through device-scope int32 atomics be cheap enough for the decode GEMM?
// NB blocks; block b adds its 32x64 int32 partial (2048 dwords) into tile (b % NT) of a [NT][2048] buffer, so every
// word receives NB/NT adds from blocks on different XCDs; then a ticket per tile, last arriver reads back with
// atomicExch(.,0) (self-cleaning) and checks the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>  // 0 = atomics + ticket + exch readback, 1 = atomics only, 2 = plain stores to private slab (baseline)
__global__ void __launch_bounds__(1024) k(int* buf, int* tickets, int* slab, int* out_ok, int NT, int S)
{
    const int tile = blockIdx.x % NT;
    int* base = buf + tile * 2048;
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 2048; i += 1024) slab[blockIdx.x * 2048 + i] = i + 1;
        return;
    }
    for (int i = threadIdx.x; i < 2048; i += 1024) __hip_atomic_fetch_add(base + i, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) return;
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        int t = __hip_atomic_fetch_add(tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == S - 1);
        if (last) { __hip_atomic_store(tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
    __syncthreads();
    if (last) {
        int bad = 0;
        for (int i = threadIdx.x; i < 2048; i += 1024) {
            int v = __hip_atomic_exchange(base + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += (v != S * (i + 1));
        }
        if (bad) atomicAdd(out_ok, bad);
    }
}

int main()
{
    const int NB = 320, S = 4, NT = NB / S;
    int *buf, *tickets, *slab, *ok;
    CK(hipMalloc(&buf, NT * 2048 * 4)); CK(hipMalloc(&tickets, NT * 4)); CK(hipMalloc(&slab, NB * 2048 * 4)); CK(hipMalloc(&ok, 4));
    CK(hipMemset(buf, 0, NT * 2048 * 4)); CK(hipMemset(tickets, 0, NT * 4)); CK(hipMemset(ok, 0, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[4] = {"atomics + FENCES + ticket + exch", "atomics only", "plain slab stores", "atomics + ticket + exch, NO fences"};
    for (int mode : {2, 1, 0, 3, 3, 3}) {
        if (mode == 1) { }
        const int batch = 20;
        float best = 1e9;
        for (int r = 0; r < (mode == 3 ? 200 : 5); ++r) {
            if (mode == 1) CK(hipMemset(buf, 0, NT * 2048 * 4));
            CK(hipEventRecord(a));
            for (int i = 0; i < batch; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(1024), 0, 0, buf, tickets, slab, ok, NT, S);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NB), dim3(1024), 0, 0, buf, tickets, slab, ok, NT, S);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(NB), dim3(1024), 0, 0, buf, tickets, slab, ok, NT, S);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(NB), dim3(1024), 0, 0, buf, tickets, slab, ok, NT, S);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= batch; best = ms < best ? ms : best;
        }
        int hok; CK(hipMemcpy(&hok, ok, 4, hipMemcpyDeviceToHost));
        printf("%-36s: %.2f us per launch (320 blocks x 2048 dwords, 4 adds per word)  mismatches so far: %d\n", names[mode], best * 1e3, hok);
        if (mode == 1) CK(hipMemset(buf, 0, NT * 2048 * 4));
    }
    return 0;
}
