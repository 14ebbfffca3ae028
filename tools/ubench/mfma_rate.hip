// Microbenchmark: issue interval / dependent latency of gfx950 i8 MFMAs, and ds_read_b128
// throughput for the p8 fragment-read pattern.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC, int SHAPE>  // SHAPE 0: 32x32x32, 1: 16x16x64
__global__ void __launch_bounds__(512) mfma_loop(int iters, int* out, long long* cycles)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    v16i acc[NACC];
    v4i acc4[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (v16i){0}; acc4[i] = (v4i){0}; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[i], 0, 0, 0);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) s += acc[i][j]; for (int j = 0; j < 4; ++j) s += acc4[i][j]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ds_read_b128 pattern test: MODE 0 = p8 swizzled fragment reads, 1 = linear rows (conflicting)
template <int MODE>
__global__ void __launch_bounds__(512) lds_loop(int iters, int* out, long long* cycles)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((int*)lds)[i] = i;
    __syncthreads();
    const int frow = lane & 31, sw = MODE == 0 ? ((frow >> 1) & 7) : 0, hi = lane >> 5;
    int foff[4];
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + ((((ks * 2 + hi) ^ sw)) << 4);
    const char* base = lds + (wave & 3) * 4096;
    v4i s = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v4i v = *(const v4i*)(base + u * 16384 + foff[ks]);
                s += v;
            }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <class F> float time_ms(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main()
{
    int* out; long long* cyc; CK(hipMalloc(&out, 4096 * 512 * 4)); CK(hipMalloc(&cyc, 4096 * 8));
    const int iters = 20000;
    long long hc[8];
#define RUN(NACC, SHAPE, BLOCKS, THREADS, label) { \
        float ms = time_ms([&] { hipLaunchKernelGGL((mfma_loop<NACC, SHAPE>), dim3(BLOCKS), dim3(THREADS), 0, 0, iters, out, cyc); }); \
        CK(hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost)); \
        double nm = (double)iters * 8 * (THREADS / 64) * BLOCKS; \
        double ops = nm * (SHAPE == 0 ? 2.0 * 32 * 32 * 32 : 2.0 * 16 * 16 * 64); \
        printf("%-44s blocks=%4d thr=%3d nacc=%d : %.3f ms  %.0f TOPS  wave-cycles/MFMA=%.1f (memtime ticks) clk~%.2f GHz-equiv\n", label, BLOCKS, THREADS, NACC, ms, ops / ms / 1e9, \
               (double)hc[0] / (iters * 8.0), 0.0); }
    RUN(8, 0, 256, 256, "32x32x32 1 wave/SIMD, 8 indep acc");
    RUN(2, 0, 256, 256, "32x32x32 1 wave/SIMD, 2 alternating acc");
    RUN(1, 0, 256, 256, "32x32x32 1 wave/SIMD, 1 acc (dependent)");
    RUN(8, 0, 256, 512, "32x32x32 2 waves/SIMD, 8 indep acc");
    RUN(2, 0, 256, 512, "32x32x32 2 waves/SIMD, 2 alternating acc");
    RUN(2, 0, 16, 512, "32x32x32 2 waves/SIMD, 2 acc, 16 blocks only");
    RUN(8, 1, 256, 256, "16x16x64 1 wave/SIMD, 8 indep acc");
    RUN(2, 1, 256, 256, "16x16x64 1 wave/SIMD, 2 alternating acc");
    RUN(1, 1, 256, 256, "16x16x64 1 wave/SIMD, 1 acc (dependent)");
    RUN(8, 1, 256, 512, "16x16x64 2 waves/SIMD, 8 indep acc");
    RUN(4, 1, 256, 512, "16x16x64 2 waves/SIMD, 4 acc");
    {
        const int li = 2000;
        for (int mode = 0; mode < 2; ++mode) for (int thr : {256, 512}) {
            auto kfn = mode == 0 ? lds_loop<0> : lds_loop<1>;
            CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            float ms = time_ms([&] { hipLaunchKernelGGL(kfn, dim3(256), dim3(thr), 65536, 0, li, out, cyc); });
            CK(hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost));
            double bytes = (double)li * 16 * 1024 * (thr / 64) * 256;
            printf("ds_read_b128 %s thr=%d: %.3f ms, %.1f TB/s aggregate, %.1f ticks per wave-read\n", mode == 0 ? "swizzled(p8)" : "linear", thr, ms, bytes / ms / 1e9,
                   (double)hc[0] / (li * 16.0));
        }
    }
    return 0;
}
