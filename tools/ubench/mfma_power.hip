// Does the INT8 MFMA rate survive real (random) operand data?  Sustained full-chip runs of
// v_mfma_i32_32x32x32_i8 with (a) zero, (b) constant small, (c) random full-range operands,
// 2 waves/SIMD, ~50 ms each (long enough for DVFS to settle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE> __global__ void __launch_bounds__(512) k(int iters, int* out)
{
    v4i a[4], b[4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u; unsigned r1 = s; s = s * 1664525u + 1013904223u; unsigned r2 = s;
            a[i][j] = MODE == 0 ? 0 : (MODE == 1 ? 0x01010101 : (int)(r1 ^ (r1 >> 13)));
            b[i][j] = MODE == 0 ? 0 : (MODE == 1 ? 0x01010101 : (int)(r2 ^ (r2 >> 11)));
        }
    v16i acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v16i){0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main()
{
    int* out; CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"zeros", "const 0x01", "random full-range"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 200000;  // 1.6M MFMAs per wave
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, iters, out);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, iters, out);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double ops = 2.0 * 32 * 32 * 32 * 8.0 * iters * 8 * 256;
            double clk = (8.0 * iters * 2 * 32) / (ms * 1e-3) / 1e9;  // cycles each SIMD spent (2 waves x 32 cyc/MFMA) / time
            printf("%-20s: %.1f ms  %.0f TOPS  -> effective MFMA clock %.2f GHz\n", names[mode], ms, ops / ms / 1e9, clk);
        }
    return 0;
}
