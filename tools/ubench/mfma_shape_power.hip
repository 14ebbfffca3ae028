// Which INT8 MFMA shape costs fewer joules per MAC on real operand data?  Under the 1.4 kW socket limit the sustained rate of a GEMM
// is set by energy, not by the issue rate (profiles/r2_clock_power_evidence.md), so the question is not the peak of
// v_mfma_i32_32x32x32_i8 (32 cycles, 32768 MACs, 16 accumulator registers) vs v_mfma_i32_16x16x64_i8 (16 cycles, 16384 MACs, 4 accumulator
// registers) but what each sustains for ~1 s on (a) zeros, (b) bench-like operands (weights rms ~22, activations mostly in [-3, 3]),
// (c) uniform int8.  Operands static in registers, 8 independent accumulators, 2 waves per SIMD, 256 blocks.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
__device__ __forceinline__ int gauss_byte(unsigned &s, float rms)   // ~N(0, rms^2), clamped to int8
{
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += (float)(lcg(s) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f;   // sum of 4 U(-1,1): variance 4/3
    int v = (int)rintf(t * rms * 0.8660254f);
    v = v > 127 ? 127 : (v < -128 ? -128 : v);
    return v & 0xFF;
}

template <int SHAPE, int MODE> __global__ void __launch_bounds__(512) k(int iters, int *out)
{
    v4i a[4], b[4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            unsigned wa = 0, wb = 0;
            for (int byte = 0; byte < 4; ++byte) {
                int x = 0, y = 0;
                if (MODE == 1) { x = gauss_byte(s, 21.7f); y = gauss_byte(s, 2.9f); }
                if (MODE == 2) { x = lcg(s) & 0xFF; y = lcg(s) & 0xFF; }
                wa |= (unsigned)x << (8 * byte); wb |= (unsigned)y << (8 * byte);
            }
            a[i][j] = (int)wa; b[i][j] = (int)wb;
        }
    v16i acc[8]; v4i acc4[8];
    for (int i = 0; i < 8; ++i) { acc[i] = (v16i){0}; acc4[i] = (v4i){0}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < (SHAPE == 0 ? 1 : 2); ++r)   // the same MACs per iteration for both shapes
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(i + r) & 3], b[(i >> 1) & 3], acc4[i], 0, 0, 0);
            }
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 16; ++j) r += acc[i][j]; for (int j = 0; j < 4; ++j) r += acc4[i][j]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE, int MODE> static double run(int *out, double seconds)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;   // 8 x 32768 MACs x 20000 per wave: ~2-3 ms per launch
    double best_tail = 0; int n = 0; double total_ms = 0, tail_sum = 0; int tail_n = 0;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < 10; ++l) hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(512), 0, 0, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms; ++n;
        const double tops = 2.0 * 8 * 32768.0 * iters * 8 * 256 * 10 / (ms * 1e-3) / 1e12;
        if (total_ms > seconds * 500) { tail_sum += tops; ++tail_n; }
        best_tail = tops;
    }
    return tail_n ? tail_sum / tail_n : best_tail;
}

// p16's quadrant (2 weight fragments x 4 activation fragments x 2 k-steps = 16 instructions on 8 accumulators) in different issue orders:
//   0: k-step, weight fragment, activation fragment   (the weight fragment stays for 4 instructions)  -- what gemm_i8_p16 issues
//   1: k-step, activation fragment, weight fragment   (the activation fragment stays for 2)
//   2: as 0 with the activation fragments snaking (.., x3 | x3, x2, ..)
//   3: as 0 with the roles swapped (activations = A operand, weights = B)
template <int ORDER> __global__ void __launch_bounds__(512) kq(int iters, int *out)
{
    v4i wf[2][2], xf[4][2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 999u;
    for (int i = 0; i < 2; ++i) for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) {
        unsigned v = 0; for (int b = 0; b < 4; ++b) v |= (unsigned)gauss_byte(s, 21.7f) << (8 * b); wf[i][kk][j] = (int)v; }
    for (int i = 0; i < 4; ++i) for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) {
        unsigned v = 0; for (int b = 0; b < 4; ++b) v |= (unsigned)gauss_byte(s, 2.9f) << (8 * b); xf[i][kk][j] = (int)v; }
    v4i acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4i){0};
    for (int it_ = 0; it_ < iters; ++it_) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (ORDER == 1) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int it = 0; it < 2; ++it) acc[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[it][kk], xf[jt][kk], acc[jt][it], 0, 0, 0);
            } else {
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jt = (ORDER == 2 && it == 1) ? 3 - j : j;
                        if (ORDER == 3) acc[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xf[jt][kk], wf[it][kk], acc[jt][it], 0, 0, 0);
                        else acc[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[it][kk], xf[jt][kk], acc[jt][it], 0, 0, 0);
                    }
            }
        }
    }
    int r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int ORDER> static double runq(int *out, double seconds)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    double total_ms = 0, tail_sum = 0; int tail_n = 0;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < 10; ++l) hipLaunchKernelGGL((kq<ORDER>), dim3(256), dim3(512), 0, 0, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        const double tops = 2.0 * 16 * 16384.0 * iters * 8 * 256 * 10 / (ms * 1e-3) / 1e12;
        if (total_ms > seconds * 500) { tail_sum += tops; ++tail_n; }
    }
    return tail_sum / (tail_n ? tail_n : 1);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    int *out; CK(hipMalloc(&out, 256 * 512 * 4));
    const char *modes[3] = {"zeros", "bench-like (w rms 21.7, x rms 2.9)", "uniform int8"};
    for (int rep = 0; rep < 2; ++rep) {
        double t[2][3];
        t[0][0] = run<0, 0>(out, seconds); t[1][0] = run<1, 0>(out, seconds);
        t[0][1] = run<0, 1>(out, seconds); t[1][1] = run<1, 1>(out, seconds);
        t[0][2] = run<0, 2>(out, seconds); t[1][2] = run<1, 2>(out, seconds);
        for (int m = 0; m < 3; ++m)
            printf("%-38s 32x32x32: %6.0f TOPS   16x16x64: %6.0f TOPS   ratio %.3f\n", modes[m], t[0][m], t[1][m], t[1][m] / t[0][m]);
    }
    for (int rep = 0; rep < 2; ++rep)
        printf("16x16x64, p16 quadrant, bench-like operands: order 0 (W stays 4) %6.0f | 1 (X stays 2) %6.0f | 2 (snake) %6.0f | 3 (roles swapped) %6.0f TOPS\n",
               runq<0>(out, seconds), runq<1>(out, seconds), runq<2>(out, seconds), runq<3>(out, seconds));
    return 0;
}
