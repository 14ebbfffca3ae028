// Does an OFFSET on the int8 operands buy matrix-core energy?  (round 4)
// profiles/r2_clock_power_evidence.md section 3: under the 1.4 kW socket limit the sustained INT8 MFMA rate is set by the operands' switching
// energy, and most of it is two's-complement SIGN EXTENSION (abs(W) x abs(X) sustains 4728 TOPS against 3860 for W x X on the bench's data).
// The int8 MFMA is signed x signed, but an offset per token row / per channel row is exact:
//     sum_k W[n,k] X[m,k] = sum_k (W[n,k] + cw[n]) (X[m,k] + cx[m])  -  cx[m] * sum_k W[n,k]  -  cw[n] * sum_k (X[m,k] + cx[m])
// with cx[m] <= 127 - max_k X[m,k] and cw[n] <= 127 - max_k W[n,k] (no clamping, so no sparse correction), both corrections rank-1 in the epilogue.
// This probe measures what the matrix cores sustain on p16's quadrant (v_mfma_i32_16x16x64_i8, activation fragment stays for two instructions)
// for offsets (cw, cx): operands static in registers, 2 waves per SIMD, 256 blocks, ~1 s per arm.
//   W: N(0, 21.7^2) (bench weights);  X: 99 % N(0, 1.41^2) + 1 % outlier k positions N(0, 28^2) (bench activations: rms ~3.1, 98.5 % in [-3, 3])
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bias_power.hip -o mfma_bias_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
__device__ __forceinline__ int gauss_int(unsigned &s, float rms)
{
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += (float)(lcg(s) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f;   // sum of 4 U(-1,1): variance 4/3
    return (int)rintf(t * rms * 0.8660254f);
}
__device__ __forceinline__ int clamp8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }

// mode 0: offsets (cw, cx) added, clamped to int8 (the product path never clamps: it picks the offset from the row maximum; here the clamp only touches the far tail)
// mode 1: abs(W) x abs(X) (the r2 reference point); mode 2: zeros
// accb: accumulators start at this value instead of 0 (a large positive start keeps the running sums away from the sign boundary)
__global__ void __launch_bounds__(512) kq(int iters, int *out, int mode, int cw, int cx, int accb, float wrms, float xrms, float xout_rms, int xout_per_1024)
{
    v4i wf[2][2], xf[4][2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 999u;
    for (int i = 0; i < 2; ++i) for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            int w = clamp8(gauss_int(s, wrms));
            if (mode == 0) w = clamp8(w + cw);
            if (mode == 1) w = w < 0 ? -w : w, w = w > 127 ? 127 : w;
            if (mode == 2) w = 0;
            v |= (unsigned)(w & 0xFF) << (8 * b);
        }
        wf[i][kk][j] = (int)v; }
    for (int i = 0; i < 4; ++i) for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            const bool outl = (int)(lcg(s) & 1023) < xout_per_1024;
            int x = clamp8(gauss_int(s, outl ? xout_rms : xrms));
            if (mode == 0) x = clamp8(x + cx);
            if (mode == 1) x = x < 0 ? -x : x, x = x > 127 ? 127 : x;
            if (mode == 2) x = 0;
            v |= (unsigned)(x & 0xFF) << (8 * b);
        }
        xf[i][kk][j] = (int)v; }
    v4i acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4i){accb, accb, accb, accb};
    for (int it_ = 0; it_ < iters; ++it_) {
        if ((it_ & 31) == 0) {   // a GEMM's accumulators start over every K / 64 steps; keep the running sums in a GEMM-like range (K = 4096: 64 steps, here 32 x 2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(acc[i][j])); acc[i][j] = (acc[i][j] & 1) + (v4i){accb, accb, accb, accb}; }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it) acc[jt][it] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[it][kk], xf[jt][kk], acc[jt][it], 0, 0, 0);
    }
    int r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// does the int32 accumulation wrap (two's complement) or saturate?  one instruction: A = B = 1 everywhere -> every output element gains 64
__global__ void wrap_test(int *out)
{
    const v4i one = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    v4i acc = {0x7FFFFFF0, (int)0x80000010, 0x7FFFFFFF, -1};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(one, one, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = acc[3]; }
}

struct Arm { const char *name; int mode, cw, cx, accb; };

static double run(const Arm &a, int *out, double seconds, float wrms, float xrms, float xo, int xop)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    double total_ms = 0, tail_sum = 0; int tail_n = 0;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(kq, dim3(256), dim3(512), 0, 0, iters, out, a.mode, a.cw, a.cx, a.accb, wrms, xrms, xo, xop);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        const double tops = 2.0 * 16 * 16384.0 * iters * 8 * 256 * 10 / (ms * 1e-3) / 1e12;
        if (total_ms > seconds * 500) { tail_sum += tops; ++tail_n; }
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return tail_sum / (tail_n ? tail_n : 1);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    int *out; CK(hipMalloc(&out, 256 * 512 * 4));
    {
        hipLaunchKernelGGL(wrap_test, dim3(1), dim3(64), 0, 0, out);
        int h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("wrap test (acc + 64): 0x7FFFFFF0 -> 0x%08X (wraps to 0x80000030: %s), 0x80000010 -> 0x%08X, 0x7FFFFFFF -> 0x%08X, -1 -> %d\n", h[0], h[0] == (int)0x80000030 ? "yes" : "NO",
               h[1], h[2], h[3]);
    }
    const Arm arms[] = {
        {"W x X (what the GEMM multiplies today)", 0, 0, 0, 0},
        {"W x X, accumulators start at 2^24", 0, 0, 0, 1 << 24},
        {"W x (X+2)", 0, 0, 2, 0}, {"W x (X+3)", 0, 0, 3, 0}, {"W x (X+4)", 0, 0, 4, 0}, {"W x (X+6)", 0, 0, 6, 0}, {"W x (X+8)", 0, 0, 8, 0}, {"W x (X+16)", 0, 0, 16, 0},
        {"W x (X+32)", 0, 0, 32, 0}, {"W x (X+64)", 0, 0, 64, 0},
        {"(W+32) x X", 0, 32, 0, 0}, {"(W+48) x X", 0, 48, 0, 0}, {"(W+64) x X", 0, 64, 0, 0},
        {"(W+32) x (X+3)", 0, 32, 3, 0}, {"(W+48) x (X+3)", 0, 48, 3, 0}, {"(W+64) x (X+3)", 0, 64, 3, 0},
        {"(W+48) x (X+4)", 0, 48, 4, 0}, {"(W+48) x (X+8)", 0, 48, 8, 0}, {"(W+64) x (X+8)", 0, 64, 8, 0}, {"(W+48) x (X+64)", 0, 48, 64, 0},
        {"(W-48) x (X-4)   (both mostly negative)", 0, -48, -4, 0},
        {"(W+48) x (X-3)   (products mostly negative)", 0, 48, -3, 0},
        {"(W-48) x (X+3)", 0, -48, 3, 0},
        {"(W+48) x X, accumulators start at 2^24", 0, 48, 0, 1 << 24},
        {"(W+48) x X, accumulators start at -2^24", 0, 48, 0, -(1 << 24)},
        // the trajectories of the exact scheme (64 steps of ~9.2k each = 590k): which end of it should sit at zero?
        {"(W+48) x (X+3), start 0        (0 -> +590k)", 0, 48, 3, 0},
        {"(W+48) x (X+3), start -590k    (-590k -> 0: the correction as start value)", 0, 48, 3, -590000},
        {"(W+48) x (X+3), start -2^24", 0, 48, 3, -(1 << 24)},
        {"(W+48) x (X-3), start 0        (0 -> -590k)", 0, 48, -3, 0},
        {"(W+48) x (X-3), start +590k    (+590k -> 0: the correction as start value)", 0, 48, -3, 590000},
        {"(W+48) x (X-3), start +2^24", 0, 48, -3, 1 << 24},
        {"abs(W) x abs(X)", 1, 0, 0, 0},
        {"zeros", 2, 0, 0, 0},
        {"W x X again (drift check)", 0, 0, 0, 0},
    };
    printf("p16 quadrant on v_mfma_i32_16x16x64_i8, bench-like operands (W rms 21.7; X rms 1.41 + 1 %% outlier k positions rms 28), %.1f s per arm\n", seconds);
    double base = 0;
    for (const Arm &a : arms) {
        const double t = run(a, out, seconds, 21.7f, 1.41f, 28.f, 10);
        if (base == 0) base = t;
        printf("  %-78s %6.0f TOPS  (%+5.1f %%)\n", a.name, t, 100.0 * (t / base - 1.0));
        fflush(stdout);
    }
    printf("uniform-rms activations (X rms 2.9, no outlier positions), for comparison with profiles/r3_mfma_shape_power.txt\n");
    const Arm arms2[] = {{"W x X", 0, 0, 0, 0}, {"W x (X+8)", 0, 0, 8, 0}, {"(W+48) x (X+8)", 0, 48, 8, 0}};
    base = 0;
    for (const Arm &a : arms2) {
        const double t = run(a, out, seconds, 21.7f, 2.9f, 2.9f, 0);
        if (base == 0) base = t;
        printf("  %-78s %6.0f TOPS  (%+5.1f %%)\n", a.name, t, 100.0 * (t / base - 1.0));
    }
    return 0;
}
