// Dev probe: does the hardware fp32 -> fp8 convert (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, gfx950 = OCP formats)
// produce the same byte as the software round-to-nearest-even encoders of asq_fp8.hip (which are bit-identical
// to tensor.to(torch.float8_*))?  Sweeps ALL 2^32 fp32 bit patterns and counts mismatches per input class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t sw_e4m3(float f)
{
    const uint32_t u = __float_as_uint(f), sign = (u >> 24) & 0x80u, a = u & 0x7FFFFFFFu;
    uint32_t r;
    if (a >= 0x43F00000u) r = 0x7Fu;
    else if (a < 0x3C800000u) r = __float_as_uint(__uint_as_float(a) + 16384.0f) - 0x46800000u;
    else { uint32_t v = a - (120u << 23); v += 0x7FFFFu + ((v >> 20) & 1u); r = v >> 20; }
    return r | sign;
}
__device__ __forceinline__ uint32_t sw_e5m2(float f)
{
    const uint32_t u = __float_as_uint(f), sign = (u >> 24) & 0x80u, a = u & 0x7FFFFFFFu;
    uint32_t r;
    if (a > 0x7F800000u) r = 0x7Fu;
    else if (a >= 0x47800000u) r = 0x7Cu;
    else if (a < 0x38800000u) r = __float_as_uint(__uint_as_float(a) + 128.0f) - 0x43000000u;
    else { uint32_t v = a - (112u << 23); v += 0xFFFFFu + ((v >> 21) & 1u); r = v >> 21; }
    return r | sign;
}

// classes: 0 = |f| <= max finite (448 / 57344), 1 = between max finite and the overflow boundary, 2 = beyond (finite), 3 = inf, 4 = NaN
__global__ void sweep(unsigned long long *bad /*[2][5]*/, unsigned *example /*[2][5]*/)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long nb[2][5] = {};
    for (uint64_t p = tid; p < (1ull << 32); p += nthreads) {
        const uint32_t u = (uint32_t)p, a = u & 0x7FFFFFFFu;
        const float f = __uint_as_float(u);
        const uint32_t hw8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false) & 0xFFu;
        const uint32_t hwb = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(f, f, 0, false) & 0xFFu;
        const int c8 = a > 0x7F800000u ? 4 : a == 0x7F800000u ? 3 : a >= 0x43F00000u ? 2 : a > 0x43E00000u ? 1 : 0;
        const int cb = a > 0x7F800000u ? 4 : a == 0x7F800000u ? 3 : a >= 0x47800000u ? 2 : a > 0x47600000u ? 1 : 0;
        const uint32_t s8 = sw_e4m3(f), sb = sw_e5m2(f);
        const bool nan8 = c8 == 4, nanb = cb == 4;  // any NaN byte counts as equal to any other NaN byte
        if (hw8 != s8 && !(nan8 && (hw8 & 0x7F) == 0x7F)) { if (!nb[0][c8]) atomicCAS(&example[c8], 0u, u); ++nb[0][c8]; }
        if (hwb != sb && !(nanb && (hwb & 0x7C) == 0x7C && (hwb & 3))) { if (!nb[1][cb]) atomicCAS(&example[5 + cb], 0u, u); ++nb[1][cb]; }
    }
    for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 5; ++c)
            if (nb[i][c]) atomicAdd(&bad[i * 5 + c], nb[i][c]);
}

int main()
{
    unsigned long long *bad; unsigned *ex;
    CK(hipMalloc(&bad, 80)); CK(hipMalloc(&ex, 40));
    CK(hipMemset(bad, 0, 80)); CK(hipMemset(ex, 0, 40));
    hipLaunchKernelGGL(sweep, dim3(8192), dim3(256), 0, 0, bad, ex);
    CK(hipDeviceSynchronize());
    unsigned long long h[10]; unsigned exh[10];
    CK(hipMemcpy(h, bad, 80, hipMemcpyDeviceToHost)); CK(hipMemcpy(exh, ex, 40, hipMemcpyDeviceToHost));
    const char *cls[5] = {"in range", "max..overflow boundary", "beyond (finite)", "inf", "NaN"};
    for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 5; ++c) printf("%s  %-24s mismatches %llu  (first input bits %08x)\n", i ? "e5m2" : "e4m3", cls[c], h[i * 5 + c], exh[i * 5 + c]);
    return 0;
}
