// Microbenchmark: issue cost (shader cycles per wave instruction) of the VALU instructions a GEMM epilogue is made of
// (int32 -> fp32 -> scaled -> packed fp16), with ONE and with TWO waves per SIMD.  One block on one CU.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum { CVT_I2F = 0, MUL, PKMUL, CVTPK, SEQ_PK, SEQ_MUL, SEQ_MAGIC, ADDU, ACCREAD, FMA, SEQ_FMA_MAGIC, NMODE };
static const char *names[NMODE] = {
    "32 x v_cvt_f32_i32", "32 x v_mul_f32", "16 x v_pk_mul_f32 (32 products)", "16 x v_cvt_pk_f16_f32 (32 values)",
    "32 outputs: 32 cvt + 16 pk_mul + 16 cvt_pk (shipped epilogue mix)", "32 outputs: 32 cvt + 32 mul + 16 cvt_pk",
    "32 outputs: 32 add_u32 + 32 sub_f32 + 32 mul + 16 cvt_pk (magic-number int->float)", "32 x v_add_u32",
    "32 x v_accvgpr_read_b32", "32 x v_fma_f32", "32 outputs: 32 add_u32 + 32 sub + 32 mul, no cvt_pk"};

template <int MODE>
__global__ void __launch_bounds__(512) body(int iters, float scale, int *out, long long *cycles)
{
    int a[32];
    float f[32];
    unsigned h[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a[i] = (int)threadIdx.x * 37 + i * 1000; f[i] = (float)(threadIdx.x + i); }
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0;
    v2f sc2 = {scale, scale};
    if constexpr (MODE == ACCREAD) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_write_b32 a%0, %1" : : "n"(i), "v"(a[i]));
    }
    asm volatile("s_nop 4" ::: "memory");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == CVT_I2F) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        } else if constexpr (MODE == MUL) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "v"(scale));
        } else if constexpr (MODE == FMA) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(f[i]) : "v"(scale));
        } else if constexpr (MODE == PKMUL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v2f p = {f[2 * i], f[2 * i + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p) : "v"(sc2));
                f[2 * i] = p[0]; f[2 * i + 1] = p[1];
            }
        } else if constexpr (MODE == CVTPK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(f[2 * i]), "v"(f[2 * i + 1]));
        } else if constexpr (MODE == ADDU) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(it));
        } else if constexpr (MODE == ACCREAD) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(a[i]) : "n"(i));
        } else if constexpr (MODE == SEQ_PK) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(t[j]) : "v"(a[4 * g + j]));
                v2f p0 = {t[0], t[1]}, p1 = {t[2], t[3]};
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p0) : "v"(sc2));
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p1) : "v"(sc2));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g]) : "v"(p0[0]), "v"(p0[1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g + 1]) : "v"(p1[0]), "v"(p1[1]));
            }
        } else if constexpr (MODE == SEQ_MUL) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(t[j]) : "v"(a[4 * g + j]));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t[j]) : "v"(scale));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g]) : "v"(t[0]), "v"(t[1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g + 1]) : "v"(t[2]), "v"(t[3]));
            }
        } else if constexpr (MODE == SEQ_MAGIC || MODE == SEQ_FMA_MAGIC) {
            const unsigned magic = 0x4B400000u;
            const float magicf = 12582912.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(t[j]) : "v"(a[4 * g + j]), "v"(magic));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(t[j]) : "v"(magicf));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t[j]) : "v"(scale));
                if constexpr (MODE == SEQ_MAGIC) {
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g]) : "v"(t[0]), "v"(t[1]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[2 * g + 1]) : "v"(t[2]), "v"(t[3]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[4 * g + j] = t[j];
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] + (int)f[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (int)h[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
static void run(int *out, long long *cyc)
{
    const int iters = 2000;
    int ninstr[NMODE] = {32, 32, 16, 16, 64, 80, 112, 32, 32, 32, 96};
    for (int waves = 4; waves <= 8; waves += 4) {
        body<MODE><<<1, waves * 64>>>(10, 0.37f, out, cyc);
        body<MODE><<<1, waves * 64>>>(iters, 0.37f, out, cyc);
        CK(hipDeviceSynchronize());
        long long h[8];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        const double per_iter = (double)mx / iters;
        printf("%-86s %d wave/SIMD: %7.1f cycles/iter = %5.2f cycles per instruction, %5.2f per output-pair of waves-worth (SIMD time per 32 items of ONE wave: %6.1f)\n",
               names[MODE], waves / 4, per_iter, per_iter / ninstr[MODE], per_iter / (waves / 4), per_iter / (waves / 4));
    }
}

int main()
{
    int *out; long long *cyc;
    CK(hipMalloc(&out, 512 * 4));
    CK(hipMalloc(&cyc, 8 * 8));
    run<CVT_I2F>(out, cyc); run<MUL>(out, cyc); run<FMA>(out, cyc); run<PKMUL>(out, cyc); run<CVTPK>(out, cyc); run<ADDU>(out, cyc); run<ACCREAD>(out, cyc);
    run<SEQ_PK>(out, cyc); run<SEQ_MUL>(out, cyc); run<SEQ_MAGIC>(out, cyc); run<SEQ_FMA_MAGIC>(out, cyc);
    return 0;
}
