// Calibrates the achievable read bandwidth for "one pass over W" at decode sizes:
//   A: contiguous   -- each wave-instruction reads 1 KiB contiguous (perfect coalescing)
//   B: fragment     -- each wave-instruction reads 16 rows x 64 B (the MFMA 16x16x64 A-fragment shape)
//   C: fragment32   -- 8 rows x 128 B per instruction (full lines; lane = row*8 + chunk)
// Each block (1024 thr) covers 16 rows x K bytes, like gemm_i8_skinny.  Sum-reduces so loads stay live.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, bool NT> __global__ void __launch_bounds__(1024) rd(const int8_t* __restrict__ w, long K, int* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n0 = (long)blockIdx.x * 16;
    v4i s = {0, 0, 0, 0};
    const int nunits = (int)(K / 128);
    for (int u0 = wave; u0 < nunits; u0 += 64) {
        v4i v[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int u = u0 + i * 16;
            if (u < nunits) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int8_t* p;
                    if (MODE == 0) p = w + n0 * K + ((long)(u * 2 + h) * 16 * 64 / 64) * 64 + 0;  // placeholder, replaced below
                    if (MODE == 0) { long idx = ((long)u * 2 + h) * 1024 + lane * 16; p = w + n0 * K + idx; }           // block's 16*K bytes read as one contiguous span
                    if (MODE == 1) p = w + (n0 + (lane & 15)) * K + (long)u * 128 + 64 * h + 16 * (lane >> 4);          // 16 rows x 64 B
                    if (MODE == 2) p = w + (n0 + 8 * h + (lane >> 3)) * K + (long)u * 128 + 16 * (lane & 7);            // 8 rows x 128 B
                    v[i][h] = NT ? __builtin_nontemporal_load((const v4i*)p) : *(const v4i*)p;
                }
            } else { v[i][0] = (v4i){0,0,0,0}; v[i][1] = (v4i){0,0,0,0}; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { s += v[i][0]; s += v[i][1]; }
    }
    if (s[0] + s[1] + s[2] + s[3] == 0x12345678) out[0] = 1;
}

template <int MODE, bool NT> void run(const int8_t* w, long N, long K, int* out, const char* tag)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rd<MODE, NT>), dim3(N / 16), dim3(1024), 0, 0, w, K, out);
    CK(hipDeviceSynchronize());
    float best = 1e9, sum = 0; const int it = 30;
    for (int i = 0; i < it; ++i) {
        CK(hipEventRecord(a)); hipLaunchKernelGGL((rd<MODE, NT>), dim3(N / 16), dim3(1024), 0, 0, w, K, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
    }
    printf("  %-28s N=%5ld K=%5ld (%.1f MB): min %.1f us avg %.1f us -> %.2f TB/s (min)\n", tag, N, K, N * K / 1e6, best * 1e3, sum / it * 1e3, N * K / (best * 1e-3) / 1e12);
}

__global__ void empty() {}

int main()
{
    int8_t* w; int* out; CK(hipMalloc(&w, 5120L * 20480)); CK(hipMalloc(&out, 4)); CK(hipMemset(w, 1, 5120L * 20480));
    {   // empty-kernel event floor
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e9;
        for (int i = 0; i < 30; ++i) { CK(hipEventRecord(a)); hipLaunchKernelGGL(empty, dim3(256), dim3(1024), 0, 0); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; }
        printf("empty kernel (256 x 1024 thr) event-to-event floor: %.1f us\n", best * 1e3);
    }
    for (auto [N, K] : std::vector<std::pair<long, long>>{{4096, 4096}, {11008, 4096}, {4096, 11008}, {5120, 20480}}) {
        run<0, false>(w, N, K, out, "contiguous");
        run<0, true>(w, N, K, out, "contiguous nt");
        run<1, false>(w, N, K, out, "fragment 16x64B");
        run<1, true>(w, N, K, out, "fragment 16x64B nt");
        run<2, false>(w, N, K, out, "rows 8x128B");
        run<2, true>(w, N, K, out, "rows 8x128B nt");
    }
    return 0;
}
