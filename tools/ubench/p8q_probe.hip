// Timing probe for gemm_i8_p8q schedule variants (dev tool).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DP8Q_VAR=n] p8q_probe.hip
#include "../../autosmoothquant_amd/csrc/asq_gemm_kernels.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
using namespace asq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace asq { int forced_kernel() { return -1; } }
void asq_set_error(const char *, ...) {}
int asq_debug_sync() { return 0; }

int main(int argc, char **argv)
{
    const int64_t K = argc > 3 ? atoll(argv[3]) : 4096;
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1024, N = argc > 2 ? atoll(argv[2]) : 4096;
    std::vector<int8_t> hx(M * K), hw(N * K);
    unsigned s = 12345;
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (int8_t)((s >> 24) % 7) - 3; }
    for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    int8_t *x, *w; void *o;
    CK(hipMalloc(&x, M * K)); CK(hipMalloc(&w, N * K)); CK(hipMalloc(&o, M * N * 2));
    CK(hipMemcpy(x, hx.data(), M * K, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), N * K, hipMemcpyHostToDevice));
    EpiDequant<ASQ_F16, false, false, false> e{o, N, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch, const char *name, double tiles) {
        for (int i = 0; i < 50; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
        }
        printf("%s M=%lld N=%lld K=%lld: %.2f us/launch -> %.0f TOPS (%.0f tiles)\n", name, (long long)M, (long long)N, (long long)K, best / 20 * 1e3, 2.0 * M * N * K / (best / 20) / 1e9, tiles);
    };
    {
        auto kfn = gemm_i8_p8q<decltype(e)>;
        CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8Q_LDS_BYTES));
        const int tm = (int)((M + 127) / 128), tn = (int)((N + 127) / 128);
        time([&] { hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8Q_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, e, (const uint8_t *)nullptr, (const uint8_t *)nullptr); }, "p8q", tm * tn);
    }
    {
        auto kfn = gemm_i8_p8h<decltype(e)>;
        CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8H_LDS_BYTES));
        const int tm = (int)((M + 127) / 128), tn = (int)((N + 255) / 256);
        time([&] { hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8H_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, e); }, "p8h", tm * tn);
    }
    return 0;
}
