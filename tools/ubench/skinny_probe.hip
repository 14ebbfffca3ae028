// Timing probe for the skinny (M <= 64) GEMM path straight through the C-ABI (no Python launch
// overhead): batches of back-to-back launches over rotating weight buffers (> 256 MiB MALL).
#include "../../include/asq_hip.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    struct S { long M, N, K; };
    std::vector<S> shapes = {{1, 4096, 4096}, {4, 4096, 4096}, {16, 4096, 4096}, {32, 4096, 4096}, {64, 4096, 4096}, {32, 11008, 4096}, {32, 4096, 11008}, {32, 5120, 20480}, {32, 14336, 4096}, {32, 1024, 4096}, {64, 11008, 4096}, {64, 14336, 4096}, {16, 11008, 4096}, {32, 20480, 5120}, {32, 8192, 8192}, {64, 8192, 8192}};
    const size_t maxw = 5120L * 20480;
    const int NB = 6;  // 6 x 105 MB > 256 MiB
    int8_t* w[NB]; int8_t* x; void* out;
    std::vector<int8_t> h(maxw);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&w[i], maxw)); CK(hipMemcpy(w[i], h.data(), maxw, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&x, 64L * 20480)); CK(hipMemcpy(x, h.data(), 64L * 20480, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 64L * 14336 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto sh : shapes) {
        const int batch = 24;
        int nb = (int)((300e6 / (double)(sh.N * sh.K)) + 1); if (nb > NB) nb = NB;   // rotate through > 256 MiB when the shape allows
        // weights for shape sh live at the start of each buffer (contiguous [N,K])
        for (int i = 0; i < 5; ++i) asq_linear_w8a8(x, w[i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
        CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0; const int it = 8;
        for (int r = 0; r < it; ++r) {
            CK(hipEventRecord(a));
            for (int i = 0; i < batch; ++i) {
                int rc = asq_linear_w8a8(x, w[i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
                if (rc) { printf("rc=%d %s\n", rc, asq_last_error()); return 1; }
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= batch; best = ms < best ? ms : best; sum += ms;
        }
        double bytes = (double)sh.N * sh.K + sh.M * sh.K + 2.0 * sh.M * sh.N;
        printf("%-8s M=%3ld N=%5ld K=%5ld: min %6.2f us avg %6.2f us | %5.2f TB/s algorithmic (of ~6.3 achievable) | %6.1f TOPS\n", asq_gemm_kernel_name(sh.M, sh.N, sh.K), sh.M, sh.N,
               sh.K, best * 1e3, sum / it * 1e3, bytes / (best * 1e-3) / 1e12, 2.0 * sh.M * sh.N * sh.K / (best * 1e-3) / 1e12);
    }
    return 0;
}
