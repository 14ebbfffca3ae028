// Ablation of the skinny GEMM: which operand stream costs what?  Build three times:
//   hipcc ... skinny_abl.hip -o skinny_abl_full ; -DSK_ABL_NOX -o skinny_abl_nox ; -DSK_ABL_NOW -o skinny_abl_now
#include "../../autosmoothquant_amd/csrc/asq_api.hip"
#include "../../autosmoothquant_amd/csrc/asq_quant.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f32.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f32_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f16.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_bf16.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f16_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_bf16_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm.hip"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main()
{
    struct S { long M, N, K; };
    std::vector<S> shapes = {{1, 4096, 4096}, {32, 4096, 4096}, {64, 4096, 4096}, {32, 5120, 20480}, {32, 14336, 4096}};
    const size_t maxw = 5120L * 20480; const int NB = 4;
    int8_t* w[NB]; int8_t* x; void* out;
    std::vector<int8_t> h(maxw); unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&w[i], maxw)); CK(hipMemcpy(w[i], h.data(), maxw, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&x, 64L * 20480)); CK(hipMemcpy(x, h.data(), 64L * 20480, hipMemcpyHostToDevice)); CK(hipMalloc(&out, 64L * 14336 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto sh : shapes) {
        const int batch = 24;
        for (int i = 0; i < 5; ++i) asq_linear_w8a8(x, w[i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(a));
            for (int i = 0; i < batch; ++i) asq_linear_w8a8(x, w[i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= batch; best = ms < best ? ms : best;
        }
        printf("  M=%3ld N=%5ld K=%5ld: %6.2f us\n", sh.M, sh.N, sh.K, best * 1e3);
    }
    return 0;
}
