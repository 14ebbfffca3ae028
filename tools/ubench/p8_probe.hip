// Ablation probe for the p8 GEMM schedule (dev tool; timing only, ablated results are invalid).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off p8_probe.hip -o p8_probe
#define ASQ_P8_PROBE 1
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

template <int ABL> float run(const int8_t* x, const int8_t* w, int32_t* out, int64_t M, int64_t N, int64_t K, int iters)
{
    EpiI32 epi{out, N, true};
    auto kfn = gemm_i8_p8<EpiI32, ABL>;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
    int tm = (M + 255) / 256, tn = (N + 255) / 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, epi, OffsetArgs{});
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, epi, OffsetArgs{});
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
    }
    printf("  ABL=%2d (%s%s%s%s) M=%5lld: min %.1f us avg %.1f us  -> %.0f TOPS-equivalent\n", ABL, (ABL & 1) ? "noDMA " : "", (ABL & 2) ? "noDSREAD " : "",
           (ABL & 4) ? "noMFMA " : "", (ABL & 16) ? "2xDSREAD " : (ABL & 8) ? "noBAR " : "", (long long)M, best * 1e3, sum / iters * 1e3, 2.0 * M * N * K / (sum / iters) / 1e9);
    return best;
}

template <class Epi> void blk_timeline(const int8_t* x, const int8_t* w, Epi epi, int64_t M, int64_t N, int64_t K, const char* tag)
{
    auto kfn = gemm_i8_p8<Epi, 128>;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
    int tm = (M + 255) / 256, tn = (N + 255) / 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, epi, OffsetArgs{});
    CK(hipDeviceSynchronize());
    static unsigned long long h[4096][8];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
    int nb = tm * tn; if (nb > 4096) nb = 4096;
    unsigned long long t0 = ~0ull, t3 = 0;
    for (int b = 0; b < nb; ++b) { if (h[b][0] < t0) t0 = h[b][0]; if (h[b][3] > t3) t3 = h[b][3]; }
    double s[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        double d[4] = {(double)(h[b][0] - t0), (double)(h[b][1] - h[b][0]), (double)(h[b][2] - h[b][1]), (double)(h[b][3] - h[b][2])};
        for (int i = 0; i < 4; ++i) { s[i] += d[i]; if (d[i] > mx[i]) mx[i] = d[i]; }
    }
    printf("  [%s] %d blocks, span %llu ticks; avg/max ticks: start-skew %.0f/%.0f  prologue %.0f/%.0f  kloop %.0f/%.0f  epilogue %.0f/%.0f\n", tag, nb, t3 - t0,
           s[0] / nb, mx[0], s[1] / nb, mx[1], s[2] / nb, mx[2], s[3] / nb, mx[3]);
    double xs[8] = {0}; int xn[8] = {0};
    for (int b = 0; b < nb; ++b) { int xc = (int)(h[b][4] & 7); xs[xc] += (double)(h[b][2] - h[b][1]); xn[xc]++; }
    printf("    kloop avg by XCC:");
    for (int i = 0; i < 8; ++i) printf(" %d:%.0f(n=%d)", i, xn[i] ? xs[i] / xn[i] : 0.0, xn[i]);
    int c = 0; for (int b = 0; b < nb; ++b) c += ((b & 7) == (int)(h[b][4] & 7));
    printf("\n    blockIdx%%8 == xcc for %d of %d blocks\n", c, nb);
}

template <class Epi> void blk_timeline_p8h(const int8_t* x, const int8_t* w, Epi epi, int64_t M, int64_t N, int64_t K, const char* tag)
{
    auto kfn = gemm_i8_p8h<Epi, true>;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8H_LDS_BYTES));
    int tm = (M + 127) / 128, tn = (N + 255) / 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8H_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, epi);
    CK(hipDeviceSynchronize());
    static unsigned long long h[4096][8];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
    int nb = tm * tn; if (nb > 4096) nb = 4096;
    double s[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        double d[3] = {(double)(h[b][1] - h[b][0]), (double)(h[b][2] - h[b][1]), (double)(h[b][3] - h[b][2])};
        for (int i = 0; i < 3; ++i) { s[i] += d[i]; if (d[i] > mx[i]) mx[i] = d[i]; }
    }
    printf("  [p8h %s] %d blocks; avg/max cycles: prologue %.0f/%.0f  kloop %.0f/%.0f  epilogue %.0f/%.0f   (K-loop floor %lld)\n", tag, nb, s[0] / nb, mx[0], s[1] / nb, mx[1],
           s[2] / nb, mx[2], (long long)(K / 128) * 1024);
}

int main()
{
    const int64_t N = 4096, K = 4096, MMAX = 4096;
    std::vector<int8_t> hx(MMAX * K), hw(N * K);
    uint32_t s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    int8_t *x, *w; int32_t* out;
    CK(hipMalloc(&x, hx.size())); CK(hipMalloc(&w, hw.size())); CK(hipMalloc(&out, MMAX * N * 4));
    CK(hipMemcpy(x, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice));
    {
        void* o16; CK(hipMalloc(&o16, MMAX * N * 2));
        EpiDequant<ASQ_F16, false, false, false> e16{o16, N, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};
        blk_timeline(x, w, e16, 4096, N, K, "4096^3 f16 epilogue");
        blk_timeline(x, w, e16, 256, N, K, "M=256 f16 epilogue");
        EpiI32 e32{out, N, true};
        blk_timeline(x, w, e32, 4096, N, K, "4096^3 i32 epilogue");
        blk_timeline_p8h(x, w, e16, 4096, N, K, "4096^3 f16 epilogue (512 blocks, 2 per CU in turn)");
        blk_timeline_p8h(x, w, e16, 2048, N, K, "M=2048 f16 epilogue (256 blocks)");
        blk_timeline_p8h(x, w, e16, 256, N, K, "M=256 f16 epilogue (32 blocks)");
    }
    for (int64_t M : {256, 4096}) {
        run<32>(x, w, out, M, N, K, 3);
        {
            unsigned long long h[2][4][8];
            CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_dbg), sizeof(h)));
            const char* names[7] = {"issue2glds", "issue_dsreads", "vmcnt_wait", "barrier1", "lgkm_wait", "mfma_issue", "barrier2"};
            int nt = (int)(K / 128);
            for (int g = 0; g < 2; ++g) {
                printf("    wave %d (group %d): avg cycles per phase segment over %d K-tiles\n", g * 4, g, nt);
                for (int ph = 0; ph < 4; ++ph) {
                    printf("      P%d:", ph + 1);
                    double tot = 0;
                    for (int q = 0; q < 7; ++q) { printf(" %s=%.0f", names[q], (double)h[g][ph][q] / nt); tot += (double)h[g][ph][q] / nt; }
                    printf("  | total %.0f\n", tot);
                }
            }
        }
        printf("M=%lld N=%lld K=%lld\n", (long long)M, (long long)N, (long long)K);
        run<0>(x, w, out, M, N, K, 20);
        run<1>(x, w, out, M, N, K, 20);
        run<2>(x, w, out, M, N, K, 20);
        run<4>(x, w, out, M, N, K, 20);
        run<8>(x, w, out, M, N, K, 20);
        run<3>(x, w, out, M, N, K, 20);
        run<6>(x, w, out, M, N, K, 20);
        run<5>(x, w, out, M, N, K, 20);
        run<7>(x, w, out, M, N, K, 20);
        run<11>(x, w, out, M, N, K, 20);
        run<16>(x, w, out, M, N, K, 20);   // fragment reads doubled
    }
    return 0;
}
