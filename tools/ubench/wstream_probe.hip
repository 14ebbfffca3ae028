// A/B probe for the two weight-streaming kernels through the C-ABI (no Python launch overhead):
//   old = gemm_i8_skinny   (no workspace passed)          new = gemm_i8_wstream (initialised workspace passed)
// 1. exactness: int32 output and fp16 dequant output of `new` against `old` (itself pinned by the test suite), every element,
//    three launches each (tickets must come back to zero), one of them with a different M in between;
// 2. timing: batches of back-to-back launches over rotating weight buffers (> 256 MiB, so weights come from HBM) and, with
//    `warm`, over ONE buffer (weights resident in L2 / Infinity Cache).
// usage: wstream_probe [check|time|both] [warm]     env: ASQ_WS_GRID, ASQ_WS_NT, ASQ_SK_IMPL (read by the library)
#include "../../include/asq_hip.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("rc=%d %s (line %d)\n", rc_, asq_last_error(), __LINE__); exit(1);} } while (0)

struct S { long M, N, K; };

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "both";
    const bool warm = argc > 2 && !strcmp(argv[2], "warm");
    std::vector<S> shapes;
    const long nk[][2] = {{4096, 4096}, {11008, 4096}, {4096, 11008}, {5120, 20480}, {20480, 5120}, {14336, 4096}, {4096, 14336}, {1024, 4096}, {12288, 4096}, {8192, 8192}, {5120, 5120}};
    const long ms[] = {1, 4, 8, 16, 32, 64, 128};
    for (auto& s : nk) for (long m : ms) shapes.push_back({m, s[0], s[1]});
    const size_t maxw = 5120L * 20480;
    const int NB = 6;  // 6 x 105 MB > 256 MiB
    int8_t* w[NB]; int8_t* x; void *out, *out2, *ws;
    std::vector<int8_t> h(maxw);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&w[i], maxw)); CK(hipMemcpy(w[i], h.data(), maxw, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&x, 128L * 20480)); CK(hipMemcpy(x, h.data() + 12345, 128L * 20480, hipMemcpyHostToDevice));
    const size_t outb = 128L * 20480 * 4;
    CK(hipMalloc(&out, outb)); CK(hipMalloc(&out2, outb));
    const size_t wsb = 96u << 20;
    CK(hipMalloc(&ws, wsb));
    CK(hipMemset(ws, 0xA5, wsb));  // garbage first: the header must come from asq_workspace_init alone
    RC(asq_workspace_init(ws, wsb, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<int32_t> a(128L * 20480), b(128L * 20480);
    const size_t hdrw = asq_workspace_header_bytes() / 4;
    std::vector<uint32_t> hdr(hdrw);

    if (!strcmp(mode, "one")) {  // wstream_probe one M N K [nows]: 40 launches of one shape over rotating weights (for rocprofv3 passes)
        const long M = atol(argv[2]), N = atol(argv[3]), K = atol(argv[4]);
        const bool nows = argc > 5 && !strcmp(argv[5], "nows");
        int nb = (int)(300e6 / ((double)N * K)) + 1; if (nb > NB) nb = NB;
        for (int i = 0; i < 40; ++i) RC(asq_linear_w8a8(x, w[i % nb], out, ASQ_F16, M, N, K, 1e-4f, nullptr, nullptr, nullptr, 0, nows ? nullptr : ws, nows ? 0 : wsb, nullptr));
        CK(hipDeviceSynchronize());
        {   // timed batches too (A/B of launcher knobs from the environment)
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9;
            for (int r = 0; r < 8; ++r) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 24; ++i) RC(asq_linear_w8a8(x, w[i % nb], out, ASQ_F16, M, N, K, 1e-4f, nullptr, nullptr, nullptr, 0, nows ? nullptr : ws, nows ? 0 : wsb, nullptr));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / 24 < best ? ms / 24 : best;
            }
            printf("one: %.2f us per launch (min of 8 batches of 24)\n", best * 1e3);
        }
        printf("one: M=%ld N=%ld K=%ld %s, 40 launches over %d weight buffers, kernel class %s, workspace need %zu\n", M, N, K, nows ? "no workspace" : "workspace", nb,
               asq_gemm_kernel_name(M, N, K), asq_gemm_workspace_bytes(M, N, K));
        return 0;
    }
    if (!strcmp(mode, "check") || !strcmp(mode, "both")) {
        // ragged shapes first (N not a multiple of 128 / 16, M not a multiple of 16), then the grid
        std::vector<S> cs = {{3, 200, 256}, {17, 1000, 384}, {33, 130, 128}, {70, 4100, 1024}, {128, 16, 4096}, {5, 129, 128 * 7}};
        for (auto& sh : shapes) cs.push_back(sh);
        long bad_total = 0;
        for (auto sh : cs) {
            if (asq_gemm_workspace_bytes(sh.M, sh.N, sh.K) == 0 || strcmp(asq_gemm_kernel_name(sh.M, sh.N, sh.K), "skinny")) continue;
            const size_t n = (size_t)sh.M * sh.N;
            RC(asq_gemm_i8_i32(x, w[1], (int32_t*)out, sh.M, sh.N, sh.K, nullptr, 0, nullptr));
            CK(hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost));
            long bad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(out2, 0xFF, n * 4));
                RC(asq_gemm_i8_i32(x, w[1], (int32_t*)out2, sh.M, sh.N, sh.K, ws, wsb, nullptr));
                CK(hipMemcpy(b.data(), out2, n * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; ++i) {
                    if (a[i] != b[i] && getenv("DETAIL") && bad < 48)
                        printf("  rep %d m=%zu n=%zu (group %zu wave %zu fg %zu j %zu) got %d want %d\n", rep, i / sh.N, i % sh.N, (i % sh.N) / 128, ((i % sh.N) % 128) / 16, ((i % sh.N) % 16) / 4, (i % sh.N) % 4, b[i], a[i]);
                    bad += a[i] != b[i];
                }
                if (rep == 0) RC(asq_gemm_i8_i32(x, w[2], (int32_t*)out2, sh.M > 1 ? sh.M - 1 : 1, sh.N, sh.K, ws, wsb, nullptr));  // another shape in between
            }
            // fp16 epilogue with per-row scales and bias
            float *srow, *bias; CK(hipMalloc(&srow, 128 * 4)); CK(hipMalloc(&bias, sh.N * 4));
            std::vector<float> f(sh.N > 128 ? sh.N : 128);
            for (size_t i = 0; i < f.size(); ++i) f[i] = 1e-3f * (float)(1 + i % 7);
            CK(hipMemcpy(srow, f.data(), 128 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, f.data(), sh.N * 4, hipMemcpyHostToDevice));
            RC(asq_linear_w8a8(x, w[1], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, srow, nullptr, bias, 0, nullptr, 0, nullptr));
            RC(asq_linear_w8a8(x, w[1], out2, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, srow, nullptr, bias, 0, ws, wsb, nullptr));
            std::vector<uint16_t> ha(n), hb(n);
            CK(hipMemcpy(ha.data(), out, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), out2, n * 2, hipMemcpyDeviceToHost));
            long badh = 0;
            for (size_t i = 0; i < n; ++i) badh += ha[i] != hb[i];
            CK(hipFree(srow)); CK(hipFree(bias));
            CK(hipMemcpy(hdr.data(), ws, hdrw * 4, hipMemcpyDeviceToHost));
            long dirty = 0;
            for (size_t i = 4; i < hdrw; ++i) dirty += hdr[i] != 0;  // accumulator tiles back at zero
            if (bad || badh || dirty) printf("MISMATCH M=%ld N=%ld K=%ld: i32 %ld  f16 %ld  dirty tickets %ld\n", sh.M, sh.N, sh.K, bad, badh, dirty);
            bad_total += bad + badh + dirty;
        }
        printf("check: %zu shapes, %ld mismatching elements / dirty tickets\n", cs.size(), bad_total);
    }
    if (!strcmp(mode, "time") || !strcmp(mode, "both")) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("%s weights; us per launch (min / avg of 8 batches of 24 launches), TB/s = algorithmic bytes / min\n", warm ? "WARM (one buffer)" : "COLD (rotating > 256 MiB)");
        for (auto sh : shapes) {
            if (strcmp(asq_gemm_kernel_name(sh.M, sh.N, sh.K), "skinny")) continue;
            double res[2][2];
            for (int impl = 0; impl < 2; ++impl) {
                void* wsp = impl ? ws : nullptr; const size_t wsn = impl ? wsb : 0;
                const int batch = 24;
                for (int i = 0; i < 5; ++i) RC(asq_linear_w8a8(x, w[warm ? 0 : i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, wsp, wsn, nullptr));
                CK(hipDeviceSynchronize());
                float best = 1e9, sum = 0; const int it = 8;
                for (int r = 0; r < it; ++r) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < batch; ++i) RC(asq_linear_w8a8(x, w[warm ? 0 : i % NB], out, ASQ_F16, sh.M, sh.N, sh.K, 1e-4f, nullptr, nullptr, nullptr, 0, wsp, wsn, nullptr));
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= batch; best = ms < best ? ms : best; sum += ms;
                }
                res[impl][0] = best * 1e3; res[impl][1] = sum / it * 1e3;
            }
            const double bytes = (double)sh.N * sh.K + sh.M * sh.K + 2.0 * sh.M * sh.N;
            printf("M=%3ld N=%5ld K=%5ld: old %6.2f / %6.2f   new %6.2f / %6.2f   (%+5.1f %%)   new %5.2f TB/s\n", sh.M, sh.N, sh.K, res[0][0], res[0][1], res[1][0], res[1][1],
                   100.0 * (res[1][0] / res[0][0] - 1.0), bytes / (res[1][0] * 1e-6) / 1e12);
        }
    }
    return 0;
}
