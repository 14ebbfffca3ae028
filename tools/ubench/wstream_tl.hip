// Per-block phase timeline of gemm_i8_wstream (dev tool): s_memrealtime stamps at block start, first data, end of the stream,
// after the slab-store drain, after the ticket, after each finished group, at exit.
// Build: make wstream_tl      usage: wstream_tl M N K [G] [iters]
#define ASQ_WS_STAMPS 1
#include "../../autosmoothquant_amd/csrc/asq_gemm_kernels.h"
#include <vector>
#include <algorithm>
void asq_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
int asq_debug_sync() { return 0; }
namespace asq { int forced_kernel() { return -1; } }
using namespace asq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MT> void run(const int8_t* x, const int8_t* const* w, int nw, void* out, char* ws, long M, long N, long K, int G, int iters)
{
    using Epi = EpiDequant<ASQ_F16, false, false, false>;
    Epi epi{out, N, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};
    WsPlan p = plan_wstream(M, N, K);
    if (G > 0) { p.G = G; const long nun_max = (p.T + G - 1) / G; p.maxseg = (int)((nun_max + p.KU - 1) / p.KU + 1); }
    auto kfn = gemm_i8_wstream<Epi, MT, false>;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, WsCfg<MT>::LDS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(kfn, dim3(p.G), dim3(512), WsCfg<MT>::LDS, 0, x, w[i % nw], M, N, K, p.KU, p.T, p.maxseg, ws, epi);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kfn, dim3(p.G), dim3(512), WsCfg<MT>::LDS, 0, x, w[i % nw], M, N, K, p.KU, p.T, p.maxseg, ws, epi);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    static unsigned long long h[1024][8];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ws_stamps), sizeof(h)));
    unsigned long long t0 = ~0ull, te = 0;
    for (int i = 0; i < p.G; ++i) { t0 = std::min(t0, h[i][0]); te = std::max(te, h[i][6]); }
    const char* names[7] = {"start", "first data", "stream end", "stores drained", "ticket known", "last group done", "exit"};
    printf("M=%ld N=%ld K=%ld G=%d MT=%d: %.2f us per launch (batch of %d); last launch spans %.2f us first start -> last exit (10 ns ticks)\n", M, N, K, p.G, MT, ms * 1e3 / iters, iters, (te - t0) * 0.01);
    for (int s = 0; s < 7; ++s) {
        std::vector<double> v;
        for (int i = 0; i < p.G; ++i) if (h[i][s] >= t0 && h[i][s] <= te) v.push_back((h[i][s] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("   %-16s n=%3zu  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us after the first block's start\n", names[s], v.size(), v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
}

int main(int argc, char** argv)
{
    const long M = argc > 1 ? atol(argv[1]) : 32, N = argc > 2 ? atol(argv[2]) : 4096, K = argc > 3 ? atol(argv[3]) : 4096;
    const int G = argc > 4 ? atoi(argv[4]) : 0, iters = argc > 5 ? atoi(argv[5]) : 24;
    const size_t wb = (size_t)N * K;
    int nw = (int)(300e6 / (double)wb) + 1; if (nw > 6) nw = 6;
    std::vector<int8_t> hbuf(wb);
    unsigned s = 777;
    for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    int8_t* w[6]; int8_t* x; void* out; char* ws;
    for (int i = 0; i < nw; ++i) { CK(hipMalloc(&w[i], wb)); CK(hipMemcpy(w[i], hbuf.data(), wb, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&x, 128 * K)); CK(hipMemcpy(x, hbuf.data(), 128 * K < (long)wb ? 128 * K : wb, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 128 * N * 4)); CK(hipMalloc(&ws, 96u << 20));
    hipLaunchKernelGGL(ws_init_header, dim3(1), dim3(256), 0, 0, (unsigned*)ws);
    CK(hipDeviceSynchronize());
    const int mt = M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : 8;
    if (mt == 1) run<1>(x, w, nw, out, ws, M, N, K, G, iters);
    else if (mt == 2) run<2>(x, w, nw, out, ws, M, N, K, G, iters);
    else if (mt == 4) run<4>(x, w, nw, out, ws, M, N, K, G, iters);
    else run<8>(x, w, nw, out, ws, M, N, K, G, iters);
    return 0;
}
