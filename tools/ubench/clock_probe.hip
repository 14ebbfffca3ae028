// Clock / power evidence for the dominant kernel (dev tool, MI355X).
//
// Question: is gemm_i8_p8 at 4096^3 stall-bound or power(clock)-bound?
// Every block of the stamped build (ABL = 128: stamps only, schedule unchanged) records
//     s_memtime      (shader-clock ticks)           at block start / prologue end / K-loop end / block end
//     s_memrealtime  (constant 100 MHz ticks)       at block start / block end
// so   effective shader clock = d(memtime) / d(memrealtime) * 100 MHz   inside the kernel itself, no SMI needed.
// The same stamps go into an MFMA-only kernel (operands in registers, nothing but v_mfma_i32_32x32x32_i8):
// its rate on the SAME operand statistics is the matrix-core ceiling for that data.
//
// Operand statistics:
//   bench   what bench.py feeds the kernel: weights N(0, 0.02^2) absmax-quantised (int8 std ~23), activations
//           N(0,1) with 1 % outlier channels x20, scaled by absmax/127 (most entries |x| <= 3)
//   uniform full-range uniform int8 (the r1 ceiling of 3500 TOPS was measured on this)
//   zeros
//
//   usage: clock_probe [seconds per arm = 1.5] [what = all | gemm | mfma | abl | pmc | p4 | epi | p16]
//   build: see Makefile (-DASQ_P8_PROBE is set by this file)
#define ASQ_P8_PROBE 1
#include "../../autosmoothquant_amd/csrc/asq_api.hip"
#include "../../autosmoothquant_amd/csrc/asq_quant.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f32.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f16.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_bf16.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f32_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_f16_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm_inst_bf16_row.hip"
#include "../../autosmoothquant_amd/csrc/asq_gemm.hip"
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <math.h>
#include <glob.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------- operand generators
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    double uni() { return (u32() + 0.5) / 4294967296.0; }
    double gauss() { const double a = uni(), b = uni(); return sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b); }
};
static int8_t q8(double v) { double r = nearbyint(v); r = r > 127 ? 127 : (r < -128 ? -128 : r); return (int8_t)r; }

static void gen_weights_bench(std::vector<int8_t> &w, int64_t N, int64_t K, Rng &r)
{
    std::vector<float> f(N * K);
    float amax = 0;
    for (auto &v : f) { v = (float)(0.02 * r.gauss()); amax = fmaxf(amax, fabsf(v)); }
    const float sc = amax / 127.0f;
    for (int64_t i = 0; i < N * K; ++i) w[i] = q8(f[i] / sc);
}
static void gen_acts_bench(std::vector<int8_t> &x, int64_t M, int64_t K, Rng &r)
{
    std::vector<float> mul(K);
    for (auto &m : mul) m = r.uni() < 0.01 ? 20.0f : 1.0f;
    std::vector<float> f(M * K);
    float amax = 0;
    for (int64_t m = 0; m < M; ++m)
        for (int64_t k = 0; k < K; ++k) { float v = (float)r.gauss() * mul[k]; f[m * K + k] = v; amax = fmaxf(amax, fabsf(v)); }
    const float sc = amax / 127.0f;
    for (int64_t i = 0; i < M * K; ++i) x[i] = q8(f[i] / sc);
}
static void gen_uniform(std::vector<int8_t> &v, Rng &r) { for (auto &e : v) e = (int8_t)(r.u32() >> 24); }
static void stats(const char *name, const std::vector<int8_t> &v)
{
    double s2 = 0; int64_t small = 0;
    for (auto e : v) { s2 += (double)e * e; small += (e >= -3 && e <= 3); }
    printf("    %-10s int8 rms %.1f, %.1f %% of entries in [-3,3]\n", name, sqrt(s2 / v.size()), 100.0 * small / v.size());
}

// ---------------------------------------------------------------- sysfs power / clock sampler (best effort)
struct Sampler {
    std::vector<std::string> power_files, freq_files;
    std::atomic<bool> run{false};
    std::thread th;
    std::vector<double> pw, fq;
    static std::vector<std::string> globv(const char *pat)
    {
        std::vector<std::string> r; glob_t g;
        if (glob(pat, 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) r.push_back(g.gl_pathv[i]); globfree(&g); }
        return r;
    }
    static double readnum(const std::string &p) { FILE *f = fopen(p.c_str(), "r"); if (!f) return -1; double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f); return v; }
    Sampler()
    {
        for (const char *pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"})
            for (auto &p : globv(pat)) power_files.push_back(p);
        for (auto &p : globv("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")) freq_files.push_back(p);
        printf("  sysfs sampler: %zu power file(s), %zu sclk file(s)%s\n", power_files.size(), freq_files.size(), power_files.empty() ? " (none: rely on the in-kernel clock and the SMI log)" : "");
        for (auto &p : power_files) printf("      %s = %.0f\n", p.c_str(), readnum(p));
        for (auto &p : freq_files) printf("      %s = %.0f\n", p.c_str(), readnum(p));
    }
    void start()
    {
        pw.clear(); fq.clear(); run = true;
        th = std::thread([this] {
            while (run) {
                double p = 0, f = 0; int np = 0, nf = 0;
                for (auto &s : power_files) { double v = readnum(s); if (v > 0) { p = fmax(p, v); ++np; } }   // busiest device
                for (auto &s : freq_files) { double v = readnum(s); if (v > 0) { f = fmax(f, v); ++nf; } }
                if (np) pw.push_back(p / 1e6);
                if (nf) fq.push_back(f / 1e6);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    void stop(const char *tag)
    {
        run = false; th.join();
        auto mean_tail = [](const std::vector<double> &v) { if (v.empty()) return -1.0; size_t a = v.size() / 2; double s = 0; for (size_t i = a; i < v.size(); ++i) s += v[i]; return s / (v.size() - a); };
        if (!pw.empty() || !fq.empty())
            printf("    [sysfs %s: %zu samples, 2nd-half mean power %.0f W]\n", tag, pw.size() > fq.size() ? pw.size() : fq.size(), mean_tail(pw));
        else printf("\n");
    }
};

// ---------------------------------------------------------------- MFMA-only kernels with clock stamps
// PAT: order / operand sharing of consecutive MFMAs (64 per loop trip, operands static in registers)
//   0 distinct   every MFMA gets an A and a B fragment different from its predecessor's
//   1 r1         tools/ubench/mfma_power.hip's pattern: 4 A x 4 B fragments, neighbours share one operand
//   2 p8         gemm_i8_p8's K-loop order: per phase 4 k-steps x 2 token tiles, pairs share the W fragment
//   3 snake      2 x 2 (W x X) fragments per k-step walked (w0,x0)(w0,x1)(w1,x1)(w1,x0): 3 of 4 steps share an operand
// GAP: s_nop 15 (16 idle issue cycles) inserted after every 8 MFMAs -> matrix-pipe duty below 100 % (run with 4 waves
//      per block = one wave per SIMD so the idle is real)
__device__ unsigned long long mf_stamp[256][4];
#define MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_, b_, acc_, 0, 0, 0)
template <int GAP> __device__ __forceinline__ void mf_gap()
{
#pragma unroll
    for (int g = 0; g < GAP; ++g) asm volatile("s_nop 15");
}
template <int PAT, int GAP> __global__ void __launch_bounds__(512) mfma_var(const v4i *__restrict__ ab, int iters, int *out)
{
    v4i a[8], b[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < 8; ++i) a[i] = ab[((blockIdx.x * 8 + wave) * 24 + i) * 64 + lane];
    for (int i = 0; i < 16; ++i) b[i] = ab[((blockIdx.x * 8 + wave) * 24 + 8 + i) * 64 + lane];
    v16i acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v16i){0};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); }
    for (int it = 0; it < iters; it += 8) {
        if constexpr (PAT == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) MF(acc[i], a[(i + j) & 7], b[(i * 3 + j) & 15]);
                mf_gap<GAP>();
            }
        } else if constexpr (PAT == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) MF(acc[i], a[i & 3], b[(i >> 1) & 3]);
                mf_gap<GAP>();
            }
        } else if constexpr (PAT == 2) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {  // P1 (wa, xf), P2 (wb, xf), P3 (wb, xf'), P4 (wa, xf')
                    const int wsel = (ph == 1 || ph == 2) ? 4 : 0, xsel = (ph >= 2) ? 8 : 0;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int j = 0; j < 2; ++j) MF(acc[2 * ph + j], a[wsel + ks], b[xsel + j * 4 + ks]);
                    mf_gap<GAP>();
                }
            }
        } else {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        MF(acc[4 * h + 0], a[ks], b[8 * h + ks]);
                        MF(acc[4 * h + 1], a[ks], b[8 * h + 4 + ks]);
                        MF(acc[4 * h + 3], a[4 + ks], b[8 * h + 4 + ks]);
                        MF(acc[4 * h + 2], a[4 + ks], b[8 * h + ks]);
                        if (ks & 1) mf_gap<GAP>();
                    }
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        mf_stamp[blockIdx.x][0] = t0; mf_stamp[blockIdx.x][1] = __builtin_amdgcn_s_memtime();
        mf_stamp[blockIdx.x][2] = r0; mf_stamp[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static v4i *mf_operands(const std::vector<int8_t> &wsrc, const std::vector<int8_t> &xsrc)
{
    // fragments: lane l of a wave gets 16 consecutive k-bytes of row (l & 31), k-half (l >> 5) -- the GEMM's own layout;
    // per wave 8 W fragments then 16 X fragments
    const int nfr = 256 * 8 * 24;
    std::vector<int8_t> h((size_t)nfr * 64 * 16);
    Rng r(99);
    for (int f = 0; f < nfr; ++f) {
        const std::vector<int8_t> &src = ((f % 24) < 8) ? wsrc : xsrc;
        const size_t rows = src.size() / 4096, row0 = r.u32() % (rows - 32), k0 = (r.u32() % (4096 / 32)) * 32;
        for (int l = 0; l < 64; ++l) memcpy(&h[((size_t)f * 64 + l) * 16], &src[(row0 + (l & 31)) * 4096 + k0 + 16 * (l >> 5)], 16);
    }
    v4i *ab;
    CK(hipMalloc(&ab, h.size()));
    CK(hipMemcpy(ab, h.data(), h.size(), hipMemcpyHostToDevice));
    return ab;
}

template <int PAT, int GAP> static void run_mfma_var(const char *pat, const char *name, const v4i *ab, int waves, double seconds, Sampler &smp)
{
    static int *out = nullptr;
    if (!out) CK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = waves == 8 ? 20000 : 40000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto kfn = mfma_var<PAT, GAP>;
    hipLaunchKernelGGL(kfn, dim3(256), dim3(waves * 64), 0, 0, ab, iters, out);
    CK(hipDeviceSynchronize());
    const auto tstart = std::chrono::steady_clock::now();
    smp.start();
    std::vector<double> series, mss;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() < seconds) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kfn, dim3(256), dim3(waves * 64), 0, 0, ab, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[256][4];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(mf_stamp), sizeof(st)));
        double clk = 0;
        for (int b = 0; b < 256; ++b) clk += (double)(st[b][1] - st[b][0]) / (double)(st[b][3] - st[b][2]) * 0.1;  // GHz
        series.push_back(clk / 256); mss.push_back(ms);
    }
    auto tail = [](const std::vector<double> &v) { double s = 0; size_t a = v.size() / 2; for (size_t i = a; i < v.size(); ++i) s += v[i]; return s / (v.size() - a); };
    const double ops = 2.0 * 32 * 32 * 32 * 8.0 * iters * waves * 256, ms = tail(mss), clk = tail(series), tops = ops / ms / 1e9;
    printf("  MFMA-only %-8s %-7s gap %d, %d waves/CU: %.2f ms -> %5.0f TOPS @ %.3f GHz in-kernel; matrix-pipe duty %.1f %% (op/clk/CU %.0f of 8192)", pat, name, GAP, waves, ms, tops, clk,
           100.0 * tops * 1e12 / (clk * 1e9) / 256 / 8192, tops * 1e12 / (clk * 1e9) / 256);
    smp.stop("");
    fflush(stdout);
}
#define RUN_MF(PAT, GAP, patname, dname, ab, waves) run_mfma_var<PAT, GAP>(patname, dname, ab, waves, seconds, smp)

// ---------------------------------------------------------------- the production schedule with stamps
template <class Epi, int XABL = 0> static void run_gemm(const char *name, const int8_t *x, const int8_t *w, Epi epi, int64_t M, int64_t N, int64_t K, double seconds, Sampler &smp)
{
    auto kfn = gemm_i8_p8<Epi, 128 | XABL>;   // stamps only (+ the store policy under test)
    auto kprod = gemm_i8_p8<Epi, XABL>;       // without stamps, for the wall-clock comparison
    CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)kprod, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
    const int tm = (int)((M + 255) / 256), tn = (int)((N + 255) / 256), nb = tm * tn < 4096 ? tm * tn : 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int BATCH = 50;
    auto launch = [&](bool prod) {
        if (prod) hipLaunchKernelGGL(kprod, dim3(tm * tn), dim3(512), P8_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, epi, OffsetArgs{});
        else hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P8_LDS_BYTES, 0, x, w, M, N, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, epi, OffsetArgs{});
    };
    for (int i = 0; i < 5; ++i) launch(false);
    CK(hipDeviceSynchronize());
    smp.start();
    const auto tstart = std::chrono::steady_clock::now();
    std::vector<double> s_clk, s_us, s_cyc[4];
    static unsigned long long h[4096][8];
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < BATCH; ++i) launch(false);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
        double clk = 0, c[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) {
            clk += (double)(h[b][3] - h[b][0]) / (double)(h[b][7] - h[b][6]) * 0.1;
            c[0] += (double)(h[b][1] - h[b][0]); c[1] += (double)(h[b][2] - h[b][1]); c[2] += (double)(h[b][3] - h[b][2]); c[3] += (double)(h[b][3] - h[b][0]);
        }
        s_clk.push_back(clk / nb); s_us.push_back(ms * 1e3 / BATCH);
        for (int i = 0; i < 4; ++i) s_cyc[i].push_back(c[i] / nb);
    }
    smp.stop(name);
    // the production instantiation right behind it (same DVFS state)
    CK(hipEventRecord(e0));
    for (int i = 0; i < BATCH; ++i) launch(true);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float msp; CK(hipEventElapsedTime(&msp, e0, e1));
    auto tail = [](const std::vector<double> &v) { double s = 0; size_t a = v.size() / 2; for (size_t i = a; i < v.size(); ++i) s += v[i]; return s / (v.size() - a); };
    const double us = tail(s_us), clk = tail(s_clk), cyc = tail(s_cyc[3]);
    const double tops = 2.0 * M * N * K / us / 1e6;
    const double floor_cyc = (double)(K / 128) * 2048.0;  // 64 MFMAs per SIMD per K-tile x 32 cycles
    printf("  gemm_i8_p8 %-8s M=%lld N=%lld K=%lld: %zu batches of %d; steady state %.2f us/launch (stamped) / %.2f (production) -> %.0f TOPS = %.1f %% of 5033\n", name, (long long)M,
           (long long)N, (long long)K, s_us.size(), BATCH, us, msp * 1e3 / BATCH, tops, tops / 50.33);
    printf("      in-kernel shader clock %.3f GHz; block = %.0f cycles (prologue %.0f + K-loop %.0f + epilogue %.0f); MFMA floor %.0f cycles = %.1f %% of the block;\n", clk, cyc, tail(s_cyc[0]),
           tail(s_cyc[1]), tail(s_cyc[2]), floor_cyc, 100.0 * floor_cyc / cyc);
    printf("      block time %.2f us (cycles / clock) vs launch-to-launch %.2f us; clock-normalised: %.0f TOPS at 2.4 GHz; first batch %.2f us @ %.3f GHz\n", cyc / clk / 1e3, us,
           tops * 2.4 / clk, s_us[0], s_clk[0]);
    printf("      clock series (GHz):");
    for (size_t i = 0; i < s_clk.size(); i += s_clk.size() / 10 + 1) printf(" %.2f", s_clk[i]);
    printf("\n");
}

// the 4-wave kernel (gemm_i8_p4) with the same stamps
template <class Epi, int PB = 1> static void run_gemm_p4(const char *name, const int8_t *x, const int8_t *w, Epi epi, int64_t M, int64_t N, int64_t K, double seconds, Sampler &smp)
{
    auto kfn = gemm_i8_p4<Epi, PB>;
    CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS_BYTES));
    const int tm = (int)((M + 255) / 256), tn = (int)((N + 255) / 256), nb = tm * tn < 4096 ? tm * tn : 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int BATCH = 50;
    auto launch = [&]() { hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(256), P4_LDS_BYTES, 0, x, w, M, N, K, tm, tn, epi); };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    smp.start();
    const auto tstart = std::chrono::steady_clock::now();
    std::vector<double> s_clk, s_us, s_cyc[4];
    static unsigned long long h[4096][8];
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < BATCH; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
        double clk = 0, c[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) {
            clk += (double)(h[b][3] - h[b][0]) / (double)(h[b][7] - h[b][6]) * 0.1;
            c[0] += (double)(h[b][1] - h[b][0]); c[1] += (double)(h[b][2] - h[b][1]); c[2] += (double)(h[b][3] - h[b][2]); c[3] += (double)(h[b][3] - h[b][0]);
        }
        s_clk.push_back(clk / nb); s_us.push_back(ms * 1e3 / BATCH);
        for (int i = 0; i < 4; ++i) s_cyc[i].push_back(c[i] / nb);
    }
    smp.stop(name);
    auto tail = [](const std::vector<double> &v) { double s = 0; size_t a = v.size() / 2; for (size_t i = a; i < v.size(); ++i) s += v[i]; return s / (v.size() - a); };
    const double us = tail(s_us), clk = tail(s_clk), cyc = tail(s_cyc[3]);
    printf("  gemm_i8_p4 %-8s M=%lld N=%lld K=%lld: %.2f us/launch -> %.0f TOPS = %.1f %% of 5033; clock %.3f GHz; block %.0f cycles (prologue %.0f + K-loop %.0f + epilogue %.0f), MFMA floor %.1f %%\n",
           name, (long long)M, (long long)N, (long long)K, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 50.33, clk, cyc, tail(s_cyc[0]), tail(s_cyc[1]), tail(s_cyc[2]),
           100.0 * (double)(K / 128) * 2048.0 / cyc);
}

// the default 256 x 256 kernel (gemm_i8_p16: p8's schedule on v_mfma_i32_16x16x64_i8) with the same stamps
template <class Epi> static void run_gemm_p16(const char *name, const int8_t *x, const int8_t *w, Epi epi, int64_t M, int64_t N, int64_t K, double seconds, Sampler &smp, OffsetArgs off = OffsetArgs{})
{
    auto kfn = gemm_i8_p16<Epi, 128>;
    CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P16_LDS_BYTES));
    const int tm = (int)((M + 255) / 256), tn = (int)((N + 255) / 256), nb = tm * tn < 4096 ? tm * tn : 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int BATCH = 50;
    auto launch = [&]() { hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), P16_LDS_BYTES, 0, x, w, M, N, K, tm, tn, epi, off); };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    smp.start();
    const auto tstart = std::chrono::steady_clock::now();
    std::vector<double> s_clk, s_us, s_cyc[4], sk_start, sk_end, sk_span, xcd_end, xcd_dur;
    static unsigned long long h[4096][8];
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < BATCH; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
        double clk = 0, c[4] = {0, 0, 0, 0};
        unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;   // the LAST launch of the batch (s_memrealtime, 10 ns ticks)
        for (int b = 0; b < nb; ++b) {
            clk += (double)(h[b][3] - h[b][0]) / (double)(h[b][7] - h[b][6]) * 0.1;
            c[0] += (double)(h[b][1] - h[b][0]); c[1] += (double)(h[b][2] - h[b][1]); c[2] += (double)(h[b][3] - h[b][2]); c[3] += (double)(h[b][3] - h[b][0]);
            s_min = h[b][6] < s_min ? h[b][6] : s_min; s_max = h[b][6] > s_max ? h[b][6] : s_max;
            e_min = h[b][7] < e_min ? h[b][7] : e_min; e_max = h[b][7] > e_max ? h[b][7] : e_max;
        }
        s_clk.push_back(clk / nb); s_us.push_back(ms * 1e3 / BATCH);
        for (int i = 0; i < 4; ++i) s_cyc[i].push_back(c[i] / nb);
        sk_start.push_back((double)(s_max - s_min) * 0.01); sk_end.push_back((double)(e_max - e_min) * 0.01); sk_span.push_back((double)(e_max - s_min) * 0.01);
        {   // per XCD (block b runs on XCD b % 8): when its last block ended, and the mean block duration on it
            unsigned long long xe[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double xd[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int xn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < nb; ++b) { const int x = b & 7; xe[x] = h[b][7] > xe[x] ? h[b][7] : xe[x]; xd[x] += (double)(h[b][7] - h[b][6]) * 0.01; ++xn[x]; }
            unsigned long long lo = ~0ull, hi = 0; double dlo = 1e30, dhi = 0;
            for (int x = 0; x < 8; ++x) { lo = xe[x] < lo ? xe[x] : lo; hi = xe[x] > hi ? xe[x] : hi; const double d = xd[x] / (xn[x] ? xn[x] : 1); dlo = d < dlo ? d : dlo; dhi = d > dhi ? d : dhi; }
            xcd_end.push_back((double)(hi - lo) * 0.01); xcd_dur.push_back(dhi / dlo);
        }
    }
    smp.stop(name);
    auto tail = [](const std::vector<double> &v) { double s = 0; size_t a = v.size() / 2; for (size_t i = a; i < v.size(); ++i) s += v[i]; return s / (v.size() - a); };
    const double us = tail(s_us), clk = tail(s_clk), cyc = tail(s_cyc[3]);
    printf("      one launch on the 100 MHz clock: first block start -> last block start %.2f us, first block end -> last block end %.2f us, first start -> last end %.2f us; "
           "launch-to-launch minus that span = %.2f us with no block running\n", tail(sk_start), tail(sk_end), tail(sk_span), us - tail(sk_span));
    printf("      per XCD (blocks b %% 8): last block of the fastest XCD -> last block of the slowest %.2f us; mean block duration slowest / fastest XCD %.3f\n", tail(xcd_end), tail(xcd_dur));
    printf("  gemm_i8_p16 %-8s M=%lld N=%lld K=%lld: %.2f us/launch -> %.0f TOPS = %.1f %% of 5033; clock %.3f GHz; block %.0f cycles (prologue %.0f + K-loop %.0f + epilogue %.0f), MFMA floor %.1f %%\n",
           name, (long long)M, (long long)N, (long long)K, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 50.33, clk, cyc, tail(s_cyc[0]), tail(s_cyc[1]), tail(s_cyc[2]),
           100.0 * (double)(K / 128) * 2048.0 / cyc);
}

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    const std::string what = argc > 2 ? argv[2] : "all";
    const int64_t N = 4096, K = 4096, M = 4096, KL = 16384;
    Rng r(1234);
    std::vector<int8_t> xb(M * KL), wb(N * KL), xu(M * K), wu(N * K), z(M * K, 0);
    gen_weights_bench(wb, N, KL, r);
    gen_acts_bench(xb, M, KL, r);
    gen_uniform(xu, r); gen_uniform(wu, r);
    printf("operand statistics:\n");
    stats("bench W", wb); stats("bench X", xb); stats("uniform", xu);
    int8_t *dxb, *dwb, *dxu, *dwu, *dz; void *o16;
    CK(hipMalloc(&dxb, xb.size())); CK(hipMalloc(&dwb, wb.size())); CK(hipMalloc(&dxu, xu.size())); CK(hipMalloc(&dwu, wu.size())); CK(hipMalloc(&dz, z.size()));
    CK(hipMalloc(&o16, M * N * 2));
    CK(hipMemcpy(dxb, xb.data(), xb.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dwb, wb.data(), wb.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dxu, xu.data(), xu.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dwu, wu.data(), wu.size(), hipMemcpyHostToDevice));
    CK(hipMemset(dz, 0, z.size()));
    EpiDequant<ASQ_F16, false, false, false> e16{o16, N, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};

    if (what == "pmc") {
        // for rocprofv3 --pmc: the PRODUCTION kernel on bench data at two K depths (a fixed per-dispatch overhead in
        // GRBM_GUI_ACTIVE shows as a different MfmaUtil at K = 16384) and an empty kernel (the overhead itself)
        auto kprod = gemm_i8_p8<decltype(e16), 0>;
        CK(hipFuncSetAttribute((const void *)kprod, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
        for (int i = 0; i < 30; ++i) {
            hipLaunchKernelGGL(kprod, dim3(256), dim3(512), P8_LDS_BYTES, 0, dxb, dwb, M, N, K, 16, 16, 1, (const int *)nullptr, 0, (char *)nullptr, e16, OffsetArgs{});
            hipLaunchKernelGGL(kprod, dim3(256), dim3(512), P8_LDS_BYTES, 0, dxb, dwb, M, N, KL, 16, 16, 1, (const int *)nullptr, 0, (char *)nullptr, e16, OffsetArgs{});
            hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, 0, (int *)nullptr);
        }
        CK(hipDeviceSynchronize());
        return 0;
    }

    if (what == "multi") {
        // Many tiles per CU (the cfg3 regime): per-CU timeline of consecutive blocks from the stamps -- how long a block takes when its
        // neighbours are out of step, and how long a CU sits between the end of one block and the start of the next.
        struct Shape { int64_t M, N; };
        for (Shape sh : {Shape{4096, 4096}, Shape{16384, 4096}, Shape{8192, 11008}}) {
            const int64_t M2 = sh.M, N2 = sh.N;
            int8_t *x2, *w2; void *o2;
            CK(hipMalloc(&x2, M2 * K)); CK(hipMalloc(&w2, N2 * K)); CK(hipMalloc(&o2, M2 * N2 * 2));
            for (int64_t off = 0; off < M2 * K; off += M * K) CK(hipMemcpy(x2 + off, dxb, std::min<int64_t>(M * K, M2 * K - off), hipMemcpyDeviceToDevice));
            for (int64_t off = 0; off < N2 * K; off += N * K) CK(hipMemcpy(w2 + off, dwb, std::min<int64_t>(N * K, N2 * K - off), hipMemcpyDeviceToDevice));
            EpiDequant<ASQ_F16, false, false, false> e2{o2, N2, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};
            auto kfn = gemm_i8_p8<decltype(e2), 128>;
            auto kprod = gemm_i8_p8<decltype(e2), 0>;
            CK(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
            CK(hipFuncSetAttribute((const void *)kprod, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES));
            const int tm = (int)(M2 / 256), tn = (int)(N2 / 256), nb = tm * tn;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto t_start = std::chrono::steady_clock::now();
            float ms = 0, msp = 0;
            const int BATCH = 20;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < seconds) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < BATCH; ++i) hipLaunchKernelGGL(kprod, dim3(nb), dim3(512), P8_LDS_BYTES, 0, x2, w2, M2, N2, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, e2, OffsetArgs{});
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&msp, e0, e1));
                CK(hipEventRecord(e0));
                for (int i = 0; i < BATCH; ++i) hipLaunchKernelGGL(kfn, dim3(nb), dim3(512), P8_LDS_BYTES, 0, x2, w2, M2, N2, K, tm, tn, 1, (const int *)nullptr, 0, (char *)nullptr, e2, OffsetArgs{});
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            static unsigned long long h[4096][8];
            CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(p8_blk), sizeof(h)));
            // group by CU (XCC + se/sh/cu bits of HW_ID), order by start time
            std::map<unsigned, std::vector<int>> per_cu;
            for (int b = 0; b < nb && b < 4096; ++b) per_cu[(unsigned)((h[b][4] & 7) | (((h[b][4] >> 16) & 0xFF) << 3))].push_back(b);
            double dur_first = 0, dur_rest = 0, gap = 0, cyc_first[3] = {0, 0, 0}, cyc_rest[3] = {0, 0, 0}; int nf = 0, nr = 0, ng = 0; double gmax = 0, span = 0;
            std::vector<double> gaps;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (auto &kv : per_cu) {
                auto &v = kv.second;
                std::sort(v.begin(), v.end(), [&](int a, int b) { return h[a][6] < h[b][6]; });
                for (size_t i = 0; i < v.size(); ++i) {
                    const int b = v[i];
                    const double d = (double)(h[b][7] - h[b][6]) * 0.01;  // us (100 MHz)
                    t0 = std::min(t0, h[b][6]); t1 = std::max(t1, h[b][7]);
                    double *c = i == 0 ? cyc_first : cyc_rest;
                    for (int j = 0; j < 3; ++j) c[j] += (double)(h[b][j + 1] - h[b][j]);
                    if (i == 0) { dur_first += d; ++nf; } else { dur_rest += d; ++nr; }
                    if (i > 0) { const double g = ((double)h[b][6] - (double)h[v[i - 1]][7]) * 0.01; gap += g; ++ng; gmax = std::max(gmax, g); gaps.push_back(g); }
                }
            }
            span = (double)(t1 - t0) * 0.01;
            std::sort(gaps.begin(), gaps.end());
            printf("  p8 M=%lld N=%lld K=%lld: %d tiles on %zu CUs (%.2f per CU); production %.1f us/launch = %.0f TOPS; stamped %.1f us; first-to-last stamp span %.1f us\n", (long long)M2, (long long)N2,
                   (long long)K, nb, per_cu.size(), (double)nb / per_cu.size(), msp * 1e3 / BATCH, 2.0 * M2 * N2 * K / (msp * 1e3 / BATCH) / 1e6, ms * 1e3 / BATCH, span);
            printf("      block wall time: first on its CU %.2f us (%d), later ones %.2f us (%d); cycles prologue/K-loop/epilogue first %.0f/%.0f/%.0f later %.0f/%.0f/%.0f\n", nf ? dur_first / nf : 0, nf,
                   nr ? dur_rest / nr : 0, nr, nf ? cyc_first[0] / nf : 0, nf ? cyc_first[1] / nf : 0, nf ? cyc_first[2] / nf : 0, nr ? cyc_rest[0] / nr : 0, nr ? cyc_rest[1] / nr : 0, nr ? cyc_rest[2] / nr : 0);
            if (ng) printf("      CU idle between consecutive blocks: mean %.2f us, median %.2f, p90 %.2f, max %.2f (%d gaps)\n", gap / ng, gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10], gmax, ng);
            CK(hipFree(x2)); CK(hipFree(w2)); CK(hipFree(o2));
        }
        return 0;
    }
    Sampler smp;
    if (what == "all" || what == "mfma") {
        printf("== MFMA-only (v_mfma_i32_32x32x32_i8, 256 blocks, operands static in registers) ==\n");
        v4i *ab_b = mf_operands(wb, xb), *ab_u = mf_operands(wu, xu), *ab_z = mf_operands(z, z);
        RUN_MF(1, 0, "r1", "zeros", ab_z, 8);
        RUN_MF(0, 0, "distinct", "zeros", ab_z, 8);
        RUN_MF(0, 0, "distinct", "bench", ab_b, 8);
        RUN_MF(1, 0, "r1", "bench", ab_b, 8);
        RUN_MF(2, 0, "p8", "bench", ab_b, 8);
        RUN_MF(3, 0, "snake", "bench", ab_b, 8);
        {   // roles swapped: activations as the matrix-core A operand, weights as B (the GEMM uses W = A, X = B)
            v4i *ab_s = mf_operands(xb, wb);
            RUN_MF(2, 0, "p8", "bench-sw", ab_s, 8);
            RUN_MF(0, 0, "distinct", "bench-sw", ab_s, 8);
        }
        {   // does the SIGN toggling of the small activations cost matrix-core energy?  X + 3 (mostly 0..6, all high bits zero) vs X; |X| (sign removed)
            std::vector<int8_t> xo(xb), xa(xb);
            for (auto &v : xo) { int t = (int)v + 3; v = (int8_t)(t > 127 ? 127 : t); }
            for (auto &v : xa) v = (int8_t)(v < 0 ? (v == -128 ? 127 : -v) : v);
            v4i *ab_o = mf_operands(wb, xo), *ab_a = mf_operands(wb, xa);
            RUN_MF(2, 0, "p8", "bench", ab_b, 8);
            RUN_MF(2, 0, "p8", "bench X+3", ab_o, 8);
            RUN_MF(2, 0, "p8", "bench |X|", ab_a, 8);
            std::vector<int8_t> wa2(wb);
            for (auto &v : wa2) v = (int8_t)(v < 0 ? (v == -128 ? 127 : -v) : v);
            v4i *ab_w = mf_operands(wa2, xb), *ab_ww = mf_operands(wa2, xa);
            RUN_MF(2, 0, "p8", "bench |W|", ab_w, 8);
            RUN_MF(2, 0, "p8", "|W|,|X|", ab_ww, 8);
        }
        RUN_MF(0, 0, "distinct", "uniform", ab_u, 8);
        RUN_MF(1, 0, "r1", "uniform", ab_u, 8);
        RUN_MF(2, 0, "p8", "uniform", ab_u, 8);
        RUN_MF(3, 0, "snake", "uniform", ab_u, 8);
        printf("  -- duty sweep, one wave per SIMD, snake order --\n");
        RUN_MF(3, 0, "snake", "bench", ab_b, 4);
        RUN_MF(3, 1, "snake", "bench", ab_b, 4);
        RUN_MF(3, 2, "snake", "bench", ab_b, 4);
        RUN_MF(3, 4, "snake", "bench", ab_b, 4);
        RUN_MF(3, 8, "snake", "bench", ab_b, 4);
        RUN_MF(1, 0, "r1", "bench", ab_b, 4);
        RUN_MF(0, 0, "distinct", "bench", ab_b, 4);
        RUN_MF(3, 0, "snake", "uniform", ab_u, 4);
        RUN_MF(3, 2, "snake", "uniform", ab_u, 4);
        RUN_MF(3, 4, "snake", "uniform", ab_u, 4);
    }
    if (what == "all" || what == "gemm") {
        printf("== gemm_i8_p8 (production schedule + stamps), f16 epilogue ==\n");
        run_gemm("bench", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 256>("bench nt", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 512>("bench sc1", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 768>("bench wt", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm("bench", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm("uniform", dxu, dwu, e16, M, N, K, seconds, smp);
        run_gemm("zeros", dz, dz, e16, M, N, K, seconds, smp);
        run_gemm("benchK16k", dxb, dwb, e16, M, N, KL, seconds, smp);
    }
    if (what == "p4") {
        run_gemm("p8 bench", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4("bench", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4<decltype(e16), 3>("noDMA", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4<decltype(e16), 5>("noSync", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4<decltype(e16), 9>("noDSRD", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4<decltype(e16), 7>("noDMA+Sy", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4<decltype(e16), 15>("MFMAonly", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4("zeros", dz, dz, e16, M, N, K, seconds, smp);
        run_gemm_p4("benchK16k", dxb, dwb, e16, M, N, KL, seconds, smp);
        run_gemm("p8 bench", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm_p4("bench", dxb, dwb, e16, M, N, K, seconds, smp);
    }
    if (what == "off") {   // round 4: offset operand images (asq_gemm_kernels.h: OffsetArgs) -- where do the cycles and the clock go?
        std::vector<int8_t> wi(N * K), xi(M * K);
        std::vector<int32_t> col(2 * N), row(2 * M), zc(2 * N, 0), zr(2 * M, 0);
        for (int64_t n = 0; n < N; ++n) {
            int mx = -128; long long sm = 0;
            for (int64_t k = 0; k < K; ++k) { const int v = wb[n * K + k]; mx = v > mx ? v : mx; sm += v; }
            const int cw = std::min(64, 127 - mx);
            col[2 * n] = cw; col[2 * n + 1] = (int32_t)sm;
            for (int64_t k = 0; k < K; ++k) wi[n * K + k] = (int8_t)(wb[n * K + k] + cw);
        }
        for (int64_t m = 0; m < M; ++m) {
            int mx = -128, mn = 127; long long sm = 0;
            for (int64_t k = 0; k < K; ++k) { const int v = xb[m * K + k]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; sm += v; }
            const int cx = mx <= 124 ? 3 : (mn >= -125 ? -3 : 0);
            row[2 * m] = cx; row[2 * m + 1] = (int32_t)(sm + cx * K);
            for (int64_t k = 0; k < K; ++k) xi[m * K + k] = (int8_t)(xb[m * K + k] + cx);
        }
        int8_t *dwi, *dxi; int32_t *dcol, *drow, *dzc, *dzr;
        CK(hipMalloc(&dwi, N * K)); CK(hipMalloc(&dxi, M * K)); CK(hipMalloc(&dcol, 8 * N)); CK(hipMalloc(&drow, 8 * M)); CK(hipMalloc(&dzc, 8 * N)); CK(hipMalloc(&dzr, 8 * M));
        CK(hipMemcpy(dwi, wi.data(), N * K, hipMemcpyHostToDevice)); CK(hipMemcpy(dxi, xi.data(), M * K, hipMemcpyHostToDevice));
        CK(hipMemcpy(dcol, col.data(), 8 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(drow, row.data(), 8 * M, hipMemcpyHostToDevice));
        CK(hipMemset(dzc, 0, 8 * N)); CK(hipMemset(dzr, 0, 8 * M));
        // (the first M * K / N * K bytes of the K = 16384 buffers are what run_gemm_p16 multiplies at K = 4096: rows contiguous at the K it is given)
        std::vector<int8_t> w4(wb.begin(), wb.begin() + N * K), x4(xb.begin(), xb.begin() + M * K);
        CK(hipMemcpy(dwb, w4.data(), N * K, hipMemcpyHostToDevice)); CK(hipMemcpy(dxb, x4.data(), M * K, hipMemcpyHostToDevice));
        Sampler smp;
        for (int rep = 0; rep < 2; ++rep) {
            run_gemm_p16("plain", dxb, dwb, e16, M, N, K, seconds, smp);
            run_gemm_p16("images, plain launch (timing only)", dxi, dwi, e16, M, N, K, seconds, smp);
            run_gemm_p16("images + start values (exact)", dxi, dwi, e16, M, N, K, seconds, smp, OffsetArgs{drow, dcol});
            run_gemm_p16("images + zero vectors (timing only)", dxi, dwi, e16, M, N, K, seconds, smp, OffsetArgs{dzr, dzc});
            run_gemm_p16("plain operands + zero vectors", dxb, dwb, e16, M, N, K, seconds, smp, OffsetArgs{dzr, dzc});
        }
        return 0;
    }

    if (what == "p16") {   // round 3: the 16x16x64 kernel next to the 32x32x32 one, stamped, alternating in one process
        for (int rep = 0; rep < 2; ++rep) {
            run_gemm("p8 bench", dxb, dwb, e16, M, N, K, seconds, smp);
            run_gemm_p16("bench", dxb, dwb, e16, M, N, K, seconds, smp);
        }
        run_gemm("p8 unif", dxu, dwu, e16, M, N, K, seconds, smp);
        run_gemm_p16("uniform", dxu, dwu, e16, M, N, K, seconds, smp);
        run_gemm_p16("zeros", dz, dz, e16, M, N, K, seconds, smp);
        run_gemm("p8 K16k", dxb, dwb, e16, M, N, KL, seconds, smp);
        run_gemm_p16("benchK16k", dxb, dwb, e16, M, N, KL, seconds, smp);
        {   // 1024 tiles = 4 rounds of 256 (M = 16384: the activation rows four times over): how unevenly the statically assigned XCDs finish a many-round launch
            const int64_t M2 = 4 * M;
            int8_t *x2; void *o2;
            CK(hipMalloc(&x2, M2 * K)); CK(hipMalloc(&o2, M2 * N * 2));
            for (int64_t off = 0; off < M2 * K; off += M * K) CK(hipMemcpy(x2 + off, dxb, M * K, hipMemcpyDeviceToDevice));
            EpiDequant<ASQ_F16, false, false, false> e2{o2, N, nullptr, nullptr, nullptr, nullptr, 1e-4f, 0, true};
            run_gemm_p16("4 rounds", x2, dwb, e2, M2, N, K, seconds, smp);
            CK(hipFree(x2)); CK(hipFree(o2));
        }
    }
    if (what == "epi") {   // round 3: the pipelined interior epilogue (epilogue_wave_rows); p4 against its round-2 epilogue in one process (for p8 run the previous build's "gemm" arm beside this)
        for (int rep = 0; rep < 2; ++rep) {
            run_gemm("p8 rows", dxb, dwb, e16, M, N, K, seconds, smp);
            run_gemm_p4<decltype(e16), 17>("r2epi", dxb, dwb, e16, M, N, K, seconds, smp);
            run_gemm_p4("rows", dxb, dwb, e16, M, N, K, seconds, smp);
        }
        run_gemm("u rows", dxu, dwu, e16, M, N, K, seconds, smp);
        run_gemm_p4("u rows", dxu, dwu, e16, M, N, K, seconds, smp);
    }
    if (what == "all" || what == "abl") {
        printf("== gemm_i8_p8 ablations on bench data: what each part of the schedule costs in time, clock and power (results invalid) ==\n");
        run_gemm("full", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 4>("noMFMA", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 6>("DMAonly", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 5>("dsRDonly", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 7>("skeleton", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 2>("no dsRD", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 1>("no DMA", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 3>("MFMAonly", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm<decltype(e16), 16>("2x dsRD", dxb, dwb, e16, M, N, K, seconds, smp);
        run_gemm("full", dxb, dwb, e16, M, N, K, seconds, smp);
    }
    return 0;
}
