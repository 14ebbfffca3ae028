/* Plain-C restatement of the reference's native GEMM entry point
 *   I8CUGEMM::linear_a8_w8_o32_  (/root/reference csrc/int8gemm/bindings.cpp:69-84)
 *   -> cublasINT8MMWrapper::Gemm_(int*...) (csrc/int8gemm/cublasINT8MMWrapper.cc:224-354)
 * i.e. out[M,N] (int32) = x[M,K] (int8) . w[N,K]^T (int8), alpha=1, beta=0, exact.
 *
 * TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Used as the checker for the
 * HIP kernels and as the "port" CPU baseline in bench.py; never on the product path.
 */
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* one output row-block: rows [m0,m1) x all N.  Multi-versioned so the same .so
 * uses AVX-512/AVX2 when the host has it (the build box and the GPU box differ). */
__attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
static void igemm_rows(const int8_t *x, const int8_t *w, int32_t *out,
                       long m0, long m1, long N, long K)
{
    for (long m = m0; m < m1; ++m) {
        const int8_t *xr = x + m * K;
        for (long n = 0; n < N; ++n) {
            const int8_t *wr = w + n * K;
            int32_t acc = 0;
            for (long k = 0; k < K; ++k)
                acc += (int32_t)xr[k] * (int32_t)wr[k];
            out[m * N + n] = acc;
        }
    }
}

/* column-block variant so skinny-M problems still use every thread */
__attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
static void igemm_cols(const int8_t *x, const int8_t *w, int32_t *out,
                       long M, long n0, long n1, long N, long K)
{
    for (long n = n0; n < n1; ++n) {
        const int8_t *wr = w + n * K;
        for (long m = 0; m < M; ++m) {
            const int8_t *xr = x + m * K;
            int32_t acc = 0;
            for (long k = 0; k < K; ++k)
                acc += (int32_t)xr[k] * (int32_t)wr[k];
            out[m * N + n] = acc;
        }
    }
}

void asq_oracle_igemm(const int8_t *x, const int8_t *w, int32_t *out,
                      long M, long N, long K, int threads)
{
    if (M <= 0 || N <= 0) return;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (M >= 64) {
        const long RB = 4;
        const long nb = (M + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
        for (long b = 0; b < nb; ++b) {
            long m0 = b * RB, m1 = m0 + RB > M ? M : m0 + RB;
            igemm_rows(x, w, out, m0, m1, N, K);
        }
    } else {
        const long CB = 16;
        const long nb = (N + CB - 1) / CB;
#pragma omp parallel for schedule(dynamic, 1)
        for (long b = 0; b < nb; ++b) {
            long n0 = b * CB, n1 = n0 + CB > N ? N : n0 + CB;
            igemm_cols(x, w, out, M, n0, n1, N, K);
        }
    }
}

int asq_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
