// fp-epilogue instantiations of the INT8 GEMM kernels for out dtype BF16, per-token (row scale) variants (own TU: parallel build)
#include "asq_gemm_kernels.h"
namespace asq {
template <> int launch_dequant_half<ASQ_BF16, true>(const DequantArgs &a, hipStream_t s) { return launch_dequant_half_impl<ASQ_BF16, true>(a, s); }
}  // namespace asq
