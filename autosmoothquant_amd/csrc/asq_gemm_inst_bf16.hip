// fp-epilogue instantiations of the INT8 GEMM kernels for out dtype BF16 (own TU: parallel build; the per-token variants are in the _row TU)
#include "asq_gemm_kernels.h"
namespace asq {
template <> int launch_dequant_half<ASQ_BF16, false>(const DequantArgs &a, hipStream_t s) { return launch_dequant_half_impl<ASQ_BF16, false>(a, s); }
template <> int launch_dequant<ASQ_BF16>(const DequantArgs &a, hipStream_t s) { return launch_dequant_impl<ASQ_BF16>(a, s); }
template <> int launch_dequant_q<ASQ_BF16>(const DequantQArgs &a, hipStream_t s) { return launch_dequant_q_impl<ASQ_BF16>(a, s); }
}  // namespace asq
