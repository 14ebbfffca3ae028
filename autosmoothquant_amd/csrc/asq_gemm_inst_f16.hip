// fp-epilogue instantiations of the INT8 GEMM kernels for out dtype F16 (own TU: parallel build; the per-token variants are in the _row TU)
#include "asq_gemm_kernels.h"
namespace asq {
template <> int launch_dequant_half<ASQ_F16, false>(const DequantArgs &a, hipStream_t s) { return launch_dequant_half_impl<ASQ_F16, false>(a, s); }
template <> int launch_dequant<ASQ_F16>(const DequantArgs &a, hipStream_t s) { return launch_dequant_impl<ASQ_F16>(a, s); }
template <> int launch_dequant_q<ASQ_F16>(const DequantQArgs &a, hipStream_t s) { return launch_dequant_q_impl<ASQ_F16>(a, s); }
}  // namespace asq
