// fp-epilogue instantiations of the INT8 GEMM kernels for out dtype F32 (own TU: parallel build; the per-token variants are in the _row TU)
#include "asq_gemm_kernels.h"
namespace asq {
template <> int launch_dequant_half<ASQ_F32, false>(const DequantArgs &a, hipStream_t s) { return launch_dequant_half_impl<ASQ_F32, false>(a, s); }
template <> int launch_dequant<ASQ_F32>(const DequantArgs &a, hipStream_t s) { return launch_dequant_impl<ASQ_F32>(a, s); }
template <> int launch_dequant_q<ASQ_F32>(const DequantQArgs &a, hipStream_t s) { return launch_dequant_q_impl<ASQ_F32>(a, s); }
}  // namespace asq
