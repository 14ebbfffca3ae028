// v_cvt_pk_bf16_f32 (+ the NaN patch of asq_common.h::f32x2_to_bf16x2_bits) against the software round-to-nearest-even with canonical NaN, all 2^32 inputs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value bf16_cvt_check.hip -o bf16_cvt_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../autosmoothquant_amd/csrc/asq_common.h"
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t sw(float f) { uint32_t u = __float_as_uint(f); if (f != f) return 0x7FC0; u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
__global__ void k(unsigned long long *bad, unsigned *ex)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, n = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (unsigned long long i = tid; i < (1ull << 32); i += n) {
        v2f v = {__uint_as_float((unsigned)i), __uint_as_float(~(unsigned)i)};
        asm("" : "+v"(v));
        const uint32_t bits = f32x2_to_bf16x2_bits(v[0], v[1]);
        if ((bits & 0xFFFF) != sw(v[0]) || (bits >> 16) != sw(v[1])) { ++b; if (b == 1) { ex[0] = (unsigned)i; ex[1] = bits; } }
    }
    atomicAdd(bad, b);
}
int main() { unsigned long long *bad; unsigned *ex; hipMalloc(&bad, 8); hipMalloc(&ex, 8); hipMemset(bad, 0, 8); hipMemset(ex, 0, 8);
  k<<<4096, 256>>>(bad, ex); unsigned long long hb; unsigned he[2]; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 8, hipMemcpyDeviceToHost);
  printf("f32x2_to_bf16x2_bits vs software RNE + canonical NaN over 2^32 inputs (both halves): %llu mismatches (first: in %08x -> %08x)\n", hb, he[0], he[1]); return 0; }
