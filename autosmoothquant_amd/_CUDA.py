"""Drop-in for the reference's native module ``autosmoothquant._CUDA`` (pybind11 class
``I8CUGEMM``, csrc/int8gemm/bindings.cpp:145-155): same class name, ctor and the five
method names / argument orders, implemented over libasq_hip.so.

Differences that are deliberate:
  * stateless: no cuBLASLt handle, no process-wide mutex, and the CURRENT torch stream is
    used per call (the reference captures one stream at construction, bindings.cpp:13);
  * arguments are validated (dtype / device / shape / contiguity) instead of passing raw
    data_ptr blindly (bindings.cpp:73-80);
  * ``linear_a8_w8_o32`` / ``linear_a8_w8_o8`` take plain row-major operands -- cuBLASLt's
    COL32 / COL32_2R_4R4 layouts (cublasINT8MMWrapper.cc:70-108) do not exist on gfx950.
"""
import torch

from . import ops


class I8CUGEMM:
    def __init__(self):
        pass

    # out[M,N] int32 = input[M,K] . weight[N,K]^T
    def linear_a8_w8_o32(self, input, weight, out):
        ops.gemm_i8_i32(input, weight, out)

    def linear_a8_w8_o32_(self, input, weight, out):
        ops.gemm_i8_i32(input, weight, out)

    # out[M,N] int8 = sat(alpha * acc)
    def linear_a8_w8_o8(self, input, weight, out, alpha):
        ops.gemm_i8_i8(input, weight, out, alpha, 0.0)

    # out[M,N] int8 = sat(alpha * acc + beta * out)
    def linear_a8_w8_o8_(self, input, weight, out, alpha, beta=0.0):
        ops.gemm_i8_i8(input, weight, out, alpha, beta)

    # returns int8 [M,N] = sat(alpha * acc + beta * bias[n])   (bindings.cpp:123-142)
    def linear_a8_w8_b8_o8_(self, input, weight, bias, alpha, beta):
        out = bias.view(1, -1).repeat(input.shape[0], 1)
        ops.gemm_i8_i8(input, weight, out, alpha, beta)
        return out
