from .linear import (Int8GEMM, W8A8BFP32OFP32Linear, W8A8BFP32OFP32QKVLinear,  # noqa: F401
                     W8A8BFP32OFP32LinearWithQuantScale)
