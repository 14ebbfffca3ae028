"""autosmoothquant_amd -- MI355X (gfx950) implementation of AutoSmoothQuant's W8A8 linear
hot path behind the reference's own module / operator API.

    from autosmoothquant_amd._CUDA import I8CUGEMM                    # low boundary
    from autosmoothquant_amd.layers.nn.linear import (                # high boundary
        W8A8BFP32OFP32Linear, W8A8BFP32OFP32QKVLinear, W8A8BFP32OFP32LinearWithQuantScale)

All compute runs in hand-written HIP kernels inside ``libasq_hip.so`` (C-ABI in
``include/asq_hip.h``).  There is no CPU or eager-PyTorch fallback: importing works
anywhere, but calling an op without the library or on a non-HIP tensor raises.
"""
__version__ = "0.1.0"
