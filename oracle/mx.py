"""Oracle for the MX (OCP Microscaling Formats v1.0, MXFP8 E4M3) opt-in of the fp8 path -- TEST INFRASTRUCTURE, not product code.

The reference has no block-scaled mode (its fp8 classes use one scale per tensor or per token, layers/nn/linear.py:336-644,
layers/functional/quantization.py:144-211); SURVEY 8f N4 lists "MX-scaled fp8" as an accuracy-checked extension.  This file restates the
OCP spec's conversion (section 6.3: shared exponent = floor(log2(max |v|)) - emax_elem, elements = saturating round-to-nearest-even of
v / 2^shared_exp) and the block-scaled dot product, so the HIP kernels (asq_quantize_mxfp8, asq_linear_mxfp8) can be checked: the quantiser
bit for bit, the GEMM within the fp32-accumulation tolerance used for the other fp8 linears.  PARITY UNPINNED against the reference (it has no
counterpart to pin to): the anchor is the published spec, through the hand-derived known answers of tests/test_mx_oracle_cpu.py."""
import numpy as np

from . import fp8 as F8
from .w8a8 import round_to

BLOCK = 32
E4M3_EMAX = 8


def mx_quantize_e4m3(x):
    """x [M, K] (K % 32 == 0) -> (codes uint8 [M, K], scales uint8 [M, K/32]); scale byte = E8M0 (value 2^(byte - 127), 0xFF = NaN)."""
    x = np.asarray(x, np.float32)
    M, K = x.shape
    assert K % BLOCK == 0
    xb = x.reshape(M, K // BLOCK, BLOCK)
    amax_bits = np.max(xb.view(np.uint32) & np.uint32(0x7FFFFFFF), axis=-1)
    e = (amax_bits >> np.uint32(23)).astype(np.int64) - E4M3_EMAX          # biased exponent of 2^(floor(log2 amax) - 8)
    e = np.clip(e, 0, 254)
    scales = np.where(amax_bits >= np.uint32(0x7F800000), 255, e).astype(np.uint8)
    inv = np.ldexp(np.float32(1.0), (127 - e).astype(np.int32)).astype(np.float32)   # 2^-(e - 127): exact, a subnormal for e = 254
    with np.errstate(over="ignore", invalid="ignore"):
        v = (xb * inv[..., None]).astype(np.float32)
        v = np.where(np.isnan(v), v, np.clip(v, -448.0, 448.0)).astype(np.float32)
    return F8.f32_to_e4m3fn(v.reshape(M, K)), scales


def mx_dequantize(codes, scales):
    codes = np.asarray(codes, np.uint8)
    M, K = codes.shape
    v = F8.e4m3fn_to_f32(codes).astype(np.float64).reshape(M, K // BLOCK, BLOCK)
    s = np.where(scales == 255, np.nan, np.ldexp(1.0, scales.astype(np.int32) - 127))
    return (v * s[..., None]).reshape(M, K)


def mx_linear(xq, xs, wq, ws, bias=None, dt="f32"):
    """the exact block-scaled product (float64), + bias, rounded to dt"""
    y = mx_dequantize(xq, xs) @ mx_dequantize(wq, ws).T
    if bias is not None:
        y = y + np.asarray(bias, np.float64)[None, :]
    return round_to(y.astype(np.float32), dt)
