"""NumPy restatement of the norm -> int8 / SiLU*up -> int8 fusions.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

What the reference owns of this arithmetic (all dead or host-side code there):
  * ``LayerNormQ.forward``                  autosmoothquant/layers/nn/fused.py:10-15
  * ``dq_add_layernorm_q``                  csrc/kernels/fused.cu:5-25, layers/functional/fused.py:5-12
  * scale-folded RMSNorm + per-tensor round  models/baichuan.py:49-59 + thirdparty/.../modeling_baichuan.py:161-176
                                             (LLaMA: models/llama.py:27-37) followed by layers/nn/linear.py:95-96
  * SiLU(gate) * up feeding a WithQuantScale linear   models/llama.py:206-211 / HF LlamaMLP, linear.py:283-292

Two kinds of functions live here:

``*_kernel_order``  repeat the HIP kernels' fp32 operation sequence exactly (csrc/asq_quant.hip:
    ``norm_quant_cached`` / ``silu_mul_quant_cached``): per-thread accumulation order (thread t of 256 owns the 16-byte
    vectors t, t+256, ... of the row and adds their elements in order), 64-lane butterfly, ``(r0+r1)+(r2+r3)``
    across the four waves, ``1/sqrt`` from one IEEE sqrt and one IEEE division, and ``exp_det`` -- a fixed sequence
    of fp32 multiplies and adds.  The GPU tests compare with these BIT FOR BIT.

``*_reference``     the reference's formulas with float64 statistics (no particular summation order).  The kernel-order
    result may differ from them -- and from the reference's ATen execution captured in tests/golden/g8_n1.npz -- only
    where the normalised value lies within a few fp32 ulps of a rounding boundary of the int8 grid; the CPU tests
    assert exactly that (every mismatch is +-1 and sits on such a boundary).

Parity status: the int8 quantisation steps are the pinned ``oracle.w8a8`` quantisers; the float parts are pinned to
g8 through the boundary argument above.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import w8a8 as O

F32 = np.float32


def _vec(dt: str) -> int:
    return 4 if dt == O.F32 else 8


# --------------------------------------------------------------------------
# fixed-order fp32 primitives (mirror csrc/asq_quant.hip)
# --------------------------------------------------------------------------
def block_sum_256(v: np.ndarray, dt: str) -> np.ndarray:
    """v [M, K] fp32 -> [M] fp32, summed like block_sum_256 over a 256-thread block whose thread t holds the
    VEC-element vectors t, t+256, ... of the row (K % VEC == 0)."""
    v = np.asarray(v, dtype=F32)
    M, K = v.shape
    vec = _vec(dt)
    assert K % vec == 0
    nvec = K // vec
    nv = (nvec + 255) // 256
    pad = np.zeros((M, nv * 256 * vec), dtype=F32)
    pad[:, :K] = v
    pad = pad.reshape(M, nv, 256, vec)
    valid = (np.arange(nv)[:, None] * 256 + np.arange(256)[None, :]) < nvec  # [nv, 256]
    acc = np.zeros((M, 256), dtype=F32)
    for i in range(nv):
        for j in range(vec):
            acc = np.where(valid[i][None, :], acc + pad[:, i, :, j], acc).astype(F32)
    acc = acc.reshape(M, 4, 64)
    lanes = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        acc = (acc + acc[:, :, lanes ^ off]).astype(F32)
    red = acc[:, :, 0]
    return ((red[:, 0] + red[:, 1]).astype(F32) + (red[:, 2] + red[:, 3]).astype(F32)).astype(F32)


def rsqrt_exact(v: np.ndarray) -> np.ndarray:
    return (F32(1.0) / np.sqrt(np.asarray(v, dtype=F32))).astype(F32)


def exp_det(x: np.ndarray) -> np.ndarray:
    """csrc/asq_quant.hip::exp_det, operation by operation."""
    x = np.asarray(x, dtype=F32)
    nan = np.isnan(x)
    xc = np.minimum(np.maximum(np.where(nan, F32(0), x), F32(-104.0)), F32(89.0)).astype(F32)
    n = np.rint((xc * F32(1.44269502162933349609375)).astype(F32)).astype(F32)
    r = (xc + (-(n * F32(0.693138122558593750)).astype(F32))).astype(F32)
    r = (r + (-(n * F32(9.05800061445916071534156799316e-06)).astype(F32))).astype(F32)
    p = np.full_like(r, F32(1.0) / F32(5040.0))
    for c in (F32(1.0) / F32(720.0), F32(1.0) / F32(120.0), F32(1.0) / F32(24.0), F32(1.0) / F32(6.0), F32(0.5), F32(1.0), F32(1.0)):
        p = ((p * r).astype(F32) + c).astype(F32)
    ni = n.astype(np.int32)
    n1 = ni >> 1
    n2 = ni - n1

    def pow2i(k):
        return ((k + 127).astype(np.int32) << 23).view(F32)
    with np.errstate(over="ignore", under="ignore"):
        out = ((p * pow2i(n1)).astype(F32) * pow2i(n2)).astype(F32)
    return np.where(nan, x, out).astype(F32)


# --------------------------------------------------------------------------
# kernel-order restatements
# --------------------------------------------------------------------------
def _norm_y_kernel_order(h: np.ndarray, dt: str, weight: np.ndarray, bias: Optional[np.ndarray], eps: float) -> np.ndarray:
    h = np.asarray(h, dtype=F32)
    K = h.shape[1]
    w = np.asarray(weight, dtype=F32)[None, :]
    if bias is not None:  # LayerNorm: two-pass variance
        mean = (block_sum_256(h, dt) / F32(K)).astype(F32)[:, None]
        d = (h + (-mean)).astype(F32)
        var = (block_sum_256((d * d).astype(F32), dt) / F32(K)).astype(F32)
        rs = rsqrt_exact((var + F32(eps)).astype(F32))[:, None]
        y = (((d * rs).astype(F32) * w).astype(F32) + np.asarray(bias, dtype=F32)[None, :]).astype(F32)
        return O.round_to(y, dt)
    var = (block_sum_256((h * h).astype(F32), dt) / F32(K)).astype(F32)
    rs = rsqrt_exact((var + F32(eps)).astype(F32))[:, None]
    n = O.round_to((h * rs).astype(F32), dt)          # hidden_states.to(input_dtype)
    return O.round_to((w * n).astype(F32), dt)        # weight * hidden_states


def _quantise(y: np.ndarray, dt: str, per_token: bool) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    if per_token:
        xq, qs = O.act_quant_per_token(y, dt)
        return xq, qs
    return O.act_quant_round(y, dt), None


def norm_quant_kernel_order(x, dt: str, weight, bias=None, eps: float = 1e-5, per_token: bool = False):
    """asq_norm_quantize: (xq int8 [M,K], s_row f32 [M] | None)."""
    return _quantise(_norm_y_kernel_order(x, dt, weight, bias, eps), dt, per_token)


def add_norm_quant_kernel_order(x, residual, dt: str, weight, bias=None, eps: float = 1e-5, per_token: bool = False):
    """asq_add_norm_quantize: (h [M,K] in dt, xq, s_row | None) with h = dt(residual + x)."""
    h = O.round_to((np.asarray(residual, dtype=F32) + np.asarray(x, dtype=F32)).astype(F32), dt)
    xq, s = _quantise(_norm_y_kernel_order(h, dt, weight, bias, eps), dt, per_token)
    return h, xq, s


def silu_mul_quant_kernel_order(gate, up, dt: str, per_token: bool = True, quant_scale: float = 1.0):
    """asq_silu_mul_quantize: a = dt(dt(g / (1 + exp_det(-g))) * u), then the consumer's quantiser."""
    g, u = np.asarray(gate, dtype=F32), np.asarray(up, dtype=F32)
    with np.errstate(over="ignore", invalid="ignore"):
        sl = O.round_to((g / (F32(1.0) + exp_det(-g)).astype(F32)).astype(F32), dt)
        a = O.round_to((sl * u).astype(F32), dt)
    if per_token:
        return O.act_quant_per_token(a, dt)
    return O.act_quant_div(a, dt, quant_scale), None


def rmsnorm_kernel_order(x, dt: str, weight, eps: float = 1e-5) -> np.ndarray:
    """asq_rmsnorm: y = dt(w * dt(x * rsqrt(mean(x^2) + eps))) with the block kernel's summation order (the fp-out twin of norm_quant_kernel_order)."""
    return _norm_y_kernel_order(x, dt, weight, None, eps)


def silu_mul_kernel_order(gate, up, dt: str) -> np.ndarray:
    """asq_silu_mul (exact form): a = dt(dt(g / (1 + exp_det(-g))) * u) -- silu_mul_quant_kernel_order's activation without the quantiser."""
    g, u = np.asarray(gate, dtype=F32), np.asarray(up, dtype=F32)
    with np.errstate(over="ignore", invalid="ignore"):
        sl = O.round_to((g / (F32(1.0) + exp_det(-g)).astype(F32)).astype(F32), dt)
        return O.round_to((sl * u).astype(F32), dt)


def rope_kernel_order(x, cos, sin, dt: str) -> np.ndarray:
    """asq_rope: x [B, S, H, D], cos / sin [S, D/2] (values of dtype dt), rotate_half convention, positions 0 .. S-1.
        out[.., :D/2] = dt(dt(x1 * cos) - x2 * sin)      out[.., D/2:] = dt(dt(x2 * cos) + x1 * sin)
    fp16: the sum is rounded ONCE to fp16 (the kernel's v_fma_mixlo/hi_f16; the product of two fp16 values and its sum with an fp16 value are exact in float64);
    bf16 / fp32: product and sum each rounded to fp32, then the result to dt (the kernel's fp32 path)."""
    x = np.asarray(x, dtype=F32)
    B, S, H, D = x.shape
    c = np.asarray(cos, dtype=F32).reshape(1, S, 1, D // 2)
    sn = np.asarray(sin, dtype=F32).reshape(1, S, 1, D // 2)
    x1, x2 = x[..., : D // 2], x[..., D // 2:]
    t1, t2 = O.round_to((x1 * c).astype(F32), dt), O.round_to((x2 * c).astype(F32), dt)
    out = np.empty_like(x)
    if dt == O.F16:
        out[..., : D // 2] = (t1.astype(np.float64) - x2.astype(np.float64) * sn.astype(np.float64)).astype(np.float16).astype(F32)
        out[..., D // 2:] = (t2.astype(np.float64) + x1.astype(np.float64) * sn.astype(np.float64)).astype(np.float16).astype(F32)
    else:
        out[..., : D // 2] = O.round_to((t1 + ((-x2) * sn).astype(F32)).astype(F32), dt)
        out[..., D // 2:] = O.round_to((t2 + (x1 * sn).astype(F32)).astype(F32), dt)
    return out


def silu_mul_quant_fp8_kernel_order(gate, up, dt: str):
    """asq_silu_mul_quantize_fp8 (exact form): a = dt(dt(g / (1 + exp_det(-g))) * u) -- silu_mul_quant_kernel_order's activation -- then the reference's
    per_token_quantize_fp8 (layers/functional/quantization.py:173-191, restated and G5-pinned in oracle/fp8.py).  Returns (e4m3fn bytes [M,K], scale f32 [M,1])."""
    from . import fp8 as F8
    g, u = np.asarray(gate, dtype=F32), np.asarray(up, dtype=F32)
    with np.errstate(over="ignore", invalid="ignore"):
        sl = O.round_to((g / (F32(1.0) + exp_det(-g)).astype(F32)).astype(F32), dt)
        a = O.round_to((sl * u).astype(F32), dt)
    return F8.per_token_quantize_fp8(a, dt)


def gate_up_silu_kernel_order(xq, w_gate, w_up, dt: str, s_gate: float, s_up: float, s_row=None) -> np.ndarray:
    """asq_linear_w8a8_gate_up with the fixed-operation-order SiLU (flags = 0): the two projections' dequantised outputs in `dt` -- exactly the epilogue of
    W8A8BFP32OFP32Linear (oracle/w8a8.py::dequant_epilogue, reference layers/nn/linear.py:93-104) -- then a = dt(dt(g / (1 + exp_det(-g))) * u)
    (silu_mul_quant_kernel_order above; reference models/llama.py:206-211 computes the same expression with torch's silu)."""
    g = O.dequant_epilogue(O.igemm(xq, w_gate), F32(s_gate), s_row, None, dt)
    u = O.dequant_epilogue(O.igemm(xq, w_up), F32(s_up), s_row, None, dt)
    with np.errstate(over="ignore", invalid="ignore"):
        sl = O.round_to((g / (F32(1.0) + exp_det(-g)).astype(F32)).astype(F32), dt)
        return O.round_to((sl * u).astype(F32), dt)


def linear_q8_forward(xq, wq, dt: str, s_scalar: float, s_row=None, s_col=None, bias=None, act: Optional[str] = None,
                      qmode: str = "per-tensor-round", quant_scale: float = 1.0) -> np.ndarray:
    """asq_linear_w8a8_q8: the linear's output in dt (oracle.w8a8.dequant_epilogue, layers/nn/linear.py:104-105), an optional ReLU
    (models/opt.py:127-128), then the consumer's per-tensor prologue (linear.py:95-96 round / :289-292 divide) -> int8.
    Exact: every step is elementwise."""
    acc = O.igemm(np.asarray(xq, dtype=np.int8), np.asarray(wq, dtype=np.int8))
    y = O.dequant_epilogue(acc, s_col if s_col is not None else s_scalar, s_row, bias, dt)
    if act == "relu":
        y = np.where(y < 0, F32(0), y).astype(F32)
    return O.act_quant_div(y, dt, quant_scale) if qmode == "per-tensor-div" else O.act_quant_round(y, dt)


# --------------------------------------------------------------------------
# reference formulas (float64 statistics) + the boundary test
# --------------------------------------------------------------------------
def norm_y_reference(h, dt: str, weight, bias=None, eps: float = 1e-5) -> np.ndarray:
    h64 = np.asarray(h, dtype=np.float64)
    w = np.asarray(weight, dtype=F32)[None, :]
    if bias is not None:
        mean = h64.mean(axis=1, keepdims=True)
        var = ((h64 - mean) ** 2).mean(axis=1, keepdims=True)
        rs = (1.0 / np.sqrt(var + eps))
        y = ((h64 - mean) * rs).astype(F32) * w + np.asarray(bias, dtype=F32)[None, :]
        return O.round_to(y.astype(F32), dt)
    var = (h64 ** 2).mean(axis=1, keepdims=True)
    rs = (1.0 / np.sqrt(var + eps)).astype(F32)
    n = O.round_to((np.asarray(h, dtype=F32) * rs).astype(F32), dt)
    return O.round_to((w * n).astype(F32), dt)


def silu_mul_reference(gate, up, dt: str) -> np.ndarray:
    g64 = np.asarray(gate, dtype=np.float64)
    with np.errstate(over="ignore"):
        sl = O.round_to((g64 / (1.0 + np.exp(-g64))).astype(F32), dt)
    return O.round_to((sl * np.asarray(up, dtype=F32)).astype(F32), dt)


def near_rounding_boundary(y, rel: float = 4e-6, abs_: float = 1e-6) -> np.ndarray:
    """True where y is within a few fp32 ulps (plus the ulps its inputs carry) of k + 0.5: the only places where a
    different summation order / exp implementation may move round(y) by one."""
    y = np.asarray(y, dtype=np.float64)
    frac = np.abs(y - np.floor(y) - 0.5)
    return frac <= (np.abs(y) * rel + abs_)
