"""NumPy restatement of the reference's FP8 (OCP e4m3fn / e5m2) linear path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Citations into ``/root/reference``:
  layers/functional/quantization.py:144-211   per_tensor / per_token / static quantisers
  layers/nn/linear.py:336-369                 easy_fp8_gemm (native_fp8_support = False:
                                              dequantise both operands, F.linear)
  layers/nn/linear.py:373-452, 503-580        FP8LinearDynamic / FP8LinearStatic

FP8 tensors travel as raw ``uint8`` bit patterns.  Conventions for floating tensors are those
of ``oracle/w8a8.py`` (fp32 arrays holding values representable in the logical dtype).
"""
from __future__ import annotations

import numpy as np

from .w8a8 import round_to

E4M3_MAX = np.float32(448.0)   # torch.finfo(torch.float8_e4m3fn).max
E5M2_MAX = np.float32(57344.0)


# --------------------------------------------------------------------------
# codecs: fp32 -> fp8 bits with round-to-nearest-even (what tensor.to(torch.float8_*) does)
# --------------------------------------------------------------------------
def _encode(a: np.ndarray, ebits: int, mbits: int, bias: int, has_inf: bool) -> np.ndarray:
    """Generic IEEE-style narrowing of fp32 to a 1+ebits+mbits format, RNE.
    e4m3fn: no infinities, exponent field all-ones is finite except mantissa all-ones (= NaN);
    values that round above the max finite become NaN (the cast is unsaturated).
    e5m2: IEEE-like (inf / NaN in the top exponent)."""
    a = np.asarray(a, dtype=np.float32)
    shape = a.shape
    f = np.ascontiguousarray(a).reshape(-1)
    sign = (f.view(np.uint32) >> 31).astype(np.uint8) << 7
    mag = np.abs(f).astype(np.float64)
    mag_safe = np.where(np.isfinite(mag), mag, 0.0)
    out = np.zeros(f.shape, dtype=np.uint8)
    emin = 1 - bias                                   # exponent of the smallest normal
    nan_code = 0x7F if not has_inf else (((1 << ebits) - 1) << mbits) | (1 << (mbits - 1))
    inf_code = ((1 << ebits) - 1) << mbits
    finite = np.isfinite(mag)
    nz = finite & (mag > 0)
    e = np.zeros(f.shape, dtype=np.int64)
    e[nz] = np.floor(np.log2(mag[nz])).astype(np.int64)
    # guard against log2 rounding at exact powers of two
    fix_hi = nz & (np.ldexp(1.0, e + 1) <= mag)
    e[fix_hi] += 1
    fix_lo = nz & (np.ldexp(1.0, e) > mag)
    e[fix_lo] -= 1
    e_eff = np.maximum(e, emin)                       # subnormals share emin
    q = np.rint(np.ldexp(mag_safe, mbits - e_eff))    # mantissa incl. hidden bit, RNE (np.rint = half-even)
    # mantissa overflow from rounding -> next binade
    ovf = q >= (1 << (mbits + 1))
    e_eff = np.where(ovf, e_eff + 1, e_eff)
    q = np.where(ovf, q / 2, q)
    is_norm = q >= (1 << mbits)
    exp_field = np.where(is_norm, e_eff + bias, 0).astype(np.int64)
    man_field = np.where(is_norm, q - (1 << mbits), q).astype(np.int64)
    code = (exp_field << mbits) | man_field
    if has_inf:
        too_big = exp_field >= ((1 << ebits) - 1)
        code = np.where(too_big, inf_code, code)
    else:
        max_code = (((1 << ebits) - 1) << mbits) | ((1 << mbits) - 2)   # 0x7E for e4m3fn
        code = np.where(code > max_code, nan_code, code)
    out[nz] = code[nz].astype(np.uint8)
    out[~finite] = nan_code if not has_inf else np.where(np.isnan(mag[~finite]), nan_code, inf_code).astype(np.uint8)
    out = out | sign
    return out.reshape(shape)


def _decode(b: np.ndarray, ebits: int, mbits: int, bias: int, has_inf: bool) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = ((b & 0x7F) >> mbits).astype(np.int64)
    m = (b & ((1 << mbits) - 1)).astype(np.float64)
    val = np.where(e == 0, np.ldexp(m, 1 - bias - mbits), np.ldexp(m + (1 << mbits), e - bias - mbits))
    top = (1 << ebits) - 1
    if has_inf:
        val = np.where((e == top) & (m == 0), np.inf, val)
        val = np.where((e == top) & (m != 0), np.nan, val)
    else:
        val = np.where((e == top) & (m == (1 << mbits) - 1), np.nan, val)
    return (s * val).astype(np.float32)


def f32_to_e4m3fn(a):
    return _encode(a, 4, 3, 7, False)


def e4m3fn_to_f32(b):
    return _decode(b, 4, 3, 7, False)


def f32_to_e5m2(a):
    return _encode(a, 5, 2, 15, True)


def e5m2_to_f32(b):
    return _decode(b, 5, 2, 15, True)


# --------------------------------------------------------------------------
# quantisers (quantization.py:144-211)
# --------------------------------------------------------------------------
def _clamp448(v):
    with np.errstate(invalid="ignore"):
        return np.clip(v, -E4M3_MAX, E4M3_MAX)


def per_tensor_quantize_fp8(t, dt):
    """quantization.py:144-170: scale = absmax/448 in t's dtype (0-dim); q = clamp(t/scale) -> e4m3fn.
    Empty tensors use the fixed range [-16, 16] (empty MoE experts)."""
    t = np.asarray(t, np.float32)
    amax = np.float32(16.0) if t.size == 0 else np.float32(np.max(np.abs(t)))
    scale = round_to(amax / E4M3_MAX, dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = round_to(t / np.float32(scale), dt)   # tensor / 0-dim tensor of the same dtype stays in dt
    return f32_to_e4m3fn(_clamp448(v)), np.float32(scale)


def per_token_quantize_fp8(t, dt):
    """quantization.py:173-191: scale[M,1] = (rowabsmax / 448 in t's dtype).to(f32); t / scale promotes to fp32."""
    t = np.asarray(t, np.float32)
    scale = round_to(np.max(np.abs(t), axis=-1, keepdims=True) / E4M3_MAX, dt).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = (t / scale).astype(np.float32)
    return f32_to_e4m3fn(_clamp448(v)), scale


def static_per_tensor_quantize_fp8(t, dt, inv_scale):
    """quantization.py:208-211: (t / inv_scale).clamp -> e4m3fn; the 0-dim fp32 scale does not promote t."""
    t = np.asarray(t, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = round_to(t / np.float32(inv_scale), dt)
    return f32_to_e4m3fn(_clamp448(v))


# --------------------------------------------------------------------------
# GEMM + module forwards (fp32 activations: the only dtype the reference's CPU path accepts with a
# bias or per-token scales -- F.linear rejects mixed Float/Half operands, linear.py:364-368)
# --------------------------------------------------------------------------
def easy_fp8_gemm(aq, a_scale, wq, w_scale, bias, out_dt="f32"):
    """linear.py:336-369 with native_fp8_support = False:
    F.linear(A.to(dt) * A_scale, B.to(dt) * B_scale, bias).to(dt).  Restated with fp32 accumulation
    (the reference's BLAS order of summation is not specified: compare with a tolerance)."""
    if aq.size == 0:
        return np.zeros((0, wq.shape[0]), np.float32)
    a = round_to(e4m3fn_to_f32(aq) * np.asarray(a_scale, np.float32), out_dt)
    w = round_to(e4m3fn_to_f32(wq) * np.float32(w_scale), out_dt)
    out = a.astype(np.float64) @ w.astype(np.float64).T
    if bias is not None:
        out = out + np.asarray(bias, np.float64)[None, :]
    return round_to(out.astype(np.float32), out_dt)


def fp8_linear_dynamic_forward(x, dt, wq, w_scale, bias, act_quant):
    """FP8LinearDynamic.forward, linear.py:413-427."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if act_quant == "per-token":
        q, s = per_token_quantize_fp8(x2, dt)
    else:
        q, s = per_tensor_quantize_fp8(x2, dt)
    return easy_fp8_gemm(q, s, wq, w_scale, bias, dt).reshape(*lead, -1)


def fp8_linear_static_forward(x, dt, wq, w_scale, input_scale, output_scale, bias):
    """FP8LinearStatic.forward, linear.py:553-566 (output re-quantised iff output_scale is truthy)."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    q = static_per_tensor_quantize_fp8(x2, dt, input_scale)
    out = easy_fp8_gemm(q, np.float32(input_scale), wq, w_scale, bias, dt)
    if output_scale:
        qo = static_per_tensor_quantize_fp8(out, dt, output_scale)
        out = round_to(e4m3fn_to_f32(qo) * np.float32(output_scale), dt)
    return out.reshape(*lead, -1)


class StaticQuantizerState:
    """Running state of the reference's calibration-time FP8StaticLinearQuantizer (layers/nn/linear.py:455-500): the maxima of the dynamic per-tensor
    input (and, with quantize_output, output) scales seen so far."""

    def __init__(self, quantize_output=False):
        self.input_scale, self.output_scale, self.quantize_output = None, None, quantize_output


def fp8_static_quantizer_forward(state, x, dt, wq, w_scale, bias):
    """One calibration forward (linear.py:474-499): quantise x per tensor, keep the running MAXIMUM of the scale, and -- as the reference does --
    dequantise this batch's codes with the RUNNING scale (not this batch's own) in the GEMM; optionally the same for the output."""
    q, s = per_tensor_quantize_fp8(x, dt)
    s = np.float32(s)
    if state.input_scale is None or s > state.input_scale:
        state.input_scale = s
    out = easy_fp8_gemm(q, state.input_scale, wq, w_scale, bias, dt)
    if state.quantize_output:
        qo, so = per_tensor_quantize_fp8(out, dt)
        so = np.float32(so)
        if state.output_scale is None or so > state.output_scale:
            state.output_scale = so
        # qoutput.to(output.dtype) * output_scale (linear.py:497): this batch's codes times THIS batch's scale, in the activation dtype
        from .w8a8 import round_to
        out = round_to(round_to(e4m3fn_to_f32(qo), dt) * round_to(np.float32(so), dt), dt)
    return out
