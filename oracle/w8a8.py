"""NumPy restatement of the reference's W8A8 ``Int8Linear`` arithmetic (CPU path).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Citations are into
``/root/reference``.

Conventions
-----------
* Floating tensors travel as ``np.float32`` arrays whose values are exactly
  representable in the *logical* dtype named by ``dt`` ("f32", "f16", "bf16").
  NumPy has no bfloat16, so bf16 is emulated: every reference op that would
  produce a bf16/fp16 tensor is restated as "compute in fp32, round once to the
  logical dtype" -- which is what ATen's CPU kernels do for reduced floating
  types (opmath = float).
* All fp32 arithmetic is done with NumPy float32 ufuncs: one IEEE rounding per
  operation, no FMA contraction.
* int8 x int8 -> int32 products are exact integers (csrc/int8gemm is an exact
  GEMM: cublasINT8MMWrapper.cc:231,276-277).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

F32, F16, BF16 = "f32", "f16", "bf16"
_DTS = (F32, F16, BF16)


# --------------------------------------------------------------------------
# logical-dtype helpers
# --------------------------------------------------------------------------
def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32, NaN preserved."""
    a = np.asarray(a, dtype=np.float32)
    shape = a.shape
    a = np.ascontiguousarray(a).reshape(-1)
    u = a.view(np.uint32).astype(np.uint64)
    bias = ((u >> 16) & 1) + 0x7FFF
    r = ((u + bias) & 0xFFFF0000).astype(np.uint32)
    r = np.where(np.isnan(a), np.uint32(0x7FC00000), r).astype(np.uint32)
    return r.view(np.float32).reshape(shape)


def round_to(a: np.ndarray, dt: str) -> np.ndarray:
    """Round an fp32 array once to logical dtype ``dt``; result stays fp32."""
    assert dt in _DTS, dt
    a = np.asarray(a, dtype=np.float32)
    if dt == F32:
        return a
    if dt == F16:
        with np.errstate(over="ignore"):
            return a.astype(np.float16).astype(np.float32)
    return bf16_round(a)


def is_representable(a: np.ndarray, dt: str) -> bool:
    a = np.asarray(a, dtype=np.float32)
    r = round_to(a, dt)
    return bool(np.array_equal(a, r, equal_nan=True))


def _to_int8(v: np.ndarray) -> np.ndarray:
    """``.round().clamp(-128, 127).to(torch.int8)`` (linear.py:92,96,292).

    round = half-to-even; clamp propagates NaN; the NaN -> int8 cast yields 0 on
    the reference's CPU path (observed; SURVEY 7 "all-zero rows")."""
    with np.errstate(invalid="ignore"):
        q = np.rint(v.astype(np.float32))
        q = np.clip(q, np.float32(-128), np.float32(127))
        q = np.where(np.isnan(q), np.float32(0), q)
    return q.astype(np.int8)


# --------------------------------------------------------------------------
# activation quantisers (GEMM prologue)
# --------------------------------------------------------------------------
def act_quant_per_token(x: np.ndarray, dt: str) -> Tuple[np.ndarray, np.ndarray]:
    """Per-token dynamic quantisation, linear.py:88-92 (= :164-168, :283-292).

    quant_scale = x.abs().max(-1, keepdim)[0].div(127.0).to(float32)
        -> absmax is exact; the division happens in x's dtype (fp32 opmath,
           one rounding to ``dt``), then widens to fp32.
    x_q = (x / quant_scale).round().clamp(-128, 127).to(int8)
        -> x[dt] / qs[f32 tensor] promotes to fp32: true IEEE division.
    Returns (x_q int8 [M,K], quant_scale f32 [M])."""
    x = np.asarray(x, dtype=np.float32)
    assert x.ndim == 2
    absmax = np.max(np.abs(x), axis=1) if x.shape[1] else np.zeros(x.shape[0], np.float32)
    qs = round_to(absmax.astype(np.float32) / np.float32(127.0), dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = x / qs[:, None]
    return _to_int8(v), qs.astype(np.float32)


def act_quant_round(x: np.ndarray, dt: str) -> np.ndarray:
    """Per-tensor branch of W8A8BFP32OFP32Linear / QKVLinear (linear.py:95-96,
    :174): ``x.round().clamp(-128,127).to(int8)`` -- no scale, 1/input_scale was
    folded into the preceding norm (models/llama.py:326-339)."""
    return _to_int8(np.asarray(x, dtype=np.float32))


def act_quant_div(x: np.ndarray, dt: str, quant_scale: float) -> np.ndarray:
    """Per-tensor branch of ...LinearWithQuantScale (linear.py:289-292):
    ``(x / quant_scale.item()).round().clamp().to(int8)``.  Division by a Python
    scalar keeps x's dtype: fp32 opmath true division, one rounding to ``dt``."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = round_to(x / np.float32(quant_scale), dt)
    return _to_int8(v)


# --------------------------------------------------------------------------
# exact integer GEMM  (K1: bindings.cpp:69-84 -> cublasINT8MMWrapper.cc:224-354)
# --------------------------------------------------------------------------
_CLIB = None


def _load_clib():
    global _CLIB
    if _CLIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasq_oracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.asq_oracle_igemm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int]
            lib.asq_oracle_igemm.restype = None
            _CLIB = lib
        else:
            _CLIB = False
    return _CLIB


def igemm_numpy(xq: np.ndarray, wq: np.ndarray) -> np.ndarray:
    """out[M,N] = xq[M,K] . wq[N,K]^T in exact integers.

    Uses fp32 BLAS on K-chunks of 1024: every partial sum is an integer of
    magnitude <= 1024*128*128 = 2^24, hence exactly representable in fp32 in any
    summation order; chunk results are accumulated in int64."""
    xq = np.asarray(xq, dtype=np.int8)
    wq = np.asarray(wq, dtype=np.int8)
    M, K = xq.shape
    N, K2 = wq.shape
    assert K == K2
    acc = np.zeros((M, N), dtype=np.int64)
    for k0 in range(0, K, 1024):
        a = xq[:, k0:k0 + 1024].astype(np.float32)
        b = wq[:, k0:k0 + 1024].astype(np.float32)
        acc += (a @ b.T).astype(np.int64)
    assert np.abs(acc).max(initial=0) < 2 ** 31
    return acc.astype(np.int32)


def igemm_c(xq: np.ndarray, wq: np.ndarray, threads: int = 0) -> np.ndarray:
    """Same product through the plain-C restatement (oracle/igemm_ref.c)."""
    lib = _load_clib()
    if not lib:
        raise RuntimeError("oracle/libasq_oracle.so not built (run `make -C oracle`)")
    xq = np.ascontiguousarray(xq, dtype=np.int8)
    wq = np.ascontiguousarray(wq, dtype=np.int8)
    M, K = xq.shape
    N = wq.shape[0]
    out = np.empty((M, N), dtype=np.int32)
    lib.asq_oracle_igemm(xq.ctypes.data, wq.ctypes.data, out.ctypes.data, M, N, K, int(threads))
    return out


def igemm(xq: np.ndarray, wq: np.ndarray) -> np.ndarray:
    return igemm_numpy(xq, wq)


# --------------------------------------------------------------------------
# dequant / bias epilogue
# --------------------------------------------------------------------------
def dequant_epilogue(acc: np.ndarray, s_col, s_row: Optional[np.ndarray], bias: Optional[np.ndarray],
                     out_dt: str, order: str = "scale_first") -> np.ndarray:
    """out = dequant(acc) (+ bias) -> out_dt.

    order="scale_first" (module semantics, linear.py:93,104 / :200-206):
        ds = s_col * s_row      (fp32; when s_row is None ds = s_col)
        out = ds * float(acc)   (fp32)
        out = out + bias        (fp32, separate rounding)
    order="acc_first" (functional helpers quantization.py:103-120):
        out = (float(acc) * s_col) * s_row
    s_col is a scalar or an fp32 [N] vector (per-channel / QKV segments)."""
    accf = acc.astype(np.float32)  # int32 -> fp32, round-to-nearest-even
    s_col = np.asarray(s_col, dtype=np.float32)
    if s_col.ndim == 1:
        s_col = s_col[None, :]
    if order == "scale_first":
        ds = s_col if s_row is None else (s_col * np.asarray(s_row, np.float32)[:, None]).astype(np.float32)
        out = (ds * accf).astype(np.float32)
    elif order == "acc_first":
        out = (accf * s_col).astype(np.float32)
        if s_row is not None:
            out = (out * np.asarray(s_row, np.float32)[:, None]).astype(np.float32)
    else:
        raise ValueError(order)
    if bias is not None:
        out = (out + np.asarray(bias, np.float32)[None, :]).astype(np.float32)
    return round_to(out, out_dt)


# --------------------------------------------------------------------------
# module forwards
# --------------------------------------------------------------------------
def linear_forward(x, dt, wq, dequant_scale: float, bias=None, act_quant="per-tensor",
                   return_intermediates=False):
    """W8A8BFP32OFP32Linear.forward, linear.py:83-106."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if act_quant == "per-token":
        xq, qs = act_quant_per_token(x2, dt)
        s_row = qs
    else:
        xq, qs, s_row = act_quant_round(x2, dt), None, None
    acc = igemm(xq, wq)
    out = dequant_epilogue(acc, np.float32(dequant_scale), s_row, bias, dt)
    out = out.reshape(*lead, -1)
    if return_intermediates:
        return out, xq, qs, acc
    return out


def qkv_scale_vector(scales: Sequence[float], qkv_size: Sequence[int]) -> np.ndarray:
    return np.concatenate([np.full(int(n), np.float32(s), np.float32) for s, n in zip(scales, qkv_size)])


def qkv_linear_forward(x, dt, wq, qkv_scales: Sequence[float], qkv_size: Sequence[int], bias=None,
                       act_quant="per-tensor", return_intermediates=False):
    """W8A8BFP32OFP32QKVLinear.forward, linear.py:158-208: one GEMM, three scalar
    dequant scales on three column segments (split/cat), bias split likewise."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if act_quant == "per-token":
        xq, qs = act_quant_per_token(x2, dt)
        s_row = qs
    else:
        xq, qs, s_row = act_quant_round(x2, dt), None, None
    acc = igemm(xq, wq)
    out = dequant_epilogue(acc, qkv_scale_vector(qkv_scales, qkv_size), s_row, bias, dt)
    out = out.reshape(*lead, -1)
    if return_intermediates:
        return out, xq, qs, acc
    return out


def linear_with_quant_scale_forward(x, dt, wq, dequant_scale: float, quant_scale: Optional[float] = None,
                                    bias=None, act_quant="per-token", return_intermediates=False):
    """W8A8BFP32OFP32LinearWithQuantScale.forward, linear.py:278-302."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if act_quant == "per-token":
        xq, qs = act_quant_per_token(x2, dt)
        s_row = qs
    else:
        xq, qs, s_row = act_quant_div(x2, dt, quant_scale), None, None
    acc = igemm(xq, wq)
    out = dequant_epilogue(acc, np.float32(dequant_scale), s_row, bias, dt)
    out = out.reshape(*lead, -1)
    if return_intermediates:
        return out, xq, qs, acc
    return out


# --------------------------------------------------------------------------
# weight-side conversion (from_float) and functional helpers
# --------------------------------------------------------------------------
def quantize_per_tensor_absmax(w, dt) -> Tuple[np.ndarray, np.float32]:
    """layers/functional/quantization.py:9-18 (CPU branch): scale = absmax/127 in
    w's dtype; w.float().div_(scale).round_() -> int8 (no clamp)."""
    w = np.asarray(w, np.float32)
    scale = round_to(np.float32(np.max(np.abs(w))) / np.float32(127), dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.rint(w / np.float32(scale))
    return q.astype(np.int8), np.float32(scale)


def quantize_weight_per_channel_absmax(w, dt) -> Tuple[np.ndarray, np.ndarray]:
    """quantization.py:40-50: scales[N,1] = rowabsmax/127 (w dtype);
    w.float().div_(scales).round_().clamp_(-128,127) -> int8."""
    w = np.asarray(w, np.float32)
    scales = round_to(np.max(np.abs(w), axis=1) / np.float32(127), dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = _to_int8(w / scales[:, None])
    return q, scales.astype(np.float32)


def linear_from_float(w, w_dt, input_scale: float, act_quant="per-tensor"):
    """W8A8BFP32OFP32Linear.from_float, linear.py:108-129 -> (wq, dequant_scale f32)."""
    wq, ws = quantize_per_tensor_absmax(w, w_dt)
    if act_quant == "per-token":
        alpha = np.float32(ws)
    else:
        # input_scale (python float) * weight_scale (0-dim tensor of w dtype) -> w dtype
        alpha = np.float32(round_to(np.float32(input_scale) * np.float32(ws), w_dt))
    return wq, alpha


def qkv_from_float(w, w_dt, input_scale: float, qkv_size, act_quant="per-tensor"):
    """W8A8BFP32OFP32QKVLinear.from_float, linear.py:210-245."""
    w = np.asarray(w, np.float32)
    wqs, scales, o = [], [], 0
    for n in qkv_size:
        q, s = quantize_per_tensor_absmax(w[o:o + n], w_dt)
        if act_quant == "per-tensor":
            # weight_scale (0-dim tensor, w dtype) * input_scale (python float)
            s = np.float32(round_to(np.float32(s) * np.float32(input_scale), w_dt))
        wqs.append(q)
        scales.append(np.float32(s))
        o += n
    return np.concatenate(wqs, axis=0), scales


def dynamic_quantize_activation_per_token_absmax(t, dt):
    """quantization.py:78-84: max_val = clamp(rowabsmax, 1e-8)/127 (t dtype, [M,1]);
    t.div_(max_val).round_().clamp_() in t dtype -> int8."""
    t = np.asarray(t, np.float32)
    mv = round_to(np.max(np.abs(t), axis=-1, keepdims=True), dt)
    mv = round_to(np.maximum(mv, round_to(np.float32(1e-8), dt)), dt)
    mv = round_to(mv / np.float32(127), dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = round_to(t / mv, dt)
    return _to_int8(v), mv


def dynamic_quantize_activation_per_tensor_absmax(t, dt):
    """quantization.py:70-75."""
    t = np.asarray(t, np.float32)
    mv = np.float32(np.max(np.abs(t)))
    mv = round_to(np.maximum(mv, round_to(np.float32(1e-8), dt)), dt)
    mv = round_to(mv / np.float32(127), dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = round_to(t / mv, dt)
    return _to_int8(v), np.float32(mv)


def dequantize_activation_w_per_channel_a_per_token(acc, w_scales, a_scales, a_dt):
    """quantization.py:103-111: q.float().mul_(w_scales[1,N]).mul_(a_scales[M,1]).to(a dtype)."""
    return dequant_epilogue(acc, np.asarray(w_scales, np.float32).reshape(-1),
                            np.asarray(a_scales, np.float32).reshape(-1), None, a_dt, order="acc_first")


def dequantize_activation_w_per_channel_a_per_tensor(acc, w_scales, a_scale, a_dt):
    """quantization.py:113-120: (q.float() * w_scales[1,N]) * a_scales."""
    out = (acc.astype(np.float32) * np.asarray(w_scales, np.float32).reshape(1, -1)).astype(np.float32)
    out = (out * np.float32(a_scale)).astype(np.float32)
    return round_to(out, a_dt)


# --------------------------------------------------------------------------
# SmoothQuant scale folding (a-11) -- what makes the per-tensor prologue scale-free
# --------------------------------------------------------------------------
def fold_norm_weight(norm_weight, dt, input_scale: float):
    """QuantizedLlamaRMSNorm.from_float, models/llama.py:27-37: weight / input_scale
    (python-float divisor: stays in weight's dtype)."""
    w = np.asarray(norm_weight, np.float32)
    return round_to(w / np.float32(input_scale), dt)


# --------------------------------------------------------------------------
# optional fast exact backends for the CPU-baseline leg of bench.py
# --------------------------------------------------------------------------
def igemm_torch(xq: np.ndarray, wq: np.ndarray) -> np.ndarray:
    """torch._int_mm (oneDNN int8 -> int32, exact) -- the "torch int8 matmul" the reference's
    CPU path would use; falls back to an int32 matmul for shapes _int_mm rejects."""
    import torch
    a = torch.from_numpy(np.ascontiguousarray(xq, dtype=np.int8))
    b = torch.from_numpy(np.ascontiguousarray(wq, dtype=np.int8))
    try:
        return torch._int_mm(a, b.t()).numpy()
    except RuntimeError:
        return (a.to(torch.int32) @ b.to(torch.int32).t()).numpy()


_BACKENDS = {"numpy": igemm_numpy, "c": igemm_c, "torch": igemm_torch}


def set_igemm_backend(name: str):
    """Select the exact-integer GEMM used by the module forwards (all are bit-identical)."""
    global igemm
    igemm = _BACKENDS[name]
