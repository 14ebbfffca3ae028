"""NumPy restatement of the reference's one executable decoder-layer composition:
``Int8BaichuanLayer`` with position_embedding "ALIBI" and ``attention_mask=None``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Citations into ``/root/reference/autosmoothquant``:
  models/baichuan.py:289-331    Int8BaichuanLayer.forward (pre-norm, in-place residual adds)
  models/baichuan.py:110-183    Int8BaichuanAttention.forward (W_pack -> fp16, eager softmax attention, o_proj)
  models/baichuan.py:225-229    Int8BaichuanMLP.forward (act(gate.to(fp16)) * up -> down)
  models/baichuan.py:258-287    from_float: which linear class / act_quant each projection gets; norm weight
                                divided by the input scale iff that group is per-tensor
  thirdparty/baichuan/modeling_baichuan.py:161-176   RMSNorm

The linears are the bit-exact restatements of ``oracle/w8a8.py``.  The glue between them (fp16 attention
matmuls, softmax, SiLU) is torch CPU arithmetic in the reference, whose summation order is not specified:
this file rounds to fp16 at the same points and is compared with the golden outputs by tolerance.
"""
from __future__ import annotations

import numpy as np

from . import w8a8 as O

DEFAULT_QC = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}


def rmsnorm(x, weight, eps):
    """modeling_baichuan.py:167-176 with fp32 weights: x * rsqrt(mean(x^2) + eps), then weight * (.)"""
    x = np.asarray(x, np.float32)
    var = np.mean(x.astype(np.float32) ** 2, axis=-1, keepdims=True, dtype=np.float32)
    return (np.asarray(weight, np.float32) * (x * (np.float32(1.0) / np.sqrt(var + np.float32(eps))))).astype(np.float32)


def convert(W, scales, qc):
    """Int8BaichuanLayer.from_float (models/baichuan.py:258-287) on fp32 source weights."""
    H = W["o_proj"].shape[0]
    a_in, o_in, m_in, d_in = [float(s) for s in scales]
    P = {"qc": dict(qc), "qkv_size": [H, H, H]}
    P["wpack_q"], P["wpack_s"] = O.qkv_from_float(W["W_pack"], "f32", a_in, P["qkv_size"], qc["qkv"])
    P["o_q"], P["o_s"] = O.linear_from_float(W["o_proj"], "f32", o_in, qc["out"])
    P["gate_q"], P["gate_s"] = O.linear_from_float(W["gate_proj"], "f32", m_in, qc["fc1"])
    P["up_q"], P["up_s"] = O.linear_from_float(W["up_proj"], "f32", m_in, qc["fc1"])
    P["down_q"], P["down_s"] = O.linear_from_float(W["down_proj"], "f32", d_in, qc["fc2"])
    P["o_qs"], P["down_qs"] = np.float32(o_in), np.float32(d_in)  # quant_scale buffers of the WithQuantScale linears (linear.py:324)
    P["ln1"] = O.fold_norm_weight(W["ln1"], "f32", a_in) if qc["qkv"] == "per-tensor" else np.asarray(W["ln1"], np.float32)
    P["ln2"] = O.fold_norm_weight(W["ln2"], "f32", m_in) if qc["fc1"] == "per-tensor" else np.asarray(W["ln2"], np.float32)
    return P


def attention(h, P, heads):
    B, S, H = h.shape
    hd = H // heads
    qc = P["qc"]
    proj = O.round_to(O.qkv_linear_forward(h, "f32", P["wpack_q"], P["wpack_s"], P["qkv_size"], None, qc["qkv"]), "f16")  # .to(torch.float16)
    q, k, v = [proj[..., i * H:(i + 1) * H].reshape(B, S, heads, hd).transpose(0, 2, 1, 3) for i in range(3)]
    s = O.round_to(np.matmul(q, k.transpose(0, 1, 3, 2)), "f16")
    s = O.round_to(s / np.float32(np.sqrt(hd)), "f16")                      # fp16 tensor / python float
    e = np.exp(s - s.max(axis=-1, keepdims=True))
    p = O.round_to(e / e.sum(axis=-1, keepdims=True), "f16")                # softmax computes in fp32, returns fp16
    a = O.round_to(np.matmul(p, v), "f16").transpose(0, 2, 1, 3).reshape(B, S, H)
    return O.linear_with_quant_scale_forward(a, "f16", P["o_q"], P["o_s"], P["o_qs"], None, qc["out"])


def mlp(h, P):
    qc = P["qc"]
    g = O.round_to(O.linear_forward(h, "f32", P["gate_q"], P["gate_s"], None, qc["fc1"]), "f16")
    act = O.round_to(g / (np.float32(1.0) + np.exp(-g)), "f16")             # SiLU on an fp16 tensor
    hidden = (act * O.linear_forward(h, "f32", P["up_q"], P["up_s"], None, qc["fc1"])).astype(np.float32)  # fp16 * fp32 -> fp32
    return O.linear_with_quant_scale_forward(hidden, "f32", P["down_q"], P["down_s"], P["down_qs"], None, qc["fc2"])


def layer_forward(x, P, heads, eps=1e-6):
    """Int8BaichuanLayer.forward, models/baichuan.py:289-331 (fp32 hidden states)."""
    res = np.array(x, np.float32, copy=True)
    res += attention(rmsnorm(res, P["ln1"], eps), P, heads).astype(np.float32)
    res += mlp(rmsnorm(res, P["ln2"], eps), P).astype(np.float32)
    return res
