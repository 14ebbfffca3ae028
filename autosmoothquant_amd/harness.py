"""LLaMA-architecture decoder-layer harness around the W8A8 linears (measurement plumbing).

The reference does not own this arithmetic: it borrows ``forward`` from HF ``transformers==4.42.3``
(reference models/llama.py:111,218,289) and only decides (1) which quantised linear class and
``act_quant`` each projection uses (:99-106, :206-211) and (2) when ``1/input_scale`` is folded
into the preceding RMSNorm weight (:27-37, :326-339).  Those two things are reproduced here;
RMSNorm, RoPE, causal attention and SiLU are stock PyTorch-ROCm ops.
Nothing in here is on the product's hot path except the seven linears.
"""
import math

import os

import torch
import torch.nn.functional as F

from .layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale
from .layers.nn.fused import DeferredResidual


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden, eps=1e-5):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(hidden))
        self.eps = eps

    def forward(self, x):
        w = self.weight
        if (GLUE_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.dtype == x.dtype and x.shape[-1] % 8 == 0 and x.shape[-1] <= 16384
                and not torch.is_grad_enabled()):
            # one pass instead of seven torch kernels (round 6: 14.6 % of the GPU time of BASELINE configs[2]'s forward in the reference's module composition); HF
            # LlamaRMSNorm's roundings, this library's summation order (bit-identical to oracle/n1.py, within an ulp of the torch ops below)
            from . import ops
            x2 = x.reshape(-1, x.shape[-1])
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            if x2.data_ptr() % 16 == 0 and w.is_contiguous() and w.data_ptr() % 16 == 0:
                return ops.rmsnorm(x2, w.detach(), self.eps).view(x.shape)
        v = x.float()
        v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.eps)
        return self.weight * v.to(x.dtype)

    def folded(self, input_scale):
        """weight / input_scale (reference QuantizedLlamaRMSNorm.from_float, models/llama.py:27-37):
        the norm then emits activations already in int8 units for a per-tensor linear."""
        n = RMSNorm(self.weight.numel(), self.eps)
        n.weight = torch.nn.Parameter(self.weight.detach() / input_scale)
        return n.to(self.weight.device, self.weight.dtype)


_ROPE_CACHE = {}


def _rope_tables(S, d, device, dtype, theta=10000.0):
    key = (S, d, str(device), dtype, theta)
    if key not in _ROPE_CACHE:
        inv = 1.0 / (theta ** (torch.arange(0, d, 2, device=device, dtype=torch.float32) / d))
        ang = torch.arange(S, device=device, dtype=torch.float32)[:, None] * inv[None, :]
        _ROPE_CACHE.clear()   # one entry: the harness runs one (S, d) at a time
        _ROPE_CACHE[key] = (torch.cos(ang)[None, None].to(dtype), torch.sin(ang)[None, None].to(dtype))
    return _ROPE_CACHE[key]


def _rope(x, theta=10000.0):
    """Rotary embedding, HF "rotate_half" convention, positions 0..S-1; x [B, H, S, D] -- here always the transposed view of a projection's [B, S, H, D] output: that
    case is ONE kernel (ops.rope / asq_rope, round 6: the four strided torch kernels per tensor were 8.5 % of the GPU time of BASELINE configs[2]'s forward); anything
    else takes the torch composition, which the kernel reproduces bit for bit."""
    d = x.shape[-1]
    if x.is_cuda and x.dim() == 4 and d % 16 == 0 and x.dtype in (torch.float16, torch.bfloat16) and ROPE_KERNEL:
        xb = x.transpose(1, 2)   # [B, S, H, D]: dense (a projection's own output) or a slice of the fused q || k || v GEMM's output (one row pitch)
        st, (B_, S_, H_, _) = xb.stride(), xb.shape
        if st[3] == 1 and st[2] == d and st[1] >= H_ * d and st[1] % 8 == 0 and st[0] == S_ * st[1] and xb.data_ptr() % 16 == 0:
            from . import ops
            cos, sin = _rope_tables(x.shape[-2], d, x.device, x.dtype, theta)
            return ops.rope(xb, cos.view(-1, d // 2), sin.view(-1, d // 2)).transpose(1, 2)
    return _rope_torch(x, theta)


ROPE_KERNEL = os.environ.get("ASQ_ROPE_KERNEL", "1") != "0"   # 0: the torch composition everywhere (A/B)
GLUE_KERNELS = os.environ.get("ASQ_GLUE_KERNELS", "1") != "0"  # 0: RMSNorm / SiLU * up of the reference composition as torch ops (A/B)


def _silu_mul(gate, up):
    """F.silu(gate) * up; on the device in the 16-bit dtypes one pass (ops.silu_mul, round 6) instead of two torch kernels"""
    if (GLUE_KERNELS and gate.is_cuda and gate.dtype in (torch.float16, torch.bfloat16) and gate.shape == up.shape and gate.is_contiguous() and up.is_contiguous()
            and gate.numel() % 8 == 0 and gate.data_ptr() % 16 == 0 and up.data_ptr() % 16 == 0 and not torch.is_grad_enabled()):
        from . import ops
        return ops.silu_mul(gate, up)
    return F.silu(gate) * up


def _rope_qk(q, k, theta=10000.0):
    """_rope of q and k [B, H, S, D]; when both are adjacent slices of ONE fused q || k || v output (the QKV linear) the two rotations are one kernel launch over the
    [B, S, Hq + Hk, D] view of that buffer (a decode step saves a launch per layer; same bytes at prefill sizes)."""
    d = q.shape[-1]
    if (ROPE_KERNEL and q.is_cuda and q.dtype in (torch.float16, torch.bfloat16) and d % 16 == 0 and q.dim() == 4 and k.dim() == 4 and q.dtype == k.dtype
            and q.shape[0] == k.shape[0] and q.shape[2] == k.shape[2] and k.shape[3] == d):
        qb, kb = q.transpose(1, 2), k.transpose(1, 2)      # [B, S, H, D] views of the projection output
        B_, S_, Hq, _ = qb.shape
        Hk = kb.shape[2]
        st, sk = qb.stride(), kb.stride()
        # the row pitch of the fused buffer (strides of size-1 dimensions carry no information: a [1, 1, H, D] view of a slice reports a dense pitch)
        if S_ > 1:
            ld, ok = st[1], (sk[1] == st[1] and (B_ == 1 or (st[0] == S_ * st[1] and sk[0] == st[0])))
        elif B_ > 1:
            ld, ok = st[0], sk[0] == st[0]
        else:
            ld, ok = (Hq + Hk) * d, True
        if (ok and st[3] == 1 and sk[3] == 1 and st[2] == d and sk[2] == d and ld >= (Hq + Hk) * d and ld % 8 == 0
                and kb.data_ptr() == qb.data_ptr() + Hq * d * q.element_size() and qb.data_ptr() % 16 == 0):
            from . import ops
            cos, sin = _rope_tables(S_, d, q.device, q.dtype, theta)
            both = torch.as_strided(qb, (B_, S_, Hq + Hk, d), (S_ * ld, ld, d, 1))
            out = ops.rope(both, cos.view(-1, d // 2), sin.view(-1, d // 2))
            return out[:, :, :Hq].transpose(1, 2), out[:, :, Hq:].transpose(1, 2)
    return _rope(q, theta), _rope(k, theta)


def _rope_torch(x, theta=10000.0):
    """The same embedding as torch ops: two fused multiply-adds per half written straight into the output (no concatenation pass), tables cached per (S, D)."""
    d = x.shape[-1]
    cos, sin = _rope_tables(x.shape[-2], d, x.device, x.dtype, theta)
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    out = torch.empty_like(x)
    torch.addcmul(x1 * cos, x2, sin, value=-1, out=out[..., : d // 2])
    torch.addcmul(x2 * cos, x1, sin, out=out[..., d // 2:])
    return out


SHARED_INPUT_MIN_ELEMS = 1 << 20   # below this the single fused C-ABI call per module is cheaper than a separate quantiser launch (host-bound regime)


def shared_input(x, *mods):
    """One explicit activation quantisation for modules that read the SAME tensor (q/k/v, gate/up): the reference quantises it inside each
    forward (layers/nn/linear.py:88-96), three HBM passes over one hidden state.  Returns a QuantizedActivation when `x` is a float tensor
    large enough to be bandwidth-bound and every module has the same input contract (`input_signature()`), else `x` unchanged.  No caching, no
    identity or version heuristics: the caller states that the modules share the input by calling this."""
    if not isinstance(x, torch.Tensor) or x.numel() < SHARED_INPUT_MIN_ELEMS or len(mods) < 2:
        return x
    sigs = {m.input_signature() if hasattr(m, "input_signature") else None for m in mods}
    if len(sigs) != 1 or None in sigs:
        return x
    return mods[0].quantize_input(x, consumers=mods)


# ---- independent linears of one input on side streams (measured, OFF by default) ----------------------------------------------------------
# q/k/v (gate/up) read one activation and write three (two) outputs: nothing orders them but the stream.  On one stream every GEMM ends with a
# tail (the last tiles draining) and the next one starts with a ramp -- 5-6 us launch to launch at 4096^3 (profiles/r2_clock_power_evidence.md).
# Forking the 2nd / 3rd module onto side streams and joining afterwards was VERDICT r2's lever 2(c); measured (profiles/r3_gemm_levers.md): the
# default step goes from 0.2445 to 0.2868 ms (-15 %): three 256-block GEMMs that each want every CU's whole LDS interleave block by block and
# the fork / join events cost more than the two gaps.  Kept behind ASQ_SIDE_STREAMS=1 for the A/B; results are identical either way.
import os as _os

SIDE_STREAM_MIN_ROWS = 1024     # below this the GEMMs are short and the fork / join events cost more than the gaps they hide
side_streams_enabled = _os.environ.get("ASQ_SIDE_STREAMS", "0") == "1"
_SIDE = {}


def concurrent_linears(mods, xi):
    """[m(xi) for m in mods] with modules 2.. running on side streams (fork after `xi` is ready on the current stream, join before returning)."""
    t = xi.xq if hasattr(xi, "xq") else xi
    if (not side_streams_enabled or len(mods) < 2 or not isinstance(t, torch.Tensor) or not t.is_cuda
            or t.numel() // t.shape[-1] < SIDE_STREAM_MIN_ROWS):
        return [m(xi) for m in mods]
    cur = torch.cuda.current_stream(t.device)
    pool = _SIDE.setdefault((t.device.index, cur.cuda_stream), [])
    while len(pool) < len(mods) - 1:
        pool.append(torch.cuda.Stream(device=t.device))
    fork = torch.cuda.Event()
    fork.record(cur)
    outs = [None] * len(mods)
    for i, m in enumerate(mods[1:], 1):
        s = pool[i - 1]
        s.wait_event(fork)
        with torch.cuda.stream(s):
            outs[i] = m(xi)
        outs[i].record_stream(cur)     # allocated on the side stream's pool, consumed (and later freed) on the current stream's timeline
    outs[0] = mods[0](xi)
    for s in pool[:len(mods) - 1]:
        cur.wait_stream(s)
    return outs


class LlamaLayer(torch.nn.Module):
    """Float decoder layer (pre-norm, MHA/GQA with RoPE, SiLU-gated MLP)."""

    rope_theta = 10000.0   # config.json "rope_theta" (HF LlamaConfig default); checkpoint.load_reference_checkpoint sets it per model

    def __init__(self, hidden=4096, inter=11008, heads=32, kv_heads=None, eps=1e-5, rope_theta=10000.0):
        super().__init__()
        kv_heads = kv_heads or heads
        self.hidden, self.heads, self.kv_heads, self.hd = hidden, heads, kv_heads, hidden // heads
        self.rope_theta = float(rope_theta)
        L = torch.nn.Linear
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(hidden, eps), RMSNorm(hidden, eps)
        self.q_proj, self.k_proj, self.v_proj = L(hidden, heads * self.hd, bias=False), L(hidden, kv_heads * self.hd, bias=False), L(hidden, kv_heads * self.hd, bias=False)
        self.o_proj = L(heads * self.hd, hidden, bias=False)
        self.gate_proj, self.up_proj, self.down_proj = L(hidden, inter, bias=False), L(hidden, inter, bias=False), L(inter, hidden, bias=False)

    def attention(self, h, record=None):
        B, S, _ = (h.base if isinstance(h, DeferredResidual) else h).shape
        alt = getattr(self, "use_fused", False)   # to_w8a8(both=True): the fused (N1) modules sit beside the reference composition
        norm = self.input_layernorm_q if alt else self.input_layernorm
        one_qkv = getattr(self, "qkv_proj", None) is not None and (alt or getattr(self, "q_proj", None) is None)
        readers = (self.qkv_proj,) if one_qkv else (self.q_proj, self.k_proj, self.v_proj)   # (a fused norm emits the activation's offset image when all of them take it)
        if isinstance(h, DeferredResidual):       # the previous layer left its residual add to this layer's input norm
            if hasattr(norm, "add_forward"):
                h, x = norm.add_forward(h.delta, h.base, consumers=readers)
            else:
                h = h.materialize()
                x = norm(h)
        else:
            x = norm(h, consumers=readers) if hasattr(norm, "add_forward") else norm(h)   # a float tensor, or a QuantizedActivation when the norm is fused (N1)
        if record is not None:
            record["attn_in"] = x
        if one_qkv:   # one GEMM over [q;k;v] (the reference's QKVLinear, as its Baichuan W_pack)
            nq, nkv = self.heads * self.hd, self.kv_heads * self.hd
            q, k, v = self.qkv_proj(x).split([nq, nkv, nkv], dim=-1)
        else:
            xi = shared_input(x, self.q_proj, self.k_proj, self.v_proj)
            q, k, v = concurrent_linears([self.q_proj, self.k_proj, self.v_proj], xi)
        q = q.view(B, S, self.heads, self.hd).transpose(1, 2)
        k = k.view(B, S, self.kv_heads, self.hd).transpose(1, 2)
        v = v.view(B, S, self.kv_heads, self.hd).transpose(1, 2)
        q, k = _rope_qk(q, k, self.rope_theta)
        if self.kv_heads != self.heads:
            r = self.heads // self.kv_heads
            k, v = k.repeat_interleave(r, dim=1), v.repeat_interleave(r, dim=1)
        a = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B, S, self.heads * self.hd)
        if record is not None:
            record["o_in"] = a
        o = self.o_proj(a)
        if getattr(self, "defer_residual", False):   # h + o is added inside the post-attention norm (asq_add_norm_quantize)
            return DeferredResidual(h, o)
        return h + o

    def mlp(self, h, record=None):
        alt = getattr(self, "use_fused", False)
        norm = self.post_attention_layernorm_q if alt else self.post_attention_layernorm
        readers = (self.gate_proj, self.up_proj)
        gu = getattr(self, "gate_up", None) if ((alt or getattr(self, "fuse_act", False)) and getattr(self, "fuse_gate_up", True)) else None
        if gu is not None:
            hb = h.base if isinstance(h, DeferredResidual) else h
            if gu.supported(hb.numel() // hb.shape[-1], hb.dtype):
                readers = (gu,)   # gate || up as ONE GEMM with the SiLU * up epilogue (layers/nn/fused.py::GateUpSiLU)
            else:
                gu = None
        if isinstance(h, DeferredResidual):
            if hasattr(norm, "add_forward"):
                h, x = norm.add_forward(h.delta, h.base, consumers=readers)
            else:
                h = h.materialize()
                x = norm(h)
        else:
            x = norm(h, consumers=readers) if hasattr(norm, "add_forward") else norm(h)
        if record is not None:
            record["mlp_in"] = x
        if gu is not None:
            from .layers.nn.fused import QuantizedActivation
            qa = x if isinstance(x, QuantizedActivation) else self.gate_proj.quantize_input(x, consumers=(gu,))
            a = gu(qa, fast=getattr(self, "fast_silu", None), consumer=self.down_proj)   # (a per-tensor down_proj gets int8 straight from the epilogue)
            if a is not None:
                d = self.down_proj(a)
                return DeferredResidual(h, d) if getattr(self, "defer_residual", False) else h + d
            x = qa
        xi = shared_input(x, self.gate_proj, self.up_proj)
        gate, up = concurrent_linears([self.gate_proj, self.up_proj], xi)
        if getattr(self, "fuse_act", False) or alt:  # SiLU * up quantised in one pass; the fp product never reaches HBM
            from .layers.nn.fused import silu_mul_q
            d = self.down_proj(silu_mul_q(gate, up, self.down_proj, fast=getattr(self, "fast_silu", None)))
            return DeferredResidual(h, d) if getattr(self, "defer_residual", False) else h + d
        g = _silu_mul(gate, up)
        if record is not None:
            record["down_in"] = g
        return h + self.down_proj(g)

    def forward(self, h, record=None):
        return self.mlp(self.attention(h, record), record)


@torch.no_grad()
def init_llama_layer(layer, std=0.02, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, p in layer.named_parameters():
        if p.dim() == 2:
            p.copy_(torch.randn(p.shape, generator=g) * std)
    return layer


@torch.no_grad()
def calibrate(layer, h):
    """Per-tensor absmax/127 input scales of the four linear groups on a calibration batch
    (what quantize/calibration.py:185-244 collects with forward hooks)."""
    rec = {}
    layer(h, rec)
    return {k: float(v.abs().max()) / 127.0 for k, v in rec.items()}


@torch.no_grad()
def refresh_derived_operands(root):
    """After ANY in-place weight update, before replaying a hipGraph captured on `root` (a layer, a stack of layers, a model): rebuild every operand DERIVED from
    a weight INTO its existing buffers -- the linears' offset images (_W8A8Base.refresh_offset_image), GateUpSiLU's interleaved gate || up operand and its image
    (GateUpSiLU.refresh), MixtralLayer's stack images and grouped gate || up operand (MixtralLayer.refresh_offset_images).  A graph bakes those buffers in; an eager
    forward notices a changed weight by itself (storage + version key), a replay cannot.  Returns the number of operands rebuilt."""
    from .layers.nn.fused import GateUpSiLU
    from .layers.nn.linear import _W8A8Base
    n = 0
    for m in root.modules():
        if isinstance(m, _W8A8Base):
            n += bool(m.refresh_offset_image())
        elif isinstance(m, GateUpSiLU):
            n += bool(m.refresh())
        if isinstance(m, MixtralLayer) and hasattr(m, "_w1_stack"):
            m.refresh_offset_images()
            n += 1
    return n


@torch.no_grad()
def qkv_from_parts(q_proj, k_proj, v_proj):
    """One W8A8BFP32OFP32QKVLinear over three already-quantised projections: the int8 rows are concatenated as they are
    and each segment keeps its own dequant scale -- exactly what QKVLinear.from_float produces from the concatenated
    float weight (per-segment absmax scales, reference linear.py:226-232)."""
    from .layers.nn.linear import W8A8BFP32OFP32QKVLinear
    sizes = [q_proj.out_features, k_proj.out_features, v_proj.out_features]
    parts = (q_proj, k_proj, v_proj)
    has_bias = [bool(p.use_bias) for p in parts]
    if any(has_bias) and not all(has_bias):
        raise ValueError("qkv_from_parts: either all of q/k/v carry a bias or none (a missing one would silently become zeros)")
    if len({p.act_quant for p in parts}) != 1 or len({p.in_features for p in parts}) != 1:
        raise ValueError("qkv_from_parts: q/k/v must share act_quant and in_features")
    m = W8A8BFP32OFP32QKVLinear(sizes, q_proj.in_features, sum(sizes), all(has_bias), q_proj.act_quant)
    m.weight = torch.cat([q_proj.weight, k_proj.weight, v_proj.weight], dim=0)
    if all(has_bias):   # OPT's biased projections: the fused module's bias is the three biases, concatenated (reference linear.py:236-238)
        m.bias = torch.cat([p.bias.detach().to(torch.float32) for p in parts])
    for name, part in zip(m._host_scalars, (q_proj, k_proj, v_proj)):
        setattr(m, name, part._buffers["dequant_scale"].detach().clone())
    return m.to(q_proj.weight.device)


@torch.no_grad()
def to_w8a8(layer, scales, quant_config=None, fuse_norm=False, fuse_qkv=False, both=False):
    """Quantised copy of `layer`, composed exactly like the reference's
    QuantizedLlamaDecoderLayer.from_float_to_int8 (models/llama.py:289-339):
      q/k/v, gate/up : W8A8BFP32OFP32Linear(act_quant = cfg["qkv"] / cfg["fc1"]), norm weight folded iff per-tensor
      o, down        : W8A8BFP32OFP32LinearWithQuantScale(act_quant = cfg["out"] / cfg["fc2"]).
    both=True keeps that composition AND attaches the fused (N1) modules beside it -- RMSNormQ twins of the two norms and a
    QKVLinear over the same int8 rows; `layer.use_fused = True/False` selects the path per forward (bench.py times both
    on one set of weights)."""
    cfg = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    cfg.update(quant_config or {})
    dev = layer.q_proj.weight.device
    q = LlamaLayer.__new__(LlamaLayer)
    torch.nn.Module.__init__(q)
    q.hidden, q.heads, q.kv_heads, q.hd, q.rope_theta = layer.hidden, layer.heads, layer.kv_heads, layer.hd, layer.rope_theta
    q.calib_scales, q.quant_config = dict(scales), dict(cfg)   # (kept for all_per_tensor_view)

    def conv(lin, cls, scale, aq):
        src = torch.nn.Linear(lin.in_features, lin.out_features, bias=False)
        src.weight = torch.nn.Parameter(lin.weight.detach().float().clone())  # from_float rounds an fp32 source in place
        return cls.from_float(src, scale, save_device=dev, act_quant=aq).to(dev)

    if fuse_qkv:   # q/k/v as ONE W8A8BFP32OFP32QKVLinear (per-segment weight scales, linear.py:210-245): one quantise + one GEMM
        from .layers.nn.linear import W8A8BFP32OFP32QKVLinear
        cat = torch.nn.Linear(layer.q_proj.in_features, layer.q_proj.out_features + 2 * layer.k_proj.out_features, bias=False)
        cat.weight = torch.nn.Parameter(torch.cat([layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight]).detach().float().clone())
        sizes = [layer.q_proj.out_features, layer.k_proj.out_features, layer.v_proj.out_features]
        q.qkv_proj = W8A8BFP32OFP32QKVLinear.from_float(cat, scales["attn_in"], sizes, save_device=dev, act_quant=cfg["qkv"]).to(dev)
    else:
        q.q_proj = conv(layer.q_proj, W8A8BFP32OFP32Linear, scales["attn_in"], cfg["qkv"])
        q.k_proj = conv(layer.k_proj, W8A8BFP32OFP32Linear, scales["attn_in"], cfg["qkv"])
        q.v_proj = conv(layer.v_proj, W8A8BFP32OFP32Linear, scales["attn_in"], cfg["qkv"])
    q.o_proj = conv(layer.o_proj, W8A8BFP32OFP32LinearWithQuantScale, scales["o_in"], cfg["out"])
    q.gate_proj = conv(layer.gate_proj, W8A8BFP32OFP32Linear, scales["mlp_in"], cfg["fc1"])
    q.up_proj = conv(layer.up_proj, W8A8BFP32OFP32Linear, scales["mlp_in"], cfg["fc1"])
    q.down_proj = conv(layer.down_proj, W8A8BFP32OFP32LinearWithQuantScale, scales["down_in"], cfg["fc2"])
    if fuse_norm:  # the norms emit int8 directly; q/k/v and gate/up share one quantised activation
        from .layers.nn.fused import RMSNormQ
        q.input_layernorm = RMSNormQ.from_float(layer.input_layernorm, scales["attn_in"], per_token=cfg["qkv"] == "per-token")
        q.post_attention_layernorm = RMSNormQ.from_float(layer.post_attention_layernorm, scales["mlp_in"], per_token=cfg["fc1"] == "per-token")
        q.fuse_act = True
        from .layers.nn.fused import GateUpSiLU
        q.gate_up = GateUpSiLU(q.gate_proj, q.up_proj)   # (round 5) gate || up as one GEMM with the SiLU * up epilogue, where the shape runs on it
        return q
    q.input_layernorm = layer.input_layernorm.folded(scales["attn_in"]) if cfg["qkv"] == "per-tensor" else layer.input_layernorm
    q.post_attention_layernorm = layer.post_attention_layernorm.folded(scales["mlp_in"]) if cfg["fc1"] == "per-tensor" else layer.post_attention_layernorm
    if both:
        from .layers.nn.fused import RMSNormQ
        q.input_layernorm_q = RMSNormQ.from_float(layer.input_layernorm, scales["attn_in"], per_token=cfg["qkv"] == "per-token")
        q.post_attention_layernorm_q = RMSNormQ.from_float(layer.post_attention_layernorm, scales["mlp_in"], per_token=cfg["fc1"] == "per-token")
        if not fuse_qkv:
            q.qkv_proj = qkv_from_parts(q.q_proj, q.k_proj, q.v_proj)
        from .layers.nn.fused import GateUpSiLU
        q.gate_up = GateUpSiLU(q.gate_proj, q.up_proj)   # (round 5) gate || up as one GEMM with the SiLU * up epilogue, where the shape runs on it
        q.use_fused = False
    return q


@torch.no_grad()
def per_tensor_twin(mod, input_scale):
    """The per-tensor W8A8BFP32OFP32LinearWithQuantScale that `from_float(..., act_quant="per-tensor")` would have produced from the float weight behind the
    per-token module `mod` -- over the SAME int8 buffer (weight quantisation does not depend on act_quant, reference linear.py:304-329): quant_scale =
    input_scale, dequant_scale = input_scale * weight_scale (mod's own dequant_scale IS the weight scale)."""
    from .layers.nn.linear import W8A8BFP32OFP32LinearWithQuantScale
    if mod.act_quant != "per-token":
        raise ValueError("per_tensor_twin: expected a per-token module")
    t = W8A8BFP32OFP32LinearWithQuantScale(mod.in_features, mod.out_features, mod.use_bias, "per-tensor")
    t.weight = mod._buffers["weight"]
    if mod.use_bias:
        t.bias = mod._buffers["bias"]
    wscale = mod._buffers["dequant_scale"].detach().to("cpu", torch.float32)
    t.dequant_scale = (input_scale * wscale).to(torch.float32)
    t.quant_scale = torch.tensor(input_scale, dtype=torch.float32)
    t._pin_scalars()
    if "_offset_cache" in mod.__dict__:   # one weight, one image
        t.__dict__["_offset_cache"] = mod.__dict__["_offset_cache"]
    return t


@torch.no_grad()
def all_per_tensor_view(qlayer):
    """A second LlamaLayer object over the SAME quantised weights as `qlayer` (to_w8a8 with the default config) whose o_proj / down_proj quantise their inputs
    per TENSOR with the calibrated scales: the "linears all per-tensor" composition BASELINE configs[2] / SURVEY 8d cfg3 name (quant_config = {qkv, out, fc1,
    fc2: per-tensor}, reference models/llama.py:289-339).  Shares every module except the two twins; `use_fused` etc. are per view."""
    import copy
    v = copy.copy(qlayer)
    v._modules = dict(qlayer._modules)
    v.__dict__.pop("_asq_arena", None)
    v.o_proj = per_tensor_twin(qlayer.o_proj, qlayer.calib_scales["o_in"])
    v.down_proj = per_tensor_twin(qlayer.down_proj, qlayer.calib_scales["down_in"])
    v.quant_config = dict(qlayer.quant_config, out="per-tensor", fc2="per-tensor")
    return v


# ---------------------------------------------------------------------------------------------
# Baichuan-architecture layer: the one decoder-layer composition the reference can execute without the
# pinned HF version (models/baichuan.py), hence the one pinned by golden vectors (tests/golden/g7_block.npz).
# ---------------------------------------------------------------------------------------------
class BaichuanRMSNorm(torch.nn.Module):
    """thirdparty/baichuan/modeling_baichuan.py:161-176: x * rsqrt(mean(x^2) + eps), cast to the weight's
    half dtype if it has one, times weight."""

    def __init__(self, hidden, eps=1e-6):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(hidden))
        self.epsilon = eps

    def forward(self, x):
        v = x * torch.rsqrt(x.to(torch.float32).pow(2).mean(-1, keepdim=True) + self.epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            v = v.to(self.weight.dtype)
        return self.weight * v


class BaichuanLayer(torch.nn.Module):
    """Float layer: pre-norm, fused W_pack QKV projection, eager softmax attention (no mask: the "ALIBI" branch
    with attention_mask=None, the only one the reference's int8 layer runs on CPU), SiLU-gated MLP."""

    def __init__(self, hidden=256, inter=704, heads=4, eps=1e-6):
        super().__init__()
        self.hidden, self.heads, self.hd = hidden, heads, hidden // heads
        L = torch.nn.Linear
        self.input_layernorm, self.post_attention_layernorm = BaichuanRMSNorm(hidden, eps), BaichuanRMSNorm(hidden, eps)
        self.W_pack, self.o_proj = L(hidden, 3 * hidden, bias=False), L(hidden, hidden, bias=False)
        self.gate_proj, self.up_proj, self.down_proj = L(hidden, inter, bias=False), L(hidden, inter, bias=False), L(inter, hidden, bias=False)
        self.int8 = False

    def _attend(self, proj):
        B, S, _ = proj.shape
        q, k, v = [t.view(B, S, self.heads, self.hd).transpose(1, 2) for t in proj.split(self.hidden, dim=-1)]
        w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(self.hd)
        w = torch.nn.functional.softmax(w, dim=-1).to(v.dtype)
        return torch.matmul(w, v).transpose(1, 2).reshape(B, S, self.hidden)

    def forward(self, h, record=None):
        x = self.input_layernorm(h)
        if record is not None:
            record["attn_in"] = x
        proj = self.W_pack(x)
        if self.int8:
            proj = proj.to(torch.float16)  # models/baichuan.py:120
        a = self._attend(proj)
        if record is not None:
            record["o_in"] = a
        h = h + self.o_proj(a).to(h.dtype)
        x = self.post_attention_layernorm(h)
        if record is not None:
            record["mlp_in"] = x
        xi = shared_input(x, self.gate_proj, self.up_proj)
        g = self.gate_proj(xi)
        if self.int8:
            g = g.to(torch.float16)        # models/baichuan.py:227
        g = F.silu(g) * self.up_proj(xi)
        if record is not None:
            record["down_in"] = g
        return h + self.down_proj(g).to(h.dtype)


@torch.no_grad()
def to_w8a8_baichuan(layer, scales, quant_config=None):
    """Quantised copy composed like Int8BaichuanLayer.from_float (models/baichuan.py:258-287):
    W_pack -> W8A8BFP32OFP32QKVLinear (three per-segment scales), gate/up -> W8A8BFP32OFP32Linear,
    o/down -> W8A8BFP32OFP32LinearWithQuantScale; norm weights divided by the input scale iff per-tensor."""
    from .layers.nn.linear import W8A8BFP32OFP32QKVLinear
    cfg = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    cfg.update(quant_config or {})
    dev = layer.W_pack.weight.device
    q = BaichuanLayer.__new__(BaichuanLayer)
    torch.nn.Module.__init__(q)
    q.hidden, q.heads, q.hd, q.int8 = layer.hidden, layer.heads, layer.hd, True

    def src(lin):
        s = torch.nn.Linear(lin.in_features, lin.out_features, bias=False)
        s.weight = torch.nn.Parameter(lin.weight.detach().float().clone())  # from_float rounds an fp32 source in place
        return s

    q.W_pack = W8A8BFP32OFP32QKVLinear.from_float(src(layer.W_pack), scales["attn_in"], [layer.hidden] * 3, save_device=dev, act_quant=cfg["qkv"]).to(dev)
    q.o_proj = W8A8BFP32OFP32LinearWithQuantScale.from_float(src(layer.o_proj), scales["o_in"], save_device=dev, act_quant=cfg["out"]).to(dev)
    q.gate_proj = W8A8BFP32OFP32Linear.from_float(src(layer.gate_proj), scales["mlp_in"], save_device=dev, act_quant=cfg["fc1"]).to(dev)
    q.up_proj = W8A8BFP32OFP32Linear.from_float(src(layer.up_proj), scales["mlp_in"], save_device=dev, act_quant=cfg["fc1"]).to(dev)
    q.down_proj = W8A8BFP32OFP32LinearWithQuantScale.from_float(src(layer.down_proj), scales["down_in"], save_device=dev, act_quant=cfg["fc2"]).to(dev)

    def norm(n, scale, fold):
        m = BaichuanRMSNorm(n.weight.numel(), n.epsilon)
        m.weight = torch.nn.Parameter(n.weight.detach() / scale if fold else n.weight.detach().clone())
        return m.to(dev)

    q.input_layernorm = norm(layer.input_layernorm, scales["attn_in"], cfg["qkv"] == "per-tensor")
    q.post_attention_layernorm = norm(layer.post_attention_layernorm, scales["mlp_in"], cfg["fc1"] == "per-tensor")
    return q


# ---------------------------------------------------------------------------------------------
# OPT-architecture layer (reference models/opt.py): LayerNorm with weight AND bias folded (:20-29), biased q/k/v/out_proj
# (:78-81), fc1 -> ReLU -> fc2 (:127-128).  The reference borrows OPTDecoderLayer.forward from HF (:131); restated here:
# pre-LN (do_layer_norm_before, every size but 350m) or post-LN, q scaled by head_dim^-0.5 BEFORE the score product, causal mask.
# ---------------------------------------------------------------------------------------------
class OptLayer(torch.nn.Module):
    def __init__(self, hidden=5120, ffn=20480, heads=40, eps=1e-5, bias=True, do_layer_norm_before=True):
        super().__init__()
        self.hidden, self.heads, self.hd, self.pre_ln = hidden, heads, hidden // heads, do_layer_norm_before
        L = torch.nn.Linear
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (L(hidden, hidden, bias=bias) for _ in range(4))
        self.fc1, self.fc2 = L(hidden, ffn, bias=bias), L(ffn, hidden, bias=bias)
        self.self_attn_layer_norm, self.final_layer_norm = torch.nn.LayerNorm(hidden, eps), torch.nn.LayerNorm(hidden, eps)

    def forward(self, h, record=None):
        B, S, _ = h.shape
        res = h
        ln = self.self_attn_layer_norm
        fusedq = hasattr(ln, "add_forward")                       # a LayerNormQ (N1): returns a QuantizedActivation -- the offset image when q/k/v all take it
        x = (ln(h, consumers=(self.q_proj, self.k_proj, self.v_proj)) if fusedq else ln(h)) if self.pre_ln else h
        if record is not None:
            record["attn_in"] = x
        xi = shared_input(x, self.q_proj, self.k_proj, self.v_proj)
        q = self.q_proj(xi) * (self.hd ** -0.5)                  # OPTAttention: query_states = q_proj(x) * scaling
        k, v = self.k_proj(xi), self.v_proj(xi)
        q, k, v = (t.view(B, S, self.heads, self.hd).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=1.0).transpose(1, 2).reshape(B, S, self.hidden)
        if record is not None:
            record["o_in"] = a
        h = res + self.out_proj(a)
        if not self.pre_ln:
            h = self.self_attn_layer_norm(h)
        res = h
        one_launch = record is None and hasattr(self.fc1, "forward_q") and getattr(self.fc2, "act_quant", None) == "per-tensor"
        fl = self.final_layer_norm
        x = ((fl(h, consumers=() if one_launch else (self.fc1,)) if hasattr(fl, "add_forward") else fl(h))) if self.pre_ln else h   # (forward_q's int8-out GEMM takes plain operands)
        if record is not None:
            record["fc1_in"] = x
        if one_launch:
            y = self.fc2(self.fc1.forward_q(x, self.fc2, act="relu"))   # fc1 + ReLU + fc2's prologue in ONE launch (asq_linear_w8a8_q8): bit-identical
        else:
            g = F.relu(self.fc1(x))
            if record is not None:
                record["fc2_in"] = g
            y = self.fc2(g)
        h = res + y
        return h if self.pre_ln else self.final_layer_norm(h)


@torch.no_grad()
def to_w8a8_opt(layer, scales, quant_config=None, fuse_norm=False):
    """Quantised copy composed like Int8OPTDecoderLayer.from_float (models/opt.py:133-164): q/k/v, fc1 -> W8A8BFP32OFP32Linear(cfg qkv / fc1),
    out_proj, fc2 -> W8A8BFP32OFP32LinearWithQuantScale(cfg out / fc2), biases kept, LayerNorm weight AND bias divided by the input scale iff the
    consumers are per-tensor.  scales: attn_in, o_in, fc1_in, fc2_in.  fuse_norm: the two LayerNorms become LayerNormQ (emit int8 directly, N1)."""
    cfg = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    cfg.update(quant_config or {})
    dev = layer.q_proj.weight.device
    q = OptLayer.__new__(OptLayer)
    torch.nn.Module.__init__(q)
    q.hidden, q.heads, q.hd, q.pre_ln = layer.hidden, layer.heads, layer.hd, layer.pre_ln

    def conv(lin, cls, scale, aq):
        src = torch.nn.Linear(lin.in_features, lin.out_features, bias=lin.bias is not None)
        src.weight = torch.nn.Parameter(lin.weight.detach().float().clone())  # from_float rounds an fp32 source in place
        if lin.bias is not None:
            src.bias = torch.nn.Parameter(lin.bias.detach().float().clone())
        return cls.from_float(src, scale, save_device=dev, act_quant=aq).to(dev)

    for n in ("q_proj", "k_proj", "v_proj"):
        setattr(q, n, conv(getattr(layer, n), W8A8BFP32OFP32Linear, scales["attn_in"], cfg["qkv"]))
    q.out_proj = conv(layer.out_proj, W8A8BFP32OFP32LinearWithQuantScale, scales["o_in"], cfg["out"])
    q.fc1 = conv(layer.fc1, W8A8BFP32OFP32Linear, scales["fc1_in"], cfg["fc1"])
    q.fc2 = conv(layer.fc2, W8A8BFP32OFP32LinearWithQuantScale, scales["fc2_in"], cfg["fc2"])

    def norm(n, scale, aq):
        if fuse_norm:
            from .layers.nn.fused import LayerNormQ
            return LayerNormQ.from_float(n, scale, per_token=aq == "per-token")
        m = torch.nn.LayerNorm(n.normalized_shape[0], n.eps).to(dev, n.weight.dtype)
        s = scale if aq == "per-tensor" else 1.0
        m.weight, m.bias = torch.nn.Parameter(n.weight.detach() / s), torch.nn.Parameter(n.bias.detach() / s)
        return m
    q.self_attn_layer_norm = norm(layer.self_attn_layer_norm, scales["attn_in"], cfg["qkv"])
    q.final_layer_norm = norm(layer.final_layer_norm, scales["fc1_in"], cfg["fc1"])
    return q


# ---------------------------------------------------------------------------------------------
# Mixtral-architecture layer (reference models/mixtral.py): LLaMA-style attention (GQA, RoPE with the config's rope_theta) + a sparse MoE block:
# float router `gate` (:137, "we do not apply quant to gate"), experts w1 / w3 = W8A8BFP32OFP32Linear(fc1), w2 = ...WithQuantScale(fc2) (:99-101).
# The reference borrows MixtralSparseMoeBlock.forward (:145): softmax in fp32 -> top-k -> renormalise -> a Python loop of per-expert module
# calls -> index_add.  Here the routed rows are sorted by expert once and each projection of ALL experts is ONE grouped launch
# (asq_linear_w8a8_grouped, bit-identical to the per-expert calls) whenever w2 is per-token; per-tensor w2 (one calibrated scale PER expert)
# falls back to the per-expert loop.
# ---------------------------------------------------------------------------------------------
class ExpertMLP(torch.nn.Module):
    """MixtralBlockSparseTop2MLP: w2(silu(w1 x) * w3 x).  Float linears when built with sizes; `ExpertMLP()` is an empty shell the quantised
    modules are attached to (to_w8a8_mixtral, checkpoint loader)."""

    def __init__(self, hidden=None, inter=None):
        super().__init__()
        if hidden is not None:
            L = torch.nn.Linear
            self.w1, self.w2, self.w3 = L(hidden, inter, bias=False), L(inter, hidden, bias=False), L(hidden, inter, bias=False)

    def forward(self, x):
        return self.w2(F.silu(self.w1(x)) * self.w3(x))


class MixtralLayer(LlamaLayer):
    def __init__(self, hidden=4096, inter=14336, heads=32, kv_heads=8, experts=8, top_k=2, eps=1e-5, rope_theta=1e6):
        torch.nn.Module.__init__(self)
        self.hidden, self.heads, self.kv_heads, self.hd, self.rope_theta = hidden, heads, kv_heads, hidden // heads, float(rope_theta)
        self.top_k = top_k
        L = torch.nn.Linear
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(hidden, eps), RMSNorm(hidden, eps)
        self.q_proj, self.k_proj, self.v_proj = L(hidden, heads * self.hd, bias=False), L(hidden, kv_heads * self.hd, bias=False), L(hidden, kv_heads * self.hd, bias=False)
        self.o_proj = L(heads * self.hd, hidden, bias=False)
        self.gate = L(hidden, experts, bias=False)
        self.experts = torch.nn.ModuleList([ExpertMLP(hidden, inter) for _ in range(experts)])

    def route(self, x2):
        w = F.softmax(self.gate(x2), dim=1, dtype=torch.float)
        w, sel = torch.topk(w, self.top_k, dim=-1)
        w = (w / w.sum(dim=-1, keepdim=True)).to(x2.dtype)
        flat = sel.reshape(-1)
        order = torch.sort(flat, stable=True).indices          # routed (token, slot) pairs sorted by expert, token order kept inside an expert
        counts = torch.bincount(flat, minlength=len(self.experts))
        return order // self.top_k, w.reshape(-1)[order], counts

    @torch.no_grad()
    def stack_experts(self):
        """[E, N, K] stacks of the experts' int8 weights for the grouped launches; the per-expert modules' `weight` buffers become views of
        the stacks (one copy of every weight in memory), their scalar scales go into device vectors."""
        ex = self.experts
        if not hasattr(ex[0].w1, "input_signature"):
            return self
        for name in ("w1", "w3", "w2"):
            mods = [getattr(e, name) for e in ex]
            st = torch.stack([m.weight for m in mods]).contiguous()
            for i, m in enumerate(mods):
                m._buffers["weight"] = st[i]
            host = [float(m._buffers["dequant_scale"]) for m in mods]
            setattr(self, f"_{name}_stack", st)
            setattr(self, f"_{name}_scale", torch.tensor(host, dtype=torch.float32, device=st.device))
            self.__dict__[f"_{name}_scale_host"] = host
        return self

    def _stack_image(self, name):
        """(image int8 [E,N,K], col_off int32 [E,N,2]) of one expert stack, rebuilt when the stack is (keyed on its storage and version counter)"""
        from . import ops
        st = getattr(self, f"_{name}_stack")
        try:
            ver = st._version
        except RuntimeError:
            # a stack without a version counter (built under inference_mode): writes into it cannot be seen.  build_offset_images() is the opt-in -- the caller
            # then owns the refresh (refresh_offset_images), as with _W8A8Base.build_offset_image; otherwise plain operands (logged once)
            if not self.__dict__.get("_images_pinned", False):
                if not self.__dict__.get("_images_off_logged", False):
                    self.__dict__["_images_off_logged"] = True
                    import logging
                    logging.getLogger("autosmoothquant_amd").info("MixtralLayer: expert stacks have no version counter (inference tensors): offset images and the grouped "
                                                                  "gate || up launch are off; call build_offset_images() to opt in (then refresh_offset_images() after weight writes)")
                return None
            ver = -1
        key = (st.data_ptr(), ver, st.device)
        hit = self.__dict__.get(f"_{name}_image")
        if hit is None or hit[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None   # (see _W8A8Base.offset_image)
            E, N, K = st.shape
            out = None   # rebuild INTO the existing buffers when they still fit: a hipGraph captured on them then replays the new image
            if hit is not None and hit[1][0].device == st.device and tuple(hit[1][0].shape) == (E, N, K):
                out = (hit[1][0].view(E * N, K), hit[1][1].view(E * N, 2))
            try:
                img, col = ops.weight_offset_image(st.view(E * N, K), out=out)
            except torch.cuda.OutOfMemoryError:
                self.offsets = False
                self.__dict__.pop(f"_{name}_image", None)
                return None
            hit = (key, (img.view(E, N, K), col.view(E, N, 2)))
            self.__dict__[f"_{name}_image"] = hit
        return hit[1]

    def _w13_operand(self, want_image):
        """w1 || w3 of every expert interleaved in blocks of 16 channels, [E, 2F, K] (ops.interleave_gate_up_stack), and its offset image when asked for: the operand
        of the grouped gate || up launch (opt-in `fuse_gate_up`: one more int8 copy of the two stacks, plus the image).  Follows the stacks (storage + version)."""
        from . import ops
        s1, s3 = self._w1_stack, self._w3_stack
        try:
            key = (s1.data_ptr(), s1._version, s3.data_ptr(), s3._version, s1.device)
        except RuntimeError:
            if not self.__dict__.get("_images_pinned", False):
                return None, None
            key = (s1.data_ptr(), -1, s3.data_ptr(), -1, s1.device)
        hit = self.__dict__.get("_w13_cache")   # [key, operand, image or None, image is current]
        if hit is None or hit[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None, None
            # rebuild INTO the cached buffers when they still fit: a hipGraph captured on them then replays the new weights instead of freed memory
            E, F_, K = s1.shape
            fits = hit is not None and hit[1].device == s1.device and tuple(hit[1].shape) == (E, 2 * F_, K)
            try:
                w13 = ops.interleave_gate_up_stack(s1, s3, out=hit[1] if fits else None)
            except torch.cuda.OutOfMemoryError:
                self.fuse_gate_up = False
                self.__dict__.pop("_w13_cache", None)
                return None, None
            hit = [key, w13, hit[2] if fits else None, False]
            self.__dict__["_w13_cache"] = hit
        if want_image and not hit[3]:
            if torch.cuda.is_current_stream_capturing():
                return hit[1], None
            E, N2, K = hit[1].shape
            out = None if hit[2] is None else (hit[2][0].view(E * N2, K), hit[2][1])
            try:
                img, col = ops.weight_offset_image(hit[1].view(E * N2, K), out=out)
            except torch.cuda.OutOfMemoryError:
                return hit[1], None
            hit[2], hit[3] = (img.view(E, N2, K), col), True
        return hit[1], (hit[2] if (want_image and hit[3]) else None)

    def build_offset_images(self):
        """Build the three stacks' images (and, with fuse_gate_up, the w1 || w3 operand and its image) now -- load time -- instead of inside the first grouped forward,
        and opt in for stacks without a version counter (a model built under torch.inference_mode()): the caller then calls refresh_offset_images() after writing
        weights.  Mirrors _W8A8Base.build_offset_image.  Returns True when every image exists."""
        self.__dict__["_images_pinned"] = True
        ok = all(self._stack_image(n) is not None for n in ("w1", "w3", "w2"))
        if getattr(self, "fuse_gate_up", False):
            ok = self._w13_operand(True)[1] is not None and ok
        return ok

    def refresh_offset_images(self):
        """Rebuild the three stacks' images from the current stacks into their existing buffers (after a weight update, before replaying a hipGraph captured
        on them; see _W8A8Base.refresh_offset_image)."""
        for name in ("w1", "w3", "w2"):
            hit = self.__dict__.get(f"_{name}_image")
            if hit is not None:
                self.__dict__[f"_{name}_image"] = (None, hit[1])   # (key None never matches: the next _stack_image call rebuilds in place)
                self._stack_image(name)
        hit = self.__dict__.get("_w13_cache")   # the grouped gate || up operand and its image follow (same buffers: graphs captured on them stay valid)
        if hit is not None:
            had_image = hit[2] is not None
            hit[0] = None
            self._w13_operand(had_image)

    def _stacks_current(self):
        """The per-expert modules stay the source of truth (state_dict, load_state_dict, .to(), replica.broadcast_quantized all act on THEIR buffers): the
        stacks are current iff every expert's `weight` is still the stack's own row block and its dequant scale the recorded one.  Anything that re-homes
        a buffer (QuantArena.pack, Module.to) or rewrites a scale leaves a mismatch here, and moe() restacks before the grouped launch."""
        for name in ("w1", "w3", "w2"):
            st, host = getattr(self, f"_{name}_stack", None), self.__dict__.get(f"_{name}_scale_host")
            if st is None or host is None or st.shape[0] != len(self.experts):
                return False
            for i, e in enumerate(self.experts):
                m = getattr(e, name)
                w = m._buffers["weight"]
                if w.device != st.device or w.data_ptr() != st[i].data_ptr() or float(m._buffers["dequant_scale"]) != host[i]:
                    return False
        return True

    def moe(self, x):
        B, S, H = x.shape
        x2 = x.reshape(-1, H)
        tok, wts, counts = self.route(x2)
        xs = x2[tok]
        ex = self.experts
        grouped = (hasattr(self, "_w1_stack") and ex[0].w2.act_quant == "per-token" and xs.is_cuda
                   and H % 128 == 0 and self._w2_stack.shape[-1] % 128 == 0)   # (the grouped launch runs on the tiled kernel: K % 128 == 0)
        if grouped and not self._stacks_current():   # a buffer was re-homed or a scale rewritten since stack_experts(): rebuild from the per-expert modules
            self.stack_experts()
        if grouped:
            from . import ops
            offs = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
            mode, qs = ex[0].w1._input_mode()
            R = xs.shape[0]
            if getattr(self, "offsets", True) and ops.grouped_offsets_supported(R, self._w1_stack.shape[1], H, x.dtype) and ops.grouped_offsets_supported(R, H, self._w2_stack.shape[-1], x.dtype):
                # offset operand images (include/asq_hip.h): same products, less matrix-core energy; the stacks' images are built once and follow the stacks
                i1, i3, i2 = (self._stack_image(n) for n in ("w1", "w3", "w2"))
            else:
                i1 = i3 = i2 = None
            F2 = self._w1_stack.shape[1]
            gate_up = (getattr(self, "fuse_gate_up", False) and mode != "per-token" and ops.grouped_gate_up_supported(R, F2, H, x.dtype))
            if gate_up:
                # opt-in (round 5): w1 || w3 of all experts as ONE grouped launch whose epilogue writes SiLU(w1 x) * (w3 x) -- the fused-SiLU arithmetic (+-1 int8 of torch's
                # silu at rounding boundaries, like ops.silu_mul_quantize), which is why the reference composition below stays the default
                w13, img13 = self._w13_operand(i1 is not None and i2 is not None)
                gate_up = w13 is not None
            if gate_up:
                if img13 is not None and i2 is not None:
                    xq, _, ro = ops.quantize_act_off(xs.contiguous(), mode, qs)
                    a = ops.linear_w8a8_grouped_gate_up(xq, img13[0], offs, self._w1_scale, self._w3_scale, x.dtype, getattr(self, "fast_silu", None), ro, img13[1])
                    aq, arow, ao = ops.quantize_act_off(a, "per-token")
                    y = ops.linear_w8a8_grouped_off(aq, i2[0], ao, i2[1], offs, self._w2_scale, x.dtype, arow)
                else:
                    xq, _ = ops.quantize_act(xs.contiguous(), mode, qs)
                    a = ops.linear_w8a8_grouped_gate_up(xq, w13, offs, self._w1_scale, self._w3_scale, x.dtype, getattr(self, "fast_silu", None))
                    aq, arow = ops.quantize_act(a, "per-token")
                    y = ops.linear_w8a8_grouped(aq, self._w2_stack, offs, self._w2_scale, x.dtype, arow)
            elif i1 is not None and i3 is not None and i2 is not None:
                xq, srow, ro = ops.quantize_act_off(xs.contiguous(), mode, qs)
                h1 = ops.linear_w8a8_grouped_off(xq, i1[0], ro, i1[1], offs, self._w1_scale, x.dtype, srow)
                h3 = ops.linear_w8a8_grouped_off(xq, i3[0], ro, i3[1], offs, self._w3_scale, x.dtype, srow)
                aq, arow, ao = ops.quantize_act_off(F.silu(h1) * h3, "per-token")
                y = ops.linear_w8a8_grouped_off(aq, i2[0], ao, i2[1], offs, self._w2_scale, x.dtype, arow)
            else:
                xq, srow = ops.quantize_act(xs.contiguous(), mode, qs)
                h1 = ops.linear_w8a8_grouped(xq, self._w1_stack, offs, self._w1_scale, x.dtype, srow)
                h3 = ops.linear_w8a8_grouped(xq, self._w3_stack, offs, self._w3_scale, x.dtype, srow)
                aq, arow = ops.quantize_act(F.silu(h1) * h3, "per-token")
                y = ops.linear_w8a8_grouped(aq, self._w2_stack, offs, self._w2_scale, x.dtype, arow)
        else:   # the reference's loop (also the float layer's path)
            y, o = torch.empty_like(xs), 0
            for e, c in zip(ex, counts.tolist()):
                if c:
                    y[o:o + c] = e(xs[o:o + c])
                o += c
        out = torch.zeros_like(x2)
        out.index_add_(0, tok, y * wts[:, None])
        return out.view(B, S, H)

    def mlp(self, h, record=None):
        if isinstance(h, DeferredResidual):
            h = h.materialize()
        x = self.post_attention_layernorm(h)
        if record is not None:
            record["mlp_in"] = x
        return h + self.moe(x)


@torch.no_grad()
def to_w8a8_mixtral(layer, scales, quant_config=None):
    """Quantised copy composed like Int8MixtralDecoderLayer.from_float (models/mixtral.py:166-200): attention as LLaMA, every expert's w1 / w3 with
    the shared moe input scale, w2 with ITS OWN down input scale (`scales["down_in"]`: a list, one per expert), the router left in floating point,
    RMSNorm weights divided by the input scale iff per-tensor."""
    cfg = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    cfg.update(quant_config or {})
    dev = layer.q_proj.weight.device
    q = MixtralLayer.__new__(MixtralLayer)
    torch.nn.Module.__init__(q)
    q.hidden, q.heads, q.kv_heads, q.hd, q.rope_theta, q.top_k = layer.hidden, layer.heads, layer.kv_heads, layer.hd, layer.rope_theta, layer.top_k

    def conv(lin, cls, scale, aq):
        src = torch.nn.Linear(lin.in_features, lin.out_features, bias=False)
        src.weight = torch.nn.Parameter(lin.weight.detach().float().clone())
        return cls.from_float(src, scale, save_device=dev, act_quant=aq).to(dev)

    for n in ("q_proj", "k_proj", "v_proj"):
        setattr(q, n, conv(getattr(layer, n), W8A8BFP32OFP32Linear, scales["attn_in"], cfg["qkv"]))
    q.o_proj = conv(layer.o_proj, W8A8BFP32OFP32LinearWithQuantScale, scales["o_in"], cfg["out"])
    q.gate = layer.gate
    q.experts = torch.nn.ModuleList()
    for e, ds in zip(layer.experts, scales["down_in"]):
        m = ExpertMLP()
        m.w1 = conv(e.w1, W8A8BFP32OFP32Linear, scales["mlp_in"], cfg["fc1"])
        m.w3 = conv(e.w3, W8A8BFP32OFP32Linear, scales["mlp_in"], cfg["fc1"])
        m.w2 = conv(e.w2, W8A8BFP32OFP32LinearWithQuantScale, ds, cfg["fc2"])
        q.experts.append(m)
    q.input_layernorm = layer.input_layernorm.folded(scales["attn_in"]) if cfg["qkv"] == "per-tensor" else layer.input_layernorm
    q.post_attention_layernorm = layer.post_attention_layernorm.folded(scales["mlp_in"]) if cfg["fc1"] == "per-tensor" else layer.post_attention_layernorm
    return q.stack_experts()
