"""Norm -> int8 fusion: the built counterpart of the reference's ``LayerNormQ``
(reference autosmoothquant/layers/nn/fused.py:10-15, which imports native symbols its extension
never exported) and of the scale-folded norms ``QuantizedLlamaRMSNorm`` / ``Int8OPTLayerNorm``
(reference models/llama.py:27-37, models/opt.py:20-29).

``RMSNormQ`` / ``LayerNormQ`` return a :class:`QuantizedActivation` (int8 tensor [+ per-token scales])
that the W8A8 linears accept in place of a floating tensor, so the q/k/v (or gate/up) projections
share ONE quantised activation and the fp16 norm output never touches HBM."""
import torch

from ... import ops


class QuantizedActivation:
    """int8 activation [.., K] (+ per-token scales) produced by a fused norm; `out_dtype` is the
    floating dtype the consuming linear should emit (the dtype of the norm's input).
    row_off (int32 [M,2] or None): when set, `xq` is the OFFSET image xq + cx[m] and row_off[m] = {cx[m], sum_k image[m,k]} (include/asq_hip.h,
    "offset operand images"); the consuming linear multiplies it with its weight's image and gets the plain product bit for bit."""
    __slots__ = ("xq", "s_row", "out_dtype", "lead", "row_off")

    def __init__(self, xq, s_row, out_dtype, lead, row_off=None):
        self.xq, self.s_row, self.out_dtype, self.lead, self.row_off = xq, s_row, out_dtype, lead, row_off

    def plain_xq(self):
        """the int8 activation without its row offsets (a torch pass; only consumers that cannot take the image need it)"""
        if self.row_off is None:
            return self.xq
        return (self.xq.to(torch.int16) - self.row_off[:, :1].to(torch.int16)).to(torch.int8)

    @property
    def shape(self):
        return (*self.lead, self.xq.shape[-1])


def wants_offsets(consumers, M, dtype):
    """True when every consumer is a W8A8 linear that runs M rows (outputs of `dtype`) on offset operand images."""
    if not consumers:
        return False
    return all(hasattr(m, "offset_image") and m.offset_image(M, dtype) is not None for m in consumers)


class _NormQ(torch.nn.Module):
    def __init__(self, dim, eps, per_token, with_bias):
        super().__init__()
        self.eps, self.per_token = eps, per_token
        self.register_buffer("weight", torch.ones(dim))
        if with_bias:
            self.register_buffer("bias", torch.zeros(dim))
        else:
            self.bias = None

    @torch.no_grad()
    def forward(self, x, consumers=None):
        """consumers: the W8A8 linears that will read the result.  When every one of them runs this many rows on offset operand images
        (module.offset_image), the int8 activation is emitted as its offset image (row_off set) -- same product, less matrix-core energy."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w = self.weight if self.weight.dtype == x.dtype else self.weight.to(x.dtype)
        b = None if self.bias is None else (self.bias if self.bias.dtype == x.dtype else self.bias.to(x.dtype))
        r = ops.norm_quantize(x2, w, b, self.eps, self.per_token, offsets=wants_offsets(consumers, x2.shape[0], x.dtype))
        return QuantizedActivation(r[0], r[1], x.dtype, lead, r[2] if len(r) > 2 else None)


    @torch.no_grad()
    def add_forward(self, x, residual, consumers=None):
        """The residual-add form (reference dq_add_layernorm_q, csrc/kernels/fused.cu:5-25): h = residual + x is written once
        and normalised + quantised in the same pass.  Returns (h, QuantizedActivation of norm(h))."""
        lead = residual.shape[:-1]
        x2, r2 = x.reshape(-1, x.shape[-1]), residual.reshape(-1, residual.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = r2 if r2.is_contiguous() else r2.contiguous()
        w = self.weight if self.weight.dtype == x.dtype else self.weight.to(x.dtype)
        b = None if self.bias is None else (self.bias if self.bias.dtype == x.dtype else self.bias.to(x.dtype))
        r = ops.add_norm_quantize(x2, r2, w, b, self.eps, self.per_token, offsets=wants_offsets(consumers, x2.shape[0], x.dtype))
        return r[0].view(residual.shape), QuantizedActivation(r[1], r[2], x.dtype, lead, r[3] if len(r) > 3 else None)


class DeferredResidual:
    """`base + delta` not yet added: a decoder layer in fused mode hands its MLP output to the NEXT layer's input norm, which
    performs the add inside asq_add_norm_quantize (one pass instead of add + norm + quantise)."""
    __slots__ = ("base", "delta")

    def __init__(self, base, delta):
        self.base, self.delta = base, delta

    def materialize(self):
        return self.base + self.delta


class RMSNormQ(_NormQ):
    """RMSNorm with weight / input_scale, emitting int8 (per-tensor) or int8 + row scales (per-token)."""

    def __init__(self, dim, eps=1e-5, per_token=False):
        super().__init__(dim, eps, per_token, with_bias=False)

    @staticmethod
    def from_float(norm, input_scale=1.0, per_token=False):
        """`norm` has .weight and .eps / .variance_epsilon.  per-tensor: weight / input_scale (reference
        models/llama.py:27-37); per-token: weight unchanged (reference :326-339 folds only for per-tensor)."""
        eps = getattr(norm, "variance_epsilon", getattr(norm, "eps", 1e-5))
        q = RMSNormQ(norm.weight.numel(), eps, per_token)
        w = norm.weight.detach()
        q.weight = (w if per_token else w / input_scale).clone()
        return q.to(w.device)


class LayerNormQ(_NormQ):
    """LayerNorm with weight and bias / output_scale emitting int8 (the reference's LayerNormQ, fused.py:10-32)."""

    def __init__(self, dim, eps=1e-5, per_token=False):
        super().__init__(dim, eps, per_token, with_bias=True)

    @staticmethod
    def from_float(module: torch.nn.LayerNorm, output_scale: float, per_token=False):
        assert module.normalized_shape[0] == module.weight.numel() == module.bias.numel()
        q = LayerNormQ(module.normalized_shape[0], module.eps, per_token)
        s = 1.0 if per_token else output_scale
        q.weight = (module.weight.detach() / s).clone()
        q.bias = (module.bias.detach() / s).clone()
        return q.to(module.weight.device)


@torch.no_grad()
def silu_mul_q(gate, up, consumer, fast=None):
    """silu(gate) * up quantised for `consumer` (a W8A8BFP32OFP32LinearWithQuantScale such as down_proj / w2):
    per-token, or per-tensor with the consumer's calibrated quant_scale.  Returns a QuantizedActivation.  fast: ops.silu_mul_quantize's opt-in
    hardware-transcendental variant."""
    lead = gate.shape[:-1]
    g2, u2 = gate.reshape(-1, gate.shape[-1]), up.reshape(-1, up.shape[-1])
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    u2 = u2 if u2.is_contiguous() else u2.contiguous()
    per_token = consumer.act_quant == "per-token"
    qs = 1.0 if per_token else float(consumer.quant_scale)
    r = ops.silu_mul_quantize(g2, u2, per_token, qs, fast, offsets=wants_offsets((consumer,), g2.shape[0], gate.dtype))
    return QuantizedActivation(r[0], r[1], gate.dtype, lead, r[2] if len(r) > 2 else None)


class QuantizedActivationFp8:
    """e4m3 activation [.., K] + per-token scales f32 [M,1] produced by a fused kernel for an FP8LinearDynamic(act_quant="per-token"); `out_dtype` is the floating
    dtype the consuming linear emits.  The fp8 counterpart of QuantizedActivation."""
    __slots__ = ("xq", "scale", "out_dtype", "lead")

    def __init__(self, xq, scale, out_dtype, lead):
        self.xq, self.scale, self.out_dtype, self.lead = xq, scale, out_dtype, lead

    @property
    def shape(self):
        return (*self.lead, self.xq.shape[-1])


@torch.no_grad()
def silu_mul_q_fp8(gate, up, consumer=None, fast=None):
    """silu(gate) * up quantised per token to e4m3 for an FP8LinearDynamic (Mixtral's w2 / LLaMA's down_proj in an fp8 checkpoint; reference models/mixtral.py:99-101 on
    FP8LinearDynamic modules, linear.py:413-427): ONE pass (ops.silu_mul_quantize_fp8) instead of silu, mul and per_token_quantize_fp8.  Returns a
    QuantizedActivationFp8 the consumer's forward accepts."""
    if consumer is not None and getattr(consumer, "act_quant", None) != "per-token":
        raise ValueError("silu_mul_q_fp8 needs a per-token FP8LinearDynamic consumer (a per-tensor scale depends on the whole tensor)")
    lead = gate.shape[:-1]
    g2, u2 = gate.reshape(-1, gate.shape[-1]), up.reshape(-1, up.shape[-1])
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    u2 = u2 if u2.is_contiguous() else u2.contiguous()
    q, sc = ops.silu_mul_quantize_fp8(g2, u2, fast)
    return QuantizedActivationFp8(q, sc, gate.dtype, lead)


class GateUpSiLU(torch.nn.Module):
    """SiLU(gate_proj(x)) * up_proj(x) as ONE GEMM over the two projections' interleaved int8 rows (ops.linear_w8a8_gate_up; reference models/llama.py:206-211 /
    HF LlamaMLP).  A VIEW of two existing W8A8BFP32OFP32Linear modules: they stay the source of truth (state_dict, load_state_dict, the arena broadcast act on THEIR
    buffers); this module derives the interleaved operand -- one more int8 copy of the two weights, plus its offset image where the shape runs on images -- on
    the first forward that can use it and rebuilds it (into the same buffers) when either weight changes.
    forward(qa) -> the [.., F] activation in qa.out_dtype, or None when the shape does not run on the fused kernel (the caller then runs the two linears and
    silu_mul_q).  Bit-identical to that two-linear + asq_silu_mul_quantize composition (same `fast` flag)."""

    offsets = True

    def __init__(self, gate, up):
        super().__init__()
        if gate.use_bias or up.use_bias or gate.act_quant != up.act_quant or gate.in_features != up.in_features or gate.out_features != up.out_features:
            raise ValueError("GateUpSiLU: gate / up must be bias-free W8A8 linears of one shape and one act_quant")
        self.__dict__["gate"], self.__dict__["up"] = gate, up      # (not registered: this module owns no checkpoint state)
        self.in_features, self.out_features, self.act_quant = gate.in_features, gate.out_features, gate.act_quant

    def input_signature(self):
        return self.gate.input_signature()

    def supported(self, M, dtype):
        g, u = self.gate, self.up
        return (g._buffers["weight"].is_cuda and g.input_signature() == u.input_signature() and self.out_features % 16 == 0
                and ops.gate_up_supported(M, self.out_features, self.in_features, dtype))

    def _key(self):
        return tuple(m._weight_key(m._buffers["weight"]) for m in (self.gate, self.up))

    def _operand(self):
        key = self._key()
        if any(k[1] is None for k in key) and not self.__dict__.get("_pinned", False):
            return None   # a weight without a version counter: writes into it cannot be seen -> the unfused composition (build() is the opt-in: the caller then owns refresh())
        if self.__dict__.get("_disabled", False):
            return None
        hit = self.__dict__.get("_wgu")
        if hit is None or hit[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None
            out = hit[1] if hit is not None and hit[1].device == self.gate.weight.device else None
            try:
                hit = (key, ops.interleave_gate_up(self.gate._buffers["weight"], self.up._buffers["weight"], out=out))
            except torch.cuda.OutOfMemoryError:
                # one more int8 copy of gate + up did not fit: this module answers None from now on and the caller runs the two linears + silu_mul_q
                self.__dict__["_disabled"] = True
                self.__dict__.pop("_wgu", None)
                self.__dict__.pop("_wgu_image", None)
                return None
            self.__dict__["_wgu"] = hit
            self.__dict__["_wgu_image_key"] = None
        return hit[1]

    def build(self, M=None, dtype=None):
        """Build the interleaved operand now (load time) instead of inside the first forward that can use it, and -- given the (M, dtype) of the forwards to come -- its
        offset image too.  Also the opt-in for weights without a version counter (inference tensors): the caller then calls refresh() after writing them.
        Returns True when the module holds the operand."""
        self.__dict__["_pinned"] = True
        ok = self._operand() is not None
        if ok and M is not None and dtype is not None:
            self.offset_image(M, dtype)
        return ok

    def refresh(self):
        """Rebuild the interleaved operand -- and its offset image, if one exists -- from the CURRENT gate / up weights INTO the existing buffers.  Call after any
        weight update before replaying a hipGraph that was captured on the fused path (the graph bakes these derived buffers in; an eager forward would notice the
        change by itself, a replay cannot), next to gate.refresh_offset_image() / up.refresh_offset_image().  No-op (False) when the
        module holds no operand."""
        hit = self.__dict__.get("_wgu")
        if hit is None or torch.cuda.is_current_stream_capturing():
            return False
        self.__dict__["_wgu"] = (None, hit[1])          # (key None never matches: _operand rebuilds into hit[1])
        had_image = self.__dict__.get("_wgu_image") is not None
        w = self._operand()
        if w is None:
            return False
        if had_image:
            old = self.__dict__["_wgu_image"]
            self.__dict__["_wgu_image"] = ops.weight_offset_image(w, out=old if old[0].device == w.device else None)
            self.__dict__["_wgu_image_key"] = self._key()
        return True

    def offset_image(self, M, dtype):
        """the interleaved operand's offset image when a forward of M rows runs on images, else None (so a fused norm in front emits the activation's image for it)"""
        if not (self.offsets and self.supported(M, dtype) and ops.offsets_supported(int(M), 2 * self.out_features, self.in_features, dtype)):
            return None
        w = self._operand()
        if w is None:
            return None
        if self.__dict__.get("_wgu_image_key") != self._key():
            if torch.cuda.is_current_stream_capturing():
                return None
            old = self.__dict__.get("_wgu_image")
            try:
                self.__dict__["_wgu_image"] = ops.weight_offset_image(w, out=old if old is not None and old[0].device == w.device else None)
            except torch.cuda.OutOfMemoryError:
                self.offsets = False   # (a third int8 copy did not fit: the fused GEMM runs on the plain interleaved operand from now on)
                self.__dict__.pop("_wgu_image", None)
                return None
            self.__dict__["_wgu_image_key"] = self._key()
        return self.__dict__["_wgu_image"]

    @torch.no_grad()
    def forward(self, qa, fast=None, consumer=None):
        """consumer (optional): the linear that reads the result.  A per-tensor consumer (a W8A8BFP32OFP32LinearWithQuantScale with act_quant "per-tensor") gets its
        int8 input straight from the GEMM's epilogue (ops.linear_w8a8_gate_up_q8) as a QuantizedActivation; otherwise the [.., F] tensor in the activation dtype."""
        if not isinstance(qa, QuantizedActivation):
            raise TypeError("GateUpSiLU takes the QuantizedActivation its projections share (mod.quantize_input / a fused norm)")
        M = qa.xq.shape[0]
        if not self.supported(M, qa.out_dtype) or (self.act_quant == "per-token") != (qa.s_row is not None):
            return None
        w = self._operand()
        if w is None:
            return None
        sg, su = self.gate._scalar("dequant_scale"), self.up._scalar("dequant_scale")
        q8 = consumer is not None and getattr(consumer, "act_quant", None) == "per-tensor" and hasattr(consumer, "quant_scale") and consumer.in_features == self.out_features
        qs = float(consumer.quant_scale) if q8 else None
        ro = col = None
        if qa.row_off is not None:
            image = self.offset_image(M, qa.out_dtype)
            if image is not None:
                xq, w, ro, col = qa.xq, image[0], qa.row_off, image[1]
            else:
                xq = qa.plain_xq()
        else:
            xq = qa.xq
        if q8:
            out = ops.linear_w8a8_gate_up_q8(xq, w, qa.out_dtype, sg, su, qs, qa.s_row, fast, ro, col)
            return QuantizedActivation(out, None, qa.out_dtype, qa.lead)
        out = ops.linear_w8a8_gate_up(xq, w, qa.out_dtype, sg, su, qa.s_row, fast, ro, col)
        return out.view(*qa.lead, self.out_features)
