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
