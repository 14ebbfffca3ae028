"""W8A8 linear modules with the reference's class names, constructor signatures, buffer
names/dtypes and ``from_float`` contracts (reference autosmoothquant/layers/nn/linear.py),
whose ``forward`` is two HIP launches on the current stream -- activation quantiser, then
the MFMA INT8 GEMM with the dequant/bias epilogue fused -- instead of ~10 eager ATen
launches around a cuBLASLt call (reference :83-106, :158-208, :278-302).

Checkpoint contract (what ``state_dict`` holds; reference :49-66, :138-149, :253-256):
    weight            int8  [out_features, in_features]
    bias              f32   [out_features]            (iff use_bias)
    dequant_scale     f32   []                        (Linear, LinearWithQuantScale)
    {q,k,v}_dequant_scale f32 []                      (QKVLinear)
    quant_scale       f32   []                        (LinearWithQuantScale, per-tensor only)
Scalar scales live on the HOST after any ``.cuda()/.half()/.to()`` (reference ``_apply``
:68-72), so reading them never synchronises the device.
"""
import threading

import torch

from .. import functional as _F  # noqa: F401  (package marker)
from ..functional.quantization import quantize_per_tensor_absmax
from ... import ops
from ..._CUDA import I8CUGEMM

_ACT_QUANT = ("per-token", "per-tensor")


def _check_act_quant(act_quant):
    assert act_quant in _ACT_QUANT, '"act_quant must be "per-token" or "per-tensor"'


class Int8GEMM(object):
    """Process-wide holder of one ``I8CUGEMM`` (reference :17-32).  Kept for API parity:
    the MI355X op object is stateless, so sharing it has no serialising effect."""
    _guard = threading.Lock()
    _singleton = None

    def __new__(cls, *args, **kwargs):
        if Int8GEMM._singleton is None:
            with Int8GEMM._guard:
                if Int8GEMM._singleton is None:
                    inst = object.__new__(cls)
                    inst.i8cugemm = I8CUGEMM()
                    Int8GEMM._singleton = inst
        return Int8GEMM._singleton

    def get_i8cugemm(self):
        return self.i8cugemm


class _W8A8Base(torch.nn.Module):
    """Shared plumbing: buffers, host-pinned scalar scales, shape handling."""
    _host_scalars = ("dequant_scale",)

    def __init__(self, in_features, out_features, use_bias=False, act_quant="per-tensor"):
        super().__init__()
        _check_act_quant(act_quant)
        self.in_features, self.out_features = in_features, out_features
        self.use_bias, self.act_quant = use_bias, act_quant
        self.i8cugemm = Int8GEMM().get_i8cugemm()
        self.register_buffer("weight", torch.empty(out_features, in_features, dtype=torch.int8, requires_grad=False))
        if use_bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32, requires_grad=False))
        for name in self._host_scalars:
            self.register_buffer(name, torch.tensor(1.0, dtype=torch.float32, requires_grad=False))

    # scalar scales stay fp32 on the host; bias stays fp32 (reference :68-81)
    def _pin_scalars(self):
        for name in self._host_scalars:
            if name in self._buffers and self._buffers[name] is not None:
                self._buffers[name] = self._buffers[name].detach().to("cpu", torch.float32)

    def _apply(self, fn, *args, **kwargs):
        bias = self._buffers.get("bias")
        super()._apply(fn, *args, **kwargs)
        if bias is not None and self._buffers["bias"].dtype != torch.float32:
            # .half()/.bfloat16() must not touch the fp32 bias: redo the move without the cast
            self._buffers["bias"] = bias.to(self._buffers["bias"].device)
        self._pin_scalars()
        return self

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        if self.use_bias:
            self.bias = self.bias.to(torch.float32)
        self._pin_scalars()
        return self

    def _scalar(self, name):
        t = self._buffers[name]
        if t.device.type != "cpu":  # e.g. from_float(save_device="cuda"): pin once, then host reads are free
            self._pin_scalars()
            t = self._buffers[name]
        return float(t)

    def _flatten(self, x):
        if x.shape[-1] != self.in_features:
            raise ValueError(f"expected last dim {self.in_features}, got {tuple(x.shape)}")
        x2 = x.reshape(-1, self.in_features)
        return x2 if x2.is_contiguous() else x2.contiguous()

    def _bias_on(self, device):
        if not self.use_bias:
            return None
        if self.bias.device != device or self.bias.dtype != torch.float32:
            self.bias = self.bias.to(device=device, dtype=torch.float32)
        return self.bias


class W8A8BFP32OFP32Linear(_W8A8Base):
    """int8 weight, int8 activation, fp32 bias, output in the input's dtype.
    per-tensor: the input is already in int8 units (1/input_scale folded into the preceding
    norm, reference models/llama.py:326-339) -> round+clamp only.  per-token: dynamic
    absmax/127 per row (reference :83-106)."""

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        mode = "per-token" if self.act_quant == "per-token" else "per-tensor-round"
        out = ops.linear_w8a8_forward(self._flatten(x), self.weight, mode, 1.0, self._scalar("dequant_scale"),
                                      None, self._bias_on(x.device))
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear, input_scale=1.0, save_device=torch.device("cpu"), act_quant="per-tensor"):
        """reference :108-129.  NOTE: like the reference, an fp32 source weight is rounded
        IN PLACE (quantize_per_tensor_absmax uses div_/round_ to save memory)."""
        _check_act_quant(act_quant)
        has_bias = module.bias is not None
        q = W8A8BFP32OFP32Linear(module.in_features, module.out_features, has_bias, act_quant)
        wq, wscale = quantize_per_tensor_absmax(module.weight)
        alpha = wscale if act_quant == "per-token" else input_scale * wscale
        q.dequant_scale = alpha.to(torch.float32).to(save_device)
        q.weight = wq.to(save_device)
        if has_bias:
            q.bias = module.bias.to(torch.float32).to(save_device)
        return q


class W8A8BFP32OFP32QKVLinear(_W8A8Base):
    """Fused QKV projection: one GEMM over N = sum(qkv_size), three scalar dequant scales on
    three column segments (reference :132-245).  The reference applies them with
    split / 3 muls / cat; here they are one cached fp32 [N] vector consumed by the GEMM
    epilogue (same fp32 arithmetic per element, no extra pass over the output)."""
    _host_scalars = ("q_dequant_scale", "k_dequant_scale", "v_dequant_scale")

    def __init__(self, qkv_size, *args, **kwargs):
        self.qkv_size = qkv_size
        super().__init__(*args, **kwargs)
        self._scol_cache = None

    def _scale_vector(self, device):
        vals = tuple(self._scalar(n) for n in self._host_scalars)
        key = (vals, tuple(self.qkv_size), str(device))
        if self._scol_cache is None or self._scol_cache[0] != key:
            parts = [torch.full((int(n),), v, dtype=torch.float32) for v, n in zip(vals, self.qkv_size)]
            self._scol_cache = (key, torch.cat(parts).to(device))
        return self._scol_cache[1]

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        mode = "per-token" if self.act_quant == "per-token" else "per-tensor-round"
        out = ops.linear_w8a8_forward(self._flatten(x), self.weight, mode, 1.0, 1.0, self._scale_vector(x.device),
                                      self._bias_on(x.device))
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear, input_scale, qkv_size, save_device=torch.device("cpu"), act_quant="per-tensor"):
        """reference :210-245: each of the q/k/v row blocks gets its own per-tensor weight scale."""
        _check_act_quant(act_quant)
        has_bias = module.bias is not None
        q = W8A8BFP32OFP32QKVLinear(qkv_size, module.in_features, module.out_features, has_bias, act_quant)
        blocks, scales = [], []
        for blk in module.weight.data.split(qkv_size, dim=0):
            wq, ws = quantize_per_tensor_absmax(blk)
            blocks.append(wq)
            scales.append(ws * input_scale if act_quant == "per-tensor" else ws)
        q.weight = torch.cat(blocks, dim=0).to(save_device)
        for name, s in zip(W8A8BFP32OFP32QKVLinear._host_scalars, scales):
            setattr(q, name, s.to(torch.float32).to(save_device))
        if has_bias:
            q.bias = module.bias.to(torch.float32).to(save_device)
        return q


class W8A8BFP32OFP32LinearWithQuantScale(_W8A8Base):
    """For projections whose input is not preceded by a norm (o_proj, down_proj, fc2, w2):
    per-tensor mode divides by the calibrated ``quant_scale`` in the input's dtype; per-token
    is the same dynamic scheme as above (reference :248-329)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.act_quant == "per-tensor":
            self._host_scalars = ("dequant_scale", "quant_scale")
            self.register_buffer("quant_scale", torch.tensor(1.0, dtype=torch.float32, requires_grad=False))

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        if self.act_quant == "per-token":
            mode, qs = "per-token", 1.0
        else:
            mode, qs = "per-tensor-div", self._scalar("quant_scale")
        out = ops.linear_w8a8_forward(self._flatten(x), self.weight, mode, qs, self._scalar("dequant_scale"), None,
                                      self._bias_on(x.device))
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear, input_scale, save_device=torch.device("cpu"), act_quant="per-token"):
        """reference :304-329 (note the default act_quant is per-token here)."""
        _check_act_quant(act_quant)
        has_bias = module.bias is not None
        q = W8A8BFP32OFP32LinearWithQuantScale(module.in_features, module.out_features, has_bias, act_quant)
        wq, wscale = quantize_per_tensor_absmax(module.weight)
        if act_quant == "per-token":
            alpha = wscale
        else:
            alpha = input_scale * wscale
            q.quant_scale = torch.tensor(input_scale, dtype=torch.float32).to(save_device)
        q.dequant_scale = alpha.to(torch.float32).to(save_device)
        q.weight = wq.to(save_device)
        if has_bias:
            q.bias = module.bias.to(torch.float32).to(save_device)
        return q
