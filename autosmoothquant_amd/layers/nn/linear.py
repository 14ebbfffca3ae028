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
from .fused import QuantizedActivation, QuantizedActivationFp8

_ACT_CODE = dict(ops._ACT)   # quantiser mode name -> C-ABI code
_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device   # (the raw getter: ~0.1 us instead of ~0.4)

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


# The reference decorates every forward with @torch.no_grad().  The int8 forwards below never record an autograd
# graph in the first place (their outputs are fresh buffers filled through the C-ABI), so the decorator's ~2 us of
# context management per call is dropped on this hot path; the result is the same: outputs never require grad.
class _W8A8Base(torch.nn.Module):
    """Shared plumbing: buffers, host-pinned scalar scales, shape handling."""
    _host_scalars = ("dequant_scale",)
    _fast_ok = False   # classes whose forward is exactly _module_forward(mode, qs, dequant_scale, no column vector) switch the cached decode call on

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

    def _fast(self, x):
        """The cached call of a decode-sized forward (round 6): _module_forward records, per module, the last (shape, dtype, device) it ran WITHOUT an
        offset image together with everything that was derived from the module's state -- C-ABI entry, quantiser code, the host scalars as Python floats, output
        shape.  A later forward with the same input geometry whose state tensors are still the same OBJECTS at the same versions (weight, bias, every host scalar)
        skips the re-derivation and the re-validation: empty + one C-ABI call.  Anything else returns None and takes the full path, which refreshes the record.
        Not used for shapes that run on offset images (their image follows the weight's version; those forwards are GPU-bound anyway)."""
        fc = self.__dict__.get("_fast_state")
        if fc is None or x.shape != fc[0] or x.dtype is not fc[1] or not x.is_cuda or x.get_device() != fc[2] or not x.is_contiguous():
            return None
        b = self._buffers
        if b["weight"] is not fc[3] or b.get("bias") is not fc[4] or _cur_dev() != fc[2]:
            return None
        for name, t, ver in fc[5]:
            u = b[name]
            if u is not t or u._version != ver:
                return None
        out = torch.empty(fc[6], dtype=fc[1], device=fc[7])
        stream = ops._stream(x)
        ws, nbytes = ops._forward_ws(fc[8], fc[9], fc[10], fc[11], fc[7], stream)
        rc = fc[12](x.data_ptr(), fc[13], fc[3].data_ptr(), out.data_ptr(), fc[9], fc[10], fc[11], fc[14], fc[15], fc[16], None, None if fc[4] is None else fc[4].data_ptr(),
                    None if ws is None else ws.data_ptr(), nbytes, stream)
        if rc:
            ops.L.check(rc, "asq_linear_w8a8_forward")
        return out

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

    # (mode, quant_scale) of this module's own activation quantiser, and (s_scalar, s_col) of its dequant epilogue
    def _input_mode(self):
        return ("per-token", 1.0) if self.act_quant == "per-token" else ("per-tensor-round", 1.0)

    def _epilogue_scales(self, device):
        return self._scalar("dequant_scale"), None

    def input_signature(self):
        """What this module's prologue computes from its input: (quantiser mode, scale).  Modules with equal signatures produce the same int8
        activation from the same tensor, so one `quantize_input` serves them all."""
        mode, qs = self._input_mode()
        return (mode, float(qs))

    def quantize_input(self, x, consumers=None):
        """The module's own prologue as a separate, explicit step (one asq_quantize_act launch): the QuantizedActivation it returns is accepted by
        this module and by every module with the same `input_signature()`.  Bit-identical to what forward(x) computes internally.
        consumers: the modules that will read the result (default: this one).  When every one of them runs this many rows on offset operand
        images (`offset_image`), the activation is emitted as its offset image (asq_quantize_act_off) -- same product, less matrix-core energy."""
        mode, qs = self._input_mode()
        x2 = self._flatten(x)
        mods = (self,) if consumers is None else tuple(consumers)
        # (the image quantiser has no scalar path for a contiguous view that does not start on a 16-byte boundary; the plain one has)
        if x2.is_cuda and x2.data_ptr() % 16 == 0 and all(isinstance(m, _W8A8Base) and m.offset_image(x2.shape[0], x.dtype) is not None for m in mods):
            xq, s_row, row_off = ops.quantize_act_off(x2, mode, qs)
            return QuantizedActivation(xq, s_row, x.dtype, x.shape[:-1], row_off)
        xq, s_row = ops.quantize_act(x2, mode, qs)
        return QuantizedActivation(xq, s_row, x.dtype, x.shape[:-1])

    # ---- offset operand image of the weight (include/asq_hip.h): built on the first forward whose shape the C-ABI runs on it (or up front by
    # build_offset_image), kept in ONE pair of buffers for the module's lifetime on a device and rebuilt INTO them when the weight changes
    offsets = True   # class / instance switch (ASQ_OFFSETS=0 in the environment switches the C-ABI side off for the whole process)

    @staticmethod
    def _weight_key(w):
        try:
            ver = w._version
        except RuntimeError:   # an inference tensor has no version counter: writes into it cannot be seen
            ver = None
        return (w.data_ptr(), ver, w.device)

    def _image_buffers(self, w, rebuild_key):
        """(re)build the image for weight `w`; reuses the module's buffers when they still fit (same shape and device), so a hipGraph captured on them sees the new
        image at its next replay instead of a freed one.  Allocation failure switches images off for this module: the forward runs on plain operands."""
        hit = self.__dict__.get("_offset_cache")
        out = None
        if hit is not None and hit[1][0].device == w.device and tuple(hit[1][0].shape) == tuple(w.shape):
            out = hit[1]
        try:
            image = ops.weight_offset_image(w, out=out)
        except torch.cuda.OutOfMemoryError:
            self.offsets = False   # (a second int8 copy of the weight did not fit: stay on plain operands from now on)
            self.__dict__.pop("_offset_cache", None)
            return None
        self.__dict__["_offset_cache"] = (rebuild_key, image)
        self.__dict__.pop("_offset_dirty", None)
        return image

    def offset_image(self, M, dtype):
        """(w_off int8 [N,K], col_off int32 [N,2]) when a forward of M rows with `dtype` outputs runs on offset operand images, else None.
        The image is a second int8 copy of the weight -- N x K bytes more per module that sees prefill-sized inputs (build_offset_image builds it at load
        time instead of inside the first such forward).  It is keyed on the weight's storage and version counter: in-place torch operations, .to() and
        replica.broadcast_quantized are seen and the image is rebuilt into the same buffers; load_state_dict marks it stale explicitly.
        A weight without a version counter (an inference tensor) gets an image only through build_offset_image(): writes into it are invisible, the caller
        then owns the refresh (refresh_offset_image).
        hipGraphs: a graph captured while an image exists bakes its buffers in.  They are stable, but their CONTENT follows the weight only when some eager call
        notices the change -- after writing the weight, call refresh_offset_image() before the next replay."""
        w = self._buffers["weight"]
        if not (self.offsets and w.is_cuda):
            return None
        ok = self.__dict__.setdefault("_offset_ok", {})   # (M, dtype) -> bool: one C-ABI query per shape, not per call
        sup = ok.get((M, dtype))
        if sup is None:
            if len(ok) > 64:
                ok.clear()
            sup = ok[(M, dtype)] = ops.offsets_supported(int(M), self.out_features, self.in_features, dtype)
        if not sup:
            return None
        key = self._weight_key(w)
        hit = self.__dict__.get("_offset_cache")
        if key[1] is None and not self.__dict__.get("_offset_pinned", False):
            return None   # no version counter and nobody promised to refresh: plain operands
        if hit is not None and hit[0] == key and not self.__dict__.get("_offset_dirty", False):
            return hit[1]
        if torch.cuda.is_current_stream_capturing():
            return None   # (an image built inside a capture would only exist after the first replay: this capture runs on the plain operands)
        return self._image_buffers(w, key)

    def build_offset_image(self):
        """Build the image now (load time) instead of lazily inside the first prefill-sized forward: the N x K extra bytes are allocated where an OOM is
        cheap to handle, and a later capture finds the image in place.  Returns True when the module now holds one.  For a weight without a version counter
        (inference tensor) this is also the opt-in: the caller refreshes after writes."""
        w = self._buffers["weight"]
        if not (self.offsets and w.is_cuda):
            return False
        self.__dict__["_offset_pinned"] = True
        return self._image_buffers(w, self._weight_key(w)) is not None

    def refresh_offset_image(self):
        """Rebuild the image from the current weight INTO the existing buffers (one pass over the weight).  Call after a weight update that torch cannot
        see -- a raw pointer write, an external library, copy_ into an inference tensor -- and after ANY weight update before replaying a hipGraph that was
        captured on the image.  No-op (returns False) when the module holds no image."""
        w = self._buffers["weight"]
        if self.__dict__.get("_offset_cache") is None or not (self.offsets and w.is_cuda):
            return False
        return self._image_buffers(w, self._weight_key(w)) is not None

    def invalidate_offset_image(self):
        """Mark the cached image stale: the next eager forward that wants it rebuilds it into the same buffers.  (In-place torch operations on a weight with a
        version counter, .to() and replica.broadcast_quantized are detected without this; load_state_dict calls it.)"""
        if self.__dict__.get("_offset_cache") is not None:
            self.__dict__["_offset_dirty"] = True

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + "weight" in state_dict:
            self.invalidate_offset_image()   # (copy_ into an inference tensor bumps no version counter)

    def forward_q(self, x, consumer, act=None):
        """This linear, an optional activation (act = "relu": OPT's fc1 -> ReLU -> fc2, reference models/opt.py:127-128) and the
        per-tensor prologue of `consumer` (the next W8A8 linear) in ONE GEMM launch with an int8-out epilogue
        (asq_linear_w8a8_q8).  Returns the QuantizedActivation `consumer` accepts; bit-identical to
        consumer's own quantisation of act(self(x))."""
        if getattr(consumer, "act_quant", None) != "per-tensor":
            raise ValueError("forward_q needs a per-tensor consumer (a per-token scale depends on the whole output row)")
        if isinstance(x, QuantizedActivation):
            per_token = self.act_quant == "per-token"
            if per_token != (x.s_row is not None):
                raise ValueError("QuantizedActivation does not match this module's act_quant")
            xq, s_row, lead, dt = x.plain_xq(), x.s_row, x.lead, x.out_dtype
        else:
            lead, dt = x.shape[:-1], x.dtype
            mode, qs = self._input_mode()
            xq, s_row = ops.quantize_act(self._flatten(x), mode, qs)
        s_scalar, s_col = self._epilogue_scales(xq.device)
        div = "quant_scale" in consumer._buffers
        out = ops.linear_w8a8_q8(xq, self.weight, dt, s_scalar, s_row, s_col, self._bias_on(xq.device), act,
                                 "per-tensor-div" if div else "per-tensor-round", consumer._scalar("quant_scale") if div else 1.0)
        return QuantizedActivation(out, None, dt, lead)


def _prequantized_forward(mod, qa, s_scalar, s_col):
    """Forward on an activation already quantised by a fused norm (layers/nn/fused.py): no prologue launch."""
    per_token = mod.act_quant == "per-token"
    if per_token != (qa.s_row is not None):
        raise ValueError(f"{type(mod).__name__}(act_quant={mod.act_quant!r}) got a {'per-token' if qa.s_row is not None else 'per-tensor'} QuantizedActivation")
    if qa.xq.shape[-1] != mod.in_features:
        raise ValueError(f"expected last dim {mod.in_features}, got {tuple(qa.shape)}")
    if qa.row_off is not None:
        image = mod.offset_image(qa.xq.shape[0], qa.out_dtype)
        if image is not None:
            out = ops.linear_w8a8_off(qa.xq, image[0], qa.row_off, image[1], qa.out_dtype, s_scalar, qa.s_row, s_col, mod._bias_on(qa.xq.device))
            return out.view(*qa.lead, mod.out_features)
        xq = qa.plain_xq()   # (a consumer the producer was not told about: correct, one extra pass)
    else:
        xq = qa.xq
    out = ops.linear_w8a8(xq, mod.weight, qa.out_dtype, s_scalar, qa.s_row, s_col, mod._bias_on(qa.xq.device))
    return out.view(*qa.lead, mod.out_features)


def _module_forward(mod, x, mode, qs, s_scalar, s_col):
    """quantise -> GEMM + epilogue for a floating input: every forward quantises its own input, as the reference does
    (layers/nn/linear.py:88-96).  Callers that KNOW several modules read one tensor (q/k/v, gate/up) quantise it once,
    explicitly: `qa = mod.quantize_input(x)` and pass `qa` to each (harness.shared_input)."""
    if not x.is_cuda:   # (before anything is moved for it)
        raise RuntimeError(f"x is on {x.device}: autosmoothquant_amd ops need a HIP (cuda:N) tensor; there is no CPU fallback")
    lead = x.shape[:-1]
    x2 = mod._flatten(x)
    w = mod._buffers["weight"]
    bias = mod._bias_on(x.device)
    # the module's own tensors are validated once per tensor OBJECT (a new object -- .to(), load_state_dict(assign=True), a re-homed buffer -- is checked again;
    # dtype / device / contiguity of an existing object only change through `.data = ...`, which the device check of the trusted op still catches)
    seen = mod.__dict__.get("_checked")
    if seen is None or seen[0] is not w or seen[1] is not bias or seen[2] is not s_col:
        ops._dev(w, "weight")
        if w.dtype != torch.int8 or w.dim() != 2 or tuple(w.shape) != (mod.out_features, mod.in_features):
            raise ValueError(f"weight must be int8 [{mod.out_features}, {mod.in_features}], got {tuple(w.shape)} {w.dtype}")
        for name, t in (("s_col", s_col), ("bias", bias)):
            if t is not None:
                ops._dev(t, name)
                if t.dtype != torch.float32 or t.numel() != mod.out_features or t.device != w.device:
                    raise ValueError(f"{name} must be float32 with {mod.out_features} elements on {w.device}")
        mod.__dict__["_checked"] = (w, bias, s_col)
    image = mod.offset_image(x2.shape[0], x.dtype)
    out = ops.linear_w8a8_forward_trusted(x2, w, _ACT_CODE[mode], float(qs), float(s_scalar), s_col, bias, image, mod.out_features, mod.in_features)
    if image is None and s_col is None and x2.shape[0] <= 64 and x.is_contiguous() and type(mod)._fast_ok:
        # decode-sized, no image: remember the derived call for _fast() (see there); host scalars by object + version, so an in-place rewrite of a scale is seen
        try:
            scal = tuple((n, mod._buffers[n], mod._buffers[n]._version) for n in mod._host_scalars)
            lib = ops.L.lib()
            mod.__dict__["_fast_state"] = (x.shape, x.dtype, x.get_device(), w, bias, scal, (*lead, mod.out_features), x.device, lib, x2.shape[0], mod.out_features, mod.in_features,
                                           lib.asq_linear_w8a8_forward, ops._DT[x.dtype], _ACT_CODE[mode], float(qs), float(s_scalar))
        except RuntimeError:   # (inference tensors have no version counter: no cached call)
            mod.__dict__.pop("_fast_state", None)
    else:
        mod.__dict__.pop("_fast_state", None)
    return out.view(*lead, mod.out_features)


class W8A8BFP32OFP32Linear(_W8A8Base):
    """int8 weight, int8 activation, fp32 bias, output in the input's dtype.
    per-tensor: the input is already in int8 units (1/input_scale folded into the preceding
    norm, reference models/llama.py:326-339) -> round+clamp only.  per-token: dynamic
    absmax/127 per row (reference :83-106)."""

    _fast_ok = True

    def forward(self, x):
        if isinstance(x, QuantizedActivation):
            return _prequantized_forward(self, x, self._scalar("dequant_scale"), None)
        y = self._fast(x)
        if y is not None:
            return y
        return _module_forward(self, x, "per-token" if self.act_quant == "per-token" else "per-tensor-round", 1.0, self._scalar("dequant_scale"), None)

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
            q.bias = module.bias.detach().to(torch.float32).to(save_device)
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

    def _epilogue_scales(self, device):
        return 1.0, self._scale_vector(device)

    def forward(self, x):
        if isinstance(x, QuantizedActivation):
            return _prequantized_forward(self, x, 1.0, self._scale_vector(x.xq.device))
        return _module_forward(self, x, "per-token" if self.act_quant == "per-token" else "per-tensor-round", 1.0, 1.0, self._scale_vector(x.device))

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
            q.bias = module.bias.detach().to(torch.float32).to(save_device)
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

    def _input_mode(self):
        return ("per-token", 1.0) if self.act_quant == "per-token" else ("per-tensor-div", self._scalar("quant_scale"))

    _fast_ok = True

    def forward(self, x):
        if isinstance(x, QuantizedActivation):  # already quantised for this module (fused.silu_mul_q)
            return _prequantized_forward(self, x, self._scalar("dequant_scale"), None)
        y = self._fast(x)
        if y is not None:
            return y
        mode, qs = self._input_mode()
        return _module_forward(self, x, mode, qs, self._scalar("dequant_scale"), None)

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
            q.bias = module.bias.detach().to(torch.float32).to(save_device)
        return q


##### FP8 linears (reference linear.py:332-644) ###################################################
# The reference hard-codes native_fp8_support = False and runs F.linear on dequantised operands
# (easy_fp8_gemm, :336-369).  Here the product runs on the e4m3 / e5m2 matrix cores with fp32
# accumulation and the two scales are applied once in the epilogue: agreement is to fp32 rounding
# (tests: rtol 1e-3), and every activation dtype works (the reference's path only accepts fp32
# activations once a bias or per-token scales are involved: F.linear rejects Float x Half).
from ..functional.quantization import per_tensor_quantize_fp8  # noqa: E402


def easy_fp8_gemm(A, A_scale, B, B_scale, bias, out_dtype):
    """out = (A . B^T) * A_scale * B_scale (+ bias) in out_dtype; empty A (empty MoE expert) gives an
    empty output (reference :336-340)."""
    if A.numel() == 0:
        return torch.empty(size=(0, B.shape[0]), dtype=out_dtype, device=A.device)
    lead = A.shape[:-1]
    out = ops.linear_fp8(A.reshape(-1, A.shape[-1]), A_scale, B, float(B_scale), bias, out_dtype)
    return out.view(*lead, B.shape[0])


class _FP8Base(torch.nn.Module):
    _host_scalars = ("weight_scale",)
    _weight_dtype = torch.float8_e4m3fn

    def __init__(self, in_features, out_features, use_bias=False):
        super().__init__()
        self.in_features, self.out_features, self.use_bias = in_features, out_features, use_bias
        self.register_buffer("weight", torch.empty(out_features, in_features, dtype=self._weight_dtype, requires_grad=False))
        if use_bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32, requires_grad=False))
        for name in self._host_scalars:
            self.register_buffer(name, torch.tensor(1.0, dtype=torch.float32, requires_grad=False))

    def _apply(self, fn, *args, **kwargs):
        bias = self._buffers.get("bias")
        super()._apply(fn, *args, **kwargs)
        if bias is not None and self._buffers["bias"].dtype != torch.float32:
            self._buffers["bias"] = bias.to(self._buffers["bias"].device)
        for name in self._host_scalars:  # scalar scales live on the host (reference :406-410, :543-549)
            if self._buffers.get(name) is not None:
                self._buffers[name] = self._buffers[name].detach().to("cpu", torch.float32)
        return self

    def _bias_dev(self, device):
        if not self.use_bias or self._buffers.get("bias") is None:
            return None
        if self.bias.device != device or self.bias.dtype != torch.float32:
            self.bias = self.bias.detach().to(device=device, dtype=torch.float32)
        return self.bias

    def _x2d(self, x):
        x2 = x.reshape(-1, x.shape[-1])
        return x2 if x2.is_contiguous() else x2.contiguous()


class FP8LinearDynamic(_FP8Base):
    """Dynamic activation scaling, per-tensor weight scale (reference :373-452)."""

    def __init__(self, in_features, out_features, act_quant, use_bias=False):
        super().__init__(in_features, out_features, use_bias)
        self.act_quant = act_quant

    @torch.no_grad()
    def forward(self, x):
        if isinstance(x, QuantizedActivationFp8):   # already quantised for this module by a fused kernel (fused.silu_mul_q_fp8): no prologue launch
            if self.act_quant != "per-token":
                raise ValueError("a QuantizedActivationFp8 carries per-token scales; this FP8LinearDynamic is per-tensor")
            if x.xq.shape[-1] != self.in_features:
                raise ValueError(f"expected last dim {self.in_features}, got {tuple(x.shape)}")
            if x.xq.numel() == 0:
                return torch.empty(*x.lead, self.out_features, dtype=x.out_dtype, device=x.xq.device)
            out = ops.linear_fp8(x.xq, x.scale, self.weight, float(self.weight_scale), self._bias_dev(x.xq.device), x.out_dtype)
            return out.view(*x.lead, self.out_features)
        lead = x.shape[:-1]
        x2 = self._x2d(x)
        if x2.numel() == 0:
            return torch.empty(*lead, self.out_features, dtype=x.dtype, device=x.device)
        q, s = ops.quantize_act_fp8(x2, "per-token" if self.act_quant == "per-token" else "per-tensor")
        out = ops.linear_fp8(q, s, self.weight, float(self.weight_scale), self._bias_dev(x.device), x.dtype)
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear, input_scale=1.0, save_device=torch.device("cpu"), act_quant="per-token"):
        """reference :429-452.  Kept bug-for-bug: the reference passes `use_bias` positionally into the
        `act_quant` slot (:444-446), so the converted module has act_quant = True/False (=> the
        per-tensor branch) and use_bias = False (the bias is dropped)."""
        assert act_quant == "per-token"
        qw, wscale = per_tensor_quantize_fp8(module.weight)
        use_bias = module.bias is not None
        if use_bias:
            import warnings
            warnings.warn("FP8LinearDynamic.from_float reproduces the reference's positional-argument slip (linear.py:444-446): the converted "
                          "module runs the per-tensor branch and DROPS the bias from forward (it stays in the state_dict); construct "
                          "FP8LinearDynamic(in, out, act_quant, use_bias=True) directly for a biased linear", stacklevel=2)
        m = FP8LinearDynamic(module.in_features, module.out_features, use_bias)
        m.weight = qw
        m.weight_scale = input_scale * wscale
        if use_bias:
            import copy
            m.bias = copy.deepcopy(module.bias)  # registered (a Parameter) but unused: use_bias is False
        return m


class FP8StaticLinearQuantizer(torch.nn.Module):
    """Calibration-time wrapper that records the running max of the dynamic per-tensor input (and
    optionally output) scale (reference :455-500)."""

    def __init__(self, in_features, out_features, weight, weight_scale, bias, quantize_output=False):
        super().__init__()
        self.weight = torch.nn.Parameter(weight, requires_grad=False)
        self.weight_scale = torch.nn.Parameter(weight_scale, requires_grad=False)
        self.bias = bias
        self.input_scale = None
        self.output_scale = None
        self.quantize_output = quantize_output
        self.in_features, self.out_features = in_features, out_features

    @staticmethod
    def _track(cur, new):
        if cur is None or bool(new > cur):
            return torch.nn.Parameter(new.detach().clone(), requires_grad=False)
        return cur

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        q, s = ops.quantize_act_fp8(x2, "per-tensor")
        self.input_scale = self._track(self.input_scale, s)
        bias = None if self.bias is None else self.bias.detach().to(device=x.device, dtype=torch.float32)
        # as the reference: the GEMM uses the RUNNING input scale, not this batch's
        out = ops.linear_fp8(q, self.input_scale.detach().to(x.device), self.weight, float(self.weight_scale), bias, x.dtype)
        if self.quantize_output:
            qo, so = ops.quantize_act_fp8(out, "per-tensor")
            self.output_scale = self._track(self.output_scale, so)
            out = qo.to(out.dtype) * so
        return out.view(*lead, self.out_features)


class FP8LinearStatic(_FP8Base):
    """Static (calibrated) input / output scales (reference :503-580)."""
    _host_scalars = ("weight_scale", "input_scale", "output_scale")

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        x2 = self._x2d(x)
        if x2.numel() == 0:
            return torch.empty(*lead, self.out_features, dtype=x.dtype, device=x.device)
        in_scale = float(self.input_scale)
        q, _ = ops.quantize_act_fp8(x2, "static", in_scale)
        out = ops.linear_fp8(q, in_scale, self.weight, float(self.weight_scale), self._bias_dev(x.device), x.dtype)
        out_scale = float(self.output_scale)
        if out_scale:  # `if self.output_scale:` in the reference (:562): re-quantise the output when non-zero
            qo, _ = ops.quantize_act_fp8(out, "static", out_scale)
            out = qo.to(out.dtype) * out_scale
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(quantizer):
        use_bias = quantizer.bias is not None
        m = FP8LinearStatic(quantizer.in_features, quantizer.out_features, use_bias)
        m.weight = quantizer.weight.detach()
        if use_bias:
            m.bias = quantizer.bias.detach().to(torch.float32)
        m.weight_scale = quantizer.weight_scale.detach().to("cpu", torch.float32)
        m.input_scale = quantizer.input_scale.detach().to("cpu", torch.float32)
        m.output_scale = (quantizer.output_scale.detach() if quantizer.output_scale is not None else torch.tensor(0.0)).to("cpu", torch.float32)
        return m


class FP8E5M2Linear(_FP8Base):
    """Unscaled e5m2 weights and activations (reference :583-644).  The reference's forward calls
    torch._scaled_mm(qinput, self.weight, scale_a=None, ...), which current PyTorch rejects (and which
    lacks the weight transpose); restated from its intent: y = e5m2(x) . e5m2(W)^T + bias."""
    _host_scalars = ()
    _weight_dtype = torch.float8_e5m2

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        x2 = self._x2d(x)
        if x2.numel() == 0:
            return torch.empty(*lead, self.out_features, dtype=x.dtype, device=x.device)
        out = ops.linear_fp8(ops.cast_e5m2(x2), 1.0, self.weight, 1.0, self._bias_dev(x.device), x.dtype)
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear):
        use_bias = module.bias is not None
        m = FP8E5M2Linear(module.in_features, module.out_features, use_bias=use_bias)
        m.weight = module.weight.detach().to(torch.float8_e5m2)
        if use_bias:
            m.bias = module.bias.detach().clone().to(torch.float32)
        return m


class FP8LinearMX(_FP8Base):
    """Extension beyond the reference: e4m3 weights AND activations with OCP Microscaling block scales (one E8M0 byte per 32
    consecutive k), the product on the scaled matrix-core instruction (ops.linear_mxfp8).  Opt-in only: on SmoothQuant-like activations
    FP8LinearDynamic (per-token e4m3) is both more accurate and faster.  Buffers: weight float8_e4m3fn [N, K],
    weight_scale_mx uint8 [N, K/32], bias f32 [N]; in_features must be a multiple of 64."""
    _host_scalars = ()

    def __init__(self, in_features, out_features, use_bias=False):
        if in_features % 64 != 0:
            raise ValueError("FP8LinearMX needs in_features % 64 == 0 (MX blocks of 32, two per matrix-core instruction)")
        super().__init__(in_features, out_features, use_bias)
        self.register_buffer("weight_scale_mx", torch.full((out_features, in_features // 32), 127, dtype=torch.uint8, requires_grad=False))

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        x2 = self._x2d(x)
        if x2.numel() == 0:
            return torch.empty(*lead, self.out_features, dtype=x.dtype, device=x.device)
        xq, xs = ops.quantize_mxfp8(x2)
        out = ops.linear_mxfp8(xq, xs, self.weight, self.weight_scale_mx, x.dtype, self._bias_dev(x.device))
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear):
        """The source module must live on the HIP device: the weight is quantised by the same kernel as the activations (no CPU path)."""
        use_bias = module.bias is not None
        m = FP8LinearMX(module.in_features, module.out_features, use_bias=use_bias)
        wq, ws = ops.quantize_mxfp8(module.weight.detach().contiguous())
        m.weight, m.weight_scale_mx = wq, ws
        if use_bias:
            m.bias = module.bias.detach().clone().to(torch.float32)
        return m
