"""Weight-side conversion and functional quantisers with the reference's names and
semantics (reference autosmoothquant/layers/functional/quantization.py:9-120).

These are offline / conversion-time helpers (plain torch ops on whatever device the
tensor lives on); the per-forward activation quantisers of the nn modules are the HIP
prologue kernels in csrc/asq_quant.hip, reached through ``autosmoothquant_amd.ops``.
Arithmetic follows the reference branch for branch: scales are computed in the tensor's own
dtype; the division runs in fp32 for host tensors and IN THE TENSOR'S OWN DTYPE for device
tensors (reference :11-16 -- its real conversion flow loads fp16 weights onto the GPU, where the
fp16 quotient is rounded to half before ``round_()``; a device-side conversion therefore gives
the reference's device results, a host-side one its host results; they can differ by +-1 on a
few percent of large-magnitude entries of an fp16 weight, and are identical for fp32 weights).
"""
import torch


def _fp32_view(t):
    # the reference converts only non-CUDA tensors ("half rounding is not supported on CPU",
    # reference :11-13, :28-30, :45-47); device tensors are divided and rounded in place in
    # their own dtype, exactly like the reference's GPU conversion flow.
    return t if t.is_cuda else t.float()


@torch.no_grad()
def quantize_per_tensor_absmax(t):
    """(int8 tensor, scale) with scale = absmax/127 in t's dtype; an fp32 ``t`` is rounded in
    place, exactly like the reference (no clamp: |t/scale| <= 127 by construction)."""
    scale = t.abs().max() / 127
    work = _fp32_view(t)
    work.div_(scale).round_()
    return work.to(torch.int8), scale


@torch.no_grad()
def quantize_fused_tensor_absmax(linears):
    """Several Linear layers sharing one scale = max over their absmax/127 (reference :21-37)."""
    scale = 0
    for lin in linears:
        scale = max(lin.weight.abs().max() / 127, scale)
    out = []
    for lin in linears:
        work = _fp32_view(lin.weight)
        work.div_(scale).round_()
        out.append(work.to(torch.int8))
    return out, scale


@torch.no_grad()
def quantize_weight_per_channel_absmax(w):
    """w [out, in] -> (int8, scales [out, 1]) with one absmax/127 scale per output channel."""
    scales = (w.abs().max(dim=1)[0] / 127).view(-1, 1)
    work = _fp32_view(w)
    work.div_(scales).round_().clamp_(-128, 127)
    return work.to(torch.int8), scales


def _floor_scale(max_val):
    return torch.clamp(max_val, min=1e-8) / 127


@torch.no_grad()
def dynamic_quantize_activation_per_tensor_absmax(t):
    s = _floor_scale(t.abs().max())
    return (t / s).round().clamp(-128, 127).to(torch.int8), s


@torch.no_grad()
def dynamic_quantize_activation_per_token_absmax(t):
    s = _floor_scale(t.abs().max(dim=-1, keepdim=True)[0])
    t.div_(s).round_().clamp_(-128, 127)
    return t.to(torch.int8), s


@torch.no_grad()
def fake_quantize_activation_per_tensor_absmax(t):
    s = _floor_scale(t.abs().max())
    return t.div_(s).round_().clamp_(-128, 127).mul_(s)


@torch.no_grad()
def fake_quantize_activation_per_token_absmax(t):
    s = _floor_scale(t.abs().max(dim=-1, keepdim=True)[0])
    return t.div_(s).round_().clamp_(-128, 127).mul_(s)


@torch.no_grad()
def dequantize_activation_w_per_channel_a_per_token(q_act, w_scales, a_scales):
    """q_act i32 [B, dim], w_scales [dim], a_scales [B, 1] -> a_scales.dtype.
    On a HIP device this is the GEMM epilogue order ASQ_EPI_ACC_FIRST; this helper is the
    stand-alone form for already-materialised accumulators."""
    out = q_act.to(torch.float32)
    out.mul_(w_scales.reshape(1, -1)).mul_(a_scales.reshape(-1, 1))
    return out.to(a_scales.dtype)


@torch.no_grad()
def dequantize_activation_w_per_channel_a_per_tensor(q_act, w_scales, a_scales):
    out = q_act.to(torch.float32) * w_scales.reshape(1, -1) * a_scales
    return out.to(a_scales.dtype)


##### FP8 (OCP e4m3fn) -- reference quantization.py:126-211 ########################################
import re as _re


def new_dtype_byte_size(dtype):
    """Byte size of a dtype parsed from its name (float8_e4m3fn -> 1): the reference installs this over
    transformers.modeling_utils.dtype_byte_size so fp8 checkpoints can be sharded (quantization.py:126-136)."""
    if dtype == torch.bool:
        return 1 / 8
    m = _re.search(r"[^\d](\d+)_?", str(dtype))
    if m is None:
        raise ValueError(f"`dtype` is not a valid dtype: {dtype}.")
    return int(m.groups()[0]) // 8


try:  # same monkey-patch as the reference, when the installed transformers still has the hook
    import transformers.modeling_utils as _tmu
    if hasattr(_tmu, "dtype_byte_size"):
        _tmu.dtype_byte_size = new_dtype_byte_size
except Exception:  # transformers absent or incompatible: nothing to patch
    pass

_E4M3 = torch.finfo(torch.float8_e4m3fn)


def per_tensor_quantize_fp8(tensor):
    """(float8_e4m3fn tensor, scale) with scale = absmax / 448 as a 0-dim tensor of the input dtype.
    Empty tensors (empty MoE experts) use the fixed range [-16, 16].  On a HIP tensor the per-forward
    version of this is ops.quantize_act_fp8(x, "per-tensor")."""
    if tensor.numel() == 0:
        lo, hi = torch.tensor(-16.0, dtype=tensor.dtype), torch.tensor(16.0, dtype=tensor.dtype)
    else:
        lo, hi = tensor.aminmax()
    scale = torch.maximum(lo.abs(), hi.abs()) / _E4M3.max
    q = (tensor / scale).clamp(min=_E4M3.min, max=_E4M3.max).to(torch.float8_e4m3fn)
    return q, scale


def per_token_quantize_fp8(tensor):
    assert tensor.numel() > 0
    scale = tensor.abs().max(dim=-1, keepdim=True)[0].div(_E4M3.max).to(torch.float32)
    q = (tensor / scale).clamp(min=_E4M3.min, max=_E4M3.max).to(torch.float8_e4m3fn)
    return q, scale


def static_per_tensor_quantize_fp8(tensor, inv_scale):
    return (tensor / inv_scale).clamp(min=_E4M3.min, max=_E4M3.max).to(torch.float8_e4m3fn)


def fake_per_tensor_quantize_fp8(tensor):
    assert tensor.numel() > 0
    scale = tensor.abs().max().div(_E4M3.max).to(torch.float32)
    return (tensor / scale).clamp(min=_E4M3.min, max=_E4M3.max).mul(scale)


def fake_per_token_quantize_fp8(tensor):
    assert tensor.numel() > 0
    scale = tensor.abs().max(dim=-1, keepdim=True)[0].div(_E4M3.max).to(torch.float32)
    return (tensor / scale).clamp(min=_E4M3.min, max=_E4M3.max).mul(scale)
