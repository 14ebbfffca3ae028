"""Weight-side conversion and functional quantisers with the reference's names and
semantics (reference autosmoothquant/layers/functional/quantization.py:9-120).

These are offline / conversion-time helpers (plain torch ops on whatever device the
tensor lives on); the per-forward activation quantisers of the nn modules are the HIP
prologue kernels in csrc/asq_quant.hip, reached through ``autosmoothquant_amd.ops``.
Arithmetic follows the reference's CPU branch: scales are computed in the tensor's own
dtype, the division itself in fp32.
"""
import torch


def _fp32_view(t):
    # the reference converts only non-CUDA tensors ("half rounding is not supported on CPU",
    # reference :11-13); doing it everywhere keeps results identical to its CPU path.
    return t if t.dtype == torch.float32 else t.float()


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
