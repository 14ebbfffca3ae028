"""SmoothQuant scale migration between a norm and the linears it feeds.

Mirror of the reference's ``smooth_ln_fcs`` (quantize/smooth.py:10-40): same name, arguments and in-place
effect, so the generate-scale -> smooth -> quantise flow can run on the MI355X box on whatever device the
weights live on (plain torch elementwise ops; nothing here is on the hot path).

    s_j = clamp( act_j^alpha / clamp(max_i |W_ij|, 1e-5)^(1-alpha), 1e-5 )      per input channel j
    norm.weight /= s   (and norm.bias for a LayerNorm);    W[:, j] *= s_j

so that  linear(norm(x))  is unchanged while activation outliers move into the weights.
"""
import torch


@torch.no_grad()
def smooth_ln_fcs(ln, fcs, act_scales, model_type="transformers", alpha=0.5):
    """ln: the norm module (LayerNorm for model_type "transformers", an RMSNorm otherwise); fcs: one Linear
    or a list of Linears reading ln's output; act_scales: per-channel activation absmax [in_features]."""
    if not isinstance(fcs, (list, tuple)):
        fcs = [fcs]
    for fc in fcs:
        if not isinstance(fc, torch.nn.Linear):
            raise TypeError("smooth_ln_fcs: fcs must be torch.nn.Linear modules")
        if not (ln.weight.numel() == fc.in_features == act_scales.numel()):
            raise ValueError("smooth_ln_fcs: norm / linear / act_scales channel counts differ")
    has_bias = model_type == "transformers"
    if has_bias and not isinstance(ln, torch.nn.LayerNorm):
        raise TypeError('smooth_ln_fcs: model_type "transformers" expects torch.nn.LayerNorm')
    device, dtype = fcs[0].weight.device, fcs[0].weight.dtype
    act = act_scales.to(device=device, dtype=dtype)
    col_absmax = torch.stack([fc.weight.abs().amax(dim=0) for fc in fcs], dim=0).amax(dim=0).clamp(min=1e-5)
    s = (act.pow(alpha) / col_absmax.pow(1 - alpha)).clamp(min=1e-5).to(device=device, dtype=dtype)
    ln.weight.div_(s)
    if has_bias:
        ln.bias.div_(s)
    for fc in fcs:
        fc.weight.mul_(s.view(1, -1))
    return s
