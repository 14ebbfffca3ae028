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


def _layer_kind(m):
    """Which decoder-layer family a module is, by the attribute names the reference's isinstance checks stand for (quantize/smooth.py:42-93 tests
    OPTDecoderLayer / LlamaDecoderLayer / BaichuanLayer / MixtralDecoderLayer of a pinned transformers version)."""
    attn = getattr(m, "self_attn", None)
    if attn is None:
        return None
    if hasattr(m, "self_attn_layer_norm") and hasattr(m, "fc1") and hasattr(attn, "q_proj"):
        return "transformers"
    if hasattr(m, "input_layernorm") and hasattr(m, "post_attention_layernorm"):
        if hasattr(m, "block_sparse_moe"):
            return "mixtral"
        if hasattr(attn, "W_pack"):
            return "baichuan"
        if hasattr(getattr(m, "mlp", None), "gate_proj") and hasattr(attn, "q_proj"):
            return "llama"
    return None


@torch.no_grad()
def smooth_lm(model, scales, alpha=0.5):
    """The reference's per-architecture walk (quantize/smooth.py:42-93): for every decoder layer, migrate the activation outliers of the attention input into
    q/k/v (or W_pack) and of the MLP input into fc1 / gate+up / the router and every expert's w1, w3.  scales: get_act_scales' dict
    {linear module name: per-channel absmax}.  Returns the number of layers smoothed."""
    n = 0
    for name, m in model.named_modules():
        kind = _layer_kind(m)
        if kind == "transformers":
            smooth_ln_fcs(m.self_attn_layer_norm, [m.self_attn.q_proj, m.self_attn.k_proj, m.self_attn.v_proj], scales[name + ".self_attn.q_proj"], kind, alpha)
            smooth_ln_fcs(m.final_layer_norm, m.fc1, scales[name + ".fc1"], kind, alpha)
        elif kind == "llama":
            smooth_ln_fcs(m.input_layernorm, [m.self_attn.q_proj, m.self_attn.k_proj, m.self_attn.v_proj], scales[name + ".self_attn.q_proj"], kind, alpha)
            smooth_ln_fcs(m.post_attention_layernorm, [m.mlp.gate_proj, m.mlp.up_proj], scales[name + ".mlp.gate_proj"], kind, alpha)
        elif kind == "baichuan":
            smooth_ln_fcs(m.input_layernorm, m.self_attn.W_pack, scales[name + ".self_attn.W_pack"], kind, alpha)
            smooth_ln_fcs(m.post_attention_layernorm, [m.mlp.gate_proj, m.mlp.up_proj], scales[name + ".mlp.gate_proj"], kind, alpha)
        elif kind == "mixtral":
            smooth_ln_fcs(m.input_layernorm, [m.self_attn.q_proj, m.self_attn.k_proj, m.self_attn.v_proj], scales[name + ".self_attn.q_proj"], kind, alpha)
            fcs = [m.block_sparse_moe.gate]
            for e in m.block_sparse_moe.experts:
                fcs += [e.w1, e.w3]
            smooth_ln_fcs(m.post_attention_layernorm, fcs, scales[name + ".block_sparse_moe.gate"], kind, alpha)
        else:
            continue
        n += 1
    return n
