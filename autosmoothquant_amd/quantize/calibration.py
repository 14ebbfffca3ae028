"""Activation-scale collectors: the forward-hook statistics the reference gathers before
smoothing / quantising (quantize/calibration.py:44-88 ``get_act_scales``, :185-244
``get_static_decoder_layer_scales`` and the per-architecture ``collect_*_layer_scales`` :91-183).

Same statistics, same result keys, but decoupled from the reference's dataset / tokenizer plumbing and from a
pinned ``transformers`` version: the caller passes an iterable of model inputs (tensors, tuples or kwargs
dicts), the collectors only rely on ``torch.nn.Linear`` module names.  Offline tooling -- nothing here is on
the hot path; it runs on whatever device the model lives on.
"""
import functools
import json
from collections import defaultdict

import torch


import contextlib


@contextlib.contextmanager
def all_experts_routed(model):
    """Mixtral calibration: route every token to EVERY expert (`block_sparse_moe.top_k = num_local_experts`) for the duration
    of the run so each expert's w2 sees activations, then restore -- the reference's _model_preprocess / _model_postprocess
    (quantize/calibration.py:23-42).  A no-op for models without block_sparse_moe modules."""
    touched = []
    for m in model.modules():
        moe = getattr(m, "block_sparse_moe", None)
        if moe is not None and hasattr(moe, "top_k"):
            n = getattr(moe, "num_experts", None) or len(getattr(moe, "experts", [])) or moe.top_k
            touched.append((moe, moe.top_k))
            moe.top_k = n
    try:
        yield len(touched)
    finally:
        for moe, k in touched:
            moe.top_k = k


def _run(model, batches):
    model.eval()
    for b in batches:
        if isinstance(b, dict):
            model(**b)
        elif isinstance(b, (tuple, list)):
            model(*b)
        else:
            model(b)


@torch.no_grad()
def get_act_scales(model, batches):
    """Per-input-channel absmax of every ``nn.Linear`` input over the calibration batches
    (reference get_act_scales, calibration.py:44-88): {module name: float32 CPU tensor [in_features]}.
    Feeds ``smooth_ln_fcs``."""
    act_scales = {}

    def hook(m, x, y, name):
        x = x[0] if isinstance(x, tuple) else x
        cur = x.detach().reshape(-1, x.shape[-1]).abs().amax(dim=0).float().cpu()
        act_scales[name] = torch.maximum(act_scales[name], cur) if name in act_scales else cur

    hooks = [m.register_forward_hook(functools.partial(hook, name=n)) for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)]
    try:
        _run(model, batches)
    finally:
        for h in hooks:
            h.remove()
    return act_scales


@torch.no_grad()
def get_io_absmax(model, batches):
    """Per-tensor absmax of every ``nn.Linear`` input and output (reference stat_io_hook, calibration.py:196-210):
    {module name: {"input": float, "output": float}}."""
    act_dict = defaultdict(dict)

    def hook(m, x, y, name):
        x = x[0] if isinstance(x, tuple) else x
        y = y[0] if isinstance(y, tuple) else y
        for key, t in (("input", x), ("output", y)):
            v = float(t.detach().abs().max())
            act_dict[name][key] = max(act_dict[name].get(key, 0.0), v)

    hooks = [m.register_forward_hook(functools.partial(hook, name=n)) for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)]
    try:
        _run(model, batches)
    finally:
        for h in hooks:
            h.remove()
    return act_dict


# name of the Linear whose input / output absmax gives each scale, per architecture
# (reference collect_{transformers,llama,baichuan,mixtral}_layer_scales, calibration.py:91-183)
_LAYER_KEYS = {
    "transformers": ("model.decoder.layers.{i}.", {
        "attn_input_scale": ("self_attn.q_proj", "input"), "q_output_scale": ("self_attn.q_proj", "output"),
        "k_output_scale": ("self_attn.k_proj", "output"), "v_output_scale": ("self_attn.v_proj", "output"),
        "out_input_scale": ("self_attn.out_proj", "input"), "fc1_input_scale": ("fc1", "input"), "fc2_input_scale": ("fc2", "input")}),
    "llama": ("model.layers.{i}.", {
        "attn_input_scale": ("self_attn.q_proj", "input"), "q_output_scale": ("self_attn.q_proj", "output"),
        "k_output_scale": ("self_attn.k_proj", "output"), "v_output_scale": ("self_attn.v_proj", "output"),
        "out_input_scale": ("self_attn.o_proj", "input"), "gate_input_scale": ("mlp.gate_proj", "input"),
        "down_input_scale": ("mlp.down_proj", "input")}),
    "baichuan": ("model.layers.{i}.", {
        "attn_input_scale": ("self_attn.W_pack", "input"), "attn_output_scale": ("self_attn.W_pack", "output"),
        "out_input_scale": ("self_attn.o_proj", "input"), "gate_input_scale": ("mlp.gate_proj", "input"),
        "down_input_scale": ("mlp.down_proj", "input")}),
    "mixtral": ("model.layers.{i}.", {
        "attn_input_scale": ("self_attn.q_proj", "input"), "q_output_scale": ("self_attn.q_proj", "output"),
        "k_output_scale": ("self_attn.k_proj", "output"), "v_output_scale": ("self_attn.v_proj", "output"),
        "out_input_scale": ("self_attn.o_proj", "input"), "moe_input_scale": ("block_sparse_moe.gate", "input")}),
}


def decoder_layer_scales(act_dict, num_layers, model_type="transformers", num_local_experts=0):
    """absmax / 127 per decoder layer under the reference's key names; ``model_type`` in
    {"transformers" (OPT), "llama", "baichuan", "mixtral"}.  Mixtral adds ``down_input_scales`` (one per expert w2)."""
    if model_type not in _LAYER_KEYS:
        raise ValueError(f"unsupport model type: {model_type}")
    prefix, keys = _LAYER_KEYS[model_type]
    out = []
    for i in range(num_layers):
        p = prefix.format(i=i)
        d = {k: act_dict[p + mod][io] / 127 for k, (mod, io) in keys.items()}
        if model_type == "mixtral":
            missing = [e for e in range(num_local_experts) if "input" not in act_dict.get(f"{p}block_sparse_moe.experts.{e}.w2", {})]
            if missing:
                raise RuntimeError(f"layer {i}: experts {missing} received no calibration tokens (no w2 activation statistics); calibrate inside "
                                   "`with all_experts_routed(model):` (the reference forces top_k = num_local_experts, quantize/calibration.py:23-35)")
            d["down_input_scales"] = [act_dict[f"{p}block_sparse_moe.experts.{e}.w2"]["input"] / 127 for e in range(num_local_experts)]
        out.append(d)
    return out


def get_static_decoder_layer_scales(model, batches, num_layers, model_type="transformers", num_local_experts=0):
    """(decoder_layer_scales, act_dict), as the reference's function of the same name (calibration.py:185-244).  For Mixtral the run
    happens with every expert routed (all_experts_routed), as in the reference."""
    with all_experts_routed(model) if model_type == "mixtral" else contextlib.nullcontext():
        act_dict = get_io_absmax(model, batches)
    return decoder_layer_scales(act_dict, num_layers, model_type, num_local_experts), act_dict


def parse_quant_config(config_path):
    """quant_config.json -> dict, e.g. {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    (reference utils/utils.py:35-39)."""
    with open(config_path, "r", encoding="utf-8") as f:
        cfg = json.load(f)
    bad = {k: v for k, v in cfg.items() if k in ("qkv", "out", "fc1", "fc2") and v not in ("per-tensor", "per-token")}
    if bad:
        raise ValueError(f"quant_config: unknown act_quant values {bad}")
    return cfg
