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


# ---- the reference's dataset / tokenizer entry points (quantize/calibration.py:44-88, :185-244, :292-339): thin adapters over the collectors above -----------
def dataset_batches(tokenizer, dataset_path, num_samples, seq_len, device=None):
    """The model inputs the reference's calibration loops build: `load_dataset("json", data_files=dataset_path, split="train").shuffle(seed=42)`, the first
    num_samples texts through `tokenizer(text, return_tensors="pt", max_length=seq_len, truncation=True).input_ids` (calibration.py:73-85)."""
    from datasets import load_dataset
    dataset = load_dataset("json", data_files=dataset_path, split="train").shuffle(seed=42)
    for i in range(num_samples):
        ids = tokenizer(dataset[i]["text"], return_tensors="pt", max_length=seq_len, truncation=True).input_ids
        yield ids.to(device) if device is not None else ids


def _model_device(model):
    p = next(model.parameters(), None)
    return p.device if p is not None else None


def get_act_scales_from_dataset(model, tokenizer, dataset_path, num_samples=512, seq_len=512):
    """`get_act_scales(model, tokenizer, dataset_path, num_samples, seq_len)` with the reference's signature (calibration.py:44)."""
    return get_act_scales(model, dataset_batches(tokenizer, dataset_path, num_samples, seq_len, _model_device(model)))


def get_static_decoder_layer_scales_from_dataset(model, tokenizer, dataset_path, num_samples=512, seq_len=512, model_type="transformers"):
    """The reference's `get_static_decoder_layer_scales(model, tokenizer, dataset_path, num_samples, seq_len, model_type)` (calibration.py:185-244):
    num_layers and num_local_experts come from model.config as there."""
    cfg = model.config
    return get_static_decoder_layer_scales(model, dataset_batches(tokenizer, dataset_path, num_samples, seq_len, _model_device(model)), cfg.num_hidden_layers,
                                           model_type, getattr(cfg, "num_local_experts", 0))


def replace_module(model, name, new_module):
    """setattr on the parent of the dotted module path `name` (reference calibration.py:247-256)."""
    if "." in name:
        parent_name, child_name = name.rsplit(".", 1)
        parent = model.get_submodule(parent_name)
    else:
        parent, child_name = model, name
    setattr(parent, child_name, new_module)


def get_layers_to_ignore(model, ignore_patterns):
    """Names of the nn.Linear modules matched by the patterns: "re:<regex>" -> re.search on the name, anything else an exact name
    (reference calibration.py:259-279)."""
    import re
    ignored = set()
    for name, linear in model.named_modules():
        if not isinstance(linear, torch.nn.Linear):
            continue
        for pat in ignore_patterns:
            if pat.startswith("re:"):
                if re.search(pat[3:], name):
                    ignored.add(name)
            elif pat == name:
                ignored.add(name)
    return list(ignored)


@torch.no_grad()
def quantize_activations_fp8(model, tokenizer=None, calibration_path=None, ignore_patterns=(), num_samples=None, batches=None):
    """fp8 (e4m3) STATIC activation calibration, the reference's function of the same name (calibration.py:292-339): every nn.Linear that no pattern
    ignores becomes an FP8StaticLinearQuantizer over its per-tensor-quantised weight (weight_scale = absmax / 448), then the calibration inputs run
    through the model and each quantizer keeps the running maximum of its dynamic per-tensor input scale.  Afterwards FP8LinearStatic.from_float(quantizer)
    freezes a module.  Inputs: the reference's (tokenizer, calibration_path, num_samples) -- max_length 1024 as there -- or `batches`, any iterable of
    model inputs.  The model must live on the HIP device: the quantizers run the library's fp8 quantiser and GEMM."""
    import copy
    from ..layers.nn.linear import FP8StaticLinearQuantizer
    from ..layers.functional.quantization import per_tensor_quantize_fp8
    ignored = set(get_layers_to_ignore(model, list(ignore_patterns)))
    for name, linear in list(model.named_modules()):
        if not isinstance(linear, torch.nn.Linear) or name in ignored:
            continue
        qw, ws = per_tensor_quantize_fp8(linear.weight)
        bias = copy.deepcopy(linear.bias) if linear.bias is not None else None
        replace_module(model, name, FP8StaticLinearQuantizer(linear.in_features, linear.out_features, weight=qw.to(linear.weight.device), weight_scale=ws, bias=bias,
                                                             quantize_output=False))
    if batches is None:
        if tokenizer is None or calibration_path is None or num_samples is None:
            raise ValueError("quantize_activations_fp8: pass (tokenizer, calibration_path, num_samples) or batches")
        batches = dataset_batches(tokenizer, calibration_path, num_samples, 1024, _model_device(model))
    _run(model, batches)
    return model
