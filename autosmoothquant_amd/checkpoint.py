"""Loader for REFERENCE-FORMAT quantised checkpoints.

The reference writes a quantised model with ``quant_model.save_pretrained(output_path)`` plus a ``quant_config.json`` beside it
(reference examples/smoothquant_model.py:92-99) and reads it back through ``Int8*ForCausalLM.from_pretrained(path, quant_config)``
(examples/test_model.py:26-40).  What is in the directory:

    config.json           the HF model config (hidden_size, num_attention_heads, num_hidden_layers, rope_theta, ffn_dim / intermediate_size ...)
    quant_config.json     {"qkv": ..., "out": ..., "fc1": ..., "fc2": ...} with values "per-tensor" | "per-token"   (utils/utils.py:35-39)
                          + optional "type": "int8" | "fp8" (= "fp8_e4m3") | "fp8_e4m3" | "fp8_e5m2" and "activation_scheme": "dynamic" | "static"
                          (examples/smoothquant_model.py:62-70; LLaMA only: models/llama.py:76-103,187-210)
    model.safetensors | model-0000x-of-0000y.safetensors + model.safetensors.index.json | pytorch_model.bin

    LLaMA     model.layers.N.self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj, {input,post_attention}_layernorm.weight      (models/llama.py)
    Baichuan  model.layers.N.self_attn.{W_pack (QKVLinear: {q,k,v}_dequant_scale), o_proj}, mlp.*, *_layernorm.weight            (models/baichuan.py)
    OPT       model.decoder.layers.N.self_attn.{q,k,v,out}_proj (+ .bias f32), fc1, fc2, {self_attn,final}_layer_norm.{weight,bias}  (models/opt.py:76-81,125-129)
    Mixtral   model.layers.N.self_attn.{q,k,v,o}_proj, block_sparse_moe.gate.weight (float), block_sparse_moe.experts.E.{w1,w2,w3},
              {input,post_attention}_layernorm.weight                                                                          (models/mixtral.py:64-67,99-145)
    int8 linears: .weight int8 [N,K], .bias f32 [N] (iff biased), .dequant_scale f32 [] (+ .quant_scale f32 [] for per-tensor o / down / fc2 / w2;
                  layers/nn/linear.py:49-66,138-149,253-256); fp8: .weight float8_e4m3fn | float8_e5m2, .weight_scale [, .input_scale, .output_scale]
                  (:387-410,515-549,595-608).  Norm weights (and OPT's norm biases) arrive ALREADY divided by the input scale when the consuming
                  linears are per-tensor int8 (models/llama.py:27-37,326-339; opt.py:20-29,150-163; mixtral.py:22-30; baichuan.py:49-59).
    embeddings, final norm, lm_head (and OPT's embed_positions / project_in / project_out): float, unquantised.

`load_reference_checkpoint` walks that directory and builds the decoder stack out of this package's own modules (the drop-in classes of
layers/nn/linear.py inside the harness layers), moving every buffer to the target device.  The module skeleton is derived from quant_config + the
key names, then each module's buffers are loaded STRICTLY (unexpected / missing / mis-typed entries raise): the loader doubles as the check of the
buffer-name contract.

What the returned stack computes: the LAYERS.  The reference borrows every layer's `forward` from Hugging Face (attention masks, KV cache, rotary
variants, ALiBi are transformers code); the harness layers restate one fixed variant each -- causal attention with RoPE at the config's
`rope_theta` (LLaMA, Mixtral; any `rope_scaling` is refused), causal attention with pre-scaled queries (OPT), and for Baichuan the branch the reference
itself can execute on CPU (no mask, no positional term) -- and are checked against layer I/O recorded from the reference's modules
(tests/golden/ckpt_*).  Embedding, positions and sampling stay with the caller.
"""
import json
import os
import re

import torch

from . import harness
from .layers.nn.linear import (W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, W8A8BFP32OFP32QKVLinear, FP8LinearDynamic, FP8LinearStatic,
                               FP8E5M2Linear)
from .quantize.calibration import parse_quant_config

_F32 = ("bias", "dequant_scale", "quant_scale", "q_dequant_scale", "k_dequant_scale", "v_dequant_scale", "weight_scale", "input_scale", "output_scale")


def read_tensors(path):
    """name -> CPU tensor for every weight file in the directory (safetensors, sharded safetensors or pytorch_model.bin)."""
    files = []
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(path, "model.safetensors")):
        files = ["model.safetensors"]
    out = {}
    if files:
        from safetensors import safe_open
        for fn in files:
            with safe_open(os.path.join(path, fn), "pt", device="cpu") as f:
                for k in f.keys():
                    out[k] = f.get_tensor(k)
        return out
    binf = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(binf):
        return dict(torch.load(binf, map_location="cpu", weights_only=True))
    raise FileNotFoundError(f"{path}: no model.safetensors[.index.json] or pytorch_model.bin")


def _take(tensors, prefix):
    n = len(prefix)
    return {k[n:]: tensors.pop(k) for k in [k for k in tensors if k.startswith(prefix)]}


def _load_module(mod, sd, where):
    for k, t in sd.items():
        want = mod._buffers["weight"].dtype if k == "weight" else (torch.float32 if k in _F32 else None)
        if want is not None and t.dtype != want:
            raise TypeError(f"{where}.{k}: checkpoint dtype {t.dtype}, the contract (layers/nn/linear.py) says {want}")
    mod.load_state_dict(sd, strict=True)   # names and shapes; raises on missing / unexpected keys
    if hasattr(mod, "_pin_scalars"):
        mod._pin_scalars()
    return mod


def quant_type(qc):
    t = qc.get("type", "int8")
    t = "fp8_e4m3" if t == "fp8" else t                       # examples/smoothquant_model.py:66-67
    if t not in ("int8", "fp8_e4m3", "fp8_e5m2"):
        raise ValueError(f"quant_config type {t!r}: expected int8 | fp8 | fp8_e4m3 | fp8_e5m2")
    if t == "fp8_e4m3" and qc.get("activation_scheme", "dynamic") not in ("dynamic", "static"):
        raise ValueError(f"quant_config activation_scheme {qc.get('activation_scheme')!r}: expected dynamic | static")
    return t


def _linear(cls, sd, act_quant, where, qkv_size=None):
    n, k = sd["weight"].shape
    args = (k, n, "bias" in sd, act_quant)
    mod = cls(qkv_size, *args) if qkv_size is not None else cls(*args)
    return _load_module(mod, sd, where)


def _proj(sd, qc, kind, where, with_quant_scale):
    """One projection of a LLaMA-family layer: the module class follows quant_config['type'] exactly as the reference's constructors do
    (models/llama.py:76-103 attention, :187-210 MLP)."""
    t = quant_type(qc)
    if t == "int8":
        return _linear(W8A8BFP32OFP32LinearWithQuantScale if with_quant_scale else W8A8BFP32OFP32Linear, sd, qc[kind], where)
    n, k = sd["weight"].shape
    if t == "fp8_e5m2":
        mod = FP8E5M2Linear(k, n, use_bias="bias" in sd)
    elif qc.get("activation_scheme", "dynamic") == "static":
        mod = FP8LinearStatic(k, n, use_bias="bias" in sd)
    else:
        mod = FP8LinearDynamic(k, n, qc[kind], use_bias="bias" in sd)
    return _load_module(mod, sd, where)


class QuantizedDecoder(torch.nn.Module):
    """The loaded stack: `layers` (harness layers holding the W8A8 modules), plus the unquantised embedding / final norm / lm_head
    when the checkpoint has them.  forward(hidden_states [B,S,H]) runs the decoder layers."""

    def __init__(self, layers, config, quant_config, arch, extras):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)
        self.config, self.quant_config, self.arch = config, quant_config, arch
        for name, t in extras.items():
            self.register_buffer(name.replace(".", "_"), t)

    @torch.no_grad()
    def forward(self, h):
        for l in self.layers:
            h = l(h)
        return h.materialize() if hasattr(h, "materialize") else h


def _norm_weight(sub, name, where, with_bias=False):
    w = _take(sub, name + ".")
    want = {"weight", "bias"} if with_bias else {"weight"}
    if set(w) != want:
        raise KeyError(f"{where}{name}: expected exactly {sorted(want)}, got {sorted(w)}")
    return w


def _cast(t, dtype):
    return t.to(dtype) if dtype is not None and t.is_floating_point() else t


def load_reference_checkpoint(path, device="cuda", dtype=None):
    """Directory written by the reference (see the module docstring) -> QuantizedDecoder on `device`.
    `dtype`: floating dtype for the norm weights / embeddings / router (default: as stored).  int8 / fp8 weights, fp32 biases and the
    host-pinned fp32 scalar scales keep their contract dtypes whatever `dtype` is."""
    with open(os.path.join(path, "config.json")) as f:
        config = json.load(f)
    qc = parse_quant_config(os.path.join(path, "quant_config.json"))
    for k in ("qkv", "out", "fc1", "fc2"):
        if k not in qc:
            raise KeyError(f"quant_config.json lacks {k!r} (reference utils/utils.py:35-39 / models/*.py read all four)")
    qtype = quant_type(qc)
    tensors = read_tensors(path)
    keys = list(tensors)
    m = next((re.match(r"^(.*?)layers\.0\.", k) for k in keys if re.match(r"^(.*?)layers\.0\.", k)), None)
    if m is None:
        raise ValueError(f"{path}: no '<prefix>layers.0.*' entries")
    prefix = m.group(1)
    arch = ("opt" if any(".fc1." in k for k in keys) else "mixtral" if any(".block_sparse_moe." in k for k in keys)
            else "baichuan" if any(".self_attn.W_pack." in k for k in keys) else "llama")
    if qtype != "int8" and arch != "llama":
        raise NotImplementedError(f"quant_config type {qtype!r} on a {arch} checkpoint: the reference builds fp8 modules for LLaMA only "
                                  "(models/llama.py:137-176,246-283; opt.py / mixtral.py / baichuan.py have no from_float_to_fp8)")
    if config.get("rope_scaling") not in (None, {}):
        raise NotImplementedError(f"config.json rope_scaling = {config['rope_scaling']!r}: the layer harness implements plain RoPE (rope_theta only)")
    H, heads, L = config["hidden_size"], config["num_attention_heads"], config["num_hidden_layers"]
    layers = []
    for i in range(L):
        p = f"{prefix}layers.{i}."
        sub = _take(tensors, p)
        if not sub:
            raise KeyError(f"layer {i}: no tensors under {p}")

        def part(name):
            return _take(sub, name + ".")
        if arch == "opt":
            lay = harness.OptLayer.__new__(harness.OptLayer)
            torch.nn.Module.__init__(lay)
            lay.hidden, lay.heads, lay.hd, lay.pre_ln = H, heads, H // heads, bool(config.get("do_layer_norm_before", True))
            for n in ("q_proj", "k_proj", "v_proj"):   # models/opt.py:78-80
                setattr(lay, n, _linear(W8A8BFP32OFP32Linear, part("self_attn." + n), qc["qkv"], p + "self_attn." + n))
            lay.out_proj = _linear(W8A8BFP32OFP32LinearWithQuantScale, part("self_attn.out_proj"), qc["out"], p + "self_attn.out_proj")   # :81
            lay.fc1 = _linear(W8A8BFP32OFP32Linear, part("fc1"), qc["fc1"], p + "fc1")                                                      # :127
            lay.fc2 = _linear(W8A8BFP32OFP32LinearWithQuantScale, part("fc2"), qc["fc2"], p + "fc2")                                        # :128
            for n in ("self_attn_layer_norm", "final_layer_norm"):
                w = _norm_weight(sub, n, p, with_bias=True)
                nm = torch.nn.LayerNorm(H, 1e-5)    # HF OPT uses nn.LayerNorm's default eps
                nm.weight = torch.nn.Parameter(_cast(w["weight"], dtype), requires_grad=False)
                nm.bias = torch.nn.Parameter(_cast(w["bias"], dtype), requires_grad=False)
                setattr(lay, n, nm)
        else:
            eps = config.get("rms_norm_eps", 1e-6)
            kvh = config.get("num_key_value_heads") or heads
            if arch == "baichuan":
                lay = harness.BaichuanLayer.__new__(harness.BaichuanLayer)
                torch.nn.Module.__init__(lay)
                lay.hidden, lay.heads, lay.hd, lay.int8 = H, heads, H // heads, True
                lay.W_pack = _linear(W8A8BFP32OFP32QKVLinear, part("self_attn.W_pack"), qc["qkv"], p + "self_attn.W_pack", qkv_size=[H, H, H])
                norm_cls = harness.BaichuanRMSNorm
            else:
                cls = harness.MixtralLayer if arch == "mixtral" else harness.LlamaLayer
                lay = cls.__new__(cls)
                torch.nn.Module.__init__(lay)
                lay.hidden, lay.heads, lay.kv_heads, lay.hd = H, heads, kvh, H // heads
                lay.rope_theta = float(config.get("rope_theta", 10000.0))
                for n in ("q_proj", "k_proj", "v_proj"):
                    setattr(lay, n, _proj(part("self_attn." + n), qc, "qkv", p + "self_attn." + n, False))
                norm_cls = harness.RMSNorm
            lay.o_proj = _proj(part("self_attn.o_proj"), qc, "out", p + "self_attn.o_proj", True)
            if arch == "mixtral":
                lay.top_k = int(config.get("num_experts_per_tok", 2))
                g = part("block_sparse_moe.gate")
                if set(g) != {"weight"}:
                    raise KeyError(f"{p}block_sparse_moe.gate: expected exactly 'weight' (the router is not quantised, models/mixtral.py:137), got {sorted(g)}")
                lay.gate = torch.nn.Linear(H, g["weight"].shape[0], bias=False)
                lay.gate.weight = torch.nn.Parameter(_cast(g["weight"], dtype), requires_grad=False)
                E = int(config.get("num_local_experts", g["weight"].shape[0]))
                lay.experts = torch.nn.ModuleList()
                for e in range(E):
                    pe = f"block_sparse_moe.experts.{e}."
                    ex = harness.ExpertMLP()
                    ex.w1 = _linear(W8A8BFP32OFP32Linear, part(pe + "w1"), qc["fc1"], p + pe + "w1")                    # models/mixtral.py:99
                    ex.w2 = _linear(W8A8BFP32OFP32LinearWithQuantScale, part(pe + "w2"), qc["fc2"], p + pe + "w2")      # :100
                    ex.w3 = _linear(W8A8BFP32OFP32Linear, part(pe + "w3"), qc["fc1"], p + pe + "w3")                    # :101
                    lay.experts.append(ex)
            else:
                lay.gate_proj = _proj(part("mlp.gate_proj"), qc, "fc1", p + "mlp.gate_proj", False)
                lay.up_proj = _proj(part("mlp.up_proj"), qc, "fc1", p + "mlp.up_proj", False)
                lay.down_proj = _proj(part("mlp.down_proj"), qc, "fc2", p + "mlp.down_proj", True)
            for n in ("input_layernorm", "post_attention_layernorm"):
                w = _norm_weight(sub, n, p)
                nm = norm_cls(H, eps)
                nm.weight = torch.nn.Parameter(_cast(w["weight"], dtype), requires_grad=False)
                setattr(lay, n, nm)
        sub = {k: v for k, v in sub.items() if not k.endswith("rotary_emb.inv_freq")}   # (HF persists this buffer in old versions)
        if sub:
            raise KeyError(f"{p}: unexpected tensors {sorted(sub)}")
        layers.append(lay)
    extras = {}
    tail = ("embed_tokens.weight", "norm.weight", "embed_positions.weight", "final_layer_norm.weight", "final_layer_norm.bias", "project_in.weight",
            "project_out.weight")
    for k in list(tensors):
        if k == "lm_head.weight" or (k.startswith(prefix) and k[len(prefix):] in tail):
            extras[k[len(prefix):] if k != "lm_head.weight" else k] = _cast(tensors.pop(k), dtype)
    if tensors:
        raise KeyError(f"{path}: tensors not consumed by the loader: {sorted(tensors)[:8]}{' ...' if len(tensors) > 8 else ''}")
    model = QuantizedDecoder(layers, config, qc, arch, extras).to(device)
    if arch == "mixtral":
        for lay in model.layers:
            lay.stack_experts()   # (after the move: the stacks are device tensors, the per-expert buffers views of them)
    return model
