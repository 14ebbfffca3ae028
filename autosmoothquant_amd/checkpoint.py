"""Loader for REFERENCE-FORMAT quantised checkpoints.

The reference writes a quantised model with ``quant_model.save_pretrained(output_path)`` plus a ``quant_config.json`` beside it
(reference examples/smoothquant_model.py:92-99) and reads it back through ``Int8*ForCausalLM.from_pretrained(path, quant_config)``
(examples/test_model.py:26-40).  What is in the directory:

    config.json           the HF model config (hidden_size, intermediate_size, num_attention_heads, num_hidden_layers, rms_norm_eps ...)
    quant_config.json     {"qkv": ..., "out": ..., "fc1": ..., "fc2": ...} with values "per-tensor" | "per-token"   (utils/utils.py:35-39)
    model.safetensors | model-0000x-of-0000y.safetensors + model.safetensors.index.json | pytorch_model.bin
        <prefix>.layers.<i>.self_attn.{q,k,v}_proj | W_pack .weight int8 [N,K], .bias f32 [N] (iff biased),
                 .dequant_scale f32 [] | .{q,k,v}_dequant_scale f32 []                      (layers/nn/linear.py:49-66, :138-149)
        <prefix>.layers.<i>.self_attn.o_proj / mlp.down_proj   + .quant_scale f32 [] iff per-tensor          (:253-256)
        <prefix>.layers.<i>.mlp.{gate,up}_proj
        <prefix>.layers.<i>.{input,post_attention}_layernorm.weight      float, ALREADY divided by the input scale when the
                 consuming linears are per-tensor (models/llama.py:27-37,326-339; models/baichuan.py:49-59)
        <prefix>.embed_tokens.weight, <prefix>.norm.weight, lm_head.weight     float, unquantised (models/llama.py:409-410)

`load_reference_checkpoint` walks that directory and builds the decoder stack out of this package's own modules (the drop-in classes
of layers/nn/linear.py inside the harness layers), moving every buffer to the target device.  The module skeleton is derived
from quant_config + the key names, then each module's buffers are loaded STRICTLY (unexpected / missing / mis-typed entries raise):
the loader doubles as the check of the buffer-name contract.  LLaMA-style (separate q/k/v) and Baichuan-style (W_pack) layers are
supported -- the two layer harnesses this package has; OPT / Mixtral directories raise NotImplementedError naming the first
unsupported key.
"""
import json
import os
import re

import torch

from . import harness
from .layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, W8A8BFP32OFP32QKVLinear
from .quantize.calibration import parse_quant_config

_EXPECT = {"weight": torch.int8, "bias": torch.float32, "dequant_scale": torch.float32, "quant_scale": torch.float32,
           "q_dequant_scale": torch.float32, "k_dequant_scale": torch.float32, "v_dequant_scale": torch.float32}


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
        if k in _EXPECT and t.dtype != _EXPECT[k]:
            raise TypeError(f"{where}.{k}: checkpoint dtype {t.dtype}, the contract (layers/nn/linear.py) says {_EXPECT[k]}")
    mod.load_state_dict(sd, strict=True)   # names and shapes; raises on missing / unexpected keys
    mod._pin_scalars()
    return mod


def _linear(cls, sd, act_quant, where, qkv_size=None):
    n, k = sd["weight"].shape
    args = (k, n, "bias" in sd, act_quant)
    mod = cls(qkv_size, *args) if qkv_size is not None else cls(*args)
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


def load_reference_checkpoint(path, device="cuda", dtype=None):
    """Directory written by the reference (see the module docstring) -> QuantizedDecoder on `device`.
    `dtype`: floating dtype for the norm weights / embeddings (default: as stored).  int8 weights, fp32 biases and the
    host-pinned fp32 scalar scales keep their contract dtypes whatever `dtype` is."""
    with open(os.path.join(path, "config.json")) as f:
        config = json.load(f)
    qc = parse_quant_config(os.path.join(path, "quant_config.json"))
    for k in ("qkv", "out", "fc1", "fc2"):
        if k not in qc:
            raise KeyError(f"quant_config.json lacks {k!r} (reference utils/utils.py:35-39 / models/*.py read all four)")
    if qc.get("type", "int8") not in ("int8",):
        raise NotImplementedError(f"quant_config type {qc.get('type')!r}: this loader handles the int8 (W8A8) checkpoints")
    tensors = read_tensors(path)
    keys = list(tensors)
    m = next((re.match(r"^(.*?)layers\.0\.", k) for k in keys if re.match(r"^(.*?)layers\.0\.", k)), None)
    if m is None:
        raise ValueError(f"{path}: no '<prefix>layers.0.*' entries")
    prefix = m.group(1)
    if any(".fc1." in k or "block_sparse_moe" in k for k in keys):
        bad = next(k for k in keys if ".fc1." in k or "block_sparse_moe" in k)
        raise NotImplementedError(f"{bad}: OPT / Mixtral layer harnesses are not part of this package (LLaMA- and Baichuan-style layers are)")
    H, heads, L = config["hidden_size"], config["num_attention_heads"], config["num_hidden_layers"]
    eps = config.get("rms_norm_eps", 1e-6)
    baichuan = any(".self_attn.W_pack." in k for k in keys)
    layers = []
    for i in range(L):
        p = f"{prefix}layers.{i}."
        sub = _take(tensors, p)
        if not sub:
            raise KeyError(f"layer {i}: no tensors under {p}")

        def part(name):
            return _take(sub, name + ".")
        if baichuan:
            lay = harness.BaichuanLayer.__new__(harness.BaichuanLayer)
            torch.nn.Module.__init__(lay)
            lay.hidden, lay.heads, lay.hd, lay.int8 = H, heads, H // heads, True
            lay.W_pack = _linear(W8A8BFP32OFP32QKVLinear, part("self_attn.W_pack"), qc["qkv"], p + "self_attn.W_pack", qkv_size=[H, H, H])
            norm_cls = harness.BaichuanRMSNorm
        else:
            lay = harness.LlamaLayer.__new__(harness.LlamaLayer)
            torch.nn.Module.__init__(lay)
            kvh = config.get("num_key_value_heads") or heads
            lay.hidden, lay.heads, lay.kv_heads, lay.hd = H, heads, kvh, H // heads
            for n in ("q_proj", "k_proj", "v_proj"):
                setattr(lay, n, _linear(W8A8BFP32OFP32Linear, part("self_attn." + n), qc["qkv"], p + "self_attn." + n))
            norm_cls = harness.RMSNorm
        lay.o_proj = _linear(W8A8BFP32OFP32LinearWithQuantScale, part("self_attn.o_proj"), qc["out"], p + "self_attn.o_proj")
        lay.gate_proj = _linear(W8A8BFP32OFP32Linear, part("mlp.gate_proj"), qc["fc1"], p + "mlp.gate_proj")
        lay.up_proj = _linear(W8A8BFP32OFP32Linear, part("mlp.up_proj"), qc["fc1"], p + "mlp.up_proj")
        lay.down_proj = _linear(W8A8BFP32OFP32LinearWithQuantScale, part("mlp.down_proj"), qc["fc2"], p + "mlp.down_proj")
        for n in ("input_layernorm", "post_attention_layernorm"):
            w = part(n)
            if set(w) != {"weight"}:
                raise KeyError(f"{p}{n}: expected exactly 'weight', got {sorted(w)}")
            nm = norm_cls(H, eps)
            nm.weight = torch.nn.Parameter(w["weight"].to(dtype) if dtype is not None else w["weight"], requires_grad=False)
            setattr(lay, n, nm)
        sub = {k: v for k, v in sub.items() if not k.endswith("rotary_emb.inv_freq")}   # (HF persists this buffer in old versions)
        if sub:
            raise KeyError(f"{p}: unexpected tensors {sorted(sub)}")
        layers.append(lay)
    extras = {}
    for k in list(tensors):
        if k in (prefix + "embed_tokens.weight", prefix + "norm.weight", "lm_head.weight"):
            t = tensors.pop(k)
            extras[k[len(prefix):] if k.startswith(prefix) and prefix else k] = t.to(dtype) if dtype is not None else t
    if tensors:
        raise KeyError(f"{path}: tensors not consumed by the loader: {sorted(tensors)[:8]}{' ...' if len(tensors) > 8 else ''}")
    model = QuantizedDecoder(layers, config, qc, "baichuan" if baichuan else "llama", extras)
    return model.to(device)
