#!/usr/bin/env python3
"""Reference-format quantised checkpoints for the OTHER model families: OPT, Mixtral, and an
fp8 (e4m3, dynamic) LLaMA with GQA and a non-default rope_theta.  Data only (safetensors + JSON + recorded layer I/O).

The reference's own wrappers (models/opt.py, models/mixtral.py, models/llama.py) subclass `transformers == 4.42.3` internals that the
installed transformers no longer has, so they cannot be imported here.  What CAN be imported is everything they are made of:

  * the reference's linear classes (layers/nn/linear.py: W8A8BFP32OFP32Linear / ...WithQuantScale / FP8LinearDynamic, `from_float`), with
    the 6-line `_CUDA` stub whose GEMM is an exact integer matmul;
  * Hugging Face's float decoder layers (OPTDecoderLayer, MixtralAttention, LlamaDecoderLayer of the installed transformers), whose
    `forward` is what the reference borrows (models/opt.py:84,131; models/mixtral.py:75,250; models/llama.py:111,218,289).

Each quantised layer below is therefore the HF layer with its projections REPLACED by reference modules exactly as the reference's
`from_float` / `from_float_to_fp8` do (which class, which act_quant, which input scale: models/opt.py:97-104,150-163;
models/mixtral.py:88-91,114-116,166-176; models/llama.py:137-176,246-283), norm weights folded by the reference's formulas
(opt.py:20-29, mixtral.py:22-30) iff the consuming linears are per-tensor.  Mixtral's sparse-MoE block is restated from transformers
4.42.3's MixtralSparseMoeBlock.forward (softmax in fp32 -> top-k -> renormalise -> per-expert w2(act(w1 x) * w3 x) -> index_add),
because the installed transformers fuses the experts into 3-D parameters.  Tensors are stored under the reference's key names
(what `save_pretrained` of Int8OPTForCausalLM / Int8MixtralForCausalLM / QuantizedLlamaForCausalLM writes).

  tests/golden/ckpt_opt_w8a8/         model.decoder.layers.N.{self_attn.{q,k,v,out}_proj, fc1, fc2, self_attn_layer_norm, final_layer_norm}
  tests/golden/ckpt_mixtral_w8a8/     model.layers.N.{self_attn.{q,k,v,o}_proj, block_sparse_moe.{gate, experts.E.{w1,w2,w3}}, *_layernorm}
  tests/golden/ckpt_llama_fp8_e4m3/   model.layers.N.{self_attn.*_proj, mlp.{gate,up,down}_proj} as FP8LinearDynamic buffers
each with config.json, quant_config.json, io.npz (x, y_layers = the hidden state after every layer, causal attention).
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt_models.py
"""
import copy
import json
import os
import shutil
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

stub = types.ModuleType("autosmoothquant._CUDA")


class I8CUGEMM:  # csrc/int8gemm/bindings.cpp:145-155; exact integer matmul == CUBLAS_COMPUTE_32I, alpha 1, beta 0
    def linear_a8_w8_o32_(self, x, w, out):
        out.copy_(x.to(torch.int32) @ w.to(torch.int32).t())


stub.I8CUGEMM = I8CUGEMM
sys.modules["autosmoothquant._CUDA"] = stub
torch.cuda.current_device = lambda: torch.device("cpu")
sys.path.insert(0, "/root/reference")
import autosmoothquant.layers.nn.linear as RL  # noqa: E402

import detrng  # noqa: E402
from safetensors.torch import save_file  # noqa: E402


def det_init(module, seed, std=0.05):
    with torch.no_grad():
        for i, (n, p) in enumerate(module.named_parameters()):
            v = detrng.normal(seed, i, tuple(p.shape)).astype(np.float32) * std
            if ("norm" in n and n.endswith("weight")):
                v = v + 1.0
            p.copy_(torch.from_numpy(v))


def causal_mask(S):
    return torch.full((S, S), float("-inf")).triu(1)[None, None]


def absmax_hooks(mods, rec):
    return [m.register_forward_hook(lambda m_, i, o, n=n: rec.__setitem__(n, max(rec.get(n, 0.0), float(i[0].abs().max()) / 127.0))) for n, m in mods]


def finish(out, tensors, config, qc, x, ys, extra=None):
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(out, "model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(out, "config.json"), "w") as f:
        json.dump(config, f, indent=1, sort_keys=True)
    with open(os.path.join(out, "quant_config.json"), "w") as f:   # examples/smoothquant_model.py:96-99
        json.dump(qc, f, indent=4)
    np.savez_compressed(os.path.join(out, "io.npz"), x=x, y_layers=np.stack(ys), **(extra or {}))
    print(os.path.basename(out), sorted(os.listdir(out)), {k: (str(v.dtype), tuple(v.shape)) for k, v in list(tensors.items())[:3]})


def put(tensors, prefix, mod):
    for k, v in mod.state_dict().items():
        tensors[prefix + k] = v.detach().clone()


# ------------------------------------------------------------------------------------------------------------------ OPT
def make_opt():
    from transformers.models.opt.modeling_opt import OPTConfig, OPTDecoderLayer
    H, FFN, HEADS, L, V, B, S = 128, 256, 4, 2, 40, 2, 20
    cfg = OPTConfig(hidden_size=H, ffn_dim=FFN, num_attention_heads=HEADS, num_hidden_layers=L, vocab_size=V, max_position_embeddings=64,
                    word_embed_proj_dim=H, do_layer_norm_before=True, enable_bias=True, dropout=0.0, attention_dropout=0.0, activation_function="relu")
    cfg._attn_implementation = "eager"
    qc = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    layers = [OPTDecoderLayer(cfg, layer_idx=i).eval() for i in range(L)]
    for i, l in enumerate(layers):
        det_init(l, 910 + i)
    x = detrng.normal(911, 0, (B, S, H)).astype(np.float32)
    x[..., 7] *= 10.0
    mask = causal_mask(S)
    scales, h = [], torch.from_numpy(x.copy())
    with torch.no_grad():
        for l in layers:
            rec = {}
            hk = absmax_hooks([("attn_input_scale", l.self_attn.q_proj), ("out_input_scale", l.self_attn.out_proj), ("fc1_input_scale", l.fc1),
                               ("fc2_input_scale", l.fc2)], rec)
            h = l(h, attention_mask=mask)
            h = h[0] if isinstance(h, tuple) else h
            for k_ in hk:
                k_.remove()
            scales.append(rec)
    y_float = h.numpy().copy()
    tensors, qlayers = {}, []
    for i, (l, sc) in enumerate(zip(layers, scales)):
        q = copy.deepcopy(l)
        a = q.self_attn
        # models/opt.py:97-104
        a.q_proj = RL.W8A8BFP32OFP32Linear.from_float(a.q_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.k_proj = RL.W8A8BFP32OFP32Linear.from_float(a.k_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.v_proj = RL.W8A8BFP32OFP32Linear.from_float(a.v_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.out_proj = RL.W8A8BFP32OFP32LinearWithQuantScale.from_float(a.out_proj, sc["out_input_scale"], act_quant=qc["out"])
        # models/opt.py:146-149
        q.fc1 = RL.W8A8BFP32OFP32Linear.from_float(q.fc1, sc["fc1_input_scale"], act_quant=qc["fc1"])
        q.fc2 = RL.W8A8BFP32OFP32LinearWithQuantScale.from_float(q.fc2, sc["fc2_input_scale"], act_quant=qc["fc2"])
        with torch.no_grad():   # Int8OPTLayerNorm.from_float (models/opt.py:20-29), applied iff per-tensor (:150-163)
            if qc["qkv"] == "per-tensor":
                q.self_attn_layer_norm.weight.div_(sc["attn_input_scale"])
                q.self_attn_layer_norm.bias.div_(sc["attn_input_scale"])
            if qc["fc1"] == "per-tensor":
                q.final_layer_norm.weight.div_(sc["fc1_input_scale"])
                q.final_layer_norm.bias.div_(sc["fc1_input_scale"])
        qlayers.append(q)
        p = f"model.decoder.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            put(tensors, p + f"self_attn.{name}.", getattr(a, name))
        put(tensors, p + "fc1.", q.fc1)
        put(tensors, p + "fc2.", q.fc2)
        put(tensors, p + "self_attn_layer_norm.", q.self_attn_layer_norm)
        put(tensors, p + "final_layer_norm.", q.final_layer_norm)
    # the unquantised rest of Int8OPTDecoder / Int8OPTForCausalLM (models/opt.py:180-204,262)
    tensors["model.decoder.embed_tokens.weight"] = torch.from_numpy(detrng.normal(912, 0, (V, H)).astype(np.float32) * 0.05)
    tensors["model.decoder.embed_positions.weight"] = torch.from_numpy(detrng.normal(912, 1, (64 + 2, H)).astype(np.float32) * 0.05)
    tensors["model.decoder.final_layer_norm.weight"] = torch.ones(H)
    tensors["model.decoder.final_layer_norm.bias"] = torch.zeros(H)
    tensors["lm_head.weight"] = tensors["model.decoder.embed_tokens.weight"].clone()
    ys, hq = [], torch.from_numpy(x.copy())
    with torch.no_grad():
        for q in qlayers:
            hq = q(hq, attention_mask=mask)
            hq = hq[0] if isinstance(hq, tuple) else hq
            ys.append(hq.numpy().copy())
    conf = {"model_type": "opt", "hidden_size": H, "ffn_dim": FFN, "num_attention_heads": HEADS, "num_hidden_layers": L, "vocab_size": V,
            "max_position_embeddings": 64, "word_embed_proj_dim": H, "do_layer_norm_before": True, "enable_bias": True,
            "activation_function": "relu", "layer_norm_elementwise_affine": True}
    finish(os.path.join(HERE, "ckpt_opt_w8a8"), tensors, conf, qc, x, ys,
           {"y_float": y_float, "scales": np.array([[s["attn_input_scale"], s["out_input_scale"], s["fc1_input_scale"], s["fc2_input_scale"]] for s in scales])})


# -------------------------------------------------------------------------------------------------------------- Mixtral
def moe_forward(h, gate, experts, top_k):
    """transformers 4.42.3 MixtralSparseMoeBlock.forward, which the reference borrows (models/mixtral.py:145)."""
    B, S, H = h.shape
    x = h.view(-1, H)
    logits = gate(x)
    w = torch.nn.functional.softmax(logits, dim=1, dtype=torch.float)
    w, sel = torch.topk(w, top_k, dim=-1)
    w = w / w.sum(dim=-1, keepdim=True)
    w = w.to(x.dtype)
    out = torch.zeros_like(x)
    mask = torch.nn.functional.one_hot(sel, num_classes=len(experts)).permute(2, 1, 0)
    for e, ex in enumerate(experts):
        idx, top_x = torch.where(mask[e])
        if top_x.numel() == 0:
            continue
        cur = x[None, top_x].reshape(-1, H)
        y = ex["w2"](torch.nn.functional.silu(ex["w1"](cur)) * ex["w3"](cur)) * w[top_x, idx, None]   # MixtralBlockSparseTop2MLP.forward
        out.index_add_(0, top_x, y.to(x.dtype))
    return out.view(B, S, H)


def make_mixtral():
    from transformers.models.mixtral.modeling_mixtral import MixtralConfig, MixtralAttention, MixtralRMSNorm, MixtralRotaryEmbedding
    H, INTER, HEADS, KVH, L, E, TOPK, V, B, S = 128, 256, 4, 2, 2, 4, 2, 40, 2, 20
    cfg = MixtralConfig(hidden_size=H, intermediate_size=INTER, num_attention_heads=HEADS, num_key_value_heads=KVH, num_hidden_layers=L, vocab_size=V,
                        num_local_experts=E, num_experts_per_tok=TOPK, max_position_embeddings=64, rope_theta=1e6, sliding_window=None, rms_norm_eps=1e-5)
    cfg._attn_implementation = "eager"
    qc = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}
    rot = MixtralRotaryEmbedding(cfg)
    x = detrng.normal(921, 0, (B, S, H)).astype(np.float32)
    x[..., 3] *= 8.0
    pos = torch.arange(S)[None].expand(B, -1)
    mask = causal_mask(S)

    def build(i):
        lay = {"attn": MixtralAttention(cfg, layer_idx=i).eval(), "ln1": MixtralRMSNorm(H, eps=1e-5), "ln2": MixtralRMSNorm(H, eps=1e-5),
               "gate": torch.nn.Linear(H, E, bias=False),
               "experts": [{k: torch.nn.Linear(*(s_), bias=False) for k, s_ in (("w1", (H, INTER)), ("w2", (INTER, H)), ("w3", (H, INTER)))} for _ in range(E)]}
        mods = torch.nn.ModuleList([lay["attn"], lay["ln1"], lay["ln2"], lay["gate"]] + [m for ex in lay["experts"] for m in ex.values()])
        det_init(mods, 920 + 7 * i)
        return lay

    def run(lay, h):
        cos, sin = rot(h, pos)
        a = lay["attn"](lay["ln1"](h), position_embeddings=(cos, sin), attention_mask=mask)[0]
        h = h + a
        return h + moe_forward(lay["ln2"](h), lay["gate"], lay["experts"], TOPK)

    layers = [build(i) for i in range(L)]
    scales, h = [], torch.from_numpy(x.copy())
    with torch.no_grad():
        for lay in layers:
            rec = {}
            hk = absmax_hooks([("attn_input_scale", lay["attn"].q_proj), ("out_input_scale", lay["attn"].o_proj), ("moe_input_scale", lay["gate"])]
                              + [(f"down_input_scale_{e}", ex["w2"]) for e, ex in enumerate(lay["experts"])], rec)
            h = run(lay, h)
            for k_ in hk:
                k_.remove()
            for e in range(E):
                rec.setdefault(f"down_input_scale_{e}", 1.0)   # an expert no token was routed to (quantize/calibration.py keeps its default)
            scales.append(rec)
    y_float = h.numpy().copy()
    tensors, qlayers = {}, []
    for i, (lay, sc) in enumerate(zip(layers, scales)):
        q = {"attn": copy.deepcopy(lay["attn"]), "ln1": copy.deepcopy(lay["ln1"]), "ln2": copy.deepcopy(lay["ln2"]), "gate": lay["gate"], "experts": []}
        a = q["attn"]
        # models/mixtral.py:88-91
        a.q_proj = RL.W8A8BFP32OFP32Linear.from_float(a.q_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.k_proj = RL.W8A8BFP32OFP32Linear.from_float(a.k_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.v_proj = RL.W8A8BFP32OFP32Linear.from_float(a.v_proj, sc["attn_input_scale"], act_quant=qc["qkv"])
        a.o_proj = RL.W8A8BFP32OFP32LinearWithQuantScale.from_float(a.o_proj, sc["out_input_scale"], act_quant=qc["out"])
        for e, ex in enumerate(lay["experts"]):   # models/mixtral.py:114-116
            src = copy.deepcopy(ex)
            q["experts"].append({"w1": RL.W8A8BFP32OFP32Linear.from_float(src["w1"], sc["moe_input_scale"], act_quant=qc["fc1"]),
                                 "w2": RL.W8A8BFP32OFP32LinearWithQuantScale.from_float(src["w2"], sc[f"down_input_scale_{e}"], act_quant=qc["fc2"]),
                                 "w3": RL.W8A8BFP32OFP32Linear.from_float(src["w3"], sc["moe_input_scale"], act_quant=qc["fc1"])})
        with torch.no_grad():   # Int8MixtralRMSNorm.from_float (models/mixtral.py:22-30), iff per-tensor
            if qc["qkv"] == "per-tensor":
                q["ln1"].weight.div_(sc["attn_input_scale"])
            if qc["fc1"] == "per-tensor":
                q["ln2"].weight.div_(sc["moe_input_scale"])
        qlayers.append(q)
        p = f"model.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            put(tensors, p + f"self_attn.{name}.", getattr(a, name))
        tensors[p + "block_sparse_moe.gate.weight"] = q["gate"].weight.detach().clone()
        for e, ex in enumerate(q["experts"]):
            for name in ("w1", "w2", "w3"):
                put(tensors, p + f"block_sparse_moe.experts.{e}.{name}.", ex[name])
        tensors[p + "input_layernorm.weight"] = q["ln1"].weight.detach().clone()
        tensors[p + "post_attention_layernorm.weight"] = q["ln2"].weight.detach().clone()
    tensors["model.embed_tokens.weight"] = torch.from_numpy(detrng.normal(922, 0, (V, H)).astype(np.float32) * 0.05)
    tensors["model.norm.weight"] = torch.ones(H)
    tensors["lm_head.weight"] = torch.from_numpy(detrng.normal(922, 1, (V, H)).astype(np.float32) * 0.05)
    ys, hq = [], torch.from_numpy(x.copy())
    # NOTE: with the folded ln2 the (float, unquantised) router sees ln2(h) / moe_input_scale -- exactly what the reference's layer feeds its
    # gate (models/mixtral.py borrows MixtralDecoderLayer.forward: one post_attention_layernorm output goes to gate AND experts)
    with torch.no_grad():
        for q in qlayers:
            hq = run(q, hq)
            ys.append(hq.numpy().copy())
    conf = {"model_type": "mixtral", "hidden_size": H, "intermediate_size": INTER, "num_attention_heads": HEADS, "num_key_value_heads": KVH,
            "num_hidden_layers": L, "vocab_size": V, "num_local_experts": E, "num_experts_per_tok": TOPK, "max_position_embeddings": 64,
            "rope_theta": 1e6, "rms_norm_eps": 1e-5, "hidden_act": "silu", "sliding_window": None}
    finish(os.path.join(HERE, "ckpt_mixtral_w8a8"), tensors, conf, qc, x, ys, {"y_float": y_float})


# ---------------------------------------------------------------------------------------------------------- LLaMA, fp8
def make_llama_fp8():
    from transformers.models.llama.modeling_llama import LlamaConfig, LlamaDecoderLayer, LlamaRotaryEmbedding
    H, INTER, HEADS, KVH, L, V, B, S = 128, 256, 4, 2, 2, 40, 2, 20
    cfg = LlamaConfig(hidden_size=H, intermediate_size=INTER, num_attention_heads=HEADS, num_key_value_heads=KVH, num_hidden_layers=L, vocab_size=V,
                      max_position_embeddings=64, rope_theta=500000.0, rms_norm_eps=1e-5, attention_bias=False, mlp_bias=False)
    cfg._attn_implementation = "eager"
    # examples/smoothquant_model.py:62-70: `type` / `activation_scheme` travel in quant_config.json; models/llama.py:76-91,187-200 build FP8LinearDynamic
    qc = {"qkv": "per-token", "out": "per-token", "fc1": "per-token", "fc2": "per-tensor", "type": "fp8_e4m3", "activation_scheme": "dynamic"}
    rot = LlamaRotaryEmbedding(cfg)
    layers = [LlamaDecoderLayer(cfg, layer_idx=i).eval() for i in range(L)]
    for i, l in enumerate(layers):
        det_init(l, 930 + i)
    x = detrng.normal(931, 0, (B, S, H)).astype(np.float32)
    pos = torch.arange(S)[None].expand(B, -1)
    mask = causal_mask(S)

    def run(l, h):
        cos, sin = rot(h, pos)
        o = l(h, attention_mask=mask, position_embeddings=(cos, sin))
        return o[0] if isinstance(o, tuple) else o

    h = torch.from_numpy(x.copy())
    with torch.no_grad():
        for l in layers:
            h = run(l, h)
    y_float = h.numpy().copy()
    tensors, qlayers = {}, []
    kinds = {"q_proj": "qkv", "k_proj": "qkv", "v_proj": "qkv", "o_proj": "out", "gate_proj": "fc1", "up_proj": "fc1", "down_proj": "fc2"}
    for i, l in enumerate(layers):
        q = copy.deepcopy(l)
        for parent, names in ((q.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (q.mlp, ("gate_proj", "up_proj", "down_proj"))):
            for name in names:
                src = getattr(parent, name)
                # what a LOADED reference model holds: QuantizedLlamaAttention / MLP.__init__ construct FP8LinearDynamic(in, out, act_quant=quant_config[..])
                # (models/llama.py:83-90,193-198) and from_pretrained fills weight / weight_scale, which from_float_to_fp8 computed with
                # FP8LinearDynamic.from_float -> per_tensor_quantize_fp8 (layers/nn/linear.py:429-452)
                conv = RL.FP8LinearDynamic.from_float(copy.deepcopy(src))
                m = RL.FP8LinearDynamic(src.in_features, src.out_features, act_quant=qc[kinds[name]])
                m.weight = conv.weight
                m.weight_scale = conv.weight_scale
                setattr(parent, name, m)
        qlayers.append(q)
        p = f"model.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            put(tensors, p + f"self_attn.{name}.", getattr(q.self_attn, name))
        for name in ("gate_proj", "up_proj", "down_proj"):
            put(tensors, p + f"mlp.{name}.", getattr(q.mlp, name))
        tensors[p + "input_layernorm.weight"] = q.input_layernorm.weight.detach().clone()
        tensors[p + "post_attention_layernorm.weight"] = q.post_attention_layernorm.weight.detach().clone()
    tensors["model.embed_tokens.weight"] = torch.from_numpy(detrng.normal(932, 0, (V, H)).astype(np.float32) * 0.05)
    tensors["model.norm.weight"] = torch.ones(H)
    tensors["lm_head.weight"] = torch.from_numpy(detrng.normal(932, 1, (V, H)).astype(np.float32) * 0.05)
    ys, hq = [], torch.from_numpy(x.copy())
    with torch.no_grad():
        for q in qlayers:
            hq = run(q, hq)
            ys.append(hq.numpy().copy())
    conf = {"model_type": "llama", "hidden_size": H, "intermediate_size": INTER, "num_attention_heads": HEADS, "num_key_value_heads": KVH,
            "num_hidden_layers": L, "vocab_size": V, "max_position_embeddings": 64, "rope_theta": 500000.0, "rms_norm_eps": 1e-5, "hidden_act": "silu"}
    finish(os.path.join(HERE, "ckpt_llama_fp8_e4m3"), tensors, conf, qc, x, ys, {"y_float": y_float})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    make_opt()
    make_mixtral()
    make_llama_fp8()
