"""CPU: the reference-format checkpoint loader (autosmoothquant_amd/checkpoint.py) on
  * tests/golden/ckpt_baichuan_w8a8 -- written by the REFERENCE itself (Int8BaichuanForCausalLM.from_float(...).save_pretrained +
    quant_config.json; tests/golden/make_golden_ckpt.py), and
  * a LLaMA-named directory synthesised here from this package's own from_float modules (the reference's models/llama.py cannot be
    imported under the installed transformers; the key names follow models/llama.py:99-106,206-211 and the buffer contract of
    layers/nn/linear.py:49-66,253-256).
No forward pass here (the modules have no CPU compute); the GPU test replays the reference's recorded outputs."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from autosmoothquant_amd import checkpoint, harness
from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, W8A8BFP32OFP32QKVLinear

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_baichuan_w8a8")


def test_loads_the_reference_written_checkpoint():
    m = checkpoint.load_reference_checkpoint(CKPT, device="cpu")
    assert m.arch == "baichuan" and len(m.layers) == 2
    assert m.quant_config == {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-tensor"}
    raw = checkpoint.read_tensors(CKPT)
    for i, lay in enumerate(m.layers):
        assert isinstance(lay.W_pack, W8A8BFP32OFP32QKVLinear) and lay.W_pack.act_quant == "per-tensor" and lay.W_pack.qkv_size == [256, 256, 256]
        assert isinstance(lay.o_proj, W8A8BFP32OFP32LinearWithQuantScale) and lay.o_proj.act_quant == "per-token" and "quant_scale" not in lay.o_proj._buffers
        assert isinstance(lay.down_proj, W8A8BFP32OFP32LinearWithQuantScale) and lay.down_proj.act_quant == "per-tensor"
        assert isinstance(lay.gate_proj, W8A8BFP32OFP32Linear) and isinstance(lay.up_proj, W8A8BFP32OFP32Linear)
        p = f"model.layers.{i}."
        for mod, name in ((lay.W_pack, "self_attn.W_pack"), (lay.o_proj, "self_attn.o_proj"), (lay.gate_proj, "mlp.gate_proj"),
                          (lay.up_proj, "mlp.up_proj"), (lay.down_proj, "mlp.down_proj")):
            sd = mod.state_dict()
            want = {k[len(p + name) + 1:]: v for k, v in raw.items() if k.startswith(p + name + ".")}
            assert set(sd) == set(want)
            for k, v in want.items():
                assert sd[k].dtype == v.dtype and torch.equal(sd[k], v), (name, k)
            for s in mod._host_scalars:   # scalar scales: fp32, on the host
                assert mod._buffers[s].device.type == "cpu" and mod._buffers[s].dtype == torch.float32 and mod._buffers[s].dim() == 0
        assert torch.equal(lay.input_layernorm.weight, raw[p + "input_layernorm.weight"])
    assert tuple(m.embed_tokens_weight.shape) == (64, 256) and tuple(m.lm_head_weight.shape) == (64, 256) and tuple(m.norm_weight.shape) == (256,)


def _write(dirpath, tensors, cfg, qc):
    from safetensors.torch import save_file
    os.makedirs(dirpath, exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(dirpath, "model.safetensors"))
    json.dump(cfg, open(os.path.join(dirpath, "config.json"), "w"))
    json.dump(qc, open(os.path.join(dirpath, "quant_config.json"), "w"))


def test_contract_violations_raise(tmp_path):
    raw = checkpoint.read_tensors(CKPT)
    cfg = json.load(open(os.path.join(CKPT, "config.json")))
    qc = json.load(open(os.path.join(CKPT, "quant_config.json")))
    bad = dict(raw)
    bad["model.layers.0.mlp.down_proj.weight"] = bad["model.layers.0.mlp.down_proj.weight"].to(torch.int16)
    _write(tmp_path / "dtype", bad, cfg, qc)
    with pytest.raises(TypeError, match="down_proj.weight"):
        checkpoint.load_reference_checkpoint(str(tmp_path / "dtype"), device="cpu")
    bad = dict(raw)
    del bad["model.layers.1.mlp.down_proj.quant_scale"]        # fc2 is per-tensor: the buffer must be there
    _write(tmp_path / "missing", bad, cfg, qc)
    with pytest.raises(RuntimeError, match="quant_scale"):
        checkpoint.load_reference_checkpoint(str(tmp_path / "missing"), device="cpu")
    bad = dict(raw)
    bad["model.layers.0.self_attn.o_proj.quant_scale"] = torch.tensor(0.5)   # out is per-token: no such buffer (linear.py:253-256)
    _write(tmp_path / "unexpected", bad, cfg, qc)
    with pytest.raises(RuntimeError, match="quant_scale"):
        checkpoint.load_reference_checkpoint(str(tmp_path / "unexpected"), device="cpu")
    _write(tmp_path / "qc", raw, cfg, {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor"})
    with pytest.raises(KeyError, match="fc2"):
        checkpoint.load_reference_checkpoint(str(tmp_path / "qc"), device="cpu")
    bad = dict(raw)
    bad["model.layers.0.mlp.fc1.weight"] = torch.zeros(4, 4, dtype=torch.int8)
    _write(tmp_path / "opt", bad, cfg, qc)
    with pytest.raises(NotImplementedError):
        checkpoint.load_reference_checkpoint(str(tmp_path / "opt"), device="cpu")


def test_llama_named_directory_round_trip(tmp_path):
    torch.manual_seed(0)
    H, I, heads = 128, 256, 4
    tensors, ref_layers = {}, []
    qc = {"qkv": "per-token", "out": "per-tensor", "fc1": "per-tensor", "fc2": "per-token"}
    for i in range(2):
        fl = harness.init_llama_layer(harness.LlamaLayer(H, I, heads), std=0.05, seed=i)
        scales = {"attn_in": 0.11, "o_in": 0.07, "mlp_in": 0.13, "down_in": 0.05}
        q = harness.to_w8a8(fl, scales, qc)
        ref_layers.append(q)
        names = {"q_proj": "self_attn.q_proj", "k_proj": "self_attn.k_proj", "v_proj": "self_attn.v_proj", "o_proj": "self_attn.o_proj",
                 "gate_proj": "mlp.gate_proj", "up_proj": "mlp.up_proj", "down_proj": "mlp.down_proj",
                 "input_layernorm": "input_layernorm", "post_attention_layernorm": "post_attention_layernorm"}
        for k, v in q.state_dict().items():
            mod, rest = k.split(".", 1)
            tensors[f"model.layers.{i}.{names[mod]}.{rest}"] = v.detach().clone()
    tensors["model.embed_tokens.weight"] = torch.randn(32, H)
    tensors["model.norm.weight"] = torch.ones(H)
    tensors["lm_head.weight"] = torch.randn(32, H)
    cfg = {"hidden_size": H, "intermediate_size": I, "num_attention_heads": heads, "num_hidden_layers": 2, "rms_norm_eps": 1e-5,
           "architectures": ["LlamaForCausalLM"]}
    # sharded layout with an index file, as save_pretrained writes large models
    from safetensors.torch import save_file
    d = tmp_path / "llama"
    os.makedirs(d)
    ks = sorted(tensors)
    shards = {"model-00001-of-00002.safetensors": ks[:len(ks) // 2], "model-00002-of-00002.safetensors": ks[len(ks) // 2:]}
    for fn, kk in shards.items():
        save_file({k: tensors[k].contiguous() for k in kk}, str(d / fn))
    json.dump({"metadata": {}, "weight_map": {k: fn for fn, kk in shards.items() for k in kk}}, open(d / "model.safetensors.index.json", "w"))
    json.dump(cfg, open(d / "config.json", "w"))
    json.dump(qc, open(d / "quant_config.json", "w"))
    m = checkpoint.load_reference_checkpoint(str(d), device="cpu", dtype=torch.float16)
    assert m.arch == "llama" and len(m.layers) == 2
    for got, ref in zip(m.layers, ref_layers):
        assert got.q_proj.act_quant == "per-token" and got.o_proj.act_quant == "per-tensor" and "quant_scale" in got.o_proj._buffers
        for n in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"):
            a, b = getattr(got, n).state_dict(), getattr(ref, n).state_dict()
            assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a), n
        assert got.input_layernorm.weight.dtype == torch.float16   # `dtype` applies to the floating tensors only
        assert got.q_proj.weight.dtype == torch.int8 and got.o_proj._buffers["quant_scale"].dtype == torch.float32
