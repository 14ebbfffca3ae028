"""CPU: the calibration collectors and the smooth -> quantise flow on a toy LLaMA-named model."""
import json

import numpy as np
import pytest
import torch

from autosmoothquant_amd import harness
from autosmoothquant_amd.quantize import (decoder_layer_scales, get_act_scales, get_io_absmax, get_static_decoder_layer_scales,
                                          parse_quant_config, smooth_ln_fcs)


class _Attn(torch.nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = layer.q_proj, layer.k_proj, layer.v_proj, layer.o_proj


class _Mlp(torch.nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = layer.gate_proj, layer.up_proj, layer.down_proj


class _Layer(torch.nn.Module):  # exposes a harness.LlamaLayer under HF's module names (model.layers.N.self_attn.q_proj ...)
    def __init__(self, hidden, inter, heads, seed):
        super().__init__()
        self.impl = [harness.init_llama_layer(harness.LlamaLayer(hidden, inter, heads), std=0.05, seed=seed)]
        self.self_attn, self.mlp = _Attn(self.impl[0]), _Mlp(self.impl[0])
        self.input_layernorm, self.post_attention_layernorm = self.impl[0].input_layernorm, self.impl[0].post_attention_layernorm

    def forward(self, h):
        return self.impl[0](h)


class _Model(torch.nn.Module):
    def __init__(self, n=2):
        super().__init__()
        self.model = torch.nn.Module()
        self.model.layers = torch.nn.ModuleList([_Layer(64, 96, 4, s) for s in range(n)])

    def forward(self, h):
        for layer in self.model.layers:
            h = layer(h)
        return h


def _batches():
    g = torch.Generator().manual_seed(5)
    return [torch.randn(2, 12, 64, generator=g) * (1 + i) for i in range(3)]


def test_collectors_track_running_absmax_under_reference_keys():
    m = _Model()
    act = get_act_scales(m, _batches())
    io = get_io_absmax(m, _batches())
    assert set(act) == set(io) and len(act) == 14 and "model.layers.1.mlp.down_proj" in act
    # recompute by hand: the inputs of layer 0's q_proj are input_layernorm(h) for each batch
    with torch.no_grad():
        rows = torch.cat([m.model.layers[0].input_layernorm(b).reshape(-1, 64) for b in _batches()])
    assert torch.equal(act["model.layers.0.self_attn.q_proj"], rows.abs().amax(0).float())
    assert io["model.layers.0.self_attn.q_proj"]["input"] == float(rows.abs().max())
    assert torch.equal(act["model.layers.0.self_attn.k_proj"], act["model.layers.0.self_attn.q_proj"])
    scales, io2 = get_static_decoder_layer_scales(m, _batches(), 2, "llama")
    assert io2 == io and len(scales) == 2
    assert sorted(scales[0]) == ["attn_input_scale", "down_input_scale", "gate_input_scale", "k_output_scale", "out_input_scale", "q_output_scale", "v_output_scale"]
    assert scales[1]["down_input_scale"] == io["model.layers.1.mlp.down_proj"]["input"] / 127
    assert not any(h for mod in m.modules() for h in mod._forward_hooks.values())   # hooks removed
    with pytest.raises(ValueError):
        decoder_layer_scales(io, 2, "gpt2")
    with pytest.raises(KeyError):
        decoder_layer_scales(io, 2, "baichuan")   # no W_pack in a llama-named model


def test_smoothing_preserves_the_float_function_and_flattens_outliers():
    m = _Model(1)
    layer = m.model.layers[0]
    with torch.no_grad():
        layer.input_layernorm.weight[5] *= 30.0          # an outlier channel
    x = _batches()[0]
    with torch.no_grad():
        ref = m(x)
    act = get_act_scales(m, [x])
    key = "model.layers.0.self_attn.q_proj"
    before = float(act[key].max() / act[key].median())
    smooth_ln_fcs(layer.input_layernorm, [layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj], act[key], "llama", 0.5)
    with torch.no_grad():
        got = m(x)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5)
    after_act = get_act_scales(m, [x])[key]
    assert float(after_act.max() / after_act.median()) < before / 3


def test_parse_quant_config(tmp_path):
    p = tmp_path / "quant_config.json"
    p.write_text(json.dumps({"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"}))
    assert parse_quant_config(str(p))["out"] == "per-token"
    p.write_text(json.dumps({"qkv": "per-channel"}))
    with pytest.raises(ValueError):
        parse_quant_config(str(p))
