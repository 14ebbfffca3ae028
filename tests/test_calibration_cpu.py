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


# ---------------------------------------------------------------------------------------------------------------------------------------
# N3 pinned to the REFERENCE: tests/golden/g9_calib.npz holds what quantize/calibration.py's own get_act_scales and
# get_static_decoder_layer_scales returned on the toy HF models of tests/calib_toy.py (tests/golden/make_golden_calib.py); the collectors
# of this package, fed the same token ids in the same order, must return the same numbers.
# ---------------------------------------------------------------------------------------------------------------------------------------
import os

G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_calib.npz")


def check_against_reference_fixture(tag, device="cpu", rtol=0.0):
    import calib_toy
    z = np.load(G9)
    model = (calib_toy.build_llama() if tag == "llama" else calib_toy.build_opt()).to(device)
    mtype = "llama" if tag == "llama" else "transformers"
    names = [str(n) for n in z[f"{tag}_names"]]
    ids_act = [torch.from_numpy(z[f"{tag}_ids_act_{j}"]).to(device) for j in range(6)]
    ids_static = [torch.from_numpy(z[f"{tag}_ids_static_{j}"]).to(device) for j in range(5)]
    act = get_act_scales(model, ids_act)
    assert sorted(act) == names                                  # every nn.Linear, the reference's module names (incl. lm_head)
    for n in names:
        want = z[f"{tag}_act::{n}"]
        assert act[n].dtype == torch.float32 and act[n].device.type == "cpu" and act[n].shape == want.shape
        np.testing.assert_allclose(act[n].numpy(), want, rtol=rtol, atol=0.0, err_msg=n)
    scales, act_dict = get_static_decoder_layer_scales(model, ids_static, 2, mtype)
    io = np.array([[act_dict[n]["input"], act_dict[n]["output"]] for n in names])
    np.testing.assert_allclose(io, z[f"{tag}_io"], rtol=rtol, atol=0.0)
    keys = [str(k) for k in z[f"{tag}_scale_keys"]]
    assert sorted(scales[0]) == keys and len(scales) == 2
    np.testing.assert_allclose(np.array([[s[k] for k in keys] for s in scales]), z[f"{tag}_scales"], rtol=rtol, atol=0.0)


@pytest.mark.parametrize("tag", ["llama", "opt"])
def test_collectors_equal_the_reference_collectors(tag):
    check_against_reference_fixture(tag)      # same torch CPU kernels on the same inputs: exact


def test_reference_dataset_fixture_matches_the_toy_corpus():
    import calib_toy
    lines = [json.loads(l)["text"] for l in open(os.path.join(os.path.dirname(G9), "calib_dataset.jsonl"))]
    assert lines == calib_toy.DATASET
    # the reference shuffles (seed 42) and truncates: the recorded ids are a permutation of the tokenised corpus prefixes
    z = np.load(G9)
    tok = calib_toy.ToyTokenizer()
    want = sorted(tuple(tok(t, max_length=24, truncation=True).input_ids[0].tolist()) for t in calib_toy.DATASET)
    got = [tuple(z[f"llama_ids_act_{j}"][0].tolist()) for j in range(6)]
    assert all(g in want for g in got) and len(set(got)) == 6
