"""GPU (-m gpu): the LLaMA-architecture layer harness composes the W8A8 linears the way the reference's
QuantizedLlamaDecoderLayer does (models/llama.py:289-339) and stays within quantisation error of the
float layer (SURVEY 8c observed ~1-2e-2 relative error for the reference's own block at hidden=256)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [None, {"qkv": "per-token", "fc1": "per-token"}, {"out": "per-tensor", "fc2": "per-tensor"}])
def test_w8a8_layer_close_to_float_layer(cfg):
    from autosmoothquant_amd import harness
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layer = harness.init_llama_layer(harness.LlamaLayer(hidden=256, inter=512, heads=4), std=0.05).to(dev)
    h = torch.randn(2, 48, 256, device=dev)
    scales = harness.calibrate(layer, h)
    q = harness.to_w8a8(layer, scales, cfg)
    assert isinstance(q.q_proj, W8A8BFP32OFP32Linear) and isinstance(q.gate_proj, W8A8BFP32OFP32Linear)
    assert isinstance(q.o_proj, W8A8BFP32OFP32LinearWithQuantScale) and isinstance(q.down_proj, W8A8BFP32OFP32LinearWithQuantScale)
    eff = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token", **(cfg or {})}
    assert q.q_proj.act_quant == eff["qkv"] and q.o_proj.act_quant == eff["out"] and q.up_proj.act_quant == eff["fc1"] and q.down_proj.act_quant == eff["fc2"]
    # norm weight folded iff the following linears are per-tensor (models/llama.py:326-339)
    folded = not torch.equal(q.input_layernorm.weight, layer.input_layernorm.weight)
    assert folded == (eff["qkv"] == "per-tensor")
    with torch.no_grad():
        ref, got = layer(h), q(h)
    # compare the layer's update (output - residual input): the residual itself is exact
    err = ((got - h) - (ref - h)).norm() / (ref - h).norm()
    assert torch.isfinite(got).all() and err < 5e-2, float(err)
    # fp16 activations run too
    with torch.no_grad():
        got16 = harness.to_w8a8(layer.half(), scales, cfg)(h.half())
    assert got16.dtype == torch.float16 and torch.isfinite(got16).all()
