"""CPU: harness.all_per_tensor_view -- the "linears all per-tensor" composition (BASELINE configs[2]) as a view over the weights of the default
(out / fc2 per-token) layer: its twins equal what to_w8a8(quant_config=all per-tensor) builds from the float layer, and share the int8 buffers."""
import torch

from autosmoothquant_amd import harness


def test_all_per_tensor_view_equals_a_fresh_conversion_and_shares_the_weights():
    torch.manual_seed(0)
    fl = harness.LlamaLayer(64, 96, 4)
    harness.init_llama_layer(fl, 0.05)
    sc = harness.calibrate(fl, torch.randn(2, 8, 64))
    q = harness.to_w8a8(fl, sc, both=True)
    v = harness.all_per_tensor_view(q)
    ref = harness.to_w8a8(fl, sc, quant_config={"out": "per-tensor", "fc2": "per-tensor"})
    for n in ("o_proj", "down_proj"):
        a, b = getattr(v, n), getattr(ref, n)
        assert a.act_quant == "per-tensor" and getattr(q, n).act_quant == "per-token"
        assert torch.equal(a.weight, b.weight) and a.weight.data_ptr() == getattr(q, n).weight.data_ptr()
        assert float(a.dequant_scale) == float(b.dequant_scale) and float(a.quant_scale) == float(b.quant_scale)
        assert a.input_signature() == b.input_signature()
    assert v.q_proj is q.q_proj and v.gate_proj is q.gate_proj and v.qkv_proj is q.qkv_proj
    assert q.o_proj is not v.o_proj and q._modules["o_proj"].act_quant == "per-token"   # the original layer is untouched
    assert v.quant_config == {"qkv": "per-tensor", "out": "per-tensor", "fc1": "per-tensor", "fc2": "per-tensor"}
