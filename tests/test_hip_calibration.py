"""GPU (-m gpu): N3 + N2 end to end on the device -- the reference's generate-scale -> smooth -> quantise -> run flow
(examples/smoothquant_model.py:41-99: get_act_scales, smooth_lm, get_static_decoder_layer_scales, from_float_to_int8) with the collectors,
the smoothing and the conversion all executing on HIP tensors, then the quantised layers running on the HIP kernels."""
import pytest
import torch

from test_calibration_cpu import _Model, _batches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_calibrate_smooth_quantise_run_on_device():
    from autosmoothquant_amd import harness
    from autosmoothquant_amd.quantize import get_act_scales, get_static_decoder_layer_scales, smooth_ln_fcs
    cpu_model = _Model()
    dev_model = _Model().to(DEV)
    dev_model.load_state_dict(cpu_model.state_dict())
    batches_cpu = _batches()
    batches = [b.to(DEV) for b in batches_cpu]
    # collectors on the device == collectors on the host (same hooks, same keys; fp32 matmul order differs in the last bits)
    act_dev, act_cpu = get_act_scales(dev_model, batches), get_act_scales(cpu_model, batches_cpu)
    assert set(act_dev) == set(act_cpu)
    for k in act_cpu:
        assert act_dev[k].device.type == "cpu"          # the reference keeps the statistics on the host (calibration.py:56)
        assert torch.allclose(act_dev[k], act_cpu[k], rtol=1e-4, atol=1e-5), k
    # smoothing on device tensors preserves the float function
    x = batches[0]
    with torch.no_grad():
        ref = dev_model(x)
    for i, layer in enumerate(dev_model.model.layers):
        p = f"model.layers.{i}."
        smooth_ln_fcs(layer.input_layernorm, [layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj], act_dev[p + "self_attn.q_proj"].to(DEV), "llama", 0.5)
        smooth_ln_fcs(layer.post_attention_layernorm, [layer.mlp.gate_proj, layer.mlp.up_proj], act_dev[p + "mlp.gate_proj"].to(DEV), "llama", 0.5)
    with torch.no_grad():
        smoothed = dev_model(x)
    assert torch.allclose(smoothed, ref, rtol=2e-4, atol=2e-5)
    # static scales of the smoothed model, then the reference's layer composition on the HIP kernels
    scales, _ = get_static_decoder_layer_scales(dev_model, batches, 2, "llama")
    h = x
    with torch.no_grad():
        for i, layer in enumerate(dev_model.model.layers):
            s = scales[i]
            q = harness.to_w8a8(layer.impl[0], {"attn_in": s["attn_input_scale"], "o_in": s["out_input_scale"], "mlp_in": s["gate_input_scale"],
                                                "down_in": s["down_input_scale"]})
            assert q.q_proj.weight.is_cuda and q.q_proj.weight.dtype == torch.int8
            h = q(h)
    err = float((h - ref).norm() / ref.norm())
    assert torch.isfinite(h).all() and err < 6e-2, err


@pytest.mark.parametrize("tag", ["llama", "opt"])
def test_device_collectors_reproduce_the_reference_fixture(tag):
    """The collectors running on HIP tensors against what the REFERENCE's quantize/calibration.py returned on the same toy model and token ids
    (tests/golden/g9_calib.npz): the device's fp32 GEMMs sum in another order than the recording's CPU run, hence a relative tolerance."""
    from test_calibration_cpu import check_against_reference_fixture
    check_against_reference_fixture(tag, device=DEV, rtol=2e-4)
