"""GPU (-m gpu): N2.  (1) The checkpoint the REFERENCE wrote (tests/golden/ckpt_baichuan_w8a8, make_golden_ckpt.py) loaded onto the
device reproduces the reference model's own per-layer hidden states for the recorded input; (2) weight-side conversion
(`from_float`, quantize_weight_per_channel_absmax) executed ON THE DEVICE equals the golden G3 conversions for fp32 sources
(the reference's host and device branches coincide there, layers/functional/quantization.py:9-18)."""
import os

import numpy as np
import pytest
import torch

import goldenio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_baichuan_w8a8")


def test_reference_checkpoint_reproduces_reference_outputs():
    from autosmoothquant_amd import checkpoint
    m = checkpoint.load_reference_checkpoint(CKPT, device=DEV)
    for lay in m.layers:
        for mod in (lay.W_pack, lay.o_proj, lay.gate_proj, lay.up_proj, lay.down_proj):
            assert mod.weight.is_cuda and mod.weight.dtype == torch.int8
            assert all(mod._buffers[s].device.type == "cpu" for s in mod._host_scalars)
    z = np.load(os.path.join(CKPT, "io.npz"))
    for i, lay in enumerate(m.layers):
        # each layer on the REFERENCE's input to that layer (its recorded previous output): per-layer parity, no drift carried over
        h_in = torch.from_numpy(z["x"] if i == 0 else z["y_layers"][i - 1]).to(DEV)
        with torch.no_grad():
            h = lay(h_in)
        want = torch.from_numpy(z["y_layers"][i]).to(DEV)
        # the linears are bit-exact; the fp16 attention / SiLU glue runs on a different BLAS than the reference's CPU run, which
        # moves a few activations across int8 rounding boundaries (same bound as the G7 block test)
        err = float((h - want).abs().max() / want.abs().max())
        assert err < 2e-3, (i, err)
    # whole-stack forward through the loader's module (the boundary flips of layer 0 propagate through layer 1)
    with torch.no_grad():
        y = m(torch.from_numpy(z["x"]).to(DEV))
    assert float((y - torch.from_numpy(z["y_layers"][-1]).to(DEV)).abs().max() / np.abs(z["y_layers"][-1]).max()) < 6e-3
    # and it is a quantised model of the float one it came from
    yf = torch.from_numpy(z["y_float"]).to(DEV)
    assert float((y - yf).norm() / yf.norm()) < 5e-2


def test_on_device_from_float_equals_golden_g3():
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, W8A8BFP32OFP32QKVLinear
    from autosmoothquant_amd.layers.functional import quantization as Q
    z, index = goldenio.load_g3()
    n = 0
    for name, wdt, kind, aq, iscale, qkv in index:
        if wdt != "f32":
            continue   # 16-bit sources follow the reference's DEVICE branch (quotient rounded in the source dtype): not what G3 (host) holds
        lin = torch.nn.Linear(64, 48, bias=True)
        lin.weight.data = torch.from_numpy(z["W_f32"].copy())
        lin.bias.data = torch.from_numpy(z["b_f32"].copy())
        lin = lin.to(DEV)
        qkv_size = [int(v) for v in qkv.split(",")]
        if kind == "linear":
            m = W8A8BFP32OFP32Linear.from_float(lin, float(iscale), save_device=DEV, act_quant=aq)
        elif kind == "quantscale":
            m = W8A8BFP32OFP32LinearWithQuantScale.from_float(lin, float(iscale), save_device=DEV, act_quant=aq)
        else:
            m = W8A8BFP32OFP32QKVLinear.from_float(lin, float(iscale), qkv_size, save_device=DEV, act_quant=aq)
        assert m.weight.is_cuda and np.array_equal(m.weight.cpu().numpy(), z[name + "_wq"]), name
        assert np.array_equal(m.bias.detach().cpu().numpy(), z[name + "_bias"])
        if kind == "qkv":
            got = np.array([m._scalar("q_dequant_scale"), m._scalar("k_dequant_scale"), m._scalar("v_dequant_scale")], np.float32)
            assert np.array_equal(got, z[name + "_qkv_scales"]), name
        else:
            assert np.float32(m._scalar("dequant_scale")) == z[name + "_dequant_scale"], name
        assert np.array_equal(lin.weight.data.cpu().numpy(), z[name + "_src_after"]), name + " (in-place rounding of the fp32 source)"
        n += 1
    assert n == 6
    wq, sc = Q.quantize_weight_per_channel_absmax(torch.from_numpy(z["W_f32"].copy()).to(DEV))
    assert np.array_equal(wq.cpu().numpy(), z["pc_f32_wq"]) and np.array_equal(sc.cpu().numpy().reshape(-1), z["pc_f32_scales"])
    # a 16-bit source converted on the device: the reference's device branch (fp16 quotient, then round) -- int8 within +-1 of the host result
    w16 = torch.from_numpy(z["W_f16"].copy()).half().to(DEV)
    q16, s16 = Q.quantize_per_tensor_absmax(w16.clone())
    host = z["f16_linear_per-token_wq"]
    d = np.abs(q16.cpu().numpy().astype(np.int32) - host.astype(np.int32))
    assert d.max() <= 1 and q16.dtype == torch.int8


# ---- OPT / Mixtral / fp8-LLaMA directories (tests/golden/make_golden_ckpt_models.py): recorded layer I/O of the reference's linear modules inside HF's layers
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _replay(name, tol_layer, tol_stack):
    from autosmoothquant_amd import checkpoint
    d = os.path.join(GOLD, name)
    m = checkpoint.load_reference_checkpoint(d, device=DEV)
    z = np.load(os.path.join(d, "io.npz"))
    errs = []
    for i, lay in enumerate(m.layers):
        h_in = torch.from_numpy(z["x"] if i == 0 else z["y_layers"][i - 1]).to(DEV)   # the REFERENCE's input to that layer: per-layer parity
        with torch.no_grad():
            h = lay(h_in)
        want = torch.from_numpy(z["y_layers"][i]).to(DEV)
        # error of what the layer ADDS to the residual stream (the stream itself would hide it), relative to that contribution's size
        d_got, d_want = h - h_in, want - h_in
        errs.append(float((d_got - d_want).abs().max() / d_want.abs().max()))
        assert errs[-1] < tol_layer, (name, i, errs)
    with torch.no_grad():
        y = m(torch.from_numpy(z["x"]).to(DEV))
    want = torch.from_numpy(z["y_layers"][-1]).to(DEV)
    x0 = torch.from_numpy(z["x"]).to(DEV)
    assert float(((y - x0) - (want - x0)).abs().max() / (want - x0).abs().max()) < tol_stack, name
    return m, errs


def test_opt_checkpoint_replays_recorded_layer_io():
    """Biased q/k/v/out_proj, LayerNorm with folded weight AND bias, fc1 -> ReLU -> fc2 (models/opt.py:76-81,125-129,150-163).  The int8 linears are
    bit-exact; attention runs through torch SDPA here and through HF's eager CPU code in the recording, which moves a few activations across int8
    rounding boundaries."""
    m, errs = _replay("ckpt_opt_w8a8", 1e-2, 2e-2)
    lay = m.layers[0]
    # fc2 is per-token in this directory -> two launches; with a per-tensor fc2 the layer uses ONE int8-out launch for fc1 + ReLU + fc2's prologue
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32LinearWithQuantScale
    z = np.load(os.path.join(GOLD, "ckpt_opt_w8a8", "io.npz"))
    h = torch.from_numpy(z["x"]).to(DEV)
    fc2 = lay.fc2
    alt = W8A8BFP32OFP32LinearWithQuantScale(fc2.in_features, fc2.out_features, True, "per-tensor")
    alt.weight, alt.bias = fc2.weight, fc2.bias
    alt.dequant_scale, alt.quant_scale = torch.tensor(float(fc2.dequant_scale) * 0.05), torch.tensor(0.05)
    lay.fc2 = alt.to(DEV)
    with torch.no_grad():
        fused = lay(h)                 # forward_q path
        rec = {}
        plain = lay(h, rec)            # record != None takes the two-step path (fc1, relu, fc2's own quantiser)
    assert torch.equal(fused, plain)


def test_mixtral_checkpoint_replays_recorded_layer_io_and_grouped_equals_loop():
    m, errs = _replay("ckpt_mixtral_w8a8", 1e-2, 2e-2)
    lay = m.layers[1]
    z = np.load(os.path.join(GOLD, "ckpt_mixtral_w8a8", "io.npz"))
    x = lay.post_attention_layernorm(torch.from_numpy(z["y_layers"][0]).to(DEV))
    with torch.no_grad():
        grouped = lay.moe(x)
        stacks = {k: lay.__dict__.pop(k) for k in [k for k in lay.__dict__ if k.endswith("_stack")]}
        loop = lay.moe(x)              # the reference's per-expert module loop
        lay.__dict__.update(stacks)
    assert torch.equal(grouped, loop)
    _, _, counts = lay.route(x.reshape(-1, x.shape[-1]))
    assert int(counts.sum()) == 2 * x.shape[0] * x.shape[1]


def test_fp8_llama_checkpoint_replays_recorded_layer_io():
    """FP8LinearDynamic modules (per-token and per-tensor dynamic e4m3 activations), GQA, rope_theta = 5e5.  The reference dequantises both operands and
    calls F.linear in fp32 (layers/nn/linear.py:336-369); here the product runs on the fp8 matrix cores: same quantised operands, fp32 accumulation in a
    different order."""
    _replay("ckpt_llama_fp8_e4m3", 2e-2, 4e-2)
