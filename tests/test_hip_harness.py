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


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("per_token", [False, True])
def test_fused_norm_quant_matches_two_step_path(dt, per_token):
    """asq_norm_quantize == (torch RMSNorm with folded weight, then asq_quantize_act) except where the fp32
    reduction order moves a value across a rounding boundary: <= 1e-3 of the entries, each by +-1."""
    from autosmoothquant_amd import harness, ops
    from autosmoothquant_amd.layers.nn.fused import RMSNormQ, LayerNormQ
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    M, K = 300, 4096
    x = (torch.randn(M, K, device=dev) * 3).to(dt)
    norm = harness.RMSNorm(K, 1e-5).to(dev).to(dt)
    with torch.no_grad():
        norm.weight.copy_((1 + 0.1 * torch.randn(K, device=dev)).to(dt))
    scale = 0.037
    ref_norm = norm if per_token else norm.folded(scale)
    with torch.no_grad():
        y = ref_norm(x)
    rq, rs = ops.quantize_act(y.contiguous(), "per-token" if per_token else "per-tensor-round")
    qa = RMSNormQ.from_float(norm, scale, per_token)(x.view(3, 100, K))
    assert qa.xq.dtype == torch.int8 and qa.shape == (3, 100, K) and qa.out_dtype == dt
    diff = (qa.xq.int() - rq.int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 1e-3
    if per_token:
        assert torch.allclose(qa.s_row, rs, rtol=2e-3)
    # LayerNorm flavour (OPT)
    ln = torch.nn.LayerNorm(K).to(dev).to(dt)
    with torch.no_grad():
        ln.weight.copy_((1 + 0.1 * torch.randn(K, device=dev)).to(dt))
        ln.bias.copy_((0.1 * torch.randn(K, device=dev)).to(dt))
        yl = torch.nn.functional.layer_norm(x, (K,), ln.weight / scale, ln.bias / scale, ln.eps)
    rq, _ = ops.quantize_act(yl.contiguous(), "per-tensor-round")
    ql = LayerNormQ.from_float(ln, scale)(x)
    diff = (ql.xq.int() - rq.int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 2e-3


def test_fused_norm_layer_close_to_unfused_layer():
    from autosmoothquant_amd import harness
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layer = harness.init_llama_layer(harness.LlamaLayer(hidden=256, inter=512, heads=4), std=0.05).to(dev).half()
    h = torch.randn(2, 48, 256, device=dev).half()
    scales = harness.calibrate(layer.float(), h.float())
    layer = layer.half()
    for cfg in (None, {"qkv": "per-token", "fc1": "per-token"}):
        with torch.no_grad():
            a = harness.to_w8a8(layer, scales, cfg)(h)
            b = harness.to_w8a8(layer, scales, cfg, fuse_norm=True)(h)
        err = ((a - h).float() - (b - h).float()).norm() / (a - h).float().norm()
        assert torch.isfinite(b).all() and err < 2e-2, float(err)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("per_token", [True, False])
def test_silu_mul_quant_matches_two_step_path(dt, per_token):
    from autosmoothquant_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    M, K = 200, 11008 if dt != torch.float32 else 4096
    g = (torch.randn(M, K, device=dev) * 2).to(dt)
    u = (torch.randn(M, K, device=dev) * 2).to(dt)
    a = (torch.nn.functional.silu(g) * u).contiguous()
    rq, rs = ops.quantize_act(a, "per-token" if per_token else "per-tensor-div", 0.05)
    xq, s_row = ops.silu_mul_quantize(g, u, per_token, 0.05)
    diff = (xq.int() - rq.int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 2e-3
    if per_token:
        assert torch.allclose(s_row, rs, rtol=4e-3)
