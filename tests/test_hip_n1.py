"""GPU (-m gpu): the N1 fusions through the C-ABI against oracle/n1.py, BIT FOR BIT (the oracle repeats the kernels' fp32
operation order, see its header), and against the reference's own outputs in tests/golden/g8_n1.npz up to the documented
rounding-boundary bound."""
import numpy as np
import pytest
import torch

import detrng
import goldenio
from oracle import n1, w8a8 as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _t(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(TDT[dt]).contiguous()


def _n(t):
    return t.float().cpu().numpy()


def _inputs(seed, M, K, dt, scale=3.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32) * scale
    x[:, 5 % K] *= 12.0
    if M > 2:
        x[2] = 0.0                      # an all-zero row (per-token: scale 0, NaN quotients -> 0)
    w = (rng.standard_normal(K).astype(np.float32) * 0.1 + 1.0) / 0.04
    b = rng.standard_normal(K).astype(np.float32) * 3.0
    return O.round_to(x, dt), O.round_to(w, dt), O.round_to(b, dt)


SHAPES = [(1, 64), (7, 320), (33, 4096), (5, 8192), (130, 1024)]


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("per_token", [False, True])
@pytest.mark.parametrize("layernorm", [False, True])
def test_norm_quantize_equals_oracle(dt, per_token, layernorm):
    from autosmoothquant_amd import ops
    for si, (M, K) in enumerate(SHAPES):
        if dt == "f32" and K > 4096:
            continue
        x, w, b = _inputs(100 + si, M, K, dt)
        bias = b if layernorm else None
        xq, s = ops.norm_quantize(_t(x, dt), _t(w, dt), None if bias is None else _t(bias, dt), 1e-5, per_token)
        rq, rs = n1.norm_quant_kernel_order(x, dt, w, bias, 1e-5, per_token)
        assert np.array_equal(xq.cpu().numpy(), rq), (M, K)
        if per_token:
            assert np.array_equal(s.cpu().numpy(), rs.reshape(-1))


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("per_token", [False, True])
@pytest.mark.parametrize("layernorm", [False, True])
def test_add_norm_quantize_equals_oracle(dt, per_token, layernorm):
    from autosmoothquant_amd import ops
    for si, (M, K) in enumerate(SHAPES):
        if dt == "f32" and K > 4096:
            continue
        x, w, b = _inputs(200 + si, M, K, dt)
        res, _, _ = _inputs(300 + si, M, K, dt, scale=5.0)
        bias = b if layernorm else None
        h, xq, s = ops.add_norm_quantize(_t(x, dt), _t(res, dt), _t(w, dt), None if bias is None else _t(bias, dt), 1e-5, per_token)
        rh, rq, rs = n1.add_norm_quant_kernel_order(x, res, dt, w, bias, 1e-5, per_token)
        assert np.array_equal(_n(h), rh), (M, K)
        assert np.array_equal(xq.cpu().numpy(), rq), (M, K)
        if per_token:
            assert np.array_equal(s.cpu().numpy(), rs.reshape(-1))
    # in place on the residual stream
    x, w, _ = _inputs(7, 9, 512, dt)
    res, _, _ = _inputs(8, 9, 512, dt)
    rt = _t(res, dt)
    h, xq, _ = ops.add_norm_quantize(_t(x, dt), rt, _t(w, dt), None, 1e-5, per_token, out=rt)
    rh, rq, _ = n1.add_norm_quant_kernel_order(x, res, dt, w, None, 1e-5, per_token)
    assert h.data_ptr() == rt.data_ptr() and np.array_equal(_n(rt), rh) and np.array_equal(xq.cpu().numpy(), rq)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("per_token", [True, False])
def test_silu_mul_quantize_equals_oracle(dt, per_token):
    from autosmoothquant_amd import ops
    for si, (M, K) in enumerate([(1, 64), (9, 704), (33, 11008), (4, 14336), (70, 2048)]):
        if dt == "f32" and K > 8192:
            continue
        rng = np.random.default_rng(400 + si)
        g = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 4, dt)
        u = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 4, dt)
        g[0, :6] = O.round_to(np.array([0.0, -0.0, 30.0, -30.0, 88.0, -100.0], np.float32), dt)   # saturating branches of exp_det
        xq, s = ops.silu_mul_quantize(_t(g, dt), _t(u, dt), per_token, 0.21, fast=False)
        rq, rs = n1.silu_mul_quant_kernel_order(g, u, dt, per_token, 0.21)
        assert np.array_equal(xq.cpu().numpy(), rq), (M, K)
        if per_token:
            assert np.array_equal(s.cpu().numpy(), rs.reshape(-1))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("per_token", [True, False])
def test_silu_mul_quantize_every_16bit_gate(dt, per_token):
    """Exhaustive in the gate: all 65 536 bit patterns of the 16-bit type (zeros, subnormals, every exponent, +-inf, NaNs),
    each against several `up` values -- the whole domain of exp_det and of the division, not a sample of it."""
    from autosmoothquant_amd import ops
    bits = np.arange(65536, dtype=np.uint32)
    if dt == "f16":
        g = bits.astype(np.uint16).view(np.float16).astype(np.float32)
    else:
        g = (bits << 16).astype(np.uint32).view(np.float32)
    g = g.reshape(64, 1024)
    rng = np.random.default_rng(5)
    for k, scale in enumerate([1.0, 1e-3, 300.0]):
        u = O.round_to(rng.standard_normal(g.shape).astype(np.float32) * scale, dt)
        with np.errstate(all="ignore"):
            rq, rs = n1.silu_mul_quant_kernel_order(g, u, dt, per_token, 0.21)
        xq, s = ops.silu_mul_quantize(_t(g, dt), _t(u, dt), per_token, 0.21, fast=False)
        assert np.array_equal(xq.cpu().numpy(), rq), (dt, k)
        if per_token:
            assert np.array_equal(s.cpu().numpy(), rs.reshape(-1), equal_nan=True)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("per_token", [True, False])
def test_silu_mul_quantize_fast_variant_is_within_one_of_the_oracle(dt, per_token):
    """The opt-in ASQ_SILU_FAST kernel (hardware v_exp_f32 / v_rcp_f32 instead of the bit-reproducible sequence): never more than +-1 int8 away from the
    oracle-exact result (+-2 for a handful of bf16 elements), and only on a small fraction of the elements (where silu(g) * up sits within an ulp of a rounding boundary); the per-token
    scale within one ulp of the activation dtype."""
    from autosmoothquant_amd import ops
    for si, (M, K) in enumerate([(9, 704), (33, 11008), (70, 2048)]):
        if dt == "f32" and K > 8192:
            continue
        rng = np.random.default_rng(600 + si)
        g = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 4, dt)
        u = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 4, dt)
        g[0, :6] = O.round_to(np.array([0.0, -0.0, 30.0, -30.0, 88.0, -100.0], np.float32), dt)
        xq, s = ops.silu_mul_quantize(_t(g, dt), _t(u, dt), per_token, 0.21, fast=True)
        rq, rs = n1.silu_mul_quant_kernel_order(g, u, dt, per_token, 0.21)
        d = np.abs(xq.cpu().numpy().astype(np.int32) - rq.astype(np.int32))
        # f16 / f32: +-1.  bf16: silu and the product are each rounded to 8 significant bits, so a 1-ulp silu can move the bf16 quotient by two of its
        # 0.5-wide steps near |x / s| = 64..127: +-2 there, +-1 everywhere else
        assert d.max() <= (2 if dt == "bf16" else 1), (dt, M, K, int(d.max()))
        assert (d > 1).mean() <= 1e-4 and (d != 0).mean() <= (2e-2 if per_token else 5e-3), (dt, M, K, float((d != 0).mean()))
        if per_token:
            np.testing.assert_allclose(s.cpu().numpy(), rs.reshape(-1), rtol={"f16": 1e-3, "bf16": 8e-3, "f32": 1e-6}[dt])


@pytest.mark.parametrize("c", [c for c in goldenio.load_g8() if c["kind"] in ("lnq", "rmsq")], ids=lambda c: c["id"])
def test_reference_fixtures_through_hip(c):
    """The reference's own LayerNormQ / folded-RMSNorm+round inputs through the HIP kernel: equal to the oracle exactly and to the
    reference's int8 except on rounding boundaries (+-1)."""
    from autosmoothquant_amd import ops
    dt = "f32" if c["kind"] == "lnq" else c["dt"]   # LayerNormQ computes in fp32 whatever x was (fused.py:11)
    bias = c.get("b")
    xq, _ = ops.norm_quantize(_t(c["x"], dt), _t(c["w"], dt), None if bias is None else _t(bias, dt), c["eps"], False)
    rq, _ = n1.norm_quant_kernel_order(c["x"], dt, c["w"], bias, c["eps"], False)
    got = xq.cpu().numpy()
    assert np.array_equal(got, rq)
    d = np.abs(got.astype(np.int32) - c["out"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() <= 5e-3


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("shape", [(32, 512, 384), (5, 100, 52), (512, 512, 1024), (3072, 256, 3072), (300, 1024, 1000)], ids=str)
def test_linear_q8_equals_oracle(dt, shape):
    """asq_linear_w8a8_q8 (int8-out epilogue feeding the next linear) on every dispatcher path (weight-streaming, generic, p8h,
    p8, ragged N) == linear -> relu -> consumer's per-tensor prologue, exactly."""
    from autosmoothquant_amd import ops
    M, K, N = shape
    rng = np.random.default_rng(M * 7 + N)
    xq = rng.integers(-128, 128, (M, K), dtype=np.int8)
    wq = rng.integers(-128, 128, (N, K), dtype=np.int8)
    bias = (rng.standard_normal(N) * 20).astype(np.float32)
    s_row = np.abs(rng.standard_normal(M)).astype(np.float32) * 0.02 + 0.01
    s_col = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.02 + 0.01
    txq, tw = torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV)
    cases = [dict(s_scalar=0.004, act="relu", qmode="per-tensor-div", quant_scale=0.37, bias=bias),
             dict(s_scalar=0.004, act=None, qmode="per-tensor-round", quant_scale=1.0),
             dict(s_scalar=1.0, s_row=s_row, s_col=s_col, bias=bias, act="relu", qmode="per-tensor-div", quant_scale=0.011)]
    for c in cases:
        dev = {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
        got = ops.linear_w8a8_q8(txq, tw, TDT[dt], dev["s_scalar"], dev.get("s_row"), dev.get("s_col"), dev.get("bias"), dev["act"], dev["qmode"], dev["quant_scale"])
        ref = n1.linear_q8_forward(xq, wq, dt, c["s_scalar"], c.get("s_row"), c.get("s_col"), c.get("bias"), c["act"], c["qmode"], c["quant_scale"])
        assert got.dtype == torch.int8 and np.array_equal(got.cpu().numpy(), ref), (shape, c["qmode"])
        # and the two-launch path it replaces
        y = ops.linear_w8a8(txq, tw, TDT[dt], dev["s_scalar"], dev.get("s_row"), dev.get("s_col"), dev.get("bias"))
        if c["act"] == "relu":
            y = torch.relu(y)
        two, _ = ops.quantize_act(y, c["qmode"], c["quant_scale"])
        assert torch.equal(got, two)


def test_opt_mlp_fused_chain_equals_unfused():
    """OPT MLP (reference models/opt.py:123-128): final_layer_norm(folded) -> fc1 (per-tensor Linear, bias) -> ReLU -> fc2 (per-tensor
    WithQuantScale, bias).  Fused: LayerNormQ -> fc1.forward_q(relu, consumer=fc2) -> fc2 on the int8 it was handed: three launches,
    no standalone quantiser, same bits as the module-by-module path fed with the same int8 norm output."""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale
    from autosmoothquant_amd.layers.nn.fused import LayerNormQ
    torch.manual_seed(3)
    H, F, M = 512, 2048, 200
    fc1f, fc2f = torch.nn.Linear(H, F), torch.nn.Linear(F, H)
    ln = torch.nn.LayerNorm(H)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(H)); ln.bias.copy_(0.1 * torch.randn(H))
    x = torch.randn(M, H) * 2
    with torch.no_grad():
        a1 = ln(x)
        a2 = torch.relu(fc1f(a1))
        ref = fc2f(a2).to(DEV)      # before from_float: like the reference, it rounds an fp32 source weight in place
    s1, s2 = float(a1.abs().max()) / 127, float(a2.abs().max()) / 127
    fc1 = W8A8BFP32OFP32Linear.from_float(fc1f, s1, act_quant="per-tensor").to(DEV)
    fc2 = W8A8BFP32OFP32LinearWithQuantScale.from_float(fc2f, s2, act_quant="per-tensor").to(DEV)
    lnq = LayerNormQ.from_float(ln, s1).to(DEV)
    for dt in (torch.float16, torch.float32):
        xd = x.to(DEV).to(dt)
        qa = lnq(xd)
        fused = fc2(fc1.forward_q(qa, fc2, act="relu"))
        unfused = fc2(torch.relu(fc1(qa)))
        assert fused.dtype == dt and torch.equal(fused, unfused)
        assert float((fused.float() - ref).norm() / ref.norm()) < 5e-2
    with pytest.raises(ValueError):
        fc1.forward_q(qa, W8A8BFP32OFP32LinearWithQuantScale(F, H, False, "per-token").to(DEV))


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(5, 64), (33, 4096), (7, 5120), (3, 11008), (2, 8192)])
def test_rmsnorm_and_silu_mul_fp_out_vs_oracle_and_torch(dt, shape):
    """Round 6 glue kernels for the reference's MODULE composition: asq_rmsnorm (fp out) == oracle/n1.py::rmsnorm_kernel_order bit for bit and within an ulp of the
    torch ops of HF's LlamaRMSNorm; asq_silu_mul exact form == oracle/n1.py::silu_mul_kernel_order bit for bit, both forms within an ulp or two of F.silu(g) * u."""
    from autosmoothquant_amd import ops
    M, K = shape
    if dt == "f32" and K > 8192:
        pytest.skip("fp32 rows up to 8192")
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    x = O.round_to(detrng.act_like(701, M + K, (M, K), scale=2.0), dt)
    w = O.round_to((detrng.act_like(702, K, (1, K), scale=0.3)[0] + 1.0), dt)
    xt, wt = torch.from_numpy(x).to(tdt).to(DEV), torch.from_numpy(w).to(tdt).to(DEV)
    y = ops.rmsnorm(xt, wt, 1e-5)
    assert np.array_equal(y.float().cpu().numpy(), n1.rmsnorm_kernel_order(x, dt, w, 1e-5))
    v = xt.float()
    ref = wt * (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)).to(tdt)
    ulp = {"f32": 2.0 ** -22, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    assert float(((y.float() - ref.float()).abs() / ref.float().abs().clamp_min(1e-3)).max()) <= 2 * ulp
    g = O.round_to(detrng.act_like(703, M * 3 + K, (M, K), scale=2.5), dt)
    u = O.round_to(detrng.act_like(704, M * 5 + K, (M, K), scale=1.5), dt)
    gt, ut = torch.from_numpy(g).to(tdt).to(DEV), torch.from_numpy(u).to(tdt).to(DEV)
    a = ops.silu_mul(gt, ut, fast=False)
    assert np.array_equal(a.float().cpu().numpy(), n1.silu_mul_kernel_order(g, u, dt))
    want = (torch.nn.functional.silu(gt) * ut).float()
    for fast in (False, True):
        got = ops.silu_mul(gt, ut, fast=fast).float()
        assert float(((got - want).abs() / want.abs().clamp_min(1e-2)).max()) <= 4 * ulp, (dt, shape, fast)
    with pytest.raises(ValueError):
        ops.silu_mul(gt, ut[:, : K // 2])


def test_harness_rmsnorm_module_takes_the_kernel():
    from autosmoothquant_amd import harness
    n = harness.RMSNorm(4096, 1e-5).to(DEV).half()
    with torch.no_grad():
        n.weight.copy_(torch.rand(4096, device=DEV) + 0.5)
        x = (torch.randn(2, 33, 4096, device=DEV) * 2).half()
        y = n(x)
        v = x.float()
        ref = n.weight * (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)).half()
    assert y.shape == x.shape and float((y.float() - ref.float()).abs().max()) <= 2.0 ** -9 * float(ref.float().abs().max())
    assert torch.equal(y, __import__("autosmoothquant_amd").ops.rmsnorm(x.view(-1, 4096), n.weight.detach(), 1e-5).view(x.shape))
