"""GPU (-m gpu): the ONE-launch module forward for <= 16 rows (asq_gemm_skinny_fq.h: the activation quantiser as the GEMM's prologue, reference
layers/nn/linear.py:88-96, :283-292 in front of :93-104) against
  (1) the two-launch composition it replaces (ops.quantize_act + ops.linear_w8a8 never fuse) -- bit for bit, every dtype x mode x epilogue operand set,
  (2) the oracle's module forwards, incl. the edge rows the reference's tests hold (all-zero rows, inf / NaN, saturating values),
  (3) asq_forward_fused_supported(): the cases below really take the fused launch (the test is not vacuous), the neighbours do not."""
import numpy as np
import pytest
import torch

import detrng
from oracle import w8a8 as O

pytestmark = pytest.mark.gpu

TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
DEV = torch.device("cuda:0")


def _x(seed, M, K, dt, scale=30.0):
    x = O.round_to(detrng.act_like(seed, M + K, (M, K), scale=scale), dt)
    return x, torch.from_numpy(x).to(TDT[dt]).to(DEV)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("mode", ["per-tensor-round", "per-tensor-div", "per-token"])
def test_fused_equals_two_launches_and_the_oracle(dt, mode):
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(7)
    for (M, N, K) in ((1, 4096, 4096), (4, 4096, 4096), (3, 256, 512), (5, 11008, 4096), (8, 4096, 4096), (9, 1000, 1024), (16, 4096, 4096), (16, 12288, 4096),
                      (4, 4096, 11008), (2, 5120, 13824), (13, 100, 256), (7, 20, 128), (4, 36, 2048)):
        takes = M <= 4 or (mode != "per-token" and (N + 15) // 16 <= 256)   # what the default forward fuses by itself; the explicit entry point below runs every case
        assert ops.forward_is_fused(M, N, K, TDT[dt], mode) == takes, (M, N, K, mode)
        w = rng.integers(-128, 128, (N, K), dtype=np.int8)
        wt = torch.from_numpy(w).to(DEV)
        x, xt = _x(1000 + M + N, M, K, dt, scale=2.5 if mode == "per-tensor-round" else 30.0)
        if M > 2:
            x[M // 2, K // 3] = 3e4
            x[1, :] = 0.0                       # an all-zero row: per-token scale 0 -> 0 / 0 -> NaN -> int8 0
            xt = torch.from_numpy(x).to(TDT[dt]).to(DEV)
        qs = 0.7312
        for (has_col, has_bias) in ((False, False), (True, True), (False, True)):
            s_col = torch.from_numpy(rng.uniform(1e-4, 2e-4, N).astype(np.float32)).to(DEV) if has_col else None
            bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV) if has_bias else None
            ds = 1.0 if has_col else 1.37e-4
            got = ops.linear_w8a8_forward_fused(xt, wt, mode, qs, ds, s_col, bias)
            if takes:
                assert torch.equal(got.view(torch.uint8), ops.linear_w8a8_forward(xt, wt, mode, qs, ds, s_col, bias).view(torch.uint8))
            xq, s_row = ops.quantize_act(xt, mode, qs)
            want = ops.linear_w8a8(xq, wt, TDT[dt], ds, s_row, s_col, bias)
            assert torch.equal(got.view(torch.int32 if dt == "f32" else torch.int16), want.view(torch.int32 if dt == "f32" else torch.int16)), (M, N, K, has_col, has_bias)
        # the oracle directly (scalar scale + bias)
        b = rng.standard_normal(N).astype(np.float32)
        got = ops.linear_w8a8_forward_fused(xt, wt, mode, qs, 1.37e-4, None, torch.from_numpy(b).to(DEV)).float().cpu().numpy()
        with np.errstate(all="ignore"):
            if mode == "per-token":
                ref = O.linear_forward(x, dt, w, 1.37e-4, b, "per-token")
            elif mode == "per-tensor-round":
                ref = O.linear_forward(x, dt, w, 1.37e-4, b, "per-tensor")
            else:
                ref = O.linear_with_quant_scale_forward(x, dt, w, 1.37e-4, np.float32(qs), b, "per-tensor")
        assert np.array_equal(got, ref, equal_nan=True), (M, N, K)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_fused_edge_rows(dt):
    """inf / NaN / saturating rows through the fused prologue: the row maximum propagates NaN like torch.max, inf rows quantise to 0 / +-127 as the reference's
    CPU path does, quant scales outside the fast division's range take the plain division."""
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(11)
    M, N, K = 8, 512, 1024
    w = rng.integers(-128, 128, (N, K), dtype=np.int8)
    wt = torch.from_numpy(w).to(DEV)
    x = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 50, dt)
    x[0, 5] = np.inf
    x[1, 7] = np.nan
    x[2, :] = 0.0
    x[3, 9] = -np.inf
    x[4, :] = 6e4 if dt == "f16" else 3e38
    x[5, 11] = 1e-7
    xt = torch.from_numpy(x).to(TDT[dt]).to(DEV)
    for mode, qs in (("per-token", 1.0), ("per-tensor-round", 1.0), ("per-tensor-div", 0.5), ("per-tensor-div", 2.0 ** -61), ("per-tensor-div", 2.0 ** 61)):
        got = ops.linear_w8a8_forward_fused(xt, wt, mode, qs, 1e-4, None, None)
        xq, s_row = ops.quantize_act(xt, mode, qs)
        want = ops.linear_w8a8(xq, wt, TDT[dt], 1e-4, s_row, None, None)
        v = torch.int32 if dt == "f32" else torch.int16
        assert torch.equal(got.view(v), want.view(v)), (mode, qs)


def test_which_shapes_fuse():
    from autosmoothquant_amd import ops
    f16 = torch.float16
    assert ops.forward_is_fused(4, 4096, 4096, torch.float32) and ops.forward_is_fused(4, 11008, 4096, f16) and ops.forward_is_fused(1, 16384, 4096, f16)
    assert ops.forward_is_fused(2, 4096, 11008, f16)
    assert ops.forward_is_fused(8, 4096, 4096, f16) and ops.forward_is_fused(16, 4096, 4096, f16, "per-tensor-div")   # 5 .. 16 rows: per-tensor modes, one tile per block
    assert not ops.forward_is_fused(8, 4096, 4096, f16, "per-token")   # ... a row-maximum pass in every block loses
    assert not ops.forward_is_fused(8, 11008, 4096, f16)         # ... and so do several channel tiles per block
    assert not ops.forward_is_fused(17, 4096, 4096, f16)
    assert not ops.forward_is_fused(4, 32000, 4096, f16)         # more than 1024 channel tiles: measured slower
    assert not ops.forward_is_fused(4, 4096, 4000, f16)          # K % 128
    assert not ops.forward_is_fused(4, 5120, 20480, f16)         # 4 x 20480 B does not fit the resident image
    x = torch.zeros(17, 4096, dtype=f16, device=DEV)
    w = torch.zeros(256, 4096, dtype=torch.int8, device=DEV)
    with pytest.raises(ValueError):   # ASQ_ERR_DIM
        ops.linear_w8a8_forward_fused(x, w, "per-token", 1.0, 1.0)   # the explicit entry point refuses what the kernel cannot run


def test_module_forward_is_one_kernel_and_matches_the_module_on_more_rows():
    """the module path (layers/nn/linear.py) at 4 rows == rows 0..3 of the same module on 64 rows (two launches, another GEMM kernel): rows are independent"""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale
    g = torch.Generator().manual_seed(3)
    for cls, aq in ((W8A8BFP32OFP32Linear, "per-tensor"), (W8A8BFP32OFP32Linear, "per-token"), (W8A8BFP32OFP32LinearWithQuantScale, "per-tensor"),
                    (W8A8BFP32OFP32LinearWithQuantScale, "per-token")):
        m = cls(4096, 4096, True, aq)
        m.weight = torch.randint(-128, 128, (4096, 4096), generator=g, dtype=torch.int8)
        m.bias = torch.randn(4096, generator=g)
        m.dequant_scale = torch.tensor(2e-4)
        if "quant_scale" in m._buffers:
            m.quant_scale = torch.tensor(0.31)
        m = m.to(DEV)
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            x = (torch.randn(64, 4096, generator=g) * 20).to(dt).to(DEV)
            big = m(x)
            for rows in (1, 4, 16):
                assert torch.equal(m(x[:rows]), big[:rows]), (cls.__name__, aq, dt, rows)


@pytest.mark.parametrize("cls_name,aq", [("W8A8BFP32OFP32Linear", "per-tensor"), ("W8A8BFP32OFP32Linear", "per-token"), ("W8A8BFP32OFP32LinearWithQuantScale", "per-tensor")])
def test_cached_decode_call_follows_every_state_change(cls_name, aq):
    """Round 6: decode-sized module forwards run a cached call (_W8A8Base._fast: same input geometry, same state OBJECTS at the same versions -> empty + one C-ABI
    call).  It must be invisible: equal to the oracle, and every change of the module's state -- a scale rewritten in place, a new weight object, an in-place weight
    write, a bias write, another shape, another dtype, a CPU tensor -- gives what a fresh module gives."""
    from autosmoothquant_amd.layers.nn import linear as LN
    cls = getattr(LN, cls_name)
    K, N, M = 512, 768, 4
    g = torch.Generator().manual_seed(21)
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    bias = torch.randn(N, generator=g)
    x = (torch.randn(2, M // 2, K, generator=g) * 30).half()

    def fresh(w, b, ds, qs):
        m = cls(K, N, True, aq)
        m.weight, m.bias, m.dequant_scale = w.clone(), b.clone(), torch.tensor(ds)
        if "quant_scale" in m._buffers:
            m.quant_scale = torch.tensor(qs)
        return m.to(DEV)

    def oracle(w, b, ds, qs, xx):
        xin = xx.float().numpy().reshape(-1, K)
        if cls_name == "W8A8BFP32OFP32Linear":
            return O.linear_forward(xin, "f16", w.numpy(), ds, b.numpy(), aq)
        return O.linear_with_quant_scale_forward(xin, "f16", w.numpy(), ds, qs, b.numpy(), aq)
    ds, qs = 1.0 / 4096, 0.37
    m = fresh(wq, bias, ds, qs)
    xd = x.to(DEV)
    want = oracle(wq, bias, ds, qs, x)
    for i in range(3):                                   # first call records, the next ones take the cached call
        y = m(xd)
        assert y.shape == (2, M // 2, N) and np.array_equal(y.float().cpu().numpy().reshape(M, N), want), i
    assert "_fast_state" in m.__dict__
    m.dequant_scale.fill_(2 * ds)                        # in place: the version counter moves
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), oracle(wq, bias, 2 * ds, qs, x))
    m.dequant_scale = torch.tensor(ds)                   # a new object
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), want)
    if "quant_scale" in m._buffers:
        m.quant_scale.fill_(0.5)
        assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), oracle(wq, bias, ds, 0.5, x))
        m.quant_scale.fill_(qs)
    w2 = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    m.weight.copy_(w2.to(DEV))                           # in-place weight write: read through the pointer
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), oracle(w2, bias, ds, qs, x))
    m.weight = wq.to(DEV)                                # a new weight object
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), want)
    m.bias.add_(1.0)
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), oracle(wq, bias + 1.0, ds, qs, x))
    m.bias.sub_(1.0)
    x3 = (torch.randn(3, K, generator=g) * 30).half()   # another geometry, then back
    assert np.array_equal(m(x3.to(DEV)).float().cpu().numpy(), oracle(wq, bias, ds, qs, x3))
    assert np.array_equal(m(xd).float().cpu().numpy().reshape(M, N), want)
    xb = x.bfloat16()
    if cls_name == "W8A8BFP32OFP32Linear":
        refb = O.linear_forward(xb.float().numpy().reshape(-1, K), "bf16", wq.numpy(), ds, bias.numpy(), aq)
    else:
        refb = O.linear_with_quant_scale_forward(xb.float().numpy().reshape(-1, K), "bf16", wq.numpy(), ds, qs, bias.numpy(), aq)
    assert np.array_equal(m(xb.to(DEV)).float().cpu().numpy().reshape(M, N), refb)
    with pytest.raises(RuntimeError):
        m(x)                                             # a CPU tensor still fails loudly
    xt = xd.transpose(0, 1)                              # non-contiguous: the full path flattens it
    assert torch.equal(m(xt), m(xt.contiguous()))
