"""GPU (-m gpu): the HIP path, called through the C-ABI, against
  (1) the golden vectors captured from the reference's Python hot path (G1, G2, G4),
  (2) the oracle on seeded inputs at sizes it finishes in seconds (every GEMM kernel the
      dispatcher can pick, ragged M/N, unaligned operands, all epilogues),
  (3) size-independent properties at BASELINE.json's full sizes (row/column checksums
      computed in exact integers, sampled entries, M-sharding invariance).
Tolerances: int8 activations, int32 accumulators AND the dequantised outputs are compared
BIT-EXACTLY (the epilogue mirrors the reference's op order); the north-star contract is the
looser rtol 1e-3."""
import hashlib
import os

import numpy as np
import pytest
import torch

import detrng
import goldenio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import w8a8 as O

pytestmark = pytest.mark.gpu

TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from autosmoothquant_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def t_in(a, dt, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(TDT[dt]).to(dev)


def t_out(t):
    return t.detach().float().cpu().numpy()


def build_module(c, dev):
    from autosmoothquant_amd.layers.nn.linear import (W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale,
                                                      W8A8BFP32OFP32QKVLinear)
    N, K = c["wq"].shape
    if c["kind"] == "linear":
        m = W8A8BFP32OFP32Linear(K, N, c["use_bias"], c["act_quant"])
        m.dequant_scale = torch.tensor(c["dequant_scale"], dtype=torch.float32)
    elif c["kind"] == "quantscale":
        m = W8A8BFP32OFP32LinearWithQuantScale(K, N, c["use_bias"], c["act_quant"])
        m.dequant_scale = torch.tensor(c["dequant_scale"], dtype=torch.float32)
        if c["act_quant"] == "per-tensor":
            m.quant_scale = torch.tensor(c["quant_scale"], dtype=torch.float32)
    else:
        m = W8A8BFP32OFP32QKVLinear(c["qkv_size"], K, N, c["use_bias"], c["act_quant"])
        m.q_dequant_scale, m.k_dequant_scale, m.v_dequant_scale = (torch.tensor(float(v), dtype=torch.float32)
                                                                   for v in c["qkv_scales"])
    m.weight = torch.from_numpy(c["wq"].copy())
    if c["use_bias"]:
        m.bias = torch.from_numpy(np.asarray(c["bias"], np.float32).copy())
    return m.to(dev)


def act_mode(c):
    if c["act_quant"] == "per-token":
        return "per-token", 1.0
    if c["kind"] == "quantscale":
        return "per-tensor-div", c["quant_scale"]
    return "per-tensor-round", 1.0


G1 = goldenio.load_g1()
G2 = goldenio.load_g2()
G4 = goldenio.load_g4()


@pytest.mark.parametrize("c", G1 + G2, ids=lambda c: f"{c['id']}-{c['kind']}-{c['act_quant']}-{c['dt']}")
def test_golden_module_forward(c, dev):
    """Module forward == the reference's output, bit for bit; intermediates too."""
    from autosmoothquant_amd import ops
    m = build_module(c, dev)
    x = t_in(c["x"], c["dt"], dev)
    y = m(x)
    assert y.dtype == TDT[c["dt"]] and tuple(y.shape) == tuple(c["out"].shape)
    assert np.array_equal(t_out(y), c["out"], equal_nan=True)
    # the two halves separately: prologue (int8 activations) and native boundary (int32 accumulators)
    mode, qs = act_mode(c)
    xq, s_row = ops.quantize_act(x.reshape(-1, x.shape[-1]), mode, qs)
    assert np.array_equal(xq.cpu().numpy(), c["xq"])
    acc = torch.empty(c["acc"].shape, dtype=torch.int32, device=dev)
    m.i8cugemm.linear_a8_w8_o32_(xq, m.weight, acc)
    assert np.array_equal(acc.cpu().numpy(), c["acc"])


def test_golden_bigk_accumulator(dev):
    from autosmoothquant_amd import ops
    g = goldenio.load_g2_bigk()
    xq, _ = ops.quantize_act(t_in(g["x"], "f32", dev), "per-tensor-round")
    w = torch.from_numpy(g["wq"]).to(dev)
    acc = torch.empty(g["acc"].shape, dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xq, w, acc)
    assert np.array_equal(acc.cpu().numpy(), g["acc"])
    y = ops.linear_w8a8(xq, w, torch.float32, float(g["dequant_scale"]))
    assert np.array_equal(t_out(y), g["out"])


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("c", G4, ids=lambda c: c["id"])
def test_golden_config_shapes(c, dev):
    """LLaMA-2-7B / OPT-13B / Mixtral layer shapes: SHA-256 of xq, acc and out equal the reference's."""
    from autosmoothquant_amd import ops
    x, wq, bias = goldenio.g4_inputs(c)
    cc = dict(c, wq=wq, bias=bias)
    m = build_module(cc, dev)
    xt = t_in(x, c["dt"], dev)
    y = m(xt)
    mode, qs = act_mode(c)
    xq, _ = ops.quantize_act(xt, mode, qs)
    acc = torch.empty((c["M"], c["N"]), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xq, m.weight, acc)
    assert _sha(xq.cpu().numpy()) == c["sha_xq"]
    assert _sha(acc.cpu().numpy()) == c["sha_acc"]
    assert _sha(t_out(y).astype(np.float32)) == c["sha_out"]


# ---- oracle comparisons on seeded inputs -------------------------------------------------
GEMM_SHAPES = [
    # (M, N, K)  -- chosen to hit: generic (odd K / small) and the tiled kernels (K % 128 == 0) with ragged edges
    (1, 1, 1), (3, 5, 7), (17, 33, 95), (64, 64, 64), (65, 130, 200), (33, 48, 320),
    (128, 128, 128), (256, 256, 256), (300, 520, 384), (129, 257, 1024), (512, 768, 1024),
    (1000, 300, 512), (255, 4096, 256), (2048, 128, 2048),
    # few rows, K % 128 == 0: weight-streaming kernel, all token-tile counts, ragged N
    (1, 16, 128), (4, 4096, 4096), (16, 100, 256), (17, 33, 128), (32, 1000, 512), (33, 48, 1152), (48, 130, 384),
    (64, 64, 128), (63, 4099, 256), (32, 512, 11008),
    # 32-channel items (N >= 14336, or K >= 16384): ragged N inside the second channel tile
    (32, 14352, 256), (48, 14337, 128), (16, 4100, 16384),
    # 64 < M on a small weight: weight-streaming kernel with several 64-row m-blocks (balanced, ragged M and N,
    # channel-tile count not a multiple of 8 -> padding items)
    (65, 256, 128), (100, 200, 384), (129, 257, 1024), (200, 1000, 256), (256, 4096, 512), (300, 520, 384),
    # same row counts on weights past the work threshold -> 128-row tiled kernel (p8h) with a partial tile / split-K
    (100, 11008, 4096), (160, 4096, 11008), (256, 512, 6144),
    # >= 144 tiles of 256 x 256: the 256-row kernel (p8); everything tiled above runs p8h
    (3072, 3072, 256), (2900, 3300, 384),
    # 32..128 tiles of 128 x 256: the 128 x 128 kernel (p8q), ragged M / N, 1..5 K-tiles (every ring phase), with its K split
    (384, 4096, 128), (400, 4100, 256), (1000, 2052, 384), (450, 7000, 512), (640, 5120, 640), (384, 4096, 2048), (256, 11000, 4096),
]


@pytest.mark.parametrize("shape", GEMM_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gemm_i8_i32_vs_oracle(shape, dev):
    from autosmoothquant_amd import ops
    M, N, K = shape
    x = detrng.int8_uniform(101, M * 7 + K, (M, K))
    w = detrng.int8_uniform(102, N * 5 + K, (N, K))
    out = torch.full((M, N), -7, dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), out)
    assert np.array_equal(out.cpu().numpy(), O.igemm(x, w))


def _fuzz_shapes():
    rng = np.random.default_rng(20240607)
    shapes = []
    for i in range(36):
        M = int(rng.choice([rng.integers(1, 65), rng.integers(65, 300), rng.integers(300, 900)]))
        N = int(rng.choice([rng.integers(1, 64), rng.integers(64, 600), rng.integers(600, 2200)]))
        K = int(128 * rng.integers(1, 20)) if i % 6 else int(rng.integers(1, 700))
        shapes.append((M, N, K))
    return shapes


@pytest.mark.parametrize("shape", _fuzz_shapes(), ids=lambda s: "x".join(map(str, s)))
def test_gemm_fuzz_vs_oracle(shape, dev):
    """Seeded random shapes across every dispatcher boundary (rows 1..900, ragged N, K % 128 == 0 and odd K), int32 and
    fp16 epilogues, each launched twice (a race in the LDS-DMA / barrier protocol shows up as run-to-run differences)."""
    from autosmoothquant_amd import ops
    M, N, K = shape
    x = detrng.int8_uniform(301, M * 3 + K, (M, K))
    w = detrng.int8_uniform(302, N * 7 + K, (N, K))
    xt, wt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    acc = O.igemm(x, w)
    for _ in range(2):
        out = torch.full((M, N), 5, dtype=torch.int32, device=dev)
        ops.gemm_i8_i32(xt, wt, out)
        assert np.array_equal(out.cpu().numpy(), acc)
    srow = (np.abs(detrng.normal(303, M, (M,))) * 1e-3 + 1e-4).astype(np.float32)
    bias = detrng.normal(304, N, (N,)).astype(np.float32)
    ref = O.dequant_epilogue(acc, np.float32(0.125), srow, bias, "f16")
    got = ops.linear_w8a8(xt, wt, torch.float16, 0.125, torch.from_numpy(srow).to(dev), None, torch.from_numpy(bias).to(dev))
    assert np.array_equal(got.float().cpu().numpy(), ref)


def test_gemm_transpose_detecting(dev):
    """x = one-hot rows selects rows of W: an m<->n swap or wrong fragment map cannot pass."""
    from autosmoothquant_amd import ops
    M, N, K = 256, 384, 256
    w = detrng.int8_uniform(103, 0, (N, K))
    x = np.zeros((M, K), np.int8)
    x[np.arange(M), (np.arange(M) * 37) % K] = 1
    out = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), out)
    assert np.array_equal(out.cpu().numpy(), w[:, (np.arange(M) * 37) % K].T.astype(np.int32))


def test_gemm_unaligned_operands_take_generic_path(dev):
    from autosmoothquant_amd import ops
    M, N, K = 130, 140, 256
    xb = torch.from_numpy(detrng.int8_uniform(104, 0, (M * K + 1,))).to(dev)
    wb = torch.from_numpy(detrng.int8_uniform(104, 1, (N * K + 3,))).to(dev)
    x, w = xb[1:].view(M, K), wb[3:].view(N, K)
    out = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(x, w, out)
    assert np.array_equal(out.cpu().numpy(), O.igemm(x.cpu().numpy(), w.cpu().numpy()))


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(5, 64), (33, 320), (7, 4096), (3, 11008), (2, 20480), (4, 24584), (9, 77),
                                   (1030, 512), (1500, 4096), (1025, 1000), (1026, 14336), (1100, 13312)])  # >= 1024 rows: wave-per-row kernel (the last two: 28 vectors per lane in fp16 / bf16)
def test_quantize_act_vs_oracle(dt, shape, dev):
    from autosmoothquant_amd import ops
    M, K = shape
    x = O.round_to(detrng.act_like(105, M + K, (M, K), scale=30.0), dt)
    x[M // 2, K // 3] = 1e4 if dt != "f16" else 6e4
    xt = t_in(x, dt, dev)
    xq, s = ops.quantize_act(xt, "per-token")
    rq, rs = O.act_quant_per_token(x, dt)
    assert np.array_equal(s.cpu().numpy(), rs) and np.array_equal(xq.cpu().numpy(), rq)
    xq, _ = ops.quantize_act(xt, "per-tensor-round")
    assert np.array_equal(xq.cpu().numpy(), O.act_quant_round(x, dt))
    xq, _ = ops.quantize_act(xt, "per-tensor-div", 0.7312)
    assert np.array_equal(xq.cpu().numpy(), O.act_quant_div(x, dt, np.float32(0.7312)))


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_per_tensor_div_quantiser_whole_domain(dt, dev):
    """per-tensor-div (`x / quant_scale` in x's dtype, then round / clamp: linear.py:289-292) without a per-element IEEE division: every
    16-bit input pattern (zeros, subnormals, +-inf, NaN, bf16 values beyond 2^60 that must take the plain division) against scales that put
    the quotients on and around half-integers, plus scales outside the fast range; fp32: 2^22 random bit patterns + the same edge values."""
    from autosmoothquant_amd import ops
    if dt == "f32":
        rng = np.random.default_rng(11)
        x = rng.integers(0, 2 ** 32, size=1 << 22, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
        x[:8] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3e38, -2.0 ** 60], np.float32)
        x[8:4104] = (np.arange(4096, dtype=np.float32) - 2048.0) * np.float32(0.25)
    elif dt == "f16":
        x = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16).astype(np.float32)
    else:
        x = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32).copy()
    x = x.reshape(-1, 1024)
    xt = t_in(x, dt, dev)
    for qs in (0.5, 0.7312, 1.0, 3.0, 1e-3, 0.0999755859375, 517.0, 2.0 ** -61, 2.0 ** 61, 1e-30, 6e4):
        with np.errstate(all="ignore"):
            want = O.act_quant_div(x, dt, np.float32(qs))
        got, _ = ops.quantize_act(xt, "per-tensor-div", qs)
        assert np.array_equal(got.cpu().numpy(), want), (dt, qs)


@pytest.mark.parametrize("out_dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("order", ["scale_first", "acc_first"])
@pytest.mark.parametrize("shape", [(67, 50, 96), (256, 384, 256), (130, 258, 128)])
def test_fused_epilogue_vs_oracle(out_dt, order, shape, dev):
    """All epilogue operand combinations incl. per-channel s_col (north star: 'per-channel dequant')."""
    from autosmoothquant_amd import ops
    M, N, K = shape
    xq = detrng.int8_uniform(106, M, (M, K))
    w = detrng.int8_uniform(107, N, (N, K))
    s_row = (np.abs(detrng.normal(108, 0, (M,))) * 0.01 + 1e-3).astype(np.float32)
    s_col = (np.abs(detrng.normal(108, 1, (N,))) * 0.01 + 1e-3).astype(np.float32)
    bias = detrng.normal(108, 2, (N,)).astype(np.float32)
    acc = O.igemm(xq, w)
    d = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    for use_row in (False, True):
        for use_col in (False, True):
            for use_bias in (False, True):
                got = ops.linear_w8a8(d(xq), d(w), TDT[out_dt], 0.00123, d(s_row) if use_row else None,
                                      d(s_col) if use_col else None, d(bias) if use_bias else None, order)
                ref = O.dequant_epilogue(acc, s_col if use_col else np.float32(0.00123), s_row if use_row else None,
                                         bias if use_bias else None, out_dt, order)
                assert np.array_equal(t_out(got), ref), (use_row, use_col, use_bias)


def test_int8_out_gemms(dev):
    """I8CUGEMM.linear_a8_w8_o8 / _o8_ / _b8_o8_ (no reference call sites; semantics restated:
    sat_i8(rne(alpha*acc + beta*c)))."""
    from autosmoothquant_amd._CUDA import I8CUGEMM
    g = I8CUGEMM()
    for (M, N, K) in [(37, 52, 96), (256, 256, 256), (520, 1000, 384), (300, 520, 1152), (2900, 3300, 128)]:   # generic, skinny, p8q, p8h, p8
        x = detrng.int8_uniform(109, M, (M, K))
        w = detrng.int8_uniform(110, N, (N, K))
        b = detrng.int8_uniform(111, N, (N,))
        acc = O.igemm(x, w).astype(np.float32)
        alpha, beta = np.float32(0.0009), np.float32(0.75)
        xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
        out = torch.zeros((M, N), dtype=torch.int8, device=dev)
        g.linear_a8_w8_o8(xd, wd, out, float(alpha))
        ref = np.clip(np.rint(alpha * acc), -128, 127).astype(np.int8)
        assert np.array_equal(out.cpu().numpy(), ref)
        c0 = detrng.int8_uniform(112, M, (M, N))
        out = torch.from_numpy(c0.copy()).to(dev)
        g.linear_a8_w8_o8_(xd, wd, out, float(alpha), float(beta))
        ref = np.clip(np.rint((alpha * acc).astype(np.float32) + (beta * c0.astype(np.float32)).astype(np.float32)), -128, 127).astype(np.int8)
        assert np.array_equal(out.cpu().numpy(), ref)
        out = g.linear_a8_w8_b8_o8_(xd, wd, torch.from_numpy(b).to(dev), float(alpha), float(beta))
        ref = np.clip(np.rint((alpha * acc).astype(np.float32) + (beta * b.astype(np.float32)[None, :]).astype(np.float32)), -128, 127).astype(np.int8)
        assert np.array_equal(out.cpu().numpy(), ref)


def test_empty_and_degenerate_inputs(dev):
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    m = W8A8BFP32OFP32Linear(64, 32, True, "per-token")
    m.weight = torch.from_numpy(detrng.int8_uniform(113, 0, (32, 64)))
    m = m.to(dev)
    y = m(torch.zeros(0, 64, dtype=torch.float16, device=dev))
    assert tuple(y.shape) == (0, 32) and y.dtype == torch.float16
    y = m(torch.zeros(2, 0, 64, dtype=torch.float16, device=dev))
    assert tuple(y.shape) == (2, 0, 32)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 63, dtype=torch.float16, device=dev))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 64, dtype=torch.float16))  # CPU tensor: no fallback
    assert m.bias.is_cuda                            # ... and nothing was moved for it
    # the module validates its own tensors once per tensor OBJECT: a replaced buffer is checked again, a restored one runs again
    good, x64 = m.weight, torch.randn(3, 64, device=dev).half()
    ref = m(x64)
    m.weight = good.to(torch.int16)
    with pytest.raises(ValueError):
        m(x64)
    m.weight = good.cpu()
    with pytest.raises(RuntimeError):
        m(x64)
    m.weight = good[:, :32].contiguous()
    with pytest.raises(ValueError):
        m(x64)
    m.weight = good.clone()
    assert torch.equal(m(x64), ref)
    # non-contiguous but reshape-able input
    xb = torch.randn(4, 128, device=dev).half()
    y1 = m(xb[:, ::2])
    y2 = m(xb[:, ::2].contiguous())
    assert torch.equal(y1, y2)


# ---- full-size properties (BASELINE.json sizes; the oracle would take minutes) -----------
@pytest.mark.parametrize("shape", [(4096, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (256, 5120, 20480),
                                   (1024, 4096, 4096), (128, 4096, 11008), (3072, 11008, 4096),    # 128 x 128 kernel, its split-K, tail peel
                                   (768, 11008, 4096), (640, 12288, 4096)],                         # r4's two one-round corners of the dispatcher (p16 on 129 tiles, p8h on 240)
                         ids=lambda s: "x".join(map(str, s)))
def test_full_size_checksums(shape, dev):
    """sum_n acc[m,n] == x[m,:] . (sum_n w[n,:]) and sum_m acc[m,n] == (sum_m x[m,:]) . w[n,:] in exact
    integers, plus 4096 sampled entries, plus M-sharding invariance (rows are independent)."""
    from autosmoothquant_amd import ops
    M, N, K = shape
    x = detrng.int8_uniform(120, M, (M, K))
    w = detrng.int8_uniform(121, N, (N, K))
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    out = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xd, wd, out)
    acc = out.cpu().numpy().astype(np.int64)
    assert np.array_equal(acc.sum(axis=1), x.astype(np.int64) @ w.astype(np.int64).sum(axis=0))
    assert np.array_equal(acc.sum(axis=0), w.astype(np.int64) @ x.astype(np.int64).sum(axis=0))
    idx = (detrng.u64(122, M, 4096) % np.uint64(M * N)).astype(np.int64)
    mi, ni = idx // N, idx % N
    ref = np.einsum("ik,ik->i", x[mi].astype(np.int64), w[ni].astype(np.int64))
    assert np.array_equal(acc[mi, ni], ref)
    # replica-parallel invariance: computing a row shard alone gives the same rows bit-for-bit
    lo, hi = M // 3, M // 3 + M // 4
    part = torch.empty((hi - lo, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xd[lo:hi], wd, part)
    assert torch.equal(part, out[lo:hi])


@pytest.mark.parametrize("shape", [(65536, 11008, 4096), (65536, 4096, 11008), (65536, 4096, 4096)], ids=lambda s: "x".join(map(str, s)))
def test_cfg3_row_count_checksums(shape, dev):
    """BASELINE configs[2] at its real row count (batch 32 x 2048 tokens = 65536 rows; LLaMA-2-7B gate/up, down and attention
    shapes, reference call sites models/llama.py:99-106,206-211).  The oracle would need hours, so: (1) exact-integer row and
    column checksums of the int32 accumulators, 8 chunks of 8192 rows, reduced on the device in int64 and compared with host
    integer arithmetic; (2) 2048 sampled accumulators against host dot products; (3) ONE 65536-row launch of the fused
    per-token fp16 epilogue equals the eight 8192-row launches bit for bit (rows are independent: the replica / chunk
    invariance the harness relies on)."""
    from autosmoothquant_amd import ops
    M, N, K = shape
    CH = 8192
    x = detrng.int8_uniform(150, K, (M, K))
    w = detrng.int8_uniform(151, N, (N, K))
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    wsum = w.astype(np.int64).sum(axis=0)
    col = torch.zeros(N, dtype=torch.int64, device=dev)
    acc = torch.empty((CH, N), dtype=torch.int32, device=dev)
    idx = (detrng.u64(152, M, 2048) % np.uint64(M * N)).astype(np.int64)
    mi, ni = idx // N, idx % N
    want = np.einsum("ik,ik->i", x[mi].astype(np.int64), w[ni].astype(np.int64))
    for c in range(M // CH):
        ops.gemm_i8_i32(xd[c * CH:(c + 1) * CH], wd, acc)
        a64 = acc.to(torch.int64)
        assert np.array_equal(a64.sum(dim=1).cpu().numpy(), x[c * CH:(c + 1) * CH].astype(np.int64) @ wsum), c
        col += a64.sum(dim=0)
        sel = (mi >= c * CH) & (mi < (c + 1) * CH)
        got = acc[torch.from_numpy(mi[sel] - c * CH).to(dev), torch.from_numpy(ni[sel]).to(dev)].cpu().numpy()
        assert np.array_equal(got.astype(np.int64), want[sel]), c
        del a64
    assert np.array_equal(col.cpu().numpy(), w.astype(np.int64) @ x.astype(np.int64).sum(axis=0))
    del acc
    s_row = torch.from_numpy((np.abs(detrng.normal(153, 0, (M,))) * 0.01 + 1e-3).astype(np.float32)).to(dev)
    whole = ops.linear_w8a8(xd, wd, torch.float16, 3.0e-4, s_row)
    for c in range(M // CH):
        part = ops.linear_w8a8(xd[c * CH:(c + 1) * CH], wd, torch.float16, 3.0e-4, s_row[c * CH:(c + 1) * CH].contiguous())
        assert torch.equal(whole[c * CH:(c + 1) * CH], part), c


def test_full_size_fused_forward_matches_unfused(dev):
    """4096^3 per-tensor W8A8 forward (north-star shape): fused epilogue == epilogue recomputed
    from the exact accumulators with the oracle's arithmetic."""
    from autosmoothquant_amd import ops
    M = N = K = 4096
    x = O.round_to(detrng.act_like(130, 0, (M, K), scale=40.0), "f16")
    w = detrng.int8_uniform(131, 0, (N, K))
    xd, wd = t_in(x, "f16", dev), torch.from_numpy(w).to(dev)
    y = ops.linear_w8a8_forward(xd, wd, "per-tensor-round", 1.0, 1.0 / 8192)
    xq, _ = ops.quantize_act(xd, "per-tensor-round")
    assert np.array_equal(xq.cpu().numpy(), O.act_quant_round(x, "f16"))
    acc = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xq, wd, acc)
    ref = O.dequant_epilogue(acc.cpu().numpy(), np.float32(1.0 / 8192), None, None, "f16")
    assert np.array_equal(t_out(y), ref)


@pytest.mark.parametrize("counts", [[300, 0, 17, 256, 1, 511], [40, 40], [0, 0, 5], [700]])
@pytest.mark.parametrize("per_token", [False, True])
@pytest.mark.parametrize("K", [256, 8192])   # (Mixtral w2 has the long K)
@pytest.mark.parametrize("odt", [torch.float16, torch.float32])   # (2-byte: row / staged epilogues of the 16 x 16 layout; 4-byte: its 64-row staging passes)
def test_grouped_launch_equals_per_group_calls(counts, per_token, K, odt, dev):
    """Mixtral-style grouped launch (one kernel over all experts, routing offsets on the device) ==
    one asq_linear_w8a8 call per expert, bit for bit; empty and ragged groups included."""
    from autosmoothquant_amd import ops
    G, N = len(counts), 320
    M = sum(counts)
    xq = torch.from_numpy(detrng.int8_uniform(140, M + G, (M, K))).to(dev)
    w = torch.from_numpy(detrng.int8_uniform(141, G, (G, N, K))).to(dev)
    sg = torch.from_numpy((np.abs(detrng.normal(142, G, (G,))) * 1e-3 + 1e-4).astype(np.float32)).to(dev)
    bias = torch.from_numpy(detrng.normal(143, G, (G, N)).astype(np.float32)).to(dev)
    s_row = torch.from_numpy((np.abs(detrng.normal(144, M, (M,))) * 0.01 + 1e-3).astype(np.float32)).to(dev) if per_token else None
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
    for use_bias in (False, True):
        got = ops.linear_w8a8_grouped(xq, w, offs, sg, odt, s_row, bias if use_bias else None)
        o = 0
        for g, c in enumerate(counts):
            if c:
                ref = ops.linear_w8a8(xq[o:o + c].contiguous(), w[g], odt, float(sg[g]), None if s_row is None else s_row[o:o + c].contiguous(),
                                      None, bias[g] if use_bias else None)
                assert torch.equal(got[o:o + c], ref), (g, c, use_bias)
            o += c


@pytest.mark.parametrize("counts,N,K", [
    ([256] * 8, 256, 2048),                       # 8 tiles, one per XCD: every tile is a tail tile, split in 2 (16 K tiles / 8)
    ([300, 0, 17, 256, 1, 511, 100, 90], 512, 8192),   # 22 tiles: XCDs hold 2 or 3 tail tiles, 8 K pieces each
    ([1000, 24, 700, 3, 0, 260], 1000, 4096),     # ragged N (no staged epilogue), 4 K pieces
    ([129] * 40, 256, 3072),                      # 40 groups, 5 tail tiles per XCD x 3 pieces (24 K tiles / 8)
    ([2048, 2048], 4352, 2048),                   # 272 tiles = 34 per XCD: one full round + 2 tail tiles x 2 pieces
])
def test_grouped_launch_tail_split_is_exact(counts, N, K, dev):
    """The grouped launch's scheduler (asq_gemm_p8.h): tiles shared equally by the XCDs and, with a workspace, the tiles of a less than half
    full last round split along K and summed in the launch.  With workspace == without == one call per group, bit for bit (fp16 epilogue with
    row scales and bias; int32 sums are order-independent), three launches in a row on one workspace, tickets left at zero."""
    from autosmoothquant_amd import ops, _lib as L
    lib = L.lib()
    G, M = len(counts), sum(counts)
    st = torch.cuda.current_stream().cuda_stream
    xq = torch.from_numpy(detrng.int8_uniform(190, M + G, (M, K))).to(dev)
    w = torch.from_numpy(detrng.int8_uniform(191, G + N, (G, N, K))).to(dev)
    sg = torch.from_numpy((np.abs(detrng.normal(192, G, (G,))) * 1e-3 + 1e-4).astype(np.float32)).to(dev)
    bias = torch.from_numpy(detrng.normal(193, G, (G, N)).astype(np.float32)).to(dev)
    s_row = torch.from_numpy((np.abs(detrng.normal(194, M, (M,))) * 0.01 + 1e-3).astype(np.float32)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
    n = lib.asq_grouped_workspace_bytes(M, N, K, G)
    hdr = lib.asq_workspace_header_bytes()
    assert n == hdr + (64 << 20)
    ws = torch.full((n,), 0x3C, dtype=torch.uint8, device=dev)
    L.check(lib.asq_workspace_init(ws.data_ptr(), n, st), "init")
    args = (xq.data_ptr(), w.data_ptr())
    tail = (offs.data_ptr(), G, M, N, K, sg.data_ptr(), s_row.data_ptr(), bias.data_ptr())
    plain = torch.empty((M, N), dtype=torch.float16, device=dev)
    L.check(lib.asq_linear_w8a8_grouped(*args, plain.data_ptr(), L.ASQ_F16, *tail, st), "grouped")
    for rep in range(3):
        got = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        L.check(lib.asq_linear_w8a8_grouped_ws(*args, got.data_ptr(), L.ASQ_F16, *tail, ws.data_ptr(), n, st), "grouped_ws")
        assert torch.equal(got, plain), rep
    assert int(ws[16:hdr].view(torch.int32).abs().max()) == 0
    o = 0
    for g, c in enumerate(counts):
        if c:
            ref = ops.linear_w8a8(xq[o:o + c].contiguous(), w[g], torch.float16, float(sg[g]), s_row[o:o + c].contiguous(), None, bias[g])
            assert torch.equal(plain[o:o + c], ref), (g, c)
        o += c
    assert torch.equal(ops.linear_w8a8_grouped(xq, w, offs, sg, torch.float16, s_row, bias), plain)   # (the module-level wrapper keeps its own workspace)


def test_forward_outputs_never_require_grad(dev):
    """The reference decorates forward with @torch.no_grad(); here no autograd graph is recorded at all, so even an input
    that requires grad (under grad mode) yields an output that does not."""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, W8A8BFP32OFP32QKVLinear
    x = torch.randn(5, 256, device=dev, requires_grad=True)
    with torch.enable_grad():
        for m in (W8A8BFP32OFP32Linear(256, 128), W8A8BFP32OFP32LinearWithQuantScale(256, 128, False, "per-token"),
                  W8A8BFP32OFP32QKVLinear([64, 32, 32], 256, 128)):
            y = m.to(dev)(x)
            assert not y.requires_grad and y.grad_fn is None


def test_large_shapes_with_4_byte_and_ragged_outputs(dev):
    """>= 144 tiles of 256 x 256 with outputs gemm_i8_p16 does not carry (fp32, int32) go to p8's 16 x 16 x 64 mode, ragged edges to the bounded staged /
    direct epilogues of that layout: int32 == torch._int_mm (an independent exact GEMM), fp32 / fp16 == the same fp32 operation sequence in torch."""
    from autosmoothquant_amd import ops
    for (M, N, K) in [(3072, 3072, 512), (3000, 3100, 384), (2900, 3330, 256)]:
        xq = torch.from_numpy(detrng.int8_uniform(210, M, (M, K))).to(dev)
        w = torch.from_numpy(detrng.int8_uniform(211, N, (N, K))).to(dev)
        Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8    # (_int_mm wants multiples of 8)
        xp = torch.zeros((Mp, K), dtype=torch.int8, device=dev); xp[:M] = xq
        wp = torch.zeros((Np, K), dtype=torch.int8, device=dev); wp[:N] = w
        acc = torch._int_mm(xp, wp.t())[:M, :N].contiguous()
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        ops.gemm_i8_i32(xq, w, out)
        assert torch.equal(out, acc), ("i32", M, N, K)
        s_row = torch.from_numpy((np.abs(detrng.normal(212, M, (M,))) * 0.01 + 1e-3).astype(np.float32)).to(dev)
        bias = torch.from_numpy(detrng.normal(213, N, (N,)).astype(np.float32)).to(dev)
        ds = (torch.tensor(2e-3, dtype=torch.float32, device=dev) * s_row)[:, None]
        ref = ds * acc.float() + bias[None, :]
        assert torch.equal(ops.linear_w8a8(xq, w, torch.float32, 2e-3, s_row, None, bias), ref), ("f32", M, N, K)
        assert torch.equal(ops.linear_w8a8(xq, w, torch.float16, 2e-3, s_row, None, bias), ref.half()), ("f16", M, N, K)
        assert torch.equal(ops.linear_w8a8(xq, w, torch.bfloat16, 2e-3, None, None, None), (torch.tensor(2e-3, dtype=torch.float32, device=dev) * acc.float()).bfloat16()), ("bf16", M, N, K)


@pytest.mark.parametrize("kern,ksplit,mma", [("p16", 0, 16), ("p4x16", 0, 16), ("p8", 3, 16), ("p8", 1, 16), ("p4", 0, 16), ("p8h", 4, 16), ("p8h", 0, 16), ("p8q", 0, 16), ("p8q", 3, 16),
                                             ("p8h", 4, 32), ("p8h", 0, 32), ("p8q", 0, 32), ("p8q", 3, 32), ("skinny", 0, 16), ("generic", 0, 16)])
def test_forced_kernel_paths_in_a_child_process(kern, ksplit, mma, dev):
    """Dispatcher branches the shape heuristics never pick by themselves -- notably split-K on the 256-row kernel (pick_kernel hands
    p8 only >= 144 tiles, pick_ksplit splits only < 118) the 32 x 32 x 32 kernels p8 / p4 that gemm_i8_p16 replaced as the default, p16 itself on ragged shapes -- forced through ASQ_GEMM_KERNEL / ASQ_KSPLIT (read once per process,
    hence the child) and compared with the oracle's exact GEMM, int32 and fused fp16 epilogue, ragged M and N."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import detrng
from oracle import w8a8 as O
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(300, 520, 1536), (64, 256, 512), (512, 768, 640)]:
    x, w = detrng.int8_uniform(160, M, (M, K)), detrng.int8_uniform(161, N, (N, K))
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    out = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(xd, wd, out)
    acc = O.igemm(x, w)
    assert np.array_equal(out.cpu().numpy(), acc), ("i32", M, N, K)
    bias = detrng.normal(162, N, (N,)).astype(np.float32)
    s_row = (np.abs(detrng.normal(163, M, (M,))) * 0.01 + 1e-3).astype(np.float32)
    y = ops.linear_w8a8(xd, wd, torch.float16, 2e-3, torch.from_numpy(s_row).to(dev), None, torch.from_numpy(bias).to(dev))
    ref = O.dequant_epilogue(acc, np.float32(2e-3), s_row, bias, "f16")
    assert np.array_equal(y.float().cpu().numpy(), ref), ("f16", M, N, K)
    # scalar-scale epilogue, bf16; per-channel scales + bias, f16 (the interior row epilogues have their own code for each)
    y = ops.linear_w8a8(xd, wd, torch.bfloat16, 3e-3, None, None, None)
    assert np.array_equal(y.float().cpu().numpy(), O.dequant_epilogue(acc, np.float32(3e-3), None, None, "bf16")), ("bf16", M, N, K)
    s_col = (np.abs(detrng.normal(164, N, (N,))) * 1e-3 + 1e-4).astype(np.float32)
    y = ops.linear_w8a8(xd, wd, torch.float16, 1.0, None, torch.from_numpy(s_col).to(dev), torch.from_numpy(bias).to(dev))
    assert np.array_equal(y.float().cpu().numpy(), O.dequant_epilogue(acc, s_col, None, bias, "f16")), ("f16 per-channel", M, N, K)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ASQ_GEMM_KERNEL=kern)
    if mma == 32:   # the 128-row kernels on v_mfma_i32_32x32x32_i8 (their form before the 16 x 16 x 64 one became the default; still what fp8 runs on)
        env["ASQ_MMA"] = "32"
    if ksplit:
        env["ASQ_KSPLIT"] = str(ksplit)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_stream_k_weight_stream_in_its_dispatch_region(dev):
    """gemm_i8_wstream (stream-K over 128-channel groups, in-launch ticket reduction; asq_gemm_wstream.h) at the shapes the dispatcher gives it --
    cfg4's per-GPU shape 32 x 5120 x 20480 (+ 16 rows) and Mixtral's w2 at 64 rows: exact int32 against the first-generation kernel (same call
    without a workspace; itself pinned to the oracle at these sizes by the checksum tests), exact fp16 epilogue with row scales and bias, three
    launches in a row and a hipGraph replay (the tickets must return to zero each time), header clean afterwards."""
    from autosmoothquant_amd import ops, _lib as L
    lib = L.lib()
    hdr = lib.asq_workspace_header_bytes()
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in [(32, 5120, 20480), (16, 5120, 20480), (17, 5120, 20480), (64, 4096, 14336)]:
        n = lib.asq_gemm_workspace_bytes(M, N, K)
        assert n > hdr, (M, N, K)                                            # the stream-K kernel is what this shape gets
        x = torch.from_numpy(detrng.int8_uniform(170, M, (M, K))).to(dev)
        w = torch.from_numpy(detrng.int8_uniform(171, N % 997, (N, K))).to(dev)
        ref = torch.empty((M, N), dtype=torch.int32, device=dev)
        L.check(lib.asq_gemm_i8_i32(x.data_ptr(), w.data_ptr(), ref.data_ptr(), M, N, K, None, 0, st), "first generation")
        ws = torch.full((n,), 0xA5, dtype=torch.uint8, device=dev)           # garbage first: the header comes from asq_workspace_init alone
        L.check(lib.asq_workspace_init(ws.data_ptr(), n, st), "init")
        for rep in range(3):
            out = torch.full((M, N), -1, dtype=torch.int32, device=dev)
            L.check(lib.asq_gemm_i8_i32(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, ws.data_ptr(), n, st), "stream-K")
            assert torch.equal(out, ref), (M, N, K, rep)
        assert int(ws[16:hdr].view(torch.int32).abs().max()) == 0            # tickets back at zero
        s_row = (torch.rand(M, device=dev) * 0.01 + 1e-3)
        bias = torch.randn(N, device=dev)
        y0 = torch.empty((M, N), dtype=torch.float16, device=dev)
        y1 = torch.empty_like(y0)
        a = (x.data_ptr(), w.data_ptr())
        L.check(lib.asq_linear_w8a8(*a, y0.data_ptr(), L.ASQ_F16, M, N, K, 2e-3, s_row.data_ptr(), None, bias.data_ptr(), 0, None, 0, st), "f16 first generation")
        L.check(lib.asq_linear_w8a8(*a, y1.data_ptr(), L.ASQ_F16, M, N, K, 2e-3, s_row.data_ptr(), None, bias.data_ptr(), 0, ws.data_ptr(), n, st), "f16 stream-K")
        assert torch.equal(y0, y1), (M, N, K)
    # the module path (ops keeps an initialised workspace per stream) and a captured graph replayed three times
    M, N, K = 32, 5120, 20480
    xf = (torch.randn(M, K, device=dev) * 20).half()
    w = torch.from_numpy(detrng.int8_uniform(172, 7, (N, K))).to(dev)
    bias = torch.randn(N, device=dev)
    # a shape whose GEMM needs no scratch goes first on the same stream: its int8 activations must not land on the shared buffer's header
    ops.linear_w8a8_forward(xf[:, :4096].contiguous(), w[:4096, :4096].contiguous(), "per-token", 1.0, 1e-3, None, None)
    want = ops.linear_w8a8_forward(xf, w, "per-token", 1.0, 1e-3, None, bias)
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        ops.linear_w8a8_forward(xf, w, "per-token", 1.0, 1e-3, None, bias)
    torch.cuda.current_stream().wait_stream(cap)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        got = ops.linear_w8a8_forward(xf, w, "per-token", 1.0, 1e-3, None, bias)
    for _ in range(3):
        got.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    # a capture WITHOUT a warm-up on its stream: the graph owns its workspace (init recorded with it); the eager call that follows on the same
    # stream handle must not inherit a buffer whose header was only recorded, never written
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s2):
        got2 = ops.linear_w8a8_forward(xf, w, "per-token", 1.0, 1e-3, None, bias)
    with torch.cuda.stream(s2):
        eager = ops.linear_w8a8_forward(xf, w, "per-token", 1.0, 1e-3, None, bias)
    torch.cuda.synchronize()
    assert torch.equal(eager, want)
    for _ in range(2):
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(got2, want)


def test_stream_k_everywhere_and_the_workspace_contract_in_child_processes(dev):
    """ASQ_SK_IMPL=1 sends EVERY few-row shape the kernel can run to it (ragged N, rows not a multiple of 16, one K unit, blocks spanning several
    groups, fewer units than CUs): exact against the oracle.  And the contract's loud failure: a workspace that never went through
    asq_workspace_init makes the launch trap (the process dies with a HIP error) instead of returning wrong sums."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import detrng
from oracle import w8a8 as O
from autosmoothquant_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1]
shapes = [(3, 200, 256), (17, 1000, 384), (33, 130, 128), (70, 4100, 1024), (128, 16, 4096), (5, 129, 896), (32, 4096, 4096), (1, 11008, 4096), (100, 512, 2048)]
for (M, N, K) in shapes:
    n = lib.asq_gemm_workspace_bytes(M, N, K)
    assert n > lib.asq_workspace_header_bytes(), (M, N, K, n)
    x, w = detrng.int8_uniform(180, M, (M, K)), detrng.int8_uniform(181, N, (N, K))
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    ws = torch.full((n,), 0x5A, dtype=torch.uint8, device=dev)
    if mode == "ok":
        L.check(lib.asq_workspace_init(ws.data_ptr(), n, st), "init")
    for rep in range(2):
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        L.check(lib.asq_gemm_i8_i32(xd.data_ptr(), wd.data_ptr(), out.data_ptr(), M, N, K, ws.data_ptr(), n, st), "stream-K")
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), O.igemm(x, w)), (M, N, K, rep)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ASQ_SK_IMPL="1")
    r = subprocess.run([sys.executable, "-c", code, "ok"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", code, "uninitialised"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "ok" not in r.stdout, "a launch on an uninitialised workspace must fail loudly"


def test_forward_workspace_cache_streams_growth_and_graph_replay(dev):
    """Decode-sized module calls reuse one workspace per (thread, device, stream): results stay exact when the shapes
    grow (the slot is re-allocated), when two streams run interleaved, and when a hipGraph captured against an outgrown
    slot is replayed afterwards (retired slots stay alive)."""
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(77)
    K = 512
    w_np = rng.integers(-127, 128, size=(320, K), dtype=np.int8)
    w = torch.from_numpy(w_np).to(dev)

    def want(x_np, mode):
        return O.linear_forward(x_np, "f16", w_np, 0.01, None, "per-token" if mode == "per-token" else "per-tensor")

    def case(M, mode):
        x_np = (rng.standard_normal((M, K)) * 30).astype(np.float16).astype(np.float32)
        return x_np, t_in(x_np, "f16", dev), want(x_np, mode)

    # 1. capture a graph on a side stream while its slot is small
    x_np, xg, want_g = case(4, "per-tensor-round")
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ops.linear_w8a8_forward(xg, w, "per-tensor-round", 1.0, 0.01)   # warm (allocates the slot outside capture)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            yg = ops.linear_w8a8_forward(xg, w, "per-tensor-round", 1.0, 0.01)
    # 2. growing shapes on the default stream and on the side stream, interleaved
    for M in (1, 7, 64, 700, 3000, 9000, 5):
        for mode in ("per-tensor-round", "per-token"):
            x_np, x, ref = case(M, mode)
            y0 = ops.linear_w8a8_forward(x, w, mode, 1.0, 0.01)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                y1 = ops.linear_w8a8_forward(x, w, mode, 1.0, 0.01)
            torch.cuda.current_stream(dev).wait_stream(side)
            assert np.array_equal(t_out(y0), ref) and np.array_equal(t_out(y1), ref)
    # 3. replay the early graph: its workspace pointer must still be valid
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(t_out(yg), want_g)


TAIL_SHAPES = [(1536, 11008, 4096), (4352, 4096, 4096), (1536, 11000, 4096), (3072, 11008, 8192), (1152, 14336, 4096),   # (r4: 768 x 11008 x 4096, the one p8h + tail shape, now runs as one round of p16)
               (2048, 11008, 4096), (2000, 12284, 4096)]   # (the last two: remainders of 88 / 128 tiles, run as one round of 128 x 256 tiles -- gemm_i8_p8h)


def _tail_case(shape, dev):
    M, N, K = shape
    xq = detrng.int8_uniform(901, M, (M, K))
    w = detrng.int8_uniform(902, N, (N, K))
    s_row = (np.abs(detrng.normal(903, 0, (M,))) * 0.01 + 1e-3).astype(np.float32)
    s_col = (np.abs(detrng.normal(903, 1, (N,))) * 0.01 + 1e-3).astype(np.float32)
    bias = detrng.normal(903, 2, (N,)).astype(np.float32)
    return xq, w, s_row, s_col, bias


def _tail_digests(shape, dev):
    """sha256 of the int32 accumulators and of the fp16 / bf16 / fp32 fused outputs (all operands) of one shape"""
    from autosmoothquant_amd import ops
    M, N, K = shape
    xq, w, s_row, s_col, bias = _tail_case(shape, dev)
    d = lambda a: torch.from_numpy(a).to(dev)
    txq, tw = d(xq), d(w)
    acc = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(txq, tw, acc)
    out = [_sha(acc.cpu().numpy())]
    for dt in ("f16", "bf16", "f32"):
        y = ops.linear_w8a8(txq, tw, TDT[dt], 1.0, d(s_row), d(s_col), d(bias))
        out.append(_sha(y.view(torch.int16 if dt != "f32" else torch.int32).cpu().numpy()))
    return out


def test_tail_peel_equals_the_plain_launch(dev):
    """The peeled launches (main + 128 x 128-tile remainder) produce the same bytes as the plain launch of a process started with
    ASQ_NO_TAIL=1 (the switch is read once per process, hence the child)."""
    import subprocess
    import sys
    code = ("import sys, json, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_hip_parity as T\n"
            "from autosmoothquant_amd import _lib\n"
            "assert not any(_lib.lib().asq_gemm_kernel_name(*s).endswith(b'+tail') for s in T.TAIL_SHAPES)\n"
            "print('DIGESTS ' + json.dumps([T._tail_digests(s, torch.device('cuda:0')) for s in T.TAIL_SHAPES]))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ASQ_NO_TAIL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    plain = json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")][-1][8:])
    assert plain == [_tail_digests(s, dev) for s in TAIL_SHAPES]


@pytest.mark.parametrize("shape", TAIL_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_tail_peel_launches(shape, dev):
    """Tile grids a few tiles over a multiple of 256 run as a main launch + a column remainder of 128 x 128 tiles (launch_gemm's tail peel): int32
    accumulators against the oracle on sampled rows over ALL columns, every epilogue operand on the peeled columns (s_col / bias offsets,
    per-token rows, int8-out), with and without a workspace (the remainder then runs unsplit)."""
    from autosmoothquant_amd import ops, _lib as L
    M, N, K = shape
    lib = L.lib()
    assert lib.asq_gemm_kernel_name(M, N, K).endswith(b"+tail")
    xq, w, s_row, s_col, bias = _tail_case(shape, dev)
    d = lambda a: torch.from_numpy(a).to(dev)
    txq, tw, tr, tc, tb = d(xq), d(w), d(s_row), d(s_col), d(bias)
    acc = torch.empty((M, N), dtype=torch.int32, device=dev)
    ops.gemm_i8_i32(txq, tw, acc)
    st = torch.cuda.current_stream().cuda_stream
    acc0 = torch.empty_like(acc)
    L.check(lib.asq_gemm_i8_i32(txq.data_ptr(), tw.data_ptr(), acc0.data_ptr(), M, N, K, None, 0, st), "no workspace")
    assert torch.equal(acc, acc0)
    rows = np.unique(np.concatenate([np.arange(0, M, 61), [M - 1]]))
    assert np.array_equal(acc.cpu().numpy()[rows], O.igemm(xq[rows], w))
    for dt in ("f16", "bf16", "f32"):
        got = ops.linear_w8a8(txq, tw, TDT[dt], 1.0, tr, tc, tb)
        out0 = torch.empty_like(got)
        L.check(lib.asq_linear_w8a8(txq.data_ptr(), tw.data_ptr(), out0.data_ptr(), {"f32": L.ASQ_F32, "f16": L.ASQ_F16, "bf16": L.ASQ_BF16}[dt], M, N, K, 1.0,
                                    tr.data_ptr(), tc.data_ptr(), tb.data_ptr(), L.ASQ_EPI_SCALE_FIRST, None, 0, st), "no workspace")
        assert torch.equal(got, out0), dt
        rows = np.arange(0, M, 131)
        ref = O.dequant_epilogue(acc.cpu().numpy()[rows], s_col, s_row[rows], bias, dt, "scale_first")
        assert np.array_equal(t_out(got)[rows], ref), dt
    q = ops.linear_w8a8_q8(txq, tw, torch.float16, 0.0007, None, None, tb, "relu", "per-tensor-div", 0.31)
    y = ops.linear_w8a8(txq, tw, torch.float16, 0.0007, None, None, tb)
    want, _ = ops.quantize_act(torch.relu(y), "per-tensor-div", 0.31)
    assert torch.equal(q, want)
    # the one-call forward (quantise + GEMM inside the C-ABI, its own workspace layout) on the same shape
    xf = (torch.randn(M, K, device=dev) * 20).half()
    y1 = ops.linear_w8a8_forward(xf, tw, "per-token", 1.0, 0.0007, None, tb)
    xq2, sr2 = ops.quantize_act(xf, "per-token")
    assert torch.equal(y1, ops.linear_w8a8(xq2, tw, torch.float16, 0.0007, sr2, None, tb))
