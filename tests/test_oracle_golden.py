"""CPU: the oracle (oracle/w8a8.py) replayed against the golden vectors captured from
the reference's own Python hot path (tests/golden/make_golden.py).  Bit-exact."""
import hashlib

import numpy as np
import pytest

import goldenio
from oracle import w8a8 as O


def _fwd(c, with_intermediates=True):
    if c["kind"] == "linear":
        return O.linear_forward(c["x"], c["dt"], c["wq"], c["dequant_scale"], c["bias"], c["act_quant"], True)
    if c["kind"] == "quantscale":
        return O.linear_with_quant_scale_forward(c["x"], c["dt"], c["wq"], c["dequant_scale"],
                                                 c.get("quant_scale"), c["bias"], c["act_quant"], True)
    return O.qkv_linear_forward(c["x"], c["dt"], c["wq"], c["qkv_scales"], c["qkv_size"], c["bias"],
                                c["act_quant"], True)


G1 = goldenio.load_g1()
G2 = goldenio.load_g2()


@pytest.mark.parametrize("c", G1, ids=lambda c: f"g1-{c['id']}-{c['kind']}-{c['act_quant']}-{c['dt']}")
def test_g1_matrix(c):
    out, xq, qs, acc = _fwd(c)
    assert np.array_equal(xq, c["xq"])
    assert np.array_equal(acc, c["acc"])
    assert np.array_equal(out, c["out"], equal_nan=True)
    assert O.is_representable(out, c["dt"])


@pytest.mark.parametrize("c", G2, ids=lambda c: f"g2-{c['id']}")
def test_g2_edges(c):
    out, xq, qs, acc = _fwd(c)
    assert out.shape == c["out"].shape
    assert np.array_equal(xq, c["xq"])
    assert np.array_equal(acc, c["acc"])
    assert np.array_equal(out, c["out"], equal_nan=True)


def test_g2_bigk_accumulator_magnitude():
    g = goldenio.load_g2_bigk()
    out, xq, qs, acc = O.linear_forward(g["x"], "f32", g["wq"], float(g["dequant_scale"]), None, "per-tensor", True)
    assert np.array_equal(acc, g["acc"]) and np.array_equal(out, g["out"])
    assert np.abs(acc).max() > 2 ** 28  # well past fp32's 2^24 integer range
    assert np.array_equal(O.igemm_c(xq, g["wq"]), g["acc"])


def test_g3_from_float():
    z, index = goldenio.load_g3()
    for name, wdt, kind, aq, iscale, qkv in index:
        W = z[f"W_{wdt}"]
        qkv_size = [int(v) for v in qkv.split(",")]
        if kind == "qkv":
            wq, sc = O.qkv_from_float(W, wdt, float(iscale), qkv_size, aq)
            assert np.array_equal(np.array(sc, np.float32), z[name + "_qkv_scales"])
        else:
            wq, alpha = O.linear_from_float(W, wdt, float(iscale), aq)
            assert np.float32(alpha) == z[name + "_dequant_scale"]
        assert np.array_equal(wq, z[name + "_wq"])
        assert np.array_equal(z[name + "_bias"], z[f"b_{wdt}"])
        # reference rounds the fp32 SOURCE weight in place (quantization.py:13-16); other dtypes untouched
        if wdt == "f32" and kind != "qkv":
            assert np.array_equal(z[name + "_src_after"], z[name + "_wq"].astype(np.float32))
        elif wdt != "f32":
            assert np.array_equal(z[name + "_src_after"], W)
    for wdt in ("f32", "f16", "bf16"):
        wq, sc = O.quantize_weight_per_channel_absmax(z[f"W_{wdt}"], wdt)
        assert np.array_equal(wq, z[f"pc_{wdt}_wq"]) and np.array_equal(sc, z[f"pc_{wdt}_scales"])
        q, s = O.dynamic_quantize_activation_per_token_absmax(z[f"dyn_tok_{wdt}_x"], wdt)
        assert np.array_equal(q, z[f"dyn_tok_{wdt}_q"]) and np.array_equal(s.reshape(-1), z[f"dyn_tok_{wdt}_s"])
        q, s = O.dynamic_quantize_activation_per_tensor_absmax(z[f"dyn_tok_{wdt}_x"], wdt)
        assert np.array_equal(q, z[f"dyn_ten_{wdt}_q"]) and np.float32(s) == z[f"dyn_ten_{wdt}_s"]
        r = O.dequantize_activation_w_per_channel_a_per_token(z[f"dq_{wdt}_acc"], z[f"dq_{wdt}_ws"],
                                                              z[f"dq_{wdt}_atok"], wdt)
        assert np.array_equal(r, z[f"dq_{wdt}_tok_out"])
        r = O.dequantize_activation_w_per_channel_a_per_tensor(z[f"dq_{wdt}_acc"], z[f"dq_{wdt}_ws"],
                                                               z[f"dq_{wdt}_atok"][0], wdt)
        assert np.array_equal(r, z[f"dq_{wdt}_ten_out"])


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


G4 = goldenio.load_g4()


@pytest.mark.parametrize("c", G4, ids=lambda c: c["id"])
def test_g4_config_shapes(c):
    """Config-sized cases: inputs regenerate from the build-owned RNG; SHA-256 of the
    reference's xq / acc / out must match."""
    x, wq, bias = goldenio.g4_inputs(c)
    if c["kind"] == "linear":
        out, xq, qs, acc = O.linear_forward(x, c["dt"], wq, c["dequant_scale"], bias, c["act_quant"], True)
    else:
        out, xq, qs, acc = O.linear_with_quant_scale_forward(x, c["dt"], wq, c["dequant_scale"], c["quant_scale"],
                                                             bias, c["act_quant"], True)
    assert _sha(xq) == c["sha_xq"]
    assert _sha(acc) == c["sha_acc"]
    assert _sha(out.astype(np.float32)) == c["sha_out"]
    for i, a, o in c["samples"]:
        assert int(acc.reshape(-1)[i]) == a and float(out.reshape(-1)[i]) == o


def test_c_igemm_matches_numpy_igemm():
    import detrng
    for (M, N, K) in [(1, 5, 3), (7, 33, 65), (64, 48, 1024 + 17), (130, 16, 4096)]:
        x = detrng.int8_uniform(5, M, (M, K))
        w = detrng.int8_uniform(6, N, (N, K))
        ref = x.astype(np.int64) @ w.astype(np.int64).T
        assert np.array_equal(O.igemm_numpy(x, w), ref)
        assert np.array_equal(O.igemm_c(x, w), ref)


def test_g5_fp8_quantisers_and_linears():
    """fp8 (e4m3fn) path: quantiser bytes + scales bit-exact vs the reference; linears within fp32 rounding."""
    import os
    from oracle import fp8 as F8
    z = np.load(os.path.join(goldenio.GOLDEN, "g5_fp8.npz"))
    for dt in ("f32", "f16", "bf16"):
        x = z[f"x_{dt}"]
        q, s = F8.per_tensor_quantize_fp8(x, dt)
        assert np.array_equal(q, z[f"pt_{dt}_q"]) and np.float32(s) == z[f"pt_{dt}_s"]
        q, s = F8.per_token_quantize_fp8(x, dt)
        assert np.array_equal(F8.e4m3fn_to_f32(q), F8.e4m3fn_to_f32(z[f"tok_{dt}_q"]), equal_nan=True)
        assert np.array_equal(s.reshape(-1), z[f"tok_{dt}_s"])
        assert np.array_equal(F8.static_per_tensor_quantize_fp8(x, dt, np.float32(0.0371)), z[f"st_{dt}_q"])
    wq, ws = F8.per_tensor_quantize_fp8(z["W"], "f32")
    assert np.array_equal(wq, z["wq"]) and np.float32(ws) == z["ws"]
    assert np.array_equal(z["ff_wq"], z["wq"]) and bool(z["ff_act_quant_is_true"]) and not bool(z["ff_use_bias"])
    for line in z["index"]:
        name, dt, aq, ub = str(line).split("|")
        o = F8.fp8_linear_dynamic_forward(z[name + "_x"], dt, z["wq"], z["ws"], z["b"] if int(ub) else None, aq)
        assert np.abs(o - z[name]).max() <= 1e-5 * np.abs(z[name]).max()
    o = F8.fp8_linear_dynamic_forward(z["x_f16"], "f16", z["wq"], z["ws"], None, "per-tensor")
    assert np.abs(o - z["dyn_f16_per-tensor_0"]).max() <= 4e-3 * np.abs(z["dyn_f16_per-tensor_0"]).max()
    o = F8.fp8_linear_static_forward(z["x_f32"], "f32", z["wq"], z["ws"], np.float32(0.0371), np.float32(0.0), z["b"])
    assert np.abs(o - z["static_f32_0.0"]).max() <= 1e-4
    assert np.float32(F8.per_tensor_quantize_fp8(np.zeros((0, 8), np.float32), "f32")[1]) == z["empty_scale"]


def test_fp8_codecs_match_torch_casts():
    import torch
    from oracle import fp8 as F8
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    rng = np.random.default_rng(1)
    r = np.concatenate([h, (rng.standard_normal(100000) * np.exp(rng.uniform(-12, 8, 100000))).astype(np.float32),
                        np.array([448, 464, 465, 480, -448, 2 ** -9, 2 ** -10, 1.5 * 2 ** -10, 57344, 61440, 65536, 0.0, -0.0], np.float32)])
    t = torch.from_numpy(r)
    for enc, dec, tdt in ((F8.f32_to_e4m3fn, F8.e4m3fn_to_f32, torch.float8_e4m3fn), (F8.f32_to_e5m2, F8.e5m2_to_f32, torch.float8_e5m2)):
        ref = t.to(tdt).float().numpy()
        assert np.array_equal(dec(enc(r)), ref, equal_nan=True)


# ---- G7: the reference's executable decoder-layer composition (Int8BaichuanLayer) ---------------
def test_g7_block_oracle_matches_reference_outputs():
    """oracle/block.py against the reference's Int8BaichuanLayer outputs.  The linears are bit-exact
    restatements; the fp16 attention / SiLU glue is torch CPU arithmetic with unspecified summation order,
    so the stated tolerance is 1e-6 of max |y| (observed: bit-identical for 2 of the 3 configs, 1.2e-8 for the third)."""
    from oracle import block as B
    g = goldenio.load_g7()
    for c in g["cases"]:
        P = B.convert(g["W"], g["scales"], c["qc"])
        y = B.layer_forward(g["x"], P, g["heads"], g["eps"])
        assert np.abs(y - c["y"]).max() <= 1e-6 * np.abs(c["y"]).max(), c["qc"]
        # and the int8 layer stays within quantisation error of the float layer (~1e-2 observed)
        assert np.abs(c["y"] - g["y_float"]).max() <= 2e-2 * np.abs(g["y_float"]).max()


# ---- G6: smooth_ln_fcs (weight-side conversion) --------------------------------------
def test_g6_smooth_ln_fcs_matches_reference():
    import torch
    from autosmoothquant_amd.quantize import smooth_ln_fcs
    from autosmoothquant_amd.harness import RMSNorm
    for c in goldenio.load_g6():
        tdt = torch.float16 if c["dt"] == "f16" else torch.float32
        H = c["ln_w"].size
        if c["kind"] == "layernorm":
            ln = torch.nn.LayerNorm(H)
            ln.bias.data = torch.from_numpy(c["ln_b"].copy())
        else:
            ln = RMSNorm(H)
        ln.weight.data = torch.from_numpy(c["ln_w"].copy())
        fcs = []
        for W in c["fcs"]:
            fc = torch.nn.Linear(H, W.shape[0], bias=False)
            fc.weight.data = torch.from_numpy(W.copy())
            fcs.append(fc)
        ln, fcs = ln.to(tdt), [fc.to(tdt) for fc in fcs]
        smooth_ln_fcs(ln, fcs if len(fcs) > 1 else fcs[0], torch.from_numpy(c["act"].copy()), c["model_type"], c["alpha"])
        # elementwise torch ops; pow() may differ in the last place between CPU vector ISAs
        tol = dict(rtol=2e-3, atol=0) if c["dt"] == "f16" else dict(rtol=2e-6, atol=0)
        np.testing.assert_allclose(ln.weight.detach().float().numpy(), c["ln_w_out"], **tol)
        if c["kind"] == "layernorm":
            np.testing.assert_allclose(ln.bias.detach().float().numpy(), c["ln_b_out"], **tol)
        for fc, ref in zip(fcs, c["fcs_out"]):
            np.testing.assert_allclose(fc.weight.detach().float().numpy(), ref, **tol)


def test_smooth_ln_fcs_rejects_mismatched_modules():
    import torch
    from autosmoothquant_amd.quantize import smooth_ln_fcs
    ln, fc = torch.nn.LayerNorm(8), torch.nn.Linear(8, 4)
    with pytest.raises(ValueError):
        smooth_ln_fcs(ln, fc, torch.ones(7))
    with pytest.raises(TypeError):
        smooth_ln_fcs(ln, torch.nn.Identity(), torch.ones(8))
