"""GPU (-m gpu): gate || up as ONE GEMM with the SiLU(gate) * up epilogue (asq_linear_w8a8_gate_up on gemm_i8_p16p; reference models/llama.py:206-211) against
  (1) oracle/n1.py::gate_up_silu_kernel_order on sampled rows (fixed-operation-order SiLU): bit for bit,
  (2) the composition it replaces -- linear_w8a8 (gate), linear_w8a8 (up), asq_silu_mul_quantize -- through the consumer's quantiser, both SiLU forms, per-tensor and
      per-token activations, plain operands and offset images: bit for bit,
  (3) the harness: a LLaMA layer's fused path with and without the gate || up GEMM gives the same hidden states."""
import numpy as np
import pytest
import torch

from oracle import n1 as N1
from oracle import w8a8 as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _case(M, F, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    wg = torch.randint(-128, 128, (F, K), generator=g, device=DEV, dtype=torch.int8)
    wu = torch.randint(-128, 128, (F, K), generator=g, device=DEV, dtype=torch.int8)
    x = torch.randn(M, K, generator=g, device=DEV) * 3.0
    x[:, torch.rand(K, generator=g, device=DEV) < 0.01] *= 20.0
    return wg, wu, x


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(4096, 2304, 512), (1024, 11008, 4096), (8192, 1152, 256)])
def test_gate_up_equals_the_composition_and_the_oracle(dt, shape):
    from autosmoothquant_amd import ops
    M, F, K = shape
    assert ops.gate_up_supported(M, F, K, TDT[dt])
    wg, wu, x = _case(M, F, K, M + F)
    xh = x.to(TDT[dt])
    w_gu = ops.interleave_gate_up(wg, wu)
    image = ops.weight_offset_image(w_gu)
    sg, su = 2.1e-4, 1.7e-4
    for mode in ("per-tensor-round", "per-token"):
        xq, s_row = ops.quantize_act(xh, mode)
        xo, s_row_o, row_off = ops.quantize_act_off(xh, mode)
        gate = ops.linear_w8a8(xq, wg, TDT[dt], sg, s_row)
        up = ops.linear_w8a8(xq, wu, TDT[dt], su, s_row)
        for fast in (False, True):
            a = ops.linear_w8a8_gate_up(xq, w_gu, TDT[dt], sg, su, s_row, fast)
            a_img = ops.linear_w8a8_gate_up(xo, image[0], TDT[dt], sg, su, s_row_o, fast, row_off, image[1])
            assert torch.equal(a.view(torch.int16), a_img.view(torch.int16)), (shape, mode, fast, "images")
            for per_token, qs in ((True, 1.0), (False, 0.013)):
                want = ops.silu_mul_quantize(gate, up, per_token, qs, fast=fast)
                got = ops.quantize_act(a, "per-token" if per_token else "per-tensor-div", qs)
                assert torch.equal(got[0], want[0]), (shape, mode, fast, per_token)
                if per_token:
                    assert torch.equal(got[1], want[1])
        # the oracle, fixed-order SiLU, on a row of every 256-row tile
        rows = [i * 256 + (i * 37) % 256 for i in range(M // 256)]
        a = ops.linear_w8a8_gate_up(xq, w_gu, TDT[dt], sg, su, s_row, False)[rows].float().cpu().numpy()
        ref = N1.gate_up_silu_kernel_order(xq[rows].cpu().numpy(), wg.cpu().numpy(), wu.cpu().numpy(), dt, sg, su, None if s_row is None else s_row[rows].cpu().numpy())
        assert np.array_equal(a, ref, equal_nan=True), (shape, mode)


def test_unsupported_shapes_are_refused_and_the_module_falls_back():
    from autosmoothquant_amd import ops
    f16 = torch.float16
    assert ops.gate_up_supported(65536, 11008, 4096, f16) and ops.gate_up_supported(8192, 14336, 4096, torch.bfloat16)
    assert not ops.gate_up_supported(512, 11008, 4096, f16)        # a single round: gemm_i8_p16p is the multi-round kernel
    assert not ops.gate_up_supported(4096, 11000, 4096, f16)       # F % 128
    assert not ops.gate_up_supported(4096, 11008, 4096, torch.float32)
    xq = torch.zeros(512, 256, dtype=torch.int8, device=DEV)
    w = torch.zeros(256, 256, dtype=torch.int8, device=DEV)
    with pytest.raises(ValueError):
        ops.linear_w8a8_gate_up(xq, w, f16, 1.0, 1.0)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_int8_out_epilogue_equals_the_per_tensor_quantiser_behind_it(dt):
    """asq_linear_w8a8_gate_up_q8 (a per-tensor consumer's int8 input straight from the epilogue) == quantize_act(gate || up GEMM's tensor, per-tensor-div, quant_scale) ==
    silu_mul_quantize(per-tensor) behind the two linears: both SiLU forms, per-tensor and per-token activations, plain and images, quant scales inside and outside the fast
    division's range (the huge quotients saturate alike)."""
    from autosmoothquant_amd import ops
    M, F, K = 4096, 2304, 512
    wg, wu, x = _case(M, F, K, 77)
    xh = x.to(TDT[dt])
    w_gu = ops.interleave_gate_up(wg, wu)
    image = ops.weight_offset_image(w_gu)
    sg, su = 2.1e-4, 1.7e-4
    for mode in ("per-tensor-round", "per-token"):
        xq, s_row = ops.quantize_act(xh, mode)
        xo, s_row_o, row_off = ops.quantize_act_off(xh, mode)
        gate = ops.linear_w8a8(xq, wg, TDT[dt], sg, s_row)
        up = ops.linear_w8a8(xq, wu, TDT[dt], su, s_row)
        for fast in (False, True):
            a = ops.linear_w8a8_gate_up(xq, w_gu, TDT[dt], sg, su, s_row, fast)
            for qs in (0.013, 0.37, 3e-5, 1e-19):
                want = ops.quantize_act(a, "per-tensor-div", qs)[0]
                got = ops.linear_w8a8_gate_up_q8(xq, w_gu, TDT[dt], sg, su, qs, s_row, fast)
                got_img = ops.linear_w8a8_gate_up_q8(xo, image[0], TDT[dt], sg, su, qs, s_row_o, fast, row_off, image[1])
                assert torch.equal(got, want), (dt, mode, fast, qs)
                assert torch.equal(got_img, want), (dt, mode, fast, qs, "images")
                if qs > 1e-10:
                    assert torch.equal(got, ops.silu_mul_quantize(gate, up, False, qs, fast=fast)[0]), (dt, mode, fast, qs, "composition")
            assert int((ops.linear_w8a8_gate_up_q8(xq, w_gu, TDT[dt], sg, su, 0.013, s_row, fast).to(torch.int16).abs() > 0).sum()) > M * F // 10   # (not a table of zeros)


@pytest.mark.parametrize("cfg", [None, {"qkv": "per-token", "fc1": "per-token"}, {"out": "per-tensor", "fc2": "per-tensor"}])
def test_llama_layer_fused_path_with_and_without_the_gate_up_gemm(cfg):
    """harness.LlamaLayer (hidden 512, inter 1536): the N1-fused path with the gate || up GEMM == the same path with two linears + silu_mul_q, bit for bit"""
    from autosmoothquant_amd import harness
    torch.manual_seed(3)
    fl = harness.LlamaLayer(512, 1536, 8).to(DEV)
    harness.init_llama_layer(fl, 0.05)
    fl = fl.to(DEV)
    h = torch.randn(4, 2048, 512, device=DEV)
    scales = harness.calibrate(fl, h[:1, :128])
    q = harness.to_w8a8(fl.half(), scales, cfg, both=True)
    q.use_fused = True
    hh = h.half()
    M = 4 * 2048
    assert q.gate_up.supported(M, torch.float16)
    with torch.no_grad():
        q.fuse_gate_up = True
        y1 = q(hh)
        q.fuse_gate_up = False
        y0 = q(hh)
        assert "_wgu" in q.gate_up.__dict__        # the fused kernel really ran
    assert torch.equal(y1, y0)


@pytest.mark.parametrize("images", [False, True])
def test_graph_replay_follows_a_weight_update_after_refresh(images):
    """a hipGraph captured on the fused path bakes GateUpSiLU's DERIVED buffers (the interleaved gate || up operand, its offset image) in.  They are rebuilt
    in place: after copy_ of new weights + harness.refresh_derived_operands() a replay gives what an eager forward gives; the buffers never move."""
    from autosmoothquant_amd import harness, ops
    from autosmoothquant_amd.layers.nn.fused import GateUpSiLU, QuantizedActivation
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    M, F, K = 4096, 2304, 512
    wg, wu, x = _case(M, F, K, 901)
    mods = []
    for w, ds in ((wg, 2.1e-4), (wu, 1.7e-4)):
        m = W8A8BFP32OFP32Linear(K, F, False, "per-tensor")
        m.weight = w.clone()
        m.dequant_scale = torch.tensor(ds)
        mods.append(m.to(DEV))
    gate, up = mods
    gu = GateUpSiLU(gate, up)
    root = torch.nn.ModuleDict({"gate": gate, "up": up, "gu": gu})
    xh = x.half()
    if images and not ops.offsets_supported(M, 2 * F, K, torch.float16):
        pytest.skip("offset operand images are switched off in this process (ASQ_OFFSETS=0)")
    if images:
        xo, s_row, row_off = ops.quantize_act_off(xh, "per-tensor-round")
        qa = QuantizedActivation(xo, s_row, torch.float16, (M,), row_off)
        assert gu.offset_image(M, torch.float16) is not None
    else:
        qa = gate.quantize_input(xh, consumers=(gu,))
    eager_old = gu(qa).clone()
    ptrs = (gu.__dict__["_wgu"][1].data_ptr(), gu.__dict__["_wgu_image"][0].data_ptr() if images else 0)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gu(qa)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        out_g = gu(qa)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager_old)
    wg2, wu2, _ = _case(M, F, K, 902)
    gate.weight.copy_(wg2)
    up.weight.copy_(wu2)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager_old)          # no refresh yet: the replay still reads the old derived operand (documented)
    assert harness.refresh_derived_operands(root) >= 1
    gr.replay()
    torch.cuda.synchronize()
    fresh = GateUpSiLU(gate, up)(qa)             # a module that has never seen the old weights
    assert torch.equal(out_g, fresh) and not torch.equal(fresh, eager_old)
    assert torch.equal(gu(qa), fresh)            # and the eager path agrees
    assert ptrs == (gu.__dict__["_wgu"][1].data_ptr(), gu.__dict__["_wgu_image"][0].data_ptr() if images else 0)
