"""GPU (-m gpu): the offset operand images (include/asq_hip.h) through the C-ABI.

The claim under test is exactness: the images follow the two stated rules (oracle/offsets.py), and everything computed from them -- int32
accumulators, fp outputs, module forwards -- equals the plain path (itself pinned to the reference's arithmetic by test_hip_parity.py) in every bit.
Tolerance: none."""
import numpy as np
import pytest
import torch

from oracle import offsets as OFF, w8a8 as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TDT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _edge_rows_i8(a, rng):
    """rows that exercise every branch of the offset rules"""
    n = a.shape[0]
    if n > 0: a[0] = 127
    if n > 1: a[1] = -128
    if n > 2: a[2] = 0
    if n > 3: a[3] = rng.integers(-3, 4, a.shape[1]); a[3, 0] = 127; a[3, 1] = -128      # both extremes: no offset fits
    if n > 4: a[4] = rng.integers(-3, 4, a.shape[1]); a[4, -1] = 126                      # positive side taken: negative offset
    if n > 5: a[5] = rng.integers(-3, 4, a.shape[1]); a[5, 0] = -127
    return a


def test_weight_image_equals_oracle():
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(0)
    for (N, K) in ((4, 16), (64, 128), (260, 4096), (1024, 11008), (8, 65536)):
        w = np.clip(np.rint(rng.standard_normal((N, K)) * 22), -128, 127).astype(np.int8)
        _edge_rows_i8(w, rng)
        w_off, col_off = ops.weight_offset_image(torch.from_numpy(w).to(DEV))
        rw, rc = OFF.weight_image(w)
        assert np.array_equal(w_off.cpu().numpy(), rw), (N, K)
        assert np.array_equal(col_off.cpu().numpy(), rc), (N, K)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("mode", ["per-tensor-round", "per-tensor-div", "per-token"])
def test_quantize_act_off_is_quantize_act_plus_row_offsets(dt, mode):
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(7)
    for (M, K) in ((1, 128), (9, 1024), (67, 4096), (5, 11008), (7, 14336), (6, 13312), (3, 20480), (2, 40960)):   # (14336 / 13312: 28 vectors per lane in the 2-byte per-token form)
        if dt == "f32" and K > 20480:
            continue
        x = rng.standard_normal((M, K)).astype(np.float32) * (1.4 if mode != "per-tensor-div" else 0.05)
        x[:, rng.random(K) < 0.01] *= 20.0
        if M > 2:
            x[1, 3] = 500.0; x[2, 5] = -500.0        # rows that clamp / carry the row maximum on either side
        if M > 4:
            x[4] = 0.0
        xt = torch.from_numpy(O.round_to(x, dt)).to(DEV).to(TDT[dt])
        qs = 0.0371 if mode == "per-tensor-div" else 1.0
        xq, s_row = ops.quantize_act(xt, mode, qs)
        xo, s_row2, row_off = ops.quantize_act_off(xt, mode, qs)
        rx, rr = OFF.act_image(xq.cpu().numpy())
        assert np.array_equal(xo.cpu().numpy(), rx), (M, K)
        assert np.array_equal(row_off.cpu().numpy(), rr), (M, K)
        if mode == "per-token":
            assert torch.equal(s_row, s_row2)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("mode", ["per-tensor-round", "per-tensor-div", "per-token"])
def test_two_rows_per_wave_quantisers_against_the_oracle(dt, mode):
    """>= 4096 rows of 4 / 6 / 8 / 10 / 12 sixty-four-lane vectors take the two-rows-per-wave kernel (second row's loads in flight behind the first row's arithmetic,
    waits counted by hand): plain and image outputs against the oracle, with a row count that leaves the last waves without a second row and row lengths whose
    last vector is partial (lanes past the row re-read its last vector)."""
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(11)
    vec = 4 if dt == "f32" else 8
    for (M, K) in ((4096, 8 * 64 * vec), (4101, 8 * 64 * vec - vec), (4097, 4 * 64 * vec), (4099, 6 * 64 * vec - 17 * vec), (4096, 10 * 64 * vec), (4103, 12 * 64 * vec - 63 * vec)):
        x = rng.standard_normal((M, K)).astype(np.float32) * (1.4 if mode != "per-tensor-div" else 0.05)
        x[:, rng.random(K) < 0.01] *= 20.0
        x[1, 3] = 500.0; x[2, 5] = -500.0; x[4] = 0.0
        x[M - 1, K - 1] = 77.0; x[M // 2, 0] = -91.0
        x = O.round_to(x, dt)
        xt = torch.from_numpy(x).to(DEV).to(TDT[dt])
        qs = 0.0371 if mode == "per-tensor-div" else 1.0
        if mode == "per-token":
            rq, rs = O.act_quant_per_token(x, dt)
        elif mode == "per-tensor-round":
            rq, rs = O.act_quant_round(x, dt), None
        else:
            rq, rs = O.act_quant_div(x, dt, qs), None
        xq, s_row = ops.quantize_act(xt, mode, qs)
        assert np.array_equal(xq.cpu().numpy(), rq), (M, K)
        if rs is not None:
            assert np.array_equal(s_row.cpu().numpy(), rs.reshape(-1)), (M, K)
        xo, s_row2, row_off = ops.quantize_act_off(xt, mode, qs)
        rx, rr = OFF.act_image(rq)
        assert np.array_equal(xo.cpu().numpy(), rx), (M, K)
        assert np.array_equal(row_off.cpu().numpy(), rr), (M, K)
        if rs is not None:
            assert torch.equal(s_row, s_row2)


def _operands(rng, M, N, K, kind):
    if kind == "bench":
        x = np.clip(np.rint(rng.standard_normal((M, K)) * 1.41), -128, 127)
        oc = rng.random(K) < 0.01
        x[:, oc] = np.clip(np.rint(rng.standard_normal((M, int(oc.sum()))) * 28), -128, 127)
        w = np.clip(np.rint(rng.standard_normal((N, K)) * 22), -128, 127)
    elif kind == "uniform":
        x = rng.integers(-128, 128, (M, K))
        w = rng.integers(-128, 128, (N, K))
    else:   # extremes: every accumulator at the int32 edge of what int8 x int8 can reach
        x = np.full((M, K), -128)
        w = np.full((N, K), -128)
        w[1::2] = 127
        x[1::3] = 127
    return _edge_rows_i8(x.astype(np.int8), rng), _edge_rows_i8(w.astype(np.int8), rng)


@pytest.mark.parametrize("kind", ["bench", "uniform", "extremes"])
def test_linear_off_equals_plain_linear_bit_for_bit(kind):
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(11)
    shapes = [(256, 256, 128), (300, 260, 256), (1, 4, 128), (513, 1028, 1024), (2304, 4096, 256), (700, 516, 4096)]
    if kind == "extremes":
        shapes.append((256, 256, 65536))   # |acc| = 2^30: start values and sums wrap
    for si, (M, N, K) in enumerate(shapes):
        x, w = _operands(rng, M, N, K, kind)
        xt, wt = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
        xo, ro = OFF.act_image(x)
        w_off, col_off = ops.weight_offset_image(wt)
        s_row = torch.from_numpy(rng.random(M).astype(np.float32) * 0.01 + 1e-3).to(DEV)
        s_col = torch.from_numpy(rng.random(N).astype(np.float32) * 1e-3 + 1e-4).to(DEV)
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV)
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            for (sr, sc, b) in ((None, None, None), (s_row, None, None), (None, s_col, bias), (s_row, s_col, bias)):
                for order in ("scale_first", "acc_first"):
                    ref = ops.linear_w8a8(xt, wt, dt, 1.25e-4, sr, sc, b, order)
                    got = ops.linear_w8a8_off(torch.from_numpy(xo).to(DEV), w_off, torch.from_numpy(ro).to(DEV), col_off, dt, 1.25e-4, sr, sc, b, order)
                    assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (kind, M, N, K, dt, order)   # (bit patterns: NaN-proof, any element size)
        # and the int32 accumulators themselves, through a unit-scale fp path that is exact for small sums: compare with the oracle's integer product
        if K <= 256 and kind != "extremes":
            acc = OFF.product_from_images(xo, ro, *OFF.weight_image(w))
            assert np.array_equal(acc, O.igemm(x, w))


def test_offsets_argument_errors():
    from autosmoothquant_amd import _lib
    h = _lib.lib()
    assert h.asq_linear_w8a8_off(256, 256, 256, 7, 4, 4, 128, 1.0, None, None, None, 0, 256, 256, None) == -3      # unknown out_dtype
    assert h.asq_linear_w8a8_off(256, 256, 256, 1, 4, 4, 128, 1.0, None, None, None, 0, None, 256, None) == -1     # NULL row_off
    assert h.asq_linear_w8a8_off(256, 256, 256, 1, 4, 6, 128, 1.0, None, None, None, 0, 256, 256, None) == -2      # N % 4
    assert h.asq_linear_w8a8_off(256, 256, 256, 1, 4, 4, 131072, 1.0, None, None, None, 0, 256, 256, None) == -2   # K > 65536
    assert h.asq_offsets_supported(4096, 4096, 4096, 1) == 1 and h.asq_offsets_supported(4096, 4096, 4096, 0) == 1 and h.asq_offsets_supported(4096, 4096, 4096, 9) == 0
    assert h.asq_offsets_supported(64, 4096, 4096, 1) == 0 and h.asq_offsets_supported(4096, 4096, 131072, 1) == 0


@pytest.mark.parametrize("cls_name,aq", [("W8A8BFP32OFP32Linear", "per-tensor"), ("W8A8BFP32OFP32Linear", "per-token"),
                                         ("W8A8BFP32OFP32LinearWithQuantScale", "per-tensor"), ("W8A8BFP32OFP32LinearWithQuantScale", "per-token")])
def test_module_forward_with_and_without_images_is_identical(cls_name, aq):
    """4096 rows on a 4096 x 4096 weight: the shape the offset path exists for.  The module with images (default) == the same module with
    `offsets = False` == the oracle on sampled rows (gemm_i8_p16's own fp16 output at 4096^3 against the oracle, not against another GPU kernel)."""
    import autosmoothquant_amd.layers.nn.linear as LN
    cls = getattr(LN, cls_name)
    g = torch.Generator().manual_seed(5)
    K = N = 4096
    M = 4096
    lin = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) * 0.02)
        lin.bias.copy_(torch.randn(N, generator=g) * 0.1)
    x = torch.randn(M, K, generator=g)
    x[:, torch.rand(K, generator=g) < 0.01] *= 20.0
    input_scale = float(x.abs().max()) / 127.0
    m = cls.from_float(lin, input_scale, act_quant=aq).to(DEV)
    xin = (x / input_scale if (aq == "per-tensor" and cls_name == "W8A8BFP32OFP32Linear") else x).half().to(DEV)
    assert m.offset_image(M, torch.float16) is not None
    y_img = m(xin)
    qa = m.quantize_input(xin)
    assert qa.row_off is not None
    y_shared = m(qa)
    m.offsets = False
    assert m.offset_image(M, torch.float16) is None
    y_plain = m(xin)
    assert m.quantize_input(xin).row_off is None
    assert torch.equal(y_img, y_plain) and torch.equal(y_shared, y_plain)
    # a contiguous view that does not start on a 16-byte boundary goes through the plain quantiser (which has a scalar path), not an error
    m.offsets = True
    odd = torch.empty(M * K + 8, dtype=torch.float16, device=DEV)[1:1 + M * K].view(M, K)
    odd.copy_(xin)
    assert odd.data_ptr() % 16 != 0
    qa_odd = m.quantize_input(odd)
    assert qa_odd.row_off is None and torch.equal(m(qa_odd), y_plain) and torch.equal(m(odd), y_plain)
    rows = [i * 256 + (i * 37) % 256 for i in range(16)] + [4095]   # one row in every 256-row tile x all 4096 columns: every tile of the launch meets the oracle directly
    w = m.weight.cpu().numpy()
    xs = xin[rows].float().cpu().numpy()
    ds = float(m.dequant_scale)
    if cls_name == "W8A8BFP32OFP32Linear":
        ref = O.linear_forward(xs, "f16", w, ds, m.bias.cpu().numpy(), aq)
    else:
        ref = O.linear_with_quant_scale_forward(xs, "f16", w, ds, float(m.quant_scale) if aq == "per-tensor" else None, m.bias.cpu().numpy(), aq)
    assert np.array_equal(y_img[rows].float().cpu().numpy(), ref)


def test_image_follows_the_weight():
    """the cached image is keyed on the weight's storage and version: an in-place update or load_state_dict rebuilds it"""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    g = torch.Generator().manual_seed(9)
    m = W8A8BFP32OFP32Linear(4096, 4096, False, "per-tensor")
    m.weight = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
    m.dequant_scale = torch.tensor(1e-4)
    m = m.to(DEV)
    x = torch.randint(-3, 4, (2304, 4096), generator=g).half().to(DEV)
    y0 = m(x)
    img0 = m.offset_image(2304, torch.float16)
    assert img0 is m.offset_image(2304, torch.float16)
    ptrs, old = (img0[0].data_ptr(), img0[1].data_ptr()), img0[0].clone()
    w2 = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
    m.load_state_dict({"weight": w2, "dequant_scale": torch.tensor(1e-4)})
    y1 = m(x)
    img1 = m.offset_image(2304, torch.float16)
    # the image is rebuilt INTO the module's buffers (a captured graph keeps valid pointers), with the new content
    assert (img1[0].data_ptr(), img1[1].data_ptr()) == ptrs and not torch.equal(img1[0], old)
    m.offsets = False
    assert torch.equal(y1, m(x)) and not torch.equal(y0, y1)
    # a write torch cannot see (raw pointer): invalidate_offset_image() / refresh_offset_image() are the documented hooks
    m.offsets = True
    w3 = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8).to(DEV)
    from autosmoothquant_amd import _lib
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(ctypes.c_void_p(m.weight.data_ptr()), ctypes.c_void_p(w3.data_ptr()), ctypes.c_size_t(w3.numel()), 3) == 0   # device to device, behind torch's back
    stale = m(x)
    assert torch.equal(stale, y1) or True   # (whatever the stale image gives; the point is the next two lines)
    assert m.refresh_offset_image() and (m.offset_image(2304, torch.float16)[0].data_ptr(), m.offset_image(2304, torch.float16)[1].data_ptr()) == ptrs
    y3 = m(x)
    m.offsets = False
    assert torch.equal(y3, m(x)) and not torch.equal(y3, y1)


def test_graph_on_an_image_follows_a_weight_update_after_refresh():
    """a hipGraph captured while an image was cached holds the image's buffers.  After an in-place weight update,
    refresh_offset_image() rebuilds into THOSE buffers, so the replay multiplies the new weights (it used to multiply the stale image, then freed memory)."""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    g = torch.Generator().manual_seed(12)
    m = W8A8BFP32OFP32Linear(4096, 4096, False, "per-tensor")
    m.weight = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
    m.dequant_scale = torch.tensor(1e-4)
    m = m.to(DEV)
    x = torch.randint(-3, 4, (2304, 4096), generator=g).half().to(DEV)
    assert m.build_offset_image()
    m(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        got = m(x)
    gr.replay()
    torch.cuda.synchronize()
    first = got.clone()
    m.weight.copy_(torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8))   # in place: the plain path of a graph would see it
    assert m.refresh_offset_image()
    gr.replay()
    torch.cuda.synchronize()
    m.offsets = False
    want = m(x)
    assert torch.equal(got, want) and not torch.equal(got, first)


def test_inference_tensor_weight_gets_an_image_only_on_request():
    """an inference tensor has no version counter, so an in-place write cannot be seen: such a module runs on plain operands unless the
    caller opts in with build_offset_image() -- and then load_state_dict still marks the image stale."""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    g = torch.Generator().manual_seed(13)
    with torch.inference_mode():
        m = W8A8BFP32OFP32Linear(4096, 4096, False, "per-tensor")
        m.weight = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
        m.dequant_scale = torch.tensor(1e-4)
        m = m.to(DEV)
        x = torch.randint(-3, 4, (2304, 4096), generator=g).half().to(DEV)
        assert m.offset_image(2304, torch.float16) is None
        y0 = m(x)
        assert m.build_offset_image() and m.offset_image(2304, torch.float16) is not None
        assert torch.equal(m(x), y0)
        w2 = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
        m.load_state_dict({"weight": w2, "dequant_scale": torch.tensor(1e-4)})
        y1 = m(x)
        m.offsets = False
        assert torch.equal(y1, m(x)) and not torch.equal(y1, y0)


def test_no_image_is_built_inside_a_capture():
    """a capture that meets a module without a cached image runs on the plain operands (an image built under capture would only exist after the first
    replay, and an eager call before that would multiply garbage); a module whose image exists is captured on it.  Replays equal the eager result."""
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    g = torch.Generator().manual_seed(10)
    m = W8A8BFP32OFP32Linear(4096, 4096, False, "per-tensor")
    m.weight = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
    m.dequant_scale = torch.tensor(1e-4)
    m = m.to(DEV)
    x = torch.randint(-3, 4, (2304, 4096), generator=g).half().to(DEV)
    m.offsets = False
    want = m(x)
    m.offsets = True
    torch.cuda.synchronize()
    for warm in (False, True):
        if warm:
            m(x)
            assert "_offset_cache" in m.__dict__
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            got = m(x)
        if not warm:
            assert m.__dict__.get("_offset_cache") is None
        for _ in range(2):
            got.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(got, want)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("per_token", [False, True])
def test_n1_quantisers_emit_the_offset_image(dt, per_token):
    """asq_norm_quantize_off / asq_add_norm_quantize_off / asq_silu_mul_quantize_off: the image of exactly what the plain entry points write"""
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(21)
    for (M, K) in ((3, 64), (33, 4096), (9, 8192), (130, 1024)):
        if dt == "f32" and K > 4096:
            continue
        x = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 3.0, dt)
        x[:, 5 % K] *= 12.0
        res = O.round_to(rng.standard_normal((M, K)).astype(np.float32), dt)
        w = O.round_to((rng.standard_normal(K).astype(np.float32) * 0.1 + 1.0) / 0.04, dt)
        b = O.round_to(rng.standard_normal(K).astype(np.float32) * 3.0, dt)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(TDT[dt])
        for bias in (None, b):
            xq, s = ops.norm_quantize(t(x), t(w), None if bias is None else t(bias), 1e-5, per_token)
            xo, s2, ro = ops.norm_quantize(t(x), t(w), None if bias is None else t(bias), 1e-5, per_token, offsets=True)
            rx, rr = OFF.act_image(xq.cpu().numpy())
            assert np.array_equal(xo.cpu().numpy(), rx) and np.array_equal(ro.cpu().numpy(), rr), ("norm", M, K)
            assert (s is None and s2 is None) or torch.equal(s, s2)
            h, xq, s = ops.add_norm_quantize(t(x), t(res), t(w), None if bias is None else t(bias), 1e-5, per_token)
            h2, xo, s2, ro = ops.add_norm_quantize(t(x), t(res), t(w), None if bias is None else t(bias), 1e-5, per_token, offsets=True)
            rx, rr = OFF.act_image(xq.cpu().numpy())
            assert torch.equal(h, h2) and np.array_equal(xo.cpu().numpy(), rx) and np.array_equal(ro.cpu().numpy(), rr), ("add_norm", M, K)
        g = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 2.0, dt)
        u = O.round_to(rng.standard_normal((M, K)).astype(np.float32) * 2.0, dt)
        for fast in (False, True):
            xq, s = ops.silu_mul_quantize(t(g), t(u), per_token, 0.05, fast)
            xo, s2, ro = ops.silu_mul_quantize(t(g), t(u), per_token, 0.05, fast, offsets=True)
            rx, rr = OFF.act_image(xq.cpu().numpy())
            assert np.array_equal(xo.cpu().numpy(), rx) and np.array_equal(ro.cpu().numpy(), rr), ("silu", M, K, fast)


def test_fused_llama_layer_with_and_without_images_is_identical():
    """the N1-fused layer at a prefill size (the norms and SiLU*up emit offset images, q/k/v/gate/up/down consume them) == the same layer with offsets off"""
    from autosmoothquant_amd import harness as H
    import autosmoothquant_amd.layers.nn.linear as LN
    torch.manual_seed(0)
    layer = H.init_llama_layer(H.LlamaLayer(hidden=1024, inter=2816, heads=8), seed=3).to(DEV)
    x = torch.randn(2, 2048, 1024, device=DEV)
    scales = H.calibrate(layer.float(), x[:, :128].float())
    q = H.to_w8a8(layer.half(), scales, both=True)
    q.use_fused = True
    xin = x.half()
    y_img = q(xin)
    seen = [m for m in q.modules() if isinstance(m, LN._W8A8Base) and m.__dict__.get("_offset_cache") is not None]
    for m in q.modules():
        if isinstance(m, LN._W8A8Base):
            m.offsets = False
    y_plain = q(xin)
    assert torch.equal(y_img, y_plain)
    assert len(seen) > 0   # (4096 rows on 1024-wide weights: 64 x 4 = 64 tiles of 256 x 256 for the 1024-wide projections, 176 for gate / up of 256 x 256 -- whether the dispatcher takes the 256 x 256 kernel decides; asserted so the test cannot pass vacuously)


@pytest.mark.parametrize("counts,per_token,N,K", [([300, 0, 129, 1, 700, 256], False, 512, 256), ([300, 0, 129, 1, 700, 256], True, 260, 1024),
                                                  ([1100, 900, 1300, 800, 1000, 1092, 1000, 1000], True, 4096, 2048)])
def test_grouped_launch_on_images_equals_plain_grouped_launch(counts, per_token, N, K):
    """asq_linear_w8a8_grouped_off == asq_linear_w8a8_grouped bit for bit: ragged groups (empty, one row, half tiles), per-group biases, and -- third case, 8192
    rows on 8 groups with the workspace -- the K-split tail pieces, whose partial sums get the correction exactly once."""
    from autosmoothquant_amd import ops
    rng = np.random.default_rng(31)
    G, M = len(counts), sum(counts)
    x = np.clip(np.rint(rng.standard_normal((M, K)) * 2.5), -128, 127).astype(np.int8)
    x[:: 7, 3] = 127
    x[3:: 11, 5] = -128
    w = np.clip(np.rint(rng.standard_normal((G, N, K)) * 22), -128, 127).astype(np.int8)
    xt, wt = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=DEV)
    sg = torch.from_numpy(rng.random(G).astype(np.float32) * 1e-3 + 1e-4).to(DEV)
    s_row = torch.from_numpy(rng.random(M).astype(np.float32) * 0.01 + 1e-3).to(DEV) if per_token else None
    xo, ro = OFF.act_image(x)
    img, col = ops.weight_offset_image(wt.view(G * N, K))
    rw, rc = OFF.weight_image(w.reshape(G * N, K))
    assert np.array_equal(img.cpu().numpy(), rw) and np.array_equal(col.cpu().numpy(), rc)
    for bias in (None, torch.from_numpy(rng.standard_normal((G, N)).astype(np.float32)).to(DEV)):
        for dt in (torch.float16, torch.bfloat16):
            ref = ops.linear_w8a8_grouped(xt, wt, offs, sg, dt, s_row, bias)
            got = ops.linear_w8a8_grouped_off(torch.from_numpy(xo).to(DEV), img.view(G, N, K), torch.from_numpy(ro).to(DEV), col.view(G, N, 2), offs, sg, dt, s_row, bias)
            assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (counts, per_token, N, K, dt, bias is not None)
