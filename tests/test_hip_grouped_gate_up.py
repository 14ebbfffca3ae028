"""GPU (-m gpu): Mixtral's w1 || w3 of all experts as ONE grouped launch with the SiLU(w1 x) * (w3 x) epilogue (asq_linear_w8a8_grouped_gate_up on the grouped
256 x 256 kernel; reference models/mixtral.py:99-101,142-145) against the composition it replaces -- linear_w8a8_grouped (w1), linear_w8a8_grouped (w3), the SiLU * up
of asq_silu_mul_quantize -- through the consumer's quantiser (per-token as Mixtral's w2, and per-tensor), both SiLU forms, plain operands and offset images, ragged
groups (empty, < 128 rows, not a multiple of 256), repeated launches on one workspace, and at Mixtral size with the bench's routing (the tail round's in-launch K
split runs there).  Bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _check(counts, F_, K, dt, seed):
    from autosmoothquant_amd import ops
    G, M = len(counts), sum(counts)
    g = torch.Generator(device=DEV).manual_seed(seed)
    w1 = torch.randint(-128, 128, (G, F_, K), generator=g, device=DEV, dtype=torch.int8)
    w3 = torch.randint(-128, 128, (G, F_, K), generator=g, device=DEV, dtype=torch.int8)
    x = (torch.randn(M, K, generator=g, device=DEV) * 3.0)
    x[:, torch.rand(K, generator=g, device=DEV) < 0.01] *= 20.0
    x = x.to(TDT[dt])
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=DEV)
    s1 = (torch.rand(G, generator=g, device=DEV) * 2e-4 + 1e-4)
    s3 = (torch.rand(G, generator=g, device=DEV) * 2e-4 + 1e-4)
    assert ops.grouped_gate_up_supported(M, F_, K, TDT[dt])
    w_gu = ops.interleave_gate_up_stack(w1, w3)
    w_img, col = ops.weight_offset_image(w_gu.view(G * 2 * F_, K))
    xq, _ = ops.quantize_act(x, "per-tensor-round")
    xo, _, row_off = ops.quantize_act_off(x, "per-tensor-round")
    gate = ops.linear_w8a8_grouped(xq, w1, offs, s1, TDT[dt])
    up = ops.linear_w8a8_grouped(xq, w3, offs, s3, TDT[dt])
    for fast in (False, True):
        a = ops.linear_w8a8_grouped_gate_up(xq, w_gu, offs, s1, s3, TDT[dt], fast)
        for rep in range(2):   # (the tail split's tickets must be back at zero)
            a_img = ops.linear_w8a8_grouped_gate_up(xo, w_img.view(G, 2 * F_, K), offs, s1, s3, TDT[dt], fast, row_off, col)
            assert torch.equal(a.view(torch.int16), a_img.view(torch.int16)), (counts, dt, fast, "images", rep)
        for per_token, qs in ((True, 1.0), (False, 0.013)):
            want = ops.silu_mul_quantize(gate, up, per_token, qs, fast=fast)
            got = ops.quantize_act(a, "per-token" if per_token else "per-tensor-div", qs)
            assert torch.equal(got[0], want[0]), (counts, dt, fast, per_token)
            if per_token:
                assert torch.equal(got[1], want[1])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("counts", [[300, 0, 70, 513, 129], [256, 256], [1, 127, 128, 255, 257, 640]])
def test_grouped_gate_up_equals_the_composition(dt, counts):
    _check(counts, 384, 512, dt, 7 + len(counts))


def test_grouped_gate_up_at_mixtral_size_with_the_bench_routing():
    p = torch.ones(8)
    counts = torch.bincount(torch.multinomial(p / p.sum(), 8192, replacement=True, generator=torch.Generator().manual_seed(1234)), minlength=8).tolist()
    _check(counts, 14336, 4096, "f16", 99)
    _check([4096, 2048, 1024, 512, 256, 256, 0, 0], 14336, 4096, "f16", 100)


def test_refusals():
    from autosmoothquant_amd import ops
    assert not ops.grouped_gate_up_supported(512, 200, 512, torch.float16)     # F % 128
    assert not ops.grouped_gate_up_supported(512, 256, 500, torch.float16)     # K % 128
    assert not ops.grouped_gate_up_supported(512, 256, 512, torch.float32)
    xq = torch.zeros(8, 512, dtype=torch.int8, device=DEV)
    w = torch.zeros(2, 400, 512, dtype=torch.int8, device=DEV)
    offs = torch.tensor([0, 4, 8], dtype=torch.int32, device=DEV)
    s = torch.ones(2, device=DEV)
    with pytest.raises(ValueError):                                            # F = 200: ASQ_ERR_DIM
        ops.linear_w8a8_grouped_gate_up(xq, w, offs, s, s, torch.float16)


def test_mixtral_layer_opt_in_equals_the_fused_silu_composition():
    """harness.MixtralLayer.moe with fuse_gate_up: the grouped gate || up launch gives the expert outputs of grouped w1, grouped w3 + ops.silu_mul_quantize (same SiLU
    form) bit for bit, on offset images and on plain operands; against the default (torch SiLU) composition it moves a few activations across int8 rounding boundaries."""
    from autosmoothquant_amd import harness, ops
    torch.manual_seed(5)
    fl = harness.MixtralLayer(256, 384, 4, 2, experts=4, top_k=2)
    with torch.no_grad():
        for p in fl.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape) * 0.05)
    lay = harness.to_w8a8_mixtral(fl, {"attn_in": 0.05, "o_in": 0.05, "mlp_in": 0.05, "down_in": [0.05, 0.06, 0.07, 0.08]}).to(DEV).half()
    lay.stack_experts()
    x = (torch.randn(2, 320, 256, device=DEV) * 2).half()
    with torch.no_grad():
        for offsets in (True, False):
            lay.offsets = offsets
            lay.fuse_gate_up = False
            base = lay.moe(x)
            for fast in (False, True):
                lay.fuse_gate_up, lay.fast_silu = True, fast
                got = lay.moe(x)
                # the composition with the fused SiLU kernel, spelled out
                x2 = x.reshape(-1, 256)
                tok, wts, counts = lay.route(x2)
                xs = x2[tok].contiguous()
                offs = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
                mode, qs = lay.experts[0].w1._input_mode()
                xq, _ = ops.quantize_act(xs, mode, qs)
                h1 = ops.linear_w8a8_grouped(xq, lay._w1_stack, offs, lay._w1_scale, torch.float16)
                h3 = ops.linear_w8a8_grouped(xq, lay._w3_stack, offs, lay._w3_scale, torch.float16)
                aq, arow = ops.silu_mul_quantize(h1, h3, True, 1.0, fast=fast)
                y = ops.linear_w8a8_grouped(aq, lay._w2_stack, offs, lay._w2_scale, torch.float16, arow)
                want = torch.zeros_like(x2)
                want.index_add_(0, tok, y * wts[:, None])
                assert torch.equal(got.reshape(-1, 256), want), (offsets, fast)
                assert float((got - base).abs().max()) <= 0.05 * float(base.abs().max()) + 1e-3


def test_mixtral_graph_replay_follows_a_weight_update_after_refresh():
    """MixtralLayer's w1 || w3 operand (and its image) is rebuilt INTO its buffers when the stacks change, and refresh_offset_images() covers it: a hipGraph
    captured on the grouped gate || up launch replays the new weights after the refresh, from the same addresses."""
    from autosmoothquant_amd import harness, ops
    torch.manual_seed(6)
    fl = harness.MixtralLayer(256, 384, 4, 2, experts=4, top_k=2)
    with torch.no_grad():
        for p in fl.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape) * 0.05)
    lay = harness.to_w8a8_mixtral(fl, {"attn_in": 0.05, "o_in": 0.05, "mlp_in": 0.05, "down_in": [0.05, 0.06, 0.07, 0.08]}).to(DEV).half()
    lay.stack_experts()
    lay.fuse_gate_up = True
    E, F_, K = lay._w1_stack.shape
    counts = [512, 0, 300, 212]
    M = sum(counts)
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=DEV)
    x = (torch.randn(M, K, device=DEV) * 2).half()
    assert ops.grouped_gate_up_supported(M, F_, K, torch.float16)
    xo, _, row_off = ops.quantize_act_off(x, "per-tensor-round")
    xq, _ = ops.quantize_act(x, "per-tensor-round")
    with torch.no_grad():
        w13, img = lay._w13_operand(True)
        assert w13 is not None and img is not None
        ptrs = (w13.data_ptr(), img[0].data_ptr(), img[1].data_ptr())

        def launch():
            return ops.linear_w8a8_grouped_gate_up(xo, img[0], offs, lay._w1_scale, lay._w3_scale, torch.float16, True, row_off, img[1])
        old = launch().clone()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            launch()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out_g = launch()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_g, old)
        for e in lay.experts:        # in-place writes into the per-expert views of the stacks
            e.w1.weight.copy_(torch.randint(-128, 128, e.w1.weight.shape, device=DEV, dtype=torch.int8))
            e.w3.weight.copy_(torch.randint(-128, 128, e.w3.weight.shape, device=DEV, dtype=torch.int8))
        assert harness.refresh_derived_operands(lay) >= 1
        w13b, imgb = lay._w13_operand(True)
        assert ptrs == (w13b.data_ptr(), imgb[0].data_ptr(), imgb[1].data_ptr())      # same buffers: the graph's pointers stay valid
        gr.replay()
        torch.cuda.synchronize()
        want = ops.linear_w8a8_grouped_gate_up(xq, ops.interleave_gate_up_stack(lay._w1_stack, lay._w3_stack), offs, lay._w1_scale, lay._w3_scale, torch.float16, True)
        assert torch.equal(out_g, want) and not torch.equal(want, old)


def test_mixtral_stacks_without_a_version_counter_get_images_only_on_request():
    """a MixtralLayer built under torch.inference_mode() has stacks without a version counter -- writes into them cannot be seen, so by default the grouped
    launches run on plain operands and the grouped gate || up launch is off; MixtralLayer.build_offset_images() is the opt-in (the caller then owns the refresh), after
    which the same forward runs on images + the fused launch and gives the same expert outputs as the fused-SiLU composition on plain operands; GateUpSiLU.build() likewise."""
    from autosmoothquant_amd import harness
    from autosmoothquant_amd.layers.nn.fused import GateUpSiLU
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    torch.manual_seed(8)
    fl = harness.MixtralLayer(256, 384, 4, 2, experts=4, top_k=2)
    with torch.no_grad():
        for p in fl.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape) * 0.05)
    x = (torch.randn(2, 320, 256) * 2).half()
    with torch.inference_mode():
        lay = harness.to_w8a8_mixtral(fl, {"attn_in": 0.05, "o_in": 0.05, "mlp_in": 0.05, "down_in": [0.05, 0.06, 0.07, 0.08]}).to(DEV).half()
        lay.stack_experts()
        lay.fuse_gate_up, lay.fast_silu = True, True
        xd = x.to(DEV)
        assert lay._stack_image("w1") is None and lay._w13_operand(True) == (None, None)
        y_plain = lay.moe(xd)
        assert lay.build_offset_images()
        assert lay._stack_image("w1") is not None and lay._w13_operand(True)[1] is not None
        y_img = lay.moe(xd)
        lay.fuse_gate_up = False                       # grouped w1, w3 + the fused SiLU quantiser on images: the composition the gate || up launch replaces
        y_comp = lay.moe(xd)
        assert torch.equal(y_img, y_comp)
        assert float((y_img.float() - y_plain.float()).abs().max()) <= 0.05 * float(y_plain.float().abs().max()) + 1e-3   # (plain path: torch SiLU, a few int8 steps apart)
        g = torch.Generator().manual_seed(3)
        mods = []
        for _ in range(2):
            m = W8A8BFP32OFP32Linear(512, 2304, False, "per-tensor")
            m.weight = torch.randint(-100, 100, (2304, 512), generator=g, dtype=torch.int8)
            m.dequant_scale = torch.tensor(2e-4)
            mods.append(m.to(DEV))
        gu = GateUpSiLU(*mods)
        assert gu._operand() is None                   # no version counter, nobody promised to refresh
        assert gu.build(4096, torch.float16) and gu._operand() is not None
