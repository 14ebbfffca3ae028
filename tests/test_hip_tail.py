"""GPU (-m gpu): interior-tile launches of the 256 x 256 kernel against the SAME product from a launch with an edge tile row (one extra activation row; rows of a
linear are independent).  Written for gemm_i8_p16t (asq_gemm_p16t.h, "THE TAIL": the single-round kernel whose last two K-tiles run m-half first so that half of every
block's epilogue leaves under its last matrix work) -- on a `-DASQ_P16_TAIL=1` build the interior launch runs on it and the edge launch on the plain-ended gemm_i8_p16,
which is how the experimental kernel was shown bit-identical before it was measured and left out (profiles/r6_tail_overlap_ab.txt).  On the default build both launches
run gemm_i8_p16 and the file pins that interior and edge-tile launches agree.  Plain operands and offset images, every epilogue operand set, both 2-byte dtypes,
K-tile counts 4 / 6 / 8 / 16 / 32 / 86, tile counts from 144 to 256, the oracle on a row of every tile, int8 extremes, back-to-back launches and hipGraph replays."""
import numpy as np
import pytest
import torch

from oracle import w8a8 as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _operands(M, N, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    w = torch.randint(-128, 128, (N, K), generator=g, device=DEV, dtype=torch.int8)
    x = torch.randn(M + 1, K, generator=g, device=DEV) * 3.0
    x[:, torch.rand(K, generator=g, device=DEV) < 0.01] *= 20.0
    return w, x


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(4096, 4096, 512), (4096, 4096, 768), (2304, 4096, 1024), (4096, 4096, 4096), (4096, 2304, 2048), (1280, 11264, 4096), (3072, 5120, 11008)])
def test_tail_equals_plain_end(dt, shape):
    from autosmoothquant_amd import ops
    M, N, K = shape
    tiles = (M // 256) * (N // 256)
    assert M % 256 == 0 and N % 256 == 0 and K % 256 == 0 and K >= 512 and 144 <= tiles <= 256
    w, x = _operands(M, N, K, M + N + K)
    xh = x.to(TDT[dt])
    rng = np.random.default_rng(7)
    s_col = torch.from_numpy(rng.uniform(1e-4, 2e-4, N).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV)
    image = ops.weight_offset_image(w)
    for mode in ("per-tensor-round", "per-token"):
        xq, s_row = ops.quantize_act(xh, mode)                    # M + 1 rows
        xo, s_row_o, row_off = ops.quantize_act_off(xh, mode)
        for (c, b) in ((None, None), (s_col, None), (None, bias), (s_col, bias)):
            ds = 1.0 if c is not None else 1.3e-4
            ref = ops.linear_w8a8(xq, w, TDT[dt], ds, s_row, c, b)[:M]                                                           # edge tile row: plain end
            got = ops.linear_w8a8(xq[:M], w, TDT[dt], ds, None if s_row is None else s_row[:M], c, b)                             # interior tiles only: TAIL
            got_img = ops.linear_w8a8_off(xo[:M], image[0], row_off[:M], image[1], TDT[dt], ds, None if s_row_o is None else s_row_o[:M], c, b)
            assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (shape, mode, c is not None, b is not None)
            assert torch.equal(ref.view(torch.int16), got_img.view(torch.int16)), (shape, mode, c is not None, b is not None, "images")
    # the oracle on a row of every 256-row tile (per-token + bias)
    rows = [i * 256 + (i * 53) % 256 for i in range(M // 256)]
    xq, s_row = ops.quantize_act(xh[:M], "per-token")
    got = ops.linear_w8a8(xq, w, TDT[dt], 1.3e-4, s_row, None, bias)[rows].float().cpu().numpy()
    ref = O.linear_forward(xh[rows].float().cpu().numpy(), dt, w.cpu().numpy(), 1.3e-4, bias.cpu().numpy(), "per-token")
    assert np.array_equal(got, ref)


def test_tail_int8_extremes_and_wraparound():
    """every operand at -128 / 127 (the largest accumulators of the int8 domain) through the TAIL kernel, on plain operands and on images"""
    from autosmoothquant_amd import ops
    M, N, K = 2304, 4096, 2048
    g = torch.Generator(device=DEV).manual_seed(11)
    w = (torch.randint(0, 2, (N, K), generator=g, device=DEV, dtype=torch.int8) * 255 - 128).to(torch.int8)
    xq = (torch.randint(0, 2, (M, K), generator=g, device=DEV, dtype=torch.int8) * 255 - 128).to(torch.int8)
    acc = torch._int_mm(xq, w.t().contiguous()) if hasattr(torch, "_int_mm") else (xq.int() @ w.int().t())
    want = (acc.float() * 1e-6).half()
    got = ops.linear_w8a8(xq, w, torch.float16, 1e-6)
    assert torch.equal(want.view(torch.int16), got.view(torch.int16))


def test_tail_launch_twice_and_under_a_graph():
    """no state is left behind: back-to-back launches and hipGraph replays give the same bits"""
    from autosmoothquant_amd import ops
    M, N, K = 4096, 4096, 1024
    w, x = _operands(M, N, K, 5)
    xq, _ = ops.quantize_act(x[:M].half(), "per-tensor-round")
    out = torch.empty((M, N), dtype=torch.float16, device=DEV)
    a = ops.linear_w8a8(xq, w, torch.float16, 1e-4).clone()
    ops.linear_w8a8(xq, w, torch.float16, 1e-4, out=out)
    assert torch.equal(a, out)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        ops.linear_w8a8(xq, w, torch.float16, 1e-4, out=out)
    for _ in range(3):
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(a, out)
