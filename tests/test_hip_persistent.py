"""GPU (-m gpu): gemm_i8_p16p -- the persistent form of the 256 x 256 kernel that multi-round launches take (asq_gemm_p16p.h) -- against the SAME product computed
as row chunks small enough for single-round launches of gemm_i8_p16 (rows of a linear are independent), on plain operands and on offset images, for every
epilogue operand set, K-tile counts 2 / 4 / 32 / 86, tile counts that are and are not multiples of the 256 CUs, and against the oracle on sampled rows."""
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
    x = torch.randn(M, K, generator=g, device=DEV) * 3.0
    x[:, torch.rand(K, generator=g, device=DEV) < 0.01] *= 20.0
    return w, x


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(8192, 4096, 512), (4608, 4096, 256), (16384, 4096, 4096), (4096, 11008, 4096), (2048, 12288, 1024), (8192, 5120, 11008 // 43 * 2)])
def test_persistent_equals_single_round_chunks(dt, shape):
    from autosmoothquant_amd import ops
    M, N, K = shape
    assert (M // 256) * (N // 256) > 256 and K % 256 == 0
    w, x = _operands(M, N, K, M + N + K)
    xh = x.to(TDT[dt])
    rng = np.random.default_rng(3)
    s_col = torch.from_numpy(rng.uniform(1e-4, 2e-4, N).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV)
    chunk = 256 * max(1, 256 // (N // 256))            # rows per single-round launch: <= 256 tiles
    for mode in ("per-tensor-round", "per-token"):
        xq, s_row = ops.quantize_act(xh, mode)
        xo, s_row_o, row_off = ops.quantize_act_off(xh, mode)
        image = ops.weight_offset_image(w)
        for (c, b) in ((None, None), (s_col, None), (None, bias), (s_col, bias)):
            ds = 1.0 if c is not None else 1.3e-4
            full = ops.linear_w8a8(xq, w, TDT[dt], ds, s_row, c, b)
            full_img = ops.linear_w8a8_off(xo, image[0], row_off, image[1], TDT[dt], ds, s_row_o, c, b)
            for r0 in range(0, M, chunk):
                r1 = min(M, r0 + chunk)
                part = ops.linear_w8a8(xq[r0:r1], w, TDT[dt], ds, None if s_row is None else s_row[r0:r1], c, b)
                assert torch.equal(full[r0:r1].view(torch.int16), part.view(torch.int16)), (shape, mode, c is not None, b is not None, r0)
            assert torch.equal(full.view(torch.int16), full_img.view(torch.int16)), (shape, mode, "images")
    # the oracle on a row of every 256-row tile (per-token + bias)
    rows = [i * 256 + (i * 53) % 256 for i in range(M // 256)]
    got = ops.linear_w8a8(xq, w, TDT[dt], 1.3e-4, s_row, None, bias)[rows].float().cpu().numpy()
    ref = O.linear_forward(xh[rows].float().cpu().numpy(), dt, w.cpu().numpy(), 1.3e-4, bias.cpu().numpy(), "per-token")
    assert np.array_equal(got, ref)


def test_persistent_launch_twice_and_under_a_graph():
    """the kernel leaves no state behind (no workspace, no tickets): back-to-back launches and hipGraph replays give the same bits"""
    from autosmoothquant_amd import ops
    M, N, K = 8192, 4096, 1024
    w, x = _operands(M, N, K, 5)
    xq, _ = ops.quantize_act(x.half(), "per-tensor-round")
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
