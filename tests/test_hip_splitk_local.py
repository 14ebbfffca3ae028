"""K splits of the 128 x 128 kernel reduced INSIDE the launch (gemm_i8_p8q2<Epi, true>, asq_gemm_p8q2.h; round 5: write-through register images, one
ticket per tile, the last arriver adds the others and runs the caller's epilogue) and the former form (ASQ_SPLITK_FIX=0: int32 slab launch + reduce launch):
every split count against the oracle's exact integer GEMM and its epilogues, ragged shapes, repeated launches on one workspace, a hipGraph replay, tickets
back at zero.  The env switches are read once per process, hence the child processes."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import detrng
from oracle import w8a8 as O
from autosmoothquant_amd import ops, _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
hdr = lib.asq_workspace_header_bytes()
for (M, N, K) in %s:
    x, w = detrng.int8_uniform(260, M, (M, K)), detrng.int8_uniform(261, N, (N, K))
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    acc = O.igemm(x, w)
    n = lib.asq_gemm_workspace_bytes(M, N, K)
    assert n > hdr, ("the forced split needs scratch", M, N, K, n)
    ws = torch.full((n,), 0x5A, dtype=torch.uint8, device=dev)
    L.check(lib.asq_workspace_init(ws.data_ptr(), n, st), "init")
    for rep in range(3):                                     # same workspace, three launches: the tickets must come back to zero each time
        out = torch.full((M, N), -7, dtype=torch.int32, device=dev)
        L.check(lib.asq_gemm_i8_i32(xd.data_ptr(), wd.data_ptr(), out.data_ptr(), M, N, K, ws.data_ptr(), n, st), "i32")
        assert np.array_equal(out.cpu().numpy(), acc), ("i32", M, N, K, rep)
    assert int(ws[16:hdr].view(torch.int32).abs().max()) == 0, "tickets not back at zero"
    bias = detrng.normal(262, N, (N,)).astype(np.float32)
    s_row = (np.abs(detrng.normal(263, M, (M,))) * 0.01 + 1e-3).astype(np.float32)
    s_col = (np.abs(detrng.normal(264, N, (N,))) * 1e-3 + 1e-4).astype(np.float32)
    y = ops.linear_w8a8(xd, wd, torch.float16, 2e-3, torch.from_numpy(s_row).to(dev), None, torch.from_numpy(bias).to(dev))
    assert np.array_equal(y.float().cpu().numpy(), O.dequant_epilogue(acc, np.float32(2e-3), s_row, bias, "f16")), ("f16 row + bias", M, N, K)
    y = ops.linear_w8a8(xd, wd, torch.bfloat16, 3e-3, None, None, None)
    assert np.array_equal(y.float().cpu().numpy(), O.dequant_epilogue(acc, np.float32(3e-3), None, None, "bf16")), ("bf16", M, N, K)
    y = ops.linear_w8a8(xd, wd, torch.float32, 1.0, None, torch.from_numpy(s_col).to(dev), torch.from_numpy(bias).to(dev))
    assert np.array_equal(y.cpu().numpy(), O.dequant_epilogue(acc, s_col, None, bias, "f32")), ("f32 per-channel", M, N, K)
# hipGraph: capture one split launch, replay it three times against the eager result
M, N, K = %s
xd = torch.from_numpy(detrng.int8_uniform(265, M, (M, K))).to(dev)
wd = torch.from_numpy(detrng.int8_uniform(266, N, (N, K))).to(dev)
want = ops.linear_w8a8(xd, wd, torch.float16, 1e-3, None, None, None).clone()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    ops.linear_w8a8(xd, wd, torch.float16, 1e-3, None, None, None)
torch.cuda.current_stream().wait_stream(cap)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    got = ops.linear_w8a8(xd, wd, torch.float16, 1e-3, None, None, None)
for _ in range(3):
    got.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
print("ok")
"""

SHAPES = "[(256, 512, 2048), (300, 520, 1536), (128, 1024, 4096), (513, 640, 1024), (64, 256, 2560)]"


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("ksplit", [2, 3, 4, 8])
@pytest.mark.parametrize("kern", ["p8q", "p8h"])
def test_in_launch_split_k_every_protocol(kern, mode, ksplit):
    """kern: the 128 x 128 kernel (gemm_i8_p8q2<Epi, true>: in-launch for mode 1) and the 128 x 256 one (slab launch + reduce launch in both modes: its in-launch form
    was measured and dropped, profiles/r5_p8h_splitk_in_launch_dropped.txt)."""
    code = CODE % (ROOT, os.path.join(ROOT, "tests"), SHAPES, "(256, 1024, 4096)")
    env = dict(os.environ, ASQ_GEMM_KERNEL=kern, ASQ_KSPLIT=str(ksplit), ASQ_SPLITK_FIX=str(mode))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_dispatcher_chooses_the_split_by_itself():
    """No forcing: the shapes the cost model splits (few tiles, long K) against the oracle, through the module-level op (ops keeps the workspace)."""
    code = CODE % (ROOT, os.path.join(ROOT, "tests"), "[(256, 4096, 4096), (384, 4096, 4096), (512, 4096, 4096), (256, 4096, 11008), (256, 5120, 20480), (384, 4096, 14336)]", "(256, 4096, 4096)")
    code = code.replace('assert n > hdr, ("the forced split needs scratch", M, N, K, n)', 'n = max(n, hdr); print("ws", M, N, K, n, L.lib().asq_gemm_kernel_name(M, N, K).decode())')
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
