#!/usr/bin/env python3
"""Dev tool: time the fp8 quantisers and asq_linear_fp8 (event pairs around batches of launches)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops

dev = torch.device("cuda:0")


def timed(fn, batch=10, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) / batch * 1e3)
    return sum(ts) / len(ts)


for (M, N, K) in [(4096, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 14336), (1024, 4096, 4096), (256, 4096, 4096), (32, 4096, 4096)]:
    x = torch.randn(M, K, device=dev, dtype=torch.float16) * 3
    w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
    for mode in ("per-token", "per-tensor"):
        us = timed(lambda: ops.quantize_act_fp8(x, mode))
        print(json.dumps({"op": "quantize_act_fp8", "mode": mode, "M": M, "K": K, "us": round(us, 1), "TBps": round(M * K * 3 / us / 1e6, 2)}), flush=True)
    xq, sc = ops.quantize_act_fp8(x, "per-token")
    us = timed(lambda: ops.linear_fp8(xq, sc.view(-1, 1), w, 0.01, None, torch.float16))
    print(json.dumps({"op": "linear_fp8", "shape": f"{M}x{N}x{K}", "us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 0)}), flush=True)
