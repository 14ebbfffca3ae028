#!/usr/bin/env python3
"""Dev tool: this library's INT8 GEMM next to the vendor path PyTorch-ROCm exposes (torch._int_mm ->
hipBLASLt / rocBLAS), same operands, same event timing.  Prints one JSON line per shape."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
shapes = [tuple(map(int, s.split("x"))) for s in (sys.argv[1] if len(sys.argv) > 1 else "4096x4096x4096,8192x8192x8192,4096x11008x4096,4096x4096x11008,2048x4096x4096,512x4096x4096").split(",")]


def timed(fn, iters=8, batch=20, warm=100):
    """steady state: `warm` launches first (the power controller needs ~20 ms after a change of kernel), then `batch` back-to-back
    launches per event pair (a single launch per pair reads 15-20 % high for both libraries and unevenly so)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        for _ in range(batch):
            fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / batch for a, b in evs)
    return sum(ts) / len(ts) * 1e3, ts[0] * 1e3


for M, N, K in shapes:
    for dist in ("uniform", "gauss16"):
        if dist == "uniform":
            x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
            w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
        else:  # what quantised activations / weights look like: small magnitudes, many zero high bits
            x = (torch.randn((M, K), generator=g) * 16).round().clamp(-127, 127).to(torch.int8).to(dev)
            w = (torch.randn((N, K), generator=g) * 32).round().clamp(-127, 127).to(torch.int8).to(dev)
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        wt = w.t()
        ours = timed(lambda: ops.gemm_i8_i32(x, w, out))
        out16 = torch.empty((M, N), dtype=torch.float16, device=dev)
        fused = timed(lambda: ops.linear_w8a8(x, w, torch.float16, 1e-4, out=out16))   # the product path: GEMM + dequant epilogue, fp16 out
        try:
            ref = torch._int_mm(x, wt)
            same = bool(torch.equal(ref, out))
            vend = timed(lambda: torch._int_mm(x, wt))
        except Exception as e:  # noqa
            same, vend = None, (float("nan"), float("nan"))
        f = 2.0 * M * N * K / 1e6
        print(json.dumps({"shape": f"{M}x{N}x{K}", "data": dist, "asq_us": round(ours[0], 1), "asq_tops": round(f / ours[0], 0), "asq_fused_f16_us": round(fused[0], 1), "asq_fused_f16_tops": round(f / fused[0], 0),
                          "int_mm_us": round(vend[0], 1), "int_mm_tops": round(f / vend[0], 0), "equal": same}), flush=True)
