#!/usr/bin/env python3
"""Device time of one decode-sized module forward (hipGraph replay of 20 back-to-back forwards over rotating weights), fused prologue vs
two kernels.  Run twice: ASQ_FUSE_PROLOGUE_BYTES=0 python tools/fused_prologue_ab.py ; python tools/fused_prologue_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
tag = "two-kernel" if os.environ.get("ASQ_FUSE_PROLOGUE_BYTES") == "0" else "fused"
for (N, K) in ((4096, 4096), (11008, 4096)):
    ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(20)]   # > 256 MiB: weights come from HBM
    for dt in (torch.float16, torch.float32):
        for M in (1, 4, 16, 32):
            x = (torch.randn(M, K, device=dev) * 30).to(dt)
            def run():
                for w in ws:
                    ops.linear_w8a8_forward(x, w, "per-tensor-round", 1.0, 1e-3)
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): g.replay()
            b.record(); b.synchronize()
            print(f"{tag:10s} N={N:5d} K={K} {str(dt)[6:]:8s} M={M:2d}: {a.elapsed_time(b) / 200 * 1e3:6.2f} us per forward")
