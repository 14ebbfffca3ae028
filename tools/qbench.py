#!/usr/bin/env python3
"""Dev tool: time the activation quantisers (batched launches)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
for (M, K) in [(4096, 4096), (4096, 11008), (65536, 4096), (32, 4096)]:
    for dt in (torch.float16, torch.float32):
        x = (torch.randn(M, K, device=dev) * 30).to(dt)
        for mode in ("per-token", "per-tensor-round", "per-tensor-div"):
            for _ in range(3): ops.quantize_act(x, mode, 0.7)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): ops.quantize_act(x, mode, 0.7)
            b.record(); b.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            by = M * K * (x.element_size() + 1)
            print(json.dumps({"M": M, "K": K, "dtype": str(dt), "mode": mode, "us": round(us, 1), "TBps": round(by / us / 1e6, 2)}), flush=True)
