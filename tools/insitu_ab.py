#!/usr/bin/env python3
"""Why does the 4096^3 GEMM take ~62 us inside bench.py's step but ~54 us in a back-to-back loop on one buffer set?
Back-to-back launches (events around batches of 40) with (a) one weight / one output, (b) 4 rotating weights, (c) 4 rotating outputs,
(d) both, (e) both + a per-tensor quantiser launch between GEMMs (the step's real neighbour)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(M, K, device=dev, generator=g) * 3).round().clamp(-128, 127).to(torch.int8)
ws = [(torch.randn(N, K, device=dev, generator=g) * 23).round().clamp(-127, 127).to(torch.int8) for _ in range(4)]
outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(4)]
xf = (torch.randn(M, K, device=dev, generator=g) * 3).half()
def loop(rot_w, rot_o, quant, n=40):
    for i in range(n):
        if quant:
            ops.quantize_act(xf, "per-tensor-round")
        ops.linear_w8a8(x, ws[i % 4 if rot_w else 0], torch.float16, 1e-4, out=outs[i % 4 if rot_o else 0])
for name, a in (("one weight, one output", (0, 0, 0)), ("4 weights", (1, 0, 0)), ("4 outputs", (0, 1, 0)), ("4 weights + 4 outputs", (1, 1, 0)),
                ("4 weights + 4 outputs + quantiser between", (1, 1, 1)), ("one weight, one output", (0, 0, 0))):
    for _ in range(6):
        loop(*a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop(*a); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    print(f"{name:45s}: {sum(ts)/len(ts):6.2f} us per iteration (min {min(ts):.2f})")
