#!/usr/bin/env python3
"""Dev tool (round 4): the dispatcher's choice and its time for M = 64 ... 3072 at LLaMA-7B's three (N, K), cold weights (4 rotating buffers), next to a simple floor:
max(ops at 60 % of the INT8 peak, bytes at 5.5 TB/s) + 2.5 us of launch.  usage: python tools/midsize_sweep.py [--ms 64,128,...] [--env KEY=VAL,...]"""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="64,128,256,384,512,768,1024,1536,2048,3072")
ap.add_argument("--nks", default="4096x4096,11008x4096,4096x11008")
ap.add_argument("--env", default="")
ap.add_argument("--batch", type=int, default=20)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rotate", type=int, default=4)
args = ap.parse_args()
for kv in args.env.split(","):
    if kv:
        k, v = kv.split("=", 1)
        os.environ[k] = v
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
for nk in args.nks.split(","):
    N, K = map(int, nk.split("x"))
    ws = [(torch.randn(N, K, device=dev, generator=g) * 22).round().clamp(-128, 127).to(torch.int8) for _ in range(args.rotate)]
    for M in map(int, args.ms.split(",")):
        x = (torch.randn(M, K, device=dev, generator=g) * 2.9).round().clamp(-128, 127).to(torch.int8)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        i = 0
        def run():
            global i
            for _ in range(args.batch):
                ops.linear_w8a8(x, ws[i % len(ws)], torch.float16, 1e-4, out=out)
                i += 1
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) / args.batch * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        ops_, byts = 2.0 * M * N * K, N * K + M * K + 2 * M * N
        floor = max(ops_ / (0.6 * 5033e6), byts / 5.5e6) + 2.5
        kn = ops.gemm_kernel_name(M, N, K)
        tile = {"p8q": (128, 128), "p8h": (128, 256), "p16": (256, 256), "p8": (256, 256)}.get(kn.split("+")[0])
        # bytes the launch moves L2 -> LDS (every tile streams its X and W panels): the path tops out near 38 B/clk per CU and 17-18 TB/s chip-wide
        lds_tbs = f"{-(-M // tile[0]) * -(-N // tile[1]) * K * (tile[0] + tile[1]) / max(us - 3.0, 0.1) / 1e6:5.1f} TB/s L2->LDS" if tile else ""
        print(f"{M:5d} x {N:5d} x {K:5d}  {kn:8s} {us:7.1f} us  {ops_ / us / 1e6:6.0f} TOPS {ops_ / us / 50.33e6:5.1f} %  {byts / us / 1e3:6.0f} GB/s   floor ~{floor:6.1f} us  x{us / floor:.2f}  {lds_tbs}", flush=True)
