#!/usr/bin/env python3
"""Dev tool: time / cross-check one forced GEMM kernel (env ASQ_GEMM_KERNEL) in this process.
usage: ASQ_GEMM_KERNEL=p8 python tools/kbench.py [--check] [--shapes MxNxK,...]"""
import argparse, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--check", action="store_true")
ap.add_argument("--shapes", default="4096x4096x4096,8192x8192x8192,4096x11008x4096,4096x4096x11008,2048x4096x4096,1024x4096x4096,512x4096x4096,256x4096x4096")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--epi", default="f16", choices=["f16", "i32"])
ap.add_argument("--batch", type=int, default=1, help="launches per timed event pair (amortises the ~6 us event/launch floor)")
ap.add_argument("--rotate", type=int, default=1, help="number of distinct weight buffers cycled through (defeats the 256 MiB MALL)")
args = ap.parse_args()
dev = torch.device("cuda:0")
kern = os.environ.get("ASQ_GEMM_KERNEL", "default")
g = torch.Generator().manual_seed(3)
for sh in args.shapes.split(","):
    M, N, K = map(int, sh.split("x"))
    x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
    if args.check:
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        ops.gemm_i8_i32(x, w, out)
        ref = torch._int_mm(x, w.t()) if (M > 16 and M % 8 == 0 and N % 8 == 0 and K % 8 == 0) else (x.cpu().int() @ w.cpu().int().t()).to(dev)
        bad = int((out != ref).sum())
        # repeat to screen for races
        nbad = 0
        for _ in range(10):
            o2 = torch.empty_like(out)
            ops.gemm_i8_i32(x, w, o2)
            nbad += int((o2 != ref).sum())
        print(json.dumps({"kernel": kern, "shape": sh, "check_mismatch": bad, "repeat_mismatch": nbad}), flush=True)
        continue
    out = torch.empty((M, N), dtype=torch.float16 if args.epi == "f16" else torch.int32, device=dev)
    ws = [w] + [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(dev) for _ in range(args.rotate - 1)]
    state = {"i": 0}
    def run():
        for _ in range(args.batch):
            wi = ws[state["i"] % len(ws)]; state["i"] += 1
            if args.epi == "f16":
                ops.linear_w8a8(x, wi, torch.float16, 1e-4, out=out)
            else:
                ops.gemm_i8_i32(x, wi, out)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for a, b in evs:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / args.batch for a, b in evs)
    avg = sum(ts) / len(ts)
    print(json.dumps({"kernel": kern, "name": ops.gemm_kernel_name(M, N, K), "shape": sh, "epi": args.epi, "avg_us": round(avg * 1e3, 1), "min_us": round(ts[0] * 1e3, 1),
                      "tops_avg": round(2.0 * M * N * K / avg / 1e9, 1), "tops_min": round(2.0 * M * N * K / ts[0] / 1e9, 1),
                      "GBps_avg": round((N * K + M * K + 2 * M * N) / avg / 1e6, 1), "batch": args.batch, "rotate": args.rotate}), flush=True)
