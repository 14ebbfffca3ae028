#!/usr/bin/env python3
"""Dev tool: random shapes through the dispatcher (every tiled kernel, tail peel, split-K, ragged M / N) against torch._int_mm, exact, each shape
launched several times (races show up as run-to-run differences).  usage: python tools/fuzz_gemm.py [n_shapes] [seed]"""
import os, sys, random, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops, _lib

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rnd = random.Random(seed)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(seed)
lib = _lib.lib()
seen, bad = collections.Counter(), 0
for i in range(n):
    cls = rnd.random()
    if cls < 0.25:     # tail-peel region: a few tiles over a multiple of 256
        tn = rnd.choice([16, 20, 32, 43, 48, 56]); tm = rnd.choice([t for t in range(2, 40) if 0 < (t * tn) % 256 <= 96 and t * tn > 256] or [6])
        M, N = tm * 256 - rnd.choice([0, 0, 7, 100]), tn * 256 - rnd.choice([0, 0, 4, 128])
    elif cls < 0.55:   # 128 x 128 kernel region
        M, N = rnd.randrange(260, 1300), rnd.choice([2048, 4096, 4100, 5120, 8192, 11008, 12288]) - rnd.choice([0, 0, 4, 60])
    else:
        M, N = rnd.randrange(1, 5000), rnd.randrange(8, 12000)
    K = 128 * rnd.choice([1, 2, 3, 4, 5, 8, 12, 16, 32, 32, 32, 40, 64, 86])
    if M * N > 64e6 or M * K > 2**31 or N * K > 2**31:
        continue
    x = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
    w = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    name = lib.asq_gemm_kernel_name(M, N, K).decode()
    seen[name] += 1
    if M > 16 and M % 8 == 0 and N % 8 == 0:
        ref = torch._int_mm(x, w.t())
    else:
        pad_m, pad_n = (-M) % 8 + (32 if M <= 16 else 0), (-N) % 8
        xp = torch.nn.functional.pad(x, (0, 0, 0, pad_m)); wp = torch.nn.functional.pad(w, (0, 0, 0, pad_n))
        ref = torch._int_mm(xp, wp.t())[:M, :N].contiguous()
    for rep in range(3):
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        ops.gemm_i8_i32(x, w, out)
        if not torch.equal(out, ref):
            bad += 1
            print("MISMATCH", (M, N, K), name, "rep", rep, int((out != ref).sum()), flush=True)
            break
print("shapes per kernel:", dict(seen), "mismatches:", bad)
sys.exit(1 if bad else 0)
