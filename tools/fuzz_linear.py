#!/usr/bin/env python3
"""Dev tool: asq_linear_w8a8 (GEMM + fused dequant / bias epilogue, every output dtype and operand combination, both association orders) on random
shapes through every dispatcher branch, against the same arithmetic spelled out in eager torch on the device (int32 accumulators from
torch._int_mm, then separate fp32 multiplies / add, then one cast).  Exact.  usage: python tools/fuzz_linear.py [n_shapes] [seed]"""
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
    if cls < 0.2:
        tn = rnd.choice([16, 20, 43, 48]); tm = rnd.choice([t for t in range(2, 30) if 0 < (t * tn) % 256 <= 96 and t * tn > 256] or [6])
        M, N = tm * 256 - rnd.choice([0, 9]), tn * 256 - rnd.choice([0, 8])
    elif cls < 0.5:
        M, N = rnd.randrange(100, 1300), rnd.choice([2048, 4096, 4104, 5120, 8192, 11008]) - rnd.choice([0, 0, 8, 56])
    else:
        M, N = rnd.randrange(1, 4000), 8 * rnd.randrange(1, 1400)
    K = 128 * rnd.choice([1, 2, 3, 4, 5, 8, 16, 32, 32, 40, 64, 86])
    if M * N > 48e6:
        continue
    Mp = (M + 7) // 8 * 8 + (32 if M <= 16 else 0)
    x = torch.randint(-128, 128, (Mp, K), generator=g, device=dev, dtype=torch.int8)
    w = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    acc = torch._int_mm(x, w.t())[:M].to(torch.float32)
    x = x[:M].contiguous()
    dt = rnd.choice([torch.float16, torch.bfloat16, torch.float32])
    s_scalar = rnd.choice([1.0, 3.1e-4, 0.0123])
    s_row = (torch.rand(M, generator=g, device=dev) * 0.02 + 1e-3) if rnd.random() < 0.5 else None
    s_col = (torch.rand(N, generator=g, device=dev) * 0.02 + 1e-3) if rnd.random() < 0.5 else None
    bias = (torch.randn(N, generator=g, device=dev) * 5) if rnd.random() < 0.5 else None
    order = rnd.choice(["scale_first", "acc_first"])
    sc = s_col[None, :] if s_col is not None else torch.full((1, 1), s_scalar, device=dev, dtype=torch.float32)
    if order == "scale_first":
        ref = (sc * s_row[:, None] if s_row is not None else sc) * acc
    else:
        ref = acc * sc
        if s_row is not None:
            ref = ref * s_row[:, None]
    if bias is not None:
        ref = ref + bias[None, :]
    ref = ref.to(dt)
    name = lib.asq_gemm_kernel_name(M, N, K).decode()
    seen[name] += 1
    got = ops.linear_w8a8(x, w, dt, s_scalar, s_row, s_col, bias, order)
    if not torch.equal(got.view(torch.int16 if dt != torch.float32 else torch.int32), ref.view(torch.int16 if dt != torch.float32 else torch.int32)):
        bad += 1
        d = (got.float() - ref.float()).abs()
        print("MISMATCH", (M, N, K), name, dt, order, "row" if s_row is not None else "-", "col" if s_col is not None else "-", "bias" if bias is not None else "-",
              int((got != ref).sum()), float(d.max()), flush=True)
print("shapes per kernel:", dict(seen), "mismatches:", bad)
sys.exit(1 if bad else 0)
