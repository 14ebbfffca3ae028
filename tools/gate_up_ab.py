#!/usr/bin/env python3
"""gate || up as one GEMM with the SiLU * up epilogue against the composition it replaces (two GEMMs + asq_silu_mul_quantize), LLaMA-2-7B's MLP shapes, offset images,
per-tensor activations, the consumer (down_proj) per-token or per-tensor.  One process, alternating arms.  usage: python tools/gate_up_ab.py [--ms 16384,65536]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="16384,65536")
ap.add_argument("--fk", default="11008x4096")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
F, K = map(int, args.fk.split("x"))
wg = (torch.randn(F, K, device=dev, generator=g) * 22).round().clamp(-128, 127).to(torch.int8)
wu = (torch.randn(F, K, device=dev, generator=g) * 22).round().clamp(-128, 127).to(torch.int8)
w_gu = ops.interleave_gate_up(wg, wu)
img_g, img_u, img_gu = ops.weight_offset_image(wg), ops.weight_offset_image(wu), ops.weight_offset_image(w_gu)
for M in map(int, args.ms.split(",")):
    x = torch.randn(M, K, device=dev, generator=g)
    x[:, torch.rand(K, device=dev, generator=g) < 0.01] *= 20
    x = (x / (x.abs().max() / 127)).half()
    xo, _, ro = ops.quantize_act_off(x, "per-tensor-round")
    bufs = {"g": torch.empty((M, F), dtype=torch.float16, device=dev), "u": torch.empty((M, F), dtype=torch.float16, device=dev), "a": torch.empty((M, F), dtype=torch.float16, device=dev)}

    def composed(per_token):
        ops.linear_w8a8_off(xo, img_g[0], ro, img_g[1], torch.float16, 2e-4, out=bufs["g"])
        ops.linear_w8a8_off(xo, img_u[0], ro, img_u[1], torch.float16, 2e-4, out=bufs["u"])
        return ops.silu_mul_quantize(bufs["g"], bufs["u"], per_token, 0.02, offsets=True)

    def fused(per_token):
        ops.linear_w8a8_gate_up(xo, img_gu[0], torch.float16, 2e-4, 2e-4, None, None, ro, img_gu[1], out=bufs["a"])
        return ops.quantize_act_off(bufs["a"], "per-token" if per_token else "per-tensor-div", 0.02)

    for per_token in (True, False):
        a, b = composed(per_token), fused(per_token)
        same = torch.equal(a[0], b[0]) and (a[1] is None or torch.equal(a[1], b[1]))
        ts = {"composed": [], "fused": []}
        for fn in (composed, fused):
            for _ in range(6):
                fn(per_token)
        torch.cuda.synchronize()
        for r in range(8):
            for name, fn in ((("composed", composed), ("fused", fused)) if r % 2 == 0 else (("fused", fused), ("composed", composed))):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn(per_token)
                e1.record()
                e1.synchronize()
                ts[name].append(e0.elapsed_time(e1) / 5 * 1e3)
        mc, mf = (sorted(ts[k])[len(ts[k]) // 2] for k in ("composed", "fused"))
        print(f"M={M} F={F} K={K} consumer {'per-token' if per_token else 'per-tensor'}: gate + up + silu_mul_quantize {mc:.1f} us | gate||up GEMM + quantiser {mf:.1f} us  ratio {mf / mc:.4f}  int8 equal {same}", flush=True)
