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

    # round 5, last session: a per-tensor consumer's int8 straight from the epilogue (asq_linear_w8a8_gate_up_q8), and what the consumer's GEMM pays for running on plain
    # activations behind it (down_proj: [M, F] x [K, F]^T on images after the quantiser vs on the plain int8 of the epilogue)
    wd = (torch.randn(K, F, device=dev, generator=g) * 22).round().clamp(-128, 127).to(torch.int8)
    img_d = ops.weight_offset_image(wd)
    dbuf = torch.empty((M, K), dtype=torch.float16, device=dev)

    def mlp_fused_quantiser():
        aq, _, aro = fused(False)
        return ops.linear_w8a8_off(aq, img_d[0], aro, img_d[1], torch.float16, 2e-4, out=dbuf)

    def mlp_q8():
        aq = ops.linear_w8a8_gate_up_q8(xo, img_gu[0], torch.float16, 2e-4, 2e-4, 0.02, None, None, ro, img_gu[1])
        return ops.linear_w8a8(aq, wd, torch.float16, 2e-4, out=dbuf)

    y0 = mlp_fused_quantiser().clone()
    same_q8 = torch.equal(y0, mlp_q8())
    tq = {"quantiser": [], "q8": []}
    for fn in (mlp_fused_quantiser, mlp_q8):
        for _ in range(4):
            fn()
    torch.cuda.synchronize()
    for r in range(8):
        for name, fn in ((("quantiser", mlp_fused_quantiser), ("q8", mlp_q8)) if r % 2 == 0 else (("q8", mlp_q8), ("quantiser", mlp_fused_quantiser))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            tq[name].append(e0.elapsed_time(e1) / 5 * 1e3)
    m0, m1 = (sorted(tq[k])[len(tq[k]) // 2] for k in ("quantiser", "q8"))
    print(f"M={M} F={F} K={K} whole MLP, per-tensor down_proj: gate||up + quantiser + down on images {m0:.1f} us | gate||up with int8-out epilogue + down on plain activations {m1:.1f} us  ratio {m1 / m0:.4f}  outputs equal {same_q8}", flush=True)
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
