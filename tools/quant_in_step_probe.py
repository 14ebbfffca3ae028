#!/usr/bin/env python3
"""Round 5: why do the row quantisers take 11.1 / 14.0 us inside the bench step (rocprofv3 kernel durations, profiles/r5_bench_kernel_stats.csv) and 9.5 / 12.7 us alone?
Run under `rocprofv3 --kernel-trace --stats`: three phases of 400 iterations each, separated by marker kernels (torch.zeros of distinct sizes), on 4096 x 4096 fp16:
  A  quantiser (image) alone, back to back
  B  one gemm_i8_p16 launch (4096^3 on images), then the quantiser            -- the step's situation
  C  one gemm_i8_p16 launch, ~20 us of an unrelated small kernel chain, then the quantiser
The per-phase kernel durations come from the trace (tools/quant_in_step_probe.py --report <kernel_trace.csv>)."""
import csv, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    q = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "quant_rows_wave" in r["Kernel_Name"]][1:]   # (the first call is the set-up's)
    for mi, mode in enumerate(("per-tensor image", "per-token image")):     # launch order below: mode x phase x 400 iterations
        for pi, ph in enumerate(("A alone", "B behind a p16 launch", "C behind p16 + ~20 us of small kernels")):
            v = q[(mi * 3 + pi) * 400:(mi * 3 + pi + 1) * 400][50:]
            print(f"{mode} | {ph} | n={len(v)} mean {sum(v) / len(v):.2f} us min {min(v):.2f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
x = torch.randn(M, K, device=dev)
x[:, torch.rand(K, device=dev) < 0.01] *= 20
x = (x / (x.abs().max() / 127)).half()
w = (torch.randn(N, K, device=dev) * 22).round().clamp(-128, 127).to(torch.int8)
img = ops.weight_offset_image(w)
xo, _, ro = ops.quantize_act_off(x, "per-tensor-round")
out = torch.empty(M, N, dtype=torch.float16, device=dev)
small = torch.zeros(256, device=dev)
names = {}
for mode in ("per-tensor-round", "per-token"):
    for phase in ("A", "B", "C"):
        torch.cuda.synchronize()
        for it in range(400):
            if phase in ("B", "C"):
                ops.linear_w8a8_off(xo, img[0], ro, img[1], torch.float16, 1e-4, out=out)
            if phase == "C":
                for _ in range(10):
                    small.add_(1.0)
            ops.quantize_act_off(x, mode)
        torch.cuda.synchronize()
print("done")
