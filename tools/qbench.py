#!/usr/bin/env python3
"""Dev tool: time the activation quantisers through the C-ABI (preallocated buffers, ctypes calls, batches of 50 launches per event pair; the
host side of a call is ~3 us, so launches of >= ~5 us are device-bound).  Prints algorithmic TB/s = M * K * (sizeof(x) + 1) / time.
usage: python tools/qbench.py [--shapes MxK,...] [--dtypes f16,f32]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="4096x4096,4096x11008,65536x4096,512x4096")
ap.add_argument("--dtypes", default="f16")
args = ap.parse_args()
h = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
TD = {"f16": (torch.float16, 1), "bf16": (torch.bfloat16, 2), "f32": (torch.float32, 0)}
for sh in args.shapes.split(","):
    M, K = map(int, sh.split("x"))
    for dn in args.dtypes.split(","):
        tdt, DT = TD[dn]
        x = torch.randn(M, K, device=dev)
        x[:, torch.rand(K, device=dev) < 0.01] *= 20
        x = x.to(tdt)
        xq = torch.empty(M, K, dtype=torch.int8, device=dev)
        s_row = torch.empty(M, dtype=torch.float32, device=dev)
        row_off = torch.empty(M, 2, dtype=torch.int32, device=dev)
        for mode, mname in ((2, "per-token"), (0, "per-tensor-round"), (1, "per-tensor-div")):
            for off in (False, True):
                def call():
                    if off:
                        rc = h.asq_quantize_act_off(x.data_ptr(), DT, mode, 0.7, xq.data_ptr(), s_row.data_ptr(), row_off.data_ptr(), M, K, st)
                    else:
                        rc = h.asq_quantize_act(x.data_ptr(), DT, mode, 0.7, xq.data_ptr(), s_row.data_ptr(), M, K, st)
                    assert rc == 0, h.asq_last_error()
                for _ in range(20):
                    call()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(50):
                        call()
                    b.record(); b.synchronize()
                    ts.append(a.elapsed_time(b) / 50 * 1e3)
                us = sorted(ts)[len(ts) // 2]
                by = M * K * (x.element_size() + 1)
                print(json.dumps({"M": M, "K": K, "dtype": dn, "mode": mname, "offset_image": off, "us": round(us, 2), "TBps": round(by / us / 1e6, 2)}), flush=True)
