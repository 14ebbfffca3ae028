#!/usr/bin/env python3
"""Module-call time of the decode-sized forwards with the one-launch forward on / off (ASQ_FUSED_FORWARD is read once per process: this script re-runs itself).
    python tools/fused_forward_ab.py            -> table: shape, dtype, mode | eager us per call (host loop, GPU drained at the end) | hipGraph us | kernel us (events)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(4, 4096, 4096, "f32", "per-tensor", "linear"), (4, 4096, 4096, "f16", "per-tensor", "linear"), (1, 4096, 4096, "f16", "per-token", "quantscale"),
         (8, 4096, 4096, "f16", "per-tensor", "linear"), (8, 4096, 4096, "f16", "per-token", "quantscale"), (8, 11008, 4096, "f16", "per-tensor", "linear"),
         (8, 4096, 11008, "f16", "per-token", "quantscale"),
         (16, 4096, 4096, "f16", "per-token", "quantscale"), (16, 4096, 4096, "f16", "per-tensor", "linear"),
         (16, 11008, 4096, "f16", "per-tensor", "linear"), (4, 4096, 11008, "f16", "per-token", "quantscale"), (4, 11008, 4096, "f16", "per-tensor", "linear"),
         (32, 4096, 4096, "f16", "per-tensor", "linear")]


def child():
    import torch
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale
    from autosmoothquant_amd import ops
    dev = torch.device("cuda:0")
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
    res = []
    g = torch.Generator().manual_seed(0)
    for (M, N, K, dt, aq, kind) in CASES:
        cls = W8A8BFP32OFP32Linear if kind == "linear" else W8A8BFP32OFP32LinearWithQuantScale
        m = cls(K, N, False, aq)
        m.weight = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
        m.dequant_scale = torch.tensor(1e-4)
        m = m.to(dev)
        # rotate through enough weights that they come from HBM, not the Infinity Cache (as tools/cold_grid.sh)
        nrot = max(2, int((300 << 20) // (N * K)))
        ws = [m.weight] + [m.weight.clone() for _ in range(nrot - 1)]
        x = (torch.randn(M, K, generator=g) * 3).to(tdt[dt]).to(dev)
        for _ in range(50):
            m(x)
        torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        for i in range(n):
            m._buffers["weight"] = ws[i % nrot]
            m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e6
        # device time per call: events around 200 calls
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(200):
            m._buffers["weight"] = ws[i % nrot]
            m(x)
        b.record()
        b.synchronize()
        res.append({"shape": [M, N, K], "dtype": dt, "act_quant": aq, "fused": bool(ops.forward_is_fused(M, N, K, tdt[dt], "per-token" if aq == "per-token" else "per-tensor-round")), "eager_us": round(eager, 2),
                    "event_us": round(a.elapsed_time(b) / 200 * 1e3, 2)})
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    out = {}
    for tag, v in (("fused", "1"), ("two_launches", "0")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, ASQ_FUSED_FORWARD=v), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("[")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)
        out[tag] = json.loads(line[-1])
    print(f"{'M x N x K':>20} {'dtype':>5} {'act':>11} | {'one launch: call us':>20} {'device us':>10} | {'two launches: call us':>22} {'device us':>10}")
    for a, b in zip(out["fused"], out["two_launches"]):
        s = "x".join(map(str, a["shape"]))
        print(f"{s:>20} {a['dtype']:>5} {a['act_quant']:>11} | {a['eager_us']:>20} {a['event_us']:>10} | {b['eager_us']:>22} {b['event_us']:>10}   {'(fused)' if a['fused'] else '(not a fused shape)'}")
