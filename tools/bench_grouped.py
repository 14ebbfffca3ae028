"""Times the grouped expert launches of BASELINE configs[4] (Mixtral-8x7B: w1/w3 4096 -> 14336, w2 14336 -> 4096, 8192 routed rows over 8 experts)
one kernel at a time: with / without the workspace that enables the in-launch K split of tail tiles, for several routings.
Run on the GPU box:  python tools/bench_grouped.py  ->  one line per (routing, projection, mode): median / min microseconds of 30 launches."""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from autosmoothquant_amd import _lib as L  # noqa: E402

E, H, FF, R = 8, 4096, 14336, 8192
ROUTINGS = {"bench (multinomial)": [983, 1034, 1090, 1052, 1002, 1028, 995, 1008], "exactly uniform": [1024] * 8,
            "skewed 4:2:1:1:.5:.5:.25:.25": None, "two empty experts": None}


def counts_for(name):
    if ROUTINGS[name] is not None:
        return ROUTINGS[name]
    p = [4.0, 2.0, 1.0, 1.0, 0.5, 0.5, 0.25, 0.25] if name.startswith("skewed") else [3.0, 1.0, 0.0, 1.0, 2.0, 0.0, 1.0, 1.0]
    g = torch.Generator().manual_seed(5)
    return torch.bincount(torch.multinomial(torch.tensor(p), R, replacement=True, generator=g), minlength=E).tolist()


def main():
    dev = torch.device("cuda:0")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    xs = {K: torch.randint(-128, 128, (R, K), dtype=torch.int8, device=dev, generator=g) for K in (H, FF)}
    ws_ = {"w1": torch.randint(-128, 128, (E, FF, H), dtype=torch.int8, device=dev, generator=g),
           "w2": torch.randint(-128, 128, (E, H, FF), dtype=torch.int8, device=dev, generator=g)}
    sg = torch.full((E,), 1e-4, device=dev)
    srow = torch.full((R,), 1e-2, device=dev)
    n = lib.asq_grouped_workspace_bytes(R, H, FF, E)
    wsb = torch.empty((n,), dtype=torch.uint8, device=dev)
    L.check(lib.asq_workspace_init(wsb.data_ptr(), n, st), "init")
    for name in ROUTINGS:
        counts = counts_for(name)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
        tiles = sum((c + 255) // 256 for c in counts)
        for proj in ("w1", "w2"):
            w = ws_[proj]
            N, K = w.shape[1], w.shape[2]
            out = torch.empty((R, N), dtype=torch.float16, device=dev)
            for mode in ("no workspace", "workspace"):
                def launch():
                    L.check(lib.asq_linear_w8a8_grouped_ws(xs[K].data_ptr(), w.data_ptr(), out.data_ptr(), L.ASQ_F16, offs.data_ptr(), E, R, N, K, sg.data_ptr(),
                                                           srow.data_ptr() if proj == "w2" else None, None, wsb.data_ptr() if mode == "workspace" else None,
                                                           n if mode == "workspace" else 0, st), "grouped")
                for _ in range(40):
                    launch()
                torch.cuda.synchronize()
                ts = []
                for _ in range(30):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); launch(); b.record(); b.synchronize()
                    ts.append(a.elapsed_time(b) * 1e3)
                ts.sort()
                ops = 2.0 * R * N * K
                print(f"{name:32s} rows {counts} tile rows {tiles:3d} x {(N + 255) // 256:2d} | {proj} {mode:13s} median {ts[15]:7.1f} us  min {ts[0]:7.1f} us  "
                      f"{ops / ts[15] / 1e6:7.1f} TOPS", flush=True)


if __name__ == "__main__":
    main()
