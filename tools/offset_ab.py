#!/usr/bin/env python3
"""Dev tool (round 4): what does an operand OFFSET buy the whole GEMM under the socket power limit?  TIMING ONLY -- the arms multiply different
matrices; the product form (exact rank-1 correction in the epilogue) is asq_linear_w8a8's `bias` path, see DESIGN 4.2.

Arms, one process, alternating batches on the same buffers' shapes (bench.py statistics: weights N(0, 0.02^2) absmax-quantised, activations N(0,1)
with 1 % outlier channels x20 in int8 units):
    base      W x X
    wrow      (W + cw[n]) x X             cw[n] = 127 - max_k W[n,k]  (what a per-channel offset may use without clamping), optionally capped
    wrow_x    (W + cw[n]) x (X + cx[m])   cx[m] = min(C, 127 - max_k X[m,k])
    absabs    |W| x |X|
    zeros
usage: python tools/offset_ab.py [--shapes MxNxK,...] [--cap 64] [--cx 3]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="4096x4096x4096")
ap.add_argument("--cap", type=int, default=64)
ap.add_argument("--cx", type=int, default=3)
ap.add_argument("--batch", type=int, default=50)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--kernel", default=None)
ap.add_argument("--per-token", action="store_true", help="per-token quantised activations (every row reaches +-127) and the row-scale epilogue")
ap.add_argument("--arms", default=None, help="comma-separated subset of the arms")
args = ap.parse_args()
if args.kernel:
    os.environ["ASQ_GEMM_KERNEL"] = args.kernel
h = _lib.lib()
vp, i64, f32, sz, cint = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int
h.asq_linear_w8a8.restype = cint
h.asq_linear_w8a8_off.restype = cint
h.asq_linear_w8a8_off.argtypes = [vp, vp, vp, cint, i64, i64, i64, f32, vp, vp, vp, cint, vp, vp, vp]
h.asq_linear_w8a8.argtypes = [vp, vp, vp, cint, i64, i64, i64, f32, vp, vp, vp, cint, vp, sz, vp]
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
stream = torch.cuda.current_stream().cuda_stream
for sh in args.shapes.split(","):
    M, N, K = map(int, sh.split("x"))
    xf = torch.randn(M, K, device=dev, generator=g)
    ch = torch.rand(K, device=dev, generator=g) < 0.01
    xf[:, ch] *= 20.0
    if args.per_token:
        rs = xf.abs().max(dim=1, keepdim=True).values / 127.0
        x = (xf / rs).round().clamp(-128, 127).to(torch.int8)
        s_row_t = rs.flatten().float().contiguous()
    else:
        x = (xf / (xf.abs().max() / 127.0)).round().clamp(-128, 127).to(torch.int8)
        s_row_t = None
    wf = torch.randn(N, K, device=dev, generator=g) * 0.02
    w = (wf / (wf.abs().max() / 127.0)).round().clamp(-128, 127).to(torch.int8)
    del xf, wf
    cw = (127 - w.max(dim=1).values.int()).clamp(max=args.cap)
    if args.per_token:   # the library's rule: +C if it fits, else -C, else 0
        rmax, rmin = x.max(dim=1).values.int(), x.min(dim=1).values.int()
        C = abs(args.cx)
        cx = torch.where(rmax <= 127 - C, torch.full_like(rmax, C), torch.where(rmin >= -128 + C, torch.full_like(rmax, -C), torch.zeros_like(rmax)))
    elif args.cx >= 0:
        cx = (127 - x.max(dim=1).values.int()).clamp(max=args.cx)
    else:
        cx = (-128 - x.min(dim=1).values.int()).clamp(min=args.cx)   # negative offsets: products and partial sums mostly negative, start values positive
    arms = {
        "base": (w, x),
        "wrow": ((w.int() + cw[:, None]).to(torch.int8), x),
        "wrow_x": ((w.int() + cw[:, None]).to(torch.int8), (x.int() + cx[:, None]).to(torch.int8)),
        "wneg_x": ((w.int() - (128 + w.min(dim=1).values.int()).clamp(max=args.cap)[:, None]).to(torch.int8), (x.int() + cx[:, None]).to(torch.int8)),
        "absabs": (w.int().abs().clamp(max=127).to(torch.int8), x.int().abs().clamp(max=127).to(torch.int8)),
        "zeros": (torch.zeros_like(w), torch.zeros_like(x)),
    }
    print(f"{sh}: cw mean {cw.float().mean():.1f} (min {int(cw.min())}, max {int(cw.max())}); cx mean {cx.float().mean():.2f}; "
          f"W rms {w.float().pow(2).mean().sqrt():.1f}, X rms {x.float().pow(2).mean().sqrt():.2f}", flush=True)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    # the product path: the library's own images and vectors (exact results), and the same launch with the vectors zeroed (start values 0: timing only)
    w_img, col_off = ops.weight_offset_image(w)
    x_img = (x.int() + cx[:, None]).to(torch.int8)
    row_off = torch.stack([cx, x_img.int().sum(dim=1).int()], dim=1).contiguous()
    zr, zc = torch.zeros_like(row_off), torch.zeros_like(col_off)
    row_off = row_off.contiguous(); col_off = col_off.contiguous()
    ref = torch.empty_like(out)
    assert h.asq_linear_w8a8(x.data_ptr(), w.data_ptr(), ref.data_ptr(), 1, M, N, K, 1.25e-4, s_row_t.data_ptr() if s_row_t is not None else None, None, None, 0, None, 0, stream) == 0
    got = ops.linear_w8a8_off(x_img, w_img, row_off, col_off, torch.float16, 1.25e-4, s_row_t)
    print("  asq_linear_w8a8_off == asq_linear_w8a8:", torch.equal(ref, got), flush=True)
    off_arms = {"off_exact": (w_img, x_img, row_off, col_off), "off_start0": (w_img, x_img, zr, zc), "off_plainops": (w, x, zr, zc)}
    for a in off_arms:
        arms[a] = None

    sr = s_row_t.data_ptr() if s_row_t is not None else None

    def call(a):
        if a in off_arms:
            ww, xx, ro, co = off_arms[a]
            rc = h.asq_linear_w8a8_off(xx.data_ptr(), ww.data_ptr(), out.data_ptr(), 1, M, N, K, 1.25e-4, sr, None, None, 0, ro.data_ptr(), co.data_ptr(), stream)
        else:
            ww, xx = arms[a]
            rc = h.asq_linear_w8a8(xx.data_ptr(), ww.data_ptr(), out.data_ptr(), 1, M, N, K, 1.25e-4, sr, None, None, 0, None, 0, stream)
        assert rc == 0, h.asq_last_error()

    if args.arms:
        arms = {a: arms[a] for a in args.arms.split(",")}
    ts = {a: [] for a in arms}
    names = list(arms)
    for r in range(args.rounds + 1):
        order = names if r % 2 == 0 else names[::-1]
        for a in order:
            for _ in range(args.batch * 3):   # re-settle the clocks on this arm
                call(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.batch):
                call(a)
            e1.record()
            e1.synchronize()
            if r >= 1:
                ts[a].append(e0.elapsed_time(e1) / args.batch * 1e3)
    nops = 2.0 * M * N * K
    base = sorted(ts["base"])[len(ts["base"]) // 2]
    for a in names:
        t = sorted(ts[a])
        med = t[len(t) // 2]
        if os.environ.get("AB_SAMPLES"):
            print("      samples:", " ".join(f"{v:.1f}" for v in ts[a]))
        print(f"  {a:12s} median {med:7.2f} us (min {t[0]:7.2f}) = {nops / med / 1e6:6.0f} TOPS = {nops / med / 1e6 / 50.33:5.1f} % of 5033   vs base {med / base:.4f}", flush=True)
