#!/usr/bin/env python3
"""Dev tool: A/B of TWO builds of libasq_hip.so in ONE process (same box, same DVFS state, alternating batches).

usage: python tools/lib_ab.py --a path/to/libA.so --b path/to/libB.so [--shapes MxNxK,...] [--kernel p8|p4|...] [--dtype f16|bf16]
Both libraries are dlopen'ed side by side (distinct file names -> distinct handles); asq_linear_w8a8 is called through ctypes with the same device buffers;
outputs are compared bit for bit first.  Operands: bench statistics (weights rms ~22, activations mostly in [-3, 3]) or --uniform."""
import argparse, ctypes, os, sys
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--a", required=True)
ap.add_argument("--b", required=True)
ap.add_argument("--shapes", default="4096x4096x4096")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--uniform", action="store_true")
ap.add_argument("--batch", type=int, default=50)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--per-token", action="store_true")
ap.add_argument("--bias", action="store_true")
ap.add_argument("--per-channel", action="store_true")
ap.add_argument("--ws", action="store_true", help="pass an initialised workspace (header + 64 KiB): the kernels that need one -- persistent p16, split-K -- become eligible")
ap.add_argument("--a-kernel", default=None, help="ASQ_GEMM_KERNEL seen by library A (the library reads it once, at its first launch)")
ap.add_argument("--b-kernel", default=None)
ap.add_argument("--a-env", default="", help="KEY=VAL[,KEY=VAL] set while library A makes its first launches (the libraries read their switches once)")
ap.add_argument("--b-env", default="")
args = ap.parse_args()

vp, i64, f32, sz, cint = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_int


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    h.asq_linear_w8a8.restype = cint
    h.asq_linear_w8a8.argtypes = [vp, vp, vp, cint, i64, i64, i64, f32, vp, vp, vp, cint, vp, sz, vp]
    h.asq_last_error.restype = ctypes.c_char_p
    h.asq_workspace_init.restype = cint
    h.asq_workspace_init.argtypes = [vp, sz, vp]
    return h


if os.path.abspath(args.a) == os.path.abspath(args.b):   # the same build twice (kernel A/B): dlopen needs two file names for two sets of statics
    import shutil, tempfile
    twin = os.path.join(tempfile.mkdtemp(), "libasq_hip_twin.so")
    shutil.copy(args.b, twin)
    args.b = twin
libs = {"A": load(args.a), "B": load(args.b)}
forced = {"A": args.a_kernel, "B": args.b_kernel}
extra_env = {"A": dict(kv.split("=", 1) for kv in args.a_env.split(",") if kv), "B": dict(kv.split("=", 1) for kv in args.b_env.split(",") if kv)}
dev = torch.device("cuda:0")
tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
DT = {"f16": 1, "bf16": 2}[args.dtype]
g = torch.Generator(device=dev).manual_seed(0)
stream = torch.cuda.current_stream().cuda_stream
for sh in args.shapes.split(","):
    M, N, K = map(int, sh.split("x"))
    if args.uniform:
        x = torch.randint(-128, 128, (M, K), device=dev, generator=g, dtype=torch.int8)
        w = torch.randint(-128, 128, (N, K), device=dev, generator=g, dtype=torch.int8)
    else:
        x = (torch.randn(M, K, device=dev, generator=g) * 2.9).round().clamp(-128, 127).to(torch.int8)
        w = (torch.randn(N, K, device=dev, generator=g) * 21.7).round().clamp(-127, 127).to(torch.int8)
    s_row = torch.rand(M, device=dev, generator=g) * 0.01 + 0.001 if args.per_token else None
    bias = torch.randn(N, device=dev, generator=g) if args.bias else None
    s_col = torch.rand(N, device=dev, generator=g) * 1e-3 + 1e-4 if args.per_channel else None
    outs = {k: torch.empty(M, N, dtype=tdt, device=dev) for k in libs}
    wss = {}
    if args.ws:
        for k in libs:
            libs[k].asq_gemm_workspace_bytes.restype = sz
            libs[k].asq_gemm_workspace_bytes.argtypes = [i64, i64, i64]
            wss[k] = torch.zeros(max(8192 + 65536, int(libs[k].asq_gemm_workspace_bytes(M, N, K))), dtype=torch.uint8, device=dev)   # (what the shape's own kernel asks for: stream-K slabs, split-K images)
            assert libs[k].asq_workspace_init(wss[k].data_ptr(), wss[k].numel(), stream) == 0

    def call(k):
        rc = libs[k].asq_linear_w8a8(x.data_ptr(), w.data_ptr(), outs[k].data_ptr(), DT, M, N, K, 1.25e-4, s_row.data_ptr() if s_row is not None else None, s_col.data_ptr() if s_col is not None else None,
                                     bias.data_ptr() if bias is not None else None, 0, wss[k].data_ptr() if args.ws else None, wss[k].numel() if args.ws else 0, stream)
        if rc != 0:
            raise RuntimeError(libs[k].asq_last_error().decode())

    for k in libs:
        if forced[k]:
            os.environ["ASQ_GEMM_KERNEL"] = forced[k]
        else:
            os.environ.pop("ASQ_GEMM_KERNEL", None)
        for key in set(extra_env["A"]) | set(extra_env["B"]):
            os.environ.pop(key, None)
        os.environ.update(extra_env[k])
        call(k)
    torch.cuda.synchronize()
    same = torch.equal(outs["A"].view(torch.int16), outs["B"].view(torch.int16))
    ts = {k: [] for k in libs}
    for r in range(args.rounds + 2):
        for k in (("A", "B") if r % 2 == 0 else ("B", "A")):
            for _ in range(args.batch):    # re-settle on this arm
                call(k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.batch):
                call(k)
            e1.record()
            e1.synchronize()
            if r >= 2:
                ts[k].append(e0.elapsed_time(e1) / args.batch * 1e3)
    ops = 2.0 * M * N * K
    line = f"{sh:>18s} {'uniform' if args.uniform else 'bench  '} outputs identical: {same} |"
    for k in libs:
        t = sorted(ts[k])
        med = t[len(t) // 2]
        line += f" {k}: median {med:7.2f} us (min {t[0]:7.2f}) = {ops / med / 1e6:6.0f} TOPS = {ops / med / 1e6 / 50.33:5.1f} % |"
    ta, tb = sorted(ts["A"])[len(ts["A"]) // 2], sorted(ts["B"])[len(ts["B"]) // 2]
    print(line + f" B/A time {tb / ta:.4f}", flush=True)
