import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autosmoothquant_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((4, 4096, 4096), (4, 11008, 4096), (8, 4096, 4096), (8, 11008, 4096), (4, 32000, 4096), (16, 4096, 4096), (2, 4096, 11008), (8, 8192, 4096)):
    nrot = max(2, int((300 << 20) // (N * K)))
    ws = [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(dev) for _ in range(nrot)]
    x = (torch.randn(M, K, generator=g) * 3).half().to(dev)
    out = {}
    for mode in ("per-tensor-round", "per-token"):
        for _ in range(20):
            ops.linear_w8a8_forward(x, ws[0], mode, 1.0, 1e-4)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(300):
            ops.linear_w8a8_forward(x, ws[i % nrot], mode, 1.0, 1e-4)
        b.record(); b.synchronize()
        out[mode] = a.elapsed_time(b) / 300 * 1e3
    # GEMM alone on int8 x
    xq, _ = ops.quantize_act(x, "per-tensor-round")
    for _ in range(20):
        ops.linear_w8a8(xq, ws[0], torch.float16, 1e-4)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(300):
        ops.linear_w8a8(xq, ws[i % nrot], torch.float16, 1e-4)
    b.record(); b.synchronize()
    print(f"{M}x{N}x{K} fused={ops.forward_is_fused(M,N,K,torch.float16)} stages={os.environ.get('ASQ_FQ_STAGES','auto')}: forward per-tensor {out['per-tensor-round']:.2f} us, per-token {out['per-token']:.2f} us; int8 GEMM alone {a.elapsed_time(b)/300*1e3:.2f} us", flush=True)
