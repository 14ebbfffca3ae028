import torch, sys
sys.path.insert(0, '/root/repo')
from autosmoothquant_amd import ops
DEV = torch.device('cuda:0')
g = torch.Generator(device=DEV).manual_seed(9)
M, N, K = 4096, 4096, 4096
x = torch.randn(M, K, generator=g, device=DEV)
x[:, torch.randperm(K, generator=torch.Generator().manual_seed(1))[:K // 100].to(DEV)] *= 20
w = torch.randn(N, K, generator=g, device=DEV) * 0.02
ref = x.double() @ w.double().t()
xq, sx = ops.quantize_act_fp8(x, "per-token")
ws_ = float(w.abs().max()) / 448
wq8 = (w / ws_).clamp(-448, 448).to(torch.float8_e4m3fn)
y_tok = ops.linear_fp8(xq, sx, wq8, ws_, None, torch.float32)
xm, xs = ops.quantize_mxfp8(x); wm, wsc = ops.quantize_mxfp8(w)
y_mx = ops.linear_mxfp8(xm, xs, wm, wsc, torch.float32)
rel = lambda y: float((y.double() - ref).norm() / ref.norm())
print('rel err per-token e4m3 %.4f  MX e4m3 %.4f' % (rel(y_tok), rel(y_mx)))
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); [f() for _ in range(n)]; b.record(); b.synchronize(); return a.elapsed_time(b) / n * 1e3
print('us: quantize_mxfp8 %.1f  linear_mxfp8 %.1f  (linear_fp8 %.1f)' % (t(lambda: ops.quantize_mxfp8(x)), t(lambda: ops.linear_mxfp8(xm, xs, wm, wsc, torch.float16)), t(lambda: ops.linear_fp8(xq, sx, wq8, ws_, None, torch.float16))))
