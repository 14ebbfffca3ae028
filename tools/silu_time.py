import torch, sys
sys.path.insert(0, '/root/repo')
from autosmoothquant_amd import ops
dev = torch.device('cuda:0')
for (M, K, dt) in [(65536, 11008, torch.float16), (65536, 11008, torch.bfloat16), (8192, 14336, torch.float16)]:
    g = torch.randn(M, K, device=dev, dtype=dt) * 2
    u = torch.randn(M, K, device=dev, dtype=dt)
    for pt in (True, False):
        f = lambda: ops.silu_mul_quantize(g, u, per_token=pt, quant_scale=0.05)
        for _ in range(5): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); b.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(M, K, dt, 'per_token' if pt else 'per_tensor', f'{us:.1f} us', f'{M*K*(2*g.element_size()+1)/us/1e6:.2f} TB/s')
