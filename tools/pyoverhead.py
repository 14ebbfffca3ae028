import time, torch, sys
sys.path.insert(0, '/root/repo')
from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
from autosmoothquant_amd import ops
dev = torch.device('cuda:0')
m = W8A8BFP32OFP32Linear(4096, 4096, False, 'per-tensor')
m.weight = torch.randint(-128, 128, (4096, 4096), dtype=torch.int8)
m = m.to(dev)
x = torch.randn(32, 4096, device=dev, dtype=torch.float16)
for _ in range(20): m(x)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(2000): m(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('cpu per call us', (t1 - t0) / 2000 * 1e6, 'total incl gpu', (t2 - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): m(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)

# ---- where the per-call time goes: the bare C-ABI call (two kernel launches) vs the Python around it
from autosmoothquant_amd import _lib as L
lib = L.lib()
M, K, N = 32, 4096, 4096
nbytes = lib.asq_linear_w8a8_workspace_bytes(M, N, K)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
args = (x.data_ptr(), L.ASQ_F16, m.weight.data_ptr(), out.data_ptr(), M, N, K, L.ASQ_ACT_ROUND, 1.0, 1.0, None, None, ws.data_ptr(), nbytes, st)
for _ in range(20): lib.asq_linear_w8a8_forward(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): lib.asq_linear_w8a8_forward(*args)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('bare ctypes call (quantise + GEMM launches) us', (t1 - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): torch.empty((M, N), dtype=torch.float16, device=dev); torch.empty((nbytes,), dtype=torch.uint8, device=dev)
t1 = time.perf_counter()
print('two torch.empty us', (t1 - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): ops.linear_w8a8_forward(x, m.weight, "per-tensor-round", 1.0, 1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('ops.linear_w8a8_forward us', (t1 - t0) / 2000 * 1e6)

# ---- A/B of the per-stream workspace cache (ops._forward_ws) in this process
def _uncached(lib_, M_, N_, K_, dev_, stream_):
    n_ = lib_.asq_linear_w8a8_workspace_bytes(M_, N_, K_)
    return torch.empty((n_,), dtype=torch.uint8, device=dev_), n_
cached = ops._forward_ws
for name, fn in (("cached", cached), ("uncached", _uncached), ("cached", cached), ("uncached", _uncached)):
    ops._forward_ws = fn
    for _ in range(200): m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4000): m(x)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f'module call, workspace {name}: {(t1 - t0) / 4000 * 1e6:.2f} us')
ops._forward_ws = cached
