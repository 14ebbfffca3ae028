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
