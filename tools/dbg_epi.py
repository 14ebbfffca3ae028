import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, detrng
from oracle import w8a8 as O
from autosmoothquant_amd import ops
dev = torch.device("cuda:0")
M, N, K = 256, 384, 256
xq = detrng.int8_uniform(106, M, (M, K)); w = detrng.int8_uniform(107, N, (N, K))
s_col = (np.abs(detrng.normal(108, 1, (N,))) * 0.01 + 1e-3).astype(np.float32)
acc = O.igemm(xq, w)
d = lambda a: torch.from_numpy(a).to(dev)
for rep in range(3):
    got = ops.linear_w8a8(d(xq), d(w), torch.float16, 0.00123, None, d(s_col), None, "scale_first").float().cpu().numpy()
    ref = O.dequant_epilogue(acc, s_col, None, None, "f16", "scale_first")
    bad = np.argwhere(got != ref)
    print("rep", rep, "mismatches", len(bad))
    if len(bad):
        print(" rows:", sorted(set(bad[:, 0]))[:20], " cols:", sorted(set(bad[:, 1]))[:40])
        for (i, j) in bad[:6]:
            print("  ", i, j, got[i, j], ref[i, j], acc[i, j], s_col[j], "ratio", got[i, j] / max(abs(acc[i, j]), 1))
