"""GPU (-m gpu): N4 at BASELINE configs[4] size -- Mixtral-8x7B expert projections (w1/w3: 4096 -> 14336, w2: 14336 -> 4096), 8192
routed rows (4096 tokens x top-2) over 8 experts, ONE grouped launch, checked against the ORACLE per expert (not against the HIP
per-expert path): uniform, skewed and empty-expert routings (reference models/mixtral.py:99-145 runs these as a Python loop of
Int8Linear calls; fp8: layers/nn/linear.py:336-369).

The oracle at this size would take many minutes for all 8192 x 14336 outputs, so the comparison is on SAMPLED outputs -- 6144 random
entries plus the first / last row of every expert (where a routing-offset bug would show) -- each recomputed from scratch: an exact
int64 dot product followed by oracle.w8a8.dequant_epilogue (int8: bit-exact), or a float64 dot product of the decoded fp8 operands
(fp8: rtol 1e-3, the tolerance SURVEY 8c states for easy_fp8_gemm)."""
import numpy as np
import pytest
import torch

import detrng
from oracle import fp8 as F8
from oracle import w8a8 as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E, H, FF, R = 8, 4096, 14336, 8192
ROUTINGS = {"uniform": [1.0] * 8, "skewed": [4.0, 2.0, 1.0, 1.0, 0.5, 0.5, 0.25, 0.25], "empty_expert": [3.0, 1.0, 0.0, 1.0, 2.0, 0.0, 1.0, 1.0]}


def _counts(p, seed):
    p = np.asarray(p, np.float64) / np.sum(p)
    u = detrng.uniform01(seed, 0, (R,)).astype(np.float64)
    e = np.searchsorted(np.cumsum(p), u, side="right").clip(0, E - 1)
    return np.bincount(e, minlength=E)


def _samples(counts, N, seed):
    offs = np.concatenate([[0], np.cumsum(counts)])
    idx = (detrng.u64(seed, 1, 6144) % np.uint64(R * N)).astype(np.int64)
    mi, ni = list(idx // N), list(idx % N)
    for e in range(E):
        if counts[e]:
            for m in (offs[e], offs[e + 1] - 1):
                for n in (0, N // 2 + 3, N - 1):
                    mi.append(int(m)); ni.append(n)
    mi, ni = np.array(mi), np.array(ni)
    ei = np.searchsorted(offs, mi, side="right") - 1
    return offs, mi, ni, ei


@pytest.mark.parametrize("routing", sorted(ROUTINGS))
@pytest.mark.parametrize("proj", ["w1", "w2"])
def test_grouped_int8_full_size_vs_oracle(routing, proj):
    from autosmoothquant_amd import ops
    K, N = (H, FF) if proj == "w1" else (FF, H)
    counts = _counts(ROUTINGS[routing], 170)
    assert counts.sum() == R and (routing != "empty_expert" or (counts[2] == 0 and counts[5] == 0))
    offs, mi, ni, ei = _samples(counts, N, 171)
    gen = torch.Generator(device=DEV).manual_seed(172 + K)    # operands are generated ON the device (470 MB of weights) and the sampled rows read back
    xq = torch.randint(-128, 128, (R, K), dtype=torch.int8, device=DEV, generator=gen)
    w = torch.randint(-128, 128, (E, N, K), dtype=torch.int8, device=DEV, generator=gen)
    sg = (np.abs(detrng.normal(174, 0, (E,))) * 1e-4 + 2e-5).astype(np.float32)
    per_token = proj == "w2"                                   # w2 = WithQuantScale(per-token), w1 = Linear(per-tensor)
    s_row = (np.abs(detrng.normal(175, 0, (R,))) * 0.01 + 1e-3).astype(np.float32) if per_token else None
    bias = detrng.normal(176, 0, (E, N)).astype(np.float32)
    got = ops.linear_w8a8_grouped(xq, w, torch.from_numpy(offs.astype(np.int32)).to(DEV), torch.from_numpy(sg).to(DEV), torch.float16,
                                  None if s_row is None else torch.from_numpy(s_row).to(DEV), torch.from_numpy(bias).to(DEV))
    assert tuple(got.shape) == (R, N)
    xs = xq[torch.from_numpy(mi).to(DEV)].cpu().numpy().astype(np.int64)
    ws = w[torch.from_numpy(ei).to(DEV), torch.from_numpy(ni).to(DEV)].cpu().numpy().astype(np.int64)
    acc = np.einsum("ik,ik->i", xs, ws)
    assert np.abs(acc).max() < 2 ** 31
    want = np.empty(len(mi), np.float32)
    for j in range(len(mi)):   # the module epilogue per sampled output: (s_group [* s_row]) * float(acc) + bias -> fp16
        want[j] = O.dequant_epilogue(np.array([[acc[j]]], np.int32), np.float32(sg[ei[j]]), None if s_row is None else s_row[mi[j]:mi[j] + 1],
                                     bias[ei[j], ni[j]:ni[j] + 1], "f16")[0, 0]
    g = got[torch.from_numpy(mi).to(DEV), torch.from_numpy(ni).to(DEV)].float().cpu().numpy()
    assert np.array_equal(g, want), f"{int((g != want).sum())} of {len(g)} sampled outputs differ"


@pytest.mark.parametrize("routing", ["uniform", "empty_expert"])
def test_grouped_fp8_full_size_vs_oracle(routing):
    from autosmoothquant_amd import ops
    K, N = H, FF
    counts = _counts(ROUTINGS[routing], 180)
    offs, mi, ni, ei = _samples(counts, N, 181)
    gen = torch.Generator(device=DEV).manual_seed(182)
    xb = torch.randint(0, 256, (R, K), dtype=torch.uint8, device=DEV, generator=gen)
    xb[(xb & 0x7F) == 0x7F] = 0x30                              # no NaN encodings (e4m3fn: S.1111.111)
    wb = torch.randint(0, 256, (E, N, K), dtype=torch.uint8, device=DEV, generator=gen)
    wb[(wb & 0x7F) == 0x7F] = 0x30
    wb &= 0xBF                                                  # keep |w| < 2
    xq, w = xb.view(torch.float8_e4m3fn), wb.view(torch.float8_e4m3fn)
    a_scale = (np.abs(detrng.normal(184, 0, (R,))) * 1e-3 + 1e-4).astype(np.float32)
    w_scale = (np.abs(detrng.normal(185, 0, (E,))) * 1e-2 + 1e-3).astype(np.float32)
    got = ops.linear_fp8_grouped(xq, torch.from_numpy(a_scale).to(DEV), w, torch.from_numpy(w_scale).to(DEV),
                                 torch.from_numpy(offs.astype(np.int32)).to(DEV), torch.float32)
    xs = F8.e4m3fn_to_f32(xb[torch.from_numpy(mi).to(DEV)].cpu().numpy()).astype(np.float64)
    ws = F8.e4m3fn_to_f32(wb[torch.from_numpy(ei).to(DEV), torch.from_numpy(ni).to(DEV)].cpu().numpy()).astype(np.float64)
    want = np.einsum("ik,ik->i", xs, ws) * a_scale[mi].astype(np.float64) * w_scale[ei].astype(np.float64)
    g = got[torch.from_numpy(mi).to(DEV), torch.from_numpy(ni).to(DEV)].cpu().numpy().astype(np.float64)
    # |error| <= 1e-3 of the output scale: a sum of 4096 products accumulated in fp32 (the reference's own F.linear has the same freedom)
    assert np.abs(g - want).max() <= 1e-3 * np.abs(want).max(), float(np.abs(g - want).max() / np.abs(want).max())
