#!/usr/bin/env python3
"""Generate g8_n1.npz by IMPORTING the reference in the build container -- data only (inputs + the
reference's outputs), no reference source travels.

  lnq_*     layers/nn/fused.py: LayerNormQ.from_float(nn.LayerNorm, output_scale).forward(x)            -> int8
  dqadd_*   layers/functional/fused.py: dq_add_layernorm_q_py(int32, scale, residual, gamma, beta, eps)  -> (residual_out, int8)
  rmsq_*    models/baichuan.py: Int8BaichuanRMSNorm.from_float(norm, scale)(x), then the per-tensor prologue of the
            linears that consume it (layers/nn/linear.py:95-96: round, clamp, int8)                       -> (y, int8)

The native module is stubbed (these functions never call it).  Usage:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_n1.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

stub = types.ModuleType("autosmoothquant._CUDA")
stub.I8CUGEMM = type("I8CUGEMM", (), {})
stub.dq_add_layernorm_q = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("native op not built"))
sys.modules["autosmoothquant._CUDA"] = stub
torch.cuda.current_device = lambda: torch.device("cpu")
sys.path.insert(0, "/root/reference")
_pkg = types.ModuleType("autosmoothquant.models")
_pkg.__path__ = ["/root/reference/autosmoothquant/models"]
sys.modules["autosmoothquant.models"] = _pkg
import autosmoothquant.layers.nn.fused as RF  # noqa: E402
import autosmoothquant.layers.functional.fused as RFF  # noqa: E402
import autosmoothquant.models.baichuan as RB  # noqa: E402
from autosmoothquant.thirdparty.baichuan.modeling_baichuan import RMSNorm as BRMSNorm  # noqa: E402

import detrng  # noqa: E402


def f32(seed, stream, shape, scale=1.0):
    return (detrng.normal(seed, stream, shape) * scale).astype(np.float32)


def main():
    store, index = {}, []
    ci = 0
    for H, M in ((64, 9), (320, 5), (4096, 3)):
        for dt in (torch.float32, torch.float16):
            tag = "f32" if dt == torch.float32 else "f16"
            x = f32(800 + ci, 0, (M, H), 2.0)
            x[:, 3] *= 15.0
            x = torch.from_numpy(x).to(dt)
            gamma, beta = f32(800 + ci, 1, (H,), 0.2) + 1.0, f32(800 + ci, 2, (H,), 0.3)
            scale = 0.037
            # --- LayerNormQ (weights stay fp32; x is cast to fp32 inside: fused.py:11)
            ln = torch.nn.LayerNorm(H, eps=1e-5)
            ln.weight.data, ln.bias.data = torch.from_numpy(gamma.copy()), torch.from_numpy(beta.copy())
            q = RF.LayerNormQ.from_float(ln, scale)
            with torch.no_grad():
                out = q(x.clone())
            key = f"lnq_{ci}"
            store[key + "_x"], store[key + "_w"], store[key + "_b"] = x.float().numpy().copy(), q.weight.detach().numpy().copy(), q.bias.detach().numpy().copy()
            store[key + "_out"] = out.numpy().copy()
            index.append(f"lnq|{ci}|{tag}|{H}|{M}|1e-5")
            # --- dq_add_layernorm_q_py (fp32 residual: the reference's int32 + alpha add is only meaningful there)
            if dt == torch.float32:
                acc = torch.from_numpy(((detrng.uniform01(810 + ci, 0, (M, H)) - 0.5) * 80000).astype(np.int32))
                in_scale = 3.1e-4
                res = torch.from_numpy(f32(810 + ci, 1, (M, H), 1.5))
                with torch.no_grad():
                    r_out, i8 = RFF.dq_add_layernorm_q_py(acc, in_scale, res, q.weight.detach(), q.bias.detach(), 1e-5)
                key = f"dqadd_{ci}"
                store[key + "_acc"], store[key + "_res"] = acc.numpy().copy(), res.numpy().copy()
                store[key + "_w"], store[key + "_b"] = q.weight.detach().numpy().copy(), q.bias.detach().numpy().copy()
                store[key + "_in_scale"] = np.array(in_scale, np.float64)
                store[key + "_res_out"], store[key + "_out"] = r_out.numpy().copy(), i8.numpy().copy()
                index.append(f"dqadd|{ci}|{tag}|{H}|{M}|1e-5")
            # --- scale-folded RMSNorm + per-tensor round
            norm = BRMSNorm(H, 1e-6)
            norm.weight.data = torch.from_numpy(gamma.copy())
            if dt == torch.float16:
                norm = norm.half()
            with torch.no_grad():
                qn = RB.Int8BaichuanRMSNorm.from_float(norm, scale)
                y = qn(x.clone())
                i8 = y.round().clamp(-128, 127).to(torch.int8)   # layers/nn/linear.py:95-96
            key = f"rmsq_{ci}"
            store[key + "_x"], store[key + "_w"] = x.float().numpy().copy(), qn.weight.detach().float().numpy().copy()
            store[key + "_y"], store[key + "_out"] = y.float().numpy().copy(), i8.numpy().copy()
            index.append(f"rmsq|{ci}|{tag}|{H}|{M}|1e-6")
            ci += 1
    store["index"] = np.array(index)
    np.savez_compressed(os.path.join(HERE, "g8_n1.npz"), **store)
    print("g8:", len(index), "cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
