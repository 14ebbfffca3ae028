#!/usr/bin/env python3
"""Generate the block-level golden vectors by IMPORTING the reference in the
build container -- data only (inputs + the reference's outputs), no reference source travels.

  g6_smooth.npz   quantize/smooth.py: smooth_ln_fcs on LayerNorm + 3 Linears and RMSNorm + 2 Linears,
                  alpha in {0.5, 0.8}, fp32 and fp16 weights
  g7_block.npz    models/baichuan.py: Int8BaichuanLayer.from_float(...).forward at hidden = 256 under three
                  quant configs (position_embedding "ALIBI", attention_mask None -- the only branch the
                  reference can execute on CPU)

The native module is the same exact-integer stub as in make_golden.py.  Usage:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_blocks.py
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


class I8CUGEMM:  # csrc/int8gemm/bindings.cpp:145-155; exact integer matmul == CUBLAS_COMPUTE_32I, alpha 1, beta 0
    def linear_a8_w8_o32_(self, x, w, out):
        out.copy_(torch._int_mm(x, w.t()) if x.shape[0] > 16 and x.shape[0] % 8 == 0 else x.to(torch.int32) @ w.to(torch.int32).t())


stub.I8CUGEMM = I8CUGEMM
sys.modules["autosmoothquant._CUDA"] = stub
torch.cuda.current_device = lambda: torch.device("cpu")  # linear.py:101 hard-codes the CUDA device
sys.path.insert(0, "/root/reference")
import autosmoothquant.quantize.smooth as RS  # noqa: E402
# models/__init__.py imports llama/opt/mixtral, which need transformers == 4.42.3: register an
# empty package object so that only models/baichuan.py is executed
_pkg = types.ModuleType("autosmoothquant.models")
_pkg.__path__ = ["/root/reference/autosmoothquant/models"]
sys.modules["autosmoothquant.models"] = _pkg
import autosmoothquant.models.baichuan as RB  # noqa: E402
from autosmoothquant.thirdparty.baichuan.modeling_baichuan import RMSNorm as BRMSNorm, BaichuanLayer  # noqa: E402
from autosmoothquant.thirdparty.baichuan.configuration_baichuan import BaichuanConfig  # noqa: E402
from transformers.models.llama.modeling_llama import LlamaRMSNorm  # noqa: E402

import detrng  # noqa: E402


def f32(seed, stream, shape, scale=1.0):
    return (detrng.normal(seed, stream, shape) * scale).astype(np.float32)


def gen_g6():
    store, index = {}, []
    H = 64
    for ci, (kind, outs, alpha, dt) in enumerate([
            ("layernorm", [48, 48, 32], 0.5, torch.float32), ("layernorm", [48, 48, 32], 0.8, torch.float32),
            ("llama_rms", [96, 80], 0.5, torch.float32), ("llama_rms", [96, 80], 0.8, torch.float32),
            ("baichuan_rms", [96, 80], 0.5, torch.float32), ("layernorm", [48, 32], 0.5, torch.float16)]):
        lw = f32(600 + ci, 0, (H,), 0.2) + 1.0
        lb = f32(600 + ci, 1, (H,), 0.1)
        act = np.abs(f32(600 + ci, 2, (H,), 3.0)) + 0.01
        act[3] = 1e-9       # exercises both clamp(min=1e-5) guards
        Ws = [f32(600 + ci, 10 + j, (o, H), 0.05) for j, o in enumerate(outs)]
        Ws[0][:, 3] = 0.0
        if kind == "layernorm":
            ln = torch.nn.LayerNorm(H)
            ln.bias.data = torch.from_numpy(lb.copy())
            mt = "transformers"
        elif kind == "llama_rms":
            ln, mt = LlamaRMSNorm(H), "llama"
        else:
            ln, mt = BRMSNorm(H), "baichuan"
        ln.weight.data = torch.from_numpy(lw.copy())
        fcs = []
        for W in Ws:
            fc = torch.nn.Linear(H, W.shape[0], bias=False)
            fc.weight.data = torch.from_numpy(W.copy())
            fcs.append(fc)
        if dt == torch.float16:
            ln, fcs = ln.half(), [fc.half() for fc in fcs]
        key = f"c{ci}"
        store[key + "_ln_w"] = ln.weight.detach().float().numpy().copy()
        if kind == "layernorm":
            store[key + "_ln_b"] = ln.bias.detach().float().numpy().copy()
        for j, fc in enumerate(fcs):
            store[key + f"_fc{j}"] = fc.weight.detach().float().numpy().copy()
        store[key + "_act"] = act
        RS.smooth_ln_fcs(ln, fcs, torch.from_numpy(act.copy()), mt, alpha)
        store[key + "_ln_w_out"] = ln.weight.detach().float().numpy().copy()
        if kind == "layernorm":
            store[key + "_ln_b_out"] = ln.bias.detach().float().numpy().copy()
        for j, fc in enumerate(fcs):
            store[key + f"_fc{j}_out"] = fc.weight.detach().float().numpy().copy()
        index.append(f"{ci}|{kind}|{mt}|{alpha}|{'f16' if dt == torch.float16 else 'f32'}|{len(fcs)}")
    store["index"] = np.array(index)
    np.savez_compressed(os.path.join(HERE, "g6_smooth.npz"), **store)
    print("g6:", len(index), "cases")


def gen_g7():
    H, HEADS, INTER, B, S = 256, 4, 704, 2, 24
    cfg = BaichuanConfig(hidden_size=H, intermediate_size=INTER, num_attention_heads=HEADS, num_hidden_layers=1, vocab_size=64,
                         model_max_length=64, rms_norm_eps=1e-6)
    W = {"W_pack": f32(700, 0, (3 * H, H), 0.05), "o_proj": f32(700, 1, (H, H), 0.05), "gate_proj": f32(700, 2, (INTER, H), 0.05),
         "up_proj": f32(700, 3, (INTER, H), 0.05), "down_proj": f32(700, 4, (H, INTER), 0.05),
         "ln1": f32(700, 5, (H,), 0.1) + 1.0, "ln2": f32(700, 6, (H,), 0.1) + 1.0}
    x = f32(700, 7, (B, S, H), 1.0)
    x[..., 5] *= 12.0  # an outlier channel, SmoothQuant-style

    def float_layer():
        m = BaichuanLayer(cfg)
        m.self_attn.W_pack.weight.data = torch.from_numpy(W["W_pack"].copy())
        m.self_attn.o_proj.weight.data = torch.from_numpy(W["o_proj"].copy())
        m.mlp.gate_proj.weight.data = torch.from_numpy(W["gate_proj"].copy())
        m.mlp.up_proj.weight.data = torch.from_numpy(W["up_proj"].copy())
        m.mlp.down_proj.weight.data = torch.from_numpy(W["down_proj"].copy())
        m.input_layernorm.weight.data = torch.from_numpy(W["ln1"].copy())
        m.post_attention_layernorm.weight.data = torch.from_numpy(W["ln2"].copy())
        return m

    # calibration: absmax/127 of every quantised linear's input on the float layer (quantize/calibration.py:185-244 in spirit)
    fl = float_layer()
    rec = {}
    hooks = [mod.register_forward_hook(lambda m_, i, o, n=n: rec.__setitem__(n, float(i[0].abs().max()) / 127.0))
             for n, mod in [("attn_in", fl.self_attn.W_pack), ("o_in", fl.self_attn.o_proj), ("mlp_in", fl.mlp.gate_proj), ("down_in", fl.mlp.down_proj)]]
    with torch.no_grad():
        y_float = fl(torch.from_numpy(x.copy()))[0].numpy().copy()
    for h in hooks:
        h.remove()
    store = {"x": x, "y_float": y_float, "scales": np.array([rec["attn_in"], rec["o_in"], rec["mlp_in"], rec["down_in"]], np.float64),
             "dims": np.array([H, HEADS, INTER, B, S]), "eps": np.array(1e-6)}
    for k, v in W.items():
        store["W_" + k] = v
    configs = [{"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-token"},
               {"qkv": "per-token", "out": "per-token", "fc1": "per-token", "fc2": "per-token"},
               {"qkv": "per-tensor", "out": "per-tensor", "fc1": "per-tensor", "fc2": "per-tensor"}]
    index = []
    for ci, qc in enumerate(configs):
        q = RB.Int8BaichuanLayer.from_float(float_layer(), cfg, qc, "ALIBI", rec["attn_in"], 1.0, rec["o_in"], rec["mlp_in"], rec["down_in"])
        with torch.no_grad():
            y = q(torch.from_numpy(x.copy()), attention_mask=None)[0]
        store[f"c{ci}_y"] = y.numpy().copy()
        rel = float(np.abs(store[f"c{ci}_y"] - y_float).max() / np.abs(y_float).max())
        print(f"g7 config {ci}: {qc}  rel-err vs float layer {rel:.3e}")
        index.append("|".join([str(ci), qc["qkv"], qc["out"], qc["fc1"], qc["fc2"]]))
    store["index"] = np.array(index)
    np.savez_compressed(os.path.join(HERE, "g7_block.npz"), **store)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen_g6()
    gen_g7()
