#!/usr/bin/env python3
"""Write a tiny REFERENCE-FORMAT quantised checkpoint by importing the reference in the build container:
Int8BaichuanForCausalLM.from_float(float model, decoder_layer_scales, quant_config).save_pretrained(dir) plus the quant_config.json
that examples/smoothquant_model.py:96-99 writes beside it -- exactly the directory examples/test_model.py loads.  Data only
(safetensors weights + two JSON files); also stores an input and the reference model's own hidden-state output for it.

  tests/golden/ckpt_baichuan_w8a8/{model.safetensors, config.json, quant_config.json, io.npz}

2 layers, hidden 256, 4 heads, inter 704, vocab 64; quant_config qkv/fc1 per-tensor, out per-token, fc2 per-tensor (so that every
buffer kind of the contract appears: {q,k,v}_dequant_scale, dequant_scale, quant_scale, folded norm weights).
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py
"""
import json
import os
import shutil
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

stub = types.ModuleType("autosmoothquant._CUDA")


class I8CUGEMM:  # csrc/int8gemm/bindings.cpp:145-155; exact integer matmul == CUBLAS_COMPUTE_32I, alpha 1, beta 0
    def linear_a8_w8_o32_(self, x, w, out):
        out.copy_(x.to(torch.int32) @ w.to(torch.int32).t())


stub.I8CUGEMM = I8CUGEMM
sys.modules["autosmoothquant._CUDA"] = stub
torch.cuda.current_device = lambda: torch.device("cpu")
sys.path.insert(0, "/root/reference")
_pkg = types.ModuleType("autosmoothquant.models")
_pkg.__path__ = ["/root/reference/autosmoothquant/models"]
sys.modules["autosmoothquant.models"] = _pkg
import autosmoothquant.models.baichuan as RB  # noqa: E402
from autosmoothquant.thirdparty.baichuan.modeling_baichuan import BaichuanForCausalLM  # noqa: E402
from autosmoothquant.thirdparty.baichuan.configuration_baichuan import BaichuanConfig  # noqa: E402

import detrng  # noqa: E402


def main():
    H, HEADS, INTER, L, V, B, S = 256, 4, 704, 2, 64, 2, 24
    cfg = BaichuanConfig(hidden_size=H, intermediate_size=INTER, num_attention_heads=HEADS, num_hidden_layers=L, vocab_size=V,
                         model_max_length=64, rms_norm_eps=1e-6)
    fm = BaichuanForCausalLM(cfg)
    with torch.no_grad():
        for i, (n, p) in enumerate(fm.named_parameters()):
            v = detrng.normal(900, i, tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(v * 0.05 + (1.0 if "layernorm" in n or n.endswith("norm.weight") else 0.0)))
    x = (detrng.normal(901, 0, (B, S, H))).astype(np.float32)
    x[..., 5] *= 12.0
    # calibration on the float layers: absmax/127 of each quantised linear's input (quantize/calibration.py:185-244 in spirit)
    scales = []
    h = torch.from_numpy(x.copy())
    with torch.no_grad():
        for layer in fm.model.layers:
            rec = {}
            hooks = [m.register_forward_hook(lambda m_, i, o, n=n: rec.__setitem__(n, float(i[0].abs().max()) / 127.0))
                     for n, m in [("attn_input_scale", layer.self_attn.W_pack), ("out_input_scale", layer.self_attn.o_proj),
                                  ("gate_input_scale", layer.mlp.gate_proj), ("down_input_scale", layer.mlp.down_proj)]]
            h = layer(h)[0]
            for hk in hooks:
                hk.remove()
            rec["attn_output_scale"] = 1.0
            scales.append(rec)
    qc = {"qkv": "per-tensor", "out": "per-token", "fc1": "per-tensor", "fc2": "per-tensor"}
    qm = RB.Int8BaichuanForCausalLM.from_float(fm, scales, qc)
    out = os.path.join(HERE, "ckpt_baichuan_w8a8")
    shutil.rmtree(out, ignore_errors=True)
    qm.save_pretrained(out, safe_serialization=True)
    with open(os.path.join(out, "quant_config.json"), "w") as f:      # examples/smoothquant_model.py:96-99
        json.dump(qc, f, indent=4)
    for extra in ("generation_config.json",):
        p = os.path.join(out, extra)
        if os.path.exists(p):
            os.remove(p)
    # the reference's own decoder-stack output for x (ALIBI branch, attention_mask None: the path it can run on CPU)
    with torch.no_grad():
        hq = torch.from_numpy(x.copy())
        per_layer = []
        for layer in qm.model.layers:
            hq = layer(hq, attention_mask=None)[0]
            per_layer.append(hq.numpy().copy())
    np.savez_compressed(os.path.join(out, "io.npz"), x=x, y_layers=np.stack(per_layer), y_float=h.numpy(),
                        scales=np.array([[s["attn_input_scale"], s["out_input_scale"], s["gate_input_scale"], s["down_input_scale"]] for s in scales]))
    print(sorted(os.listdir(out)))
    from safetensors import safe_open
    with safe_open(os.path.join(out, "model.safetensors"), "pt") as f:
        for k in sorted(f.keys()):
            t = f.get_tensor(k)
            print(f"  {k:60s} {str(t.dtype):14s} {tuple(t.shape)}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
