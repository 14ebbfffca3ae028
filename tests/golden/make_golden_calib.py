#!/usr/bin/env python3
"""N3 fixtures: the REFERENCE's own calibration collectors -- quantize/calibration.py `get_act_scales` (:44-88) and
`get_static_decoder_layer_scales` (:185-244, with collect_llama_layer_scales :113-136 and collect_transformers_layer_scales :91-111) -- run here on
two toy Hugging Face models (LLaMA- and OPT-named modules, tests/calib_toy.py), a tiny JSON-lines dataset and a toy tokenizer.

The module imports `autosmoothquant.models._MODEL_TYPE`; autosmoothquant/models/__init__.py pulls in the wrappers that need transformers 4.42.3, so
`autosmoothquant.models` is replaced by a stub holding the same mapping (models/__init__.py: architecture name -> model type).  Stored (data only):
  tests/golden/calib_dataset.jsonl     the dataset the reference's load_dataset("json", ...) reads
  tests/golden/g9_calib.npz            per model: the token ids in the order the reference fed them (after its shuffle(seed=42)),
                                       act_scales (per-channel absmax per nn.Linear), act_dict (per-tensor in/out absmax), decoder_layer_scales
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_calib.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
import tempfile  # noqa: E402
os.environ.setdefault("HF_HOME", tempfile.mkdtemp(prefix="asq_hf_"))   # the reference's load_dataset() caches; keep that out of the home directory
os.environ.setdefault("HF_DATASETS_OFFLINE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

stub = types.ModuleType("autosmoothquant._CUDA")


class I8CUGEMM:
    def linear_a8_w8_o32_(self, x, w, out):
        out.copy_(x.to(torch.int32) @ w.to(torch.int32).t())


stub.I8CUGEMM = I8CUGEMM
sys.modules["autosmoothquant._CUDA"] = stub
sys.path.insert(0, "/root/reference")
models = types.ModuleType("autosmoothquant.models")
models._MODEL_TYPE = {"LlamaForCausalLM": "llama", "LLaMAForCausalLM": "llama", "OPTForCausalLM": "transformers", "MixtralForCausalLM": "mixtral",
                      "BaichuanForCausalLM": "baichuan", "BaiChuanForCausalLM": "baichuan"}   # reference models/__init__.py
sys.modules["autosmoothquant.models"] = models
import autosmoothquant.quantize.calibration as RC  # noqa: E402

import calib_toy  # noqa: E402


def main():
    ds_path = os.path.join(HERE, "calib_dataset.jsonl")
    with open(ds_path, "w") as f:
        for t in calib_toy.DATASET:
            f.write(json.dumps({"text": t}) + "\n")
    out = {}
    for tag, build, mtype in (("llama", calib_toy.build_llama, "llama"), ("opt", calib_toy.build_opt, "transformers")):
        model = build()
        tok = calib_toy.ToyTokenizer()
        act = RC.get_act_scales(model, tok, ds_path, num_samples=6, seq_len=24)
        ids_a = [c.numpy() for c in tok.calls]
        tok2 = calib_toy.ToyTokenizer()
        scales, act_dict = RC.get_static_decoder_layer_scales(model, tok2, ds_path, num_samples=5, seq_len=16, model_type=mtype)
        ids_b = [c.numpy() for c in tok2.calls]
        names = sorted(act)
        out[f"{tag}_names"] = np.array(names)
        for n in names:
            out[f"{tag}_act::{n}"] = act[n].numpy()
        out[f"{tag}_io"] = np.array([[act_dict[n]["input"], act_dict[n]["output"]] for n in names], dtype=np.float64)
        keys = sorted(scales[0])
        out[f"{tag}_scale_keys"] = np.array(keys)
        out[f"{tag}_scales"] = np.array([[s[k] for k in keys] for s in scales], dtype=np.float64)
        for j, a in enumerate(ids_a):
            out[f"{tag}_ids_act_{j}"] = a
        for j, a in enumerate(ids_b):
            out[f"{tag}_ids_static_{j}"] = a
        print(tag, len(names), "linears;", keys)
    np.savez_compressed(os.path.join(HERE, "g9_calib.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
