#!/usr/bin/env python3
"""G5b: the reference's fp8 STATIC calibration, run here on CPU.

  part 1  layers/nn/linear.py:455-500 FP8StaticLinearQuantizer itself: a fixed e4m3 weight, a sequence of batches whose per-tensor scale rises, falls
          and rises again; stored per batch: the reference's output and its running input_scale (and output_scale with quantize_output=True).
  part 2  quantize/calibration.py:247-339 replace_module / get_layers_to_ignore / quantize_activations_fp8 on the toy LLaMA of tests/calib_toy.py, the
          JSON-lines dataset of G9 and the toy tokenizer; stored: which nn.Linear names were replaced / ignored, every quantizer's weight bytes, weight_scale and
          final input_scale, the token ids in the order the reference fed them, and FP8LinearStatic.from_float(quantizer) on a probe input.
While generating, part 1 is pushed through oracle/fp8.py::fp8_static_quantizer_forward and must agree (scales exactly; outputs to the fp32 summation-order
tolerance the other fp8 fixtures use).  Data only; no reference source travels.  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fp8_static.py"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
import tempfile  # noqa: E402
os.environ.setdefault("HF_HOME", tempfile.mkdtemp(prefix="asq_hf_"))
os.environ.setdefault("HF_DATASETS_OFFLINE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

stub = types.ModuleType("autosmoothquant._CUDA")


class I8CUGEMM:
    def linear_a8_w8_o32_(self, x, w, out):
        out.copy_(x.to(torch.int32) @ w.to(torch.int32).t())


stub.I8CUGEMM = I8CUGEMM
sys.modules["autosmoothquant._CUDA"] = stub
sys.path.insert(0, "/root/reference")
models = types.ModuleType("autosmoothquant.models")
models._MODEL_TYPE = {"LlamaForCausalLM": "llama", "OPTForCausalLM": "transformers"}
sys.modules["autosmoothquant.models"] = models
import autosmoothquant.layers.nn.linear as RL  # noqa: E402
import autosmoothquant.layers.functional.quantization as RQ  # noqa: E402
import autosmoothquant.quantize.calibration as RC  # noqa: E402

import calib_toy  # noqa: E402
import detrng  # noqa: E402
from oracle import fp8 as F8  # noqa: E402

u8 = lambda t: t.detach().view(torch.uint8).cpu().numpy().copy()


def part1(out):
    K, N = 96, 40
    Wf = (detrng.normal(61, 0, (N, K)) * np.float32(0.05)).astype(np.float32)
    b = (detrng.normal(62, 0, (N,)) * np.float32(0.5)).astype(np.float32)
    wq_t, ws_t = RQ.per_tensor_quantize_fp8(torch.from_numpy(Wf.copy()))
    out["p1_wq"], out["p1_ws"], out["p1_b"] = u8(wq_t), np.float32(ws_t.item()), b
    mags = [1.0, 3.0, 0.5, 2.0, 7.0, 1.5]     # the running scale rises at batches 1 and 4 and is kept through the others
    xs = [(detrng.normal(63, i, (5 + i, K)) * np.float32(m)).astype(np.float32) for i, m in enumerate(mags)]
    for qo in (False, True):
        ref = RL.FP8StaticLinearQuantizer(K, N, weight=wq_t.clone(), weight_scale=ws_t.clone(), bias=torch.from_numpy(b.copy()), quantize_output=qo)
        st = F8.StaticQuantizerState(qo)
        for i, x in enumerate(xs):
            y = ref(torch.from_numpy(x.copy())).detach().numpy().copy()
            tag = f"p1_qo{int(qo)}_{i}"
            out[tag + "_x"], out[tag + "_y"] = x, y
            out[tag + "_in_scale"] = np.float32(ref.input_scale.item())
            if qo:
                out[tag + "_out_scale"] = np.float32(ref.output_scale.item())
            o = F8.fp8_static_quantizer_forward(st, x, "f32", out["p1_wq"], out["p1_ws"], b)
            assert np.float32(st.input_scale) == out[tag + "_in_scale"], (tag, st.input_scale, out[tag + "_in_scale"])
            if qo:   # the output scale is the absmax of a float sum: summation order moves its last bits
                assert abs(float(st.output_scale) - float(out[tag + "_out_scale"])) <= 2e-6 * float(out[tag + "_out_scale"]), tag
                assert np.abs(o - y).max() <= 0.07 * float(np.abs(y).max()), tag    # one fp8 code (2^-3 relative) at a rounding boundary
            else:
                assert np.abs(o - y).max() <= 1e-5 * float(np.abs(y).max()), (tag, np.abs(o - y).max())
    out["p1_nbatches"] = np.int64(len(xs))
    # the fp16 combination the reference's CPU path accepts (no bias: F.linear rejects Float bias x Half operands)
    ref = RL.FP8StaticLinearQuantizer(K, N, weight=wq_t.clone(), weight_scale=ws_t.clone(), bias=None, quantize_output=False)
    for i, x in enumerate(xs[:3]):
        xh = torch.from_numpy(x.copy()).half()
        y = ref(xh)
        out[f"p1_f16_{i}_y"], out[f"p1_f16_{i}_in_scale"] = y.float().numpy().copy(), np.float32(ref.input_scale.float().item())


def part2(out):
    ds_path = os.path.join(HERE, "calib_dataset.jsonl")
    model = calib_toy.build_llama()
    names_before = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)]
    patterns = ["re:.*lm_head", "model.layers.1.mlp.up_proj"]
    ignored = sorted(RC.get_layers_to_ignore(model, patterns))
    tok = calib_toy.ToyTokenizer()
    RC.quantize_activations_fp8(model, tok, ds_path, patterns, 5)
    out["p2_linear_names"], out["p2_ignored"], out["p2_patterns"] = np.array(names_before), np.array(ignored), np.array(patterns)
    for j, c in enumerate(tok.calls):
        out[f"p2_ids_{j}"] = c.numpy()
    out["p2_ncalls"] = np.int64(len(tok.calls))
    replaced = [(n, m) for n, m in model.named_modules() if isinstance(m, RL.FP8StaticLinearQuantizer)]
    out["p2_replaced"] = np.array([n for n, _ in replaced])
    assert sorted(n for n, _ in replaced) == sorted(set(names_before) - set(ignored))
    probe = (detrng.normal(64, 0, (7, 64)) * np.float32(1.5)).astype(np.float32)
    out["p2_probe"] = probe
    for n, m in replaced:
        out[f"p2_wq::{n}"], out[f"p2_ws::{n}"], out[f"p2_in_scale::{n}"] = u8(m.weight), np.float32(m.weight_scale.item()), np.float32(m.input_scale.item())
        if m.in_features == 64:
            st = RL.FP8LinearStatic.from_float(m)
            st.output_scale = torch.tensor(0.0)    # (the reference copies None here and its forward then fails on `if self.output_scale`; 0 = "no output quantisation")
            out[f"p2_static_y::{n}"] = st(torch.from_numpy(probe.copy())).detach().numpy().copy()
    print("G5b part 2:", len(replaced), "quantizers,", len(ignored), "ignored,", len(tok.calls), "calibration forwards")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    store = {}
    part1(store)
    part2(store)
    np.savez_compressed(os.path.join(HERE, "g5b_fp8_static.npz"), **store)
    print("G5b written:", len(store), "arrays")
