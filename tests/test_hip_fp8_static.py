"""GPU (-m gpu): the fp8 static calibration on the device against G5b -- FP8StaticLinearQuantizer (running-maximum input scale, the GEMM on the RUNNING
scale; reference layers/nn/linear.py:455-500) and quantize_activations_fp8 (quantize/calibration.py:292-339) on the toy LLaMA, then FP8LinearStatic.from_float.
Tolerance: the scales are absmax / 448 of the same inputs -> exact for module inputs that are exact (part 1), 2e-4 relative for the first linears of the model
forward and 3e-2 behind fp8 GEMMs (part 2, see there); outputs 1e-3 relative (fp32 summation order; the fp8 matrix cores accumulate in fp32)."""
import os

import numpy as np
import pytest
import torch

import calib_toy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g5b_fp8_static.npz"), allow_pickle=False)


def _quantizer(qo, bias=True):
    from autosmoothquant_amd.layers.nn.linear import FP8StaticLinearQuantizer
    w = torch.from_numpy(G["p1_wq"]).view(torch.float8_e4m3fn).to(DEV)
    b = torch.nn.Parameter(torch.from_numpy(G["p1_b"]).to(DEV), requires_grad=False) if bias else None
    return FP8StaticLinearQuantizer(w.shape[1], w.shape[0], weight=w, weight_scale=torch.tensor(float(G["p1_ws"])), bias=b, quantize_output=qo)


@pytest.mark.parametrize("qo", [False, True])
def test_static_quantizer_equals_the_reference_batch_by_batch(qo):
    m = _quantizer(qo)
    for i in range(int(G["p1_nbatches"])):
        tag = f"p1_qo{int(qo)}_{i}"
        y = m(torch.from_numpy(G[tag + "_x"]).to(DEV)).float().cpu().numpy()
        assert np.float32(m.input_scale.item()) == G[tag + "_in_scale"], tag
        ref = G[tag + "_y"]
        if qo:
            assert abs(float(m.output_scale) - float(G[tag + "_out_scale"])) <= 1e-4 * float(G[tag + "_out_scale"]), (tag, float(m.output_scale), float(G[tag + "_out_scale"]))
            assert np.abs(y - ref).max() <= 0.07 * float(np.abs(ref).max())
        else:
            assert np.abs(y - ref).max() <= 1e-3 * float(np.abs(ref).max()), tag


def test_static_quantizer_fp16_inputs():
    m = _quantizer(False, bias=False)
    for i in range(3):
        y = m(torch.from_numpy(G[f"p1_qo0_{i}_x"]).half().to(DEV)).float().cpu().numpy()
        assert np.float32(m.input_scale.float().item()) == G[f"p1_f16_{i}_in_scale"]
        ref = G[f"p1_f16_{i}_y"]
        assert np.abs(y - ref).max() <= 4e-3 * float(np.abs(ref).max())


def test_quantize_activations_fp8_flow_matches_the_reference():
    from autosmoothquant_amd.quantize import quantize_activations_fp8
    from autosmoothquant_amd.layers.nn.linear import FP8StaticLinearQuantizer, FP8LinearStatic
    model = calib_toy.build_llama().to(DEV)
    ds = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calib_dataset.jsonl")
    tok = calib_toy.ToyTokenizer()
    quantize_activations_fp8(model, tok, ds, list(G["p2_patterns"]), int(G["p2_ncalls"]))
    for j, c in enumerate(tok.calls):
        assert np.array_equal(c.numpy(), G[f"p2_ids_{j}"])
    replaced = {n: m for n, m in model.named_modules() if isinstance(m, FP8StaticLinearQuantizer)}
    assert sorted(replaced) == sorted(str(n) for n in G["p2_replaced"])
    for n in G["p2_ignored"]:
        assert isinstance(model.get_submodule(str(n)), torch.nn.Linear)
    probe = torch.from_numpy(G["p2_probe"]).to(DEV)
    for n, m in replaced.items():
        assert np.array_equal(m.weight.detach().view(torch.uint8).cpu().numpy(), G[f"p2_wq::{n}"]), n
        assert abs(float(m.weight_scale) - float(G[f"p2_ws::{n}"])) <= 1.2e-7 * float(G[f"p2_ws::{n}"]), n   # (absmax / 448 computed by torch ON THE DEVICE: its fp32 division may differ from the host's in the last place)
        ref_s = float(G[f"p2_in_scale::{n}"])
        # layer 0's q/k/v read the (float) embedding norm: same inputs as the recording up to fp32 summation order.  Everything behind them has been through fp8
        # GEMMs: a last-place difference in a sum flips e4m3 codes (2^-3 relative each) and the absmax statistics downstream move by up to a few per cent
        first = n.startswith("model.layers.0.self_attn.") and n.rsplit(".", 1)[1] in ("q_proj", "k_proj", "v_proj")
        assert abs(float(m.input_scale) - ref_s) <= (2e-4 if first else 3e-2) * ref_s, (n, float(m.input_scale), ref_s)
        key = f"p2_static_y::{n}"
        if key in G.files:
            st = FP8LinearStatic.from_float(m).to(DEV)
            assert float(st.output_scale) == 0.0                               # no output quantisation recorded -> none applied
            y = st(probe).float().cpu().numpy()
            assert np.abs(y - G[key]).max() <= 0.07 * float(np.abs(G[key]).max())   # a 2e-4 different static scale moves a few e4m3 codes of the input
