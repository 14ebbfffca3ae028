"""Toy HF models + tokenizer shared by tests/golden/make_golden_calib.py (which runs the REFERENCE's collectors on them) and the tests that run
this package's collectors on the same models and inputs.  Weights come from the build-owned counter RNG, so both sides rebuild them bit for bit."""
import numpy as np
import torch

import detrng


def _det_init(model, seed, std=0.08):
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            v = detrng.normal(seed, i, tuple(p.shape)).astype(np.float32) * std
            if "norm" in n and n.endswith("weight"):
                v = v + 1.0
            p.copy_(torch.from_numpy(v))
    return model.eval()


def build_llama(seed=940):
    from transformers.models.llama.modeling_llama import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=64, intermediate_size=96, num_attention_heads=4, num_key_value_heads=4, num_hidden_layers=2, vocab_size=97,
                      max_position_embeddings=64, rms_norm_eps=1e-5, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    cfg.architectures = ["LlamaForCausalLM"]
    return _det_init(LlamaForCausalLM(cfg), seed)


def build_opt(seed=950):
    from transformers.models.opt.modeling_opt import OPTConfig, OPTForCausalLM
    cfg = OPTConfig(hidden_size=64, ffn_dim=96, num_attention_heads=4, num_hidden_layers=2, vocab_size=97, max_position_embeddings=64, word_embed_proj_dim=64,
                    do_layer_norm_before=True, dropout=0.0)
    cfg._attn_implementation = "eager"
    cfg.architectures = ["OPTForCausalLM"]
    return _det_init(OPTForCausalLM(cfg), seed)


class ToyTokenizer:
    """text -> ids in [3, 97): a fixed function of the characters; honours max_length / truncation like the call in quantize/calibration.py:79-80."""

    class _Out:
        def __init__(self, ids):
            self.input_ids = ids

    def __init__(self):
        self.calls = []

    def __call__(self, text, return_tensors="pt", max_length=None, truncation=False):
        ids = [3 + (ord(c) * 7 + i * 13) % 94 for i, c in enumerate(text)]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        t = torch.tensor([ids], dtype=torch.long)
        self.calls.append(t.clone())
        return self._Out(t)


DATASET = [
    "the quick brown fox jumps over the lazy dog",
    "int8 matrix cores want their operands in full 128 byte lines",
    "per-token scales are absmax over the row divided by 127",
    "smoothquant migrates the outliers of the activations into the weights",
    "a decoder layer is attention then a gated mlp",
    "calibration only needs the running maximum of every linear input",
    "weights stay resident in hbm and are broadcast once",
    "zzzz AAAA 0000 !!!!",
]
