"""CPU: MixtralLayer's [E,N,K] expert stacks follow the per-expert modules (QuantArena.pack / .to() / a rewritten scale left the grouped
launch on stale stacks).  The per-expert buffers are the source of truth; `_stacks_current()` detects every way they can move, moe() restacks."""
import torch

from autosmoothquant_amd import harness, replica


def _layer():
    torch.manual_seed(0)
    fl = harness.MixtralLayer(64, 96, 4, 2, experts=4, top_k=2)
    with torch.no_grad():
        for p in fl.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape) * 0.05)
    scales = {"attn_in": 0.05, "o_in": 0.05, "mlp_in": 0.05, "down_in": [0.05, 0.06, 0.07, 0.08]}
    return harness.to_w8a8_mixtral(fl, scales)


def test_stacks_follow_rehomed_buffers_and_rewritten_scales():
    q = _layer()
    assert q._stacks_current()
    before = [e.w1.weight.clone() for e in q.experts]
    # 1. replica.QuantArena.pack re-homes every quantised buffer into the arena: the stacks no longer alias the modules
    replica.QuantArena(q).pack("cpu")
    assert not q._stacks_current()
    assert all(torch.equal(a, e.w1.weight) for a, e in zip(before, q.experts))
    q.stack_experts()
    assert q._stacks_current()
    assert all(torch.equal(q._w1_stack[i], before[i]) and e.w1.weight.data_ptr() == q._w1_stack[i].data_ptr() for i, e in enumerate(q.experts))
    # 2. a data write through the module's buffer lands in the stack (one storage)
    q.experts[2].w3.weight.fill_(7)
    assert int(q._w3_stack[2].min()) == 7 and q._stacks_current()
    # 3. a rewritten dequant scale (load_state_dict of the scale buffers) is a mismatch until the scale vector is rebuilt
    q.experts[1].w2.dequant_scale = torch.tensor(0.125)
    assert not q._stacks_current()
    q.stack_experts()
    assert q._stacks_current() and float(q._w2_scale[1]) == 0.125
    # 4. replacing a weight buffer (what Module.to(device) does) is a mismatch
    q.experts[0].w1._buffers["weight"] = q.experts[0].w1._buffers["weight"].clone()
    assert not q._stacks_current()
