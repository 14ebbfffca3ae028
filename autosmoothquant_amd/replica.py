"""Replica-parallel forwards: one process per GPU, every rank holds a full copy of the
quantised buffers and processes its own rows.  The ONLY collective is a one-time
broadcast of {int8 weight, fp32 bias, fp32 scalar scales} from rank 0 at load time
(``torch.distributed`` backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
The reference has no distributed code at all; the correctness criterion is
"every replica's rows equal the 1-GPU rows bit for bit" -- rows of an Int8Linear are
independent (per-token scales are per row, per-tensor has no cross-row state).
"""
import torch
import torch.distributed as dist


def shard_rows(M, world_size, rank):
    """Contiguous row range of `rank`: rows[g] = M*g/G ... M*(g+1)/G."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (M * rank) // world_size, (M * (rank + 1)) // world_size


def _w8a8_modules(root):
    from .layers.nn.linear import _W8A8Base
    return [m for m in root.modules() if isinstance(m, _W8A8Base)]


def broadcast_quantized(root, src=0, group=None, device=None):
    """Broadcast every W8A8 module's buffers from `src`.  Device buffers (weight, bias) go as
    they are; the host-pinned scalar scales of all modules travel packed in ONE fp32 tensor.
    Returns the number of payload bytes moved (for GB/s reporting)."""
    mods = _w8a8_modules(root)
    nbytes = 0
    names = []
    for m in mods:
        for n in m._host_scalars:
            names.append((m, n))
    dev = device if device is not None else (mods[0].weight.device if mods else torch.device("cpu"))
    packed = torch.tensor([float(m._buffers[n]) for m, n in names], dtype=torch.float32, device=dev)
    if len(names):
        dist.broadcast(packed, src=src, group=group)
        nbytes += packed.numel() * 4
        host = packed.cpu()
        for i, (m, n) in enumerate(names):
            m._buffers[n] = host[i].clone()
    for m in mods:
        for n in ("weight", "bias"):
            t = m._buffers.get(n)
            if t is None:
                continue
            if not t.is_contiguous():
                t = t.contiguous()
                m._buffers[n] = t
            dist.broadcast(t, src=src, group=group)
            nbytes += t.numel() * t.element_size()
        if hasattr(m, "_scol_cache"):
            m._scol_cache = None
    return nbytes


def buffers_fingerprint(root):
    """Order-dependent 64-bit fingerprint of all quantised buffers (equal on every rank after
    the broadcast).  Computed with integer tensor ops on the buffers' own device."""
    acc = 0
    for m in _w8a8_modules(root):
        for n in ("weight", "bias") + tuple(m._host_scalars):
            t = m._buffers.get(n)
            if t is None:
                continue
            b = t.detach().contiguous().reshape(-1).view(torch.uint8).to(torch.int64)
            idx = torch.arange(1, b.numel() + 1, device=b.device, dtype=torch.int64)
            acc = (acc * 1000003 + int((b * (idx % 8191 + 1)).sum().item())) % (1 << 61)
    return acc


def all_ranks_equal(value, group=None, device=None):
    """True iff `value` (python int) is identical on every rank."""
    t = torch.tensor([value, -value], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0]) == value and int(t[1]) == -value
