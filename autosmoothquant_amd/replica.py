"""Replica-parallel forwards: one process per GPU, every rank holds a full copy of the
quantised buffers and processes its own rows.  The ONLY collective traffic is a one-time
broadcast of {int8 / fp8 weights, fp32 biases, fp32 scalar scales} from rank 0 at load time
(``torch.distributed`` backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
The reference has no distributed code at all; the correctness criterion is
"every replica's rows equal the 1-GPU rows bit for bit" -- rows of an Int8Linear are
independent (per-token scales are per row, per-tensor has no cross-row state).

How the broadcast is shaped for xGMI.  An MI355X node is a full mesh of
point-to-point links (7 x ~153 GB/s per GPU), not a switch: one ``broadcast`` per buffer
(~450 collectives for LLaMA-7B) runs as a ring bound by ONE link.  Instead

  1. every quantised buffer of every module (W8A8 and FP8 classes alike) is re-homed into ONE
     flat, 256-B aligned device arena (`QuantArena`; the module buffers become views of it, so
     packing is a one-time copy on the source and zero-copy on the receivers);
  2. the arena is cut into `world` equal chunks; the source SCATTERS chunk r to rank r -- G-1
     different links of the source carry 1/G of the payload each, concurrently;
  3. one in-place ALL-GATHER completes every rank's arena -- every link of the mesh busy,
     (G-1)/G of the payload in and out per GPU.

Two collectives in total, time ~ payload / G / link + payload (G-1)/G / (links x link) instead
of payload / link.  The host-pinned scalar scales travel in the arena's tail.
"""
import hashlib

import torch
import torch.distributed as dist

_ALIGN = 256
XGMI_LINK_GBPS = 153.0  # per direction, per link; 7 links per GPU


def shard_rows(M, world_size, rank):
    """Contiguous row range of `rank`: rows[g] = M*g/G ... M*(g+1)/G."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (M * rank) // world_size, (M * (rank + 1)) // world_size


def _quant_modules(root):
    from .layers.nn.linear import _W8A8Base, _FP8Base
    return [m for m in root.modules() if isinstance(m, (_W8A8Base, _FP8Base))]


def _pad(n, a=_ALIGN):
    return (n + a - 1) // a * a


class QuantArena:
    """Flat layout of every quantised buffer under `root`: (module index, buffer name, dtype, shape,
    byte offset) for the device buffers, then one fp32 slot per host-pinned scalar scale."""

    def __init__(self, root, world=1):
        self.mods = _quant_modules(root)
        self.entries, off, sig = [], 0, hashlib.sha256()
        for i, m in enumerate(self.mods):
            for n in ("weight", "bias", "weight_scale_mx"):
                t = m._buffers.get(n)
                if t is None:
                    continue
                nb = t.numel() * t.element_size()
                self.entries.append((i, n, t.dtype, tuple(t.shape), off, nb))
                sig.update(f"{type(m).__name__}.{n}:{t.dtype}:{tuple(t.shape)}@{off};".encode())
                off += _pad(nb)
        self.scalars = [(i, n) for i, m in enumerate(self.mods) for n in m._host_scalars]
        sig.update(";".join(f"{type(self.mods[i]).__name__}.{n}" for i, n in self.scalars).encode())
        self.scalar_off = off
        off += _pad(4 * len(self.scalars))
        self.payload_bytes = sum(e[5] for e in self.entries) + 4 * len(self.scalars)
        self.total = _pad(max(off, _ALIGN), _ALIGN * max(world, 1))  # equal, aligned chunks
        self.signature = int.from_bytes(sig.digest()[:7], "big")
        self.buf = None

    def _view(self, e):
        _, _, dt, shape, off, nb = e
        return self.buf[off:off + nb].view(dt).view(shape)

    def pack(self, device, copy=True):
        """Allocate the arena on `device` and re-home the module buffers as views of it; with copy=True (the source rank) every
        buffer's contents are copied in first (receivers skip the copy: their arena is about to be overwritten)."""
        self.buf = torch.zeros(self.total, dtype=torch.uint8, device=device)
        for e in self.entries:
            m, n = self.mods[e[0]], e[1]
            v = self._view(e)
            if copy:
                v.copy_(m._buffers[n].detach().to(device))
            m._buffers[n] = v
        if self.scalars and copy:
            vals = torch.tensor([float(self.mods[i]._buffers[n]) for i, n in self.scalars], dtype=torch.float32)
            self.buf[self.scalar_off:self.scalar_off + 4 * len(self.scalars)].view(torch.float32).copy_(vals.to(device))
        return self

    def unpack_scalars(self):
        """Arena tail -> the modules' host-pinned fp32 scalar buffers (one device-to-host copy)."""
        if self.scalars:
            host = self.buf[self.scalar_off:self.scalar_off + 4 * len(self.scalars)].view(torch.float32).cpu()
            for k, (i, n) in enumerate(self.scalars):
                self.mods[i]._buffers[n] = host[k].clone()
        for m in self.mods:
            if hasattr(m, "_scol_cache"):
                m._scol_cache = None


def all_ranks_equal(value, group=None, device=None):
    """True iff `value` (python int, < 2^62) is identical on every rank."""
    t = torch.tensor([value, -value], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0]) == value and int(t[1]) == -value


def broadcast_arena(arena, src=0, group=None):
    """Scatter + all-gather of a packed arena (see the module docstring).  In place on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return
    chunk = arena.total // world
    views = [arena.buf[r * chunk:(r + 1) * chunk] for r in range(world)]
    mine = views[rank]
    # `src` is a GLOBAL rank (torch.distributed's convention for scatter / broadcast); `rank` and the chunk index are group ranks
    src_in_group = src if group is None else dist.get_group_rank(group, src)
    # 1. scatter: chunk r -> group rank r (the source keeps its own); G-1 links of the source in parallel
    dist.scatter(mine, scatter_list=views if rank == src_in_group else None, src=src, group=group)
    # 2. all-gather of the chunks into every arena (input staged: in-place aliasing is backend-specific)
    dist.all_gather_into_tensor(arena.buf, mine.clone(), group=group)


def broadcast_quantized(root, src=0, group=None, device=None, timing=None):
    """Make every rank's quantised buffers (W8A8 *and* FP8 modules) equal to rank `src`'s: flat-pack, scatter,
    all-gather.  Module buffers are views of the arena afterwards (kept alive by `root._asq_arena`).
    Returns the number of payload bytes (for GB/s reporting).  `timing` (a dict) receives the wall time of the three phases
    on this rank: "pack_s" (arena allocation + one copy of every buffer), "collective_s" (scatter + all-gather between
    barriers: the xGMI part), "unpack_s" (scalar scales back to the host)."""
    import time
    world = dist.get_world_size(group)
    arena = QuantArena(root, world)
    dev = device if device is not None else (arena.mods[0].weight.device if arena.mods else torch.device("cpu"))

    def tick():
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()
    # the signature collective runs on EVERY rank first -- also on one that found no quantised module (its empty arena hashes to a constant): a
    # rank that returned early here would leave the others blocked in the all-reduce instead of raising the mismatch
    if not all_ranks_equal(arena.signature, group=group, device=dev):
        raise RuntimeError("broadcast_quantized: ranks hold different module structures (class / buffer names / shapes); "
                           "every rank must build the same quantised model skeleton before the broadcast")
    if not arena.mods:
        return 0
    t0 = tick()
    src_in_group = src if group is None else dist.get_group_rank(group, src)
    arena.pack(dev, copy=dist.get_rank(group) == src_in_group)
    t1 = tick()
    dist.barrier(group=group)
    t1b = tick()
    broadcast_arena(arena, src=src, group=group)
    t2 = tick()
    arena.unpack_scalars()
    t3 = tick()
    if timing is not None:
        timing.update(pack_s=t1 - t0, collective_s=t2 - t1b, unpack_s=t3 - t2)
    try:
        root._asq_arena = arena
    except Exception:
        pass
    return arena.payload_bytes


def buffers_fingerprint(root):
    """Order-dependent 61-bit fingerprint of all quantised buffers (equal on every rank after
    the broadcast).  Computed with integer tensor ops on the buffers' own device."""
    acc = 0
    for m in _quant_modules(root):
        for n in ("weight", "bias", "weight_scale_mx") + tuple(m._host_scalars):
            t = m._buffers.get(n)
            if t is None:
                continue
            b = t.detach().contiguous().reshape(-1).view(torch.uint8).to(torch.int64)
            idx = torch.arange(1, b.numel() + 1, device=b.device, dtype=torch.int64)
            acc = (acc * 1000003 + int((b * (idx % 8191 + 1)).sum().item())) % (1 << 61)
    return acc


def gather_fingerprints(fp, group=None, device=None):
    """[fingerprint of rank 0, ..., of rank G-1] on every rank (the bench line prints them: one glance shows that every replica holds rank 0's buffers)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [int(fp)]
    mine = torch.tensor([int(fp)], dtype=torch.int64, device=device or "cpu")
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def broadcast_report(nbytes, seconds, world, timing=None):
    """The `weight_broadcast` block of the bench line: achieved GB/s of the collective phase next to one xGMI link and the
    scatter + all-gather model (payload/G over one link, then (G-1)/G of the payload over G-1 links)."""
    model_s = (nbytes / world / (XGMI_LINK_GBPS * 1e9) + nbytes * (world - 1) / world / ((world - 1) * XGMI_LINK_GBPS * 1e9)) if world > 1 else 0.0
    coll = (timing or {}).get("collective_s", seconds)
    out = {"bytes": nbytes, "ms_total_incl_pack": seconds * 1e3, "ms": coll * 1e3, "GBps": nbytes / coll / 1e9 if coll > 0 else None,
           "xgmi_link_GBps": XGMI_LINK_GBPS, "x_one_link": (nbytes / coll / 1e9 / XGMI_LINK_GBPS) if coll > 0 else None,
           "model_ms_scatter_allgather": model_s * 1e3, "collectives": "1 scatter + 1 all-gather over one flat arena" if world > 1 else "none"}
    if timing:
        out.update({k.replace("_s", "_ms"): v * 1e3 for k, v in timing.items()})
    return out
