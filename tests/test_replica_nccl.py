"""GPU (-m gpu), world_size 2 over RCCL: the arena broadcast (autosmoothquant_amd/replica.py: flat-pack, scatter, all-gather) on real
devices, W8A8 and FP8 modules, followed by 'every replica's rows == the single-process rows' with the HIP forward.
Self-skips on a box with fewer than 2 GPUs (the gpurun box has one; the driver's multi-GPU tier has eight)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed, dev):
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear, W8A8BFP32OFP32LinearWithQuantScale, FP8LinearStatic
    g = torch.Generator().manual_seed(seed)
    mods = torch.nn.ModuleDict()
    a = W8A8BFP32OFP32Linear(512, 384, True, "per-tensor")
    b = W8A8BFP32OFP32LinearWithQuantScale(384, 512, False, "per-token")
    c = FP8LinearStatic(512, 256, use_bias=True)
    for m in (a, b):
        m.weight = torch.randint(-128, 128, m.weight.shape, generator=g, dtype=torch.int8)
    a.bias = torch.randn(384, generator=g)
    a.dequant_scale, b.dequant_scale = torch.tensor(1e-3 * (seed + 1)), torch.tensor(2e-3 * (seed + 1))
    c.weight = torch.randint(0, 120, c.weight.shape, generator=g, dtype=torch.uint8).view(torch.float8_e4m3fn)
    c.bias = torch.randn(256, generator=g)
    c.weight_scale, c.input_scale, c.output_scale = torch.tensor(0.01 * (seed + 1)), torch.tensor(0.05), torch.tensor(0.0)
    mods["a"], mods["b"], mods["c"] = a, b, c
    return mods.to(dev)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from autosmoothquant_amd import replica
        mods = _model(0 if rank == 0 else 11, dev)
        differs = not replica.all_ranks_equal(replica.buffers_fingerprint(mods), device=dev)
        nbytes = replica.broadcast_quantized(mods, src=0, device=dev)
        same = replica.all_ranks_equal(replica.buffers_fingerprint(mods), device=dev)
        M = 70
        xg = (torch.randn(M, 512, generator=torch.Generator().manual_seed(3)) * 30).half()
        lo, hi = replica.shard_rows(M, world, rank)
        y = mods["a"](xg[lo:hi].to(dev)).cpu()
        y8 = mods["c"](xg[lo:hi].to(dev)).cpu()
        q.put((rank, differs, same, nbytes, lo, hi, y, y8))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_arena_broadcast_and_row_sharding_rccl_world2():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _model(0, torch.device("cuda", 0))
    xg = (torch.randn(70, 512, generator=torch.Generator().manual_seed(3)) * 30).half().to("cuda:0")
    want, want8 = ref["a"](xg).cpu(), ref["c"](xg).cpu()
    for rank, differs, same, nbytes, lo, hi, y, y8 in res:
        assert differs and same and nbytes > 0
        assert torch.equal(y, want[lo:hi]) and torch.equal(y8, want8[lo:hi])
