"""GPU (-m gpu), 2 gloo ranks SHARING the one device of the gpurun box: the arena broadcast followed by forwards on offset operand images.
A module that held an image of ITS OWN weight before the broadcast must multiply rank 0's weight afterwards: the image cache is keyed on the weight's
storage, QuantArena.pack re-homes the weight, so the receiver rebuilds the image from the received bytes.  RCCL itself needs one
device per rank (tests/test_replica_nccl.py, self-skipping here); the code path above the collective is the same."""
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


def _module(seed, dev):
    from autosmoothquant_amd.layers.nn.linear import W8A8BFP32OFP32Linear
    g = torch.Generator().manual_seed(seed)
    m = W8A8BFP32OFP32Linear(4096, 4096, False, "per-tensor")
    m.weight = torch.randint(-100, 100, (4096, 4096), generator=g, dtype=torch.int8)
    m.dequant_scale = torch.tensor(1e-4 * (seed + 1))
    return torch.nn.ModuleDict({"m": m}).to(dev)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autosmoothquant_amd import replica, ops
        root = _module(0 if rank == 0 else 3, dev)
        m = root["m"]
        x = torch.randint(-3, 4, (2304, 4096), generator=torch.Generator().manual_seed(1)).half().to(dev)
        y_own = m(x)                                   # builds the image of this rank's own weight
        had_image = m.__dict__.get("_offset_cache") is not None
        replica.broadcast_quantized(root, src=0, device=dev)
        y = m(x)                                       # must run on an image of the RECEIVED weight
        img = m.offset_image(2304, torch.float16)
        fresh = ops.weight_offset_image(m.weight)
        image_ok = img is not None and torch.equal(img[0], fresh[0]) and torch.equal(img[1], fresh[1])
        m.offsets = False
        plain_ok = torch.equal(y, m(x))
        q.put((rank, had_image, image_ok, plain_ok, y.cpu(), bool(torch.equal(y, y_own))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_receiver_rebuilds_its_offset_image_after_the_broadcast():
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
    for rank, had_image, image_ok, plain_ok, y, same_as_own in res:
        assert had_image and image_ok and plain_ok
        assert torch.equal(y, res[0][4])               # every replica computes rank 0's rows
        assert same_as_own == (rank == 0)              # ... and rank 1's own weight gave something else before
