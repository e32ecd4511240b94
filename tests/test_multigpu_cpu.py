"""N>1 path on CPU: world_size-2 gloo processes shard a frame by pixel tiles and gather it on rank 0."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtxpt_amd import parallel  # noqa: E402


def fake_radiance(px):
    x = (px >> 16).astype(np.float32); y = (px & 0xFFFF).astype(np.float32)
    return np.stack([x * 0.5 + y, x - y * 0.25, x * y * 1e-3, np.ones_like(x)], 1).astype(np.float32)


def _worker(rank, world, port, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = parallel.shard_pixels(w, h, rank, world)
    counts = [parallel.shard_pixels(w, h, r, world).size for r in range(world)]
    packed = torch.from_numpy(fake_radiance(px))
    got = parallel.gather_packed(packed, rank, world, dist, counts)
    if rank == 0:
        img = parallel.assemble(w, h, world, [g.numpy() for g in got])
        q.put(img)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("size", [(100, 70), (256, 144)])
def test_tile_shard_gather_world2(size):
    w, h = size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    full = fake_radiance(((xx.astype(np.uint32) << 16) | yy.astype(np.uint32)).reshape(-1)).reshape(h, w, 4)
    assert np.array_equal(img, full)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_partition_the_frame(world):
    w, h = 200, 120
    seen = np.zeros((h, w), np.int32)
    sizes = []
    for r in range(world):
        px = parallel.shard_pixels(w, h, r, world)
        seen[(px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)] += 1
        sizes.append(px.size)
    assert np.all(seen == 1)
    assert max(sizes) - min(sizes) <= 2 * 32 * 32          # interleaved tiles balance the pixel counts
