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
    """Each rank fills only its own tiles of a host frame, then the LIBRARY's gather protocol (pt_gather_host: the layout, packing, un-padded
    point-to-point transfers and unpacking pt_gather runs over RCCL) moves them to rank 0 over a gloo transport supplied as C callbacks."""
    import ctypes
    import rtxpt_amd as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = pt.shard_layout(w, h, rank, world)
    assert np.array_equal(px, parallel.shard_pixels(w, h, rank, world))          # the numpy mirror agrees with the library
    frame = np.full((h, w, 4), -7.0, np.float32)                                  # poison: nothing but the owned tiles is valid on this rank
    frame[(px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)] = fake_radiance(px)
    sent = []

    def send(ptr, nbytes, peer):
        t = torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8).clone()
        sent.append(nbytes); dist.send(t, dst=peer)

    def recv(ptr, nbytes, peer):
        t = torch.empty(nbytes, dtype=torch.uint8); dist.recv(t, src=peer)
        ctypes.memmove(ptr, t.data_ptr(), nbytes)
    pt.gather_host(w, h, rank, world, frame, send, recv)
    if rank != 0:
        assert sent == [16 * px.size]                                             # exactly the rank's own bytes: no padding to the largest shard
    if rank == 0:
        q.put(frame)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("size,world", [((100, 70), 2), ((256, 144), 2), ((200, 120), 3)])
def test_tile_shard_gather_through_the_c_entry_point(size, world):
    w, h = size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
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
    import rtxpt_amd as pt
    for r in range(world):
        px = parallel.shard_pixels(w, h, r, world)
        assert np.array_equal(px, pt.shard_layout(w, h, r, world))
        seen[(px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)] += 1
        sizes.append(px.size)
    assert np.all(seen == 1)
    assert max(sizes) - min(sizes) <= 2 * 32 * 32          # interleaved tiles balance the pixel counts


def _neeat_worker(rank, world, port, w, h, q):
    """NEE-AT on tile-sharded frames: between two frames every rank needs every rank's feedback reservoirs (pt_neeat_exchange_host = the protocol pt_render runs over RCCL)."""
    import ctypes
    import rtxpt_amd as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = pt.shard_layout(w, h, rank, world); yy, xx = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
    weight = np.full((h, w), -3.0, np.float32); cand = np.full((h, w), 0xDEADBEEF, np.uint32); depth = np.full((h, w), -7.0, np.float32)          # poison outside the owned tiles
    weight[yy, xx] = (xx * 0.25 + yy).astype(np.float32); cand[yy, xx] = (xx * 7919 + yy * 104729).astype(np.uint32) | np.uint32(0x80000000) * (xx & 1).astype(np.uint32)
    depth[yy, xx] = (0.5 + xx * 0.001 + yy * 0.37).astype(np.float32)
    sent = []

    def send(ptr, nbytes, peer):
        t = torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8).clone(); sent.append((peer, nbytes)); dist.send(t, dst=peer)

    def recv(ptr, nbytes, peer):
        t = torch.empty(nbytes, dtype=torch.uint8); dist.recv(t, src=peer); ctypes.memmove(ptr, t.data_ptr(), nbytes)
    pt.neeat_exchange_host(w, h, rank, world, weight, cand, depth, send, recv)
    assert sorted(sent) == [(p, 12 * px.size) for p in range(world) if p != rank]      # its own pixels (weight, candidate, depth), un-padded, once to every other rank
    q.put((rank, weight, cand, depth))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size,world", [((100, 70), 2), ((200, 120), 3)])
def test_neeat_feedback_exchange_through_the_c_entry_point(size, world):
    w, h = size
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_neeat_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    want_w = (xx * 0.25 + yy).astype(np.float32); want_c = (xx * 7919 + yy * 104729).astype(np.uint32) | np.uint32(0x80000000) * (xx & 1).astype(np.uint32)
    want_d = (0.5 + xx * 0.001 + yy * 0.37).astype(np.float32)
    for rank, weight, cand, depth in got:
        assert np.array_equal(weight, want_w) and np.array_equal(cand, want_c) and np.array_equal(depth, want_d), "rank %d" % rank
