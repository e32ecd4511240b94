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


def _planes_worker(rank, world, port, w, h, to_root, q):
    """The exchanges of a tile-sharded realtime frame over host memory (pt_exchange_planes_host = the protocol of pt_realtime_frame's guide exchange and of pt_gather_stable_planes): planes
    of 4, 4, 8 and 16 bytes per pixel — depth, hit distance, motion vectors, a header-like plane."""
    import ctypes
    import rtxpt_amd as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = pt.shard_layout(w, h, rank, world); yy, xx = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
    full = _full_planes(w, h)
    mine = [np.full_like(a, 0xEE) for a in full]                      # poison outside the owned tiles
    for a, b in zip(mine, full): a[yy, xx] = b[yy, xx]
    sent = []

    def send(ptr, nbytes, peer):
        t = torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8).clone(); sent.append((peer, nbytes)); dist.send(t, dst=peer)

    def recv(ptr, nbytes, peer):
        t = torch.empty(nbytes, dtype=torch.uint8); dist.recv(t, src=peer); ctypes.memmove(ptr, t.data_ptr(), nbytes)
    pt.exchange_planes_host(w, h, rank, world, mine, send, recv, to_root=to_root)
    rec = 4 + 4 + 8 + 16
    assert sorted(sent) == ([(0, rec * px.size)] if (to_root and rank) else [] if to_root else [(p, rec * px.size) for p in range(world) if p != rank])      # its own records, un-padded
    q.put((rank, mine))
    dist.barrier()
    dist.destroy_process_group()


def _full_planes(w, h):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    depth = (0.5 + xx * 0.001 + yy * 0.37).astype(np.float32); hit = (xx * 3.0 - yy).astype(np.float32)
    mv = np.stack([(xx * 31 + yy) & 0xFFFF, (yy * 17 + 5) & 0xFFFF, xx & 0xFF, yy & 0xFF], -1).astype(np.uint16)
    hdr = np.stack([xx * 7919 + yy, xx ^ yy, xx + 1, yy + 2], -1).astype(np.uint32)
    return [np.ascontiguousarray(a) for a in (depth, hit, mv, hdr)]


@pytest.mark.parametrize("size,world,to_root", [((100, 70), 2, False), ((200, 120), 3, False), ((97, 61), 3, True)])
def test_plane_exchange_through_the_c_entry_point(size, world, to_root):
    w, h = size
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_planes_worker, args=(r, world, port, w, h, to_root, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    full = _full_planes(w, h)
    import rtxpt_amd as pt
    for rank, planes in got:
        if to_root and rank:      # a sender keeps what it had: its own tiles, poison elsewhere
            px = pt.shard_layout(w, h, rank, world); yy, xx = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
            for a, b in zip(planes, full):
                want = np.full_like(b, 0xEE); want[yy, xx] = b[yy, xx]; assert np.array_equal(a, want), "rank %d" % rank
        else:
            for a, b in zip(planes, full): assert np.array_equal(a, b), "rank %d" % rank
