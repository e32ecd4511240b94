"""Multi-GPU plumbing (new; the reference is single-GPU): one process per GPU, frames sharded by 32x32
pixel tiles, tile t -> rank (position of t in Morton order) % world_size; every rank holds a full scene/BVH replica and
traces only its tiles; ONE collective per frame — a gather of the packed RGBA32F tiles to rank 0 (RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests). The RNG is keyed on absolute pixel coordinates + sample index, so the image is
bit-identical for any world size.

`shard_pixels` mirrors build_shards() in rtxpt_amd/csrc/pt_api.hip exactly (same tile order, same in-tile 8x8 block order).
"""
import numpy as np

TILE = 32


def _part1by1(v):
    v = v & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def shard_pixels(width, height, rank, world):
    """Packed pixel ids (x<<16|y) owned by `rank`, in the library's pack order."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    tiles = sorted(((_part1by1(x) | (_part1by1(y) << 1)), y * tx + x) for y in range(ty) for x in range(tx))
    by, bx = np.meshgrid(np.arange(0, TILE, 8), np.arange(0, TILE, 8), indexing="ij")
    yy, xx = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    # order: block rows (by), block cols (bx), then y, x inside the block
    ox = (bx.reshape(-1, 1, 1) + xx[None]).reshape(-1)
    oy = (by.reshape(-1, 1, 1) + yy[None]).reshape(-1)
    out = []
    for order, (_, t) in enumerate(tiles):
        if order % world != rank:
            continue
        px = (t % tx) * TILE + ox
        py = (t // tx) * TILE + oy
        ok = (px < width) & (py < height)
        out.append(((px[ok].astype(np.uint32) << 16) | py[ok].astype(np.uint32)))
    return np.concatenate(out) if out else np.zeros(0, np.uint32)


def gather_packed(packed, rank, world, dist, counts):
    """Gather per-rank packed RGBA32F tile buffers (torch tensors, (n_r, 4) float32) to rank 0.
    Returns the list of tensors on rank 0 (None elsewhere). Buffers are padded to the largest shard: one gather, no all-to-all."""
    import torch
    nmax = max(counts)
    send = torch.zeros((nmax, 4), dtype=torch.float32, device=packed.device)
    send[: packed.shape[0]] = packed
    if rank == 0:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=0)
        if send.is_cuda:      # the collective only orders torch's stream; pt_unpack_shard reads these buffers on the library's own stream
            torch.cuda.current_stream(send.device).synchronize()
        return [recv[r][: counts[r]] for r in range(world)]
    dist.gather(send, gather_list=None, dst=0)
    return None


def assemble(width, height, world, shards):
    """numpy reference of pt_unpack_shard: scatter packed shards into a full (h, w, 4) frame."""
    img = np.zeros((height, width, 4), np.float32)
    for r in range(world):
        px = shard_pixels(width, height, r, world)
        img[(px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)] = shards[r]
    return img
