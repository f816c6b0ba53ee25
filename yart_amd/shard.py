"""Sharding of the intersection path across the GPUs of one node.

Rays are independent and the scene is read-only, so the path shards with NO data-path
collective: the scene is replicated in every GPU's HBM, each rank (one process per GPU)
traces its own part.  Two partitionings, both pure index arithmetic:

  * image mode   -- the reference's 32x32 tile list (Renderer::render,
                    src/renderer.cpp:161-217, tileSize_ = 32 include/renderer.h:14), tile i
                    owned by rank i % world.  The reference hands the same list to 8 threads
                    through a LIFO queue (src/scheduler.cpp:46-72); with pixels independent the
                    assignment does not change any pixel.
  * batch mode   -- contiguous slices of a ray batch (`ray_slice`).

The single exchange the path has is the framebuffer gather at the end of image mode
(`gather_framebuffer`): every rank sends its tiles' float RGB pixels to rank 0 with ONE
`torch.distributed.gather` -- RCCL over xGMI when the tensors live on GPUs (backend "nccl"),
gloo in the CPU tests.  4K float RGB is 99.5 MB in total; a direct gather to the root uses
all inbound xGMI links of the root at once, where a ring would be bound by one link.
"""
import numpy as np

TILE = 32                      # include/renderer.h:14


def tile_list(width, height, tile=TILE):
    """The reference's tile enumeration, in its order (src/renderer.cpp:163-217):
    full tiles x-major, then the right-hand strip, the bottom strip, the corner.
    -> int32 [nTiles, 4] = x0, y0, x1, y1 (end exclusive)."""
    fx, fy = width // tile, height // tile
    out = []
    for i in range(fx):
        for j in range(fy):
            out.append((i * tile, j * tile, (i + 1) * tile, (j + 1) * tile))
    if width - fx * tile > 0:
        for j in range(fy):
            out.append((fx * tile, j * tile, width, (j + 1) * tile))
    if height - fy * tile > 0:
        for i in range(fx):
            out.append((i * tile, fy * tile, (i + 1) * tile, height))
    if width - fx * tile > 0 and height - fy * tile > 0:
        out.append((fx * tile, fy * tile, width, height))
    return np.asarray(out, np.int32).reshape(-1, 4)


def tiles_of_rank(n_tiles, rank, world):
    """Round-robin ownership: tile i belongs to rank i % world."""
    return np.arange(rank, n_tiles, world, dtype=np.int64)


def tile_pixels(tiles, ids, width):
    """Row-major pixel numbers (y * width + x) of the given tiles, tile by tile, rows first --
    the order TileTask::run visits them (src/renderer.cpp:146-153)."""
    parts = []
    for t in ids:
        x0, y0, x1, y1 = (int(v) for v in tiles[t])
        ys, xs = np.meshgrid(np.arange(y0, y1, dtype=np.int64), np.arange(x0, x1, dtype=np.int64), indexing="ij")
        parts.append((ys * width + xs).ravel())
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def rank_pixels(width, height, rank, world, tile=TILE):
    """Pixel numbers of rank `rank`'s tiles (tile i -> rank i % world), in TileTask::run's order."""
    tiles = tile_list(width, height, tile)
    return tile_pixels(tiles, tiles_of_rank(len(tiles), rank, world), width)


def ray_slice(n, rank, world):
    """Batch mode: contiguous [begin, end) of an n-ray batch for `rank`; sizes differ by at most 1."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def coherence_key(rays, lo, hi, grid_bits=7):
    """The library's sort key (sort.hip: Morton cell of the origin on a 2^grid_bits grid over [lo, hi], direction octant
    in the low three bits) as torch ops on whatever device `rays` [n, 8] lives on.  Strong scaling orders ONE
    batch by it once and cuts contiguous slices, `ray_slice`, one per rank: a slice of an ordered batch is ordered."""
    import torch
    lo = torch.as_tensor(lo, dtype=torch.float32, device=rays.device)
    hi = torch.as_tensor(hi, dtype=torch.float32, device=rays.device)
    cells = float(1 << grid_bits)
    q = ((rays[:, 0:3] - lo) / (hi - lo) * cells).clamp_(0, cells - 1).to(torch.int64)
    key = torch.zeros(rays.shape[0], dtype=torch.int64, device=rays.device)
    for bit in range(grid_bits):
        for ax in range(3):
            key |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax + 3)
    key |= (rays[:, 4] < 0).to(torch.int64) | ((rays[:, 5] < 0).to(torch.int64) << 1) | ((rays[:, 6] < 0).to(torch.int64) << 2)
    return key


def strong_slice(rays, lo, hi, rank, world):
    """This rank's rays of a batch every rank holds: -> (ray numbers of the slice in coherence order, keys of the slice)."""
    import torch
    key = coherence_key(rays, lo, hi)
    order = torch.argsort(key, stable=True)
    begin, end = ray_slice(rays.shape[0], rank, world)
    return order[begin:end], key[order[begin:end]]


def allreduce_u64(values, group=None):
    """Counters / checksums summed over the ranks, in place (int64 tensor): what `yart_comm_allreduce_u64` does over RCCL,
    through torch.distributed for ranks that share a GPU or run on CPU (gloo).  world == 1 needs no process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values


def gather_framebuffer(local_rgb, width, height, rank, world, tile=TILE, group=None):
    """local_rgb: float32 tensor [nLocalPixels, 3] in `tile_pixels(tiles, tiles_of_rank(..))` order
    (CPU tensor with gloo, GPU tensor with nccl/RCCL).  Returns the full [height, width, 3] frame
    on rank 0 (same device as the input) and None elsewhere.  One collective; world == 1 needs
    no process group at all."""
    import torch
    import torch.distributed as dist

    tiles = tile_list(width, height, tile)
    counts = [int(((tiles[tiles_of_rank(len(tiles), r, world)][:, 2:] -
                    tiles[tiles_of_rank(len(tiles), r, world)][:, :2]).prod(1)).sum()) for r in range(world)]
    if local_rgb.shape != (counts[rank], 3):
        raise ValueError("rank %d: expected [%d, 3] pixels, got %s" % (rank, counts[rank], tuple(local_rgb.shape)))
    if world == 1:
        parts = [local_rgb]
    else:
        # gather needs equal shapes: pad to the largest share (shares differ by < one tile row set)
        m = max(counts)
        send = torch.zeros((m, 3), dtype=torch.float32, device=local_rgb.device)
        send[:counts[rank]] = local_rgb
        recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0, group=group)
        if rank != 0:
            return None
        parts = [recv[r][:counts[r]] for r in range(world)]
    frame = torch.zeros((height * width, 3), dtype=torch.float32, device=local_rgb.device)
    for r in range(world):
        pix = torch.from_numpy(tile_pixels(tiles, tiles_of_rank(len(tiles), r, world), width)).to(local_rgb.device)
        frame[pix] = parts[r]
    return frame.view(height, width, 3)
