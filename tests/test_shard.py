"""Multi-GPU partitioning on CPU: the tile list is the reference's, every pixel /
ray has exactly one owner, and the world_size-2 framebuffer gather (gloo here, RCCL on GPUs)
reassembles exactly the frame a single rank produces."""
import os
import socket

import numpy as np
import pytest

from yart_amd import shard

torch = pytest.importorskip("torch")


def test_tile_list_follows_reference_enumeration():
    # src/renderer.cpp:163-217 with tileSize_ = 32: 1024x768 -> 32 x 24 full tiles, x-major
    t = shard.tile_list(1024, 768)
    assert len(t) == 32 * 24
    assert tuple(t[0]) == (0, 0, 32, 32) and tuple(t[1]) == (0, 32, 32, 64) and tuple(t[24]) == (32, 0, 64, 32)
    # ragged: right strip, bottom strip, corner, in that order
    t = shard.tile_list(70, 40)
    assert [tuple(r) for r in t] == [(0, 0, 32, 32), (32, 0, 64, 32), (64, 0, 70, 32),
                                     (0, 32, 32, 40), (32, 32, 64, 40), (64, 32, 70, 40)]
    assert len(shard.tile_list(16, 16)) == 1 and tuple(shard.tile_list(16, 16)[0]) == (0, 0, 16, 16)
    assert len(shard.tile_list(0, 0)) == 0


@pytest.mark.parametrize("w,h,world", [(70, 40, 1), (70, 40, 2), (1920, 1080, 8), (33, 33, 3), (5, 5, 8)])
def test_every_pixel_has_exactly_one_owner(w, h, world):
    tiles = shard.tile_list(w, h)
    seen = np.zeros(w * h, np.int32)
    for r in range(world):
        pix = shard.tile_pixels(tiles, shard.tiles_of_rank(len(tiles), r, world), w)
        np.add.at(seen, pix, 1)
    assert (seen == 1).all()


@pytest.mark.parametrize("n,world", [(0, 4), (7, 8), (16, 8), (1 << 24, 8), (1000003, 7)])
def test_ray_slices_partition_the_batch(n, world):
    edges = [shard.ray_slice(n, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    sizes = [e - b for b, e in edges]
    assert max(sizes) - min(sizes) <= 1


def _pixel_value(pix):
    """Stand-in for a rendered pixel: any deterministic function of the pixel number."""
    p = pix.astype(np.float64)
    return np.stack([np.sin(p * 0.37), p * 1e-3, np.cos(p * 0.11) * 7], 1).astype(np.float32)


def _worker(rank, world, port, w, h, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = shard.tile_list(w, h)
    pix = shard.tile_pixels(tiles, shard.tiles_of_rank(len(tiles), rank, world), w)
    local = torch.from_numpy(_pixel_value(pix))
    frame = shard.gather_framebuffer(local, w, h, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(70, 40), (128, 96)])
def test_gloo_world2_gather_equals_single_rank(tmp_path, w, h):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    got = np.load(out)
    want = _pixel_value(np.arange(w * h)).reshape(h, w, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # world == 1 needs no process group and gives the same frame
    tiles = shard.tile_list(w, h)
    pix = shard.tile_pixels(tiles, shard.tiles_of_rank(len(tiles), 0, 1), w)
    one = shard.gather_framebuffer(torch.from_numpy(_pixel_value(pix)), w, h, 0, 1).numpy()
    assert np.array_equal(one.view(np.uint32), want.view(np.uint32))


def _strong_worker(rank, world, port, n, out_path):
    import torch.distributed as dist
    from yart_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays = torch.from_numpy(scenes.random_rays(n, seed=5))          # the ONE batch, the same on every rank
    ids, keys = shard.strong_slice(rays, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], rank, world)
    assert bool((keys[1:] >= keys[:-1]).all())                      # a slice of the ordered batch is ordered
    sums = torch.tensor([ids.numel(), int(ids.sum()), int((ids * ids % 1000003).sum()), rank + 1], dtype=torch.int64)
    shard.allreduce_u64(sums)
    edges = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(edges, torch.stack([keys.min(), keys.max()]))
    if rank == 0:
        np.save(out_path, np.concatenate([sums.numpy(), torch.stack(edges).numpy().ravel()]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_strong_scaling_slices_and_counter_reduction(tmp_path):
    """bench.py --scaling strong on two ranks, rehearsed on CPU: every ray of the one batch belongs to exactly one rank
    (count, sum and a checksum of the ray numbers reduced over the ranks, the job of yart_comm_allreduce_u64 on GPUs),
    and the slices are contiguous pieces of the coherence order (rank 0's keys end where rank 1's begin)."""
    import torch.multiprocessing as mp
    n, world = 100003, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "strong.npy")
    mp.spawn(_strong_worker, args=(world, port, n, out), nprocs=world, join=True)
    got = np.load(out)
    ids = np.arange(n, dtype=np.int64)
    assert got[:4].tolist() == [n, int(ids.sum()), int((ids * ids % 1000003).sum()), 3]
    assert got[4 + 1] <= got[4 + 2]                                 # max key of rank 0 <= min key of rank 1
    # world == 1: the whole batch, no process group
    rays = torch.from_numpy(__import__("yart_amd.scenes", fromlist=["x"]).random_rays(1000, seed=1))
    one, keys = shard.strong_slice(rays, [0.0] * 3, [1.0] * 3, 0, 1)
    assert sorted(one.tolist()) == list(range(1000)) and shard.allreduce_u64(torch.tensor([7])).tolist() == [7]
