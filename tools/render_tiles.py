#!/usr/bin/env python3
"""BASELINE.json configs[4] / SURVEY.md 8e: image tiles sharded across the GPUs of one node, scene
replicated, framebuffer gathered once at the end.

    python tools/render_tiles.py [--tris N --width W --height H --spp S --verify]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/render_tiles.py ...

One process per rank.  Each rank owns the reference's 32x32 tiles i with i % world == rank
(yart_amd/shard.py; tile list of Renderer::render, src/renderer.cpp:161-217), traces, per sample,
one closest-hit batch and one shadow batch for ITS pixels through the C-ABI, shades them (direct
light from a point light, Lambertian 0.5 -- torch plumbing, not the product) and sends its pixels to
rank 0 in one collective.  Backend: RCCL ("nccl") when every rank has a GPU of its own, gloo when
ranks share a GPU (rehearsal on a one-GPU box).  Pixels do not depend on which rank traced them:
--verify renders the whole frame on rank 0 as well and demands a bit-identical framebuffer.
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPS = 1e-4


def jitter_t(pix, sample, axis):
    """Per (pixel, sample) film offset in [-0.5, 0.5): a hash of the pixel number, so that a pixel gets
    the same samples whichever rank renders it (the reference seeds Rng(y*width+x), src/renderer.cpp:33)."""
    import torch
    m = 0xFFFFFFFF
    h = (pix * 2654435761 + (sample * 40503 + axis * 9973 + 12345)) & m
    h = h ^ (h >> 15)
    h = (h * 2246822519) & m
    h = h ^ (h >> 13)
    return (h.double() / 4294967296.0 - 0.5).float()


def camera_rays_t(pos, direction, width, height, fov, xs, ys):
    """Camera::Camera + Camera::sample (include/camera.h:9-31) in fp32 on the device."""
    import torch
    dev = xs.device
    f = torch.float32
    pos = torch.tensor(pos, dtype=f, device=dev)
    d = torch.tensor(direction, dtype=f, device=dev)
    d = d * (1.0 / d.norm())
    up = torch.tensor([0.0, 1.0, 0.0], dtype=f, device=dev)
    right = torch.linalg.cross(d, up) * (float(width) * fov / float(height))
    upv = torch.linalg.cross(right, d)
    upv = upv * (1.0 / upv.norm()) * fov
    a = (xs + 0.5) / float(width) - 0.5
    b = -(ys + 0.5) / float(height) + 0.5
    dd = right[None] * a[:, None] + upv[None] * b[:, None] + d[None]
    dd = dd * (1.0 / dd.norm(dim=1, keepdim=True))
    rays = torch.empty((len(xs), 8), dtype=f, device=dev)
    rays[:, 0:3] = pos
    rays[:, 3] = 0
    rays[:, 4:7] = dd
    rays[:, 7] = float("inf")
    return rays


def render_pixels(acc, ta, ta_k, pix, width, height, spp, cam, light):
    """-> float32 [len(pix), 3] on the accel's device, ray count, seconds inside the engine."""
    import torch
    from yart_amd import api, scenes
    dev = torch.device("cuda", acc.device)
    rgb = torch.zeros((len(pix), 3), dtype=torch.float32, device=dev)
    n_rays, t_engine = 0, 0.0
    pix_t = torch.from_numpy(pix).to(dev)
    xs = (pix_t % width).float()
    ys = (pix_t // width).float()
    for s in range(spp):
        rays = camera_rays_t(cam[0], cam[1], width, height, 0.785398,
                             xs + jitter_t(pix_t, s, 0), ys + jitter_t(pix_t, s, 1))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        hits = acc.intersect_closest(rays)
        torch.cuda.synchronize(dev)
        t_engine += time.perf_counter() - t0
        prim = hits[:, 3].contiguous().view(torch.int32)
        hit = prim >= 0
        pid = prim.clamp(min=0).long()
        P = rays[:, 0:3] + rays[:, 4:7] * hits[:, 0:1]
        k = ta_k[pid]
        nrm = torch.zeros_like(P)
        nrm.scatter_(1, k[:, None], 1.0)
        nrm.scatter_(1, ((k + 1) % 3)[:, None], ta[pid, 0:1])
        nrm.scatter_(1, ((k + 2) % 3)[:, None], ta[pid, 1:2])
        nrm = nrm / nrm.norm(dim=1, keepdim=True)
        nl = torch.where((nrm * rays[:, 4:7]).sum(1, keepdim=True) < 0, nrm, -nrm)
        to_l = light[None] - P
        dist = to_l.norm(dim=1, keepdim=True)
        wi = to_l / dist
        sh = torch.cat([P + wi * EPS, torch.zeros_like(dist), wi, dist], 1)
        sh = torch.where(hit[:, None], sh, torch.tensor([0, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device=dev)[None]).contiguous()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        occ = acc.intersect_any(sh)
        torch.cuda.synchronize(dev)
        t_engine += time.perf_counter() - t0
        n_rays += 2 * len(pix)
        lit = hit & (occ == 0)
        radiance = 0.5 / np.pi * 700.0 * (nl * wi).sum(1).abs() / (dist[:, 0] * dist[:, 0])
        rgb += torch.where(lit, radiance, torch.zeros_like(radiance))[:, None] * (1.0 / spp)
    return rgb, n_rays, t_engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--bmp", default="", help="write the gathered frame as a BMP (yart_film_write_bmp)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from yart_amd import api, scenes, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    own_gpu = n_dev >= world
    device = local_rank if own_gpu else local_rank % max(1, n_dev)
    torch.cuda.set_device(device)
    if world > 1:
        if own_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    verts, tris, sph = scenes.architectural(args.tris, seed=1)
    t0 = time.time()
    acc = api.Accel(verts, tris, sph, device=device)
    build_s = time.time() - t0
    dev = torch.device("cuda", device)
    ta_np = acc.triaccel()
    ta = torch.from_numpy(ta_np.view(np.float32).reshape(-1, 12).copy()).to(dev)
    ta_k = torch.from_numpy(ta_np["k"].astype(np.int64)).to(dev)
    cam = ([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05])
    light = torch.tensor([0.0, 9.0, 0.0], device=dev)

    tiles = shard.tile_list(args.width, args.height)
    mine = shard.tiles_of_rank(len(tiles), rank, world)
    pix = shard.tile_pixels(tiles, mine, args.width)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local, n_rays, t_engine = render_pixels(acc, ta, ta_k, pix, args.width, args.height, args.spp, cam, light)
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    t1 = time.perf_counter()
    payload = local if (own_gpu or world == 1) else local.cpu()
    frame = shard.gather_framebuffer(payload, args.width, args.height, rank, world)
    if world > 1:
        dist.barrier()
    t_gather = time.perf_counter() - t1
    stats = torch.tensor([t_render, t_engine, float(n_rays)], dtype=torch.float64)
    if world > 1:
        s = stats.to(dev) if own_gpu else stats
        tmax = s.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = s.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        t_render, t_engine, n_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
    else:
        n_total = float(n_rays)
    if rank == 0:
        frame_cpu = frame.cpu()
        out = {"config": "C5 image tiles: %dx%d, %d spp (1 closest + 1 shadow ray per sample), %d-triangle stand-in, "
                         "32x32 tiles dealt tile %% %d" % (args.width, args.height, args.spp, acc.num_triangles, world),
               "n_ranks": world, "backend": "none" if world == 1 else ("nccl" if own_gpu else "gloo (ranks share a GPU)"),
               "rays_total": int(n_total), "render_s_max_over_ranks": round(t_render, 4),
               "engine_s_max_over_ranks": round(t_engine, 4),
               "Mrays_s_engine": round(n_total / t_engine / 1e6, 1), "Mrays_s_incl_raygen_and_shading": round(n_total / t_render / 1e6, 1),
               "gather_ms": round(t_gather * 1e3, 2), "framebuffer_MB": round(frame_cpu.numel() * 4 / 1e6, 1),
               "build_s": round(build_s, 1), "frame_sum": float(frame_cpu.double().sum()),
               "lit_fraction": float((frame_cpu.sum(2) > 0).float().mean())}
        if args.verify:
            allpix = np.arange(args.width * args.height, dtype=np.int64)
            full, _, _ = render_pixels(acc, ta, ta_k, allpix, args.width, args.height, args.spp, cam, light)
            out["sharded_equals_single_rank_bitwise"] = bool(torch.equal(full.view(args.height, args.width, 3).cpu().view(torch.int32),
                                                                         frame_cpu.view(torch.int32)))
        if args.out:
            np.save(args.out, frame_cpu.numpy())
        if args.bmp:
            api.film_write_bmp(args.bmp, frame_cpu.numpy(), args.width, args.height)
            out["bmp"] = args.bmp
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
