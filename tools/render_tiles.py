#!/usr/bin/env python3
"""BASELINE.json configs[4] / SURVEY.md 8e: image tiles sharded across the GPUs of one node, scene
replicated, framebuffer gathered once at the end -- all of it through the C-ABI.

    python tools/render_tiles.py [--tris N --width W --height H --spp S --depth D --verify]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/render_tiles.py ...

One process per rank.  Each rank owns the reference's 32x32 tiles i with i % world == rank
(yart_tiles_rank_pixels; tile list of Renderer::render, src/renderer.cpp:161-217), renders ITS pixels with
the device-resident wavefront driver (yart_wavefront_render_diffuse: diffuse path tracing, tile-major
output buffer) and the parts are gathered on rank 0:
  * every rank has a GPU of its own: yart_frame_gather over an RCCL communicator (yart_comm_*), whose
    id travels through torch.distributed's store;
  * ranks share a GPU (rehearsal on a one-GPU box; RCCL refuses two ranks on one device): one
    torch.distributed.gather over gloo (yart_amd/shard.py).
Pixels do not depend on which rank traced them: --verify renders the whole frame on rank 0 as well and
compares bit for bit, and every rank re-renders a seeded sample of ITS pixels with the per-pixel CPU loop
over the oracle (oracle/diffuse_ref.py) and compares.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LIGHT, INTENSITY, ALBEDO = [0.0, 9.0, 0.0], [60.0, 55.0, 50.0], [0.5, 0.5, 0.5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--slots", type=int, default=1 << 23,
                    help="path slots = rays per batch and rank; 2^23 keeps every rank of an 8-way split in the throughput regime (DESIGN.md 5)")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--oracle-pixels", type=int, default=512, help="pixels per rank re-rendered on the CPU with --verify")
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

    W, H = args.width, args.height
    verts, tris, sph = scenes.architectural(args.tris, seed=1)
    t0 = time.time()
    acc = api.Accel(verts, tris, sph, device=device)
    acc.set_surface_data(verts, tris, np.zeros_like(verts), sph)
    build_s = time.time() - t0
    cam = api.camera_basis([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], W, H, 0.785398)
    wf = api.Wavefront(acc, args.slots)
    job = (cam, W, H, args.spp, args.depth, LIGHT, INTENSITY, ALBEDO)

    comm = None
    if own_gpu or world == 1:            # RCCL communicator of the C-ABI; its id goes through torch's store
        box = [api.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        comm = api.Comm(box[0], rank, world, device=device)

    pix = api.tiles_rank_pixels(W, H, rank, world)
    d_pix = torch.from_numpy(pix.astype(np.int32)).cuda()
    wf.render_diffuse(cam, W, H, 1, args.depth, LIGHT, INTENSITY, ALBEDO, seed=1, pixels=d_pix)      # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local, st, _ = wf.render_diffuse(*job, seed=1, pixels=d_pix)
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    t1 = time.perf_counter()
    if comm is not None:
        frame = comm.gather_frame(local, W, H)
        torch.cuda.synchronize()
    else:
        frame = shard.gather_framebuffer(local.cpu(), W, H, rank, world)
    if world > 1:
        dist.barrier()
    t_gather = time.perf_counter() - t1

    # ---- every rank: a seeded sample of its pixels against the CPU loop over the oracle
    oracle_ok, oracle_n, oracle_diff = True, 0, 0
    if args.verify and args.oracle_pixels > 0:
        from oracle import oracle as orc              # --verify only: the checker
        from oracle import diffuse_ref as dr
        o = orc.Oracle(verts, tris, sph)
        nrm = dr.Normals(o, verts, tris, sph)
        rng = np.random.RandomState(100 + rank)
        sel = np.sort(rng.choice(len(pix), size=min(args.oracle_pixels, len(pix)), replace=False))
        stats = {}
        want = dr.render(o, nrm, cam, W, H, args.spp, args.depth, LIGHT, INTENSITY, ALBEDO, seed=1, pixels=pix[sel], nthreads=8, stats=stats)
        got = local[torch.from_numpy(sel).cuda()].cpu().numpy()
        oracle_n = len(sel)
        oracle_diff = int((got.view(np.uint32) != want.view(np.uint32)).any(1).sum())
        rmse = float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)))
        # the closest and shadow rays the CPU loop sent, answered by the engine: the parity bar of SURVEY 8d
        rays, hits = stats["closest"][min(1, len(stats["closest"]) - 1)]
        g = api.hits_to_numpy(acc.intersect_closest(torch.from_numpy(rays).cuda()))
        sh, occ = stats["shadow"][0]
        m = hits["prim"] != -1
        rel = np.abs(g["t"][m] - hits["t"][m]) / np.abs(hits["t"][m])
        oracle_ok = bool(rmse <= 1e-4 and oracle_diff <= max(1, oracle_n // 500)
                         and np.array_equal(g["prim"] != -1, m) and (g["prim"] == hits["prim"]).mean() >= 0.9999
                         and (rel.size == 0 or rel.max() <= 1e-5)
                         and np.array_equal(acc.intersect_any(torch.from_numpy(sh).cuda()).cpu().numpy(), occ))

    rays_mine = float(st["closest_rays"] + st["shadow_rays"])
    engine_s = (st["ms_closest"] + st["ms_shadow"]) * 1e-3
    vals = torch.tensor([t_render, engine_s, rays_mine, float(oracle_n), float(oracle_diff), 0.0 if oracle_ok else 1.0], dtype=torch.float64)
    if world > 1:
        s = vals.cuda() if own_gpu else vals
        vmax, vsum = s.clone(), s.clone()
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        vmax, vsum = vmax.cpu(), vsum.cpu()
    else:
        vmax, vsum = vals, vals
    if rank == 0:
        frame_cpu = frame.cpu()
        n_total = float(vsum[2])
        out = {"config": "C5 image tiles: %dx%d, %d spp, depth %d diffuse path tracing, %d-triangle stand-in, 32x32 tiles dealt tile %% %d"
                         % (W, H, args.spp, args.depth, acc.num_triangles, world),
               "n_ranks": world,
               "gather": "yart_frame_gather over RCCL" if (comm is not None and world > 1) else
                         ("yart_frame_gather, one rank (local scatter)" if comm is not None else "torch.distributed.gather over gloo (ranks share a GPU)"),
               "rays_total": int(n_total), "render_s_max_over_ranks": round(float(vmax[0]), 4),
               "engine_s_max_over_ranks": round(float(vmax[1]), 4),
               "Mrays_s_end_to_end": round(n_total / float(vmax[0]) / 1e6, 1),
               "Mrays_s_engine": round(n_total / max(1e-9, float(vmax[1])) / 1e6, 1),
               "gather_ms": round(t_gather * 1e3, 2), "framebuffer_MB": round(frame_cpu.numel() * 4 / 1e6, 1),
               "build_s": round(build_s, 1), "frame_sum": float(frame_cpu.double().sum()),
               "lit_fraction": float((frame_cpu.sum(2) > 0).float().mean())}
        if args.verify:
            full, _, _ = wf.render_diffuse(*job, seed=1)
            out["sharded_equals_single_rank_bitwise"] = bool(torch.equal(full.view(H, W, 3).cpu().view(torch.int32), frame_cpu.view(torch.int32)))
            out["oracle_check"] = {"pixels_per_rank_rerendered_on_cpu": int(vsum[3]) // world, "pixels_differing": int(vsum[4]),
                                   "all_ranks_within_parity_bar": bool(float(vsum[5]) == 0.0)}
        if args.out:
            np.save(args.out, frame_cpu.numpy())
        if args.bmp:
            api.film_write_bmp(args.bmp, frame_cpu.numpy(), W, H)
            out["bmp"] = args.bmp
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
