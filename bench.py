#!/usr/bin/env python3
"""bench.py -- closest-hit throughput of the MI355X intersection engine.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- coherence sort + one traversal of the
4-wide triangle+sphere tree (yart_intersect_closest_dev) -- over one batch of synthetic rays
already resident in HBM.  Workload at N=1 (BASELINE.json configs[3], the config the
metric "Mrays/sec (closest-hit, incoherent paths)" is quoted on; it fits one GPU):
10 M-triangle random soup + 1 M spheres, 2^24 uniformly random rays (SURVEY.md 8d
C4).  With N ranks the scene is replicated on every GPU and every rank traces its own
batch of the same size (weak scaling, no data-path collective: rays are independent).

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(counts, wide, chained):
    """ray in (32 B) + hit out (16 B) + the records the engine's own traversal
    fetched, per ray and per kernel, from the counting kernel (include/yart_amd.h):
      default 4-wide tree : 48 B per node visit = 12 B per child record (four per visit; the packed node of
                            yart_amd/csrc/wide_build.h) + 16 B per triangle plane row + 32 B per second-stage
                            row + 16 B per sphere record
      reference tree      : 32 B per node record + 48 B per triangle record (+ sphere tree)"""
    c = counts.sum(0).astype(np.float64)
    n = float(len(counts))
    if wide:
        return n * (32 + 16) + 12.0 * c[0] + 16.0 * c[1] + 32.0 * c[2] + 16.0 * c[3], 0.0
    tri = n * (32 + 16 + (16 if chained else 0)) + 32.0 * c[0] + 48.0 * c[1]
    sph = (n * (32 + 16) + 32.0 * c[2] + 16.0 * c[3]) if chained else 0.0
    return tri, sph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--spheres", type=int, default=1_000_000)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--s", type=float, default=0.005, help="soup edge scale (SURVEY.md 8d sweeps 0.002/0.005/0.01)")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--static", action="store_true", help="plain grid launch instead of the persistent kernel")
    ap.add_argument("--reference-tree", action="store_true",
                    help="traverse the reference's binary tree (sphere pass + triangle pass) instead of the 4-wide tree")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget; 0 disables it")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, what the driver runs): --rays per GPU, every rank its own batch; strong: ONE batch of "
                         "--rays (default 2^26 then), ordered once by the coherence key, contiguous slices per rank (SURVEY 8e)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the s = 0.002 / 0.01 points of configs[3] (N = 1 only)")
    args = ap.parse_args()
    strong = args.scaling == "strong"
    if strong and "--rays" not in " ".join(sys.argv):
        args.rays = 1 << 26

    import torch
    import torch.distributed as dist
    from yart_amd import api, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    n_dev = torch.cuda.device_count()
    own_gpu = n_dev >= world                  # one GPU per rank: RCCL; ranks sharing a GPU (rehearsal on a one-GPU box): gloo
    device = local_rank if own_gpu else local_rank % n_dev
    torch.cuda.set_device(device)
    if world > 1:
        if own_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            log("warning: %d ranks on %d GPU(s): rehearsal over gloo, ranks share a device" % (world, n_dev))
            dist.init_process_group("gloo")

    t0 = time.time()
    verts, tris, sph = scenes.random_soup(args.tris, args.spheres, s=args.s, seed=42)
    rays = scenes.random_rays(args.rays, seed=7 if strong else 7 + rank)
    t1 = time.time()
    acc = api.Accel(verts, tris, sph, device=device)
    t2 = time.time()
    total_rays = args.rays * (1 if strong else world)
    if strong:
        # SURVEY 8e throughput mode: the batch every rank holds is ordered ONCE by the coherence key (Morton cell of the
        # origin on a 128^3 grid over the scene + direction octant, as sort.hip) and cut into contiguous slices, so a rank's
        # rays stay in one region of the scene; outside the timed region (the caller's batch arrives ordered).
        from yart_amd import shard
        d_all = torch.from_numpy(rays).cuda()
        order, _ = shard.strong_slice(d_all, verts.min(0), verts.max(0), rank, world)
        rays = d_all[order].cpu().numpy()
        del d_all, order
        torch.cuda.empty_cache()
        args.rays = len(rays)
    if rank == 0:
        log("scene: %d tris, %d spheres, %d nodes, depth %d, %.1f MB in HBM (gen %.1fs, build+upload %.1fs)"
            % (acc.num_triangles, acc.num_spheres, acc.num_nodes, acc.depth, acc.device_bytes / 1e6, t1 - t0, t2 - t1))
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((args.rays, 4), dtype=torch.float32, device="cuda")
    flags = 0
    if not args.no_sort and not strong:       # strong: the slice arrives in the coherence order already (ordered once, above)
        flags |= api.YART_SORT_RAYS
    if args.static:
        flags |= api.YART_STATIC_LAUNCH
    if args.reference_tree:
        flags |= api.YART_REFERENCE_TREE
    wide = not args.reference_tree

    # ---- fetch counts of the engine's own traversal -> algorithmic bytes per launch
    d_counts = torch.zeros((args.rays, 4), dtype=torch.int32, device="cuda")
    acc.intersect_closest(d_rays, flags, out=d_hits, counts=d_counts)
    torch.cuda.synchronize()
    counts = d_counts.cpu().numpy().astype(np.uint32)
    del d_counts
    bytes_tri, bytes_sph = algorithmic_bytes(counts, wide, acc.num_spheres > 0)
    mean_counts = counts.mean(0)
    counted_hits = d_hits.clone()

    # ---- warm-up, then the timed region
    for _ in range(args.warmup):
        acc.intersect_closest(d_rays, flags, out=d_hits)
    torch.cuda.synchronize()
    acc.profile(True)
    acc.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tic = time.perf_counter()
    for _ in range(args.steps):
        acc.intersect_closest(d_rays, flags, out=d_hits)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - tic
    prof = acc.profile_read()
    acc.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if own_gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- size-independent properties at full size (parity gate, part 1)
    same_as_counted = bool(torch.equal(d_hits.view(torch.int32), counted_hits.view(torch.int32)))            # idempotent / counting variant agrees
    plain = acc.intersect_closest(d_rays, api.YART_STATIC_LAUNCH | (flags & api.YART_REFERENCE_TREE))   # unsorted, static grid, fix-up launches on
    a, b = api.hits_to_numpy(d_hits), api.hits_to_numpy(plain)
    order_independent = bool(np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32)))
    d_occ = acc.intersect_any(d_rays, flags)
    any_consistent = bool(np.array_equal(d_occ.cpu().numpy() != 0, a["prim"] != -1))
    hit_frac = float((a["prim"] != -1).mean())
    del plain, d_occ, counted_hits
    # ---- any-hit rates (also any-hit Mrays/s; maxT = inf as above and maxT in U(0, 0.5)), outside the timed region
    any_hit = None
    if world == 1 and not strong:
        any_hit = {}
        gen = torch.Generator(device="cuda")
        gen.manual_seed(11)
        d_short = d_rays.clone()
        d_short[:, 7] = torch.rand(args.rays, device="cuda", generator=gen) * 0.5
        for name, batch in (("maxT_inf", d_rays), ("maxT_U(0,0.5)", d_short)):
            d_o = acc.intersect_any(batch, flags)
            torch.cuda.synchronize()
            t_any = time.perf_counter()
            for _ in range(args.steps):
                d_o = acc.intersect_any(batch, flags, out=d_o)
            torch.cuda.synchronize()
            t_any = (time.perf_counter() - t_any) / args.steps
            any_hit[name] = {"Mrays_s": round(args.rays / t_any / 1e6, 1), "ms_per_step": round(t_any * 1e3, 3),
                             "occluded_fraction": round(float((d_o != 0).float().mean().item()), 4)}
        del d_short, d_o

    value = total_rays * args.steps / elapsed / 1e6
    out = {
        "metric": "Mrays/sec (closest-hit, incoherent paths)",
        "value": round(value, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 (f64 sphere test)",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[3]: random soup %d tris (s=%g) + %d spheres, %d uniformly random rays %s, "
                        "closest-hit, tree=%s, sort=%s, launch=%s" % (args.tris, args.s, args.spheres, total_rays if strong else args.rays,
                                                             "in ONE batch cut into contiguous slices of the coherence order, one per GPU" if strong else "per GPU",
                                                             "reference binary (sphere pass + triangle pass)" if args.reference_tree
                                                             else "4-wide quantised, triangles+spheres",
                                                             "off" if args.no_sort else ("none per call: the batch was ordered once by morton+octant" if strong else "morton+octant"),
                                                             "static" if args.static else "persistent"),
            "rays_per_gpu": args.rays, "tris": args.tris, "spheres": args.spheres,
            "scene_replicated": True, "hit_fraction": round(hit_frac, 4),
        },
    }
    t_tri = prof["ms_triangles"] / max(1, args.steps) * 1e-3
    t_sph = prof["ms_spheres"] / max(1, args.steps) * 1e-3
    t_sort = prof["ms_sort"] / max(1, args.steps) * 1e-3
    # Roofline of the dominant kernel, HBM side: bytes that crossed between L2 and memory per launch
    # (rocprofv3 PMC passes of this same command, profiles/traffic.json: FETCH_SIZE with the factor
    # calibrated on this kernel's access shape by tools/calib_fetch.hip + WRITE_SIZE) / this run's
    # kernel time (HIP events on the launch stream).  The cache-inclusive algorithmic rate (every
    # fetch the traversal issues, whether a cache served it or not) is reported beside it; it is not
    # a fraction of anything and can exceed the HBM peak.
    algorithmic = bytes_tri / t_tri / 1e9 if t_tri > 0 else 0.0
    traffic = None
    traffic_source = None
    traffic_stale = False
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    key = "%stris=%d,spheres=%d,rays=%d,s=%g" % ("" if wide else "reftree,", args.tris, args.spheres, args.rays, args.s)
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if key in tr and not strong:
                traffic = tr[key]["hbm_bytes_per_launch"]
                traffic_source = tr[key].get("source")
                # the figure belongs to the kernel it was measured on: another tree of sources, no fraction
                traffic_stale = tr[key].get("kernel_fingerprint") != api.kernel_fingerprint()
        except Exception as exc:     # a malformed side file must not break the bench line
            log("traffic.json ignored: %r" % (exc,))
    achieved = traffic / t_tri / 1e9 if (traffic and t_tri > 0 and not traffic_stale) else None
    out["roofline"] = {
        "bound": "hbm",
        "kernel": ("traceWide<closest> (yart_amd/csrc/traverse_wide.hip)" if wide
                   else "traceKernel<triangles, closest, pruned> (yart_amd/csrc/traverse.hip)"),
        "achieved": round(achieved, 1) if achieved is not None else None,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved is not None else None,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "traffic_stale": traffic_stale,
        "kernel_fingerprint": api.kernel_fingerprint(),
        "kernel_ms": round(t_tri * 1e3, 3),
        "algorithmic_GBps": round(algorithmic, 1),
        "algorithmic_bytes_per_launch": int(bytes_tri),
        "algorithmic_bytes_per_ray": round(bytes_tri / args.rays, 1),
        "per_ray_fetches": ({"child_records_12B": round(float(mean_counts[0]), 2),
                             "triangle_plane_rows_16B": round(float(mean_counts[1]), 2),
                             "second_stage_rows_32B": round(float(mean_counts[2]), 2),
                             "sphere_records_16B": round(float(mean_counts[3]), 2)} if wide else
                            {"tri_tree_node_records_32B": round(float(mean_counts[0]), 2),
                             "triangle_records_48B": round(float(mean_counts[1]), 2),
                             "sphere_tree_node_records_32B": round(float(mean_counts[2]), 2),
                             "sphere_records_16B": round(float(mean_counts[3]), 2)}),
        "other_kernels_ms": {"sort": round(t_sort * 1e3, 3), "other_passes": round(t_sph * 1e3, 3)},
    }
    # measured ceiling of the same device: a plain 1 GiB device-to-device copy, read + write bytes
    src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        dst.copy_(src)
    ev1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del src, dst
    out["roofline"]["stream_copy_GBps"] = round(copy_gbs, 1)
    out["roofline"]["note"] = ("achieved = HBM-side bytes per launch (PMC, separate passes of this command; see traffic_source) / "
                               "this run's kernel time; frac = achieved / 8 TB/s.  algorithmic_GBps counts every fetch the traversal "
                               "issues, cached or not: most of them are served by L1 / L2, so it is not a roofline fraction.  The kernel is "
                               "bound by vector-instruction issue (a wave's step costs the same however many lanes take part), "
                               "not by bytes (DESIGN.md 3.1)"
                               if achieved is not None else
                               "no PMC traffic figure for this configuration in profiles/traffic.json: HBM-side rate not available; "
                               "algorithmic_GBps counts every fetch, cached or not")
    if any_hit is not None:
        out["any_hit"] = any_hit
    out["parity_gate"] = {"idempotent_and_counting_variant_equal": same_as_counted,
                          "sorted_persistent_equals_unsorted_static": order_independent,
                          "any_hit_consistent_with_closest": any_consistent}

    # ---- CPU baseline + sampled parity against it (rank 0, N=1 only)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"], out["parity_gate"]["sample_vs_cpu"] = cpu_baseline(
            verts, tris, sph, rays, a, args.cpu_seconds)
    # ---- the other two points of configs[3]'s sweep over the soup's edge scale, N = 1 only
    if rank == 0 and world == 1 and not strong and not args.no_sweep and args.s == 0.005:
        del d_rays, d_hits
        acc.close()
        torch.cuda.empty_cache()
        out["sweep"] = [{"s": args.s, "Mrays_s": out["value"], "ms_per_step": out["ms_per_step"], "kernel_ms": out["roofline"]["kernel_ms"],
                         "hit_fraction": out["config"]["hit_fraction"]}]
        for s_other in (0.002, 0.01):
            out["sweep"].append(sweep_point(args, s_other, rays, flags))
        out["sweep"].sort(key=lambda r: r["s"])
    # ---- N > 1: the one exchange the path has (image mode), exercised once outside the timed region:
    # every rank renders its tiles of a small frame, yart_frame_gather collects them over RCCL, rank 0 compares
    # with its own unsharded render.  Guarded by a watchdog: a stuck collective must not cost the bench line.
    stuck = False
    if world > 1 and own_gpu and os.environ.get("YART_BENCH_SKIP_GATHER_CHECK") != "1":
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(gather_check(rank, world, device)), daemon=True)
        th.start()
        th.join(180.0)
        stuck = th.is_alive()
        out["multi_gpu_gather_check"] = {"error": "timed out after 180 s"} if stuck else box
    if rank == 0:
        print(json.dumps(out), flush=True)
    if stuck:
        os._exit(3)             # the line above is printed; a hung collective must not pass for a clean run
    if world > 1:
        dist.destroy_process_group()


def sweep_point(args, s, rays, flags):
    """One more point of configs[3]: the same rays through the soup of edge scale s; timed like the headline, with the
    size-independent part of the parity gate."""
    import torch
    from yart_amd import api, scenes
    verts, tris, sph = scenes.random_soup(args.tris, args.spheres, s=s, seed=42)
    acc = api.Accel(verts, tris, sph)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    for _ in range(args.warmup):
        acc.intersect_closest(d_rays, flags, out=d_hits)
    torch.cuda.synchronize()
    acc.profile(True)
    acc.profile_read()
    tic = time.perf_counter()
    for _ in range(args.steps):
        acc.intersect_closest(d_rays, flags, out=d_hits)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - tic
    prof = acc.profile_read()
    acc.profile(False)
    plain = acc.intersect_closest(d_rays, api.YART_STATIC_LAUNCH | (flags & api.YART_REFERENCE_TREE))
    same = bool(torch.equal(plain.view(torch.int32)[:, [0, 3]], d_hits.view(torch.int32)[:, [0, 3]]))
    hit = float((d_hits.view(torch.int32)[:, 3] != -1).float().mean().item())
    acc.close()
    del d_rays, d_hits, plain
    torch.cuda.empty_cache()
    return {"s": s, "Mrays_s": round(len(rays) * args.steps / elapsed / 1e6, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "kernel_ms": round(prof["ms_triangles"] / max(1, args.steps), 3), "hit_fraction": round(hit, 4),
            "sorted_persistent_equals_unsorted_static": same}


def gather_check(rank, world, device):
    """Tile-sharded render of a small frame + yart_frame_gather over an RCCL communicator of the C-ABI;
    -> dict for the JSON line (rank 0's view)."""
    import torch
    import torch.distributed as dist
    from yart_amd import api, scenes
    try:
        torch.cuda.set_device(device)              # the current device is per thread, and this runs in a thread of its own
        W, H = 640, 360
        v, t, s = scenes.architectural(40000, seed=2)
        acc = api.Accel(v, t, s, device=device)
        acc.set_surface_data(v, t, np.zeros_like(v), s)
        cam = api.camera_basis([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], W, H, 0.785398)
        job = (cam, W, H, 2, 4, [0.0, 9.0, 0.0], [60.0, 55.0, 50.0], [0.5, 0.5, 0.5])
        wf = api.Wavefront(acc, 1 << 16)
        ids = [api.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = api.Comm(ids[0], rank, world, device=device)
        pix = torch.from_numpy(api.tiles_rank_pixels(W, H, rank, world).astype(np.int32)).cuda()
        local, _, _ = wf.render_diffuse(*job, seed=5, pixels=pix)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame = comm.gather_frame(local, W, H)
        torch.cuda.synchronize()
        dist.barrier()
        ms = (time.perf_counter() - t0) * 1e3
        res = {"ranks": world, "frame": "%dx%d" % (W, H), "gather_ms_incl_barrier": round(ms, 3)}
        # the second exchange of SURVEY 8e: counters + a checksum summed over the ranks behind the C-ABI (ncclAllReduce)
        sums = torch.tensor([pix.numel(), rank + 1, int(local.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFF],
                            dtype=torch.int64, device="cuda")
        comm.allreduce_u64(sums)
        torch.cuda.synchronize()
        sums = sums.cpu().tolist()
        res["allreduce_u64"] = {"pixels": sums[0], "pixels_ok": sums[0] == W * H, "ranks_ok": sums[1] == world * (world + 1) // 2}
        if rank == 0:
            full, _, _ = wf.render_diffuse(*job, seed=5)
            res["rccl_gathered_frame_equals_single_rank_render_bitwise"] = bool(
                torch.equal(full.view(H, W, 3).view(torch.int32), frame.view(torch.int32)))
            res["allreduce_u64"]["checksum_ok"] = (int(full.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFF) == (sums[2] & 0xFFFFFFFF)
        comm.close()
        wf.close()
        acc.close()
        return res
    except Exception as exc:                       # reported, never fatal to the bench line
        return {"error": repr(exc)}


def cpu_baseline(verts, tris, sph, rays, gpu_hits, budget_s):
    """The reference's own Scene::intersect (oracle/_ref, compiled from /root/reference in
    the authoring container) -- or, if that build is absent, the oracle restatement -- on a
    bounded sample of the SAME workload, all host threads.  Also the sampled parity check:
    the GPU results for those rays against the CPU's."""
    from oracle import oracle as orc
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
    kind = "reference" if os.path.exists(orc.REF_SO) else "port"
    t0 = time.time()
    if kind == "reference":
        cpu = orc.Reference(verts, tris, sph, gen_normals=False)
        run = lambda r: cpu.throughput(r, cores)
    else:
        cpu = orc.Oracle(verts, tris, sph)
        run = lambda r: cpu.intersect(r, 0, nthreads=cores)["t"]
    log("cpu baseline (%s) scene build %.1fs, %d threads" % (kind, time.time() - t0, cores))
    n = 256
    while True:                                   # grow the sample until it costs about budget_s
        t0 = time.perf_counter()
        t_cpu = run(rays[:n])
        dt = max(time.perf_counter() - t0, 1e-4)
        if dt >= budget_s / 3 or n >= len(rays):
            break
        n = min(len(rays), 1 << int(np.log2(n * min(64.0, max(2.0, 0.8 * budget_s / dt)))))
    mrays = n / dt / 1e6
    g = gpu_hits[:n]
    hit_equal = bool(np.array_equal(np.isfinite(t_cpu) & (t_cpu < rays[:n, 7]), g["prim"] != -1))
    m = g["prim"] != -1
    rel = np.abs(g["t"][m] - t_cpu[m]) / np.abs(t_cpu[m])
    parity = {"rays": n, "hit_miss_equal": hit_equal,
              "t_bit_equal_fraction": float((g["t"].view(np.uint32) == t_cpu.view(np.uint32)).mean()),
              "max_rel_t_err": float(rel.max()) if rel.size else 0.0}
    base = {"value": round(mrays, 6), "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": "first %d rays of the same batch, same scene (10M-class soup incl. the O(#spheres) linear sphere "
                      "loop of include/scene.h:142-144), %.1f s wall" % (n, dt)}
    # the reference's own thread count (numWorkers = 8, src/scheduler.cpp:20) on the same sample
    if cores != 8:
        t0 = time.perf_counter()
        (cpu.throughput(rays[:n], 8) if kind == "reference" else cpu.intersect(rays[:n], 0, nthreads=8))
        dt8 = max(time.perf_counter() - t0, 1e-4)
        base["at_8_threads"] = {"value": round(n / dt8 / 1e6, 6), "unit": "Mrays/s", "cores": 8,
                                "sample": "the same %d rays, %.1f s wall" % (n, dt8)}
    # the same sample without spheres: the BVH part alone (BASELINE.md plan)
    if len(sph):
        del cpu
        if kind == "reference":
            cpu2 = orc.Reference(verts, tris, np.zeros((0, 4), np.float32), gen_normals=False)
            run2 = lambda r: cpu2.throughput(r, cores)
        else:
            cpu2 = orc.Oracle(verts, tris, np.zeros((0, 4), np.float32))
            run2 = lambda r: cpu2.intersect(r, 0, nthreads=cores)["t"]
        n2 = 1 << 12
        while True:
            t0 = time.perf_counter()
            run2(rays[:n2])
            dt2 = max(time.perf_counter() - t0, 1e-4)
            if dt2 >= budget_s / 6 or n2 >= len(rays):
                break
            n2 = min(len(rays), 1 << int(np.log2(n2 * min(64.0, max(2.0, 0.4 * budget_s / dt2)))))
        base["triangles_only"] = {"value": round(n2 / dt2 / 1e6, 6), "unit": "Mrays/s", "cores": cores,
                                  "sample": "first %d rays, spheres removed, %.1f s wall" % (n2, dt2)}
        if cores != 8:
            t0 = time.perf_counter()
            run2_8 = (lambda r: cpu2.throughput(r, 8)) if kind == "reference" else (lambda r: cpu2.intersect(r, 0, nthreads=8))
            run2_8(rays[:n2])
            dt28 = max(time.perf_counter() - t0, 1e-4)
            base["triangles_only"]["at_8_threads"] = {"value": round(n2 / dt28 / 1e6, 6), "unit": "Mrays/s", "cores": 8,
                                                      "sample": "the same %d rays, %.1f s wall" % (n2, dt28)}
    return base, parity


if __name__ == "__main__":
    main()
