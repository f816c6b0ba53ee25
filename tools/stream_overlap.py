#!/usr/bin/env python3
"""Mid-size batches: the closest-hit and the shadow batch of one bounce issued
on ONE stream one after the other, or on TWO streams at the same time, on the C4 scene.  A persistent launch
of a batch below ~2^20 rays does not fill the GPU (grid = batch / 256 workgroups), so two of them can share
it.  Prints one JSON line per batch size: Mrays/s of both ways (rays of both batches / wall time)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes

tris_n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sph_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
v, t, s = scenes.random_soup(tris_n, sph_n, s=0.005, seed=42)
acc = api.Accel(v, t, s)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for logn in (14, 16, 18, 19, 20, 21, 22):
    n = 1 << logn
    rays = torch.from_numpy(scenes.random_rays(n, seed=7)).cuda()
    shadow = torch.from_numpy(scenes.random_rays(n, seed=8, tmax=0.5)).cuda()
    hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    occ = torch.empty((n,), dtype=torch.uint8, device="cuda")
    flags = api.YART_SORT_RAYS | api.YART_RAYS_REGULAR
    reps = 20 if logn <= 20 else 8
    res = {}
    for name in ("one_stream", "two_streams"):
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3 if warm else reps):
                acc.intersect_closest(rays, flags, out=hits, stream=s1)
                acc.intersect_any(shadow, flags, out=occ, stream=s1 if name == "one_stream" else s2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (3 if warm else reps)
        res[name + "_ms"] = round(dt * 1e3, 4)
        res[name + "_Mrays_s"] = round(2 * n / dt / 1e6, 1)
    res = {"rays_per_batch": n, **res, "gain": round(res["one_stream_ms"] / res["two_streams_ms"], 3)}
    print(json.dumps(res), flush=True)
