"""Throughput of the hit post-processing gather (surface.hip) on the C4 scene: 2^24 closest hits ->
YartSurface records.  Prints one JSON line (algorithmic bytes / HIP-event time vs the HBM peak).

    python tools/bench_surface.py [--tris N --spheres N --rays N]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--spheres", type=int, default=1_000_000)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--s", type=float, default=0.005)
    args = ap.parse_args()
    verts, tris, sph = scenes.random_soup(args.tris, args.spheres, s=args.s, seed=42)
    rng = np.random.default_rng(1)
    normals = rng.normal(size=verts.shape).astype(np.float32)
    acc = api.Accel(verts, tris, sph)
    acc.set_surface_data(verts, tris, normals, sph)
    rays = torch.from_numpy(scenes.random_rays(args.rays)).cuda()
    hits = acc.intersect_closest(rays, api.YART_SORT_RAYS)
    out = torch.empty((args.rays, 12), dtype=torch.float32, device="cuda")
    for _ in range(2):
        acc.surface(rays, hits, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    steps = 10
    for _ in range(steps):
        acc.surface(rays, hits, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    prim = api.hits_to_numpy(hits)["prim"]
    n_tri, n_sph = int((prim >= 0).sum()), int((prim <= -2).sum())
    alg = args.rays * (16 + 48) + n_tri * 64 + n_sph * (32 + 16)
    print(json.dumps({"kernel": "surfaceKernel", "rays": args.rays, "triangle_hits": n_tri, "sphere_hits": n_sph,
                      "ms": round(ms, 4), "Mrays_s": round(args.rays / ms / 1e3, 1),
                      "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 3)}))


if __name__ == "__main__":
    main()
