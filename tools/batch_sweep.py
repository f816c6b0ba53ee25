"""Batch-size sweep on one scene: per batch size, wall time per call and the library's own kernel
timing (sort / main kernel / other passes), with and without the coherence sort.  One JSON line per row.

    python tools/batch_sweep.py [--tris N --spheres N] [--sizes 14,16,18,20,22]
"""
import argparse
import json
import os
import sys
import time

import torch

os.environ.setdefault("YART_SORT_MIN_RAYS", "0")        # measure the sort at every size; the library's default skips it below ~1.5 M rays
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--spheres", type=int, default=1_000_000)
    ap.add_argument("--s", type=float, default=0.005)
    ap.add_argument("--sizes", default="12,14,16,18,20,22,24")
    args = ap.parse_args()
    verts, tris, sph = scenes.random_soup(args.tris, args.spheres, s=args.s, seed=42)
    acc = api.Accel(verts, tris, sph)
    all_rays = torch.from_numpy(scenes.random_rays(1 << 24)).cuda()
    for lg in [int(x) for x in args.sizes.split(",")]:
        n = 1 << lg
        rays = all_rays[:n].contiguous()
        hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        for name, flags in (("sort", api.YART_SORT_RAYS), ("nosort", 0), ("sort+regular", api.YART_SORT_RAYS | api.YART_RAYS_REGULAR)):
            steps = max(5, min(200, (1 << 25) // n))
            for _ in range(3):
                acc.intersect_closest(rays, flags, out=hits)
            torch.cuda.synchronize()
            acc.profile(False)
            t0 = time.perf_counter()
            for _ in range(steps):
                acc.intersect_closest(rays, flags, out=hits)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e3
            acc.profile(True)
            acc.profile_read()
            for _ in range(steps):
                acc.intersect_closest(rays, flags, out=hits)
            torch.cuda.synchronize()
            p = acc.profile_read()
            acc.profile(False)
            print(json.dumps({"log2_rays": lg, "mode": name, "wall_ms": round(wall, 4), "Mrays_s": round(n / wall / 1e3, 1),
                              "sort_ms": round(p["ms_sort"] / steps, 4), "kernel_ms": round(p["ms_triangles"] / steps, 4),
                              "other_ms": round(p["ms_spheres"] / steps, 4)}), flush=True)


if __name__ == "__main__":
    main()
