#!/usr/bin/env python3
"""Any-hit on the bench scene: 2^24 shadow rays (random rays with maxT 0.5), sorted; ms per call.  YART_AMD_LIB selects the build."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes
v, t, s = scenes.random_soup(10_000_000, 1_000_000, s=0.005, seed=42)
acc = api.Accel(v, t, s)
rays = torch.from_numpy(scenes.random_rays(1 << 24, seed=8, tmax=0.5)).cuda()
occ = torch.empty((1 << 24,), dtype=torch.uint8, device="cuda")
for reps in (3, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        acc.intersect_any(rays, api.YART_SORT_RAYS, out=occ)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print("any-hit: %.3f ms per call, %.1f Mrays/s, occluded fraction %.4f" % (dt * 1e3, (1 << 24) / dt / 1e6, float(occ.float().mean())))
