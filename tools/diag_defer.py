#!/usr/bin/env python3
"""Diagnostic (GPU box): how many rays of the bench batch the wide kernel leaves to the fix-up launches, and why;
the classes of the triangles' error bands.   python tools/diag_defer.py [s] [rays_log2]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes

s = float(sys.argv[1]) if len(sys.argv) > 1 else 0.005
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
v, t, sph = scenes.random_soup(10_000_000, 1_000_000, s=s, seed=42)
rays = scenes.random_rays(n, seed=7)
acc = api.Accel(v, t, sph)
d = torch.from_numpy(rays).cuda()
h = api.hits_to_numpy(acc.intersect_closest(d, api.YART_SORT_RAYS | api.YART_RAYS_REGULAR))
left = h["prim"] == api.YART_PRIM_UNRESOLVED
print("s", s, "rays", n, "unresolved", int(left.sum()))
cover = api.scene_cover(v, t, sph)
print("cover", cover[:8])
r = rays[left][:8]
print(r)
inside = ((rays[:, :3] >= cover[:3]) & (rays[:, :3] <= cover[3:6])).all(1)
print("outside cover", int((~inside).sum()), "of the unresolved", int((~inside & left).sum()))
with np.errstate(divide="ignore"):
    inv = np.abs(1.0 / rays[left][:, 4:7])
print("max |1/d| of unresolved", np.sort(inv.max(1))[-5:] if left.any() else None)
nodes, rows_a, rows_bc = acc.wide()
kw = rows_a.view(np.uint32)[:, 3]
tri = (np.abs(rows_bc[:, 4:7]).sum(1) > 0) | (np.abs(rows_bc[:, 0:3]).sum(1) > 1e30)
print("band classes of triangles", np.bincount((kw[:len(kw)] >> 29)))
