"""Time build/scheduling variants of the wide-tree kernel on the C4 scene (GPU box).
    python tools/exp_wide.py "leaf:refill:node,..." [tris spheres rays]
Each variant is KEY=VAL;KEY=VAL of environment knobs read at accel creation."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from yart_amd import api, scenes
variants = sys.argv[1].split(',')
tris_n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
sph_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rays_n = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 24
v, t, s = scenes.random_soup(tris_n, sph_n, s=0.005, seed=42)
rays = torch.from_numpy(scenes.random_rays(rays_n, seed=7)).cuda()
hits = torch.empty((rays_n, 4), dtype=torch.float32, device='cuda')
ref = None
for var in variants:
    keys = []
    for kv in var.split(';'):
        if kv:
            k, val = kv.split('=')
            os.environ[k] = val
            keys.append(k)
    t0 = time.time()
    a = api.Accel(v, t, s)
    tb = time.time() - t0
    flags = api.YART_SORT_RAYS
    a.intersect_closest(rays, flags, out=hits); torch.cuda.synchronize()
    a.profile(True); a.profile_read()
    t0 = time.perf_counter()
    for _ in range(6): a.intersect_closest(rays, flags, out=hits)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    p = a.profile_read(); a.profile(False)
    h = hits.view(torch.int32).clone()
    if ref is None: ref = h
    nd = (h != ref).any(1).sum().item()
    print("%-60s build %.1fs depth %d: step %.3f ms  main %.3f  other %.3f  sort %.3f -> %.1f Mrays/s  diff=%d  MB=%.0f"
          % (var, tb, a.wide_depth, dt * 1e3, p['ms_triangles'] / 6, p['ms_spheres'] / 6, p["ms_sort"] / 6,
             rays_n / dt / 1e6, nd, a.device_bytes / 1e6), flush=True)
    a.close()
    for k in keys: os.environ.pop(k, None)
