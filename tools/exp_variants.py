"""Time scheduling-knob variants of the traversal kernels on one scene (GPU box).
    python tools/exp_variants.py [tris spheres rays]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from yart_amd import api, scenes
tris_n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sph_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rays_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
variants = [tuple(map(int, v.split(':'))) for v in (sys.argv[4].split(',') if len(sys.argv) > 4 else
            "16:24:1,16:1:1,16:16:1,32:32:1".split(','))]
v, t, s = scenes.random_soup(tris_n, sph_n, s=0.005, seed=42)
rays = torch.from_numpy(scenes.random_rays(rays_n, seed=7)).cuda()
hits = torch.empty((rays_n, 4), dtype=torch.float32, device='cuda')
ref = None
for refill, node, xcd in variants:
    os.environ['YART_REFILL_MIN'] = str(refill); os.environ['YART_NODE_MIN'] = str(node); os.environ['YART_XCD_AWARE'] = str(xcd)
    a = api.Accel(v, t, s)
    for flags, name in ((api.YART_SORT_RAYS, 'sorted'),):
        a.intersect_closest(rays, flags, out=hits); torch.cuda.synchronize()
        a.profile(True); a.profile_read()
        t0 = time.perf_counter()
        for _ in range(3): a.intersect_closest(rays, flags, out=hits)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        p = a.profile_read(); a.profile(False)
        h = hits.view(torch.int32).clone()
        if ref is None: ref = h
        print("refill=%2d node=%2d xcd=%d %s: step %.3f ms  tri %.3f  sph %.3f  sort %.3f  -> %.1f Mrays/s  same=%s"
              % (refill, node, xcd, name, dt * 1e3, p['ms_triangles'] / 3, p['ms_spheres'] / 3, p['ms_sort'] / 3,
                 rays_n / dt / 1e6, bool(torch.equal(h, ref))), flush=True)
    a.close()
