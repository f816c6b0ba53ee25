#!/usr/bin/env python3
"""One-off parity check at BASELINE configs[3] scale (lives under tests/ because it calls the oracle):
the 10 M-triangle soup (spheres left out: the reference walks them linearly, 10^6 per ray) and 2^22 of
the bench's rays, every one of them traced by the GPU (default rule and reference rule) and by the
oracle's restatement of the reference on the host cores.  Prints one JSON line.

    python tests/parity_at_scale.py [log2_rays]
    python tests/parity_at_scale.py 18 1000000 100000      # with spheres: tris, spheres (same local density as configs[3];
                                                           # the oracle loops over all spheres per ray like the reference)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from yart_amd import api, scenes  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
n_sph = int(sys.argv[3]) if len(sys.argv) > 3 else 0
# fewer primitives in the same cube: larger ones, so that a ray meets as many of them (0.005 at 10 M triangles)
scale = 0.005 * (10_000_000 / n_tris) ** (1.0 / 3.0)
verts, tris, sph = scenes.random_soup(n_tris, n_sph, s=scale, seed=42)
rays = scenes.random_rays(1 << lg, seed=7)
acc = api.Accel(verts, tris, sph)
d = torch.from_numpy(rays).cuda()
fast = api.hits_to_numpy(acc.intersect_closest(d, api.YART_SORT_RAYS))
exact = api.hits_to_numpy(acc.intersect_closest(d, api.YART_SORT_RAYS | api.YART_REFERENCE_RULE))
threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
t0 = time.time()
o = orc.Oracle(verts, tris, sph)
t1 = time.time()
want = o.intersect(rays, mode=0, nthreads=threads)
t2 = time.time()
hit = want["prim"] != -1
out = {"scene": "%d-triangle soup + %d spheres, s=%.4f" % (n_tris, n_sph, scale), "rays": 1 << lg, "hits": int(hit.sum()),
       "reference_rule": {"same_prim": float((exact["prim"] == want["prim"]).mean()),
                          "t_bits_equal": float((exact["t"].view(np.uint32) == want["t"].view(np.uint32)).mean()),
                          "uv_bits_equal": float(((exact["u"].view(np.uint32) == want["u"].view(np.uint32)) &
                                                  (exact["v"].view(np.uint32) == want["v"].view(np.uint32)))[hit].mean())},
       "default_rule": {"hit_miss_equal": bool(np.array_equal(fast["prim"] != -1, hit)),
                        "same_prim": float((fast["prim"] == want["prim"]).mean()),
                        "t_bits_equal": float((fast["t"].view(np.uint32) == want["t"].view(np.uint32)).mean())},
       "oracle": {"build_s": round(t1 - t0, 1), "trace_s": round(t2 - t1, 1), "threads": threads,
                  "Mrays_s": round((1 << lg) / (t2 - t1) / 1e6, 3)}}
print(json.dumps(out))
