"""Tree-quality harness on the CPU (no GPU): builds the 4-wide tree of a soup with the local density of
BASELINE configs[3] (fewer primitives, edge scale enlarged to keep triangles-per-volume x size^3 equal)
under build-knob variants and reports the deterministic walk's fetch counts per ray (oracle walk of the
dumped tree): child records (4 per node visit), plane rows, second-stage rows, sphere records.
    python tools/tree_quality.py "VAR=1;VAR2=3,OTHER=2" [tris] [rays]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes
from oracle import oracle as orc

variants = sys.argv[1].split(',') if len(sys.argv) > 1 else ['']
n_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 40_000
s = 0.005 * (10_000_000 / n_tris) ** (1.0 / 3.0)
v, t, sph = scenes.random_soup(n_tris, n_tris // 10, s=s, seed=42)
rays = scenes.random_rays(n_rays, seed=7)
ref = None
for var in variants:
    keys = []
    for kv in var.split(';'):
        if kv:
            k, val = kv.split('=')
            os.environ[k] = val
            keys.append(k)
    t0 = time.time()
    nodes, ra, rbc, depth = api.build_wide_host(v, t, sph)
    tb = time.time() - t0
    hits, cnt = orc.wide_intersect(nodes, ra, rbc, rays, counts=True)
    if ref is None:
        ref = hits
    same = bool(np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32)))
    m = cnt.mean(0)
    print("%-50s build %.1fs nodes %8d rows %8d depth %2d: node visits %.2f  plane rows %.2f  stage2 %.2f  spheres %.2f  same=%s"
          % (var, tb, len(nodes), len(ra), depth, m[0] / 4, m[1], m[2], m[3], same), flush=True)
    for k in keys:
        os.environ.pop(k, None)
