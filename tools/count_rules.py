"""Per-ray fetch counts of the three traversal rules on the C4 scene (the un-pruned
reference rule and the pruned rule on the reference's tree, next to the default 4-wide tree).
    python tools/count_rules.py [log2_rays]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
verts, tris, sph = scenes.random_soup(10_000_000, 1_000_000, s=0.005, seed=42)
acc = api.Accel(verts, tris, sph)
rays = torch.from_numpy(scenes.random_rays(1 << lg, seed=7)).cuda()
for name, flags in (("default: 4-wide tree, pruned (16-B child records, plane rows, 2nd-stage rows, spheres)", 0),
                    ("reference tree, pruned (tri-tree 32-B nodes, 48-B triangles, sphere-tree nodes, spheres)", api.YART_REFERENCE_TREE),
                    ("reference rule, un-pruned (same columns)", api.YART_REFERENCE_RULE)):
    counts = torch.zeros((1 << lg, 4), dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    acc.intersect_closest(rays, flags)
    e0.record()
    acc.intersect_closest(rays, flags)
    e1.record()
    acc.intersect_closest(rays, flags, counts=counts)
    torch.cuda.synchronize()
    print(json.dumps({"rule": name, "rays": 1 << lg, "per_ray": [round(x, 2) for x in counts.float().mean(0).tolist()],
                      "ms": round(e0.elapsed_time(e1), 3)}), flush=True)
