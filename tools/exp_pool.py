#!/usr/bin/env python3
"""Experiment (GPU box): the phase-homogeneous pool kernel (YART_WIDE_POOL, traverse_wide.hip: tracePool) against the
default kernel on BASELINE configs[3]: results bit for bit, ms per sorted closest-hit call.
    python tools/exp_pool.py [--tris N --spheres N --rays N --s 0.005] [--configs "S,W,block,keep,tri,sph;..."]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000)
ap.add_argument("--spheres", type=int, default=1_000_000)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--s", type=float, default=0.005)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--configs", default="11,10,512,48,32,24")
ap.add_argument("--out", default="")
args = ap.parse_args()

v, t, s = scenes.random_soup(args.tris, args.spheres, s=args.s, seed=1)
acc = api.Accel(v, t, s)
rays = torch.from_numpy(scenes.random_rays(args.rays, seed=7)).cuda()
keys = ("YART_WIDE_POOL", "YART_WIDE_POOL_WINDOW", "YART_WIDE_POOL_BLOCK", "YART_WIDE_POOL_KEEP", "YART_WIDE_POOL_TRI", "YART_WIDE_POOL_SPH")


def run(label):
    out = acc.intersect_closest(rays, api.YART_SORT_RAYS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = acc.intersect_closest(rays, api.YART_SORT_RAYS)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    print("%-40s %.3f ms per call, %.1f Mrays/s" % (label, ms, args.rays / ms / 1e3), flush=True)
    return out.clone(), ms


for k in keys:
    os.environ.pop(k, None)
base, base_ms = run("default kernel")
res = [{"config": "default", "ms": round(base_ms, 3)}]
for cfg in args.configs.split(";"):
    vals = cfg.split(",")
    for k, x in zip(keys, vals):
        os.environ[k] = x
    got, ms = run("pool S,W,block,keep,tri,sph = " + cfg)
    same = bool(torch.equal(got.view(torch.int32), base.view(torch.int32)))
    print("    bit-identical to the default kernel:", same, flush=True)
    res.append({"config": cfg, "ms": round(ms, 3), "identical": same})
    if not same:
        d = (got.view(torch.int32) != base.view(torch.int32)).any(1).nonzero().flatten()
        print("    differing rays:", int(d.numel()), "first", d[:5].tolist(), got[d[:3]].tolist(), base[d[:3]].tolist(), flush=True)
for k in keys:
    os.environ.pop(k, None)
base2, ms2 = run("default kernel again")
res.append({"config": "default again", "ms": round(ms2, 3)})
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
