#!/usr/bin/env python3
"""Fuzz of the builders and of every traversal rule (GPU box): random scenes of many sizes and shapes, device-built and
host-built accel -- the reference's arrays byte for byte between the two, closest-hit / any-hit answers of all rules equal
to the reference's rule restated on the CPU, on every ray.
    python tools/fuzz_device_build.py [rounds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api, scenes
from oracle import oracle as orc          # the checker: this script is a test tool

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run just this round (the others only draw their random numbers)
bad = 0
for it in range(rounds):
    n_tris = int(rng.choice([0, 3, 900, 1024, 1500, 5000, 40000, 262144 - 100, 270000, 600000]))
    n_sph = int(rng.choice([0, 1, 50, 1023, 1024, 3000, 60000]))
    kind = rng.randint(4)
    if kind == 0 or n_tris < 1000:
        kind = 0
        v, t, s = scenes.random_soup(n_tris, n_sph, s=float(rng.choice([0.002, 0.02, 0.2])), seed=int(rng.randint(1 << 30)))
    elif kind == 1:
        v, t, _ = scenes.architectural(max(n_tris, 2000), seed=int(rng.randint(1 << 30)))
        _, _, s = scenes.random_soup(0, n_sph, s=0.05, seed=3)
        s = s.copy(); s[:, :3] = s[:, :3] * 20 - 10
    elif kind == 2:          # grid-aligned: ties everywhere, median ranges
        base = rng.randint(-6, 7, size=(n_tris, 1, 3)).astype(np.float32)
        v = (base + rng.randint(0, 3, size=(n_tris, 3, 3)).astype(np.float32) * 0.5).reshape(-1, 3)
        t = np.zeros((n_tris, 5), np.int32); t[:, 0] = np.arange(n_tris) * 3; t[:, 1] = t[:, 0] + 1; t[:, 2] = t[:, 0] + 2; t[:, 4] = np.arange(n_tris)
        _, _, s = scenes.random_soup(0, n_sph, s=0.05, seed=4)
    else:                    # clustered: most primitives in a small corner, a few far away
        v, t, s = scenes.random_soup(n_tris, n_sph, s=0.01, seed=int(rng.randint(1 << 30)))
        v = v.copy(); far = rng.rand(len(v)) < 0.001; v[far] *= 1000.0
    if len(t) + len(s) < 2:
        continue
    rays = scenes.random_rays(20000, seed=int(rng.randint(1 << 30)))
    if kind in (1, 2):
        rays[:, :3] = rays[:, :3] * 14 - 7
    rays[::97, 4] = 0.0                     # some axis-parallel ones for the fix-up path
    shadow = rays.copy(); shadow[:, 7] = float(rng.choice([0.3, 3.0]))
    if only >= 0 and it != only:
        continue
    if only >= 0:
        np.savez("gpurun_out/fuzz_round_%d.npz" % it, verts=v, tris=t, spheres=s, rays=rays, shadow=shadow)
    res = {}
    failed = {}
    for mode in ("1", "0"):
        os.environ["YART_BUILD_DEVICE"] = mode
        try:
            a = api.Accel(v, t, s)
        except api.YartError as exc:
            failed[mode] = str(exc)
            continue
        d, ds = torch.from_numpy(rays).cuda(), torch.from_numpy(shadow).cuda()
        res[mode] = (a.nodes().tobytes(), a.triaccel().tobytes(),
                     [a.intersect_closest(d, f).cpu().numpy().copy() for f in (0, api.YART_SORT_RAYS, api.YART_REFERENCE_TREE, api.YART_REFERENCE_RULE)],
                     [a.intersect_any(ds, f).cpu().numpy().copy() for f in (0, api.YART_REFERENCE_RULE)],
                     a.wide_built_on_device[0], a.ref_built_on_device)
        a.close()
    if failed:
        same = failed.get("1") == failed.get("0")
        print("round %2d kind %d tris %7d spheres %6d: create failed %s (%s)" % (it, kind, len(t), len(s), "both ways alike" if same else "DIFFERENTLY", failed), flush=True)
        bad += 0 if same else 1
        continue
    # (after the builds: a tree deeper than 64 levels is refused by the library; the reference, and its restatement,
    # would run over their fixed stacks, src/bvhaccel.cpp:242-244)
    o = orc.Oracle(v, t, s)
    want = o.intersect(rays, mode=0, nthreads=16)
    want_occ = o.intersect_shadow(shadow, mode=0, nthreads=16)
    why = []
    if res["1"][0] != res["0"][0]: why.append("nodes")
    if res["1"][1] != res["0"][1]: why.append("triaccel")
    # Every rule on both builders' trees against the reference's rule restated on the CPU (oracle mode 0), on every ray:
    # primitive and t bits, occlusion bytes.  No relaxed bars: slivers (kind 3), tiny far spheres (kinds 1, 2 see the
    # spheres from up to 7 units outside the scene) and exact ties (kind 2) are answered like the reference.
    names = ("default", "sorted", "reference tree", "reference rule")
    for mode in ("1", "0"):
        for k, x in enumerate(res[mode][2]):
            d = np.nonzero((x[:, 3].view(np.int32) != want["prim"]) | (x[:, 0].view(np.uint32) != want["t"].view(np.uint32)))[0]
            if len(d):
                why.append("closest, %s, build %s: %d rays differ, first %d: %s vs %s" % (names[k], mode, len(d), d[0], (x[d[0], 0], int(x[d[0], 3:].view(np.int32)[0])), (want["t"][d[0]], want["prim"][d[0]])))
        for k, x in enumerate(res[mode][3]):
            if not np.array_equal(x, want_occ): why.append("any, %s, build %s: %d differ" % (("default", "reference rule")[k], mode, int((x != want_occ).sum())))
    ok = not why
    print("round %2d kind %d tris %7d spheres %6d device-built wide=%s ref=%s median ranges %d nan records %d: %s"
          % (it, kind, len(t), len(s), res["1"][4], res["1"][5][0], res["1"][5][2], res["1"][5][3], "ok" if ok else "MISMATCH " + "; ".join(why)), flush=True)
    bad += 0 if ok else 1
    if only >= 0:
        np.savez("gpurun_out/fuzz_round_%d_results.npz" % it, **{"closest_%s_%d" % (m, k): x for m in ("1", "0") for k, x in enumerate(res[m][2])})
print("mismatches:", bad)
sys.exit(1 if bad else 0)
