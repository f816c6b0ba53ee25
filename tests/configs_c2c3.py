#!/usr/bin/env python3
"""Parity-checked measurements of the BASELINE.json configurations that are not the bench line
(C1 is tests/test_trace_dropin.py, C4 is bench.py, C5 is tools/render_tiles.py).  Lives under tests/ because it checks every batch against the oracle.

    python tests/configs_c2c3.py [--tris 262144] [--width 1920 --height 1080] [--spp 4] [--depth 8]

C2  pinhole primary rays, one per pixel centre (Camera::sample, include/camera.h:24-31), into the
    Sponza-class stand-in scene (yart_amd/scenes.py: no assets and no network in the build environment).
C3  diffuse path tracing of the same scene with the structure of the reference's trace()
    (src/renderer.cpp:29-136): per bounce a closest-hit batch, a shadow batch towards a point light
    (origin P + wi * EPS, maxT = |P - L| - eps, :101-102), Russian roulette on the path weight's
    luminance (:113-115), cosine-weighted BSDF sample (Lambertian 0.5, include/qmc.h:38-52), next ray
    from P + dir * EPS (:129).  Ray GENERATION here is torch plumbing with torch's RNG -- it
    reproduces the distribution of the reference's secondary rays, not its random stream (the
    bit-exact stream is covered by tests/golden/g4_trace.npz).

Every batch is timed with events on the stream it runs on; a seeded sample of each batch is
re-traced by the oracle (CPU restatement of the reference) and compared at the parity bar.
Prints one JSON object per configuration.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPS = 1e-4           # include/constants.h:7


def timed(fn, reps=3):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps


def parity_sample(oracle, rays_np, got, occluded=None, n=8192, seed=0):
    """GPU results of a seeded sample of the batch against the oracle (reference rule)."""
    rng = np.random.RandomState(seed)
    idx = rng.choice(len(rays_np), size=min(n, len(rays_np)), replace=False)
    r = np.ascontiguousarray(rays_np[idx])
    if occluded is not None:
        want = oracle.intersect_shadow(r, mode=0, nthreads=16)
        return {"rays": int(len(idx)), "any_hit_equal": bool(np.array_equal(want, occluded[idx]))}
    want = oracle.intersect(r, mode=0, nthreads=16)
    g = got[idx]
    m = want["prim"] != -1
    rel = np.abs(g["t"][m] - want["t"][m]) / np.abs(want["t"][m])
    return {"rays": int(len(idx)), "hit_miss_equal": bool(np.array_equal(g["prim"] != -1, m)),
            "same_prim_fraction": float((g["prim"] == want["prim"]).mean()),
            "max_rel_t_err": float(rel.max()) if rel.size else 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--obj", nargs=2, metavar=("FOLDER", "FILE"), default=None,
                    help="trace this OBJ scene (loaded by yart_obj_load, e.g. a supplied Sponza) instead of the stand-in; "
                         "camera and light are placed from its bounds")
    args = ap.parse_args()

    import torch
    from yart_amd import api, scenes
    from oracle import oracle as orc

    cam_pos, cam_dir, light_pos, label = [-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], [0.0, 9.0, 0.0], "Sponza-class stand-in"
    if args.obj:
        o = api.load_obj(args.obj[0] if args.obj[0].endswith(os.sep) else args.obj[0] + os.sep, args.obj[1])
        verts, tris, sph = o["verts"], o["tris"], o["spheres"]
        finite = verts[np.isfinite(verts).all(1)]
        lo, hi = finite.min(0), finite.max(0)
        c, e = (lo + hi) / 2, hi - lo
        cam_pos = [float(c[0] - 0.4 * e[0]), float(c[1] + 0.1 * e[1]), float(c[2])]
        light_pos = [float(c[0]), float(c[1] + 0.4 * e[1]), float(c[2])]
        label = "OBJ %s" % args.obj[1]
    else:
        verts, tris, sph = scenes.architectural(args.tris, seed=args.seed)
    t0 = time.time()
    acc = api.Accel(verts, tris, sph, device=0)
    build_s = time.time() - t0
    oracle = orc.Oracle(verts, tris, sph)
    ta = torch.from_numpy(acc.triaccel().view(np.float32).reshape(-1, 12).copy()).cuda()
    ta_k = torch.from_numpy(acc.triaccel()["k"].astype(np.int64)).cuda()
    scene = {"tris": int(acc.num_triangles), "nodes_wide": None, "wide_depth": int(acc.wide_depth),
             "device_MB": round(acc.device_bytes / 1e6, 1), "build_s": round(build_s, 2)}

    # ---- C2: primary rays
    rays_np = scenes.camera_rays(cam_pos, cam_dir, args.width, args.height, 0.785398)
    d_rays = torch.from_numpy(rays_np).cuda()
    out = {"config": "C2 primary rays %dx%d, %s (%d tris)" % (args.width, args.height, label, acc.num_triangles),
           "scene": scene, "rays": len(rays_np)}
    for name, flags in (("default", 0), ("sorted", api.YART_SORT_RAYS), ("reference_tree", api.YART_REFERENCE_TREE),
                        ("reference_rule", api.YART_REFERENCE_RULE)):
        hits, ms = timed(lambda f=flags: acc.intersect_closest(d_rays, f))
        out["%s_ms" % name] = round(ms, 3)
        out["%s_Mrays_s" % name] = round(len(rays_np) / ms / 1e3, 1)
        if name == "default":
            got = api.hits_to_numpy(hits)
            out["hit_fraction"] = float((got["prim"] != -1).mean())
            out["parity_sample_vs_oracle"] = parity_sample(oracle, rays_np, got)
    print(json.dumps(out), flush=True)

    # ---- C3: diffuse path tracing, per-bounce batches
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    W, H, spp = args.width, args.height, args.spp
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    px = np.tile(xs.ravel(), spp) + (np.random.RandomState(5).random_sample(W * H * spp).astype(np.float32) - 0.5)
    py = np.tile(ys.ravel(), spp) + (np.random.RandomState(6).random_sample(W * H * spp).astype(np.float32) - 0.5)
    rays = torch.from_numpy(scenes.camera_rays(cam_pos, cam_dir, W, H, 0.785398, px, py)).cuda()
    light = torch.tensor(light_pos, device="cuda")
    bounces = []
    for b in range(args.depth):
        n = rays.shape[0]
        if n == 0:
            break
        flags = api.YART_SORT_RAYS if b > 0 else 0
        hits, ms_c = timed(lambda: acc.intersect_closest(rays, flags))
        rec = {"bounce": b, "closest_rays": int(n), "closest_ms": round(ms_c, 3), "closest_Mrays_s": round(n / ms_c / 1e3, 1)}
        if b in (0, 1, 3):
            rec["closest_parity"] = parity_sample(oracle, rays.cpu().numpy(), api.hits_to_numpy(hits), seed=b)
        prim = hits[:, 3].contiguous().view(torch.int32)
        alive = prim >= 0
        r = rays[alive]
        t = hits[alive, 0:1]
        pid = prim[alive].long()
        P = r[:, 0:3] + r[:, 4:7] * t
        # geometric normal from the Wald record: n[k] = 1, n[ku] = n_u, n[kv] = n_v (include/triaccel.h:86-92)
        k = ta_k[pid]
        nrm = torch.zeros_like(P)
        nrm.scatter_(1, k[:, None], 1.0)
        nrm.scatter_(1, ((k + 1) % 3)[:, None], ta[pid, 0:1])
        nrm.scatter_(1, ((k + 2) % 3)[:, None], ta[pid, 1:2])
        nrm = nrm / nrm.norm(dim=1, keepdim=True)
        nl = torch.where((nrm * r[:, 4:7]).sum(1, keepdim=True) < 0, nrm, -nrm)      # src/renderer.cpp:76
        # shadow rays towards the point light (src/renderer.cpp:98-104, PointLight::sample light.h:41-52)
        to_l = light[None] - P
        dist = to_l.norm(dim=1, keepdim=True)
        wi = to_l / dist
        sh = torch.cat([P + wi * EPS, torch.zeros_like(dist), wi, dist - 0.0], 1).contiguous()
        occ, ms_s = timed(lambda: acc.intersect_any(sh, api.YART_SORT_RAYS))
        rec.update(shadow_rays=int(sh.shape[0]), shadow_ms=round(ms_s, 3), shadow_Mrays_s=round(sh.shape[0] / ms_s / 1e3, 1),
                   occluded_fraction=float(occ.float().mean().item()))
        if b in (0, 1):
            rec["shadow_parity"] = parity_sample(oracle, sh.cpu().numpy(), None, occluded=occ.cpu().numpy(), seed=10 + b)
        # Russian roulette on the weight's luminance: 1 at the first bounce, the albedo (0.5) after it
        if b > 0:
            keep = torch.rand(P.shape[0], device="cuda", generator=g) <= 0.5
            P, nl = P[keep], nl[keep]
        # cosine-weighted direction about nl (cosHemisphereSample, include/qmc.h:38-52, via a local frame)
        u1 = torch.rand(P.shape[0], device="cuda", generator=g)
        u2 = torch.rand(P.shape[0], device="cuda", generator=g)
        rr, phi = torch.sqrt(u1), 2 * np.pi * u2
        lx, ly, lz = rr * torch.cos(phi), rr * torch.sin(phi), torch.sqrt(torch.clamp(1 - u1, min=0))
        helper = torch.where(nl[:, 0:1].abs() > 0.9, torch.tensor([0.0, 1.0, 0.0], device="cuda")[None],
                             torch.tensor([1.0, 0.0, 0.0], device="cuda")[None])
        tx = torch.linalg.cross(helper.expand_as(nl), nl)
        tx = tx / tx.norm(dim=1, keepdim=True)
        ty = torch.linalg.cross(nl, tx)
        d = tx * lx[:, None] + ty * ly[:, None] + nl * lz[:, None]
        d = d / d.norm(dim=1, keepdim=True)
        inf = torch.full((P.shape[0], 1), float("inf"), device="cuda")
        rays = torch.cat([P + d * EPS, torch.zeros_like(inf), d, inf], 1).contiguous()
        bounces.append(rec)
    tot_r = sum(x["closest_rays"] + x["shadow_rays"] for x in bounces)
    tot_ms = sum(x["closest_ms"] + x["shadow_ms"] for x in bounces)
    inc = [x for x in bounces if x["bounce"] >= 2]
    print(json.dumps({"config": "C3 diffuse path tracing depth %d, %d spp, %dx%d, same scene" % (args.depth, spp, W, H),
                      "bounces": bounces,
                      "all_batches_Mrays_s": round(tot_r / tot_ms / 1e3, 1),
                      "incoherent_closest_Mrays_s(bounce>=2)": round(sum(x["closest_rays"] for x in inc) / max(1e-9, sum(x["closest_ms"] for x in inc)) / 1e3, 1)}),
          flush=True)


if __name__ == "__main__":
    main()
