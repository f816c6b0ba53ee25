#!/usr/bin/env python3
"""BASELINE configs[2] (C3) and configs[4] (C5, one GPU's view) at their stated sizes through the
device-resident wavefront driver (yart_wavefront_render_diffuse): diffuse path tracing of the Sponza- /
San-Miguel-class stand-in scenes, per-iteration batch statistics and end-to-end Mrays/s.

    python tools/wavefront_bench.py --config c3 [--spp 64 --depth 8 --slots 4194304] [--pools] [--out file.json]
    python tools/wavefront_bench.py --config c5 --spp 256

Size-independent checks at full size: the frames of runs with another slot count and in pool mode are
and without the coherence sort are compared bit for bit (a path's result does not depend on slot, batch, iteration or
the order a wave finds its candidates in: DESIGN.md section 1, order independence); every pixel finite.
The small-size check against the CPU loop over the oracle is tests/test_wavefront_dev.py."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--slots", type=int, default=1 << 22)
    ap.add_argument("--tris", type=int, default=None)
    ap.add_argument("--check-spp", type=int, default=2, help="samples of the scheduling-independence check (0: skip)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from yart_amd import api, scenes

    if args.config == "c3":
        W, H, spp, tris = 1920, 1080, args.spp or 64, args.tris or 262144
        label = "C3: diffuse path tracing depth %d, %d spp, %dx%d, Sponza-class stand-in" % (args.depth, spp, W, H)
    else:
        W, H, spp, tris = 3840, 2160, args.spp or 256, args.tris or 10_000_000
        label = "C5 (one GPU, all tiles): diffuse path tracing depth %d, %d spp, %dx%d, San-Miguel-scale stand-in" % (args.depth, spp, W, H)
    t0 = time.time()
    v, t, s = scenes.architectural(tris, seed=1)
    t1 = time.time()
    acc = api.Accel(v, t, s)
    acc.set_surface_data(v, t, np.zeros_like(v), s)
    t2 = time.time()
    cam = api.camera_basis([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], W, H, 0.785398)
    light, inten, alb = [0.0, 9.0, 0.0], [60.0, 55.0, 50.0], [0.5, 0.5, 0.5]
    out = {"config": label, "scene": {"tris": int(acc.num_triangles), "spheres": int(acc.num_spheres), "device_MB": round(acc.device_bytes / 1e6, 1),
                                      "generate_s": round(t1 - t0, 1), "build_upload_s": round(t2 - t1, 1)},
           "slots": args.slots}
    wf = api.Wavefront(acc, args.slots)

    def run(spp_, regenerate, wfx=wf, flags=api.YART_SORT_RAYS):
        rgb, st, iters = wfx.render_diffuse(cam, W, H, spp_, args.depth, light, inten, alb, seed=1, regenerate=regenerate, flags=flags)
        return rgb, st, iters

    if args.check_spp > 0:
        a, _, _ = run(args.check_spp, True)
        b, _, _ = run(args.check_spp, False)
        wf2 = api.Wavefront(acc, max(1000, args.slots // 3 + 17))
        c, _, _ = run(args.check_spp, True, wf2)
        wf2.close()
        d, _, _ = run(args.check_spp, True, flags=0)           # batches traced in slot order, no coherence sort
        diff = lambda x, y: int((x.view(torch.int32) != y.view(torch.int32)).any(1).sum().item())
        out["checks"] = {"spp": args.check_spp, "pixels": int(a.shape[0]),
                         "pixels_differing_regenerate_vs_pools": diff(a, b),
                         "pixels_differing_other_slot_count": diff(a, c),
                         "pixels_differing_unsorted": diff(a, d),
                         "all_finite": bool(torch.isfinite(a).all().item()), "mean_rgb": [round(float(x), 5) for x in a.mean(0).tolist()]}
        del a, b, c, d
    for name, regen in (("regenerate", True), ("pools", False)):
        run(1, regen)                                         # warm-up
        rgb, st, iters = run(spp, regen)
        rays = st["closest_rays"] + st["shadow_rays"]
        engine_ms = st["ms_closest"] + st["ms_shadow"]
        rec = {"paths": st["paths"], "iterations": st["iterations"], "closest_rays": st["closest_rays"], "shadow_rays": st["shadow_rays"],
               "wall_ms": round(st["ms_total"], 1),
               "ms": {k: round(st["ms_" + k], 1) for k in ("closest", "surface", "shadow", "shade")},
               "end_to_end_Mrays_s": round(rays / st["ms_total"] / 1e3, 1),
               "engine_Mrays_s": round(rays / engine_ms / 1e3, 1),
               "closest_Mrays_s": round(st["closest_rays"] / st["ms_closest"] / 1e3, 1),
               "shadow_Mrays_s": round(st["shadow_rays"] / st["ms_shadow"] / 1e3, 1),
               "batch_rays": args.slots,
               "median_live_closest_rays_per_batch": int(np.median([i["closest_rays"] for i in iters])) if iters else 0,
               "mean_rgb": [round(float(x), 5) for x in rgb.mean(0).tolist()]}
        if not regen:      # per-bounce: iteration k of a pool is bounce k
            per = {}
            b = 0
            for i in iters:
                if i["closest_rays"] == 0:
                    continue
                # a pool starts when the live count jumps back up
                if b > 0 and i["closest_rays"] > last:
                    b = 0
                d = per.setdefault(b, {"closest_rays": 0, "shadow_rays": 0, "ms_closest": 0.0, "ms_shadow": 0.0, "batches": 0})
                d["closest_rays"] += i["closest_rays"]; d["shadow_rays"] += i["shadow_rays"]
                d["ms_closest"] += i["ms_closest"]; d["ms_shadow"] += i["ms_shadow"]; d["batches"] += 1
                last = i["closest_rays"]
                b += 1
            rec["per_bounce"] = [{"bounce": k, "batches": d["batches"], "live_closest_rays_per_batch": d["closest_rays"] // d["batches"],
                                  "closest_Mrays_s": round(d["closest_rays"] / max(1e-9, d["ms_closest"]) / 1e3, 1),
                                  "shadow_Mrays_s": round(d["shadow_rays"] / max(1e-9, d["ms_shadow"]) / 1e3, 1)} for k, d in sorted(per.items())]
        out[name] = rec
        del rgb
    print(json.dumps(out), flush=True)
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
