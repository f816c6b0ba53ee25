#!/usr/bin/env python3
"""Measurement (GPU box): the device integrator against the reference's own trace() frame (tests/golden/g6_trace_diffuse.npz):
relative-luminance RMSE and bias at equal spp, for a few seeds, and what two device renders with different seeds differ by
(the noise floor of the comparison).   python tools/integrator_pin.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api

g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_trace_diffuse.npz")))
W, H, spp, depth = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["max_depth"])
acc = api.Accel(g["verts"], g["tris"], g["spheres"])
acc.set_surface_data(g["verts"], g["tris"], g["vertex_normals"], g["spheres"])
cam = api.camera_basis(g["cam_pos"], g["cam_dir"], W, H, float(g["fov"]))
wf = api.Wavefront(acc, 1 << 18)
lum = np.array([0.2126, 0.7152, 0.0722])
lr = g["rgb"].astype(np.float64) @ lum
frames = []
out = {"frame": "%dx%d" % (W, H), "spp": spp, "depth": depth, "reference_mean_luminance": float(lr.mean()), "seeds": []}
for seed in (11, 12, 13):
    rgb, _, _ = wf.render_diffuse(cam, W, H, spp, depth, g["light_pos"], g["light_intensity"], g["albedo"], seed=seed)
    lg = rgb.cpu().numpy().astype(np.float64) @ lum
    frames.append(lg)
    out["seeds"].append({"seed": seed, "rel_luminance_rmse_vs_reference": float(np.sqrt(np.mean((lg - lr) ** 2)) / lr.mean()),
                         "bias": float(lg.mean() / lr.mean() - 1.0)})
out["rel_luminance_rmse_between_two_device_seeds"] = float(np.sqrt(np.mean((frames[0] - frames[1]) ** 2)) / lr.mean())
print(json.dumps(out))
