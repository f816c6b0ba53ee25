#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- makes tests/golden/g6_trace_diffuse.npz: a small all-Lambertian scene with one point light
rendered by the REAL reference's integrator (trace(), src/renderer.cpp:29-136, at its native 4096 samples per pixel and
10 bounces, its own Lambertian, PointLight, Rng, Camera) through oracle/_ref (ref_shim.cpp: ref_trace_diffuse).  The
device integrator (yart_wavefront_render_diffuse) is held to this frame at equal spp (tests/test_wavefront_dev.py:
SURVEY 8d's bar for independently seeded renders, <= 2 % relative-luminance RMSE).  Runs only where /root/reference is.
    python oracle/make_golden_trace.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc


def quad(p, e1, e2, n1=1, n2=1):
    """(n1 x n2) grid of quads sharing vertices -> verts, triangles (vertex numbers local to the mesh)"""
    p, e1, e2 = (np.asarray(a, np.float32) for a in (p, e1, e2))
    v = [p + e1 * (i / n1) + e2 * (j / n2) for j in range(n2 + 1) for i in range(n1 + 1)]
    t = []
    for j in range(n2):
        for i in range(n1):
            a = j * (n1 + 1) + i
            t += [(a, a + 1, a + n1 + 2), (a, a + n1 + 2, a + n1 + 1)]
    return np.asarray(v, np.float32), t


def uv_sphere(c, r, nu=12, nv=8):
    """a closed mesh with shared vertices: the reference generates smooth vertex normals for it"""
    v, t = [], []
    for j in range(nv + 1):
        th = np.pi * j / nv
        for i in range(nu):
            ph = 2 * np.pi * i / nu
            v.append([c[0] + r * np.sin(th) * np.cos(ph), c[1] + r * np.cos(th), c[2] + r * np.sin(th) * np.sin(ph)])
    for j in range(nv):
        for i in range(nu):
            a, b = j * nu + i, j * nu + (i + 1) % nu
            if j > 0:
                t.append((a, b, b + nu))
            if j < nv - 1:
                t.append((a, b + nu, a + nu))
    return np.asarray(v, np.float32), t


def flat_box(lo, hi):
    """six faces with vertices of their own: flat shading through the same normal generation"""
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    d = hi - lo
    faces = [(lo, (d[0], 0, 0), (0, d[1], 0)), (lo, (0, 0, d[2]), (0, d[1], 0)), (lo, (d[0], 0, 0), (0, 0, d[2])),
             (hi, (-d[0], 0, 0), (0, -d[1], 0)), (hi, (0, 0, -d[2]), (0, -d[1], 0)), (hi, (-d[0], 0, 0), (0, 0, -d[2]))]
    vs, ts = [], []
    for p, e1, e2 in faces:
        v, t = quad(p, e1, e2)
        ts += [tuple(x + 4 * (len(vs)) for x in tri) for tri in t]
        vs.append(v)
    return np.concatenate(vs), ts


def scene():
    meshes = [quad([-7, 0, -7], [14, 0, 0], [0, 0, 14], 3, 3),          # floor
              quad([-7, 0, -4], [14, 0, 0], [0, 7, 0], 2, 2),           # back wall
              quad([-5, 0, -4], [0, 0, 9], [0, 7, 0], 2, 2),            # left wall
              uv_sphere([0.0, 1.2, 0.0], 1.2),
              flat_box([2.0, 0.0, 0.2], [3.4, 1.5, 1.6])]
    verts, tris = [], []
    base = 0
    for m, (v, t) in enumerate(meshes):
        verts.append(v)
        tris += [(a + base, b + base, c + base, m, k) for k, (a, b, c) in enumerate(t)]
        base += len(v)
    spheres = np.array([[-2.6, 0.8, 1.0, 0.8], [1.0, 0.5, 2.6, 0.5]], np.float32)
    return np.concatenate(verts).astype(np.float32), np.asarray(tris, np.int32), spheres


def main():
    verts, tris, spheres = scene()
    job = dict(albedo=[0.5, 0.6, 0.7], light_pos=[1.5, 5.0, 3.5], light_intensity=[60.0, 55.0, 50.0],
               cam_pos=[0.5, 3.0, 9.0], cam_dir=[-0.05, -0.22, -1.0], fov=0.785398, width=96, height=64)
    rgb, normals = orc.reference_trace_diffuse(verts, tris, spheres, nthreads=8, **job)
    out = os.path.join(ROOT, "tests", "golden", "g6_trace_diffuse.npz")
    np.savez_compressed(out, verts=verts, tris=tris, spheres=spheres, vertex_normals=normals, rgb=rgb, spp=np.int32(4096),
                        max_depth=np.int32(10), **{k: np.asarray(v, np.float32 if not isinstance(v, int) else np.int32) for k, v in job.items()})
    lum = rgb @ np.array([0.2126, 0.7152, 0.0722], np.float32)
    print("wrote", out, os.path.getsize(out), "bytes; pixels", len(rgb), "mean luminance", float(lum.mean()), "black pixels", int((lum == 0).sum()))


if __name__ == "__main__":
    main()
