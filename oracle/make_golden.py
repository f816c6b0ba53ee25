"""TEST INFRASTRUCTURE ONLY.  Generates the golden vectors under tests/golden/ from
the REAL reference (oracle/_ref/libyart_ref.so = /root/reference compiled by
oracle/Makefile, -ffp-contract=off) -- runs only in the authoring container.

Every fixture is data: scene arrays, ray arrays and what the reference's
Scene::intersect / Scene::intersectShadow / BvhAccel returned for them.  The
`_fma` fields hold the same outputs from the reference built with GCC's default
contraction (-ffp-contract=fast -mfma) to record the low-bit spread a different
build of the reference itself shows.

    python oracle/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from yart_amd import scenes       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
INF = np.float32(np.inf)


def run_reference(verts, tris, sph, rays, shadow_rays):
    ref = orc.Reference(verts, tris, sph)
    fma = orc.Reference(verts, tris, sph, so=orc.REF_FMA_SO)
    h = ref.intersect(rays)
    hf = fma.intersect(rays)
    assert h["ambiguous"].max() <= 1, "primitive identification ambiguous; change the fixture"
    out = dict(
        verts=verts, tris=tris, spheres=sph, rays=rays, shadow_rays=shadow_rays,
        nodes=ref.nodes().view(np.uint8), triaccel=ref.triaccel().view(np.uint8),
        vertex_normals=ref.vertex_normals(),
        hit=h["hit"].astype(np.uint8), t=h["t"], u=h["u"], v=h["v"], prim=h["prim"],
        normal=h["normal"], shading_normal=h["shadingNormal"],
        occluded=ref.intersect_shadow(shadow_rays),
        hit_fma=hf["hit"].astype(np.uint8), t_fma=hf["t"], u_fma=hf["u"], v_fma=hf["v"], prim_fma=hf["prim"],
        occluded_fma=fma.intersect_shadow(shadow_rays),
    )
    return out


def save(name, d):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **d)
    same = (d["prim"] == d["prim_fma"]).mean()
    m = (d["hit"] == 1) & (d["hit_fma"] == 1)
    rel = np.abs(d["t"][m] - d["t_fma"][m]) / np.maximum(np.abs(d["t"][m]), 1e-30)
    print("%-18s %7d rays  hits %6d  prim(fma)==prim %.6f  max rel dt(fma) %.3g  occluded %.3f  (%d KB)"
          % (name, len(d["rays"]), int(d["hit"].sum()), same, rel.max() if rel.size else 0.0,
             d["occluded"].mean(), os.path.getsize(path) // 1024))


def g1_cornell():
    verts, tris, sph = orc.reference_cornell()
    W = H = 64
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    xy = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.float32)
    rays = orc.reference_camera_rays([50.0, 48.0, 220.0], [0.0, -0.042612, -1.0], W, H, 0.785398, xy)
    rng = np.random.RandomState(11)
    sh = np.zeros((4096, 8), np.float32)
    sh[:, 0:3] = rng.random_sample((4096, 3)) * [98.0, 78.0, 228.0] + 1.0
    sh[:, 4:7] = orc.reference_uniform_sphere(rng.random_sample((4096, 2)))
    sh[:, 7] = (rng.random_sample(4096) * 120.0).astype(np.float32)
    return run_reference(verts, tris, sph, rays, sh)


def g2_soup():
    verts, tris, sph = scenes.random_soup(10000, 64, s=0.05, seed=42)
    rng = np.random.RandomState(7)
    n = 16384
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.random_sample((n, 3))
    rays[:, 4:7] = orc.reference_uniform_sphere(rng.random_sample((n, 2)))
    rays[:, 7] = INF
    sh = rays.copy()
    sh[:, 7] = (rng.random_sample(n) * 0.5).astype(np.float32)
    return run_reference(verts, tris, sph, rays, sh)


def g3_edges():
    """Edge cases of SURVEY.md 8c (G3).  One scene, 16 triangles (count % 8 == 0):
    an axis-aligned box made of flat (zero-thickness bbox) quads, two coplanar
    duplicate triangles in DIFFERENT meshes (exact t tie: first found wins), two
    degenerate triangles (inf/NaN TriAccel records), plus spheres that overlap
    each other and the box."""
    V = []
    T = []

    def mesh(vs, ts):
        base = len(V)
        mid = (T[-1][3] + 1) if T else 0
        V.extend(vs)
        for k, t in enumerate(ts):
            T.append([t[0] + base, t[1] + base, t[2] + base, mid, k])

    # box [0,2]^3 : 6 quads = 12 triangles
    c = [(0, 0, 0), (2, 0, 0), (0, 2, 0), (2, 2, 0), (0, 0, 2), (2, 0, 2), (0, 2, 2), (2, 2, 2)]
    faces = [(0, 1, 2, 3), (4, 6, 5, 7), (0, 2, 4, 6), (1, 5, 3, 7), (0, 4, 1, 5), (2, 3, 6, 7)]
    # (one mesh per TRIANGLE: the two triangles of a quad tie exactly on the shared
    # diagonal; separate meshes let the harness tell which one the reference kept)
    for f in faces:
        mesh([c[i] for i in f], [(0, 1, 2)])
        mesh([c[i] for i in f], [(3, 2, 1)])
    # coplanar duplicates at z = 1 (inside the box), separate meshes
    dup = [(0.25, 0.25, 1.0), (1.75, 0.25, 1.0), (0.25, 1.75, 1.0)]
    mesh(dup, [(0, 1, 2)])
    mesh(dup, [(0, 1, 2)])
    # degenerate: repeated vertex, and three collinear points
    mesh([(0.5, 0.5, 0.5), (0.5, 0.5, 0.5), (1.0, 1.0, 0.5)], [(0, 1, 2)])
    mesh([(0.2, 0.2, 1.5), (0.4, 0.4, 1.5), (0.8, 0.8, 1.5)], [(0, 1, 2)])
    verts = np.array(V, np.float32)
    tris = np.array(T, np.int32)
    assert len(tris) % 8 == 0
    sph = np.array([[1.0, 1.0, 1.0, 0.5],      # centre of the box
                    [1.0, 1.0, 1.0, 0.5],      # exact duplicate: first in shapes_ order wins
                    [1.2, 1.0, 1.0, 0.5],      # overlapping
                    [3.5, 1.0, 1.0, 0.75],     # outside the box
                    [1.0, 1.0, 1.0, 5.0]],     # encloses everything (origins inside -> far root)
                   np.float32)
    R = []

    def ray(o, d, tmin=0.0, tmax=np.inf):
        d = np.asarray(d, np.float64)
        d = d / np.linalg.norm(d)
        R.append([o[0], o[1], o[2], tmin, d[0], d[1], d[2], tmax])

    # axis-parallel rays (zero direction components -> inf / NaN slabs), incl. -0.0
    for o in [(1, 1, -3), (1, 1, 5), (0.5, 1.5, -1), (1, 1, 1), (0, 1, -3), (2, 1, -3), (0, 0, -3),
              (1.0, 0.25, -3), (0.25, 0.25, -3), (1, 1, 1.0), (1, 1, 0.9999)]:
        for d in [(0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0),
                  (-0.0, 0.0, 1.0), (0.0, -0.0, -1.0)]:
            ray(o, d)
    # diagonals through edges / corners, in-plane rays
    for o, d in [((-1, -1, -1), (1, 1, 1)), ((3, 3, 3), (-1, -1, -1)), ((-1, 1, 1), (1, 0, 0)),
                 ((-1, 0, 0), (1, 0, 0)), ((-1, 2, 2), (1, 0, 0)), ((1, 1, -1), (0, 1e-8, 1)),
                 ((1, -1, 1), (1e-20, 1, 0)), ((0.25, 0.25, -1), (0, 0, 1)), ((1, 1, 3), (0.001, 0.002, -1))]:
        ray(o, d)
    # finite intervals: tmax just short of / just past a surface, tmin past the first surface
    for tmax in [0.5, 1.0, 0.99999, 1.00001, 2.0, 3.0, 3.00001, 1e-4, 1e-5]:
        ray((1, 1, -1), (0, 0, 1), 0.0, tmax)
        ray((1.3, 0.7, -1), (0.1, 0.05, 1), 0.0, tmax)
    for tmin in [0.5, 1.5, 2.5, 10.0]:
        ray((1, 1, -1), (0, 0, 1), tmin, np.inf)
    rng = np.random.RandomState(5)
    base = np.array(R, np.float32)
    # random rays from inside and outside
    n = 3000
    rr = np.zeros((n, 8), np.float32)
    rr[:, 0:3] = rng.random_sample((n, 3)) * 6.0 - 2.0
    rr[:, 4:7] = orc.reference_uniform_sphere(rng.random_sample((n, 2)))
    rr[:, 7] = INF
    rays0 = np.concatenate([base, rr], 0)
    # rays that START on a surface: re-launch from the hit points of the set so far
    ref = orc.Reference(verts, tris, sph)
    h = ref.intersect(rays0)
    m = h["hit"] == 1
    p = rays0[m, 0:3] + rays0[m, 4:7] * h["t"][m, None]
    onsurf = np.zeros((int(m.sum()), 8), np.float32)
    onsurf[:, 0:3] = p
    onsurf[:, 4:7] = orc.reference_uniform_sphere(rng.random_sample((len(p), 2)))
    onsurf[:, 7] = INF
    cont = rays0[m].copy()                       # continue in the same direction from the hit point
    cont[:, 0:3] = p
    near = cont.copy()
    near[:, 0:3] = p - rays0[m, 4:7] * np.float32(1e-4)   # hit expected at t ~ 1e-4 (the acceptance threshold)
    rays = np.concatenate([rays0, onsurf, cont, near], 0).astype(np.float32)
    sh = rays.copy()
    fin = ~np.isfinite(sh[:, 7])
    sh[fin, 7] = (rng.random_sample(int(fin.sum())) * 4.0).astype(np.float32)
    return run_reference(verts, tris, sph, rays, sh)


def g4_trace():
    """the reference's own trace() (src/renderer.cpp:29-136, 4096 spp,
    <= 10 bounces, Rng(y*width+x)) on 32 pixels of the 512x512 Cornell frame of BASELINE configs[0]:
    the pixels it accumulated, and a strided sample of the ray stream it issued (camera rays,
    secondary rays after BSDF sampling, shadow rays) with the reference's answers for them."""
    verts, tris, sph = orc.reference_cornell()
    W = H = 512
    gx, gy = np.meshgrid(np.arange(8) * 64 + 29, np.arange(4) * 128 + 67, indexing="ij")
    pixels = np.stack([gx.ravel(), gy.ravel()], -1).astype(np.int32)
    rgb = orc.reference_trace_cornell(W, H, pixels, mode=0, nthreads=8)
    rgb1, closest, shadow = orc.reference_trace_cornell(W, H, pixels, mode=1)
    assert np.array_equal(rgb, rgb1)
    n_closest, n_shadow = len(closest), len(shadow)
    closest = closest[:: max(1, len(closest) // 16384)][:16384]
    shadow = shadow[:: max(1, len(shadow) // 16384)][:16384]
    d = run_reference(verts, tris, sph, np.ascontiguousarray(closest), np.ascontiguousarray(shadow))
    d.update(pixels=pixels, rgb=rgb, frame=np.array([W, H], np.int32),
             stream_counts=np.array([n_closest, n_shadow], np.int64))
    return d


if __name__ == "__main__":
    orc.build_reference()
    os.makedirs(OUT, exist_ok=True)
    save("g1_cornell.npz", g1_cornell())
    save("g2_soup.npz", g2_soup())
    save("g3_edges.npz", g3_edges())
    save("g4_trace.npz", g4_trace())
