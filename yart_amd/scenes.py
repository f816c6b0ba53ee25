"""Synthetic scene and ray-set generators for the configurations of BASELINE.json
(SURVEY.md 8d, C1-C5).  Pure numpy; deterministic for a given seed.

Exchange format (the one the C-ABI takes, include/yart_amd.h):
  verts   float32 [nV,3]
  tris    int32   [nT,5]  = v0, v1, v2 (global vertex indices), meshId, triId
                            listed mesh by mesh, triId counting from 0 inside a mesh --
                            the order the reference enumerates them (src/bvhaccel.cpp:328-343)
  spheres float32 [nS,4]  = cx, cy, cz, radius     (Scene::shapes_ order, include/scene.h:142)
  rays    float32 [n,8]   = orig xyz, minT, dir xyz, maxT   (include/vector.h:238-250)
"""
import numpy as np

INF = np.float32(np.inf)


def random_soup(n_tris, n_spheres=0, s=0.005, seed=42, tris_per_mesh=100):
    """C4: triangle soup, first vertices uniform in [0,1]^3, edge vectors uniform in
    [-s,s]^3; spheres with uniform centres and radius uniform in [0.5 s, s]."""
    rng = np.random.RandomState(seed)
    c = rng.random_sample((n_tris, 3)).astype(np.float32)
    e1 = ((rng.random_sample((n_tris, 3)) * 2 - 1) * s).astype(np.float32)
    e2 = ((rng.random_sample((n_tris, 3)) * 2 - 1) * s).astype(np.float32)
    verts = np.empty((n_tris, 3, 3), np.float32)
    verts[:, 0] = c
    verts[:, 1] = c + e1
    verts[:, 2] = c + e2
    verts = verts.reshape(-1, 3)
    idx = np.arange(n_tris, dtype=np.int64)
    tris = np.empty((n_tris, 5), np.int32)
    tris[:, 0] = idx * 3
    tris[:, 1] = idx * 3 + 1
    tris[:, 2] = idx * 3 + 2
    tris[:, 3] = idx // tris_per_mesh
    tris[:, 4] = idx % tris_per_mesh
    sph = np.empty((n_spheres, 4), np.float32)
    if n_spheres:
        sph[:, :3] = rng.random_sample((n_spheres, 3))
        sph[:, 3] = (0.5 + 0.5 * rng.random_sample(n_spheres)) * s
    return verts, tris, sph


def _grid_patch(origin, du, dv, nu, nv, rng, jitter):
    """nu x nv quad patch (2 triangles per cell) spanned by du, dv from origin."""
    us = np.linspace(0.0, 1.0, nu + 1)
    vs = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(us, vs, indexing="ij")
    p = origin[None, None, :] + uu[..., None] * du[None, None, :] + vv[..., None] * dv[None, None, :]
    if jitter > 0:
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        p = p + (rng.random_sample(uu.shape)[..., None] - 0.5) * jitter * n[None, None, :]
    p = p.reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    t1 = np.stack([a, b, a + 1], 1)
    t2 = np.stack([b + 1, a + 1, b], 1)
    return p.astype(np.float32), np.concatenate([t1, t2], 0).astype(np.int32)


def architectural(n_tris_target=262144, seed=1):
    """C2/C3/C5 stand-in for the Sponza / San Miguel class of scenes (no assets and no
    network in the build environment): a closed atrium -- floor, ceiling, four walls as
    large tessellated occluders -- two rows of tessellated columns, arches of fine
    detail and clutter of small objects, sized to about `n_tris_target` triangles.
    Extent roughly [-12,12] x [0,10] x [-5,5] like the usual Sponza export."""
    rng = np.random.RandomState(seed)
    patches = []

    def add(p, t):
        patches.append((p, t))

    # budget: 35 % shell, 35 % columns, 30 % clutter
    shell_cells = int(n_tris_target * 0.35 / 2 / 6)
    g = max(2, int(np.sqrt(shell_cells)))
    X0, X1, Y0, Y1, Z0, Z1 = -12.0, 12.0, 0.0, 10.0, -5.0, 5.0
    o = np.array
    add(*_grid_patch(o([X0, Y0, Z0]), o([X1 - X0, 0, 0]), o([0, 0, Z1 - Z0]), g, g, rng, 0.02))   # floor
    add(*_grid_patch(o([X0, Y1, Z0]), o([0, 0, Z1 - Z0]), o([X1 - X0, 0, 0]), g, g, rng, 0.02))   # ceiling
    add(*_grid_patch(o([X0, Y0, Z0]), o([0, Y1 - Y0, 0]), o([X1 - X0, 0, 0]), g, g, rng, 0.05))   # back
    add(*_grid_patch(o([X0, Y0, Z1]), o([X1 - X0, 0, 0]), o([0, Y1 - Y0, 0]), g, g, rng, 0.05))   # front
    add(*_grid_patch(o([X0, Y0, Z0]), o([0, 0, Z1 - Z0]), o([0, Y1 - Y0, 0]), g, g, rng, 0.05))   # left
    add(*_grid_patch(o([X1, Y0, Z0]), o([0, Y1 - Y0, 0]), o([0, 0, Z1 - Z0]), g, g, rng, 0.05))   # right
    # columns: tessellated cylinders
    n_cols = 20
    col_tris = int(n_tris_target * 0.35 / n_cols)
    seg = max(8, int(np.sqrt(col_tris / 2)))
    for c in range(n_cols):
        cx = X0 + 2.0 + (c // 2) * ((X1 - X0 - 4.0) / (n_cols // 2 - 1))
        cz = -2.5 if c % 2 == 0 else 2.5
        ang = np.linspace(0, 2 * np.pi, seg + 1)
        ys = np.linspace(Y0, Y1 * 0.7, seg + 1)
        aa, yy = np.meshgrid(ang, ys, indexing="ij")
        r = 0.4 + 0.03 * np.sin(8 * aa) + 0.02 * rng.random_sample(aa.shape)
        p = np.stack([cx + r * np.cos(aa), yy, cz + r * np.sin(aa)], -1).reshape(-1, 3)
        i, j = np.meshgrid(np.arange(seg), np.arange(seg), indexing="ij")
        a = (i * (seg + 1) + j).ravel()
        b = a + (seg + 1)
        t = np.concatenate([np.stack([a, b, a + 1], 1), np.stack([b + 1, a + 1, b], 1)], 0)
        add(p.astype(np.float32), t.astype(np.int32))
    # clutter: small randomly oriented tessellated blobs (plants, drapes, pots)
    have = sum(len(t) for _, t in patches)
    n_blobs = 400
    blob_tris = max(8, (n_tris_target - have) // n_blobs)
    bs = max(2, int(np.sqrt(blob_tris / 2)))
    for _ in range(n_blobs):
        ctr = np.array([rng.uniform(X0 + 1, X1 - 1), rng.uniform(Y0, Y1 * 0.8), rng.uniform(Z0 + 0.5, Z1 - 0.5)])
        th = np.linspace(0.05, np.pi - 0.05, bs + 1)
        ph = np.linspace(0, 2 * np.pi, bs + 1)
        tt, pp = np.meshgrid(th, ph, indexing="ij")
        rad = rng.uniform(0.1, 0.5) * (1 + 0.2 * np.sin(5 * pp) * np.sin(3 * tt))
        p = np.stack([ctr[0] + rad * np.sin(tt) * np.cos(pp), ctr[1] + rad * np.cos(tt),
                      ctr[2] + rad * np.sin(tt) * np.sin(pp)], -1).reshape(-1, 3)
        i, j = np.meshgrid(np.arange(bs), np.arange(bs), indexing="ij")
        a = (i * (bs + 1) + j).ravel()
        b = a + (bs + 1)
        t = np.concatenate([np.stack([a, b, a + 1], 1), np.stack([b + 1, a + 1, b], 1)], 0)
        add(p.astype(np.float32), t.astype(np.int32))

    verts, tris = [], []
    voff = 0
    mesh = 0
    for p, t in patches:
        # split big patches into meshes of <= 128 triangles that share the patch's
        # vertex block (the reference copies a mesh's vertex vector per triangle at
        # build time, include/triangle.h:188 -- small meshes keep it buildable)
        verts.append(p)
        for s0 in range(0, len(t), 128):
            tt = t[s0:s0 + 128]
            rec = np.empty((len(tt), 5), np.int32)
            rec[:, :3] = tt + voff
            rec[:, 3] = mesh
            rec[:, 4] = np.arange(len(tt))
            tris.append(rec)
            mesh += 1
        voff += len(p)
    return (np.concatenate(verts, 0).astype(np.float32), np.concatenate(tris, 0),
            np.zeros((0, 4), np.float32))


def uniform_sphere_dirs(u1, u2):
    """uniformSphereSample, include/qmc.h:82-94 (fp32)."""
    u1 = u1.astype(np.float32)
    u2 = u2.astype(np.float32)
    z = np.float32(1.0) - np.float32(2.0) * u1
    r = np.sqrt(np.maximum(np.float32(0.0), np.float32(1.0) - z * z)).astype(np.float32)
    phi = (np.float32(2.0) * np.float32(3.14159265358979) * u2).astype(np.float32)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], -1).astype(np.float32)


def random_rays(n, seed=7, lo=0.0, hi=1.0, tmax=None):
    """C4 ray set: origins uniform in the cube, directions uniform on the sphere.
    tmax=None -> +inf (closest-hit set); tmax=x -> maxT uniform in (0,x) (any-hit set)."""
    rng = np.random.RandomState(seed)
    rays = np.empty((n, 8), np.float32)
    rays[:, 0:3] = lo + (hi - lo) * rng.random_sample((n, 3))
    rays[:, 3] = 0.0
    rays[:, 4:7] = uniform_sphere_dirs(rng.random_sample(n), rng.random_sample(n))
    if tmax is None:
        rays[:, 7] = INF
    else:
        rays[:, 7] = (rng.random_sample(n) * tmax).astype(np.float32)
    return rays


def camera_rays(pos, direction, width, height, fov, xs=None, ys=None):
    """Camera::Camera + Camera::sample, include/camera.h:9-31 (fp32), pixel centres by
    default (x, y integer -> the +0.5 inside sample())."""
    f = np.float32
    pos = np.asarray(pos, f)
    d = np.asarray(direction, f)
    d = (d * (f(1.0) / np.sqrt((d * d).sum(dtype=f)))).astype(f)
    up = np.array([0, 1, 0], f)
    right = (np.cross(d, up).astype(f) * f(f(width) * f(fov) / f(height))).astype(f)
    upv = np.cross(right, d).astype(f)
    upv = (upv * (f(1.0) / np.sqrt((upv * upv).sum(dtype=f))) * f(fov)).astype(f)
    if xs is None:
        ys, xs = np.meshgrid(np.arange(height, dtype=f), np.arange(width, dtype=f), indexing="ij")
        xs = xs.ravel()
        ys = ys.ravel()
    a = ((xs.astype(f) + f(0.5)) / f(width) - f(0.5)).astype(f)
    b = (-(ys.astype(f) + f(0.5)) / f(height) + f(0.5)).astype(f)
    dd = (right[None] * a[:, None] + upv[None] * b[:, None] + d[None]).astype(f)
    dd = (dd * (f(1.0) / np.sqrt((dd * dd).sum(-1, dtype=f)))[:, None]).astype(f)
    rays = np.empty((len(a), 8), f)
    rays[:, 0:3] = pos
    rays[:, 3] = 0
    rays[:, 4:7] = dd
    rays[:, 7] = INF
    return rays
