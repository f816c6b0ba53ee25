"""GPU: BASELINE configs[3] at its full size (10 M triangles + 1 M spheres, 2^24 rays) through
size-independent properties -- the oracle cannot walk 2^24 rays through an 11 M-primitive scene in test
time (its sampled check runs inside bench.py):
  * idempotence, and independence of launch shape and ray order (sorted persistent == unsorted static);
  * any-hit on the same rays says "occluded" exactly where closest-hit found something;
  * every hit lies inside the ray interval and names an existing primitive;
  * re-tracing each ray with maxT just beyond / just before its hit distance finds the same hit / not
    that hit (the closest hit really is the first thing along the ray);
  * scaling scene and rays by a power of two scales t exactly and keeps the primitive (1 M triangles)."""
import numpy as np
import pytest

from conftest import bits
from yart_amd import api, scenes

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_c4_full_size_properties():
    verts, tris, sph = scenes.random_soup(10_000_000, 1_000_000, s=0.005, seed=42)
    acc = api.Accel(verts, tris, sph)
    rays_np = scenes.random_rays(1 << 24, seed=7)
    rays = torch.from_numpy(rays_np).cuda()
    a = acc.intersect_closest(rays, api.YART_SORT_RAYS)
    b = acc.intersect_closest(rays, api.YART_SORT_RAYS)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    c = acc.intersect_closest(rays, api.YART_STATIC_LAUNCH)
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    occ = acc.intersect_any(rays, api.YART_SORT_RAYS)
    h = api.hits_to_numpy(a)
    hit = h["prim"] != -1
    assert np.array_equal(occ.cpu().numpy() != 0, hit)
    assert 0.95 < hit.mean() < 1.0
    assert (h["t"][hit] > 1e-4).all() and np.isfinite(h["t"][hit]).all()
    assert h["prim"].max() < len(tris) and h["prim"].min() >= -(len(sph) + 1)
    tri = h["prim"] >= 0
    assert (h["u"][tri] >= 0).all() and (h["v"][tri] >= 0).all() and (h["u"][tri] + h["v"][tri] <= 1.0 + 1e-6).all()
    del b, c, occ

    # the closest hit is the first thing along the ray
    idx = np.flatnonzero(hit)[:: max(1, hit.sum() // (1 << 20))][: 1 << 20]
    beyond = rays_np[idx].copy()
    beyond[:, 7] = np.nextafter(h["t"][idx], np.float32(np.inf))
    again = api.hits_to_numpy(acc.intersect_closest(torch.from_numpy(beyond).cuda(), api.YART_SORT_RAYS))
    assert np.array_equal(again["prim"], h["prim"][idx]) and np.array_equal(bits(again["t"]), bits(h["t"][idx]))
    before = rays_np[idx].copy()
    before[:, 7] = h["t"][idx]                          # the interval is open at maxT: this hit no longer counts
    nothing = api.hits_to_numpy(acc.intersect_closest(torch.from_numpy(before).cuda(), api.YART_SORT_RAYS))
    assert (nothing["prim"] == -1).all()
    acc.close()


def test_power_of_two_scaling_is_exact():
    verts, tris, sph = scenes.random_soup(1_000_000, 100_000, s=0.01, seed=5)
    rays = scenes.random_rays(1 << 20, seed=6)
    acc1 = api.Accel(verts, tris, sph)
    h1 = api.hits_to_numpy(acc1.intersect_closest(torch.from_numpy(rays).cuda(), api.YART_REFERENCE_RULE))
    acc1.close()
    k = np.float32(4.0)
    rays2 = rays.copy()
    rays2[:, 0:3] *= k
    acc2 = api.Accel(verts * k, tris, sph * k)
    h2 = api.hits_to_numpy(acc2.intersect_closest(torch.from_numpy(rays2).cuda(), api.YART_REFERENCE_RULE))
    acc2.close()
    # every operation of the tests is exact under scaling by 4 except the comparisons with the absolute
    # epsilon 1e-4 (include/triaccel.h:181, include/sphere.h:65): a surface 5e-5 in front of the origin is
    # ignored in one scene and hit in the other.  Stay clear of it in both.
    far = (h1["prim"] != -1) & (h1["t"] > 1e-3) & (h2["prim"] != -1) & (h2["t"] > 4e-3)
    assert far.mean() > 0.9
    assert np.array_equal(h2["prim"][far], h1["prim"][far])
    assert np.array_equal(bits(h2["t"][far]), bits(h1["t"][far] * k))
    assert np.array_equal(bits(h2["u"][far]), bits(h1["u"][far])) and np.array_equal(bits(h2["v"][far]), bits(h1["v"][far]))
