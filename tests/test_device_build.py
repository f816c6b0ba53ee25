"""GPU: the 4-wide tree built ON the device (YART_BUILD_DEVICE=1, yart_amd/csrc/wide_build_dev.hip)
held to the host-built tree's bars: same structural invariants (every primitive in exactly one leaf, every
quantised child box contains what is below it, rows are the reference's TriAccel bits), the oracle's walk of
the dumped tree equals the kernel's answers bit for bit, and the answers meet the parity bar against the
golden vectors of the real reference and against the oracle on seeded soups / the architectural stand-in."""
import numpy as np
import pytest

from conftest import bits
from yart_amd import api, scenes
from test_gpu_parity import assert_exact, assert_within_bar, dev, regular_mask
from test_wide_tree_cpu import check_structure

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_build(monkeypatch):
    monkeypatch.setenv("YART_BUILD_DEVICE", "1")


def test_device_built_tree_structure_and_parity(golden):
    from oracle import oracle as orc
    g = golden
    acc = api.Accel(g["verts"], g["tris"], g["spheres"])
    on_device, ms = acc.wide_built_on_device
    assert on_device and ms > 0
    nodes, rows_a, rows_bc = acc.wide()
    check_structure(g, nodes, rows_a, rows_bc, acc.wide_depth, 1)
    # the kernel against the golden answers of the real reference
    for launch in (0, api.YART_SORT_RAYS, api.YART_STATIC_LAUNCH):
        got = api.hits_to_numpy(acc.intersect_closest(dev(g["rays"]), launch))
        assert_within_bar(got, g["prim"], g["t"])
        assert np.array_equal(got["prim"], g["prim"]) and np.array_equal(bits(got["t"]), bits(g["t"]))
        assert np.array_equal(acc.intersect_any(dev(g["shadow_rays"]), launch).cpu().numpy(), g["occluded"])
    exact = api.hits_to_numpy(acc.intersect_closest(dev(g["rays"]), api.YART_REFERENCE_RULE))
    assert_exact(exact, g["prim"], g["t"], g["u"], g["v"])
    # the oracle's walk of the dumped tree against the kernel, bit for bit
    rays = np.ascontiguousarray(g["rays"])
    want = orc.wide_intersect(nodes, rows_a, rows_bc, rays, ref_nodes=acc.nodes(), cover=api.scene_cover(g["verts"], g["tris"], g["spheres"]))
    got = api.hits_to_numpy(acc.intersect_closest(dev(rays), api.YART_RAYS_REGULAR))
    assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(bits(got["t"]), bits(want["t"]))
    acc.close()


@pytest.mark.parametrize("scene", ["soup", "architectural", "spheres_only", "with_nan"])
def test_device_built_tree_against_oracle(scene):
    from oracle import oracle as orc
    if scene == "soup":
        v, t, s = scenes.random_soup(200000, 3000, s=0.01, seed=42)
        rays = scenes.random_rays(1 << 18, seed=7)
    elif scene == "architectural":
        v, t, s = scenes.architectural(60000, seed=1)
        rays = scenes.camera_rays([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], 320, 180, 0.785398)
    elif scene == "spheres_only":
        _, _, s = scenes.random_soup(0, 3000, s=0.05, seed=12)
        v, t = np.zeros((0, 3), np.float32), np.zeros((0, 5), np.int32)
        rays = scenes.random_rays(20000, seed=13)
    else:
        v, t, s = scenes.random_soup(5000, 50, s=0.05, seed=3)
        v = v.copy()
        v[7] = np.nan
        v[100, 1] = np.inf
        rays = scenes.random_rays(20000, seed=4)
    shadow = rays.copy()
    shadow[:, 7] = 0.5
    acc = api.Accel(v, t, s)
    assert acc.wide_built_on_device[0]
    o = orc.Oracle(v, t, s)
    want = o.intersect(rays, mode=0, nthreads=16)
    want_occ = o.intersect_shadow(shadow, mode=0, nthreads=16)
    for launch in (0, api.YART_SORT_RAYS):
        got = api.hits_to_numpy(acc.intersect_closest(dev(rays), launch))
        assert_within_bar(got, want["prim"], want["t"])
        assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(bits(got["t"]), bits(want["t"]))
        assert np.array_equal(acc.intersect_any(dev(shadow), launch).cpu().numpy(), want_occ)
    nodes, rows_a, rows_bc = acc.wide()
    assert len(rows_a) <= len(t) + len(s) and acc.wide_depth * 3 + 1 <= 64
    acc.close()


def test_tiny_scenes_fall_back_to_the_host_builder():
    """Fewer than two primitives: nothing to sort; the host builder makes the tree and results are unchanged."""
    v, t, s = scenes.random_soup(1, 0, s=0.3, seed=2)
    acc = api.Accel(v, t, s)
    assert not acc.wide_built_on_device[0]
    got = api.hits_to_numpy(acc.intersect_closest(dev(scenes.random_rays(1000, seed=1))))
    assert got["prim"].max() <= 0
    acc.close()


def test_host_builders_still_serve_large_scenes_on_request(monkeypatch):
    """From 2^18 primitives the device builds by default; YART_BUILD_DEVICE=0 must keep giving the host-built trees
    (binned SAH / std::partition recursion), with the same answers."""
    from oracle import oracle as orc
    v, t, s = scenes.random_soup(300000, 3000, s=0.01, seed=17)
    rays = scenes.random_rays(1 << 17, seed=5)
    out = {}
    for mode in ("default", "0"):
        if mode == "0":
            monkeypatch.setenv("YART_BUILD_DEVICE", "0")
        else:
            monkeypatch.delenv("YART_BUILD_DEVICE", raising=False)
        acc = api.Accel(v, t, s)
        assert acc.wide_built_on_device[0] == (mode == "default")
        assert acc.ref_built_on_device[0] == (mode == "default")
        out[mode] = (api.hits_to_numpy(acc.intersect_closest(dev(rays))), acc.nodes().tobytes(), acc.triaccel().tobytes())
        acc.close()
    assert out["default"][1] == out["0"][1] and out["default"][2] == out["0"][2]      # the reference's arrays: identical
    want = orc.Oracle(v, t, s).intersect(rays, mode=0, nthreads=16)
    for mode in out:
        assert_within_bar(out[mode][0], want["prim"], want["t"])
    assert np.array_equal(out["default"][0]["prim"], out["0"][0]["prim"]) and np.array_equal(bits(out["default"][0]["t"]), bits(out["0"][0]["t"]))
