"""GPU: the REFERENCE's own tree built on the device (YART_BUILD_DEVICE=1, yart_amd/csrc/ref_build_dev.hip): the
flattened node array in the reference's order and the TriAccel records in its leaf order must be, byte for byte,
what the real reference produced (golden vectors) and what the oracle / the host builder produce on seeded
scenes -- including the cases where the data-parallel restatement is delicate: the pairing of std::partition's
swaps on inputs full of ties, one-sided splits (std::nth_element median), signed zeros in the boxes (the sign
of the first zero of a range wins), degenerate triangles (NaN-bearing records), infinities."""
import numpy as np
import pytest

from yart_amd import api, scenes
from test_gpu_parity import assert_exact, dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_build(monkeypatch):
    monkeypatch.setenv("YART_BUILD_DEVICE", "1")


def soup_tris(v):
    n = len(v) // 3
    t = np.zeros((n, 5), np.int32)
    t[:, 0] = np.arange(n) * 3
    t[:, 1] = t[:, 0] + 1
    t[:, 2] = t[:, 0] + 2
    t[:, 3] = np.arange(n) // 7
    t[:, 4] = np.arange(n) % 7
    return t


def check_against_oracle(v, t, expect_device=True):
    from oracle import oracle as orc
    s = np.zeros((0, 4), np.float32)
    acc = api.Accel(v, t, s)
    on, ms, medians, projections = acc.ref_built_on_device
    assert on == expect_device
    o = orc.Oracle(v, t, s)
    nodes, ta = acc.nodes(), acc.triaccel()
    want_nodes, want_ta = o.nodes(), o.triaccel()
    assert len(nodes) == len(want_nodes)
    assert nodes.tobytes() == want_nodes.tobytes()
    assert ta.tobytes() == want_ta.tobytes()
    return acc, o, medians, projections


def test_golden_arrays_of_the_real_reference(golden):
    g = golden
    acc = api.Accel(g["verts"], g["tris"], g["spheres"])
    assert acc.ref_built_on_device[0] == (len(g["tris"]) >= 1024)
    assert acc.nodes().tobytes() == g["nodes"].tobytes()
    assert acc.triaccel().tobytes() == g["triaccel"].tobytes()
    # the sibling-pair copy of the same tree, through the kernel that walks it with the reference's rule
    exact = api.hits_to_numpy(acc.intersect_closest(dev(g["rays"]), api.YART_REFERENCE_RULE))
    assert_exact(exact, g["prim"], g["t"], g["u"], g["v"])
    acc.close()


@pytest.mark.parametrize("n", [1024, 1025, 4099, 70001, 400000])
def test_ragged_sizes(n):
    v, t, _ = scenes.random_soup(n, 0, s=0.1 if n < 100000 else 0.01, seed=n + 1, tris_per_mesh=5)
    acc, _, _, _ = check_against_oracle(v, t)
    acc.close()


def test_architectural_scene_and_queries_through_the_reference_tree():
    v, t, s = scenes.architectural(60000, seed=1)
    acc, o, _, _ = check_against_oracle(v, t)
    rays = scenes.camera_rays([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], 320, 180, 0.785398)
    want = o.intersect(rays, mode=0, nthreads=16)
    got = api.hits_to_numpy(acc.intersect_closest(dev(rays), api.YART_REFERENCE_RULE))
    assert_exact(got, want["prim"], want["t"], want["u"], want["v"])
    acc.close()


def test_ties_everywhere():
    """Coordinates on a coarse grid: many centres equal the midpoint or each other, every level has ties."""
    rng = np.random.RandomState(5)
    n = 50000
    base = rng.randint(-8, 8, size=(n, 1, 3)).astype(np.float32)
    v = (base + rng.randint(0, 3, size=(n, 3, 3)).astype(np.float32) * 0.5).reshape(-1, 3)
    acc, _, medians, _ = check_against_oracle(v, soup_tris(v))
    assert medians > 0          # thousands of triangles share a centre: those ranges end in the median path
    acc.close()


def test_one_sided_splits_take_the_median_path():
    """2000 identical triangles among others: ranges of them split one-sidedly (src/bvhaccel.cpp:133-143)."""
    v0, _, _ = scenes.random_soup(3000, 0, s=0.1, seed=9)
    same = np.tile(np.array([[0.25, 0.25, 0.25], [0.5, 0.25, 0.25], [0.25, 0.5, 0.25]], np.float32), (2000, 1))
    v = np.concatenate([v0[:4500], same, v0[4500:]])
    acc, _, medians, _ = check_against_oracle(v, soup_tris(v))
    assert medians > 0
    acc.close()


def test_signed_zeros_in_boxes():
    """min / max keep the FIRST of two equal values: a box bound that is zero carries the sign of the first zero
    of the range in the builder's order (include/bbox.h:70-90)."""
    rng = np.random.RandomState(11)
    n = 30000
    # small triangles on a grid of eighths around the origin: a good share of the coordinates is exactly zero,
    # with a random sign
    base = rng.randint(-8, 9, size=(n, 1, 3)).astype(np.float32) / 8
    v = (base + rng.randint(-1, 2, size=(n, 3, 3)).astype(np.float32) / 16 + rng.uniform(0, 1e-3, size=(n, 1, 3)).astype(np.float32) * (rng.rand(n, 1, 3) < 0.5)).reshape(-1, 3)
    v = v.astype(np.float32)
    zero = v == 0
    assert zero.mean() > 0.01
    v[zero & (rng.rand(*v.shape) < 0.5)] = -0.0
    assert (np.signbit(v) & (v == 0)).any()
    acc, _, _, _ = check_against_oracle(v, soup_tris(v))
    acc.close()


def test_degenerate_triangles_and_infinities():
    v, t, _ = scenes.random_soup(5000, 0, s=0.05, seed=21)
    v = v.copy()
    v[30:33] = v[30]                          # a point
    v[60] = v[61]                             # a needle
    v[90:93, 2] = 0.125                       # axis-aligned, fine; then collinear:
    v[120:123] = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [0.3, 0.3, 0.3]], np.float32)
    v[150, 1] = np.inf
    v[180, 0] = -np.inf
    acc, _, _, projections = check_against_oracle(v, t)
    assert projections >= 3
    acc.close()


def test_nan_coordinates_go_to_the_host_builder():
    v, t, _ = scenes.random_soup(5000, 0, s=0.05, seed=3)
    v = v.copy()
    v[7] = np.nan
    acc, _, _, _ = check_against_oracle(v, t, expect_device=False)
    assert acc.wide_built_on_device[0]
    acc.close()


def test_small_scenes_and_the_switch_keep_the_host_builder(monkeypatch):
    v, t, _ = scenes.random_soup(1000, 0, s=0.1, seed=2)
    acc, _, _, _ = check_against_oracle(v, t, expect_device=False)
    acc.close()
    monkeypatch.setenv("YART_BUILD_DEVICE_REF", "0")
    v, t, _ = scenes.random_soup(5000, 0, s=0.1, seed=2)
    acc, _, _, _ = check_against_oracle(v, t, expect_device=False)
    assert acc.wide_built_on_device[0]
    acc.close()


def test_bad_vertex_index_is_rejected():
    v, t, s = scenes.random_soup(5000, 0, s=0.1, seed=2)
    t = t.copy()
    t[4000, 1] = len(v)
    with pytest.raises(api.YartError, match="vertex index out of range"):
        api.Accel(v, t, s)
    t[4000, 1] = -1
    with pytest.raises(api.YartError, match="vertex index out of range"):
        api.Accel(v, t, s)


def test_surface_records_need_the_leaf_order_from_the_device(monkeypatch):
    v, t, s = scenes.random_soup(20000, 100, s=0.05, seed=8)
    rng = np.random.RandomState(1)
    normals = rng.randn(len(v), 3).astype(np.float32)
    rays = scenes.random_rays(50000, seed=2)
    out = []
    for device in ("1", "0"):
        monkeypatch.setenv("YART_BUILD_DEVICE", device)
        acc = api.Accel(v, t, s)
        assert acc.ref_built_on_device[0] == (device == "1")
        acc.set_surface_data(v, t, normals, s)
        hits = acc.intersect_closest(dev(rays), api.YART_REFERENCE_RULE)
        out.append((hits.cpu().numpy().copy(), acc.surface(dev(rays), hits).cpu().numpy().copy(), acc.depth, acc.num_nodes))
        acc.close()
    assert out[0][0].tobytes() == out[1][0].tobytes()
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert out[0][2:] == out[1][2:]


def test_sphere_tree_of_the_reference_tree_path_built_on_the_device(monkeypatch):
    """From 1024 spheres the sphere tree the reference-tree kernels walk comes from the same device builder; the
    reference rule over it must be bit-exact against the oracle (which loops over the spheres like the reference) and
    identical to what the host-built accel returns, ties between duplicate spheres included."""
    from oracle import oracle as orc
    v, t, s = scenes.random_soup(6000, 5000, s=0.04, seed=31)
    s = np.concatenate([s, s[:200], np.array([[0.5, 0.5, 0.5, 0.0]] * 3, np.float32)])      # duplicates, zero radius
    rays = scenes.random_rays(1 << 16, seed=6)
    shadow = rays.copy()
    shadow[:, 7] = 0.4
    out = []
    for device in ("1", "0"):
        monkeypatch.setenv("YART_BUILD_DEVICE", device)
        acc = api.Accel(v, t, s)
        out.append((acc.intersect_closest(dev(rays), api.YART_REFERENCE_RULE).cpu().numpy().copy(),
                    acc.intersect_any(dev(shadow), api.YART_REFERENCE_RULE).cpu().numpy().copy(),
                    acc.intersect_closest(dev(rays), api.YART_REFERENCE_TREE).cpu().numpy().copy()))
        acc.close()
    for k in range(3):
        assert out[0][k].tobytes() == out[1][k].tobytes()
    o = orc.Oracle(v, t, s)
    want = o.intersect(rays, mode=0, nthreads=16)
    got = api.hits_to_numpy(torch.from_numpy(out[0][0]))
    assert_exact(got, want["prim"], want["t"], want["u"], want["v"])
    assert np.array_equal(out[0][1], o.intersect_shadow(shadow, mode=0, nthreads=16))
