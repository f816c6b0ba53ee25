"""GPU: yart_intersect_closest_noalloc_dev (the entry point that allocates nothing) against
yart_intersect_closest_dev on the same inputs -- same kernels, so the bar is bit-identical hit
records -- on two calls in a row, the second with a smaller input; the first is large enough
to go through the coherence sort (sort workspace in the static scratch), the second is not sorted;
the edge-case golden set (irregular rays: fix-up launches, which use two more counter slots of the
scratch) follows.  Size limit and error text are checked without touching the pointers."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden
from yart_amd import api, scenes

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def same_bits(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_noalloc_matches_the_allocating_entry_point(monkeypatch):
    monkeypatch.setenv("YART_SORT_MIN_RAYS", str(1 << 20))
    v, t, s = scenes.random_soup(120000, 5000, s=0.015, seed=31)
    acc = api.Accel(v, t, s)
    big = dev(scenes.random_rays(api.YART_NOALLOC_MAX_RAYS, seed=32))          # the maximum size, sorted
    small = dev(scenes.random_rays(70001, seed=33))                            # ragged, below the sort threshold
    for flags in (api.YART_SORT_RAYS, api.YART_SORT_RAYS | api.YART_REFERENCE_TREE):
        want_big = acc.intersect_closest(big, flags).clone()
        want_small = acc.intersect_closest(small, flags).clone()
        torch.cuda.synchronize()
        got_big = torch.full((big.shape[0], 4), -7.0, dtype=torch.float32, device="cuda")
        got_small = torch.full((small.shape[0], 4), -7.0, dtype=torch.float32, device="cuda")
        acc.intersect_closest(big, flags, out=got_big, noalloc=True)
        acc.intersect_closest(small, flags, out=got_small, noalloc=True)       # second call, smaller input, same scratch
        torch.cuda.synchronize()
        assert same_bits(got_big, want_big)
        assert same_bits(got_small, want_small)
    hits = api.hits_to_numpy(got_big)
    assert 0.05 < (hits["prim"] != -1).mean() < 1.0            # the comparison is not between two empty results
    acc.close()


@pytest.mark.parametrize("name", ["g3_edges.npz", "g1_cornell.npz"])
def test_noalloc_edge_case_rays_and_second_accel(name):
    """Irregular rays go through the fix-up launches (more counter slots of the scratch); a second accel
    on the same GPU shares the array.  (Loads the golden set itself: the session-scoped `golden` fixture
    would let pytest move this test between the other files' tests.)"""
    golden = load_golden(name)
    a = api.Accel(golden["verts"], golden["tris"], golden["spheres"])
    b = api.Accel(*scenes.random_soup(3000, 10, s=0.05, seed=5))
    rays = dev(golden["rays"])
    other = dev(scenes.random_rays(5000, seed=6))
    for acc, r in ((a, rays), (b, other), (a, rays[: len(rays) // 3])):
        want = acc.intersect_closest(r, api.YART_SORT_RAYS).clone()
        got = torch.full((r.shape[0], 4), -7.0, dtype=torch.float32, device="cuda")
        acc.intersect_closest(r, api.YART_SORT_RAYS, out=got, noalloc=True)
        torch.cuda.synchronize()
        assert same_bits(got, want)
    a.close()
    b.close()


def test_noalloc_rejects_batches_beyond_its_maximum():
    acc = api.Accel(*scenes.random_soup(100, 1, s=0.1, seed=1))
    L = api.lib()
    r = torch.zeros((4, 8), dtype=torch.float32, device="cuda")
    out = torch.zeros((4, 4), dtype=torch.float32, device="cuda")
    st = L.yart_intersect_closest_noalloc_dev(acc._h, ctypes.c_void_p(r.data_ptr()), api.YART_NOALLOC_MAX_RAYS + 1,
                                              ctypes.c_void_p(out.data_ptr()), 0, None)
    assert st == 1 and b"YART_NOALLOC_MAX_RAYS" in L.yart_last_error()
    acc.close()
