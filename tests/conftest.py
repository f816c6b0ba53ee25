import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library sorts a batch only from ~1.5 M rays up (YART_SORT_RAYS is a hint); the tests' batches are
# smaller and must still go through the sort path, so the threshold is lifted for every accel they create.
os.environ.setdefault("YART_SORT_MIN_RAYS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden vectors generated from the real reference by oracle/make_golden.py."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session", params=["g1_cornell.npz", "g2_soup.npz", "g3_edges.npz", "g4_trace.npz"])
def golden(request):
    d = load_golden(request.param)
    d["name"] = request.param
    return d


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def wide_walk(verts, tris, spheres, rays, any_hit=False, built=None, counts=False):
    """The default rule without a GPU, end to end: the oracle's walk of the engine's 4-wide tree (host-built unless
    `built` = (nodes, rows_a, rows_bc) of another builder is given) with everything the kernel uses -- the reference's
    nodes for the near-edge check and the tie rule, the scene's cover -- and, for the rays that walk leaves unresolved,
    the pruned walk of the reference's own tree (oracle mode 1), which is what the fix-up launches run.
    -> (results, number of rays that took the second path)"""
    from oracle import oracle as orc
    from yart_amd import api
    if built is None:
        built = api.build_wide_host(verts, tris, spheres)[:3]
    ref_nodes = api.build_host(verts, tris)[0]
    cover = api.scene_cover(verts, tris, spheres)
    rays = np.ascontiguousarray(rays, np.float32)
    got = orc.wide_intersect(built[0], built[1], built[2], rays, any_hit=any_hit, ref_nodes=ref_nodes, cover=cover, counts=counts)
    cnt = None
    if counts:
        got, cnt = got
    left = (got == orc.UNRESOLVED_OCC) if any_hit else (got["prim"] == orc.UNRESOLVED_PRIM)
    if left.any():
        o = orc.Oracle(verts, tris, spheres)
        sub = np.ascontiguousarray(rays[left])
        got[left] = o.intersect_shadow(sub, mode=1, pad_max=cover[7]) if any_hit else o.intersect(sub, mode=1, pad_max=cover[7])
    return (got, int(left.sum()), cnt) if counts else (got, int(left.sum()))
