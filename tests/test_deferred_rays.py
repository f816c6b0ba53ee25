"""GPU: rays the wide-tree kernel does not take (axis-parallel directions, minT != 0, non-finite origins: not
regularRay(), yart_amd/csrc/traverse.h) are answered by fix-up launches on the reference's own tree.  The wide kernel
lists them (up to kDeferCap = 16384 per call) so that those launches touch the listed rays only; a batch with more
is scanned whole.  Both ways must give what the reference-tree path gives for every ray, sorted or not."""
import numpy as np
import pytest

from yart_amd import api, scenes
from test_gpu_parity import dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def irregular(rays, idx, rng):
    kind = rng.randint(0, 4, size=len(idx))
    for i, k in zip(idx, kind):
        if k == 0:
            rays[i, 4 + rng.randint(3)] = 0.0              # axis-parallel: 1/d is infinite
        elif k == 1:
            rays[i, 3] = 0.01                               # minT != 0
        elif k == 2:
            rays[i, 4:7] = 0.0
            rays[i, 4 + rng.randint(3)] = 1.0               # along an axis
        else:
            rays[i, 7] = 0.3                                # regular after all: a finite maxT


@pytest.mark.parametrize("n_irregular", [1, 300, 16384, 16385, 100000])
def test_listed_and_scanned_fixups_agree_with_the_reference_tree(n_irregular, monkeypatch):
    monkeypatch.setenv("YART_SORT_MIN_RAYS", "0")        # so that YART_SORT_RAYS sorts a batch of this size
    v, t, s = scenes.random_soup(30000, 500, s=0.03, seed=5)
    acc = api.Accel(v, t, s)
    n = 1 << 18
    rays = scenes.random_rays(n, seed=9)
    rng = np.random.RandomState(n_irregular)
    idx = rng.choice(n, size=n_irregular, replace=False)
    irregular(rays, idx, rng)
    shadow = rays.copy()
    shadow[:, 7] = np.minimum(shadow[:, 7], 0.5)
    want = acc.intersect_closest(dev(rays), api.YART_REFERENCE_TREE).cpu().numpy()
    want_occ = acc.intersect_any(dev(shadow), api.YART_REFERENCE_TREE).cpu().numpy()
    for flags in (0, api.YART_SORT_RAYS, api.YART_STATIC_LAUNCH):
        got = acc.intersect_closest(dev(rays), flags).cpu().numpy()
        # the irregular rays are answered by the same kernels on the same tree: bit for bit
        assert got[idx].tobytes() == want[idx].tobytes()
        # the others by the wide tree: same primitive and distance (exact-t ties aside, none in this soup)
        assert np.array_equal(got[:, 3].view(np.int32), want[:, 3].view(np.int32))
        assert np.array_equal(got[:, 0], want[:, 0])
        assert np.array_equal(acc.intersect_any(dev(shadow), flags).cpu().numpy(), want_occ)
    acc.close()
