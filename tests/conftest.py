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
