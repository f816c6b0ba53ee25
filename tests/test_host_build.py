"""CPU: the product's host-side builder (yart_amd/csrc/bvh_build.cpp, reached through the
C-ABI's yart_build_host) must produce the reference's node and TriAccel arrays bit for
bit, and the C-ABI library must load and export every symbol include/yart_amd.h declares.
No compute entry point is called here (there is no GPU in the CPU suite)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from yart_amd import api, scenes


def test_library_exports_every_declared_symbol():
    L = api.lib()
    header = open(os.path.join(ROOT, "include", "yart_amd.h")).read()
    # every declaration: a line that starts with a return type (any C type, pointers included) and names a yart_ function
    declared = set(re.findall(r"^(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s+\*?(yart_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared, "no declarations found"
    assert declared == set(api.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_record_sizes():
    assert api.NODE_DTYPE.itemsize == 32
    assert api.TRIACCEL_DTYPE.itemsize == 48
    assert api.HIT_DTYPE.itemsize == 16


def test_builder_matches_reference_golden(golden):
    nodes, ta = api.build_host(golden["verts"], golden["tris"])
    assert nodes.tobytes() == golden["nodes"].tobytes()
    assert ta.tobytes() == golden["triaccel"].tobytes()


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 63, 64, 1000, 70001])
def test_builder_matches_oracle_on_ragged_sizes(n):
    """70001 is above the size at which the builder hands ranges to its thread pool."""
    from oracle import oracle as orc
    v, t, s = scenes.random_soup(n, 0, s=0.1, seed=n + 1, tris_per_mesh=5)
    nodes, ta = api.build_host(v, t)
    o = orc.Oracle(v, t, s)
    assert nodes.tobytes() == o.nodes().tobytes()
    assert ta.tobytes() == o.triaccel().tobytes()


def test_builder_median_fallback_and_ties():
    """All centres equal on the split axis -> std::partition leaves one side empty ->
    the std::nth_element median path (src/bvhaccel.cpp:134-143)."""
    from oracle import oracle as orc
    n = 40
    v = np.zeros((n * 3, 3), np.float32)
    for i in range(n):
        v[i * 3:(i + 1) * 3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]       # 40 identical triangles
    t = np.zeros((n, 5), np.int32)
    t[:, 0] = np.arange(n) * 3
    t[:, 1] = t[:, 0] + 1
    t[:, 2] = t[:, 0] + 2
    t[:, 3] = np.arange(n)
    nodes, ta = api.build_host(v, t)
    o = orc.Oracle(v, t, np.zeros((0, 4), np.float32))
    assert nodes.tobytes() == o.nodes().tobytes()
    assert ta.tobytes() == o.triaccel().tobytes()


def test_bad_vertex_index_is_rejected():
    v = np.zeros((3, 3), np.float32)
    t = np.array([[0, 1, 5, 0, 0]], np.int32)
    with pytest.raises(api.YartError):
        api.build_host(v, t)


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product must fail, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    v, t, s = scenes.random_soup(16)
    with pytest.raises(api.YartError):
        api.Accel(v, t, s)


def test_size_limits_are_errors_not_crashes():
    """Counts beyond what the records can address are refused before any array is touched: 2^29 leaf
    slots of the reference layout, 2^30 primitives of one kind for the 4-wide tree."""
    L = api.lib()
    v = np.zeros((3, 3), np.float32)
    t = np.zeros((1, 5), np.int32)
    n = ctypes.c_size_t(0)
    vp = ctypes.c_void_p
    assert L.yart_build_host(v.ctypes.data_as(vp), 3, t.ctypes.data_as(vp), 1 << 31, None, ctypes.byref(n), None) == 1
    assert b"too many triangles" in L.yart_last_error()
    nn, npr, depth = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_int(0)
    sph = np.zeros((1, 4), np.float32)
    assert L.yart_build_wide_host(None, 0, None, 0, sph.ctypes.data_as(vp), 1 << 30, 1, None, ctypes.byref(nn),
                                  None, None, ctypes.byref(npr), ctypes.byref(depth)) == 1
    assert b"2^30" in L.yart_last_error()


def test_camera_basis_is_the_reference_camera():
    """yart_camera_basis (Camera::Camera, include/camera.h:9-22, after rt.cpp's normal(dir)) against the oracle's fp32
    restatement, bit for bit; the product no longer borrows the checker's."""
    from oracle import diffuse_ref as dr
    for pos, d, w, h, fov in (([50.0, 48.0, 220.0], [0.0, -0.042612, -1.0], 512, 512, 0.785398),
                              ([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], 1920, 1080, 0.785398),
                              ([0.3, -2.0, 7.5], [0.2, 0.7, -0.4], 333, 77, 1.1)):
        got = api.camera_basis(pos, d, w, h, fov)
        want = dr.camera_basis(pos, d, w, h, fov)
        for a, b in zip(got, want):
            assert np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
    L = api.lib()
    assert L.yart_camera_basis(None, None, 4, 4, ctypes.c_float(1.0), None) == 1
