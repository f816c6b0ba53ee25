"""SURVEY.md 8a row L + BASELINE configs[0] (plumbing): the reference's OWN renderer drives the
engine as a drop-in.  oracle/_ref/libyart_ref.so holds the real trace() (src/renderer.cpp:29-136:
4096 spp, <= 10 bounces, Rng(y*width+x), all of yart's BSDFs / lights / sampling); an EngineShape
plugged into the reference's virtual Shape::intersect hook (include/shape.h:15, called by
Scene::intersect / intersectShadow at include/scene.h:142-144, :176-180) answers every ray of it
through the C-ABI, one ray per call (yart_intersect_closest / yart_intersect_any).

Bar: image RMSE over the sampled pixels of the 512x512 Cornell frame against the golden pixels
the unmodified reference produced (tests/golden/g4_trace.npz) <= 1e-4 absolute -- and in fact 0:
with identical hits the reference's estimator consumes identical random numbers.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc
from yart_amd import api

RMSE_BOUND = 1e-4

needs_ref = pytest.mark.skipif(not os.path.exists(orc.REF_SO), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
def test_reference_trace_matches_golden_pixels_and_stream():
    """CPU: the fixture is what the reference build in this tree produces (guards a stale _ref)."""
    g = load_golden("g4_trace.npz")
    w, h = (int(v) for v in g["frame"])
    rgb, closest, shadow = orc.reference_trace_cornell(w, h, g["pixels"][:4], mode=1)
    assert np.array_equal(rgb.view(np.uint32), g["rgb"][:4].view(np.uint32))
    assert np.array_equal(closest[0].view(np.uint32), g["rays"][0].view(np.uint32))
    assert np.isinf(closest[:, 7]).all() and np.isfinite(shadow[:, 7]).all()
    # every path starts at the camera (src/rt.cpp:23) and 4096 of them start per pixel
    at_camera = (closest[:, 0:3] == np.array([50.0, 48.0, 220.0], np.float32)).all(1)
    assert int(at_camera.sum()) == 4 * 4096


def _reference_engine(g):
    """The reference itself behind the two C-ABI engine signatures (oracle/ref_shim.cpp): lets the
    EngineShape hook and the batched driver be exercised without a GPU."""
    ref = orc.Reference(g["verts"], g["tris"], g["spheres"])
    fc = ctypes.cast(ref.lib.ref_engine_closest, ctypes.c_void_p)
    fa = ctypes.cast(ref.lib.ref_engine_any, ctypes.c_void_p)
    return ref, (fc, fa, ref.h, 0)


@needs_ref
@pytest.mark.parametrize("mode", [2, 3], ids=["shape_hook", "wavefront_driver"])
def test_harness_plumbing_on_cpu(mode):
    """CPU: with the reference answering the rays, the one-ray Shape hook (mode 2) and the batched
    driver of yart_amd/cxx/wavefront.hpp with trace() restated as its integrator (mode 3) both give
    trace()'s own pixels bit for bit -- so any difference seen on the GPU is the engine's."""
    g = load_golden("g4_trace.npz")
    w, h = (int(v) for v in g["frame"])
    ref, engine = _reference_engine(g)
    rgb, info = orc.reference_trace_cornell(w, h, g["pixels"], mode=mode, nthreads=1, engine=engine)
    assert np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32))
    total = int(g["stream_counts"].sum())
    assert (info if mode == 2 else info["rays"]) == total
    if mode == 3:
        assert info["largest_batch"] == len(g["pixels"])       # every pixel in flight at once


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["reference", "default"])
def test_wavefront_driver_reproduces_reference_image(rule):
    """the batched driver (yart_amd/cxx/wavefront.hpp) feeding the GPU engine, with the
    reference's trace() restated as its integrator (reference BSDFs / lights / RNG): bit-identical
    pixels, RMSE 0."""
    g = load_golden("g4_trace.npz")
    w, h = (int(v) for v in g["frame"])
    acc = api.Accel(g["verts"], g["tris"], g["spheres"], device=0)
    L = api.lib()
    fc = ctypes.cast(L.yart_intersect_closest, ctypes.c_void_p)
    fa = ctypes.cast(L.yart_intersect_any, ctypes.c_void_p)
    flags = api.YART_REFERENCE_RULE if rule == "reference" else 0
    rgb, info = orc.reference_trace_cornell(w, h, g["pixels"], mode=3, nthreads=4, engine=(fc, fa, acc._h, flags))
    print("wavefront driver: %d rays in %d closest-hit batches (largest %d)" % (info["rays"], info["closest_batches"], info["largest_batch"]))
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - g["rgb"].astype(np.float64)) ** 2)))
    assert rmse <= RMSE_BOUND, rmse
    assert np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32))
    assert info["rays"] == int(g["stream_counts"].sum())
    acc.close()


@needs_ref
@pytest.mark.gpu
def test_wavefront_driver_over_pinned_queue_reproduces_reference_image():
    """The same driver with its batches going through a YartQueue (device arrays mirrored in pinned host
    memory, yart_queue_*): bit-identical pixels again."""
    g = load_golden("g4_trace.npz")
    w, h = (int(v) for v in g["frame"])
    acc = api.Accel(g["verts"], g["tris"], g["spheres"], device=0)
    L = api.lib()
    fc = ctypes.cast(L.yart_intersect_closest, ctypes.c_void_p)
    fa = ctypes.cast(L.yart_intersect_any, ctypes.c_void_p)
    qapi = [ctypes.cast(getattr(L, "yart_queue_" + n), ctypes.c_void_p).value
            for n in ("create", "destroy", "rays", "hits", "surfaces", "occluded", "closest", "any")]
    rgb, info = orc.reference_trace_cornell(w, h, g["pixels"], mode=4, nthreads=4, engine=(fc, fa, acc._h, 0), queue_api=qapi)
    assert np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32))
    assert info["rays"] == int(g["stream_counts"].sum())
    acc.close()


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["reference", "default"])
def test_reference_renderer_drives_engine(rule):
    g = load_golden("g4_trace.npz")
    w, h = (int(v) for v in g["frame"])
    acc = api.Accel(g["verts"], g["tris"], g["spheres"], device=0)
    L = api.lib()
    fc = ctypes.cast(L.yart_intersect_closest, ctypes.c_void_p)
    fa = ctypes.cast(L.yart_intersect_any, ctypes.c_void_p)
    flags = api.YART_REFERENCE_RULE if rule == "reference" else 0
    pixels = g["pixels"][:16]
    rgb, calls = orc.reference_trace_cornell(w, h, pixels, mode=2, nthreads=8, engine=(fc, fa, acc._h, flags))
    want = g["rgb"][:16]
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - want.astype(np.float64)) ** 2)))
    print("trace() through the engine: %d one-ray calls, RMSE %.3g" % (calls, rmse))
    assert calls > 16 * 4096
    assert rmse <= RMSE_BOUND, rmse
    if rule == "reference":
        assert np.array_equal(rgb.view(np.uint32), want.view(np.uint32))
    acc.close()
