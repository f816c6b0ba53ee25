"""TEST INFRASTRUCTURE ONLY.  ctypes loaders for

  * ``Oracle``    -- oracle/_build/libyart_oracle.so, the CPU restatement of the
                     reference intersection path (oracle/cpu_ref.cpp);
  * ``Reference`` -- oracle/_ref/libyart_ref.so, the REAL reference compiled from
                     /root/reference by oracle/Makefile (authoring container only;
                     the GPU box gets the prebuilt file).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (yart_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libyart_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libyart_ref.so")
REF_FMA_SO = os.path.join(HERE, "_ref", "libyart_ref_fma.so")

c_sz = ctypes.c_size_t
c_vp = ctypes.c_void_p

NODE_DTYPE = np.dtype([("bounds", np.float32, 6), ("offset", np.uint32),
                       ("numTriangles", np.uint8), ("splitAxis", np.uint8),
                       ("pad", np.uint8, 2)])
TRIACCEL_DTYPE = np.dtype([("n_u", np.float32), ("n_v", np.float32), ("n_d", np.float32), ("k", np.int32),
                           ("b_u", np.float32), ("b_v", np.float32), ("b_d", np.float32), ("triIdx", np.int32),
                           ("c_u", np.float32), ("c_v", np.float32), ("c_d", np.float32), ("meshIdx", np.int32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.int32)])
REFHIT_DTYPE = np.dtype([("hit", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32),
                         ("prim", np.int32), ("ambiguous", np.int32),
                         ("normal", np.float32, 3), ("shadingNormal", np.float32, 3)])
assert NODE_DTYPE.itemsize == 32 and TRIACCEL_DTYPE.itemsize == 48
assert HIT_DTYPE.itemsize == 16 and REFHIT_DTYPE.itemsize == 48


def _p(a):
    return a.ctypes.data_as(c_vp) if a is not None else None


def build_oracle(force=False):
    """Compile the restatement (gcc only; works here and on the GPU box)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "cpu_ref.cpp")):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_reference():
    """Compile the real reference; only where /root/reference exists."""
    if not os.path.isdir("/root/reference"):
        return None
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return REF_SO


def _scene_args(verts, tris, spheres):
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 5)
    spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)
    return verts, tris, spheres


class Oracle:
    """CPU restatement (cpu_ref.cpp)."""

    def __init__(self, verts, tris, spheres):
        build_oracle()
        self.lib = L = ctypes.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = c_vp
        L.orc_scene_create.argtypes = [c_vp, c_sz, c_vp, c_sz, c_vp, c_sz]
        L.orc_scene_destroy.argtypes = [c_vp]
        L.orc_num_nodes.restype = c_sz
        L.orc_num_nodes.argtypes = [c_vp]
        L.orc_get_nodes.argtypes = [c_vp, c_vp]
        L.orc_get_triaccel.argtypes = [c_vp, c_vp]
        L.orc_get_leaf_order.argtypes = [c_vp, c_vp]
        L.orc_intersect.argtypes = [c_vp, c_vp, c_sz, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        L.orc_intersect_shadow.argtypes = [c_vp, c_vp, c_sz, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        self.verts, self.tris, self.spheres = _scene_args(verts, tris, spheres)
        self.h = L.orc_scene_create(_p(self.verts), len(self.verts), _p(self.tris), len(self.tris),
                                    _p(self.spheres), len(self.spheres))
        if not self.h:
            raise ValueError("oracle: bad scene")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def nodes(self):
        out = np.zeros(self.lib.orc_num_nodes(self.h), NODE_DTYPE)
        self.lib.orc_get_nodes(self.h, _p(out))
        return out

    def triaccel(self):
        out = np.zeros(len(self.tris), TRIACCEL_DTYPE)
        self.lib.orc_get_triaccel(self.h, _p(out))
        return out

    def leaf_order(self):
        out = np.zeros(len(self.tris), np.uint32)
        self.lib.orc_get_leaf_order(self.h, _p(out))
        return out

    def intersect(self, rays, mode=0, counts=False, nthreads=8, pad_max=0.0):
        """pad_max (mode 1): the scene's Cover::padMax, word 7 of yart_scene_cover -- how much later the pruned walk culls."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DTYPE)
        cnt = np.zeros((len(rays), 4), np.uint32) if counts else None
        self.lib.orc_intersect(self.h, _p(rays), len(rays), _p(hits), _p(cnt), mode, nthreads, float(pad_max))
        return (hits, cnt) if counts else hits

    def intersect_shadow(self, rays, mode=0, counts=False, nthreads=8, pad_max=0.0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        cnt = np.zeros((len(rays), 4), np.uint32) if counts else None
        self.lib.orc_intersect_shadow(self.h, _p(rays), len(rays), _p(occ), _p(cnt), mode, nthreads, float(pad_max))
        return (occ, cnt) if counts else occ


UNRESOLVED_PRIM = -2**31        # a ray the wide kernel leaves to the fix-up launches (yart_amd/csrc/traverse.h)
UNRESOLVED_OCC = 2


def wide_intersect(nodes, rows_a, rows_bc, rays, any_hit=False, counts=False, nthreads=8, ref_nodes=None, cover=None):
    """The engine's 4-wide tree (dump of yart_accel_get_wide) walked on the CPU with the default
    kernel's rule in its deterministic form (cpu_ref.cpp traverseWide).  ref_nodes: the reference's
    flattened nodes (near-edge check, tie rule); cover: the ten words of yart_scene_cover.  Rays the
    kernel leaves to the fix-up launches come back with prim = UNRESOLVED_PRIM / occlusion UNRESOLVED_OCC."""
    build_oracle()
    L = ctypes.CDLL(ORACLE_SO)
    L.orc_wide_intersect.argtypes = [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_sz, c_vp]
    if ref_nodes is not None:
        ref_nodes = np.ascontiguousarray(ref_nodes)
        assert ref_nodes.dtype.itemsize == 32
    if cover is not None:
        cover = np.ascontiguousarray(cover, np.float32)
        assert cover.size == 10
    nodes = np.ascontiguousarray(nodes, np.uint32)
    rows_a = np.ascontiguousarray(rows_a, np.float32)
    rows_bc = np.ascontiguousarray(rows_bc, np.float32)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros(len(rays), np.uint8 if any_hit else HIT_DTYPE)
    cnt = np.zeros((len(rays), 4), np.uint32) if counts else None
    L.orc_wide_intersect(_p(nodes), _p(rows_a), _p(rows_bc), _p(rays), len(rays), _p(out), _p(cnt),
                         1 if any_hit else 0, nthreads, _p(ref_nodes), 0 if ref_nodes is None else len(ref_nodes), _p(cover))
    return (out, cnt) if counts else out


class Reference:
    """The real reference (ref_shim.cpp around /root/reference)."""

    def __init__(self, verts, tris, spheres, gen_normals=True, so=None):
        so = so or REF_SO
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        self.lib = L = ctypes.CDLL(so)
        L.ref_scene_create.restype = c_vp
        L.ref_scene_create.argtypes = [c_vp, c_sz, c_vp, c_sz, c_vp, c_sz, ctypes.c_int]
        L.ref_scene_destroy.argtypes = [c_vp]
        L.ref_num_nodes.restype = c_sz
        L.ref_num_nodes.argtypes = [c_vp]
        L.ref_get_nodes.argtypes = [c_vp, c_vp]
        L.ref_get_triaccel.argtypes = [c_vp, c_vp]
        L.ref_intersect.argtypes = [c_vp, c_vp, c_sz, c_vp, ctypes.c_int]
        L.ref_intersect_shadow.argtypes = [c_vp, c_vp, c_sz, c_vp, ctypes.c_int]
        L.ref_intersect_throughput.argtypes = [c_vp, c_vp, c_sz, c_vp, ctypes.c_int]
        self.verts, self.tris, self.spheres = _scene_args(verts, tris, spheres)
        self.h = L.ref_scene_create(_p(self.verts), len(self.verts), _p(self.tris), len(self.tris),
                                    _p(self.spheres), len(self.spheres), 1 if gen_normals else 0)
        if not self.h:
            raise ValueError("reference: bad scene")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_scene_destroy(self.h)
            self.h = None

    def nodes(self):
        out = np.zeros(self.lib.ref_num_nodes(self.h), NODE_DTYPE)
        self.lib.ref_get_nodes(self.h, _p(out))
        return out

    def triaccel(self):
        out = np.zeros(len(self.tris), TRIACCEL_DTYPE)
        self.lib.ref_get_triaccel(self.h, _p(out))
        return out

    def vertex_normals(self):
        """The per-vertex normals the reference's TriangleMesh constructor generated."""
        nm = int(self.tris[:, 3].max()) + 1 if len(self.tris) else 0
        base = np.zeros(nm, np.int32)
        for m in range(nm):
            base[m] = self.tris[self.tris[:, 3] == m][:, :3].min()
        out = np.full((len(self.verts), 3), np.nan, np.float32)
        self.lib.ref_get_vertex_normals.argtypes = [c_vp, c_vp, c_vp]
        self.lib.ref_get_vertex_normals(self.h, _p(base), _p(out))
        return out

    def intersect(self, rays, nthreads=8):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), REFHIT_DTYPE)
        self.lib.ref_intersect(self.h, _p(rays), len(rays), _p(out), nthreads)
        return out

    def intersect_shadow(self, rays, nthreads=8):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.uint8)
        self.lib.ref_intersect_shadow(self.h, _p(rays), len(rays), _p(out), nthreads)
        return out

    def throughput(self, rays, nthreads=8):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.float32)
        self.lib.ref_intersect_throughput(self.h, _p(rays), len(rays), _p(out), nthreads)
        return out


def reference_cornell(so=None):
    """Geometry of the reference's Scene::makeCornellBox (src/scene.cpp:4-157)."""
    L = ctypes.CDLL(so or REF_SO)
    nv, nt, ns = c_sz(), c_sz(), c_sz()
    L.ref_cornell(None, ctypes.byref(nv), None, ctypes.byref(nt), None, ctypes.byref(ns))
    verts = np.zeros((nv.value, 3), np.float32)
    tris = np.zeros((nt.value, 5), np.int32)
    sph = np.zeros((ns.value, 4), np.float32)
    L.ref_cornell(_p(verts), ctypes.byref(nv), _p(tris), ctypes.byref(nt), _p(sph), ctypes.byref(ns))
    return verts, tris, sph


def reference_camera_rays(pos, direction, width, height, fov, xy, so=None):
    """Camera::sample (include/camera.h:24-31) of the real reference."""
    L = ctypes.CDLL(so or REF_SO)
    L.ref_camera_rays.argtypes = [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_vp, c_sz, c_vp]
    pos = np.ascontiguousarray(pos, np.float32)
    direction = np.ascontiguousarray(direction, np.float32)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    rays = np.zeros((len(xy), 8), np.float32)
    L.ref_camera_rays(_p(pos), _p(direction), width, height, fov, _p(xy), len(xy), _p(rays))
    return rays


def reference_uniform_sphere(u12, so=None):
    """uniformSphereSample (include/qmc.h:82-94) of the real reference."""
    L = ctypes.CDLL(so or REF_SO)
    L.ref_uniform_sphere.argtypes = [c_vp, c_sz, c_vp]
    u12 = np.ascontiguousarray(u12, np.float32).reshape(-1, 2)
    out = np.zeros((len(u12), 3), np.float32)
    L.ref_uniform_sphere(_p(u12), len(u12), _p(out))
    return out


def reference_load_obj(folder, file, so=None):
    """The reference's Scene::loadFromObj (src/scene.cpp:159-298) flattened; the material table comes
    from its bundled tinyobjloader called directly.  The reference prints the scene to stdout."""
    L = ctypes.CDLL(so or REF_SO)
    L.ref_obj_load.restype = c_vp
    L.ref_obj_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.ref_obj_counts.argtypes = [c_vp] + [ctypes.POINTER(c_sz)] * 5
    L.ref_obj_get.argtypes = [c_vp] * 7
    L.ref_obj_free.argtypes = [c_vp]
    import sys
    sys.stdout.flush()
    L.fflush(None)
    keep, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)                    # the reference lists every material and shape on stdout
    try:
        h = L.ref_obj_load(os.fsencode(folder), os.fsencode(file))
        L.fflush(None)
    finally:
        os.dup2(keep, 1)
        os.close(keep)
        os.close(null)
    n = [c_sz() for _ in range(5)]
    L.ref_obj_counts(h, *[ctypes.byref(x) for x in n])
    nm, nv, nt, ns, nmat = [x.value for x in n]
    out = {"verts": np.zeros((nv, 3), np.float32), "normals": np.zeros((nv, 3), np.float32),
           "tris": np.zeros((nt, 5), np.int32), "spheres": np.zeros((ns, 4), np.float32),
           "meshes": np.zeros((nm, 6), np.int32), "materials": np.zeros((nmat, 9), np.float32)}
    L.ref_obj_get(h, *[out[k].ctypes.data_as(c_vp) for k in ("verts", "normals", "tris", "spheres", "meshes", "materials")])
    L.ref_obj_free(h)
    return out


_ENGINE_CLOSEST = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, c_sz, c_vp, ctypes.c_uint)
_ENGINE_ANY = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, c_sz, c_vp, ctypes.c_uint)


def reference_trace_cornell(width, height, pixels, mode=0, nthreads=8, engine=None, so=None, queue_api=None):
    """The reference's own trace() (src/renderer.cpp:29-136, 4096 spp, <= 10 bounces,
    Rng(y*width+x)) on the listed (x, y) pixels of a width x height frame of
    Scene::makeCornellBox with the camera of src/rt.cpp:22-28.

      mode 0 -> rgb [n,3]                                   (the reference's intersection code)
      mode 1 -> rgb, closest_rays [m,8], shadow_rays [k,8]  (the ray stream trace() issued)
      mode 2 -> rgb, n_engine_calls; `engine` = (closest_fn_addr, any_fn_addr, accel_handle, flags):
                every Scene::intersect / intersectShadow of trace() is answered by those C-ABI
                entry points through the reference's virtual Shape::intersect hook.
      mode 3 -> rgb, stats; trace() restated as the integrator of yart_amd/cxx/wavefront.hpp: all
                listed pixels in flight, batched engine calls.
      mode 4 -> as 3, the batches going through a YartQueue; `queue_api` = addresses of yart_queue_create,
                _destroy, _rays, _hits, _surfaces, _occluded, _closest, _any."""
    L = ctypes.CDLL(so or REF_SO)
    if mode == 4:
        arr = (ctypes.c_void_p * 8)(*queue_api)
        L.ref_set_queue_api.argtypes = [ctypes.c_void_p]
        L.ref_set_queue_api(arr)
    L.ref_trace_cornell.restype = ctypes.c_long
    L.ref_trace_cornell.argtypes = [ctypes.c_int, ctypes.c_int, c_vp, c_sz, ctypes.c_int, ctypes.c_int, c_vp,
                                    ctypes.POINTER(c_vp), ctypes.POINTER(c_sz), ctypes.POINTER(c_vp),
                                    ctypes.POINTER(c_sz), c_vp, c_vp, c_vp, ctypes.c_uint]
    L.ref_free.argtypes = [c_vp]
    pixels = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
    rgb = np.zeros((len(pixels), 3), np.float32)
    cl, sh = c_vp(), c_vp()
    ncl, nsh = c_sz(0), c_sz(0)
    fc, fa, acc, flags = engine if engine is not None else (None, None, None, 0)
    rc = L.ref_trace_cornell(width, height, _p(pixels), len(pixels), mode, nthreads, _p(rgb),
                             ctypes.byref(cl), ctypes.byref(ncl), ctypes.byref(sh), ctypes.byref(nsh),
                             fc, fa, acc, flags)
    if mode == 1:
        closest = np.ctypeslib.as_array(ctypes.cast(cl, ctypes.POINTER(ctypes.c_float)), (ncl.value, 8)).copy()
        shadow = np.ctypeslib.as_array(ctypes.cast(sh, ctypes.POINTER(ctypes.c_float)), (nsh.value, 8)).copy()
        L.ref_free(cl)
        L.ref_free(sh)
        return rgb, closest, shadow
    if mode == 2:
        if rc < 0:
            raise RuntimeError("engine call failed inside trace()")
        return rgb, int(rc)
    if mode in (3, 4):
        if rc < 0:
            raise RuntimeError("engine call failed inside the wavefront driver")
        return rgb, dict(rays=int(rc), closest_batches=int(ncl.value), largest_batch=int(nsh.value))
    return rgb


def reference_trace_diffuse(verts, tris, spheres, albedo, light_pos, light_intensity, cam_pos, cam_dir, fov, width, height,
                            pixels=None, nthreads=8):
    """The real reference's trace() (4096 spp, <= 10 bounces) with its own Lambertian / PointLight on the given scene
    (ref_shim.cpp: ref_trace_diffuse) -> (rgb float32 [n, 3] in pixel-list order, vertex normals the reference's
    TriangleMesh constructor generated, float32 [nVerts, 3])."""
    L = ctypes.CDLL(REF_SO)
    verts, tris, spheres = _scene_args(verts, tris, spheres)
    if pixels is None:
        pixels = np.arange(width * height)
    xy = np.ascontiguousarray(np.stack([np.asarray(pixels) % width, np.asarray(pixels) // width], 1).astype(np.int32))
    f = lambda a: np.ascontiguousarray(a, np.float32)
    rgb = np.zeros((len(xy), 3), np.float32)
    normals = np.zeros((len(verts), 3), np.float32)
    L.ref_trace_diffuse.argtypes = [c_vp, c_sz, c_vp, c_sz, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_int,
                                    ctypes.c_int, c_vp, c_sz, ctypes.c_int, c_vp, c_vp]
    L.ref_trace_diffuse.restype = ctypes.c_int
    a, lp, li, cp, cd = f(albedo), f(light_pos), f(light_intensity), f(cam_pos), f(cam_dir)
    st = L.ref_trace_diffuse(_p(verts), len(verts), _p(tris), len(tris), _p(spheres), len(spheres), _p(a), _p(lp), _p(li),
                             _p(cp), _p(cd), float(fov), int(width), int(height), _p(xy), len(xy), nthreads, _p(rgb), _p(normals))
    if st != 0:
        raise ValueError("ref_trace_diffuse: bad scene")
    return rgb, normals
