"""ctypes binding of the C-ABI in include/yart_amd.h (libyart_amd.so).

PyTorch is used only as plumbing: device buffers and streams.  There is no CPU
fallback of any kind here -- if the HIP library is missing, or no GPU is present,
the calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YART_AMD_LIB") or os.path.join(_HERE, "libyart_amd.so")      # override: A/B runs of two builds

YART_DEFAULT = 0
YART_REFERENCE_RULE = 1 << 0
YART_STATIC_LAUNCH = 1 << 1
YART_SORT_RAYS = 1 << 2
YART_REFERENCE_TREE = 1 << 3
YART_RAYS_REGULAR = 1 << 4
YART_PRIM_UNRESOLVED = -2**31       # a ray left to the fix-up launches, seen only with YART_RAYS_REGULAR
YART_OCC_UNRESOLVED = 2
YART_NOALLOC_MAX_RAYS = 2097152

NODE_DTYPE = np.dtype([("bounds", np.float32, 6), ("offset", np.uint32),
                       ("numTriangles", np.uint8), ("splitAxis", np.uint8),
                       ("pad", np.uint8, 2)])
TRIACCEL_DTYPE = np.dtype([("n_u", np.float32), ("n_v", np.float32), ("n_d", np.float32), ("k", np.int32),
                           ("b_u", np.float32), ("b_v", np.float32), ("b_d", np.float32), ("triIdx", np.int32),
                           ("c_u", np.float32), ("c_v", np.float32), ("c_d", np.float32), ("meshIdx", np.int32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.int32)])

EXPORTS = [
    "yart_accel_create", "yart_accel_destroy",
    "yart_intersect_closest_dev", "yart_intersect_any_dev", "yart_intersect_closest_noalloc_dev",
    "yart_intersect_closest_counted_dev", "yart_intersect_any_counted_dev",
    "yart_intersect_closest", "yart_intersect_any",
    "yart_accel_num_nodes", "yart_accel_num_triangles", "yart_accel_num_spheres",
    "yart_accel_get_nodes", "yart_accel_get_triaccel", "yart_accel_depth",
    "yart_accel_device_bytes", "yart_build_host", "yart_status_string", "yart_last_error",
    "yart_accel_profile_enable", "yart_accel_profile_read",
    "yart_scene_cover", "yart_camera_basis", "yart_build_wide_host", "yart_accel_num_wide_nodes", "yart_accel_num_wide_prims", "yart_accel_wide_depth", "yart_accel_get_wide", "yart_accel_wide_built_on_device", "yart_accel_ref_built_on_device",
    "yart_obj_load", "yart_obj_free", "yart_obj_num_meshes", "yart_obj_num_vertices", "yart_obj_num_triangles",
    "yart_obj_num_materials", "yart_obj_num_spheres", "yart_obj_get", "yart_obj_mesh_name", "yart_obj_material_name",
    "yart_film_tonemap", "yart_film_write_bmp",
    "yart_accel_set_surface_data", "yart_surface_dev", "yart_surface",
    "yart_wavefront_create", "yart_wavefront_destroy", "yart_wavefront_render_diffuse",
    "yart_tiles_count", "yart_tiles_list", "yart_tiles_rank_pixels",
    "yart_comm_unique_id", "yart_comm_create", "yart_comm_destroy", "yart_frame_gather", "yart_comm_allreduce_u64",
    "yart_queue_create", "yart_queue_destroy", "yart_queue_rays", "yart_queue_hits", "yart_queue_surfaces",
    "yart_queue_occluded", "yart_queue_closest", "yart_queue_any",
]


class DiffuseJob(ctypes.Structure):
    """YartDiffuseJob (include/yart_amd.h)."""
    _fields_ = [("cam_pos", ctypes.c_float * 3), ("cam_dir", ctypes.c_float * 3), ("cam_right", ctypes.c_float * 3),
                ("cam_up", ctypes.c_float * 3), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("spp", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("light_pos", ctypes.c_float * 3),
                ("light_intensity", ctypes.c_float * 3), ("albedo", ctypes.c_float * 3), ("seed", ctypes.c_uint32),
                ("d_pixels", ctypes.c_void_p), ("n_pixels", ctypes.c_size_t), ("regenerate", ctypes.c_int32),
                ("flags", ctypes.c_uint)]


class WavefrontIter(ctypes.Structure):
    _fields_ = [("closest_rays", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64), ("ms_closest", ctypes.c_float),
                ("ms_surface", ctypes.c_float), ("ms_shadow", ctypes.c_float), ("ms_shade", ctypes.c_float)]


class WavefrontStats(ctypes.Structure):
    _fields_ = [("paths", ctypes.c_uint64), ("closest_rays", ctypes.c_uint64), ("shadow_rays", ctypes.c_uint64),
                ("iterations", ctypes.c_uint64), ("ms_total", ctypes.c_double), ("ms_closest", ctypes.c_double),
                ("ms_surface", ctypes.c_double), ("ms_shadow", ctypes.c_double), ("ms_shade", ctypes.c_double)]


SURFACE_DTYPE = np.dtype([("normal", np.float32, 3), ("t", np.float32), ("shading_normal", np.float32, 3),
                          ("u", np.float32), ("v", np.float32), ("prim", np.int32), ("mesh_idx", np.int32),
                          ("tri_idx", np.int32)])

OBJ_MESH_DTYPE = np.dtype([("first_vertex", np.int32), ("num_vertices", np.int32), ("first_triangle", np.int32),
                           ("num_triangles", np.int32), ("material", np.int32), ("normals_from_file", np.int32)])
OBJ_MATERIAL_DTYPE = np.dtype([("diffuse", np.float32, 3), ("specular", np.float32, 3), ("shininess", np.float32),
                               ("ior", np.float32), ("illum", np.int32), ("bsdf", np.int32)])

_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_lib = None


class YartError(RuntimeError):
    pass


def lib():
    """Load libyart_amd.so; raises if it was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YartError("%s not found: build it with `make -C yart_amd/csrc` "
                        "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and this binding hands torch's
    # device buffers and streams to the library, so torch's copy has to be the one both use.  Loading
    # the library first would bring in /opt/rocm's copy as a second runtime, and the one initialised
    # second finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.yart_accel_create.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, ctypes.c_int, ctypes.POINTER(_vp)]
    L.yart_accel_destroy.argtypes = [_vp]
    L.yart_accel_destroy.restype = None
    for name in ("yart_intersect_closest_dev", "yart_intersect_any_dev", "yart_intersect_closest_noalloc_dev"):
        getattr(L, name).argtypes = [_vp, _vp, _sz, _vp, ctypes.c_uint, _vp]
    for name in ("yart_intersect_closest_counted_dev", "yart_intersect_any_counted_dev"):
        getattr(L, name).argtypes = [_vp, _vp, _sz, _vp, _vp, ctypes.c_uint, _vp]
    for name in ("yart_intersect_closest", "yart_intersect_any"):
        getattr(L, name).argtypes = [_vp, _vp, _sz, _vp, ctypes.c_uint]
    for name in ("yart_accel_num_nodes", "yart_accel_num_triangles", "yart_accel_num_spheres",
                 "yart_accel_device_bytes", "yart_accel_num_wide_nodes", "yart_accel_num_wide_prims"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = _sz
    L.yart_accel_depth.argtypes = [_vp]
    L.yart_accel_wide_depth.argtypes = [_vp]
    L.yart_accel_wide_built_on_device.argtypes = [_vp, ctypes.POINTER(ctypes.c_double)]
    L.yart_accel_ref_built_on_device.argtypes = [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.yart_accel_get_wide.argtypes = [_vp, _vp, _vp, _vp]
    L.yart_accel_get_nodes.argtypes = [_vp, _vp]
    L.yart_accel_get_triaccel.argtypes = [_vp, _vp]
    L.yart_build_host.argtypes = [_vp, _sz, _vp, _sz, _vp, ctypes.POINTER(_sz), _vp]
    L.yart_build_wide_host.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, ctypes.c_int, _vp, ctypes.POINTER(_sz), _vp, _vp,
                                       ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_int)]
    L.yart_accel_profile_enable.argtypes = [_vp, ctypes.c_int]
    L.yart_accel_profile_read.argtypes = [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.yart_obj_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_vp)]
    L.yart_obj_free.argtypes = [_vp]
    L.yart_obj_free.restype = None
    for name in ("yart_obj_num_meshes", "yart_obj_num_vertices", "yart_obj_num_triangles", "yart_obj_num_materials",
                 "yart_obj_num_spheres"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = _sz
    L.yart_obj_get.argtypes = [_vp] * 7
    for name in ("yart_obj_mesh_name", "yart_obj_material_name"):
        getattr(L, name).argtypes = [_vp, _sz]
        getattr(L, name).restype = ctypes.c_char_p
    L.yart_accel_set_surface_data.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _sz]
    L.yart_surface_dev.argtypes = [_vp, _vp, _vp, _sz, _vp, _vp]
    L.yart_surface.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.yart_wavefront_create.argtypes = [_vp, _sz, ctypes.POINTER(_vp)]
    L.yart_wavefront_destroy.argtypes = [_vp]
    L.yart_wavefront_destroy.restype = None
    L.yart_wavefront_render_diffuse.argtypes = [_vp, ctypes.POINTER(DiffuseJob), _vp, ctypes.POINTER(WavefrontStats),
                                                ctypes.POINTER(WavefrontIter), _sz, _vp]
    L.yart_tiles_count.argtypes = [ctypes.c_int32] * 3
    L.yart_tiles_count.restype = _sz
    L.yart_tiles_list.argtypes = [ctypes.c_int32] * 3 + [_vp]
    L.yart_tiles_rank_pixels.argtypes = [ctypes.c_int32] * 5 + [_vp]
    L.yart_tiles_rank_pixels.restype = _sz
    L.yart_comm_unique_id.argtypes = [_vp]
    L.yart_comm_create.argtypes = [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.POINTER(_vp)]
    L.yart_comm_destroy.argtypes = [_vp]
    L.yart_comm_destroy.restype = None
    L.yart_frame_gather.argtypes = [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp]
    L.yart_comm_allreduce_u64.argtypes = [_vp, _vp, _sz, _vp]
    L.yart_film_tonemap.argtypes = [_vp, _sz, _vp]
    L.yart_film_write_bmp.argtypes = [ctypes.c_char_p, _vp, ctypes.c_int32, ctypes.c_int32]
    L.yart_status_string.argtypes = [ctypes.c_int]
    L.yart_status_string.restype = ctypes.c_char_p
    L.yart_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def _check(status):
    if status != 0:
        L = lib()
        raise YartError("%s: %s" % (L.yart_status_string(status).decode(), L.yart_last_error().decode()))


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def _scene_arrays(verts, tris, spheres):
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 5)
    if spheres is None:
        spheres = np.zeros((0, 4), np.float32)
    spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)
    return verts, tris, spheres


def build_host(verts, tris):
    """Host-only build (no GPU needed): the reference-layout node and TriAccel arrays."""
    L = lib()
    verts, tris, _ = _scene_arrays(verts, tris, None)
    n = _sz(0)
    _check(L.yart_build_host(_p(verts), len(verts), _p(tris), len(tris), None, ctypes.byref(n), None))
    nodes = np.zeros(n.value, NODE_DTYPE)
    ta = np.zeros(len(tris), TRIACCEL_DTYPE)
    _check(L.yart_build_host(_p(verts), len(verts), _p(tris), len(tris), _p(nodes), ctypes.byref(n), _p(ta)))
    return nodes, ta


def camera_basis(pos, direction, width, height, fov):
    """Camera::Camera (include/camera.h:9-22) -> (pos, dir, right, up) as float32[3] each: what Wavefront.render_diffuse
    takes as `cam` (include/yart_amd.h: yart_camera_basis)."""
    L = lib()
    L.yart_camera_basis.argtypes = [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp]
    L.yart_camera_basis.restype = ctypes.c_int
    p = np.ascontiguousarray(pos, np.float32).reshape(3)
    d = np.ascontiguousarray(direction, np.float32).reshape(3)
    out = np.zeros(12, np.float32)
    _check(L.yart_camera_basis(_p(p), _p(d), int(width), int(height), float(fov), _p(out)))
    return out[0:3].copy(), out[3:6].copy(), out[6:9].copy(), out[9:12].copy()


def kernel_fingerprint():
    """Short hash of the sources that decide what the dominant kernel fetches (the traversal kernel, the tree formats and
    builders, the padding rules).  profiles/traffic.json stamps its PMC figures with it, bench.py refuses a figure whose
    stamp is not this tree's (tools/make_traffic.py)."""
    import hashlib
    h = hashlib.sha1()
    for name in ("traverse_wide.hip", "traverse.hip", "traverse.h", "wide_build.h", "wide_build.cpp", "wide_build_dev.hip", "cover.h", "sort.hip"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def scene_cover(verts, tris, spheres=None):
    """-> float32[10]: the scene's cover (include/yart_amd.h: yart_scene_cover)."""
    L = lib()
    verts, tris, spheres = _scene_arrays(verts, tris, spheres)
    out = np.zeros(10, np.float32)
    L.yart_scene_cover.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz, _vp]
    L.yart_scene_cover.restype = ctypes.c_int
    _check(L.yart_scene_cover(_p(verts), len(verts), _p(tris), len(tris), _p(spheres), len(spheres), _p(out)))
    return out


def build_wide_host(verts, tris, spheres=None, max_leaf=1):
    """Host-only build of the 4-wide tree (no GPU needed) -> nodes uint32 [n,16], rowsA [p,4], rowsBC [p,8], depth."""
    L = lib()
    verts, tris, spheres = _scene_arrays(verts, tris, spheres)
    nn, npr, depth = _sz(0), _sz(0), ctypes.c_int(0)
    args = (_p(verts), len(verts), _p(tris), len(tris), _p(spheres), len(spheres), int(max_leaf))
    _check(L.yart_build_wide_host(*args, None, ctypes.byref(nn), None, None, ctypes.byref(npr), ctypes.byref(depth)))
    nodes = np.zeros((nn.value, 16), np.uint32)
    rows_a = np.zeros((npr.value, 4), np.float32)
    rows_bc = np.zeros((npr.value, 8), np.float32)
    _check(L.yart_build_wide_host(*args, _p(nodes), ctypes.byref(nn), _p(rows_a), _p(rows_bc), ctypes.byref(npr), ctypes.byref(depth)))
    return nodes, rows_a, rows_bc, depth.value


def load_obj(folder, file):
    """OBJ + MTL -> flat scene (yart_obj_load; host-only).  Returns a dict: verts [v,3], normals [v,3],
    tris [t,5], spheres [s,4], meshes (OBJ_MESH_DTYPE), materials (OBJ_MATERIAL_DTYPE), mesh_names,
    material_names.  verts / tris / spheres go to Accel() unchanged."""
    L = lib()
    h = _vp()
    _check(L.yart_obj_load(os.fsencode(folder), os.fsencode(file), ctypes.byref(h)))
    try:
        out = {
            "verts": np.zeros((L.yart_obj_num_vertices(h), 3), np.float32),
            "normals": np.zeros((L.yart_obj_num_vertices(h), 3), np.float32),
            "tris": np.zeros((L.yart_obj_num_triangles(h), 5), np.int32),
            "spheres": np.zeros((L.yart_obj_num_spheres(h), 4), np.float32),
            "meshes": np.zeros(L.yart_obj_num_meshes(h), OBJ_MESH_DTYPE),
            "materials": np.zeros(L.yart_obj_num_materials(h), OBJ_MATERIAL_DTYPE),
        }
        _check(L.yart_obj_get(h, *[_p(out[k]) for k in ("verts", "normals", "tris", "spheres", "meshes", "materials")]))
        out["mesh_names"] = [L.yart_obj_mesh_name(h, i).decode() for i in range(len(out["meshes"]))]
        out["material_names"] = [L.yart_obj_material_name(h, i).decode() for i in range(len(out["materials"]))]
        return out
    finally:
        L.yart_obj_free(h)


def film_tonemap(rgb):
    """Float r,g,b -> the reference's 8-bit b,g,r (yart_film_tonemap; host-only)."""
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((len(rgb), 3), np.uint8)
    _check(lib().yart_film_tonemap(_p(rgb), len(rgb), _p(out)))
    return out


def film_write_bmp(path, rgb, width, height):
    """Float framebuffer [height, width, 3] (row 0 on top) -> 24-bit BMP (yart_film_write_bmp)."""
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(height, width, 3)
    _check(lib().yart_film_write_bmp(os.fsencode(path), _p(rgb), width, height))


class Accel:
    """A scene resident in one GPU's HBM (yart_accel_create)."""

    def __init__(self, verts, tris, spheres=None, device=0):
        self._h = None
        L = lib()
        verts, tris, spheres = _scene_arrays(verts, tris, spheres)
        h = _vp()
        _check(L.yart_accel_create(_p(verts), len(verts), _p(tris), len(tris), _p(spheres), len(spheres),
                                   int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if self._h:
            lib().yart_accel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection ------------------------------------------------------
    @property
    def num_nodes(self):
        return lib().yart_accel_num_nodes(self._h)

    @property
    def num_triangles(self):
        return lib().yart_accel_num_triangles(self._h)

    @property
    def num_spheres(self):
        return lib().yart_accel_num_spheres(self._h)

    @property
    def depth(self):
        return lib().yart_accel_depth(self._h)

    @property
    def device_bytes(self):
        return lib().yart_accel_device_bytes(self._h)

    def nodes(self):
        out = np.zeros(self.num_nodes, NODE_DTYPE)
        if out.size:
            _check(lib().yart_accel_get_nodes(self._h, _p(out)))
        return out

    def triaccel(self):
        out = np.zeros(self.num_triangles, TRIACCEL_DTYPE)
        if out.size:
            _check(lib().yart_accel_get_triaccel(self._h, _p(out)))
        return out

    @property
    def wide_depth(self):
        return lib().yart_accel_wide_depth(self._h)

    @property
    def wide_built_on_device(self):
        """-> (bool, milliseconds) : whether the GPU built the 4-wide tree (YART_BUILD_DEVICE=1) and how long it took."""
        ms = ctypes.c_double()
        return bool(lib().yart_accel_wide_built_on_device(self._h, ctypes.byref(ms))), ms.value

    @property
    def ref_built_on_device(self):
        """-> (bool, milliseconds, ranges split at the median on the host, degenerate triangles projected on the host):
        whether the GPU built the reference's own tree and TriAccel records too (YART_BUILD_DEVICE=1)."""
        ms, med, prj = ctypes.c_double(), ctypes.c_size_t(), ctypes.c_size_t()
        on = lib().yart_accel_ref_built_on_device(self._h, ctypes.byref(ms), ctypes.byref(med), ctypes.byref(prj))
        return bool(on), ms.value, med.value, prj.value

    def wide(self):
        """-> (nodes uint32 [nNodes, 16], rowsA float32 [nPrims, 4], rowsBC float32 [nPrims, 8]) of the
        4-wide tree (yart_amd/csrc/wide_build.h)."""
        L = lib()
        n = L.yart_accel_num_wide_prims(self._h)
        nodes = np.zeros((L.yart_accel_num_wide_nodes(self._h), 16), np.uint32)
        rows_a = np.zeros((n, 4), np.float32)
        rows_bc = np.zeros((n, 8), np.float32)
        _check(L.yart_accel_get_wide(self._h, _p(nodes), _p(rows_a), _p(rows_bc)))
        return nodes, rows_a, rows_bc

    def profile(self, enable=True):
        _check(lib().yart_accel_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self):
        """-> dict(ms_sort, ms_spheres, ms_triangles, calls): HIP-event time of each kernel
        group since the last read, measured on the stream the kernels were launched on."""
        a, b, c, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
        _check(lib().yart_accel_profile_read(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c),
                                             ctypes.byref(n)))
        return dict(ms_sort=a.value, ms_spheres=b.value, ms_triangles=c.value, calls=n.value)

    # -- device-resident batches (torch tensors on this accel's GPU) --------
    def _stream(self, stream):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        return _vp(stream.cuda_stream)

    def _rays(self, rays):
        import torch
        if not (isinstance(rays, torch.Tensor) and rays.is_cuda and rays.dtype == torch.float32
                and rays.is_contiguous() and rays.dim() == 2 and rays.shape[1] == 8):
            raise YartError("rays must be a contiguous float32 CUDA tensor of shape [n, 8]")
        if rays.device.index != self.device:
            raise YartError("rays live on another device than the accel")
        return rays

    def intersect_closest(self, rays, flags=0, out=None, counts=None, stream=None, noalloc=False):
        """rays [n,8] float32 -> hits [n,4] float32 (t,u,v,prim-as-int32-bits).
        noalloc: through yart_intersect_closest_noalloc_dev (`out` must be given)."""
        import torch
        rays = self._rays(rays)
        n = rays.shape[0]

        def check_out(t, cols, dtype, what):
            if not (t.is_cuda and t.device == rays.device and t.is_contiguous() and t.dtype == dtype and tuple(t.shape) == (n, cols)):
                raise YartError("%s must be a contiguous [%d, %d] %s tensor on the rays' device" % (what, n, cols, dtype))
        if out is not None:
            check_out(out, 4, torch.float32, "out")
        if counts is not None:
            check_out(counts, 4, torch.int32, "counts")
        if noalloc:
            if out is None or counts is not None:
                raise YartError("noalloc: pass `out`, no `counts`")
            _check(lib().yart_intersect_closest_noalloc_dev(self._h, _vp(rays.data_ptr()), n, _vp(out.data_ptr()),
                                                            flags, self._stream(stream)))
            return out
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=rays.device)
        if counts is not None:
            _check(lib().yart_intersect_closest_counted_dev(self._h, _vp(rays.data_ptr()), n, _vp(out.data_ptr()),
                                                            _vp(counts.data_ptr()), flags, self._stream(stream)))
        else:
            _check(lib().yart_intersect_closest_dev(self._h, _vp(rays.data_ptr()), n, _vp(out.data_ptr()),
                                                    flags, self._stream(stream)))
        return out

    def intersect_any(self, rays, flags=0, out=None, counts=None, stream=None):
        """rays [n,8] float32 -> occluded [n] uint8."""
        import torch
        rays = self._rays(rays)
        n = rays.shape[0]
        if out is None:
            out = torch.empty((n,), dtype=torch.uint8, device=rays.device)
        if counts is not None:
            _check(lib().yart_intersect_any_counted_dev(self._h, _vp(rays.data_ptr()), n, _vp(out.data_ptr()),
                                                        _vp(counts.data_ptr()), flags, self._stream(stream)))
        else:
            _check(lib().yart_intersect_any_dev(self._h, _vp(rays.data_ptr()), n, _vp(out.data_ptr()),
                                                flags, self._stream(stream)))
        return out

    # -- host buffers (numpy) -----------------------------------------------
    def set_surface_data(self, verts, tris, vertex_normals, spheres=None):
        """Upload what hit post-processing needs (yart_accel_set_surface_data)."""
        verts, tris, spheres = _scene_arrays(verts, tris, spheres)
        normals = np.ascontiguousarray(vertex_normals, np.float32).reshape(-1, 3)
        if len(normals) != len(verts):
            raise YartError("vertex_normals must have one row per vertex")
        _check(lib().yart_accel_set_surface_data(self._h, _p(verts), len(verts), _p(tris), len(tris), _p(normals),
                                                 _p(spheres), len(spheres)))

    def surface(self, rays, hits, out=None, stream=None):
        """rays [n,8], hits [n,4] (device) -> [n,12] float32 device tensor of YartSurface records."""
        import torch
        rays = self._rays(rays)
        n = rays.shape[0]
        if not (isinstance(hits, torch.Tensor) and hits.is_cuda and hits.is_contiguous() and hits.shape == (n, 4)):
            raise YartError("hits must be the [n, 4] device tensor of an intersect_closest call")
        if out is None:
            out = torch.empty((n, 12), dtype=torch.float32, device=rays.device)
        _check(lib().yart_surface_dev(self._h, _vp(rays.data_ptr()), _vp(hits.data_ptr()), n, _vp(out.data_ptr()),
                                      self._stream(stream)))
        return out

    def surface_host(self, rays, hits):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.ascontiguousarray(hits)
        out = np.zeros(len(rays), SURFACE_DTYPE)
        _check(lib().yart_surface(self._h, _p(rays), _p(hits), len(rays), _p(out)))
        return out

    def intersect_closest_host(self, rays, flags=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _check(lib().yart_intersect_closest(self._h, _p(rays), len(rays), _p(hits), flags))
        return hits

    def intersect_any_host(self, rays, flags=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        _check(lib().yart_intersect_any(self._h, _p(rays), len(rays), _p(occ), flags))
        return occ


def tiles_list(width, height, tile=32):
    """Renderer::render's tile list (yart_tiles_list) -> int32 [n, 4] = x0, y0, x1, y1."""
    L = lib()
    out = np.zeros((L.yart_tiles_count(width, height, tile), 4), np.int32)
    if len(out):
        _check(L.yart_tiles_list(width, height, tile, _p(out)))
    return out


def tiles_rank_pixels(width, height, rank, world, tile=32):
    """Pixel numbers of a rank's tiles in rendering order (yart_tiles_rank_pixels) -> uint32 [n]."""
    L = lib()
    n = L.yart_tiles_rank_pixels(width, height, tile, rank, world, None)
    out = np.zeros(n, np.uint32)
    if n:
        got = L.yart_tiles_rank_pixels(width, height, tile, rank, world, _p(out))
        if got != n:
            raise YartError("yart_tiles_rank_pixels: inconsistent count")
    return out


class Comm:
    """RCCL communicator for the framebuffer gather (yart_comm_*).  unique_id() on rank 0, shared by the
    caller (torch.distributed.broadcast_object_list, a file, ...), then Comm(id, rank, world, device)."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check(lib().yart_comm_unique_id(buf))
        return bytes(buf.raw)

    def __init__(self, unique_id, rank, world, device=0):
        self._h = None
        h = _vp()
        _check(lib().yart_comm_create(ctypes.c_char_p(unique_id), rank, world, int(device), ctypes.byref(h)))
        self._h = h
        self.rank, self.world, self.device = rank, world, int(device)

    def close(self):
        if self._h:
            lib().yart_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_frame(self, local_rgb, width, height, tile=32, stream=None):
        """local_rgb: float32 CUDA tensor [n_local, 3] in tiles_rank_pixels order -> [height, width, 3] on rank 0, None elsewhere."""
        import torch
        if not (local_rgb.is_cuda and local_rgb.is_contiguous() and local_rgb.dtype == torch.float32):
            raise YartError("local_rgb must be a contiguous float32 CUDA tensor")
        frame = torch.zeros((height, width, 3), dtype=torch.float32, device=local_rgb.device) if self.rank == 0 else None
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        _check(lib().yart_frame_gather(self._h, _vp(local_rgb.data_ptr()), width, height, tile,
                                       _vp(frame.data_ptr()) if frame is not None else None, _vp(stream.cuda_stream)))
        return frame

    def allreduce_u64(self, values, stream=None):
        """values: contiguous CUDA tensor of n 64-bit integers (torch.int64 holds the uint64 bit patterns), summed over
        the ranks in place (yart_comm_allreduce_u64: ray / hit counters and a checksum)."""
        import torch
        if not (values.is_cuda and values.is_contiguous() and values.dtype in (torch.int64, torch.uint64)):
            raise YartError("values must be a contiguous 64-bit integer CUDA tensor")
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        _check(lib().yart_comm_allreduce_u64(self._h, _vp(values.data_ptr()), values.numel(), _vp(stream.cuda_stream)))
        return values


class Wavefront:
    """Device-resident path queues + the diffuse integrator (yart_wavefront_*)."""

    def __init__(self, accel, slots):
        self._h = None
        h = _vp()
        _check(lib().yart_wavefront_create(accel._h, int(slots), ctypes.byref(h)))
        self._h = h
        self.accel = accel
        self.slots = int(slots)

    def close(self):
        if self._h:
            lib().yart_wavefront_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_diffuse(self, cam, width, height, spp, max_depth, light_pos, light_intensity, albedo, seed=1,
                       pixels=None, regenerate=True, flags=YART_SORT_RAYS, max_iters=4096, stream=None):
        """cam = (pos, dir, right, up) as Camera holds them (oracle/diffuse_ref.camera_basis or the C++ shim's
        Camera).  pixels: optional int32/uint32 CUDA tensor of pixel numbers y*width+x.  -> (rgb [n,3] float32
        CUDA tensor in pixel-list order, stats dict, list of per-iteration dicts)."""
        import torch
        job = DiffuseJob()
        for name, v in (("cam_pos", cam[0]), ("cam_dir", cam[1]), ("cam_right", cam[2]), ("cam_up", cam[3]),
                        ("light_pos", light_pos), ("light_intensity", light_intensity), ("albedo", albedo)):
            setattr(job, name, (ctypes.c_float * 3)(*[float(x) for x in v]))
        job.width, job.height, job.spp, job.max_depth = int(width), int(height), int(spp), int(max_depth)
        job.seed = int(seed)
        job.regenerate = 1 if regenerate else 0
        job.flags = int(flags)
        n = width * height
        if pixels is not None:
            if not (pixels.is_cuda and pixels.is_contiguous() and pixels.dtype in (torch.int32, torch.uint32)):
                raise YartError("pixels must be a contiguous int32 CUDA tensor")
            job.d_pixels = pixels.data_ptr()
            job.n_pixels = pixels.numel()
            n = pixels.numel()
        rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda:%d" % self.accel.device)
        st = WavefrontStats()
        its = (WavefrontIter * max_iters)()
        _check(lib().yart_wavefront_render_diffuse(self._h, ctypes.byref(job), _vp(rgb.data_ptr()), ctypes.byref(st), its,
                                                   max_iters, self.accel._stream(stream)))
        stats = {k: getattr(st, k) for k, _ in WavefrontStats._fields_}
        iters = [{k: getattr(its[i], k) for k, _ in WavefrontIter._fields_} for i in range(min(max_iters, int(st.iterations)))]
        return rgb, stats, iters


def hits_to_numpy(hits_tensor):
    """[n,4] float32 device tensor -> structured numpy array (t,u,v,prim)."""
    a = hits_tensor.detach().cpu().numpy()
    return np.ascontiguousarray(a).view(HIT_DTYPE).reshape(-1)
