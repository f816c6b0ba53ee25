/* yart_amd.h -- C-ABI of the MI355X ray-intersection engine for yart.
 *
 * This is the drop-in boundary for ONE path of elaic/yart: BVH traversal plus
 * Wald-projected-triangle and sphere intersection, closest-hit and any-hit.
 * Every entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  Plain pointers and sizes only;
 * no C++ or torch types.  All functions return a yart_status (0 = success);
 * hit / miss is DATA in the output arrays, never the return value.
 *
 * Threading: an accel is immutable after creation; intersect calls may be
 * issued concurrently from several host threads on distinct streams.
 */
#ifndef YART_AMD_H
#define YART_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- records --------------------------------------------------------- */

/* Ray, 32 bytes: the reference's `struct Ray` (include/vector.h:238-250).
 * dir must be unit length, as a direction normalised in fp32 is (| |dir|^2 - 1 | <= 7 * 2^-24): the reference's sphere
 * test assumes it (include/sphere.h:67-68), and the engine's sphere boxes are sized for what that test accepts from
 * such a direction.  Triangle results do not depend on the length of dir. */
typedef struct YartRay {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;
} YartRay;

/* Hit, 16 bytes: the fields of the reference's RayHitInfo that the hot path
 * produces (t, u, v; include/vector.h:255-263) plus the primitive identity the
 * reference derives its normal / bsdf / areaLight from but never exposes
 * (src/bvhaccel.cpp:283-292, include/sphere.h:79-83).
 *   prim >= 0   triangle: index into the leaf-ordered TriAccel array
 *               (yart_accel_get_triaccel; its triIdx/meshIdx give mesh ids)
 *   prim == -1  miss; t == ray.tmax, u == v == 0
 *   prim <= -2  sphere number -(prim + 2) of the `spheres` array; u == v == 0 */
typedef struct YartHit {
    float t, u, v;
    int32_t prim;
} YartHit;

/* 48-byte Wald triangle record, the reference's `struct TriAccel`
 * (include/triaccel.h:16-32), bit-for-bit. */
typedef struct YartTriAccel {
    float n_u, n_v, n_d; int32_t k;
    float b_u, b_v, b_d; int32_t triIdx;
    float c_u, c_v, c_d; int32_t meshIdx;
} YartTriAccel;

/* 32-byte flattened BVH node, the reference's FlattenedBvhNode
 * (src/bvhaccel.cpp:64-93) in the reference's own depth-first order; the
 * reference leaves numTriangles uninitialised in interior nodes, here it is 0. */
typedef struct YartBvhNode {
    float bounds_min[3];
    float bounds_max[3];
    uint32_t offset;        /* interior: childOffset (right child); leaf: triangleOffset */
    uint8_t num_triangles;
    uint8_t split_axis;     /* 0,1,2 = interior split axis; 3 = leaf */
    uint8_t pad[2];
} YartBvhNode;

typedef struct YartAccel YartAccel;

enum yart_status {
    YART_OK = 0,
    YART_ERR_INVALID_ARGUMENT = 1,
    YART_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime failure at create */
    YART_ERR_HIP = 3,             /* a HIP call failed; see yart_last_error() */
    YART_ERR_OUT_OF_MEMORY = 4,
    YART_ERR_TREE_TOO_DEEP = 5    /* BVH deeper than the 64-entry traversal stack the
                                     reference also assumes (src/bvhaccel.cpp:242-244) */
};

/* flags for the intersect calls */
enum yart_flags {
    YART_DEFAULT        = 0,
    /* Traverse with the reference's exact rule: box interval [minT,maxT], no
     * distance pruning (include/bbox.h:30-50), every sphere considered.  Results
     * are bit-identical to the reference for every ray.  Slow; for parity runs. */
    YART_REFERENCE_RULE = 1u << 0,
    /* Launch one lane per ray over a plain grid instead of the persistent
     * work-pulling kernel (A/B measurements). */
    YART_STATIC_LAUNCH  = 1u << 1,
    /* The batch is incoherent: sort it by a Morton key of origin + direction octant before
     * traversal when that pays -- batches of at least YART_SORT_MIN_RAYS rays (environment, read
     * at create; default 1572864: measured crossover on a scene far larger than L2).  Results are
     * written back in the caller's order either way. */
    YART_SORT_RAYS      = 1u << 2,
    /* Distance-pruned traversal of the REFERENCE's binary tree (sphere-tree pass, then the
     * triangle tree, like include/scene.h:142-146) instead of the engine's own 4-wide
     * triangle+sphere tree.  Same results as the default; slower. */
    YART_REFERENCE_TREE = 1u << 3,
    /* Skip the fix-up launches.  The default kernel leaves three kinds of ray to two small launches
     * on the reference's tree that follow it: rays that are not "regular" (a non-finite origin, a
     * zero or non-finite 1/dir component, minT != 0, NaN maxT -- the reference's Ray constructor
     * and renderer produce none, include/vector.h:244-249), rays that start outside the scene's
     * cover (yart_scene_cover), and rays that meet two triangles at bit-identical t in two
     * different leaves of the reference's tree (which of them the reference keeps depends on its
     * visiting order, src/bvhaccel.cpp:255-281).  With this flag such a ray comes back UNRESOLVED:
     * YartHit.prim == YART_PRIM_UNRESOLVED, occlusion byte == YART_OCC_UNRESOLVED; the caller
     * re-submits those rays with YART_REFERENCE_TREE, or knows that its batch holds none. */
    YART_RAYS_REGULAR   = 1u << 4
};
#define YART_PRIM_UNRESOLVED INT32_MIN
#define YART_OCC_UNRESOLVED 2

/* ---- accel lifetime -------------------------------------------------- */

/* Replaces Scene::preprocess() -> BvhAccel::BvhAccel(const Scene&)
 * (include/scene.h:37-74, src/bvhaccel.cpp:309-363) and the ownership of
 * Scene::shapes_ for the sphere loop (include/scene.h:142-144).
 *
 *   verts    n_verts x 3 floats
 *   tris     n_tris x 5 int32: v0, v1, v2 (indices into verts), meshId, triId,
 *            listed in the order the reference enumerates them: mesh by mesh,
 *            triangle by triangle (src/bvhaccel.cpp:328-343) -- the builder's
 *            partition order, and so the leaf order, depends on it
 *   spheres  n_spheres x 4 floats: cx, cy, cz, radius, in Scene::shapes_ order
 *   device   HIP device ordinal the accel lives on
 *
 * The host builds the reference's tree (same topology, same leaf order, same
 * TriAccel bits) and uploads it once; nothing is read from the input arrays
 * after the call returns. */
int yart_accel_create(const float* verts, size_t n_verts,
                      const int32_t* tris, size_t n_tris,
                      const float* spheres, size_t n_spheres,
                      int device, YartAccel** out);

/* Replaces BvhAccel::~BvhAccel (src/bvhaccel.cpp:365-368). */
void yart_accel_destroy(YartAccel* accel);

/* ---- intersection: rays and results already resident in HBM ----------- */

/* Replaces Scene::intersect(const Ray&, RayHitInfo*) (include/scene.h:136-168)
 * -> Shape::intersect loop + BvhAccel::intersect (src/bvhaccel.cpp:370-373),
 * for a batch of n rays.  d_rays / d_hits are device pointers on the accel's
 * device; `stream` is a hipStream_t (NULL = default stream).  Asynchronous. */
int yart_intersect_closest_dev(YartAccel* accel, const YartRay* d_rays, size_t n,
                               YartHit* d_hits, unsigned flags, void* stream);

/* Replaces Scene::intersectShadow(const Ray&) (include/scene.h:170-194) ->
 * BvhAccel::intersectShadow (src/bvhaccel.cpp:375-381).  d_occluded[i] = 1 if
 * any primitive is accepted in (1e-4, ray.tmax), else 0. */
int yart_intersect_any_dev(YartAccel* accel, const YartRay* d_rays, size_t n,
                           uint8_t* d_occluded, unsigned flags, void* stream);

/* As above and additionally writes, per ray, FOUR uint32 fetch counters -- the terms of the
 * algorithmic bytes per ray.  Diagnostic; slower.
 *   default path (4-wide tree):            YART_REFERENCE_TREE / _RULE (reference's tree):
 *   [0] 16-byte child records fetched      [0] 32-byte node records, triangle tree
 *       (four per 64-byte node)
 *   [1] 16-byte triangle plane rows        [1] 48-byte triangle records tested
 *   [2] 32-byte second-stage rows          [2] 32-byte node records, sphere tree
 *       (primitives that passed stage one)
 *   [3] 16-byte sphere records             [3] 16-byte sphere records
 * Rays the default path defers to the reference tree (see YART_RAYS_REGULAR) get that
 * path's counters. */
int yart_intersect_closest_counted_dev(YartAccel* accel, const YartRay* d_rays, size_t n,
                                       YartHit* d_hits, uint32_t* d_counts,
                                       unsigned flags, void* stream);
int yart_intersect_any_counted_dev(YartAccel* accel, const YartRay* d_rays, size_t n,
                                   uint8_t* d_occluded, uint32_t* d_counts,
                                   unsigned flags, void* stream);

/* ---- intersection without allocation ---------------------------------- */

/* No counterpart in the reference.  For callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same contract, kernels and results as
 * yart_intersect_closest_dev, but the scratch those kernels need beyond the caller's arrays (the
 * persistent kernels' work counters, the coherence sort's keys / permutation / histograms) lives in
 * one static 40 MiB device array per GPU instead of per-stream blocks grown with hipMalloc, so the
 * call reaches no hipMalloc / hipFree / allocating library routine.  Limits: n <=
 * YART_NOALLOC_MAX_RAYS (larger batches return YART_ERR_INVALID_ARGUMENT); every call on one GPU
 * shares that GPU's array, so calls that OVERLAP on one GPU (other streams or threads, any accel)
 * race -- issue them on one stream or order them with events. */
#define YART_NOALLOC_MAX_RAYS 2097152u
int yart_intersect_closest_noalloc_dev(YartAccel* accel, const YartRay* d_rays, size_t n,
                                       YartHit* d_hits, unsigned flags, void* stream);

/* ---- intersection: host buffers (copies in, runs, copies out, syncs) --- */

/* Same contracts as the _dev calls; the batch-of-one form of these is what the
 * C++ shim (yart_amd/cxx/yart_shim.hpp) forwards Scene::intersect /
 * Scene::intersectShadow to. */
int yart_intersect_closest(YartAccel* accel, const YartRay* rays, size_t n,
                           YartHit* hits, unsigned flags);
int yart_intersect_any(YartAccel* accel, const YartRay* rays, size_t n,
                       uint8_t* occluded, unsigned flags);

/* ---- introspection (parity dumps) ------------------------------------- */

/* The reference's private BvhAccel::optimizedAccel_ / triangles_
 * (include/bvhaccel.h:57-58) as built by this library. */
size_t yart_accel_num_nodes(const YartAccel* accel);
size_t yart_accel_num_triangles(const YartAccel* accel);
size_t yart_accel_num_spheres(const YartAccel* accel);
int yart_accel_get_nodes(const YartAccel* accel, YartBvhNode* out);
int yart_accel_get_triaccel(const YartAccel* accel, YartTriAccel* out);
/* The engine's own 4-wide tree (no counterpart in the reference): node count (64 bytes
 * each), primitives in its leaves, depth, and a dump of its arrays (nodes; 16 bytes per primitive
 * of first-stage leaf rows; 32 bytes per primitive of second-stage rows -- layout:
 * yart_amd/csrc/wide_build.h) for the CPU-side cross-check of the traversal counters. */
size_t yart_accel_num_wide_nodes(const YartAccel* accel);
size_t yart_accel_num_wide_prims(const YartAccel* accel);
int yart_accel_wide_depth(const YartAccel* accel);
int yart_accel_get_wide(const YartAccel* accel, void* nodes_out, void* rows_a_out, void* rows_bc_out);
/* 1 if the 4-wide tree of this accel was built on the GPU (Morton-ordered linear BVH whose subtrees of <= 256
 * primitives are split again with binned SAH, collapse plan, quantised nodes and leaf rows by kernels; no
 * counterpart in the reference, whose build is src/bvhaccel.cpp:100-151), 0 if the host builder made it (binned
 * SAH on a thread pool: seconds for 10^7 primitives, a tree the kernel walks 1-2 % faster).  The GPU builds scenes
 * of 2^18 primitives and more unless the environment says otherwise at create (YART_BUILD_DEVICE=1 / 0: always /
 * never).  ms_out (optional): wall milliseconds of the device build. */
int yart_accel_wide_built_on_device(const YartAccel* accel, double* ms_out);
/* 1 if the reference's own arrays of this accel -- the flattened nodes in the reference's order, the TriAccel
 * records in its leaf order: what BvhAccel's constructor builds on one host thread, src/bvhaccel.cpp:100-151,
 * :215-234, :309-363 -- were built on the GPU as well (YART_BUILD_DEVICE=1, triangle count >= 1024, no NaN
 * coordinate; YART_BUILD_DEVICE_REF=0 keeps this build on the host), bit for bit the arrays of the host
 * builder; 0 if the host built them.  Optional outputs: wall milliseconds of that build, the ranges whose
 * one-sided midpoint split was settled by std::nth_element on the host, and the degenerate triangles whose
 * TriAccel record (it holds NaNs) was computed on the host. */
int yart_accel_ref_built_on_device(const YartAccel* accel, double* ms_out, size_t* host_median_splits, size_t* host_projections);
/* Depth of the triangle tree (root = 1) and bytes resident in HBM. */
int yart_accel_depth(const YartAccel* accel);
size_t yart_accel_device_bytes(const YartAccel* accel);

/* Host-only build (no HIP call): fills the same arrays the accel would hold.
 * Pass NULL outputs to query the counts.  Used by CPU-side tests. */
int yart_build_host(const float* verts, size_t n_verts,
                    const int32_t* tris, size_t n_tris,
                    YartBvhNode* nodes_out, size_t* n_nodes,
                    YartTriAccel* triaccel_out);

/* Replaces Camera::Camera (include/camera.h:9-22) for callers that hand a camera to yart_wavefront_render_diffuse or
 * generate primary rays themselves: the four vectors a Camera holds, in fp32 with the reference's expressions --
 * out[0..2] = position, out[3..5] = normal(dir) (src/rt.cpp:22-28 normalises before it constructs the camera,
 * include/vector.h:165-167), out[6..8] = right_ = cross(direction, (0,1,0)) * (width * fov / height), out[9..11] = up_ =
 * normal(cross(right_, direction)) * fov.  Camera::sample(x, y) (include/camera.h:24-31) then is
 * normal(direction + right_ * (x / width - 0.5) + up_ * (y / height - 0.5)).  No HIP call. */
int yart_camera_basis(const float pos[3], const float dir[3], int32_t width, int32_t height, float fov, float out[12]);

/* The scene's COVER (no HIP call; no counterpart in the reference): the box of ray origins for which the
 * boxes of the engine's own tree are guaranteed to hold everything the reference's primitive tests can
 * accept, rounding noise included (yart_amd/csrc/cover.h) -- the scene's bounds widened by 1/16 to 8 of its
 * diagonals.  Rays that start outside it are answered on the reference's tree (fix-up launches).
 * out[0..2] = lo, out[3..5] = hi, out[6] = M (largest |coordinate| of a covered origin), out[7] = the farthest a
 * triangle's error band reaches beyond the triangle's own box (the pruned walk of the reference's tree,
 * YART_REFERENCE_TREE and the fix-up launches, culls that much later), out[8] unused, out[9] = bits of a uint32: 0
 * for an empty scene (every origin covered). */
int yart_scene_cover(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                     const float* spheres, size_t n_spheres, float out[10]);

/* Host-only build of the engine's 4-wide tree (no HIP call; layout as yart_accel_get_wide).
 * Call with NULL array outputs to query n_nodes / n_prims, then again with arrays of that
 * size.  max_leaf in 1..8 (the accel uses 1).  Used by CPU-side tests. */
int yart_build_wide_host(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                         const float* spheres, size_t n_spheres, int max_leaf,
                         void* nodes_out, size_t* n_nodes, void* rows_a_out, void* rows_bc_out,
                         size_t* n_prims, int* depth);

/* ---- Hit post-processing --------------------------------------------------
 * What the reference derives from a hit before it shades: for a triangle the geometric normal
 * normal(cross(e1,e2)) and the shading normal n0(1-u-v) + n1 u + n2 v, not normalised
 * (src/bvhaccel.cpp:283-292, include/triangle.h:158-171); for a sphere normal(orig + dir t - position)
 * for both (include/sphere.h:79-83).  One record per ray, the float fields of RayHitInfo
 * (include/vector.h:255-263) in its order, the borrowed Bsdf* / AreaLight* replaced by what
 * identifies them: for a triangle mesh_idx / tri_idx (Bsdf = the mesh's), for a sphere mesh_idx = -1
 * and tri_idx = its number in Scene::shapes_, for a miss both -1 and the normals 0.  Bit-identical
 * to the reference's RayHitInfo for the same hit. */
typedef struct YartSurface {    /* 48 bytes */
    float normal[3];
    float t;
    float shading_normal[3];
    float u;
    float v;
    int32_t prim;               /* as in YartHit */
    int32_t mesh_idx;
    int32_t tri_idx;
} YartSurface;

/* Uploads what the gather needs: per triangle (leaf order) its geometric normal and its three vertex
 * normals, per sphere its record.  verts / tris / spheres: the arrays yart_accel_create was given;
 * vertex_normals: 3 floats per vertex, same numbering as verts (TriangleMesh::normals_ of every
 * mesh, concatenated).  +64 bytes of HBM per triangle.  Part of setting the accel up: not to be
 * called while yart_surface* calls on the same accel are in flight. */
int yart_accel_set_surface_data(YartAccel* accel, const float* verts, size_t n_verts,
                                const int32_t* tris, size_t n_tris, const float* vertex_normals,
                                const float* spheres, size_t n_spheres);
/* d_rays, d_hits (results of yart_intersect_closest*), d_out: device pointers; asynchronous on `stream`. */
int yart_surface_dev(YartAccel* accel, const YartRay* d_rays, const YartHit* d_hits, size_t n,
                     YartSurface* d_out, void* stream);
/* Host buffers; synchronous. */
int yart_surface(YartAccel* accel, const YartRay* rays, const YartHit* hits, size_t n, YartSurface* out);

/* ---- intersection: ray queue with pinned host mirrors ------------------------------------------
 * For a renderer whose shading stays on the host (the reference's BSDFs, lights and Rng,
 * src/renderer.cpp:59-129) but which submits rays in batches: device arrays for `capacity` rays and
 * their results, mirrored in page-locked host memory.  The caller writes rays into the mirror,
 * yart_queue_closest / _any copy them to the device (DMA from pinned memory on the queue's own stream),
 * run the batch -- and, on request, the hit post-processing of row K on the device -- copy the results
 * back and return when they are in the mirrors.  One batch at a time per queue; queues are independent.
 * Used by yart_amd/cxx/wavefront.hpp; no counterpart in the reference (one ray per call). */
typedef struct YartQueue YartQueue;
int yart_queue_create(YartAccel* accel, size_t capacity, YartQueue** out);
void yart_queue_destroy(YartQueue* queue);
YartRay* yart_queue_rays(YartQueue* queue);                 /* host mirror, `capacity` entries, caller writes */
const YartHit* yart_queue_hits(const YartQueue* queue);     /* valid after yart_queue_closest */
const YartSurface* yart_queue_surfaces(const YartQueue* queue);   /* valid after yart_queue_closest(.., with_surface = 1) */
const uint8_t* yart_queue_occluded(const YartQueue* queue); /* valid after yart_queue_any */
int yart_queue_closest(YartQueue* queue, size_t n, unsigned flags, int with_surface);
int yart_queue_any(YartQueue* queue, size_t n, unsigned flags);

/* ---- Wavefront driver, device resident ----------------------------------------------
 * Replaces the per-pixel synchronous loop of the reference's renderer -- TileTask::run -> trace()
 * (src/renderer.cpp:29-136, :146-153), which calls Scene::intersect / intersectShadow one ray at a time
 * -- by path queues that never leave HBM: a pool of `slots` paths, each with its pending ray, hit,
 * surface record (row K), shadow ray and occlusion flag, advanced one bounce per iteration by
 *     closest-hit batch -> yart_surface_dev -> shade kernel -> any-hit batch -> apply kernel,
 * every batch `slots` rays wide.  A path that ends frees its slot; with `regenerate` the next
 * iteration starts a new camera path there, so batches stay full until the frame's paths run out
 * (the rays of one batch then come from different bounces: the incoherent stream BASELINE's metric is
 * quoted on); without it a pool is traced bounce by bounce to the end before the next one starts
 * (per-bounce batch statistics).
 *
 * The integrator that runs on the device is the diffuse path tracer of BASELINE configs[2]: trace()'s
 * structure with one point light and Lambertian surfaces (camera sample with pixel jitter
 * include/camera.h:24-31; light sample include/light.h:41-52; shadow ray from P + wi EPS, :101-102;
 * Russian roulette on the weight's luminance, :113-115; cosine-weighted bounce in the hit's Frame,
 * include/frame.h:18-32; next ray from P + dir EPS, :129; colour x 1/spp per sample, :132).  Its random
 * numbers are a counter-based hash of (seed, pixel number y*width+x as in :33, sample, bounce, purpose),
 * not the reference's std::mt19937_64 stream: a path's result does not depend on which slot or batch
 * traced it, and the frame is reproducible bit for bit (samples of a pixel are summed in order).  A
 * renderer whose shading stays on the host (the reference's BSDFs, lights and Rng) uses
 * yart_amd/cxx/wavefront.hpp over the same device calls instead. */
typedef struct YartWavefront YartWavefront;

typedef struct YartDiffuseJob {
    float cam_pos[3];           /* Camera::position_ */
    float cam_dir[3];           /* Camera::direction_ (unit) */
    float cam_right[3];         /* Camera::right_ = cross(dir, up) * (width * fov / height) */
    float cam_up[3];            /* Camera::up_    = normal(cross(right, dir)) * fov */
    int32_t width, height;
    int32_t spp;                /* samples per pixel (the reference: 4096, src/renderer.cpp:36) */
    int32_t max_depth;          /* closest-hit rays per path at most (the reference: 10, :58) */
    float light_pos[3];         /* PointLight */
    float light_intensity[3];
    float albedo[3];            /* Lambertian reflectance of every surface */
    uint32_t seed;
    const uint32_t* d_pixels;   /* device array of pixel numbers y*width+x to render (a rank's share of the
                                   tile list, yart_tiles_pixels), or NULL for all pixels in row-major order */
    size_t n_pixels;            /* entries of d_pixels; ignored when it is NULL */
    int32_t regenerate;         /* 1: refill freed slots every iteration; 0: pool by pool, bounce by bounce */
    unsigned flags;             /* yart_flags for the intersect calls (YART_SORT_RAYS ...) */
} YartDiffuseJob;

typedef struct YartWavefrontIter {   /* one iteration = one bounce of every path in flight */
    uint64_t closest_rays;      /* live paths traced by the closest-hit batch */
    uint64_t shadow_rays;       /* of them hit a surface and sent a shadow ray */
    float ms_closest;           /* closest-hit batch (sort + traversal), HIP events on the stream */
    float ms_surface;           /* 0 since the surface step became part of the shade kernel (kept for layout) */
    float ms_shadow;            /* any-hit batch */
    float ms_shade;             /* generate + shade + apply kernels */
} YartWavefrontIter;

typedef struct YartWavefrontStats {
    uint64_t paths, closest_rays, shadow_rays, iterations;
    double ms_total;            /* wall time of the call, host clock around a final synchronise */
    double ms_closest, ms_surface, ms_shadow, ms_shade;
} YartWavefrontStats;

/* slots: paths in flight (>= 1; 2^20 .. 2^23 fills the GPU).  The accel must have its surface data
 * (yart_accel_set_surface_data).  HBM: 176 bytes per slot + 16 bytes per path of a sample slab. */
int yart_wavefront_create(YartAccel* accel, size_t slots, YartWavefront** out);
void yart_wavefront_destroy(YartWavefront* wf);
/* Renders job->spp samples of every listed pixel into d_rgb (device, 3 floats per listed pixel, in
 * list order; overwritten).  iters / max_iters: optional per-iteration records (the first max_iters).
 * Synchronous: returns when the frame is in d_rgb. */
int yart_wavefront_render_diffuse(YartWavefront* wf, const YartDiffuseJob* job, float* d_rgb,
                                  YartWavefrontStats* stats, YartWavefrontIter* iters, size_t max_iters,
                                  void* stream);

/* ---- Multi-GPU image mode --------------------------------------------------------------
 * Replaces the reference's way of spreading a frame over workers: Renderer::render cuts the image into
 * 32x32 tiles (include/renderer.h:14, src/renderer.cpp:161-217) and 8 threads pop them from a queue
 * (src/scheduler.cpp:46-72).  Here the scene is replicated in every GPU's HBM, tile i of the SAME list
 * belongs to rank i % world (one process per GPU), every rank renders its tiles with no communication,
 * and the one exchange of the path is the framebuffer gather at the end, over RCCL (xGMI inside a node).
 * Pure index arithmetic (host, no HIP call): */
size_t yart_tiles_count(int32_t width, int32_t height, int32_t tile);
/* out: 4 int32 per tile = x0, y0, x1, y1 (end exclusive), in the reference's order: full tiles column
 * by column, then the right-hand strip, the bottom strip, the corner. */
int yart_tiles_list(int32_t width, int32_t height, int32_t tile, int32_t* out);
/* Pixel numbers y*width+x of rank `rank`'s tiles, tile by tile, rows first (the order TileTask::run
 * visits them, src/renderer.cpp:146-153).  out_pixels == NULL: returns the count only.  This list is
 * what YartDiffuseJob.d_pixels takes, and the order of a rank's part in yart_frame_gather. */
size_t yart_tiles_rank_pixels(int32_t width, int32_t height, int32_t tile, int32_t rank, int32_t world,
                              uint32_t* out_pixels);

/* Communicator over RCCL (librccl is loaded when the first communicator is made; a host without it
 * gets YART_ERR_NO_DEVICE and keeps everything else).  Rank 0 makes the 128-byte id and hands it to the
 * other ranks by whatever the host program has (a file, MPI, a socket); every rank then joins. */
typedef struct YartComm YartComm;
int yart_comm_unique_id(void* id128);
int yart_comm_create(const void* id128, int32_t rank, int32_t world, int device, YartComm** out);
void yart_comm_destroy(YartComm* comm);
/* d_local_rgb: this rank's pixels, 3 floats each, in yart_tiles_rank_pixels order (device).  Rank 0
 * receives every rank's part (one grouped ncclSend / ncclRecv exchange: a direct gather uses all inbound
 * xGMI links of the root at once) and scatters the pixels into d_frame_rgb (width*height*3 floats,
 * row-major, device; ignored on other ranks).  Asynchronous on `stream`; world == 1 is the local scatter. */
int yart_frame_gather(YartComm* comm, const float* d_local_rgb, int32_t width, int32_t height, int32_t tile,
                      float* d_frame_rgb, void* stream);
/* ONE gather may be in flight per communicator: rank 0's staging block and tile tables belong to the communicator, and a
 * call on another stream, or with another frame shape, first waits for the earlier gather's stream.
 *
 * The second exchange SURVEY 8(e) names: ray / hit counters and a checksum summed over the ranks (the reference's
 * counterpart is none -- its workers share one address space, src/scheduler.cpp:46-72).  d_values: n uint64 on the
 * communicator's device, replaced on every rank by their sum over all ranks (ncclAllReduce, ncclSum).  world == 1 leaves
 * them as they are.  Asynchronous on `stream`. */
int yart_comm_allreduce_u64(YartComm* comm, uint64_t* d_values, size_t n, void* stream);

/* ---- Scene ingestion -------------------------------------------------------------
 * Replaces Scene::loadFromObj(folder, file) (src/scene.cpp:159-298) and, beneath it, the bundled
 * tinyobjloader's LoadObj / LoadMtl (external/tinyobjloader/tiny_obj_loader.h:838-1133, :587-813):
 * Wavefront OBJ + MTL -> the flat scene yart_accel_create takes.  Same meshes (one per usemtl / g / o
 * run of faces), same per-mesh vertex numbering, same triangles, same float bits (tinyobjloader's own
 * number conversion, not strtod), per-vertex normals from the file or generated as
 * include/triangle.h:76-111 does, the mesh's material, and the light sphere the reference appends to
 * every OBJ scene.  Host-only: no HIP call.  The reference ignores load failures; this returns
 * YART_ERR_INVALID_ARGUMENT when the file cannot be opened or a face names a vertex that does
 * not exist. */
typedef struct YartObjScene YartObjScene;

typedef struct YartObjMesh {        /* 24 bytes */
    int32_t first_vertex;           /* into the concatenated vertex / normal arrays */
    int32_t num_vertices;
    int32_t first_triangle;
    int32_t num_triangles;
    int32_t material;               /* index into the materials, -1: none named (the reference
                                       indexes its Bsdf list with -1 there, src/scene.cpp:261-266) */
    int32_t normals_from_file;      /* 1: `vn` of the file, 0: generated */
} YartObjMesh;

typedef struct YartObjMaterial {    /* 40 bytes: what src/scene.cpp:172-208 reads of a material */
    float diffuse[3];
    float specular[3];
    float shininess;
    float ior;
    int32_t illum;
    int32_t bsdf;                   /* 0 Lambertian(diffuse), 1 Torrance-Sparrow conductor(shininess),
                                       2 Fresnel dielectric(ior), 3 black Lambertian (src/scene.cpp:183-207) */
} YartObjMaterial;

int yart_obj_load(const char* folder, const char* file, YartObjScene** out);
void yart_obj_free(YartObjScene* scene);
size_t yart_obj_num_meshes(const YartObjScene* scene);
size_t yart_obj_num_vertices(const YartObjScene* scene);
size_t yart_obj_num_triangles(const YartObjScene* scene);
size_t yart_obj_num_materials(const YartObjScene* scene);
size_t yart_obj_num_spheres(const YartObjScene* scene);
/* Copies out whichever arrays are non-NULL: verts / normals 3 floats per vertex (meshes
 * concatenated), tris 5 int32 per triangle as yart_accel_create takes them (vertex numbers already
 * offset by the mesh's first_vertex), spheres 4 floats each. */
int yart_obj_get(const YartObjScene* scene, float* verts, float* normals, int32_t* tris, float* spheres,
                 YartObjMesh* meshes, YartObjMaterial* materials);
const char* yart_obj_mesh_name(const YartObjScene* scene, size_t mesh);
const char* yart_obj_material_name(const YartObjScene* scene, size_t material);

/* ---- Film output -------------------------------------------------------------------
 * Replaces Camera::saveImage -> Bitmap::write (include/camera.h:53-56, src/bitmap.cpp:8-72): the
 * float framebuffer (r,g,b per pixel, row 0 on top, as Camera::accumulate fills it) is tone-mapped
 * with the reference's curve, min(255, (uint8)(powf(1 - expf(-x), 1/2.2f) * 255 + 0.5f)), and written
 * as a 24-bit bottom-up BMP with the reference's headers.  Host-only.  The reference's writer tests
 * its padding array instead of the padding size (:64) and does not compile with g++ 11; rows are
 * padded to four bytes here.  Negative / NaN radiance (undefined in the reference) maps to 0. */
int yart_film_tonemap(const float* rgb, size_t n_pixels, uint8_t* bgr_out);
int yart_film_write_bmp(const char* path, const float* rgb, int32_t width, int32_t height);

/* Per-kernel timing (no counterpart in the reference, which only has a wall-clock
 * Timer around the whole render, include/timer.h:6-26).  While enabled, every
 * intersect call brackets its launches with HIP events on the caller's stream, in three
 * groups: ms_sort (coherence sort), ms_triangles (the MAIN traversal kernel: the 4-wide tree
 * kernel, or the triangle-tree pass with YART_REFERENCE_TREE / _RULE) and ms_spheres (the other
 * passes: sphere-tree pass of the reference-tree path, fix-up launches of the default path); yart_accel_profile_read synchronises those events,
 * adds their durations (milliseconds) and launch counts to the outputs and clears
 * the list.  Diagnostic; the span list is locked, so calls from several host threads are safe. */
int yart_accel_profile_enable(YartAccel* accel, int enable);
int yart_accel_profile_read(YartAccel* accel, double* ms_sort, double* ms_spheres,
                            double* ms_triangles, uint64_t* n_calls);

const char* yart_status_string(int status);
/* Text of the last failure on the calling thread ("" if none). */
const char* yart_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* YART_AMD_H */
