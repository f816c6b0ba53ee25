// Hit post-processing on the device: what the reference derives from a hit
// before shading -- geometric normal and interpolated shading normal of the triangle
// (src/bvhaccel.cpp:283-292, include/triangle.h:158-171) or the sphere's normal at the hit point
// (include/sphere.h:79-83) -- for a whole batch, keyed by (prim, u, v) of the hit records.
//
// HBM-bound gather: per ray 16 B hit in, 48 B out, plus one 64-byte surface row = one line per
// triangle hit (geometric normal precomputed on the host with the reference's expression, three
// vertex normals, meshIdx, triIdx) or the 32-B ray + one 16-byte sphere record per sphere hit.  One ray per lane, no divergence beyond the
// triangle / sphere / miss split.  Compiled with -ffp-contract=off; sqrt and the division are the
// correctly rounded IEEE ones, so the bits are the reference's.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/yart_amd.h"
#include "traverse.h"
#include "surface_inl.h"

namespace yart {
namespace {

__global__ __launch_bounds__(256) void surfaceKernel(const float4* __restrict__ rays, const float4* __restrict__ hits,
                                                     uint32_t n, const float4* __restrict__ rows,
                                                     const float4* __restrict__ spheres,
                                                     float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = hits[i];
    const SurfacePoint sp = surfaceOfHit(h, rows, spheres, rays + static_cast<size_t>(i) * 2);
    const float nx = sp.nx, ny = sp.ny, nz = sp.nz, sx = sp.sx, sy = sp.sy, sz = sp.sz;
    const int meshIdx = sp.meshIdx, triIdx = sp.triIdx;
    out[static_cast<size_t>(i) * 3 + 0] = make_float4(nx, ny, nz, h.x);
    out[static_cast<size_t>(i) * 3 + 1] = make_float4(sx, sy, sz, h.y);
    out[static_cast<size_t>(i) * 3 + 2] = make_float4(h.z, h.w, __int_as_float(meshIdx), __int_as_float(triIdx));
}

} // namespace

// The surface rows from the scene arrays on the device (yart_accel_set_surface_data for device-built accels: the host loop
// took 0.17 s for 10 M triangles).  include/triangle.h:158-164, include/vector.h:165-168 with hipcc's correctly rounded
// `/` and sqrtf under -ffp-contract=off: the host loop's bits.  *bad is set when a vertex index is out of range.
__global__ void __launch_bounds__(256) surfaceRowsKernel(const float* __restrict__ verts, uint32_t nVerts, const int32_t* __restrict__ tris,
                                                         uint32_t nTris, const float* __restrict__ normals,
                                                         const uint32_t* __restrict__ slotOfInput, float4* __restrict__ rows,
                                                         uint32_t* __restrict__ bad)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nTris) return;
    const int32_t* t = tris + (size_t)i * 5;
    const int32_t i0 = t[0], i1 = t[1], i2 = t[2];
    if (i0 < 0 || i1 < 0 || i2 < 0 || (uint32_t)i0 >= nVerts || (uint32_t)i1 >= nVerts || (uint32_t)i2 >= nVerts) { *bad = 1u; return; }
    const float* p0 = verts + (size_t)i0 * 3;
    const float* p1 = verts + (size_t)i1 * 3;
    const float* p2 = verts + (size_t)i2 * 3;
    const float e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e1z = p1[2] - p0[2];
    const float e2x = p2[0] - p0[0], e2y = p2[1] - p0[1], e2z = p2[2] - p0[2];
    const float cx = e1y * e2z - e1z * e2y;
    const float cy = e1z * e2x - e1x * e2z;
    const float cz = e1x * e2y - e1y * e2x;
    const float inv = 1.0f / sqrtf(cx * cx + cy * cy + cz * cz);
    const float* n0 = normals + (size_t)i0 * 3;
    const float* n1 = normals + (size_t)i1 * 3;
    const float* n2 = normals + (size_t)i2 * 3;
    float4* r = rows + (size_t)slotOfInput[i] * 4;
    r[0] = make_float4(cx * inv, cy * inv, cz * inv, n0[0]);
    r[1] = make_float4(n0[1], n0[2], n1[0], n1[1]);
    r[2] = make_float4(n1[2], n2[0], n2[1], n2[2]);
    r[3] = make_float4(__int_as_float(t[3]), __int_as_float(t[4]), 0.0f, 0.0f);
}

hipError_t launchSurfaceRows(const float* verts, uint32_t nVerts, const int32_t* tris, uint32_t nTris, const float* normals,
                             const uint32_t* slotOfInput, void* rows, uint32_t* bad, hipStream_t stream)
{
    if (nTris == 0) return hipSuccess;
    surfaceRowsKernel<<<(nTris + 255) / 256, 256, 0, stream>>>(verts, nVerts, tris, nTris, normals, slotOfInput, static_cast<float4*>(rows), bad);
    return hipGetLastError();
}

hipError_t launchSurface(const void* rays, const void* hits, uint32_t n, const void* rows,
                         const void* spheres, void* out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    surfaceKernel<<<(n + 255) / 256, 256, 0, stream>>>(static_cast<const float4*>(rays), static_cast<const float4*>(hits), n,
                                                      static_cast<const float4*>(rows),
                                                      static_cast<const float4*>(spheres), static_cast<float4*>(out));
    return hipGetLastError();
}

} // namespace yart
