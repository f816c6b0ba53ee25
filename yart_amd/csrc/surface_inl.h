// surface_inl.h -- what the reference derives from a hit before it shades (src/bvhaccel.cpp:283-292,
// include/triangle.h:158-171, include/sphere.h:79-83), as one device function shared by the batch kernel
// (surface.hip: surfaceKernel) and the wavefront driver's shade kernel (wavefront.hip), which folds the step in.
// Compiled with -ffp-contract=off; sqrtf and '/' are hipcc's correctly rounded ones: the bits are the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace yart {

struct SurfacePoint {
    float nx, ny, nz;       // geometric normal (triangle) / normal at the hit point (sphere); 0 for a miss
    float sx, sy, sz;       // shading normal, not normalised
    int meshIdx, triIdx;
};

// h = the hit record (t, u, v, prim); rows = 64-byte surface rows in leaf-slot order; o, d = the ray (spheres only)
__device__ __forceinline__ SurfacePoint surfaceOfHit(const float4& h, const float4* __restrict__ rows, const float4* __restrict__ spheres,
                                                     const float4* __restrict__ ray)
{
    SurfacePoint s;
    s.nx = s.ny = s.nz = s.sx = s.sy = s.sz = 0.0f;
    s.meshIdx = -1; s.triIdx = -1;
    const int prim = __float_as_int(h.w);
    if (prim >= 0) {
        const float4 r0 = rows[static_cast<size_t>(prim) * 4 + 0];     // geometric normal, n0.x
        const float4 r1 = rows[static_cast<size_t>(prim) * 4 + 1];     // n0.yz, n1.xy
        const float4 r2 = rows[static_cast<size_t>(prim) * 4 + 2];     // n1.z, n2.xyz
        const float4 r3 = rows[static_cast<size_t>(prim) * 4 + 3];     // meshIdx, triIdx
        s.meshIdx = __float_as_int(r3.x);
        s.triIdx = __float_as_int(r3.y);
        s.nx = r0.x; s.ny = r0.y; s.nz = r0.z;
        const float u = h.y, v = h.z;
        const float w = 1.0f - u - v;                                   // include/triangle.h:168
        s.sx = r0.w * w + r1.z * u + r2.y * v;
        s.sy = r1.x * w + r1.w * u + r2.z * v;
        s.sz = r1.y * w + r2.x * u + r2.w * v;
    } else if (prim <= -2) {
        const float4 o = ray[0], d = ray[1];
        const float4 sp = spheres[-(prim + 2)];
        const float t = h.x;
        const float x = o.x + d.x * t - sp.x, y = o.y + d.y * t - sp.y, z = o.z + d.z * t - sp.z;  // include/sphere.h:80
        const float inv = 1.0f / sqrtf(x * x + y * y + z * z);       // include/vector.h:165-168; hipcc's default
        // sqrtf and '/' are the correctly rounded ones (__fsqrt_rn is the NATIVE approximation here)
        s.nx = x * inv; s.ny = y * inv; s.nz = z * inv;
        s.sx = s.nx; s.sy = s.ny; s.sz = s.nz;
        s.triIdx = -(prim + 2);
    }
    return s;
}

} // namespace yart
