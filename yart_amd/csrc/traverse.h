// Internal interface between the C-ABI layer (capi.cpp) and the kernels (traverse.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_build.h"

namespace yart {

struct TraceParams {
    const float4* pairs;        // DevNodePair array viewed as 4 x float4 per pair
    const float4* prims;        // triangles: 3 x float4 per TriAccel; spheres: 1 x float4
    const uint32_t* primIndex;  // spheres only: leaf slot -> caller's sphere number
    const float4* rays;         // 2 x float4 per YartRay
    const uint32_t* perm;       // optional: traversal slot -> ray number (sorted batches)
    uint32_t n;
    float4* hits;               // closest-hit output (YartHit)
    uint8_t* occluded;          // any-hit output
    uint32_t* counts;           // optional 4 x uint32 per ray (see include/yart_amd.h)
    uint32_t* workCounter;      // persistent launch: 256-byte slot = 8 counters 32 B apart, zeroed by the launcher
    uint32_t chained;           // 1: triangle pass running after the sphere pass (hits/occluded/counts hold its results)
    uint32_t refillMin;         // persistent launch: refill a wave once this many lanes are idle
    uint32_t nodeMin;           // leave the descend loop when fewer lanes than this are at interior nodes
};

// prim: 0 = triangle tree, 1 = sphere tree.
hipError_t launchTrace(const TraceParams& P, int prim, bool any, bool prune, bool count,
                       bool persistent, int numCUs, hipStream_t stream);

// Ray sorting (sort.hip): fills perm[0..n) with ray numbers ordered by a Morton key of
// the ray origin inside [lo,hi] plus the direction octant.  `keys` and `tmp` are
// n-element scratch arrays.
hipError_t sortRays(const float4* rays, uint32_t n, const float lo[3], const float hi[3],
                    uint32_t* perm, void* workspace, size_t workspaceBytes, hipStream_t stream);
size_t sortWorkspaceBytes(uint32_t n);

} // namespace yart
