// Internal interface between the C-ABI layer (capi.cpp) and the kernels (traverse.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_build.h"
#include "cover.h"

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
    const uint32_t* deferFlag;  // fix-up launch after the wide kernel: deferFlag[0] = rays it left behind, deferFlag[1] != 0: more
                                // than its list holds
    const uint32_t* deferList;  // their ray numbers, deferFlag[0] of them (at most deferCap)
    uint32_t deferCap;
    uint32_t onlyDeferred;      // 1: trace only the rays the wide kernel left behind: the listed ones, or -- when the list
                                // overflowed -- every ray of the batch whose output holds the marker below
    Cover cover;                // sphere tree: a ray that starts outside it gets its own pad on the sphere boxes
    float sceneLo[3], sceneHi[3];   // bounds of the scene (for that pad)
};

// What the wide kernel leaves in the output of a ray it does not answer (irregular, outside the cover, too nearly
// axis-parallel for the box arithmetic) until the fix-up launches overwrite it; the fix-up sphere pass keeps the ray marked
// for the triangle pass (kUnresolvedV in YartHit.v / kUnresolvedOcc).  include/yart_amd.h: YART_PRIM_UNRESOLVED.
constexpr int32_t kUnresolvedPrim = INT32_MIN;
constexpr uint8_t kUnresolvedOcc = 2;
constexpr uint32_t kUnresolvedV = 0x7FC0DEFEu;

// The wide-tree kernel (traverse_wide.hip) handles "regular" rays: finite origin, finite non-zero
// 1/dir, minT == 0 (what the reference's Ray constructor sets, include/vector.h:244-249) and a maxT
// that is not NaN.  For anything else the reference's result depends on how ITS tree's boxes treat
// NaN plane distances and minT (include/bbox.h:30-50), so those rays are deferred to the kernels
// that walk the reference's own tree.  One predicate, used by both sides.
__host__ __device__ inline bool regularRay(float ox, float oy, float oz, float tmin,
                                           float dx, float dy, float dz, float tmax)
{
    const float inf = __builtin_inff();
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    return (ix != 0.0f) && (__builtin_fabsf(ix) < inf) && (iy != 0.0f) && (__builtin_fabsf(iy) < inf) &&
           (iz != 0.0f) && (__builtin_fabsf(iz) < inf) &&
           (__builtin_fabsf(ox) < inf) && (__builtin_fabsf(oy) < inf) && (__builtin_fabsf(oz) < inf) &&
           (tmin == 0.0f) && (tmax == tmax);
}

struct WideParams {
    const float4* units;        // the packed tree (wide_build.h: PackedWide), 16-byte units
    const float4* triaccel;     // the reference's leaf-ordered TriAccel records, 3 x float4 each: second-stage rows of a triangle
    const float4* rays;
    const uint32_t* perm;
    uint32_t n;
    float4* hits;
    uint8_t* occluded;
    uint32_t* counts;
    uint32_t* workCounter;      // 256-byte slot zeroed by the launcher: counters at words 0, 8, .. 56
    uint32_t* deferFlag;        // words 1, 2 of the same slot: rays left for the fix-up launches (counted, and listed in deferList
                                // while it has room) / set when the list overflowed
    uint32_t* deferList;
    uint32_t deferCap;
    uint32_t refillMin;
    uint32_t nodeMin;
    uint32_t sphereMin;         // sphere leaves are tested once this many lanes of a wave hold one
    uint32_t hasSpheres;        // 0: a tree of triangle leaves only (the kernel instance without the sphere side)
    uint32_t laneCap;           // lanes of a wave that take rays (0 or 64: all): small batches spread over more, emptier waves
    uint32_t maxBlocksPerCU;    // 0 = as many as fit (8)
    uint32_t chunk;             // rays a wave takes from its range counter at a time (multiple of 64)
    const float4* refNodes;     // the reference's flattened nodes (YartBvhNode, 2 x float4 each): the box of a triangle's leaf
                                // there (near-edge check), its topology (which of two tied leaves its walk reaches first)
    const uint32_t* leafOfSlot; // leaf slot -> index of its leaf in refNodes
    Cover cover;                // rays that start outside it are left to the fix-up launches (cover.h)
    float cullReach;            // max over the axes of (longest way from a covered origin to the far side of the tree's bounds
                                // + 1.25 x their extent): |A| + 255 |B| of slab4 <= cullReach x the largest |1 / d|
    float cullMargin;           // 2^-21 cullReach
    // experiment (YART_WIDE_POOL, traverse_wide.hip: tracePool): slots per lane class (0: off), stack entries per slot in
    // LDS, threads per block, lanes at interior nodes below which a wave goes back to the masks, leaf-step thresholds
    uint32_t poolSlots, poolWindow, poolBlock, poolKeep, poolTriMin, poolSphMin;
};

hipError_t launchTraceWide(const WideParams& P, bool any, bool count, bool persistent, int numCUs,
                           hipStream_t stream);

// leafOfSlot[slot] = index of the slot's leaf in `nodes` (YartBvhNode array in the reference's order).
hipError_t launchLeafOfSlot(const void* nodes, size_t numNodes, uint32_t* leafOfSlot, hipStream_t stream);

// prim: 0 = triangle tree, 1 = sphere tree.
hipError_t launchTrace(const TraceParams& P, int prim, bool any, bool prune, bool count,
                       bool persistent, int numCUs, hipStream_t stream);

// Ray sorting (sort.hip): (*perm)[0..n) = ray numbers ordered by a Morton key of
// the ray origin inside [lo,hi] plus the direction octant.  *perm points into `workspace`
// (sortWorkspaceBytes(n) bytes), valid until the workspace is reused.
hipError_t sortRays(const float4* rays, uint32_t n, const float lo[3], const float hi[3], uint32_t octShift,
                    uint32_t gridBits, const uint32_t** perm, void* workspace, size_t workspaceBytes, hipStream_t stream);
size_t sortWorkspaceBytes(uint32_t n);

// Hit post-processing (surface.hip): rows = 64 B per triangle in leaf-slot order (geometric normal, the three
// vertex normals, meshIdx, triIdx, 8 B unused), spheres = cx,cy,cz,r in the scene's order.
hipError_t launchSurface(const void* rays, const void* hits, uint32_t n, const void* rows,
                         const void* spheres, void* out, hipStream_t stream);

} // namespace yart
