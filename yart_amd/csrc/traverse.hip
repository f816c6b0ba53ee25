// traverse.hip -- BVH traversal + primitive intersection kernels for gfx950 (MI355X).
//
// One ray per lane, wave64.  The tree is the reference's binary BVH
// (src/bvhaccel.cpp) stored as 64-byte SIBLING PAIRS of the reference's 32-byte
// node records (bvh_build.h): one 4 x dwordx4 fetch brings both children of a
// node, both are slab-tested (include/bbox.h:30-50) and visited front to back by
// the sign of ray.dir[splitAxis] exactly as the reference orders them
// (src/bvhaccel.cpp:255-261).  Leaves hold up to 7 Wald-projected triangles of
// 48 bytes (include/triaccel.h:16-32, test :169-201) or up to 4 spheres of 16
// bytes (fp64 test, include/sphere.h:63-97).
//
// Pass order is the reference's (include/scene.h:142-146, :176-182): the sphere
// tree first, then the triangle tree bounded by the sphere hit.
//
// Execution shape: persistent workgroups; the batch is cut into 8 ranges (one per
// XCD) with a counter each; every wave pulls chunks of consecutive rays, its own
// XCD's range first, and refills idle lanes from its chunk, so lanes that finish
// early do not idle until the slowest ray of a fixed 64-ray group is done.
// Per-lane traversal stack: the top kLdsStack entries live in LDS (bank = lane,
// conflict free for any mix of stack heights), deeper entries spill to scratch.
// Loop structure is while-while: lanes descend until fewer than `nodeMin` of them
// are at an interior node, then the lanes that reached a leaf test its primitives.
//
// What bounds it (measured, profiles/): every lane fetches its own 64-byte pair, so
// a wave instruction touches up to 64 different cache lines; the vector memory
// pipe retires about one 16-byte lane request per clock per CU for such scattered
// loads.  The kernel therefore spends its effort on issuing FEWER 16-byte requests
// (rows B and C of a triangle record are only fetched once row A has produced a t
// inside the ray interval) rather than on arithmetic.
//
// Arithmetic: built with -ffp-contract=off; every expression rounds exactly as the
// reference's C++ (no FMA contraction, IEEE division / sqrt), so with
// YART_REFERENCE_RULE the results are bit-identical to the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "traverse.h"

namespace yart {
namespace {

constexpr int kBlock = 256;          // 4 waves
constexpr int kLdsStack = 16;        // stack entries per lane kept in LDS
constexpr int kOvfStack = kMaxDepth - kLdsStack;
constexpr uint32_t kIdle = kEmptyLeaf;
constexpr uint32_t kChunk = 256;     // rays a wave takes from a range counter at a time

struct RayState {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;                // 1.0f / dir, the division of include/bbox.h:36 hoisted
    float tmin, tmax;
    float pad;                       // sphere tree: widening of every box for a ray that starts outside the scene's cover;
                                     // triangle tree, pruned: how much later a node is culled (see startRay)
};

// include/bbox.h:30-50, operation for operation.  `tFar0` is ray.maxT in the
// reference; the pruned rule passes the current closest t.  The per-axis early-outs
// of the reference are folded into one final test: tNear only grows, tFar only
// shrinks and neither can become NaN (NaN operands fail every comparison that would
// assign them).  The compare/select form keeps the reference's behaviour when a
// plane distance is NaN (0 * inf: origin on a box plane and a zero direction
// component).
__device__ __forceinline__ bool slabExact(float lx, float ly, float lz, float hx, float hy, float hz,
                                          const RayState& r, float tFar0)
{
    float tNear = r.tmin, tFar = tFar0;
    float t0, t1, s0, s1;
    t0 = (lx - r.ox) * r.ix; t1 = (hx - r.ox) * r.ix;
    s0 = (t1 < t0) ? t1 : t0; s1 = (t1 < t0) ? t0 : t1;
    tNear = (tNear < s0) ? s0 : tNear; tFar = (tFar > s1) ? s1 : tFar;
    t0 = (ly - r.oy) * r.iy; t1 = (hy - r.oy) * r.iy;
    s0 = (t1 < t0) ? t1 : t0; s1 = (t1 < t0) ? t0 : t1;
    tNear = (tNear < s0) ? s0 : tNear; tFar = (tFar > s1) ? s1 : tFar;
    t0 = (lz - r.oz) * r.iz; t1 = (hz - r.oz) * r.iz;
    s0 = (t1 < t0) ? t1 : t0; s1 = (t1 < t0) ? t0 : t1;
    tNear = (tNear < s0) ? s0 : tNear; tFar = (tFar > s1) ? s1 : tFar;
    return !(tFar < tNear) && (tFar > r.tmin) && (tNear < r.tmax);
}

// The same test for rays whose six plane distances cannot be NaN (finite origin,
// finite non-zero 1/dir): every select above then IS a min or a max, so the
// v_min/v_max/v_min3/v_max3 forms give the same tNear / tFar up to the sign of a
// zero, which no comparison can see.  Same products, same roundings.
__device__ __forceinline__ bool slabFast(float lx, float ly, float lz, float hx, float hy, float hz,
                                         const RayState& r, float tFar0)
{
    const float ax = (lx - r.ox) * r.ix, bx = (hx - r.ox) * r.ix;
    const float ay = (ly - r.oy) * r.iy, by = (hy - r.oy) * r.iy;
    const float az = (lz - r.oz) * r.iz, bz = (hz - r.oz) * r.iz;
    const float tNear = fmaxf(fmaxf(r.tmin, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float tFar = fminf(fminf(tFar0, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    return !(tFar < tNear) && (tFar > r.tmin) && (tNear < r.tmax);
}

// first ray of work range `i` of 8 (multiples of kChunk, last range takes the remainder)
__device__ __forceinline__ uint32_t rangeBegin(uint32_t n, uint32_t i)
{
    if (i >= 8) return n;
    const uint32_t chunks = (n + kChunk - 1) / kChunk;
    const uint32_t b = (uint32_t)(((uint64_t)chunks * i) / 8) * kChunk;
    return b < n ? b : n;
}

__device__ __forceinline__ float sel3(int i, float a, float b, float c)
{
    return (i == 0) ? a : ((i == 1) ? b : c);
}

__device__ __forceinline__ bool finiteNonZero(float v)
{
    return (v != 0.0f) && (fabsf(v) < __builtin_inff());
}

constexpr float kCullSlack = 1.0f + 0x1p-18f;

template <int PRIM, bool ANY, bool PRUNE, bool COUNT, bool PERSIST>
__global__ void __launch_bounds__(kBlock) traceKernel(TraceParams P)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    uint32_t ovf[kOvfStack];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t laneBit = 1ull << lane;

    RayState r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.tmin = r.tmax = r.pad = 0.0f;
    uint32_t cur = kIdle;            // encoded node reference, see bvh_build.h
    int sp = 0;
    bool live = false;               // lane owns a ray whose result is not written yet
    bool odd = false;                // this ray needs the NaN-preserving slab test
    uint32_t rayId = 0;
    float tcur = 0.0f, hu = 0.0f, hv = 0.0f;
    int32_t prim = -1;
    int32_t minS = -1, maxL = -1;    // sphere winner bookkeeping (see sphere leaf below)
    uint32_t occ = 0;
    uint32_t cNodes = 0, cPrims = 0;

    // wave-uniform work queue state
    uint32_t chunkNext = 0, chunkEnd = 0;
    bool exhausted = false;
    bool firstPass = true;
    // HW_REG_XCC_ID (id 20), bits 3:0 -- which of the 8 XCDs this wave runs on
    const uint32_t homeXcd = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    uint32_t rangesDone = 0;
    const int refillMin = (int)P.refillMin;
    const int nodeMin = (int)P.nodeMin;
    // Fix-up launch after the wide kernel: the rays it listed (or, when its list overflowed, every ray of the batch
    // whose output still holds the wide kernel's marker); nothing to do when it deferred none.  Wave-uniform.
    uint32_t nRays = P.n;
    const uint32_t* perm = P.perm;
    bool listed = false;
    if (P.onlyDeferred) {
        const uint32_t c = P.deferFlag[0];
        if (c == 0) return;
        if (P.deferFlag[1] == 0 && c <= P.deferCap) { nRays = c; perm = P.deferList; listed = true; }
    }

    auto push = [&](uint32_t v) {
        if (sp < kLdsStack) ldsStack[sp * kBlock + tid] = v;
        else ovf[sp - kLdsStack] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return kIdle;
        --sp;
        return (sp < kLdsStack) ? ldsStack[sp * kBlock + tid] : ovf[sp - kLdsStack];
    };

    auto startRay = [&](uint32_t slot) {
        rayId = perm ? perm[slot] : slot;
        const float4 a = P.rays[(size_t)rayId * 2 + 0];
        const float4 b = P.rays[(size_t)rayId * 2 + 1];
        r.ox = a.x; r.oy = a.y; r.oz = a.z; r.tmin = a.w;
        r.dx = b.x; r.dy = b.y; r.dz = b.z; r.tmax = b.w;
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
        if (P.onlyDeferred && !listed) {
            bool marked;
            if (ANY) marked = P.occluded[rayId] == kUnresolvedOcc;
            else {
                const float4 h = P.hits[rayId];
                marked = __float_as_int(h.w) == kUnresolvedPrim || __float_as_uint(h.z) == kUnresolvedV;
            }
            if (!marked) { live = false; cur = kIdle; sp = 0; return; }        // the wide kernel has answered this ray
        }
        // The reference's sphere loop never looks at ray.minT (include/sphere.h:63-97 accepts
        // any t in (1e-4, current t)); only its triangle BVH culls by it (include/bbox.h:32,49).
        if (PRIM == 1) {
            r.tmin = 0.0f;
            // The sphere boxes hold what the sphere test can accept from a ray of the scene's cover (cover.h: radius
            // sqrt(r^2 + 2^-20 D^2)).  A ray from outside it can have a longer |op|: all boxes are widened by
            // 2^-10 x the distance from its origin to the far corner of the scene, which bounds the excess.
            r.pad = 0.0f;
            if (!coverHolds(P.cover, r.ox, r.oy, r.oz)) {
                const float fx = fmaxf(fabsf(r.ox - P.sceneLo[0]), fabsf(P.sceneHi[0] - r.ox));
                const float fy = fmaxf(fabsf(r.oy - P.sceneLo[1]), fabsf(P.sceneHi[1] - r.oy));
                const float fz = fmaxf(fabsf(r.oz - P.sceneLo[2]), fabsf(P.sceneHi[2] - r.oz));
                r.pad = 0x1.04p-10f * (fx + fy + fz);
            }
        }
        if (PRIM == 0) {
            // A triangle's test can accept a point its error band (cover.h) beyond the triangle's box, hence beyond the
            // box of its leaf: up to padMax in space, padMax x the largest |1/d| along the ray.  Distance pruning
            // allows for it; a scene of well-conditioned triangles has a padMax of a few ulps.
            r.pad = P.cover.padMax * fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz));
            if (!(r.pad == r.pad)) r.pad = __builtin_inff();             // 0 x inf of an irregular ray: no pruning
        }
        odd = !(finiteNonZero(r.ix) && finiteNonZero(r.iy) && finiteNonZero(r.iz) &&
                fabsf(r.ox) < __builtin_inff() && fabsf(r.oy) < __builtin_inff() && fabsf(r.oz) < __builtin_inff() &&
                r.tmin == r.tmin && r.tmax == r.tmax);
        tcur = r.tmax; hu = 0.0f; hv = 0.0f; prim = -1; minS = -1; maxL = -1; occ = 0;
        cNodes = 0; cPrims = 0;
        sp = 0;
        live = true;
        cur = 0u;                    // pair 0 = { root, empty leaf }, axis bits 0
        if (P.chained) {             // triangle pass after the sphere pass
            if (ANY) {
                if (P.occluded[rayId] == 1) { live = false; cur = kIdle; }   // a sphere already occludes (2: marked by the sphere pass of a fix-up)
            } else {
                const float4 h = P.hits[rayId];                         // result of the sphere pass
                tcur = h.x; prim = __float_as_int(h.w);
            }
        }
    };

    for (;;) {
        // ------------------------------------------------------------ fetch work
        if (PERSIST) {
            const uint64_t idleMask = __ballot(cur == kIdle);
            const int nIdle = __popcll(idleMask);
            if (!exhausted && (nIdle >= refillMin || idleMask == ~0ull)) {
                // A wave drains the range of the XCD it runs on first, so that (with a sorted
                // batch) each XCD's private L2 serves one compact region of the scene, and only
                // then helps with the other ranges.
                while (chunkNext == chunkEnd && !exhausted) {
                    const uint32_t range = (homeXcd + rangesDone) & 7u;
                    const uint32_t rBegin = rangeBegin(nRays, range), rEnd = rangeBegin(nRays, range + 1);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.workCounter + range * 8, kChunk);
                    base = __builtin_amdgcn_readfirstlane(base) + rBegin;
                    if (base >= rEnd) { if (++rangesDone == 8) exhausted = true; }
                    else { chunkNext = base; chunkEnd = (rEnd - base < kChunk) ? rEnd : base + kChunk; }
                }
                if (!exhausted) {
                    const uint32_t avail = chunkEnd - chunkNext;
                    const uint32_t rank = __popcll(idleMask & (laneBit - 1));
                    if (cur == kIdle && rank < avail) startRay(chunkNext + rank);
                    chunkNext += ((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail;
                }
            }
            if (__ballot(cur != kIdle) == 0) {
                if (exhausted) break;
                continue;            // chunk ran dry with every lane idle: fetch the next one
            }
        } else {
            if (firstPass) {
                const uint32_t slot = blockIdx.x * kBlock + tid;
                if (slot < nRays) startRay(slot);
                firstPass = false;
            }
            if (__ballot(cur != kIdle) == 0) break;
        }

        // ------------------------------------------------------------ descend
        for (;;) {
            const uint64_t inner = __ballot((cur & kLeafBit) == 0);
            if (inner == 0) break;
            // leave for the leaf phase once few lanes still descend and some lane waits at a leaf
            if (__popcll(inner) < nodeMin && __ballot(cur != kIdle && (cur & kLeafBit)) != 0) break;
            if ((cur & kLeafBit) == 0) {
                const float4* p = P.pairs + (size_t)(cur & 0x1FFFFFFFu) * 4;
                const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                const uint32_t axis = cur >> 29;
                const float bound = (PRUNE && !ANY) ? __builtin_fmaf(tcur, kCullSlack, PRIM == 0 ? r.pad : 0.0f) : r.tmax;   // slack: see traverse_wide.hip
                bool h0, h1;
                const float pd = PRIM == 1 ? r.pad : 0.0f;      // x - 0 is x: the triangle tree's boxes are the reference's own
                if (__ballot(odd) != 0) {            // rare: some lane's ray can produce NaN plane distances
                    h0 = slabExact(q0.x - pd, q0.y - pd, q0.z - pd, q0.w + pd, q1.x + pd, q1.y + pd, r, bound);
                    h1 = slabExact(q2.x - pd, q2.y - pd, q2.z - pd, q2.w + pd, q3.x + pd, q3.y + pd, r, bound);
                } else {
                    h0 = slabFast(q0.x - pd, q0.y - pd, q0.z - pd, q0.w + pd, q1.x + pd, q1.y + pd, r, bound);
                    h1 = slabFast(q2.x - pd, q2.y - pd, q2.z - pd, q2.w + pd, q3.x + pd, q3.y + pd, r, bound);
                }
                const uint32_t r0 = __float_as_uint(q1.z), r1 = __float_as_uint(q3.z);
                if (COUNT) cNodes += 2;
                // src/bvhaccel.cpp:255: left child first iff ray.dir[splitAxis] > 0
                const bool leftFirst = sel3((int)axis, r.dx, r.dy, r.dz) > 0.0f;
                const uint32_t first = leftFirst ? r0 : r1, second = leftFirst ? r1 : r0;
                const bool hf = leftFirst ? h0 : h1, hs = leftFirst ? h1 : h0;
                if (hf) { if (hs) push(second); cur = first; }
                else if (hs) cur = second;
                else cur = pop();
            }
        }

        // ------------------------------------------------------------ leaf
        if (cur != kIdle && (cur & kLeafBit)) {
            const uint32_t slot0 = (cur & 0x7FFFFFFFu) >> 3;
            const uint32_t cnt = cur & 7u;
            bool stop = false;
            if (PRIM == 0) {
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4* tp = P.prims + (size_t)(slot0 + i) * 3;
                    const float4 A = tp[0];
                    if (COUNT) ++cPrims;
                    // include/triaccel.h:169-201
                    const int k = __float_as_int(A.w);
                    const float dk = sel3(k, r.dx, r.dy, r.dz), du = sel3(k, r.dy, r.dz, r.dx),
                                dv = sel3(k, r.dz, r.dx, r.dy);
                    const float ok = sel3(k, r.ox, r.oy, r.oz), ou = sel3(k, r.oy, r.oz, r.ox),
                                ov = sel3(k, r.oz, r.ox, r.oy);
                    const float nd = 1.0f / (dk + A.x * du + A.y * dv);
                    const float t = (A.z - ok - A.x * ou - A.y * ov) * nd;
                    if (!(tcur > t && t > 0.0001f)) continue;
                    const float4 B = tp[1], C = tp[2];             // rows b_*, c_* only for a t in range
                    const float phu = ou + t * du;
                    const float phv = ov + t * dv;
                    const float lambda = phu * B.x + phv * B.y + B.z;
                    if (lambda < 0.0f) continue;
                    const float mue = phu * C.x + phv * C.y + C.z;
                    if (mue < 0.0f) continue;
                    if (lambda + mue > 1.0f) continue;
                    if (ANY) { occ = 1; stop = true; break; }      // src/bvhaccel.cpp:266-272
                    tcur = t; hv = lambda; hu = mue;               // include/triaccel.h:194-196
                    prim = (int32_t)(slot0 + i);
                }
            } else {
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4 S = P.prims[slot0 + i];
                    if (COUNT) ++cPrims;
                    // include/sphere.h:63-97
                    const float opx = S.x - r.ox, opy = S.y - r.oy, opz = S.z - r.oz;
                    const double b = (double)(opx * r.dx + opy * r.dy + opz * r.dz);
                    double det = b * b - (double)(opx * opx + opy * opy + opz * opz) + (double)(S.w * S.w);
                    if (det < 0) continue;
                    det = sqrt(det);
                    const double eps = (double)1e-4f;
                    double t = b - det;
                    if (!(t > eps)) t = b + det;     // the far root is only tried when the near one fails
                    if (!(t > eps && t < (double)r.tmax)) continue;
                    if (ANY) { occ = 1; stop = true; break; }      // include/scene.h:176-180
                    // The reference walks the spheres in array order and accepts sphere i iff
                    // t_i < (double)current float t (include/sphere.h:78,87).  The outcome of
                    // that sequential scan, for ANY visiting order: F = min over candidates of
                    // float(t); among candidates with float(t) == F the winner is the larger of
                    // (lowest index) and (highest index whose double t is still < F).
                    const float f = (float)t;
                    if (minS >= 0 && f > tcur) continue;
                    const int32_t idx = (int32_t)P.primIndex[slot0 + i];
                    const bool below = t < (double)f;
                    if (minS < 0 || f < tcur) { tcur = f; minS = idx; maxL = below ? idx : -1; }
                    else {
                        minS = (idx < minS) ? idx : minS;
                        if (below && idx > maxL) maxL = idx;
                    }
                }
            }
            if (stop) { cur = kIdle; sp = 0; }
            else cur = pop();
        }

        // ------------------------------------------------------------ retire
        if (cur == kIdle && live) {
            live = false;
            // the sphere pass of a fix-up keeps the ray marked for the triangle pass that follows it
            const bool keepMark = PRIM == 1 && P.onlyDeferred;
            if (ANY) {
                P.occluded[rayId] = (keepMark && !occ) ? kUnresolvedOcc : (uint8_t)occ;
            } else {
                if (PRIM == 1) prim = (minS < 0) ? -1 : -((minS > maxL ? minS : maxL) + 2);
                // include/scene.h:167: a hit only counts if t < ray.maxT
                if (PRIM == 0 && !(tcur < r.tmax)) { prim = -1; }
                if (prim == -1) { hu = 0.0f; hv = 0.0f; }
                P.hits[rayId] = make_float4(tcur, hu, keepMark ? __uint_as_float(kUnresolvedV) : hv, __int_as_float(prim));
            }
            if (COUNT) {
                uint32_t* c = P.counts + (size_t)rayId * 4;
                c[2 * PRIM] = cNodes; c[2 * PRIM + 1] = cPrims;
                if (!P.chained) { c[2 - 2 * PRIM] = 0; c[3 - 2 * PRIM] = 0; }
            }
        }
    }
}

template <int PRIM, bool ANY, bool PRUNE, bool COUNT>
hipError_t launchT(const TraceParams& P, bool persistent, int numCUs, hipStream_t stream)
{
    if (P.n == 0) return hipSuccess;
    if (persistent) {
        int perCU = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &perCU, traceKernel<PRIM, ANY, PRUNE, COUNT, true>, kBlock, 0);
        if (e != hipSuccess) return e;
        if (perCU < 1) perCU = 1;
        if (perCU > 8) perCU = 8;
        uint32_t grid = (uint32_t)numCUs * (uint32_t)perCU;
        const uint32_t need = (P.n + kChunk - 1) / kChunk;       // one wave per chunk at most
        const uint32_t needBlocks = (need + 3) / 4;
        if (grid > needBlocks) grid = needBlocks;
        e = hipMemsetAsync(P.workCounter, 0, 256, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((traceKernel<PRIM, ANY, PRUNE, COUNT, true>), dim3(grid), dim3(kBlock), 0, stream, P);
    } else {
        const uint32_t grid = (P.n + kBlock - 1) / kBlock;
        hipLaunchKernelGGL((traceKernel<PRIM, ANY, PRUNE, COUNT, false>), dim3(grid), dim3(kBlock), 0, stream, P);
    }
    return hipGetLastError();
}

template <int PRIM, bool ANY>
hipError_t launchPC(const TraceParams& P, bool prune, bool count, bool persistent, int numCUs, hipStream_t s)
{
    if (prune) {
        return count ? launchT<PRIM, ANY, true, true>(P, persistent, numCUs, s)
                     : launchT<PRIM, ANY, true, false>(P, persistent, numCUs, s);
    }
    return count ? launchT<PRIM, ANY, false, true>(P, persistent, numCUs, s)
                 : launchT<PRIM, ANY, false, false>(P, persistent, numCUs, s);
}

} // namespace

hipError_t launchTrace(const TraceParams& P, int prim, bool any, bool prune, bool count,
                       bool persistent, int numCUs, hipStream_t stream)
{
    if (prim == 0) {
        return any ? launchPC<0, true>(P, prune, count, persistent, numCUs, stream)
                   : launchPC<0, false>(P, prune, count, persistent, numCUs, stream);
    }
    return any ? launchPC<1, true>(P, prune, count, persistent, numCUs, stream)
               : launchPC<1, false>(P, prune, count, persistent, numCUs, stream);
}

} // namespace yart
