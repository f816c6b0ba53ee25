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
// Execution shape: persistent workgroups; every wave pulls chunks of consecutive
// rays from a global counter and refills idle lanes from its chunk, so lanes that
// finish early do not idle until the slowest ray of a fixed 64-ray group is done.
// Per-lane traversal stack: the top kLdsStack entries live in LDS (bank = lane,
// conflict free for any mix of stack heights), deeper entries spill to scratch.
// Loop structure is while-while: all lanes descend until none is at an interior
// node, then all lanes that reached a leaf test its primitives together.
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
constexpr uint32_t kChunk = 256;     // rays a wave takes from the global counter at a time
constexpr int kRefillMin = 8;        // refill once this many lanes are idle

struct RayState {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;                // 1.0f / dir, the division of include/bbox.h:36 hoisted
    float tmin, tmax;
};

// include/bbox.h:30-50.  `tFar0` is ray.maxT in the reference; the pruned rule
// passes the current closest t.  The per-axis early-outs of the reference are
// folded into one final test: tNear only grows, tFar only shrinks and neither can
// become NaN (NaN operands fail every comparison that would assign them).
__device__ __forceinline__ bool slab(float lx, float ly, float lz, float hx, float hy, float hz,
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

__device__ __forceinline__ float sel3(int i, float a, float b, float c)
{
    return (i == 0) ? a : ((i == 1) ? b : c);
}

template <int PRIM, bool ANY, bool PRUNE, bool COUNT, bool PERSIST>
__global__ void __launch_bounds__(kBlock) traceKernel(TraceParams P)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    uint32_t ovf[kOvfStack];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t laneBit = 1ull << lane;

    RayState r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.tmin = r.tmax = 0.0f;
    uint32_t cur = kIdle;            // encoded node reference, see bvh_build.h
    int sp = 0;
    bool live = false;               // lane owns a ray whose result is not written yet
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
        rayId = P.perm ? P.perm[slot] : slot;
        const float4 a = P.rays[(size_t)rayId * 2 + 0];
        const float4 b = P.rays[(size_t)rayId * 2 + 1];
        r.ox = a.x; r.oy = a.y; r.oz = a.z; r.tmin = a.w;
        r.dx = b.x; r.dy = b.y; r.dz = b.z; r.tmax = b.w;
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
        // The reference's sphere loop never looks at ray.minT (include/sphere.h:63-97 accepts
        // any t in (1e-4, current t)); only its triangle BVH culls by it (include/bbox.h:32,49).
        if (PRIM == 1) r.tmin = 0.0f;
        tcur = r.tmax; hu = 0.0f; hv = 0.0f; prim = -1; minS = -1; maxL = -1; occ = 0;
        cNodes = 0; cPrims = 0;
        sp = 0;
        live = true;
        cur = 0u;                    // pair 0 = { root, empty leaf }, axis bits 0
        if (P.chained) {
            if (ANY) {
                if (P.occluded[rayId]) { live = false; cur = kIdle; }   // a sphere already occludes
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
            if (!exhausted && (nIdle >= kRefillMin || idleMask == ~0ull)) {
                if (chunkNext == chunkEnd) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.workCounter, kChunk);
                    base = __shfl(base, 0);
                    if (base >= P.n) { exhausted = true; }
                    else { chunkNext = base; chunkEnd = (P.n - base < kChunk) ? P.n : base + kChunk; }
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
                if (slot < P.n) startRay(slot);
                firstPass = false;
            }
            if (__ballot(cur != kIdle) == 0) break;
        }

        // ------------------------------------------------------------ descend
        while (__ballot((cur & kLeafBit) == 0) != 0) {
            if ((cur & kLeafBit) == 0) {
                const float4* p = P.pairs + (size_t)(cur & 0x1FFFFFFFu) * 4;
                const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                const uint32_t axis = cur >> 29;
                const float bound = (PRUNE && !ANY) ? tcur : r.tmax;
                const bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, bound);
                const bool h1 = slab(q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, r, bound);
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
        if (cur != kIdle) {
            const uint32_t slot0 = (cur & 0x7FFFFFFFu) >> 3;
            const uint32_t cnt = cur & 7u;
            bool stop = false;
            if (PRIM == 0) {
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4* tp = P.prims + (size_t)(slot0 + i) * 3;
                    const float4 A = tp[0], B = tp[1], C = tp[2];
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
                    const int32_t idx = (int32_t)P.primIndex[slot0 + i];
                    const bool below = t < (double)f;
                    if (minS < 0 || f < tcur) { tcur = f; minS = idx; maxL = below ? idx : -1; }
                    else if (f == tcur) {
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
            if (ANY) {
                P.occluded[rayId] = (uint8_t)occ;
            } else {
                if (PRIM == 1) prim = (minS < 0) ? -1 : -((minS > maxL ? minS : maxL) + 2);
                // include/scene.h:167: a hit only counts if t < ray.maxT
                if (PRIM == 0 && !(tcur < r.tmax)) { prim = -1; }
                if (prim == -1) { hu = 0.0f; hv = 0.0f; }
                P.hits[rayId] = make_float4(tcur, hu, hv, __int_as_float(prim));
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
