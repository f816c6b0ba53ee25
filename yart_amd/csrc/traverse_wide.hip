// traverse_wide.hip -- the default closest-hit / any-hit kernel: one pass over the engine's own
// 4-wide BVH of triangles AND spheres (wide_build.h), distance-pruned by the best hit of either kind.
//
// Why this shape (measured on MI355X, tools/calib_gather.hip, tools/calib_valu.hip + profiles/): a one-ray-per-lane
// traversal pays per wave STEP -- the vector instructions of a node step or a leaf step are issued once for the
// wave however many of its lanes take part -- and per scattered line request (~165 G/s from L2, ~50 G/s beyond
// it), not per byte.  So: one node = four quantised child boxes per dependent fetch, three 16-byte requests on one
// 64-byte line (the binary tree's sibling pair costs four for two boxes), half the levels; the child references
// are one block address and four byte offsets (wide_build.h: PackedWide); leaf rows packed densely, the rows every
// primitive's first rejection needs apart from the rest; spheres in the same tree, so neither kind is traversed
// beyond the other's hit (the reference's order -- all spheres, then the triangle tree, include/scene.h:142-146 --
// cannot prune the sphere loop at all).
//
// Semantics reproduced (per ray, independent of visiting order):
//   triangles  include/triaccel.h:169-201 on the reference's TriAccel bits; accepted iff
//              1e-4 < t < min(best triangle t, sphere t)         (strict, as `info->t > t`)
//   spheres    include/sphere.h:63-97 in fp64; the sequential "accept iff t < (double)current float t"
//              scan over Scene::shapes_ is reproduced by the order-independent rule of traverse.hip
//   result     the triangle iff its t is strictly below the sphere's float t, else the sphere
//              (include/scene.h:142-167: spheres first, triangles must be strictly closer)
// Exact-t ties between two different triangles: the reference keeps the first one ITS traversal finds.  Inside one
// of its leaves that is the lower slot; across two of its leaves it is the leaf its walk reaches first -- at the lowest
// common ancestor of the two leaves the child on the side the ray points to (src/bvhaccel.cpp:255-261) -- which this
// kernel finds by descending the reference's own node array (tiedFirst below; rare, a few dependent loads).
//
// False positives of the reference's tests (cover.h): every primitive's box holds all its test can accept from a
// covered ray, and a triangle hit within the triangle's error band of an edge counts only if the ray passes the box
// of the triangle's leaf in the reference's tree (include/bbox.h:30-50) -- the condition under which the reference
// reaches that test at all.
//
// Rays that are not regularRay() (traverse.h) or start outside the scene's cover are left to the fix-up launches on the
// reference tree.  Such a ray's output holds a marker (kUnresolvedPrim / kUnresolvedOcc) until the fix-up launch
// overwrites it.
// Built with -ffp-contract=off: every tested expression rounds as the reference's C++ does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <mutex>

#include "traverse.h"
#include "wide_build.h"

namespace yart {
namespace {

constexpr int kBlock = 256;
constexpr int kLdsStack = 19;
constexpr int kOvfStack = kWideStack - kLdsStack;
constexpr uint32_t kIdle = kWideNone;
// Nodes are culled against the best hit so far times this: the plane distances of slab4 carry the rounding
// of the ray-side arithmetic (a few ulp), and a box whose computed entry lands one ulp beyond an equally
// distant hit found earlier would otherwise be skipped or not depending on the order in which a wave
// happened to find the two -- the result for a ray through a shared edge then changed from run to run.
// The primitive tests keep the exact bound.  (oracle/cpu_ref.cpp traverseWide mirrors it.)
constexpr float kCullSlack = 1.0f + 0x1p-18f;
constexpr uint32_t kNeedPop = 0xFFFFFFFDu;       // "take the next stack entry": leaf-like (bit 0), not a possible reference (unit < 2^30 - 2)
constexpr uint32_t kStepMask = 0x7FE00000u;      // a grid step in the packed node: exponent + two mantissa bits (wide_build.h)

__device__ __forceinline__ uint32_t rangeBegin(uint32_t n, uint32_t i, uint32_t kChunk)
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

struct WRay {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, tmax;
};

// include/bbox.h:30-50 for a regular ray (no NaN plane distance, minT = 0): the reference's own slab arithmetic
// against [0, maxT], on the box of a leaf of ITS tree (q0 = lo.xyz, hi.x; q1 = hi.y, hi.z).
__device__ __forceinline__ bool slabRef(const float4& q0, const float4& q1, const WRay& r)
{
    const float ax = (q0.x - r.ox) * r.ix, bx = (q0.w - r.ox) * r.ix;
    const float ay = (q0.y - r.oy) * r.iy, by = (q1.x - r.oy) * r.iy;
    const float az = (q0.z - r.oz) * r.iz, bz = (q1.y - r.oz) * r.iz;
    const float tNear = fmaxf(fmaxf(0.0f, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float tFar = fminf(fminf(r.tmax, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    return !(tFar < tNear) && (tFar > 0.0f) && (tNear < r.tmax);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Slab test of the four child boxes of a node against [0, bound] (minT is 0 for every ray this
// kernel takes).  Per axis the distance to grid plane q is t(q) = A + q * B with A = (origin - o) / d,
// B = scale / d; the sign of d says which of a child's two planes is the entry plane, so the packed
// plane dwords are swapped once per node instead of taking a min and a max per child and axis.
// Two children share one packed fma (v_pk_fma_f32).  The boxes are the engine's own (quantised,
// conservative), so this arithmetic is free to use fma; the primitive tests are what has to round
// like the reference.  tn[c] = entry distance, returns hit flags in h[c].
__device__ __forceinline__ void slab4(const float4& q0, const float4& q1, const float4& q2, const WRay& r,
                                      float bound, float margin, float tn[4], bool h[4])
{
    const uint32_t steps = __float_as_uint(q0.w);
    const float sx = __uint_as_float(steps & kStepMask), sy = __uint_as_float((steps << 10) & kStepMask),
                sz = __uint_as_float((steps << 20) & kStepMask);
    const float Ax = (q0.x - r.ox) * r.ix, Bx = sx * r.ix;
    const float Ay = (q0.y - r.oy) * r.iy, By = sy * r.iy;
    const float Az = (q0.z - r.oz) * r.iz, Bz = sz * r.iz;
    const uint32_t lx = __float_as_uint(q1.x), ly = __float_as_uint(q1.y), lz = __float_as_uint(q1.z);
    const uint32_t hx = __float_as_uint(q1.w), hy = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
    const bool negx = r.ix < 0.0f, negy = r.iy < 0.0f, negz = r.iz < 0.0f;
    const uint32_t nx = negx ? hx : lx, fx = negx ? lx : hx;      // byte c = entry / exit plane of child c
    const uint32_t ny = negy ? hy : ly, fy = negy ? ly : hy;
    const uint32_t nz = negz ? hz : lz, fz = negz ? lz : hz;
    // Entry planes are taken EARLIER by what the rounding of t = q B + A can amount to on that axis: A = (origin_node - o) / d
    // and B = step / d carry up to 2^-22 (|A| + 255 |B|) <= 2^-22 cullReach |1 / d| (every node's grid lies inside the
    // bounds of the tree, o inside the cover); margin = twice that per unit of |1 / d|.  So a box is never culled against a
    // hit that lies behind its true entry: without this the result for a ray through an edge shared by two triangles
    // depended on which of them a wave happened to find first.  Per axis, because a ray nearly parallel to an axis has a
    // huge |1 / d| there, and that axis must not loosen the other two.
    const float Axn = __builtin_fmaf(fabsf(r.ix), -margin, Ax), Ayn = __builtin_fmaf(fabsf(r.iy), -margin, Ay),
                Azn = __builtin_fmaf(fabsf(r.iz), -margin, Az);
    const f32x2 ax = { Ax, Ax }, bx = { Bx, Bx }, ay = { Ay, Ay }, by = { By, By }, az = { Az, Az }, bz = { Bz, Bz };
    const f32x2 axn = { Axn, Axn }, ayn = { Ayn, Ayn }, azn = { Azn, Azn };
#define YART_PAIR(w, s) (f32x2){ (float)(((w) >> (s)) & 255u), (float)(((w) >> ((s) + 8)) & 255u) }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 tnx = __builtin_elementwise_fma(YART_PAIR(nx, 16 * p), bx, axn);
        const f32x2 tny = __builtin_elementwise_fma(YART_PAIR(ny, 16 * p), by, ayn);
        const f32x2 tnz = __builtin_elementwise_fma(YART_PAIR(nz, 16 * p), bz, azn);
        const f32x2 tfx = __builtin_elementwise_fma(YART_PAIR(fx, 16 * p), bx, ax);
        const f32x2 tfy = __builtin_elementwise_fma(YART_PAIR(fy, 16 * p), by, ay);
        const f32x2 tfz = __builtin_elementwise_fma(YART_PAIR(fz, 16 * p), bz, az);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float tNear = fmaxf(fmaxf(tnx[e], tny[e]), fmaxf(tnz[e], 0.0f));
            const float tFar = fminf(fminf(tfx[e], tfy[e]), fminf(tfz[e], bound));
            tn[2 * p + e] = tNear;
            // tNear >= 0 and tFar <= bound <= tmax, neither can be NaN (0 and bound are operands), so this
            // differs from the full interval test only for a box that ends exactly at the origin or begins
            // exactly at tmax: visiting those is harmless, no primitive test can accept there
            h[2 * p + e] = !(tFar < tNear);
        }
    }
#undef YART_PAIR
}

// Which of two leaves of the reference's tree (indices a != b into its flattened, pre-order node array: left child at
// i + 1, right child at `offset`, src/bvhaccel.cpp:215-234) its walk reaches first for a ray of direction d: at their
// lowest common ancestor the left subtree iff d[splitAxis] > 0 (src/bvhaccel.cpp:255-261).
__device__ __forceinline__ uint32_t tiedFirst(const float4* __restrict__ refNodes, uint32_t a, uint32_t b, float dx, float dy, float dz)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    uint32_t i = 0;
    for (int guard = 0; guard < 128; ++guard) {
        const float4 w = refNodes[2 * (size_t)i + 1];              // bounds_max.y, .z, offset, num | axis
        const uint32_t right = __float_as_uint(w.z);
        const uint32_t axis = (__float_as_uint(w.w) >> 8) & 255u;
        if (axis == 3u) break;                                      // a leaf: cannot happen for a != b
        if (hi < right) { i = i + 1; continue; }
        if (lo >= right) { i = right; continue; }
        const float d = axis == 0u ? dx : (axis == 1u ? dy : dz);
        return d > 0.0f ? lo : hi;                                  // lo lies in the left subtree, hi in the right one
    }
    return lo;
}

// The triangle of leaf row `leafRef` gave exactly the t of the triangle held (primTri).  It takes its place iff it passes
// the whole test (include/triaccel.h:169-201 with the bound one ulp above, the near-edge check of the common path) and the
// reference's walk meets it first: the lower slot inside one of its leaves, the leaf tiedFirst names across two.
__device__ __forceinline__ void tieBreak(const WideParams& P, const WRay& r, uint32_t leafRef, float tTri, float& hu, float& hv, int32_t& primTri)
{
    const float4 A = P.units[leafRef >> 2];
    const uint32_t kw = __float_as_uint(A.w);
    const int k = (int)(kw & 3u);
    const uint32_t slot = (kw & kPackSlotMask) >> 2;
    const float dk = sel3(k, r.dx, r.dy, r.dz), du = sel3(k, r.dy, r.dz, r.dx), dv = sel3(k, r.dz, r.dx, r.dy);
    const float ok = sel3(k, r.ox, r.oy, r.oz), ou = sel3(k, r.oy, r.oz, r.ox), ov = sel3(k, r.oz, r.ox, r.oy);
    const float nd = 1.0f / (dk + A.x * du + A.y * dv);
    const float t = (A.z - ok - A.x * ou - A.y * ov) * nd;
    if (!(t == tTri && t > 0.0001f)) return;
    const float4* ta = P.triaccel + (size_t)slot * 3;
    const float4 B = ta[1], C = ta[2];
    const float phu = ou + t * du;
    const float phv = ov + t * dv;
    const float lambda = phu * B.x + phv * B.y + B.z;
    if (lambda < 0.0f) return;
    const float mue = phu * C.x + phv * C.y + C.z;
    if (mue < 0.0f) return;
    if (lambda + mue > 1.0f) return;
    const uint32_t mine = P.leafOfSlot[slot], held = P.leafOfSlot[primTri];
    if ((kw >> kPackBandShift) && fminf(fminf(lambda, mue), 1.0f - (lambda + mue)) < bandOfClass(kw >> kPackBandShift)) {
        const float4* lb = P.refNodes + 2 * (size_t)mine;
        const float4 b0 = lb[0], b1 = lb[1];
        if (!slabRef(b0, b1, r)) return;
    }
    const bool first = mine == held ? (int32_t)slot < primTri : tiedFirst(P.refNodes, mine, held, r.dx, r.dy, r.dz) == mine;
    if (!first) return;
    hv = lambda; hu = mue;
    primTri = (int32_t)slot;
}

#define YART_CSWAP(i, j)                                                   \
    {                                                                      \
        const bool s_ = tn[j] < tn[i];                                     \
        const float tf_ = s_ ? tn[j] : tn[i], tg_ = s_ ? tn[i] : tn[j];    \
        const uint32_t rf_ = s_ ? rf[j] : rf[i], rg_ = s_ ? rf[i] : rf[j]; \
        tn[i] = tf_; tn[j] = tg_; rf[i] = rf_; rf[j] = rg_;                \
    }

// SPH: the tree holds sphere leaves.  A scene of triangles only runs an instance without the sphere slot, the sphere
// test and the sphere side of the bookkeeping.
template <bool ANY, bool COUNT, bool PERSIST, bool SPH>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 8))) traceWide(WideParams P)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    uint32_t ovf[kOvfStack];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t laneBit = 1ull << lane;

    WRay r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.tmax = 0.0f;
    uint32_t cur = kIdle;
    uint32_t pend = kIdle;                         // a triangle leaf set aside while the lane keeps descending
    uint32_t pendS = kIdle;                        // a sphere leaf set aside likewise: the sphere test is double-precision code
                                                   // of its own, run only once enough lanes hold one (sphereMin)
    int sp = 0;
    bool live = false;
    uint32_t rayId = 0;
    float tTri = 0.0f, hu = 0.0f, hv = 0.0f;      // best triangle
    int32_t primTri = -1;
    float F = 0.0f;                                // best sphere: float t, candidates bookkeeping
    int32_t minS = -1, maxL = -1;
    uint32_t occ = 0;
    uint32_t cNodes = 0, cTri = 0, cSph = 0, cStage2 = 0;
#ifdef YART_STEP_STATS
    uint32_t stNodeSteps = 0, stNodeLanes = 0, stLeafSteps = 0, stLeafLanes = 0, stIters = 0, stLive = 0, stSphSteps = 0, stSphLanes = 0;
#endif

    uint32_t chunkNext = 0, chunkEnd = 0;
    bool exhausted = false;
    bool firstPass = true;
    const uint32_t homeXcd = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID
    uint32_t rangesDone = 0;
    const int refillMin = (int)P.refillMin;
    const int nodeMin = (int)P.nodeMin;
    const int sphereMin = (int)P.sphereMin;
    const uint32_t kChunk = P.chunk;             // rays a wave takes from a range counter at a time
    const uint64_t capMask = (P.laneCap == 0 || P.laneCap >= 64) ? ~0ull : ((1ull << P.laneCap) - 1ull);

    auto push = [&](uint32_t v) {
        if (sp < kLdsStack) ldsStack[sp * kBlock + tid] = v;
        else ovf[sp - kLdsStack] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return kIdle;
        --sp;
        if (__builtin_expect(sp < kLdsStack, 1)) return ldsStack[sp * kBlock + tid];
        asm volatile("; deep stack entry" ::: "memory");      // keeps the two paths apart: a ds_read here, a scratch
        return ovf[sp - kLdsStack];                           // load there, instead of one flat load for both
    };

    // A ray this kernel does not answer: the fix-up launches on the reference tree do, taking the listed rays only; a
    // batch with more such rays than the list holds is scanned whole for the marker left in the ray's output (no
    // atomic once the list is full).
    auto deferRay = [&](uint32_t id) {
        uint32_t c = __hip_atomic_load(P.deferFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < P.deferCap) c = atomicAdd(P.deferFlag, 1u);
        if (c < P.deferCap) P.deferList[c] = id;
        else P.deferFlag[1] = 1u;
        if (ANY) P.occluded[id] = kUnresolvedOcc;
        else P.hits[id] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(kUnresolvedPrim));
    };

    auto startRay = [&](uint32_t slot) {
        rayId = P.perm ? P.perm[slot] : slot;
        const float4 a = P.rays[(size_t)rayId * 2 + 0];
        const float4 b = P.rays[(size_t)rayId * 2 + 1];
        if (!(b.w > 0.0001f) && b.w == b.w) {
            // maxT <= 1e-4: no test can accept (`info->t > t && t > 0.0001f`, include/triaccel.h:181; `t > eps && t <
            // hitInfo->t`, include/sphere.h:78,87), whatever else the ray holds: a miss.  The wavefront driver's idle
            // slots carry such rays (wavefront.hip: deadRay).
            if (ANY) P.occluded[rayId] = 0;
            else P.hits[rayId] = make_float4(b.w, 0.0f, 0.0f, __int_as_float(-1));
            if (COUNT) { uint32_t* c = P.counts + (size_t)rayId * 4; c[0] = c[1] = c[2] = c[3] = 0; }
            return;
        }
        if (!regularRay(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w) || !coverHolds(P.cover, a.x, a.y, a.z)) {
            deferRay(rayId);
            return;
        }
        r.ox = a.x; r.oy = a.y; r.oz = a.z;
        r.dx = b.x; r.dy = b.y; r.dz = b.z; r.tmax = b.w;
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
        // slab4 would see infinite planes of the wrong sign if A = (origin_node - o) / d or 255 B overflowed
        if (!(fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz)) * P.cullReach < __builtin_inff())) { deferRay(rayId); return; }
        tTri = r.tmax; hu = 0.0f; hv = 0.0f; primTri = -1;
        F = __builtin_inff(); minS = -1; maxL = -1; occ = 0;
        cNodes = 0; cTri = 0; cSph = 0; cStage2 = 0;
        sp = 0;
        live = true;
        pend = kIdle;
        pendS = kIdle;
        cur = 0u;                                  // root node
    };

    for (;;) {
        // ------------------------------------------------------------ fetch work
        if (PERSIST) {
            const uint64_t idleMask = __ballot(!live) & capMask;
            const int nIdle = __popcll(idleMask);
            if (!exhausted && (nIdle >= refillMin || idleMask == capMask)) {
                while (chunkNext == chunkEnd && !exhausted) {
                    const uint32_t range = (homeXcd + rangesDone) & 7u;
                    const uint32_t rBegin = rangeBegin(P.n, range, kChunk), rEnd = rangeBegin(P.n, range + 1, kChunk);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.workCounter + range * 8, kChunk);
                    base = __builtin_amdgcn_readfirstlane(base) + rBegin;
                    if (base >= rEnd) { if (++rangesDone == 8) exhausted = true; }
                    else { chunkNext = base; chunkEnd = (rEnd - base < kChunk) ? rEnd : base + kChunk; }
                }
                if (!exhausted) {
                    const uint32_t avail = chunkEnd - chunkNext;
                    const uint32_t rank = __popcll(idleMask & (laneBit - 1));
                    if ((idleMask & laneBit) && rank < avail) startRay(chunkNext + rank);     // idle AND below the lane cap
                    chunkNext += ((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail;
                }
            }
            if (__ballot(live) == 0) {
                if (exhausted) break;
                continue;
            }
        } else {
            if (firstPass) {
                const uint32_t slot = blockIdx.x * kBlock + tid;
                if (slot < P.n) startRay(slot);
                firstPass = false;
            }
            if (__ballot(live) == 0) break;
        }

        // ------------------------------------------------------------ descend
        // Speculative descent: a lane that reaches a leaf sets it aside (one of each kind per lane) and goes on
        // with the next node of its stack instead of idling until the wave turns to the leaves; the
        // leaf is tested in the next leaf phase.  Nodes visited meanwhile are tested against a bound
        // that does not yet know that leaf's hit -- a little extra work for lanes that stay busy.
        // A triangle leaf and a sphere leaf are set aside in slots of their own: a lane that holds one of a kind is
        // stopped only by a second leaf of the same kind (with one slot for both, 5.01 -> 4.78 ms on C4).  The sphere
        // leaves wait until sphereMin lanes hold one, unless a lane cannot go on without (its next node is another
        // sphere leaf) or the wave has no interior node left; the threshold matters little (1..24: within 0.3 %), and
        // letting lanes wait for it without that exception costs more than the fuller sphere steps save.
        bool sphNow = false;
        for (;;) {
            if ((cur & kPackLeaf) && cur < kNeedPop) {              // a leaf (the two markers are the largest values)
                if (SPH && (cur & kPackSphere)) { if (pendS == kIdle) { pendS = cur; cur = kNeedPop; } }
                else if (pend == kIdle) { pend = cur; cur = kNeedPop; }
            }
            if (cur == kNeedPop) cur = pop();
            const uint64_t inner = __ballot((cur & kPackLeaf) == 0);
            if (inner == 0) { sphNow = true; break; }
            if (__popcll(inner) < nodeMin) {
                const uint64_t sphHeld = SPH ? __ballot(pendS != kIdle) : 0ull;
                sphNow = SPH && sphHeld != 0 && (__popcll(sphHeld) >= sphereMin ||
                                          __ballot((cur & kPackLeaf) && cur < kNeedPop && (cur & kPackSphere)) != 0);
                if (sphNow || __ballot(pend != kIdle) != 0) break;
            }
#ifdef YART_STEP_STATS
            ++stNodeSteps; stNodeLanes += __popcll(inner);
#endif
            if ((cur & kPackLeaf) == 0) {
                // three 16-byte requests per visit: the child references are one block address and four byte offsets
                const float4* p = P.units + (cur >> 2);
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                // Wave priority: low for the box arithmetic that follows, raised again once the next node is known
                // (below) -- a wave on its way to its next loads goes ahead of waves that are computing, so its memory
                // latency starts earlier.  C4 4.77 -> 4.68 ms, s = 0.01 7.81 -> 7.50 ms; the level (1..3) does not
                // matter, and lowering it for the primitive tests as well gives half of that back.
                __builtin_amdgcn_s_setprio(0);
                // culled against the best hit so far, a relative hair later (kCullSlack); slab4 allows for its own rounding
                const float bound = (ANY ? r.tmax : (SPH ? fminf(tTri, F) : tTri)) * kCullSlack;
                float tn[4];
                uint32_t rf[4];
                const uint32_t blk = __float_as_uint(q2.z), offs = __float_as_uint(q2.w);
                rf[0] = blk + (offs & 255u); rf[1] = blk + ((offs >> 8) & 255u);
                rf[2] = blk + ((offs >> 16) & 255u); rf[3] = blk + (offs >> 24);
                bool h[4];
                slab4(q0, q1, q2, r, bound, P.cullMargin, tn, h);
                // an unused slot needs no test of its own: its box is inverted on every axis (qlo 255, qhi 0:
                // wide_build.cpp clear()), so it exits before it enters
                if (COUNT) cNodes += 4;
                {
                    // nearest child next, the others pushed far to near.  A hit child has tn < tmax, so
                    // +inf marks the missed ones and sorts them last.
                    const float inf = __builtin_inff();
#pragma unroll
                    for (int c = 0; c < 4; ++c) tn[c] = h[c] ? tn[c] : inf;
                    YART_CSWAP(0, 1) YART_CSWAP(2, 3) YART_CSWAP(0, 2) YART_CSWAP(1, 3) YART_CSWAP(1, 2)
                    cur = (tn[0] < inf) ? rf[0] : kNeedPop;
                    __builtin_amdgcn_s_setprio(2);              // before the pushes: 0.6 % over raising it after them
                    if (tn[3] < inf) push(rf[3]);
                    if (tn[2] < inf) push(rf[2]);
                    if (tn[1] < inf) push(rf[1]);
                }
            }
        }

        // ------------------------------------------------------------ leaf
#ifdef YART_STEP_STATS
        { const uint64_t pm = __ballot(pend != kIdle); if (pm) { ++stLeafSteps; stLeafLanes += __popcll(pm); } ++stIters; stLive += __popcll(__ballot(live));
          const uint64_t sm = __ballot(sphNow && pendS != kIdle); if (sm) { ++stSphSteps; stSphLanes += __popcll(sm); } }
#endif
        bool stop = false;
        uint32_t tieRow = kIdle;
        if (pend != kIdle) {
            const uint32_t leaf = pend;
            pend = kIdle;
            const float4* row = P.units + (leaf >> 2);          // one primitive per leaf
            {
                do {
                    const float4 A = row[0];
                    if (COUNT) ++cTri;
                    // include/triaccel.h:169-201
                    const uint32_t kw = __float_as_uint(A.w);
                    const int k = (int)(kw & 3u);
                    const float dk = sel3(k, r.dx, r.dy, r.dz), du = sel3(k, r.dy, r.dz, r.dx),
                                dv = sel3(k, r.dz, r.dx, r.dy);
                    const float ok = sel3(k, r.ox, r.oy, r.oz), ou = sel3(k, r.oy, r.oz, r.ox),
                                ov = sel3(k, r.oz, r.ox, r.oy);
                    const float nd = 1.0f / (dk + A.x * du + A.y * dv);
                    const float t = (A.z - ok - A.x * ou - A.y * ov) * nd;
                    const float bound = ANY ? r.tmax : (SPH ? fminf(tTri, F) : tTri);
                    const uint32_t slot = (kw & kPackSlotMask) >> 2;
                    if (!(bound > t && t > 0.0001f)) {
                        // an exact tie with the triangle held so far is looked at after the common path (below)
                        if (!ANY && t == tTri && primTri >= 0 && (int32_t)slot != primTri) tieRow = leaf;
                        break;
                    }
                    // second stage: rows b and c of the reference's TriAccel record of this slot
                    const float4* ta = P.triaccel + (size_t)slot * 3;
                    const float4 B = ta[1], C = ta[2];
                    if (COUNT) ++cStage2;
                    const float phu = ou + t * du;
                    const float phv = ov + t * dv;
                    const float lambda = phu * B.x + phv * B.y + B.z;
                    if (lambda < 0.0f) break;
                    const float mue = phu * C.x + phv * C.y + C.z;
                    if (mue < 0.0f) break;
                    if (lambda + mue > 1.0f) break;
                    // Accepted by the reference's test.  Within the triangle's error band of an edge (cover.h) the hit
                    // point may lie outside the triangle, and then outside the box of the triangle's leaf in the
                    // reference's tree: the reference finds this hit only if the ray passes that box.
                    if (kw >> kPackBandShift) {
                        if (fminf(fminf(lambda, mue), 1.0f - (lambda + mue)) < bandOfClass(kw >> kPackBandShift)) {
                            const float4* lb = P.refNodes + 2 * (size_t)P.leafOfSlot[slot];
                            const float4 b0 = lb[0], b1 = lb[1];
                            if (!slabRef(b0, b1, r)) break;
                        }
                    }
                    if (ANY) { occ = 1; stop = true; break; }
                    tTri = t; hv = lambda; hu = mue;       // include/triaccel.h:194-196
                    primTri = (int32_t)slot;
                } while (false);
            }
        }
        if (!ANY && __ballot(tieRow != kIdle) != 0) {
            // Rare: a second triangle at bit-identical t.  The reference keeps the FIRST one its walk finds (`info->t > t`
            // is strict, include/triaccel.h:181); so the held one stays unless this one passes the whole test and comes
            // earlier in the reference's order.  The test is run again from the rows: this path costs the common one nothing.
            if (tieRow != kIdle) tieBreak(P, r, tieRow, tTri, hu, hv, primTri);
        }
        if (SPH && sphNow && pendS != kIdle && !stop) {
            const float4* row = P.units + (pendS >> 2);
            pendS = kIdle;
            {
                do {
                    const float4 S = row[0];
                    if (COUNT) ++cSph;
                    // include/sphere.h:63-97
                    const float opx = S.x - r.ox, opy = S.y - r.oy, opz = S.z - r.oz;
                    const double b = (double)(opx * r.dx + opy * r.dy + opz * r.dz);
                    double det = b * b - (double)(opx * opx + opy * opy + opz * opz) + (double)(S.w * S.w);
                    if (det < 0) break;
                    det = sqrt(det);
                    const double eps = (double)1e-4f;
                    double t = b - det;
                    if (!(t > eps)) t = b + det;
                    if (!(t > eps && t < (double)r.tmax)) break;
                    if (ANY) { occ = 1; stop = true; break; }
                    const float f = (float)t;
                    // a sphere whose float t lies beyond the best triangle cannot be the result
                    // (the triangle is strictly closer); beyond the best sphere neither
                    if (f > fminf(tTri, F)) break;
                    const int32_t idx = (int32_t)__float_as_uint(row[1].x);
                    if (COUNT) ++cStage2;
                    const bool below = t < (double)f;
                    if (minS < 0 || f < F) { F = f; minS = idx; maxL = below ? idx : -1; }
                    else {
                        minS = (idx < minS) ? idx : minS;
                        if (below && idx > maxL) maxL = idx;
                    }
                } while (false);
            }
        }
        if (stop) { cur = kIdle; sp = 0; pend = kIdle; pendS = kIdle; }

        // ------------------------------------------------------------ retire
        if (cur == kIdle && pend == kIdle && (!SPH || pendS == kIdle) && live) {
            live = false;
            if (ANY) {
                P.occluded[rayId] = (uint8_t)occ;
            } else {
                // include/scene.h:142-167: spheres first, a triangle replaces the sphere only if strictly closer
                const bool sphere = SPH && (minS >= 0) && !(tTri < F);
                float t = sphere ? F : tTri;
                int32_t prim = sphere ? -((minS > maxL ? minS : maxL) + 2) : primTri;
                if (!(t < r.tmax)) prim = -1;
                if (sphere || prim == -1) { hu = 0.0f; hv = 0.0f; }
                {
                    // written once, read by the caller: a streaming store (C4 -1.2 %; the loads are another matter)
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 h4 = { t, hu, hv, __int_as_float(prim) };
                    __builtin_nontemporal_store(h4, reinterpret_cast<f32x4*>(P.hits + rayId));
                }
            }
            if (COUNT) {
                uint32_t* c = P.counts + (size_t)rayId * 4;
                c[0] = cNodes; c[1] = cTri; c[2] = cStage2; c[3] = cSph;
            }
        }
    }
#ifdef YART_STEP_STATS
    // experiment builds only (make EXTRA=-DYART_STEP_STATS): how full the waves are in the two phases
    if (lane == 0) {
        atomicAdd(P.workCounter + 9, stNodeSteps); atomicAdd(P.workCounter + 10, stNodeLanes >> 6);
        atomicAdd(P.workCounter + 11, stLeafSteps); atomicAdd(P.workCounter + 12, stLeafLanes >> 6);
        atomicAdd(P.workCounter + 13, stIters); atomicAdd(P.workCounter + 14, stLive >> 6);
        atomicAdd(P.workCounter + 17, stSphSteps); atomicAdd(P.workCounter + 18, stSphLanes);
    }
#endif
}

template <bool ANY, bool COUNT, bool SPH>
hipError_t launchW(const WideParams& P, bool persistent, int numCUs, hipStream_t stream)
{
    if (P.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(P.workCounter, 0, 256, stream);      // work counters + defer flag
    if (e != hipSuccess) return e;
    if (persistent) {
        int perCU = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, traceWide<ANY, COUNT, true, SPH>, kBlock, 0);
        if (e != hipSuccess) return e;
        if (perCU < 1) perCU = 1;
        if (perCU > 8) perCU = 8;
        // Waves resident per SIMD (= blocks per CU).  A wave's step takes the longer the more waves share its SIMD,
        // and the launch ends with the longest rays of the last chunks, a drain at ever lower occupancy whose length
        // is a fixed number of steps: the smaller the batch, the more of its time is that drain, and the fewer,
        // faster waves pay (profiles/r02_exp_waves_per_simd.txt: 2^20 rays 4 waves -13 %, 2^22 rays 6 waves -8 %).
        int want = P.n >= (1u << 23) ? 7 : P.n >= (1u << 22) ? 6 : P.n >= (1u << 21) ? 5 : P.n >= (1u << 20) ? 4 : P.n >= (1u << 19) ? 3 : 2;
        if (P.maxBlocksPerCU) want = (int)P.maxBlocksPerCU;
        if (perCU > want) perCU = want;
        uint32_t grid = (uint32_t)numCUs * (uint32_t)perCU;
        // A batch of a few thousand rays is latency bound -- its time is that of the wave whose rays take the most steps --
        // so it is spread over more waves that fill only 16 or 32 of their lanes: 2^12 rays 0.171 -> 0.136 ms, 2^14 rays
        // 0.258 -> 0.237 ms; from 2^16 rays up full waves are faster (YART_WIDE_LANE_CAP overrides, 64 = never).
        WideParams Q = P;
        if (Q.laneCap == 0) Q.laneCap = P.n <= 8192u ? 16u : P.n <= 32768u ? 32u : 64u;
        if (Q.laneCap > 0 && Q.laneCap < 64) {
            Q.chunk = Q.laneCap;
            Q.refillMin = Q.refillMin * Q.laneCap / 64 ? Q.refillMin * Q.laneCap / 64 : 1;
            Q.nodeMin = Q.nodeMin * Q.laneCap / 64 ? Q.nodeMin * Q.laneCap / 64 : 1;
            Q.sphereMin = Q.sphereMin * Q.laneCap / 64 ? Q.sphereMin * Q.laneCap / 64 : 1;
        }
        const uint32_t need = (Q.n + Q.chunk - 1) / Q.chunk;
        const uint32_t needBlocks = (need + 3) / 4;
        if (grid > needBlocks) grid = needBlocks;
        hipLaunchKernelGGL((traceWide<ANY, COUNT, true, SPH>), dim3(grid), dim3(kBlock), 0, stream, Q);
#ifdef YART_STEP_STATS
        uint32_t st[12];
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(st + 2, P.workCounter + 9, 10 * 4, hipMemcpyDeviceToHost) == hipSuccess && P.n > 1000000)
            fprintf(stderr, "step stats (%u rays, any=%d): node steps %u, mean lanes %.1f; triangle leaf steps %u, mean lanes %.1f; sphere leaf steps %u, mean lanes %.1f; outer iterations %u, mean live lanes %.1f\n",
                    P.n, (int)ANY, st[2], 64.0 * st[3] / (st[2] ? st[2] : 1), st[4], 64.0 * st[5] / (st[4] ? st[4] : 1), st[10], (double)st[11] / (st[10] ? st[10] : 1),
                    st[6], 64.0 * st[7] / (st[6] ? st[6] : 1));
#endif
    } else {
        const uint32_t grid = (P.n + kBlock - 1) / kBlock;
        hipLaunchKernelGGL((traceWide<ANY, COUNT, false, SPH>), dim3(grid), dim3(kBlock), 0, stream, P);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// tracePool -- EXPERIMENT (YART_WIDE_POOL=S, closest hit only): phase-homogeneous wave steps.  The rays of a workgroup
// live in a pool of S slots per lane CLASS (slot index = s * 64 + lane: every LDS array is indexed so that lane L only
// ever touches bank L -- conflict-free -- and a ray never changes its lane, only the wave that works on it).  Per class
// four bit masks say which slots wait for a node step, a triangle test, a sphere test, or are free.  A wave looks at the
// masks, picks the phase most of its lanes can take part in, parks the rays it holds in another phase (two LDS writes and
// an atomic or) and claims waiting ones of the chosen phase (atomic and), so a node step runs with rays at interior
// nodes in (nearly) all 64 lanes and a leaf step with leaves in most.  No speculative descent: a ray is in one phase.
// What it costs: the state of a claimed ray comes out of LDS (11 reads + three reciprocals), LDS holds the whole ray
// state (27 dwords per slot with a 10-entry stack window, deeper entries in a global overflow area), so fewer rays are
// in flight per CU, and the claim / park code is issued nearly every time the wave comes back to the top.
// Results are the default kernel's bit for bit (order independence).  Numbers: DESIGN.md 3.9.
enum { PF_OX, PF_OY, PF_OZ, PF_DX, PF_DY, PF_DZ, PF_TMAX, PF_TTRI, PF_HU, PF_HV, PF_PRIM, PF_CUR, PF_SP, PF_RAY, PF_F, PF_MINS, PF_MAXL, PF_COUNT };
constexpr int kPoolBlockMax = 512;
enum { PH_NONE = 0, PH_NODE = 1, PH_TRI = 2, PH_SPH = 3 };

template <bool SPH>
__global__ void __launch_bounds__(kPoolBlockMax) tracePool(WideParams P, uint32_t* __restrict__ ovfAll)
{
    extern __shared__ uint32_t pl[];
    const uint32_t S = P.poolSlots, W = P.poolWindow, NS = S * 64u;
    uint32_t* const masks = pl + PF_COUNT * NS;          // [node, tri, sph, free][64]
    uint32_t* const stack = masks + 256;                 // [W][NS]
    uint32_t* const ovf = ovfAll + (size_t)blockIdx.x * NS * kWideStack;
    float* const pf = reinterpret_cast<float*>(pl);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t laneBit = 1ull << lane;
    if (tid < 64) { masks[tid] = 0; masks[64 + tid] = 0; masks[128 + tid] = 0; masks[192 + tid] = (S >= 32 ? 0xFFFFFFFFu : ((1u << S) - 1u)); }
    __syncthreads();

    WRay r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.tmax = 0.0f;
    bool have = false;
    uint32_t si = (uint32_t)lane;          // slot index of the ray held
    uint32_t cur = kIdle;
    int sp = 0;
    float bnd = 0.0f;                      // min(best triangle t, best sphere float t) of the ray held

    uint32_t chunkNext = 0, chunkEnd = 0;
    bool exhausted = false;
    const uint32_t homeXcd = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    uint32_t rangesDone = 0;
    const uint32_t kChunk = P.chunk;
    const int refillMin = (int)P.refillMin, nodeMin = (int)P.nodeMin, nodeKeep = (int)P.poolKeep;
    const int triMin = (int)P.poolTriMin, sphMin = (int)P.poolSphMin;
#ifdef YART_STEP_STATS
    uint32_t stNodeSteps = 0, stNodeLanes = 0, stLeafSteps = 0, stLeafLanes = 0, stIters = 0, stSphSteps = 0, stSphLanes = 0, stStart = 0;
#endif

    auto push = [&](uint32_t v) {
        if ((uint32_t)sp < W) stack[(uint32_t)sp * NS + si] = v;
        else __hip_atomic_store(ovf + (size_t)si * kWideStack + (sp - (int)W), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return kIdle;
        --sp;
        if (__builtin_expect((uint32_t)sp < W, 1)) return stack[(uint32_t)sp * NS + si];
        // written by whichever wave held the ray then: read past this CU's L1
        return __hip_atomic_load(ovf + (size_t)si * kWideStack + (sp - (int)W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto phaseOf = [&](uint32_t c) -> int {
        if (c == kIdle) return PH_NONE;
        if ((c & kPackLeaf) == 0) return PH_NODE;
        return (SPH && (c & kPackSphere)) ? PH_SPH : PH_TRI;
    };
    auto park = [&](int ph) {
        pl[PF_CUR * NS + si] = cur;
        pl[PF_SP * NS + si] = (uint32_t)sp;
        __hip_atomic_fetch_or(masks + (ph - 1) * 64 + lane, 1u << (si >> 6), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        have = false;
    };
    auto claim = [&](int ph, uint32_t m) {
        const uint32_t b = (uint32_t)__builtin_ctz(m);
        const uint32_t old = __hip_atomic_fetch_and(masks + (ph - 1) * 64 + lane, ~(1u << b), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!(old & (1u << b))) return;                  // another wave was faster
        si = b * 64u + (uint32_t)lane;
        have = true;
        r.ox = pf[PF_OX * NS + si]; r.oy = pf[PF_OY * NS + si]; r.oz = pf[PF_OZ * NS + si];
        r.dx = pf[PF_DX * NS + si]; r.dy = pf[PF_DY * NS + si]; r.dz = pf[PF_DZ * NS + si];
        r.tmax = pf[PF_TMAX * NS + si];
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
        cur = pl[PF_CUR * NS + si];
        sp = (int)pl[PF_SP * NS + si];
        const float tt = pf[PF_TTRI * NS + si];
        bnd = SPH ? fminf(tt, pf[PF_F * NS + si]) : tt;
    };
    auto deferRay = [&](uint32_t id) {
        uint32_t c = __hip_atomic_load(P.deferFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < P.deferCap) c = atomicAdd(P.deferFlag, 1u);
        if (c < P.deferCap) P.deferList[c] = id;
        else P.deferFlag[1] = 1u;
        P.hits[id] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(kUnresolvedPrim));
    };
    // `si` names a free slot this lane has claimed; false: the ray is answered or deferred, the slot stays free
    auto startRay = [&](uint32_t slot) -> bool {
        const uint32_t rayId = P.perm ? P.perm[slot] : slot;
        const float4 a = P.rays[(size_t)rayId * 2 + 0];
        const float4 b = P.rays[(size_t)rayId * 2 + 1];
        if (!(b.w > 0.0001f) && b.w == b.w) {
            P.hits[rayId] = make_float4(b.w, 0.0f, 0.0f, __int_as_float(-1));
            return false;
        }
        if (!regularRay(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w) || !coverHolds(P.cover, a.x, a.y, a.z)) { deferRay(rayId); return false; }
        r.ox = a.x; r.oy = a.y; r.oz = a.z;
        r.dx = b.x; r.dy = b.y; r.dz = b.z; r.tmax = b.w;
        r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
        if (!(fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz)) * P.cullReach < __builtin_inff())) { deferRay(rayId); return false; }
        pf[PF_OX * NS + si] = r.ox; pf[PF_OY * NS + si] = r.oy; pf[PF_OZ * NS + si] = r.oz;
        pf[PF_DX * NS + si] = r.dx; pf[PF_DY * NS + si] = r.dy; pf[PF_DZ * NS + si] = r.dz;
        pf[PF_TMAX * NS + si] = r.tmax; pf[PF_TTRI * NS + si] = r.tmax;
        pf[PF_HU * NS + si] = 0.0f; pf[PF_HV * NS + si] = 0.0f;
        pl[PF_PRIM * NS + si] = 0xFFFFFFFFu;
        pl[PF_RAY * NS + si] = rayId;
        if (SPH) { pf[PF_F * NS + si] = __builtin_inff(); pl[PF_MINS * NS + si] = 0xFFFFFFFFu; pl[PF_MAXL * NS + si] = 0xFFFFFFFFu; }
        bnd = r.tmax;
        sp = 0;
        cur = 0u;
        return true;
    };
    auto retire = [&]() {
        const float tTri = pf[PF_TTRI * NS + si];
        float hu = pf[PF_HU * NS + si], hv = pf[PF_HV * NS + si];
        const int32_t primTri = (int32_t)pl[PF_PRIM * NS + si];
        const uint32_t rayId = pl[PF_RAY * NS + si];
        float F = __builtin_inff();
        int32_t minS = -1, maxL = -1;
        if (SPH) { F = pf[PF_F * NS + si]; minS = (int32_t)pl[PF_MINS * NS + si]; maxL = (int32_t)pl[PF_MAXL * NS + si]; }
        const bool sphere = SPH && (minS >= 0) && !(tTri < F);
        float t = sphere ? F : tTri;
        int32_t prim = sphere ? -((minS > maxL ? minS : maxL) + 2) : primTri;
        if (!(t < r.tmax)) prim = -1;
        if (sphere || prim == -1) { hu = 0.0f; hv = 0.0f; }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 h4 = { t, hu, hv, __int_as_float(prim) };
        __builtin_nontemporal_store(h4, reinterpret_cast<f32x4*>(P.hits + rayId));
        __hip_atomic_fetch_or(masks + 192 + lane, 1u << (si >> 6), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        have = false;
    };

    for (;;) {
        if (have && cur == kIdle) retire();
        const uint32_t mN = masks[lane], mT = masks[64 + lane], mS = SPH ? masks[128 + lane] : 0u, mF = masks[192 + lane];
        const int ph = have ? phaseOf(cur) : PH_NONE;
        const int nN = __popcll(__ballot(ph == PH_NODE || mN != 0));
        const int nT = __popcll(__ballot(ph == PH_TRI || mT != 0));
        const int nS = SPH ? __popcll(__ballot(ph == PH_SPH || mS != 0)) : 0;
        const int nF = exhausted ? 0 : __popcll(__ballot(mF != 0));
        int phase;           // 0 = start rays
        if (nF >= refillMin) phase = 0;
        else if (nN >= nodeMin) phase = PH_NODE;
        else if (SPH && nS >= sphMin) phase = PH_SPH;
        else if (nT >= triMin) phase = PH_TRI;
        else if (nN > 0 && nN >= nT && nN >= nS) phase = PH_NODE;
        else if (nT > 0 && nT >= nS) phase = PH_TRI;
        else if (nS > 0) phase = PH_SPH;
        else if (nF > 0) phase = 0;
        else if (__ballot(have) == 0 && exhausted) break;
        else continue;       // every slot of every class is in another wave's hands for the moment
#ifdef YART_STEP_STATS
        ++stIters;
#endif

        if (phase == 0) {
            // ---------------------------------------------------------------- start rays in free slots
            bool mine = false;
            if (mF != 0) {
                if (have) park(ph);
                const uint32_t b = (uint32_t)__builtin_ctz(mF);
                const uint32_t old = __hip_atomic_fetch_and(masks + 192 + lane, ~(1u << b), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old & (1u << b)) { mine = true; si = b * 64u + (uint32_t)lane; }
            }
            const uint64_t takers = __ballot(mine);
            const int nTake = __popcll(takers);
            while (chunkNext == chunkEnd && !exhausted) {
                const uint32_t range = (homeXcd + rangesDone) & 7u;
                const uint32_t rBegin = rangeBegin(P.n, range, kChunk), rEnd = rangeBegin(P.n, range + 1, kChunk);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.workCounter + range * 8, kChunk);
                base = __builtin_amdgcn_readfirstlane(base) + rBegin;
                if (base >= rEnd) { if (++rangesDone == 8) exhausted = true; }
                else { chunkNext = base; chunkEnd = (rEnd - base < kChunk) ? rEnd : base + kChunk; }
            }
            bool started = false;
            if (!exhausted) {
                const uint32_t avail = chunkEnd - chunkNext;
                const uint32_t rank = __popcll(takers & (laneBit - 1));
                if (mine && rank < avail) started = startRay(chunkNext + rank);
                chunkNext += ((uint32_t)nTake < avail) ? (uint32_t)nTake : avail;
            }
            if (mine) {
                if (started) have = true;
                else __hip_atomic_fetch_or(masks + 192 + lane, 1u << (si >> 6), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#ifdef YART_STEP_STATS
            ++stStart;
#endif
            continue;
        }

        if (have && ph != phase) park(ph);
        if (!have) {
            const uint32_t m = phase == PH_NODE ? mN : (phase == PH_TRI ? mT : mS);
            if (m != 0) claim(phase, m);
        }

        if (phase == PH_NODE) {
            // ---------------------------------------------------------------- node steps
            for (;;) {
                const bool inner = have && (cur & kPackLeaf) == 0;       // kIdle has the leaf bit
                const uint64_t im = __ballot(inner);
                if (im == 0) break;
#ifdef YART_STEP_STATS
                ++stNodeSteps; stNodeLanes += __popcll(im);
#endif
                if (inner) {
                    const float4* p = P.units + (cur >> 2);
                    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                    __builtin_amdgcn_s_setprio(0);
                    const float bound = bnd * kCullSlack;
                    float tn[4];
                    uint32_t rf[4];
                    const uint32_t blk = __float_as_uint(q2.z), offs = __float_as_uint(q2.w);
                    rf[0] = blk + (offs & 255u); rf[1] = blk + ((offs >> 8) & 255u);
                    rf[2] = blk + ((offs >> 16) & 255u); rf[3] = blk + (offs >> 24);
                    bool h[4];
                    slab4(q0, q1, q2, r, bound, P.cullMargin, tn, h);
                    const float inf = __builtin_inff();
#pragma unroll
                    for (int c = 0; c < 4; ++c) tn[c] = h[c] ? tn[c] : inf;
                    YART_CSWAP(0, 1) YART_CSWAP(2, 3) YART_CSWAP(0, 2) YART_CSWAP(1, 3) YART_CSWAP(1, 2)
                    __builtin_amdgcn_s_setprio(2);
                    if (tn[3] < inf) push(rf[3]);
                    if (tn[2] < inf) push(rf[2]);
                    if (tn[1] < inf) push(rf[1]);
                    cur = (tn[0] < inf) ? rf[0] : pop();
                }
                if (__popcll(__ballot(have && (cur & kPackLeaf) == 0)) < nodeKeep) break;
            }
            continue;
        }

        if (phase == PH_TRI) {
            // ---------------------------------------------------------------- one triangle leaf per lane
            const bool act = have && cur != kIdle && (cur & kPackLeaf) && !(SPH && (cur & kPackSphere));
#ifdef YART_STEP_STATS
            { const uint64_t pm = __ballot(act); if (pm) { ++stLeafSteps; stLeafLanes += __popcll(pm); } }
#endif
            uint32_t tieRow = kIdle;
            float tTri = 0.0f, hu = 0.0f, hv = 0.0f;
            int32_t primTri = -1;
            if (act) {
                const uint32_t leaf = cur;
                tTri = pf[PF_TTRI * NS + si];
                primTri = (int32_t)pl[PF_PRIM * NS + si];
                const float4* row = P.units + (leaf >> 2);
                do {
                    const float4 A = row[0];
                    const uint32_t kw = __float_as_uint(A.w);
                    const int k = (int)(kw & 3u);
                    const float dk = sel3(k, r.dx, r.dy, r.dz), du = sel3(k, r.dy, r.dz, r.dx), dv = sel3(k, r.dz, r.dx, r.dy);
                    const float ok = sel3(k, r.ox, r.oy, r.oz), ou = sel3(k, r.oy, r.oz, r.ox), ov = sel3(k, r.oz, r.ox, r.oy);
                    const float nd = 1.0f / (dk + A.x * du + A.y * dv);
                    const float t = (A.z - ok - A.x * ou - A.y * ov) * nd;
                    const uint32_t slot = (kw & kPackSlotMask) >> 2;
                    if (!(bnd > t && t > 0.0001f)) {
                        if (t == tTri && primTri >= 0 && (int32_t)slot != primTri) tieRow = leaf;
                        break;
                    }
                    const float4* ta = P.triaccel + (size_t)slot * 3;
                    const float4 B = ta[1], C = ta[2];
                    const float phu = ou + t * du;
                    const float phv = ov + t * dv;
                    const float lambda = phu * B.x + phv * B.y + B.z;
                    if (lambda < 0.0f) break;
                    const float mue = phu * C.x + phv * C.y + C.z;
                    if (mue < 0.0f) break;
                    if (lambda + mue > 1.0f) break;
                    if (kw >> kPackBandShift) {
                        if (fminf(fminf(lambda, mue), 1.0f - (lambda + mue)) < bandOfClass(kw >> kPackBandShift)) {
                            const float4* lb = P.refNodes + 2 * (size_t)P.leafOfSlot[slot];
                            const float4 b0 = lb[0], b1 = lb[1];
                            if (!slabRef(b0, b1, r)) break;
                        }
                    }
                    tTri = t; bnd = t;
                    pf[PF_TTRI * NS + si] = t; pf[PF_HV * NS + si] = lambda; pf[PF_HU * NS + si] = mue;
                    pl[PF_PRIM * NS + si] = slot;
                } while (false);
            }
            if (__ballot(tieRow != kIdle) != 0) {
                if (tieRow != kIdle) {
                    hu = pf[PF_HU * NS + si]; hv = pf[PF_HV * NS + si];
                    const int32_t before = primTri;
                    tieBreak(P, r, tieRow, tTri, hu, hv, primTri);
                    if (primTri != before) { pf[PF_HU * NS + si] = hu; pf[PF_HV * NS + si] = hv; pl[PF_PRIM * NS + si] = (uint32_t)primTri; }
                }
            }
            if (act) cur = pop();
            continue;
        }

        // -------------------------------------------------------------------- one sphere leaf per lane
        if (SPH) {
            const bool act = have && cur != kIdle && (cur & kPackLeaf) && (cur & kPackSphere);
#ifdef YART_STEP_STATS
            { const uint64_t sm = __ballot(act); if (sm) { ++stSphSteps; stSphLanes += __popcll(sm); } }
#endif
            if (act) {
                const float4* row = P.units + (cur >> 2);
                do {
                    const float4 Sr = row[0];
                    const float opx = Sr.x - r.ox, opy = Sr.y - r.oy, opz = Sr.z - r.oz;
                    const double b = (double)(opx * r.dx + opy * r.dy + opz * r.dz);
                    double det = b * b - (double)(opx * opx + opy * opy + opz * opz) + (double)(Sr.w * Sr.w);
                    if (det < 0) break;
                    det = sqrt(det);
                    const double eps = (double)1e-4f;
                    double t = b - det;
                    if (!(t > eps)) t = b + det;
                    if (!(t > eps && t < (double)r.tmax)) break;
                    const float f = (float)t;
                    if (f > bnd) break;
                    const int32_t idx = (int32_t)__float_as_uint(row[1].x);
                    const bool below = t < (double)f;
                    const float F = pf[PF_F * NS + si];
                    int32_t minS = (int32_t)pl[PF_MINS * NS + si], maxL = (int32_t)pl[PF_MAXL * NS + si];
                    if (minS < 0 || f < F) { pf[PF_F * NS + si] = f; minS = idx; maxL = below ? idx : -1; bnd = f; }
                    else {
                        minS = (idx < minS) ? idx : minS;
                        if (below && idx > maxL) maxL = idx;
                    }
                    pl[PF_MINS * NS + si] = (uint32_t)minS; pl[PF_MAXL * NS + si] = (uint32_t)maxL;
                } while (false);
                cur = pop();
            }
        }
    }
#ifdef YART_STEP_STATS
    if (lane == 0) {
        atomicAdd(P.workCounter + 9, stNodeSteps); atomicAdd(P.workCounter + 10, stNodeLanes >> 6);
        atomicAdd(P.workCounter + 11, stLeafSteps); atomicAdd(P.workCounter + 12, stLeafLanes >> 6);
        atomicAdd(P.workCounter + 13, stIters); atomicAdd(P.workCounter + 14, stStart);
        atomicAdd(P.workCounter + 17, stSphSteps); atomicAdd(P.workCounter + 18, stSphLanes);
    }
#endif
}

template <bool SPH>
hipError_t launchPool(const WideParams& P, int numCUs, hipStream_t stream)
{
    if (P.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(P.workCounter, 0, 256, stream);
    if (e != hipSuccess) return e;
    const uint32_t NS = P.poolSlots * 64u;
    const size_t lds = ((size_t)PF_COUNT * NS + 256 + (size_t)P.poolWindow * NS) * 4;
    const int perCU = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
    const uint32_t grid = (uint32_t)numCUs * (uint32_t)perCU;
    // Experiment: one overflow area per device, sized for the largest launch so far, shared by every launch of this kernel
    // on the device.  So that callers on several threads or streams cannot collide in it, the host side is under a mutex
    // and a launch waits (on the device, by event) for the previous launch of this kernel on the same GPU.
    static std::mutex poolMutex;
    static uint32_t* ovf[16] = {};
    static size_t ovfBytes[16] = {};
    static hipEvent_t lastDone[16] = {};
    std::lock_guard<std::mutex> lock(poolMutex);
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    const size_t need = (size_t)grid * NS * kWideStack * 4;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (need > ovfBytes[dev]) {
        if ((e = hipDeviceSynchronize()) != hipSuccess) return e;      // nothing may still use the old area
        if (ovf[dev]) (void)hipFree(ovf[dev]);
        ovf[dev] = nullptr; ovfBytes[dev] = 0;
        if ((e = hipMalloc(reinterpret_cast<void**>(&ovf[dev]), need)) != hipSuccess) return e;
        ovfBytes[dev] = need;
    }
    if (!lastDone[dev]) { if ((e = hipEventCreateWithFlags(&lastDone[dev], hipEventDisableTiming)) != hipSuccess) return e; }
    else if ((e = hipStreamWaitEvent(stream, lastDone[dev], 0)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(tracePool<SPH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL((tracePool<SPH>), dim3(grid), dim3(P.poolBlock), lds, stream, P, ovf[dev]);
    if ((e = hipEventRecord(lastDone[dev], stream)) != hipSuccess) return e;
#ifdef YART_STEP_STATS
    uint32_t st[12];
    if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(st + 2, P.workCounter + 9, 10 * 4, hipMemcpyDeviceToHost) == hipSuccess && P.n > 1000000)
        fprintf(stderr, "pool step stats (%u rays, S=%u W=%u block=%u x%d per CU, lds %zu): node steps %u, mean lanes %.1f; triangle leaf steps %u, mean lanes %.1f; sphere leaf steps %u, mean lanes %.1f; top-of-loop passes %u, start steps %u\n",
                P.n, P.poolSlots, P.poolWindow, P.poolBlock, perCU, lds, st[2], 64.0 * st[3] / (st[2] ? st[2] : 1), st[4], 64.0 * st[5] / (st[4] ? st[4] : 1), st[10], (double)st[11] / (st[10] ? st[10] : 1), st[6], st[7]);
#endif
    return hipGetLastError();
}

// The leaf of every triangle slot in the reference's flattened node array (the near-edge check and the tie rule above).
__global__ void __launch_bounds__(256) leafOfSlotKernel(const uint4* __restrict__ nodes, uint32_t numNodes, uint32_t* __restrict__ leafOfSlot)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numNodes) return;
    const uint4 w = nodes[(size_t)i * 2 + 1];                    // YartBvhNode: bounds_max.y, .z, offset, num_triangles | split_axis << 8
    if (((w.w >> 8) & 255u) != 3u) return;
    const uint32_t cnt = w.w & 255u;
    for (uint32_t k = 0; k < cnt; ++k) leafOfSlot[w.z + k] = i;
}

} // namespace

hipError_t launchLeafOfSlot(const void* nodes, size_t numNodes, uint32_t* leafOfSlot, hipStream_t stream)
{
    if (numNodes == 0) return hipSuccess;
    const uint32_t n = (uint32_t)numNodes;
    hipLaunchKernelGGL(leafOfSlotKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, static_cast<const uint4*>(nodes), n, leafOfSlot);
    return hipGetLastError();
}

hipError_t launchTraceWide(const WideParams& P, bool any, bool count, bool persistent, int numCUs,
                           hipStream_t stream)
{
    if (P.poolSlots && !any && !count && persistent)
        return P.hasSpheres ? launchPool<true>(P, numCUs, stream) : launchPool<false>(P, numCUs, stream);
    if (P.hasSpheres) {
        if (any) return count ? launchW<true, true, true>(P, persistent, numCUs, stream) : launchW<true, false, true>(P, persistent, numCUs, stream);
        return count ? launchW<false, true, true>(P, persistent, numCUs, stream) : launchW<false, false, true>(P, persistent, numCUs, stream);
    }
    if (any) return count ? launchW<true, true, false>(P, persistent, numCUs, stream) : launchW<true, false, false>(P, persistent, numCUs, stream);
    return count ? launchW<false, true, false>(P, persistent, numCUs, stream) : launchW<false, false, false>(P, persistent, numCUs, stream);
}

} // namespace yart
