// ref_build_dev.hip -- the reference's triangle BVH built ON THE GPU, bit for bit.
//
// What is reproduced: elaic/yart's BvhAccel constructor (src/bvhaccel.cpp:100-151 recursive build, :215-234
// flattening, :309-363 triangle boxes and TriAccel records in leaf order; include/triaccel.h:54-108), i.e. what
// bvh_build.cpp produces on the host.  That builder is a recursion over ranges of one primitive array; every
// step of it has a data-parallel form with the SAME result:
//
//   node box        fold of std::min / std::max over the range in order.  min and max of non-NaN floats do not
//                   depend on the order except for the sign of a zero result, which is the sign of the FIRST
//                   zero in the range: boxes are reduced as ordered integers with +0 == -0, and the few
//                   coordinates that are exactly zero race for the lowest position on the side.
//   split           widest axis, midpoint double(float(hi + lo)) * 0.5, predicate double(centre) < midpoint
//   std::partition  libstdc++'s bidirectional version swaps the i-th element from the left that fails the
//                   predicate with the i-th from the right that passes it, until the two cursors meet.  The
//                   meeting point is begin + (number passing); ranks come from one exclusive scan of the
//                   predicate flags over the whole array, and every position fetches its partner's element.
//   median fallback std::nth_element when one side is empty.  Its permutation is the library's introselect;
//                   the range is brought to the host, run through the same std::nth_element, and written back
//                   (rare: a range of eight or more triangles whose centres all lie on one side of the box middle).
//   numbering       nodes in pre-order (left child = parent + 1, right = parent + 1 + size of the left subtree),
//                   sibling pairs in pre-order of their interior parent, leaves right-child-first (g++ evaluates
//                   the two recursive calls right to left): a leaf's first slot = n - (end of its range).
//
// All ranges of one depth are processed by the same launches (level-synchronous): a kernel boundary is the
// only ordering between a node and its children.  NaN coordinates make the fold order-dependent in a way this
// scheme does not follow; such a scene is handed back to the host builder (useHost).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"
#include "bvh_build.h"
#include "ref_build_dev.h"

namespace yart {
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLeafBelow = 8;              // src/bvhaccel.cpp:57 -- leaf iff count < 8
constexpr uint32_t kSphereLeafBelow = 5;        // the engine's sphere tree: leaves of at most 4 (bvh_build.cpp buildSphereBvh)
constexpr uint32_t kOneSided = 0x80000000u;     // flag in nSplit: the range was split at its median on the host
constexpr uint32_t kFixCap = 1u << 16;          // most ranges / triangles fixed up on the host before giving up

#define YART_HIP(call)                                                                              \
    do {                                                                                            \
        const hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return YART_ERR_HIP; } \
    } while (0)

// monotone float -> uint with the two zeros on one key
__device__ inline uint32_t orderedKey(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float fromKey(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Triangle boxes and centres as the reference computes them (src/bvhaccel.cpp:18-30, :337-339; include/bbox.h:70-78).
// status[0]: a vertex index is out of range; status[1]: a coordinate is NaN
__global__ void __launch_bounds__(256) refPrims(const float* __restrict__ verts, const int32_t* __restrict__ tris, uint32_t nTris, uint64_t nVerts,
                                                float4* __restrict__ primLo, float4* __restrict__ primHi, float4* __restrict__ rec,
                                                uint32_t* __restrict__ seg, uint32_t* __restrict__ status)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nTris) return;
    const int32_t* t = tris + (size_t)i * 5;
    bool ok = true;
    for (int c = 0; c < 3; ++c) ok = ok && t[c] >= 0 && (uint64_t)t[c] < nVerts;
    seg[i] = 0u;
    if (!ok) {
        status[0] = 1u;
        primLo[i] = primHi[i] = make_float4(0, 0, 0, 0);
        rec[i] = make_float4(0, 0, 0, __uint_as_float(i));
        return;
    }
    const float* v0 = verts + (size_t)t[0] * 3;
    float lo[3] = { v0[0], v0[1], v0[2] }, hi[3] = { v0[0], v0[1], v0[2] };
    bool nan = !(v0[0] == v0[0] && v0[1] == v0[1] && v0[2] == v0[2]);
    for (int k = 1; k < 3; ++k) {
        const float* p = verts + (size_t)t[k] * 3;
        for (int a = 0; a < 3; ++a) {
            lo[a] = (p[a] < lo[a]) ? p[a] : lo[a];
            hi[a] = (hi[a] < p[a]) ? p[a] : hi[a];
            nan = nan || !(p[a] == p[a]);
        }
    }
    if (nan) status[1] = 1u;
    primLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    primHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    rec[i] = make_float4((lo[0] + hi[0]) * 0.5f, (lo[1] + hi[1]) * 0.5f, (lo[2] + hi[2]) * 0.5f, __uint_as_float(i));
}

__device__ inline void initNode(uint32_t s, uint32_t b, uint32_t e, uint32_t* nBegin, uint32_t* nEnd, uint32_t* nKeys, uint32_t* nZero)
{
    nBegin[s] = b;
    nEnd[s] = e;
    for (int c = 0; c < 6; ++c) {
        nKeys[(size_t)s * 6 + c] = c < 3 ? 0xFFFFFFFFu : 0u;
        nZero[(size_t)s * 6 + c] = kNone;
    }
}

__global__ void initRoot(uint32_t n, uint32_t* nBegin, uint32_t* nEnd, uint32_t* nKeys, uint32_t* nZero)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) initNode(0u, 0u, n, nBegin, nEnd, nKeys, nZero);
}

// Boxes of the ranges of this level.  Positions of one range are neighbours, so a wave reduces the runs of equal
// range ids it holds by a segmented scan and only the last lane of a run touches memory; waves that lie inside one
// range are combined once more per block (the top levels would otherwise send every wave to the same six words).
__global__ void __launch_bounds__(1024) segBounds(const float4* __restrict__ rec, const uint32_t* __restrict__ seg, uint32_t n,
                                                  const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                                  uint32_t* __restrict__ nKeys, uint32_t* __restrict__ nZero)
{
    __shared__ uint32_t wSeg[16];
    __shared__ uint32_t wKeys[16][6];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s = i < n ? seg[i] : kNone;
    uint32_t k[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    if (s != kNone) {
        const uint32_t id = __float_as_uint(rec[i].w);
        const float4 lo = primLo[id], hi = primHi[id];
        const float v[6] = { lo.x, lo.y, lo.z, hi.x, hi.y, hi.z };
        for (int c = 0; c < 6; ++c) {
            k[c] = orderedKey(v[c]);
            if (v[c] == 0.0f) atomicMin(&nZero[(size_t)s * 6 + c], (i << 1) | (__float_as_uint(v[c]) >> 31));
        }
    }
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t os = __shfl_up(s, o);
        const bool take = lane >= o && os == s;
        for (int c = 0; c < 6; ++c) {
            const uint32_t ov = __shfl_up(k[c], o);
            if (take) k[c] = c < 3 ? min(k[c], ov) : max(k[c], ov);
        }
    }
    const uint32_t nextS = __shfl_down(s, 1);
    const bool tail = lane == 63u || nextS != s;
    const bool uniform = __all(s == __shfl(s, 0)) != 0;
    if (uniform && s != kNone) {
        if (lane == 63u) {
            wSeg[wave] = s;
            for (int c = 0; c < 6; ++c) wKeys[wave][c] = k[c];
        }
    } else {
        if (lane == 63u) wSeg[wave] = kNone;
        if (tail && s != kNone)
            for (int c = 0; c < 6; ++c) {
                if (c < 3) atomicMin(&nKeys[(size_t)s * 6 + c], k[c]);
                else atomicMax(&nKeys[(size_t)s * 6 + c], k[c]);
            }
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const int c = (int)threadIdx.x;
        uint32_t cur = kNone, acc = 0u;
        for (int w = 0; w <= 16; ++w) {
            const uint32_t ws = w < 16 ? wSeg[w] : kNone;
            if (ws != cur) {
                if (cur != kNone) {
                    if (c < 3) atomicMin(&nKeys[(size_t)cur * 6 + c], acc);
                    else atomicMax(&nKeys[(size_t)cur * 6 + c], acc);
                }
                cur = ws;
                if (ws != kNone) acc = wKeys[w][c];
            } else if (ws != kNone) {
                acc = c < 3 ? min(acc, wKeys[w][c]) : max(acc, wKeys[w][c]);
            }
        }
    }
}

// Leaf or split, widest axis (include/bbox.h:52-62) and midpoint (src/bvhaccel.cpp:125-127) of the nodes of this level.
__global__ void __launch_bounds__(256) segPlan(uint32_t first, uint32_t count, const uint32_t* __restrict__ nBegin, const uint32_t* __restrict__ nEnd,
                                               const uint32_t* __restrict__ nKeys, uint32_t* __restrict__ nAxis, double* __restrict__ nMid,
                                               uint32_t leafBelow)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t s = first + t;
    if (nEnd[s] - nBegin[s] < leafBelow) { nAxis[s] = 3u; return; }
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = fromKey(nKeys[(size_t)s * 6 + a]); hi[a] = fromKey(nKeys[(size_t)s * 6 + 3 + a]); }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const uint32_t axis = (dx > dy && dx > dz) ? 0u : ((dy > dz) ? 1u : 2u);
    nAxis[s] = axis;
    nMid[s] = (double)(hi[axis] + lo[axis]) * 0.5;
}

// Predicate flags of the positions in ranges that split; positions of ranges that became leaves retire.
__global__ void __launch_bounds__(256) segFlags(const float4* __restrict__ rec, uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ nAxis,
                                                const double* __restrict__ nMid, uint32_t* __restrict__ leafOf, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > n) return;
    uint32_t f = 0u;
    if (i < n) {
        const uint32_t s = seg[i];
        if (s != kNone) {
            const uint32_t ax = nAxis[s];
            if (ax == 3u) {
                leafOf[i] = s;
                seg[i] = kNone;
            } else {
                const float4 r = rec[i];
                const float c = ax == 0u ? r.x : (ax == 1u ? r.y : r.z);
                f = ((double)c < nMid[s]) ? 1u : 0u;
            }
        }
    }
    flags[i] = f;       // flags[n] = 0: the scan's last word is the total
}

struct OneSided { uint32_t node, begin, end, axis; };

// Split position and children of the nodes of this level.  counters[0]: nodes allocated, counters[1]: one-sided ranges.
__global__ void __launch_bounds__(256) segSplit(uint32_t first, uint32_t count, uint32_t* __restrict__ nBegin, uint32_t* __restrict__ nEnd,
                                                const uint32_t* __restrict__ nAxis, const uint32_t* __restrict__ T, uint32_t* __restrict__ nSplit,
                                                uint32_t* __restrict__ nLeft, uint32_t* __restrict__ nKeys, uint32_t* __restrict__ nZero,
                                                uint32_t* __restrict__ counters, OneSided* __restrict__ oneSided)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t s = first + t;
    const bool split = t < count && nAxis[s] != 3u;
    const uint64_t m = __ballot(split);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0u;
    if (lane == 0u && m) base = atomicAdd(counters, 2u * (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (!split) return;
    const uint32_t b = nBegin[s], e = nEnd[s];
    const uint32_t nTrue = T[e] - T[b];
    uint32_t mid = b + nTrue, mark = 0u;
    if (nTrue == 0u || nTrue == e - b) {                    // src/bvhaccel.cpp:133-141
        mid = b + (e - b) / 2u;
        mark = kOneSided;
        const uint32_t slot = atomicAdd(counters + 1, 1u);
        if (slot < kFixCap) oneSided[slot] = OneSided{ s, b, e, nAxis[s] };
    }
    nSplit[s] = mid | mark;
    const uint32_t child = base + 2u * rank;
    nLeft[s] = child;
    initNode(child, b, mid, nBegin, nEnd, nKeys, nZero);
    initNode(child + 1u, mid, e, nBegin, nEnd, nKeys, nZero);
}

// std::partition's pairing: the k-th failing element left of the split and the k-th passing one right of it,
// counted from the right, change places.
__global__ void __launch_bounds__(256) partners(const uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ nBegin,
                                                const uint32_t* __restrict__ nEnd, const uint32_t* __restrict__ nSplit,
                                                const uint32_t* __restrict__ flags, const uint32_t* __restrict__ T,
                                                uint32_t* __restrict__ failPos, uint32_t* __restrict__ passPos)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg[i];
    if (s == kNone) return;
    const uint32_t sp = nSplit[s];
    if (sp & kOneSided) return;
    const uint32_t b = nBegin[s], e = nEnd[s], f = flags[i];
    if (i < sp) {
        if (!f) failPos[b + (i - b) - (T[i] - T[b])] = i;
    } else if (f) {
        passPos[b + (T[e] - T[i] - 1u)] = i;
    }
}

__global__ void __launch_bounds__(256) applySplit(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t* __restrict__ seg, uint32_t n,
                                                  const uint32_t* __restrict__ nBegin, const uint32_t* __restrict__ nEnd,
                                                  const uint32_t* __restrict__ nSplit, const uint32_t* __restrict__ nLeft,
                                                  const uint32_t* __restrict__ flags, const uint32_t* __restrict__ T,
                                                  const uint32_t* __restrict__ failPos, const uint32_t* __restrict__ passPos)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg[i];
    uint32_t j = i;
    if (s != kNone) {
        const uint32_t sp = nSplit[s], mid = sp & ~kOneSided;
        if (!(sp & kOneSided)) {
            const uint32_t b = nBegin[s], e = nEnd[s], f = flags[i];
            if (i < mid) {
                if (!f) j = passPos[b + (i - b) - (T[i] - T[b])];
            } else if (f) {
                j = failPos[b + (T[e] - T[i] - 1u)];
            }
        }
        seg[i] = nLeft[s] + (i >= mid ? 1u : 0u);
    }
    dst[i] = src[j];
}

// node counts of the subtrees, deepest level first
__global__ void __launch_bounds__(256) subtreeSizes(uint32_t first, uint32_t count, const uint32_t* __restrict__ nAxis, const uint32_t* __restrict__ nLeft,
                                                    uint32_t* __restrict__ size, uint32_t* __restrict__ interior)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t s = first + t;
    if (nAxis[s] == 3u) { size[s] = 1u; interior[s] = 0u; return; }
    const uint32_t l = nLeft[s];
    size[s] = 1u + size[l] + size[l + 1u];
    interior[s] = 1u + interior[l] + interior[l + 1u];
}

// pre-order number of every node and pair number of every interior node (src/bvhaccel.cpp:215-234), top level first
__global__ void __launch_bounds__(256) numberNodes(uint32_t first, uint32_t count, const uint32_t* __restrict__ nAxis, const uint32_t* __restrict__ nLeft,
                                                   const uint32_t* __restrict__ size, const uint32_t* __restrict__ interior,
                                                   uint32_t* __restrict__ pre, uint32_t* __restrict__ pairOf)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint32_t s = first + t;
    if (s == 0u) { pre[0] = 0u; pairOf[0] = 1u; }
    if (nAxis[s] == 3u) return;
    const uint32_t l = nLeft[s], p = s == 0u ? 0u : pre[s], q = s == 0u ? 1u : pairOf[s];
    pre[l] = p + 1u;
    pre[l + 1u] = p + 1u + size[l];
    pairOf[l] = q + 1u;
    pairOf[l + 1u] = q + 1u + interior[l];
}

struct NodeBox { float lo[3], hi[3]; };

__device__ inline NodeBox boxOf(uint32_t s, const uint32_t* nKeys, const uint32_t* nZero)
{
    NodeBox b;
    for (int c = 0; c < 6; ++c) {
        float v = fromKey(nKeys[(size_t)s * 6 + c]);
        const uint32_t z = nZero[(size_t)s * 6 + c];
        if (v == 0.0f && z != kNone && (z & 1u)) v = -0.0f;        // the first zero of the range was a negative one
        (c < 3 ? b.lo[c] : b.hi[c - 3]) = v;
    }
    return b;
}

// first leaf slot of a leaf: the reference numbers leaves right child first (n - end of the range), the sphere tree
// left to right (begin of the range)
__device__ inline uint32_t leafSlot(uint32_t s, uint32_t n, const uint32_t* nBegin, const uint32_t* nEnd, bool leftToRight)
{
    return leftToRight ? nBegin[s] : n - nEnd[s];
}

__device__ inline DevNodeRec recordOf(uint32_t s, uint32_t n, const uint32_t* nBegin, const uint32_t* nEnd, const uint32_t* nAxis,
                                      const uint32_t* nKeys, const uint32_t* nZero, const uint32_t* pairOf, bool leftToRight)
{
    DevNodeRec r;
    const NodeBox b = boxOf(s, nKeys, nZero);
    for (int a = 0; a < 3; ++a) { r.lo[a] = b.lo[a]; r.hi[a] = b.hi[a]; }
    const uint32_t axis = nAxis[s];
    r.split_axis = (uint8_t)axis;
    r.pad[0] = r.pad[1] = 0;
    if (axis == 3u) {
        const uint32_t cnt = nEnd[s] - nBegin[s];
        r.num_prims = (uint8_t)cnt;
        r.ref = kLeafBit | (leafSlot(s, n, nBegin, nEnd, leftToRight) << 3) | cnt;
    } else {
        r.num_prims = 0;
        r.ref = (axis << 29) | pairOf[s];
    }
    return r;
}

// The flattened node array in the reference's order and the sibling pairs of the device layout (bvh_build.cpp emitPairs).
__global__ void __launch_bounds__(256) emitNodes(uint32_t total, uint32_t n, const uint32_t* __restrict__ nBegin, const uint32_t* __restrict__ nEnd,
                                                 const uint32_t* __restrict__ nAxis, const uint32_t* __restrict__ nLeft, const uint32_t* __restrict__ nKeys,
                                                 const uint32_t* __restrict__ nZero, const uint32_t* __restrict__ pre, const uint32_t* __restrict__ pairOf,
                                                 YartBvhNode* __restrict__ nodes /* may be null */, DevNodePair* __restrict__ pairs, bool leftToRight)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= total) return;
    const uint32_t axis = nAxis[s];
    if (nodes) {
        const NodeBox b = boxOf(s, nKeys, nZero);
        YartBvhNode nd;
        for (int a = 0; a < 3; ++a) { nd.bounds_min[a] = b.lo[a]; nd.bounds_max[a] = b.hi[a]; }
        nd.split_axis = (uint8_t)axis;
        nd.pad[0] = nd.pad[1] = 0;
        if (axis == 3u) {
            nd.offset = leafSlot(s, n, nBegin, nEnd, leftToRight);
            nd.num_triangles = (uint8_t)(nEnd[s] - nBegin[s]);
        } else {
            nd.offset = pre[nLeft[s] + 1u];
            nd.num_triangles = 0;
        }
        nodes[pre[s]] = nd;
    }
    if (axis != 3u) {
        DevNodePair p;
        p.child[0] = recordOf(nLeft[s], n, nBegin, nEnd, nAxis, nKeys, nZero, pairOf, leftToRight);
        p.child[1] = recordOf(nLeft[s] + 1u, n, nBegin, nEnd, nAxis, nKeys, nZero, pairOf, leftToRight);
        pairs[pairOf[s]] = p;
    }
    if (s == 0u) {
        DevNodePair p;
        p.child[0] = recordOf(0u, n, nBegin, nEnd, nAxis, nKeys, nZero, pairOf, leftToRight);
        DevNodeRec empty;
        for (int a = 0; a < 3; ++a) { empty.lo[a] = __builtin_inff(); empty.hi[a] = -__builtin_inff(); }
        empty.ref = kPadLeaf;
        empty.num_prims = 0;
        empty.split_axis = 3;
        empty.pad[0] = empty.pad[1] = 0;
        p.child[1] = empty;
        pairs[0] = p;
    }
}

// include/triaccel.h:54-108: cover.h projectRecord (the expressions of bvh_build.cpp projectTriangle)
__device__ inline void projectTriangle(float out[12], const float* a, const float* b, const float* c, int32_t triIdx, int32_t meshIdx)
{
    projectRecord(out, a, b, c, triIdx, meshIdx);
}

struct NanFix { uint32_t slot, tri; };

// TriAccel records in leaf order and the slot of every input triangle (src/bvhaccel.cpp:354-360).  A record that holds
// a NaN (degenerate triangle) is listed for the host: the sign of a NaN the hardware generates is not the host's.
__global__ void __launch_bounds__(256) emitLeaves(const float4* __restrict__ rec, const uint32_t* __restrict__ leafOf, uint32_t n,
                                                  const uint32_t* __restrict__ nBegin, const uint32_t* __restrict__ nEnd,
                                                  const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                  uint32_t* __restrict__ slotOfInput, float4* __restrict__ triaccel,
                                                  uint32_t* __restrict__ counters, NanFix* __restrict__ fixes)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint32_t s = leafOf[j];
    const uint32_t slot = n - nEnd[s] + (j - nBegin[s]);
    const uint32_t id = __float_as_uint(rec[j].w);
    slotOfInput[id] = slot;
    const int32_t* t = tris + (size_t)id * 5;
    float o[12];
    projectTriangle(o, verts + (size_t)t[0] * 3, verts + (size_t)t[1] * 3, verts + (size_t)t[2] * 3, t[4], t[3]);
    bool nan = false;
    for (int c = 0; c < 12; ++c)
        if ((c & 3) != 3 && !(o[c] == o[c])) nan = true;
    if (nan) {
        const uint32_t f = atomicAdd(counters + 2, 1u);
        if (f < kFixCap) fixes[f] = NanFix{ slot, id };
    }
    triaccel[(size_t)slot * 3 + 0] = make_float4(o[0], o[1], o[2], o[3]);
    triaccel[(size_t)slot * 3 + 1] = make_float4(o[4], o[5], o[6], o[7]);
    triaccel[(size_t)slot * 3 + 2] = make_float4(o[8], o[9], o[10], o[11]);
}

// The engine's sphere tree (bvh_build.cpp buildSphereBvh): boxes widened by a relative margin so that the fp32 slab test
// never rejects a ray the fp64 sphere test accepts, centre = the sphere's.  status[1]: a NaN (host builder's case).
__global__ void __launch_bounds__(256) spherePrims(const float* __restrict__ spheres, uint32_t n, Cover cover, float4* __restrict__ primLo,
                                                   float4* __restrict__ primHi, float4* __restrict__ rec, uint32_t* __restrict__ seg,
                                                   uint32_t* __restrict__ status)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* s = spheres + (size_t)i * 4;
    const float r = fabsf(s[3]);
    float lo[3], hi[3];
    bool nan = !(r == r);
    sphereBox(s, sqrt(sphereReff2(s, cover)), lo, hi);           // cover.h: everything the sphere test can accept
    for (int a = 0; a < 3; ++a)
        nan = nan || !(s[a] == s[a]) || !(lo[a] == lo[a]) || !(hi[a] == hi[a]);
    if (nan) status[1] = 1u;
    seg[i] = 0u;
    primLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    primHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    rec[i] = make_float4(s[0], s[1], s[2], __uint_as_float(i));
}

// sphere records in leaf-slot order (= final position: leaves are numbered left to right) and the slot -> sphere table
__global__ void __launch_bounds__(256) emitSphereLeaves(const float4* __restrict__ rec, uint32_t n, const float4* __restrict__ spheres,
                                                        float4* __restrict__ records, uint32_t* __restrict__ index)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint32_t id = __float_as_uint(rec[j].w);
    records[j] = spheres[id];
    index[j] = id;
}

struct Scratch {                    // frees whatever was allocated, whichever way the build ends
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t count)
    {
        *p = nullptr;
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

uint32_t blocks(size_t n, uint32_t per = 256) { return static_cast<uint32_t>((n + per - 1) / per); }

} // namespace

void DeviceRefTree::release()
{
    (void)hipFree(nodes); (void)hipFree(pairs); (void)hipFree(triaccel); (void)hipFree(slotOfInput);
    nodes = pairs = triaccel = slotOfInput = nullptr;
}

namespace {

// The level-synchronous partition build shared by the two trees: everything between "boxes, centres and the root are
// set" and "nodes are numbered".
struct Topology {
    Scratch S;
    uint32_t n = 0;
    float4 *primLo = nullptr, *primHi = nullptr, *recA = nullptr, *recB = nullptr, *order = nullptr;
    uint32_t *seg = nullptr, *leafOf = nullptr, *flags = nullptr, *T = nullptr, *failPos = nullptr, *passPos = nullptr, *counters = nullptr;
    uint32_t *nBegin = nullptr, *nEnd = nullptr, *nKeys = nullptr, *nZero = nullptr, *nAxis = nullptr, *nSplit = nullptr, *nLeft = nullptr;
    uint32_t *size = nullptr, *interior = nullptr, *pre = nullptr, *pairOf = nullptr;
    double* nMid = nullptr;
    OneSided* oneSided = nullptr;
    char* scanTemp = nullptr;
    size_t scanBytes = 0;
    uint32_t total = 0, nInterior = 0;
    int depth = 0;
    size_t medianSplits = 0;

    int allocate(size_t count, std::string& err)
    {
        n = static_cast<uint32_t>(count);
        const size_t cap = 2 * count;                   // both children of a split hold at least one primitive
        YART_HIP(S.alloc(&primLo, count));
        YART_HIP(S.alloc(&primHi, count));
        YART_HIP(S.alloc(&recA, count));
        YART_HIP(S.alloc(&recB, count));
        YART_HIP(S.alloc(&seg, count));
        YART_HIP(S.alloc(&leafOf, count));
        YART_HIP(S.alloc(&flags, count + 1));
        YART_HIP(S.alloc(&T, count + 1));
        YART_HIP(S.alloc(&failPos, count));
        YART_HIP(S.alloc(&passPos, count));
        YART_HIP(S.alloc(&counters, 64));
        YART_HIP(S.alloc(&nBegin, cap));
        YART_HIP(S.alloc(&nEnd, cap));
        YART_HIP(S.alloc(&nKeys, cap * 6));
        YART_HIP(S.alloc(&nZero, cap * 6));
        YART_HIP(S.alloc(&nAxis, cap));
        YART_HIP(S.alloc(&nSplit, cap));
        YART_HIP(S.alloc(&nLeft, cap));
        YART_HIP(S.alloc(&nMid, cap));
        YART_HIP(S.alloc(&oneSided, kFixCap));
        YART_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, flags, T, (int)(count + 1), nullptr));
        YART_HIP(S.alloc(&scanTemp, scanBytes));
        const uint32_t initCounters[4] = { 1u, 0u, 0u, 0u };   // nodes, one-sided ranges, NaN records; words 8, 9: status
        YART_HIP(hipMemcpyAsync(counters, initCounters, sizeof initCounters, hipMemcpyHostToDevice, nullptr));
        YART_HIP(hipMemsetAsync(counters + 8, 0, 8, nullptr));
        hipLaunchKernelGGL(initRoot, dim3(1), dim3(64), 0, nullptr, n, nBegin, nEnd, nKeys, nZero);
        YART_HIP(hipGetLastError());
        return YART_OK;
    }

    // primLo / primHi / recA / seg filled.  leafBelow: a range of fewer primitives is a leaf.
    int run(uint32_t leafBelow, const char* what, bool* useHost, std::string& err)
    {
        std::vector<uint32_t> levelStart{ 0u };
        uint32_t allocated = 1;
        float4 *src = recA, *dst = recB;
        std::vector<OneSided> ranges;
        std::vector<float4> hostRecs, hostOut;
        struct Item { float key; uint32_t idx; };
        std::vector<Item> items;
        for (;;) {
            const uint32_t first = levelStart.back(), count = allocated - first;
            if (static_cast<int>(levelStart.size()) > kMaxDepth) { err = std::string(what) + " BVH deeper than 64 levels"; return YART_ERR_TREE_TOO_DEEP; }
            hipLaunchKernelGGL(segBounds, dim3(blocks(n, 1024)), dim3(1024), 0, nullptr, src, seg, n, primLo, primHi, nKeys, nZero);
            hipLaunchKernelGGL(segPlan, dim3(blocks(count)), dim3(256), 0, nullptr, first, count, nBegin, nEnd, nKeys, nAxis, nMid, leafBelow);
            hipLaunchKernelGGL(segFlags, dim3(blocks(size_t(n) + 1)), dim3(256), 0, nullptr, src, seg, n, nAxis, nMid, leafOf, flags);
            YART_HIP(hipGetLastError());
            YART_HIP(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, flags, T, (int)(n + 1), nullptr));
            hipLaunchKernelGGL(segSplit, dim3(blocks(count)), dim3(256), 0, nullptr, first, count, nBegin, nEnd, nAxis, T, nSplit, nLeft, nKeys, nZero,
                               counters, oneSided);
            YART_HIP(hipGetLastError());
            uint32_t c[2];
            YART_HIP(hipMemcpy(c, counters, sizeof c, hipMemcpyDeviceToHost));
            levelStart.push_back(allocated);
            if (c[0] == allocated) break;                    // every node of this level is a leaf
            if (c[1]) {
                // one-sided ranges: the same std::nth_element the host builder runs, on the same order of elements
                medianSplits += c[1];
                if (c[1] > kFixCap || medianSplits > 4 * static_cast<size_t>(kFixCap)) { *useHost = true; return YART_OK; }
                ranges.resize(c[1]);
                YART_HIP(hipMemcpy(ranges.data(), oneSided, c[1] * sizeof(OneSided), hipMemcpyDeviceToHost));
                size_t elems = 0;
                for (const OneSided& r : ranges) elems += r.end - r.begin;
                if (elems > (size_t(1) << 24)) { *useHost = true; return YART_OK; }
                for (const OneSided& r : ranges) {
                    const size_t cnt = r.end - r.begin;
                    hostRecs.resize(cnt);
                    hostOut.resize(cnt);
                    items.resize(cnt);
                    YART_HIP(hipMemcpy(hostRecs.data(), src + r.begin, cnt * sizeof(float4), hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < cnt; ++i) {
                        const float4& h = hostRecs[i];
                        items[i] = Item{ r.axis == 0 ? h.x : (r.axis == 1 ? h.y : h.z), static_cast<uint32_t>(i) };
                    }
                    std::nth_element(items.begin(), items.begin() + cnt / 2, items.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
                    for (size_t i = 0; i < cnt; ++i) hostOut[i] = hostRecs[items[i].idx];
                    YART_HIP(hipMemcpy(src + r.begin, hostOut.data(), cnt * sizeof(float4), hipMemcpyHostToDevice));
                }
                YART_HIP(hipMemsetAsync(counters + 1, 0, 4, nullptr));
            }
            hipLaunchKernelGGL(partners, dim3(blocks(n)), dim3(256), 0, nullptr, seg, n, nBegin, nEnd, nSplit, flags, T, failPos, passPos);
            hipLaunchKernelGGL(applySplit, dim3(blocks(n)), dim3(256), 0, nullptr, src, dst, seg, n, nBegin, nEnd, nSplit, nLeft, flags, T, failPos, passPos);
            YART_HIP(hipGetLastError());
            std::swap(src, dst);
            allocated = c[0];
        }
        order = src;
        total = allocated;
        depth = static_cast<int>(levelStart.size()) - 1;
        if (total - 1 >= kMaxPairs) { err = "too many BVH nodes"; return YART_ERR_INVALID_ARGUMENT; }
        // numbering: subtree sizes bottom-up, pre-order and pair numbers top-down
        YART_HIP(S.alloc(&size, total));
        YART_HIP(S.alloc(&interior, total));
        YART_HIP(S.alloc(&pre, total));
        YART_HIP(S.alloc(&pairOf, total));
        for (int lv = depth - 1; lv >= 0; --lv) {
            const uint32_t first = levelStart[lv], count = levelStart[lv + 1] - first;
            hipLaunchKernelGGL(subtreeSizes, dim3(blocks(count)), dim3(256), 0, nullptr, first, count, nAxis, nLeft, size, interior);
        }
        for (int lv = 0; lv < depth; ++lv) {
            const uint32_t first = levelStart[lv], count = levelStart[lv + 1] - first;
            hipLaunchKernelGGL(numberNodes, dim3(blocks(count)), dim3(256), 0, nullptr, first, count, nAxis, nLeft, size, interior, pre, pairOf);
        }
        YART_HIP(hipGetLastError());
        YART_HIP(hipMemcpy(&nInterior, interior, 4, hipMemcpyDeviceToHost));
        return YART_OK;
    }
};

} // namespace

int buildReferenceTreeDevice(const float* dVerts, const float* verts, size_t nVerts, const int32_t* dTris, const int32_t* tris,
                             size_t nTris, DeviceRefTree& out, bool* useHost, std::string& err)
{
    out = DeviceRefTree{};
    *useHost = false;
    if (nTris < 1 || nTris >= kMaxLeafSlot) { *useHost = true; return YART_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    Topology T;
    int st = T.allocate(nTris, err);
    if (st != YART_OK) return st;
    const uint32_t n = T.n;
    NanFix* fixes;
    YART_HIP(T.S.alloc(&fixes, kFixCap));
    hipLaunchKernelGGL(refPrims, dim3(blocks(nTris)), dim3(256), 0, nullptr, dVerts, dTris, n, (uint64_t)nVerts, T.primLo, T.primHi, T.recA, T.seg, T.counters + 8);
    YART_HIP(hipGetLastError());
    uint32_t status[2];
    YART_HIP(hipMemcpy(status, T.counters + 8, sizeof status, hipMemcpyDeviceToHost));
    if (status[0]) { err = "yart_accel_create: vertex index out of range"; return YART_ERR_INVALID_ARGUMENT; }
    if (status[1]) { *useHost = true; return YART_OK; }
    st = T.run(kLeafBelow, "triangle", useHost, err);
    if (st != YART_OK || *useHost) return st;

    // the arrays the caller keeps
    DeviceRefTree r;
    r.numNodes = T.total;
    r.numPairs = static_cast<size_t>(T.nInterior) + 1;
    r.numTriangles = nTris;
    r.depth = T.depth;
    hipError_t e = hipMalloc(&r.nodes, r.numNodes * sizeof(YartBvhNode));
    if (e == hipSuccess) e = hipMalloc(&r.pairs, r.numPairs * sizeof(DevNodePair));
    if (e == hipSuccess) e = hipMalloc(&r.triaccel, nTris * sizeof(YartTriAccel));
    if (e == hipSuccess) e = hipMalloc(&r.slotOfInput, nTris * sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(emitNodes, dim3(blocks(T.total)), dim3(256), 0, nullptr, T.total, n, T.nBegin, T.nEnd, T.nAxis, T.nLeft, T.nKeys, T.nZero, T.pre, T.pairOf,
                           static_cast<YartBvhNode*>(r.nodes), static_cast<DevNodePair*>(r.pairs), false);
        hipLaunchKernelGGL(emitLeaves, dim3(blocks(nTris)), dim3(256), 0, nullptr, T.order, T.leafOf, n, T.nBegin, T.nEnd, dVerts, dTris,
                           static_cast<uint32_t*>(r.slotOfInput), static_cast<float4*>(r.triaccel), T.counters, fixes);
        e = hipGetLastError();
    }
    uint32_t nFix = 0;
    if (e == hipSuccess) e = hipMemcpy(&nFix, T.counters + 2, 4, hipMemcpyDeviceToHost);
    DevNodePair rootPair;
    if (e == hipSuccess) e = hipMemcpy(&rootPair, r.pairs, sizeof rootPair, hipMemcpyDeviceToHost);
    if (e == hipSuccess && nFix > kFixCap) { r.release(); *useHost = true; return YART_OK; }
    if (e == hipSuccess && nFix) {
        std::vector<NanFix> fx(nFix);
        e = hipMemcpy(fx.data(), fixes, nFix * sizeof(NanFix), hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < nFix && e == hipSuccess; ++i) {
            const int32_t* t = tris + static_cast<size_t>(fx[i].tri) * 5;
            YartTriAccel ta;
            projectTriangleHost(ta, verts + static_cast<size_t>(t[0]) * 3, verts + static_cast<size_t>(t[1]) * 3,
                                verts + static_cast<size_t>(t[2]) * 3, t[4], t[3]);
            e = hipMemcpy(static_cast<YartTriAccel*>(r.triaccel) + fx[i].slot, &ta, sizeof ta, hipMemcpyHostToDevice);
        }
        r.projectionsOnHost = nFix;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        r.release();
        err = std::string("device reference-tree build: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? YART_ERR_OUT_OF_MEMORY : YART_ERR_HIP;
    }
    std::memcpy(r.rootLo, rootPair.child[0].lo, 12);
    std::memcpy(r.rootHi, rootPair.child[0].hi, 12);
    r.medianSplitsOnHost = T.medianSplits;
    r.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out = r;
    return YART_OK;
}

void DeviceSphereTree::release()
{
    (void)hipFree(pairs); (void)hipFree(records); (void)hipFree(index);
    pairs = records = index = nullptr;
}

int buildSphereTreeDevice(const float* dSpheres, size_t nSpheres, const Cover& cover, DeviceSphereTree& out, bool* useHost, std::string& err)
{
    out = DeviceSphereTree{};
    *useHost = false;
    if (nSpheres < 1 || nSpheres >= kMaxLeafSlot) { *useHost = true; return YART_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    Topology T;
    int st = T.allocate(nSpheres, err);
    if (st != YART_OK) return st;
    const uint32_t n = T.n;
    hipLaunchKernelGGL(spherePrims, dim3(blocks(nSpheres)), dim3(256), 0, nullptr, dSpheres, n, cover, T.primLo, T.primHi, T.recA, T.seg, T.counters + 8);
    YART_HIP(hipGetLastError());
    uint32_t status[2];
    YART_HIP(hipMemcpy(status, T.counters + 8, sizeof status, hipMemcpyDeviceToHost));
    if (status[1]) { *useHost = true; return YART_OK; }
    st = T.run(kSphereLeafBelow, "sphere", useHost, err);
    if (st != YART_OK || *useHost) return st;

    DeviceSphereTree r;
    r.numPairs = static_cast<size_t>(T.nInterior) + 1;
    r.numSpheres = nSpheres;
    r.depth = T.depth;
    hipError_t e = hipMalloc(&r.pairs, r.numPairs * sizeof(DevNodePair));
    if (e == hipSuccess) e = hipMalloc(&r.records, nSpheres * 16);
    if (e == hipSuccess) e = hipMalloc(&r.index, nSpheres * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(emitNodes, dim3(blocks(T.total)), dim3(256), 0, nullptr, T.total, n, T.nBegin, T.nEnd, T.nAxis, T.nLeft, T.nKeys, T.nZero, T.pre, T.pairOf,
                           static_cast<YartBvhNode*>(nullptr), static_cast<DevNodePair*>(r.pairs), true);
        hipLaunchKernelGGL(emitSphereLeaves, dim3(blocks(nSpheres)), dim3(256), 0, nullptr, T.order, n, reinterpret_cast<const float4*>(dSpheres),
                           static_cast<float4*>(r.records), static_cast<uint32_t*>(r.index));
        e = hipGetLastError();
    }
    DevNodePair rootPair;
    if (e == hipSuccess) e = hipMemcpy(&rootPair, r.pairs, sizeof rootPair, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        r.release();
        err = std::string("device sphere-tree build: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? YART_ERR_OUT_OF_MEMORY : YART_ERR_HIP;
    }
    std::memcpy(r.rootLo, rootPair.child[0].lo, 12);
    std::memcpy(r.rootHi, rootPair.child[0].hi, 12);
    r.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out = r;
    return YART_OK;
}

} // namespace yart
