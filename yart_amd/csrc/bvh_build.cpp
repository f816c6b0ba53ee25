// Host-side BVH / TriAccel build.  See bvh_build.h.
//
// Compile with -ffp-contract=off: every float expression below must round
// exactly as the reference's C++ expression does, so that the node and
// TriAccel arrays are bit-identical to the reference's.
#include "bvh_build.h"
#include "task_pool.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>

namespace yart {
namespace {

constexpr int kLeafThreshold = 8;     // src/bvhaccel.cpp:57 -- leaf iff count < 8

struct Bounds {
    float lo[3], hi[3];
    void reset()
    {
        const float inf = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) { lo[a] = inf; hi[a] = -inf; }     // include/bbox.h:13-16
    }
    // include/bbox.h:80-90 with std::min / std::max argument order preserved
    void grow(const Bounds& r)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = (r.lo[a] < lo[a]) ? r.lo[a] : lo[a];
            hi[a] = (hi[a] < r.hi[a]) ? r.hi[a] : hi[a];
        }
    }
    void growPoint(const float* p)                                       // include/bbox.h:70-78
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = (p[a] < lo[a]) ? p[a] : lo[a];
            hi[a] = (hi[a] < p[a]) ? p[a] : hi[a];
        }
    }
    int widestAxis() const                                               // include/bbox.h:52-62
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        return (dy > dz) ? 1 : 2;
    }
};

struct Prim {           // what the reference keeps per triangle while building (src/bvhaccel.cpp:18-30)
    Bounds box;
    float centre[3];
    uint32_t id;        // index into the caller's primitive array
};

// Partition step of the reference builder (src/bvhaccel.cpp:125-143): midpoint of
// the node bounds on the widest axis (computed in double from a float sum),
// std::partition by centre < midpoint, std::nth_element median when one side is
// empty.  Returns the split position.
uint32_t splitRange(std::vector<Prim>& prims, uint32_t begin, uint32_t end, const Bounds& box, int axis)
{
    const double midpoint = static_cast<double>(box.hi[axis] + box.lo[axis]) * 0.5;
    auto first = prims.begin() + begin, last = prims.begin() + end;
    auto mid = std::partition(first, last, [midpoint, axis](const Prim& p) {
        return static_cast<double>(p.centre[axis]) < midpoint;
    });
    if (mid == first || mid == last) {
        mid = first + (end - begin) / 2;
        std::nth_element(first, mid, last, [axis](const Prim& a, const Prim& b) {
            return a.centre[axis] < b.centre[axis];
        });
    }
    return static_cast<uint32_t>(mid - prims.begin());
}

struct FlatTree {
    // reference order: pre-order, left child at i+1, right child at right[i]
    std::vector<Bounds> box;
    std::vector<uint32_t> right;      // interior: index of right child
    std::vector<uint8_t> axis;        // 0..2 interior, 3 leaf
    std::vector<uint32_t> begin, end; // leaf: range in the (permuted) prim array
    int depth = 0;
};

// Top-down build in two steps.  (1) The ranges are split by jobs on the host thread pool into a
// linked tree: the partition of a range depends only on the range's own primitives, and every range
// is handled by exactly one job with the same std:: algorithms, so the tree is the one the
// reference's recursion builds whatever the schedule.  (2) A serial walk numbers the nodes in the
// reference's flattened pre-order (src/bvhaccel.cpp:215-234).
struct LinkNode {                     // trivially constructible: the pool is never value-initialised
    Bounds box;
    uint32_t begin, end;
    uint32_t left;                    // interior: children at left, left + 1; leaf: 0
    uint32_t axis;
};

class LinkBuilder {
public:
    LinkBuilder(std::vector<Prim>& prims, int leafThreshold)
        : prims_(prims), leafThreshold_(leafThreshold), nodes_(new LinkNode[2 * prims.size() + 2]) { }

    void run()
    {
        unsigned nThreads = buildThreads();
        if (prims_.size() < kSpawnAbove) nThreads = 1;
        pool_.run(nThreads, Job{ 0u, 0u, static_cast<uint32_t>(prims_.size()) }, [this](const Job& j) { build(j); });
    }

    const LinkNode* nodes() const { return nodes_.get(); }

private:
    struct Job { uint32_t node, begin, end; };
    static constexpr uint32_t kSpawnAbove = 16384;

    void build(const Job& first)
    {
        std::vector<Job> stack{ first };
        while (!stack.empty()) {
            const Job j = stack.back();
            stack.pop_back();
            LinkNode& n = nodes_[j.node];
            n.box.reset();
            for (uint32_t i = j.begin; i < j.end; ++i) n.box.grow(prims_[i].box);
            n.begin = j.begin;
            n.end = j.end;
            n.left = 0;
            n.axis = 3;
            if (static_cast<int>(j.end - j.begin) < leafThreshold_) continue;
            n.axis = static_cast<uint32_t>(n.box.widestAxis());
            const uint32_t mid = splitRange(prims_, j.begin, j.end, n.box, static_cast<int>(n.axis));
            const uint32_t l = next_.fetch_add(2);
            n.left = l;
            const Job lj{ l, j.begin, mid }, rj{ l + 1, mid, j.end };
            if (j.end - j.begin > kSpawnAbove) { pool_.spawn(lj); pool_.spawn(rj); }
            else { stack.push_back(rj); stack.push_back(lj); }
        }
    }

    std::vector<Prim>& prims_;
    int leafThreshold_;
    std::unique_ptr<LinkNode[]> nodes_;
    std::atomic<uint32_t> next_{ 1 };
    TaskPool<Job> pool_;
};

void buildFlat(std::vector<Prim>& prims, int leafThreshold, FlatTree& t)
{
    LinkBuilder lb(prims, leafThreshold);
    lb.run();
    const LinkNode* ln = lb.nodes();
    struct Visit { uint32_t node; int32_t parent; int depth; };
    std::vector<Visit> stack{ Visit{ 0u, -1, 1 } };
    while (!stack.empty()) {
        const Visit v = stack.back();
        stack.pop_back();
        const LinkNode& n = ln[v.node];
        const uint32_t idx = static_cast<uint32_t>(t.box.size());
        if (v.parent >= 0) t.right[v.parent] = idx;         // only right children carry a parent
        t.box.push_back(n.box);
        t.right.push_back(0);
        t.begin.push_back(n.begin);
        t.end.push_back(n.end);
        t.axis.push_back(static_cast<uint8_t>(n.axis));
        t.depth = std::max(t.depth, v.depth);
        if (n.axis == 3) continue;
        // left child is numbered next (idx + 1); the right one when the left subtree is done
        stack.push_back(Visit{ n.left + 1, static_cast<int32_t>(idx), v.depth + 1 });
        stack.push_back(Visit{ n.left, -1, v.depth + 1 });
    }
}

// Leaf slots.  The reference appends a leaf's triangles to its leaf-order list when
// the leaf is created; its two recursive calls are function arguments
// (src/bvhaccel.cpp:146-150) and g++ evaluates them right to left, so leaves are
// numbered in right-child-first depth-first order.
void assignLeafSlots(const FlatTree& t, std::vector<uint32_t>& slotOfLeaf)
{
    slotOfLeaf.assign(t.box.size(), 0);
    std::vector<uint32_t> stack{ 0u };
    uint32_t next = 0;
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        if (t.axis[n] == 3) {
            slotOfLeaf[n] = next;
            next += t.end[n] - t.begin[n];
        } else {
            stack.push_back(n + 1);        // left: visited after ...
            stack.push_back(t.right[n]);   // ... the right subtree
        }
    }
}

// Sibling-pair layout for the device.  Pair 0 holds the root next to an empty leaf;
// the pair of interior node i holds records of its children i+1 and right[i].  Pairs
// are numbered in pre-order of their parent, so the left child's pair follows its
// parent's pair directly.
void emitPairs(const FlatTree& t, const std::vector<uint32_t>& slotOfLeaf, std::vector<DevNodePair>& pairs)
{
    const size_t n = t.box.size();
    std::vector<uint32_t> pairOf(n, 0);
    uint32_t np = 1;
    for (size_t i = 0; i < n; ++i)
        if (t.axis[i] != 3) pairOf[i] = np++;
    pairs.assign(np, DevNodePair{});
    auto record = [&](uint32_t node) {
        DevNodeRec r;
        std::memset(&r, 0, sizeof r);
        std::memcpy(r.lo, t.box[node].lo, 12);
        std::memcpy(r.hi, t.box[node].hi, 12);
        r.split_axis = t.axis[node];
        if (t.axis[node] == 3) {
            const uint32_t cnt = t.end[node] - t.begin[node];
            r.num_prims = static_cast<uint8_t>(cnt);
            r.ref = leafRef(slotOfLeaf[node], cnt);
        } else {
            r.ref = interiorRef(pairOf[node], t.axis[node]);
        }
        return r;
    };
    DevNodeRec empty;
    std::memset(&empty, 0, sizeof empty);
    const float inf = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; ++a) { empty.lo[a] = inf; empty.hi[a] = -inf; }
    empty.ref = kPadLeaf;
    empty.split_axis = 3;
    pairs[0].child[0] = record(0);
    pairs[0].child[1] = empty;
    parallelChunks(n, buildThreads(), size_t(1) << 15, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            if (t.axis[i] == 3) continue;
            pairs[pairOf[i]].child[0] = record(static_cast<uint32_t>(i) + 1);
            pairs[pairOf[i]].child[1] = record(t.right[i]);
        }
    });
}

// include/triaccel.h:54-108
void projectTriangle(YartTriAccel& ta, const float* a, const float* b, const float* c,
                     int32_t triIdx, int32_t meshIdx)
{
    const float ab[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    const float ac[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    const float nrm[3] = { ab[1] * ac[2] - ab[2] * ac[1],
                           ab[2] * ac[0] - ab[0] * ac[2],
                           ab[0] * ac[1] - ab[1] * ac[0] };
    int k = 1;
    const float ax = std::fabs(nrm[0]), ay = std::fabs(nrm[1]), az = std::fabs(nrm[2]);
    if (ax > ay) k = (ax > az) ? 0 : 2;
    else if (az > ay) k = 2;
    const int u = (k + 1) % 3, v = (k + 2) % 3;
    const float proj[3] = { nrm[0] / nrm[k], nrm[1] / nrm[k], nrm[2] / nrm[k] };
    ta.n_u = proj[u];
    ta.n_v = proj[v];
    ta.n_d = a[0] * proj[0] + a[1] * proj[1] + a[2] * proj[2];
    ta.k = k;
    const float det = ab[u] * ac[v] - ab[v] * ac[u];
    ta.b_u = -ab[v] / det;
    ta.b_v = ab[u] / det;
    ta.b_d = (ab[v] * a[u] - ab[u] * a[v]) / det;
    ta.triIdx = triIdx;
    ta.c_u = ac[v] / det;
    ta.c_v = -ac[u] / det;
    ta.c_d = -(ac[v] * a[u] - ac[u] * a[v]) / det;
    ta.meshIdx = meshIdx;
}

} // namespace

void projectTriangleHost(YartTriAccel& ta, const float* a, const float* b, const float* c, int32_t triIdx, int32_t meshIdx)
{
    projectTriangle(ta, a, b, c, triIdx, meshIdx);
}

int buildTriangleBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                     TriangleBvh& out, std::string& err)
{
    if ((nTris && (!verts || !tris)) || nTris >= kMaxLeafSlot) {
        err = "yart_accel_create: null triangle input or too many triangles";
        return YART_ERR_INVALID_ARGUMENT;
    }
    std::vector<Prim> prims(nTris);
    const unsigned nThreads = buildThreads();
    std::atomic<bool> badIndex{ false };
    parallelChunks(nTris, nThreads, size_t(1) << 16, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const int32_t* t = tris + i * 5;
            bool ok = true;
            for (int c = 0; c < 3; ++c) ok = ok && t[c] >= 0 && static_cast<size_t>(t[c]) < nVerts;
            if (!ok) { badIndex = true; return; }
            Prim& p = prims[i];
            const float* v0 = verts + static_cast<size_t>(t[0]) * 3;
            std::memcpy(p.box.lo, v0, 12);                              // BBox(point), include/bbox.h:22
            std::memcpy(p.box.hi, v0, 12);
            p.box.growPoint(verts + static_cast<size_t>(t[1]) * 3);     // src/bvhaccel.cpp:337-339
            p.box.growPoint(verts + static_cast<size_t>(t[2]) * 3);
            for (int a = 0; a < 3; ++a) p.centre[a] = (p.box.lo[a] + p.box.hi[a]) * 0.5f;   // :28
            p.id = static_cast<uint32_t>(i);
        }
    });
    if (badIndex) {
        err = "yart_accel_create: vertex index out of range";
        return YART_ERR_INVALID_ARGUMENT;
    }

    const bool timing = std::getenv("YART_BUILD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    FlatTree t;
    buildFlat(prims, kLeafThreshold, t);
    if (timing) fprintf(stderr, "reference tree: partition build %.2fs\n", now() - t0);
    const double t1 = now();
    if (t.depth > kMaxDepth) {
        err = "triangle BVH deeper than 64 levels";
        return YART_ERR_TREE_TOO_DEEP;
    }
    std::vector<uint32_t> slot;
    assignLeafSlots(t, slot);

    const size_t nNodes = t.box.size();
    out.depth = t.depth;
    out.nodes.assign(nNodes, YartBvhNode{});
    out.triaccel.assign(nTris, YartTriAccel{});
    out.slotOfInput.assign(nTris, 0);
    parallelChunks(nNodes, nThreads, size_t(1) << 14, [&](size_t nb, size_t ne) {
        for (size_t i = nb; i < ne; ++i) {
            YartBvhNode& n = out.nodes[i];
            std::memcpy(n.bounds_min, t.box[i].lo, 12);
            std::memcpy(n.bounds_max, t.box[i].hi, 12);
            n.split_axis = t.axis[i];
            if (t.axis[i] == 3) {
                n.offset = slot[i];
                n.num_triangles = static_cast<uint8_t>(t.end[i] - t.begin[i]);
                for (uint32_t j = t.begin[i]; j < t.end[i]; ++j) {      // src/bvhaccel.cpp:354-360
                    const int32_t* tr = tris + static_cast<size_t>(prims[j].id) * 5;
                    out.slotOfInput[prims[j].id] = slot[i] + (j - t.begin[i]);
                    projectTriangle(out.triaccel[slot[i] + (j - t.begin[i])],
                                    verts + static_cast<size_t>(tr[0]) * 3,
                                    verts + static_cast<size_t>(tr[1]) * 3,
                                    verts + static_cast<size_t>(tr[2]) * 3, tr[4], tr[3]);
                }
            } else {
                n.offset = t.right[i];
            }
        }
    });
    if (nNodes - 1 >= kMaxPairs) {
        err = "too many BVH nodes";
        return YART_ERR_INVALID_ARGUMENT;
    }
    const double t2 = now();
    emitPairs(t, slot, out.pairs);
    if (timing) fprintf(stderr, "reference tree: leaf slots + node records + projection %.2fs, pair layout %.2fs\n", t2 - t1, now() - t2);
    return YART_OK;
}

// Sphere BVH: the reference has none (linear loop, include/scene.h:142-144), so the
// tree shape is the engine's own: same midpoint/median splitter as the triangle
// tree, leaves of at most 4 spheres, leaf slots in left-to-right order.  A sphere's box
// holds everything the fp64 sphere test (include/sphere.h:63-97) can accept from a
// covered ray, its rounding noise included (cover.h: sphereBox).
int buildSphereBvh(const float* spheres, size_t nSpheres, const Cover& cover, SphereBvh& out, std::string& err)
{
    out = SphereBvh{};
    if (nSpheres == 0) return YART_OK;
    if (!spheres || nSpheres >= kMaxLeafSlot) {
        err = "yart_accel_create: null sphere input or too many spheres";
        return YART_ERR_INVALID_ARGUMENT;
    }
    std::vector<Prim> prims(nSpheres);
    for (size_t i = 0; i < nSpheres; ++i) {
        const float* s = spheres + i * 4;
        Prim& p = prims[i];
        sphereBox(s, std::sqrt(sphereReff2(s, cover)), p.box.lo, p.box.hi);
        for (int a = 0; a < 3; ++a) p.centre[a] = s[a];
        p.id = static_cast<uint32_t>(i);
    }
    FlatTree t;
    buildFlat(prims, 5, t);
    if (t.depth > kMaxDepth) {
        err = "sphere BVH deeper than 64 levels";
        return YART_ERR_TREE_TOO_DEEP;
    }
    std::vector<uint32_t> slot(t.box.size(), 0);
    uint32_t next = 0;
    for (size_t i = 0; i < t.box.size(); ++i)       // pre-order = left-to-right over leaves
        if (t.axis[i] == 3) { slot[i] = next; next += t.end[i] - t.begin[i]; }
    out.depth = t.depth;
    out.records.resize(nSpheres * 4);
    out.index.resize(nSpheres);
    for (size_t i = 0; i < t.box.size(); ++i) {
        if (t.axis[i] != 3) continue;
        for (uint32_t j = t.begin[i]; j < t.end[i]; ++j) {
            const uint32_t s = slot[i] + (j - t.begin[i]);
            std::memcpy(&out.records[static_cast<size_t>(s) * 4], spheres + static_cast<size_t>(prims[j].id) * 4, 16);
            out.index[s] = prims[j].id;
        }
    }
    emitPairs(t, slot, out.pairs);
    return YART_OK;
}

} // namespace yart
