// TEST INFRASTRUCTURE ONLY -- the product path (yart_amd/) never includes,
// links or calls anything in this file.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load the library built from it.
//
// cpu_ref.cpp: scalar CPU restatement of the reference's intersection hot
// path (elaic/yart), written from the reference's behaviour, each function
// citing the reference file:line it follows.  Parity is PINNED: this
// restatement is checked bit-for-bit against the real reference compiled in
// the authoring container (oracle/_ref/libyart_ref.so, see oracle/ref_shim.cpp)
// and against the golden vectors under tests/golden/ that were generated from
// that real reference (oracle/make_golden.py).
//
// Arithmetic contract: IEEE fp32/fp64 exactly as the C++ expressions of the
// reference are written, evaluated left to right, NO floating-point
// contraction (build with -ffp-contract=off).  The real reference built with
// -ffp-contract=off agrees bit-for-bit; built with GCC's default (=fast plus
// -mfma) it differs in low bits of t/u/v (spread recorded in tests/golden/).
//
// Builder order: the reference's buildRecursive passes its two recursive calls
// as function arguments (src/bvhaccel.cpp:146-150); g++ evaluates them right to
// left, so the RIGHT subtree appends its triangles to the leaf-order list
// first.  This restatement follows the g++ build of the reference (the one the
// golden vectors come from).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline float comp(const V3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

struct Box { V3 mn, mx; };

// include/vector.h:238-250
struct Ray { float o[3]; float tmin; float d[3]; float tmax; };

// 16-byte hit record of the engine boundary: t,u,v as the reference's
// RayHitInfo (include/vector.h:255-263) + the primitive identity the reference
// never exposes.
struct Hit { float t, u, v; int32_t prim; };

// include/triaccel.h:16-32
struct TriAccel {
    float n_u, n_v, n_d; int32_t k;
    float b_u, b_v, b_d; int32_t triIdx;
    float c_u, c_v, c_d; int32_t meshIdx;
};
static_assert(sizeof(TriAccel) == 48, "TriAccel is 48 bytes");

// src/bvhaccel.cpp:64-93 (normalised: numTriangles = 0 in interior nodes)
struct Node {
    float b[6];
    uint32_t offset;
    uint8_t numTriangles;
    uint8_t splitAxis;
    uint8_t pad[2];
};
static_assert(sizeof(Node) == 32, "node is 32 bytes");

struct BoundsInfo {           // src/bvhaccel.cpp:18-30
    Box bounds;
    V3 center;
    uint32_t meshId, triId;
};

inline float fmin2(float a, float b) { return b < a ? b : a; }   // std::min
inline float fmax2(float a, float b) { return a < b ? b : a; }   // std::max

inline Box emptyBox()         // include/bbox.h:13-16
{
    const float inf = std::numeric_limits<float>::infinity();
    return Box{ V3{ inf, inf, inf }, V3{ -inf, -inf, -inf } };
}
inline Box unionPoint(const Box& b, const V3& p)   // include/bbox.h:70-78
{
    return Box{ V3{ fmin2(b.mn.x, p.x), fmin2(b.mn.y, p.y), fmin2(b.mn.z, p.z) },
                V3{ fmax2(b.mx.x, p.x), fmax2(b.mx.y, p.y), fmax2(b.mx.z, p.z) } };
}
inline Box unionBox(const Box& a, const Box& b)    // include/bbox.h:80-90
{
    return Box{ V3{ fmin2(a.mn.x, b.mn.x), fmin2(a.mn.y, b.mn.y), fmin2(a.mn.z, b.mn.z) },
                V3{ fmax2(a.mx.x, b.mx.x), fmax2(a.mx.y, b.mx.y), fmax2(a.mx.z, b.mx.z) } };
}
inline int maxExtent(const Box& b)                 // include/bbox.h:52-62
{
    float dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}

// include/triaccel.h:54-108
void project(TriAccel* ta, const V3& a, const V3& b, const V3& c, int32_t triIdx, int32_t meshIdx)
{
    V3 ab{ b.x - a.x, b.y - a.y, b.z - a.z };
    V3 ac{ c.x - a.x, c.y - a.y, c.z - a.z };
    V3 n{ ab.y * ac.z - ab.z * ac.y, ab.z * ac.x - ab.x * ac.z, ab.x * ac.y - ab.y * ac.x };
    int k = 1;
    if (std::fabs(n.x) > std::fabs(n.y)) {
        if (std::fabs(n.x) > std::fabs(n.z)) k = 0; else k = 2;
    } else {
        if (std::fabs(n.z) > std::fabs(n.y)) k = 2;
    }
    int u = (k + 1) % 3, v = (k + 2) % 3;
    float nk = comp(n, k);
    V3 np{ n.x / nk, n.y / nk, n.z / nk };
    ta->n_u = comp(np, u);
    ta->n_v = comp(np, v);
    ta->n_d = a.x * np.x + a.y * np.y + a.z * np.z;
    ta->k = k;
    float det = comp(ab, u) * comp(ac, v) - comp(ab, v) * comp(ac, u);
    ta->b_u = -comp(ab, v) / det;
    ta->b_v = comp(ab, u) / det;
    ta->b_d = (comp(ab, v) * comp(a, u) - comp(ab, u) * comp(a, v)) / det;
    ta->triIdx = triIdx;
    ta->c_u = comp(ac, v) / det;
    ta->c_v = -comp(ac, u) / det;
    ta->c_d = -(comp(ac, v) * comp(a, u) - comp(ac, u) * comp(a, v)) / det;
    ta->meshIdx = meshIdx;
}

struct BuildNode {            // src/bvhaccel.cpp:32-54
    size_t triStart = 0, numTris = 0;
    Box bounds;
    int splitAxis = 3;
    BuildNode* child[2] = { nullptr, nullptr };
};

struct Scene {
    std::vector<V3> verts;
    std::vector<int32_t> tris;      // 5 per triangle: v0 v1 v2 meshId triId
    std::vector<float> spheres;     // 4 per sphere: cx cy cz r
    std::vector<Node> nodes;        // flattened, reference layout and order
    std::vector<TriAccel> triaccel; // leaf order
    std::vector<uint32_t> leafOrder;// leaf slot -> input triangle index
    std::vector<BuildNode> pool;
    size_t poolUsed = 0;
};

using Iter = std::vector<BoundsInfo>::iterator;

// src/bvhaccel.cpp:100-151
BuildNode* buildRecursive(Scene& sc, Iter begin, Iter end, std::vector<uint32_t>& order,
                          const std::vector<BoundsInfo>& all)
{
    (void)all;
    auto n = std::distance(begin, end);
    BuildNode* node = &sc.pool[sc.poolUsed++];
    if (n < 8) {                                    // :57, :106
        node->triStart = order.size();
        Box bb = emptyBox();
        for (auto it = begin; it != end; ++it) {
            bb = unionBox(bb, it->bounds);
            order.push_back(it->triId);             // triId here = input triangle index
        }
        node->numTris = (size_t)n;
        node->bounds = bb;
        node->splitAxis = 3;
        return node;
    }
    Box bb = emptyBox();
    for (auto it = begin; it != end; ++it) bb = unionBox(bb, it->bounds);
    int axis = maxExtent(bb);
    // :126  float sum, then * 0.5 (double literal) -> double midpoint
    double midpoint = (double)(comp(bb.mx, axis) + comp(bb.mn, axis)) * 0.5;
    Iter middle = std::partition(begin, end, [midpoint, axis](const BoundsInfo& b) {
        return (double)comp(b.center, axis) < midpoint;
    });
    if (middle == begin || middle == end) {
        middle = begin + n / 2;
        std::nth_element(begin, middle, end, [axis](const BoundsInfo& l, const BoundsInfo& r) {
            return comp(l.center, axis) < comp(r.center, axis);
        });
    }
    node->bounds = bb;
    node->splitAxis = axis;
    // g++ evaluates the reference's two argument expressions right to left
    BuildNode* right = buildRecursive(sc, middle, end, order, all);
    BuildNode* left = buildRecursive(sc, begin, middle, order, all);
    node->child[0] = left;
    node->child[1] = right;
    return node;
}

// src/bvhaccel.cpp:215-234
void flatten(std::vector<Node>& out, const BuildNode* n)
{
    Node r;
    std::memset(&r, 0, sizeof r);
    r.b[0] = n->bounds.mn.x; r.b[1] = n->bounds.mn.y; r.b[2] = n->bounds.mn.z;
    r.b[3] = n->bounds.mx.x; r.b[4] = n->bounds.mx.y; r.b[5] = n->bounds.mx.z;
    if (n->splitAxis != 3) {
        r.splitAxis = (uint8_t)n->splitAxis;
        out.push_back(r);
        size_t idx = out.size() - 1;
        flatten(out, n->child[0]);
        out[idx].offset = (uint32_t)out.size();
        flatten(out, n->child[1]);
    } else {
        r.splitAxis = 3;
        r.offset = (uint32_t)n->triStart;
        r.numTriangles = (uint8_t)n->numTris;
        out.push_back(r);
    }
}

// src/bvhaccel.cpp:309-363
void buildScene(Scene& sc)
{
    size_t nT = sc.tris.size() / 5;
    std::vector<BoundsInfo> info;
    info.reserve(nT);
    for (size_t i = 0; i < nT; ++i) {
        const int32_t* t = &sc.tris[i * 5];
        Box b{ sc.verts[t[0]], sc.verts[t[0]] };              // BBox(point), bbox.h:22
        b = unionPoint(b, sc.verts[t[1]]);
        b = unionPoint(b, sc.verts[t[2]]);
        BoundsInfo bi;
        bi.bounds = b;
        // :28  (min + max) * 0.5  -- operator*(float): 0.5 narrows to 0.5f
        bi.center = V3{ (b.mn.x + b.mx.x) * 0.5f, (b.mn.y + b.mx.y) * 0.5f, (b.mn.z + b.mx.z) * 0.5f };
        bi.meshId = (uint32_t)t[3];
        bi.triId = (uint32_t)i;
        info.push_back(bi);
    }
    sc.pool.resize(2 * nT + 2);
    sc.poolUsed = 0;
    sc.leafOrder.clear();
    sc.leafOrder.reserve(nT);
    BuildNode* root = buildRecursive(sc, info.begin(), info.end(), sc.leafOrder, info);
    sc.triaccel.resize(nT);
    for (size_t i = 0; i < nT; ++i) {
        const int32_t* t = &sc.tris[(size_t)sc.leafOrder[i] * 5];
        project(&sc.triaccel[i], sc.verts[t[0]], sc.verts[t[1]], sc.verts[t[2]], t[4], t[3]);
    }
    sc.nodes.clear();
    flatten(sc.nodes, root);
    sc.pool.clear();
    sc.pool.shrink_to_fit();
}

// include/bbox.h:30-50 ; tFar0 is ray.maxT in the reference.  The engine's
// distance-pruned rule passes the current closest t instead.
inline bool slab(const float* b, const Ray& r, float tFar0)
{
    float tNear = r.tmin;
    float tFar = tFar0;
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / r.d[i];
        float t0 = (b[i] - r.o[i]) * inv;
        float t1 = (b[3 + i] - r.o[i]) * inv;
        if (t1 < t0) std::swap(t0, t1);
        if (tNear < t0) tNear = t0;
        if (tFar > t1) tFar = t1;
        if (tFar < tNear) return false;
    }
    return tFar > r.tmin && tNear < r.tmax;
}

// include/triaccel.h:169-201 ; tcur is info->t
inline bool wald(const TriAccel& ta, const Ray& r, float& tcur, float& uOut, float& vOut)
{
    static const int modulo[] = { 1, 2, 0, 1 };
    const int k = ta.k, ku = modulo[k], kv = modulo[k + 1];
    const float nd = 1.0f / (r.d[k] + ta.n_u * r.d[ku] + ta.n_v * r.d[kv]);
    const float t = (ta.n_d - r.o[k] - ta.n_u * r.o[ku] - ta.n_v * r.o[kv]) * nd;
    if (!(tcur > t && t > 0.0001f)) return false;
    const float hu = r.o[ku] + t * r.d[ku];
    const float hv = r.o[kv] + t * r.d[kv];
    const float lambda = hu * ta.b_u + hv * ta.b_v + ta.b_d;
    if (lambda < 0.0f) return false;
    const float mue = hu * ta.c_u + hv * ta.c_v + ta.c_d;
    if (mue < 0.0f) return false;
    if (lambda + mue > 1.0f) return false;
    tcur = t;
    vOut = lambda;
    uOut = mue;
    return true;
}

// include/sphere.h:63-97 ; returns true and narrows t like the reference
inline bool sphere(const float* s, const Ray& r, float& tcur)
{
    static const double EPS_S = 1e-4f;
    float opx = s[0] - r.o[0], opy = s[1] - r.o[1], opz = s[2] - r.o[2];
    double b = opx * r.d[0] + opy * r.d[1] + opz * r.d[2];
    double det = b * b - (opx * opx + opy * opy + opz * opz) + s[3] * s[3];
    if (det < 0) return false;
    det = std::sqrt(det);
    double t = b - det;
    if (t > EPS_S && t < tcur) { tcur = (float)t; return true; }
    t = b + det;
    if (t > EPS_S && t < tcur) { tcur = (float)t; return true; }
    return false;
}

struct Counters { uint32_t nodes = 0, tris = 0, spheres = 0; };

// src/bvhaccel.cpp:236-305 (closest: shadow=false ; any: shadow=true)
template <bool shadow>
bool traverseRef(const Scene& sc, const Ray& r, Hit& h, Counters* c)
{
    size_t sp = 0;
    size_t stack[64];
    size_t cur = 0;
    bool hit = false;
    if (sc.nodes.empty()) return false;
    for (;;) {
        const Node& n = sc.nodes[cur];
        if (c) c->nodes++;
        if (slab(n.b, r, r.tmax)) {
            if (n.splitAxis != 3) {
                if (r.d[n.splitAxis] > 0) { stack[sp] = n.offset; cur = cur + 1; }
                else { stack[sp] = cur + 1; cur = n.offset; }
                sp++;
                continue;
            }
            int tri = -1;
            for (size_t i = 0; i < n.numTriangles; ++i) {
                size_t idx = n.offset + i;
                if (c) c->tris++;
                if (wald(sc.triaccel[idx], r, h.t, h.u, h.v)) {
                    if (shadow) return true;
                    tri = (int)idx;
                }
            }
            if (tri != -1) { hit = true; h.prim = tri; }
        }
        if (sp == 0) return hit;
        cur = stack[--sp];
    }
}

// include/scene.h:136-168
bool sceneIntersect(const Scene& sc, const Ray& r, Hit& h, Counters* c)
{
    h.t = r.tmax; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    size_t nS = sc.spheres.size() / 4;
    for (size_t i = 0; i < nS; ++i) {
        if (c) c->spheres++;
        if (sphere(&sc.spheres[i * 4], r, h.t)) { h.prim = -(int32_t)i - 2; h.u = 0.0f; h.v = 0.0f; }
    }
    traverseRef<false>(sc, r, h, c);
    return h.t < r.tmax;
}

// include/scene.h:170-194
bool sceneIntersectShadow(const Scene& sc, const Ray& r, Counters* c)
{
    size_t nS = sc.spheres.size() / 4;
    float t = r.tmax;
    for (size_t i = 0; i < nS; ++i) {
        if (c) c->spheres++;
        if (sphere(&sc.spheres[i * 4], r, t)) return true;
    }
    Hit h; h.t = r.tmax; h.u = h.v = 0.0f; h.prim = -1;   // src/bvhaccel.cpp:377-378
    return traverseRef<true>(sc, r, h, c);
}

// ---------------------------------------------------------------------------
// The ENGINE's deterministic traversal rule, restated on the CPU so that the
// per-ray fetch counts that define the algorithmic bytes can be
// obtained without a GPU and cross-checked against the counting kernel.
// Same tree, same front-to-back order by split-axis sign as the reference
// (src/bvhaccel.cpp:255-261); differences, all of which leave the result
// unchanged except for the rounding cases documented in DESIGN.md:
//   * both children of an interior node are box-tested when the node is
//     processed (they sit side by side in one 64-byte record on the device),
//   * the slab interval is [minT, current closest t] instead of [minT, maxT]
//     (distance pruning), final acceptance still tests tNear < ray.maxT.
// A "node" count is one 32-byte child record fetched.
// ---------------------------------------------------------------------------
constexpr float kCullSlack = 1.0f + 0x1p-18f;

template <bool shadow>
bool traverseEngine(const Scene& sc, const Ray& r, Hit& h, Counters* c, float padMax)
{
    if (sc.nodes.empty()) return false;
    // as yart_amd/csrc/traverse.hip: pruning allows for the triangles' error bands (cover.h: Cover::padMax)
    float cullPad = padMax * std::fmax(std::fmax(std::fabs(1.0f / r.d[0]), std::fabs(1.0f / r.d[1])), std::fabs(1.0f / r.d[2]));
    if (!(cullPad == cullPad)) cullPad = std::numeric_limits<float>::infinity();
    uint32_t stack[64];
    int sp = 0;
    bool hit = false;
    // the root sits in pair 0 next to an empty leaf: one 64-byte fetch = 2 records
    if (c) c->nodes += 2;
    if (!slab(sc.nodes[0].b, r, shadow ? r.tmax : std::fma(h.t, kCullSlack, cullPad))) return false;
    uint32_t cur = 0;
    for (;;) {
        const Node& n = sc.nodes[cur];
        if (n.splitAxis != 3) {
            uint32_t l = cur + 1, rr = n.offset;
            if (c) c->nodes += 2;
            float bound = shadow ? r.tmax : std::fma(h.t, kCullSlack, cullPad);      // as yart_amd/csrc/traverse.hip
            bool hl = slab(sc.nodes[l].b, r, bound);
            bool hr = slab(sc.nodes[rr].b, r, bound);
            uint32_t first = l, second = rr;
            bool hf = hl, hs = hr;
            if (!(r.d[n.splitAxis] > 0)) { first = rr; second = l; hf = hr; hs = hl; }
            if (hf && hs) { stack[sp++] = second; cur = first; continue; }
            if (hf) { cur = first; continue; }
            if (hs) { cur = second; continue; }
        } else {
            int tri = -1;
            for (size_t i = 0; i < n.numTriangles; ++i) {
                size_t idx = n.offset + i;
                if (c) c->tris++;
                if (wald(sc.triaccel[idx], r, h.t, h.u, h.v)) {
                    if (shadow) return true;
                    tri = (int)idx;
                }
            }
            if (tri != -1) { hit = true; h.prim = tri; }
        }
        if (sp == 0) return hit;
        cur = stack[--sp];
    }
}


// ---------------------------------------------------------------------------
// The engine's 4-wide triangle+sphere tree (yart_amd/csrc/wide_build.h), walked on the CPU from a
// dump of its arrays (yart_accel_get_wide): the deterministic form of the default kernel's rule --
// nearest child first, others pushed far to near, a leaf's primitives tested as soon as the leaf
// is reached, boxes pruned by min(best triangle t, best sphere float t).  The GPU kernel may set a
// leaf aside while it keeps descending (speculative descent), so its fetch counts are >= these; the
// hits are the same.  Node boxes are the engine's own (quantised), so their arithmetic (fma) is
// restated as the kernel does it; primitive tests are wald() / the reference sphere test above.
// ---------------------------------------------------------------------------
struct WideNodeRec {
    float origin[3];
    float scale[3];
    uint8_t qlo[3][4];
    uint8_t qhi[3][4];
    uint32_t ref[4];
};
static_assert(sizeof(WideNodeRec) == 64, "wide node is 64 bytes");

// What the kernel needs beside the tree (yart_amd/csrc/traverse_wide.hip, cover.h): the reference's own nodes (the box of
// a triangle's leaf there decides whether a hit in the triangle's error band counts, and whether two tied triangles
// share a leaf), the cover of the scene (rays that start outside it are not answered by this walk) and the bounds of
// the tree's grids (cull margin).
struct WideAux {
    const Node* refNodes = nullptr;
    const uint32_t* leafOfSlot = nullptr;       // triangle slot -> index of its leaf in refNodes
    float coverLo[3], coverHi[3];
    bool coverValid = false;
    float cullReach = 0.0f;                     // traverse.h: WideParams::cullReach (any bound of it serves)
};

// Which of two leaves (indices a != b into the reference's pre-order node array) the reference's walk reaches first
// (src/bvhaccel.cpp:255-261): at their lowest common ancestor the left subtree iff dir[splitAxis] > 0.
inline uint32_t tiedFirst(const Node* nodes, uint32_t a, uint32_t b, const Ray& r)
{
    const uint32_t lo = std::min(a, b), hi = std::max(a, b);
    uint32_t i = 0;
    while (nodes[i].splitAxis != 3) {
        const uint32_t right = nodes[i].offset;
        if (hi < right) i = i + 1;
        else if (lo >= right) i = right;
        else return r.d[nodes[i].splitAxis] > 0 ? lo : hi;
    }
    return lo;
}
constexpr int32_t kUnresolved = INT32_MIN;      // yart_amd/csrc/traverse.h kUnresolvedPrim; any-hit: 2

inline float bandOfClass(uint32_t cls)          // cover.h
{
    const uint32_t bits = (106u + 3u * cls) << 23;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

inline bool regularRay(const Ray& r)            // yart_amd/csrc/traverse.h
{
    const float inf = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; ++a) {
        const float i = 1.0f / r.d[a];
        if (!(i != 0.0f && std::fabs(i) < inf && std::fabs(r.o[a]) < inf)) return false;
    }
    return r.tmin == 0.0f && r.tmax == r.tmax;
}

// returns 0 = miss / not occluded, 1 = hit / occluded, 2 = left to the walk of the reference's tree
template <bool shadow>
int traverseWide(const WideNodeRec* nodes, const float* rowsA, const float* rowsBC, const WideAux& aux, const Ray& r, Hit& out, Counters* c)
{
    const uint32_t kNone = 0xFFFFFFFFu, kLeaf = 0x80000000u, kSph = 0x40000000u;
    const float inf = std::numeric_limits<float>::infinity();
    if (!(r.tmax > 0.0001f) && r.tmax == r.tmax) {        // as traverse_wide.hip: nothing can be accepted
        out.t = r.tmax; out.u = out.v = 0.0f; out.prim = -1;
        return 0;
    }
    bool covered = true;
    if (aux.coverValid)
        for (int a = 0; a < 3; ++a) covered = covered && r.o[a] >= aux.coverLo[a] && r.o[a] <= aux.coverHi[a];
    if (!regularRay(r) || !covered) { out.t = 0.0f; out.u = out.v = 0.0f; out.prim = kUnresolved; return 2; }
    const float ix = 1.0f / r.d[0], iy = 1.0f / r.d[1], iz = 1.0f / r.d[2];
    const float inv[3] = { ix, iy, iz };
    const float imax = std::fmax(std::fmax(std::fabs(ix), std::fabs(iy)), std::fabs(iz));
    if (!(imax * aux.cullReach < inf)) { out.t = 0.0f; out.u = out.v = 0.0f; out.prim = kUnresolved; return 2; }
    const float margin = 0x1p-21f * aux.cullReach;      // per unit of |1 / d|: entry planes are taken that much earlier, per axis
    float tTri = r.tmax, hu = 0.0f, hv = 0.0f, F = inf;
    int32_t primTri = -1, minS = -1, maxL = -1;
    uint32_t stack[128];
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        if (!(cur & kLeaf)) {
            const WideNodeRec& n = nodes[cur];
            if (c) c->nodes += 4;
            const float bound = (shadow ? r.tmax : fmin2(tTri, F)) * kCullSlack;     // as traverse_wide.hip
            float tn[4];
            uint32_t rf[4];
            int nHit = 0;
            for (int ch = 0; ch < 4; ++ch) {
                float tNear = 0.0f, tFar = bound;
                for (int a = 0; a < 3; ++a) {
                    const float A = (n.origin[a] - r.o[a]) * inv[a], B = n.scale[a] * inv[a];
                    const bool neg = inv[a] < 0.0f;
                    const float qn = (float)(neg ? n.qhi[a][ch] : n.qlo[a][ch]), qf = (float)(neg ? n.qlo[a][ch] : n.qhi[a][ch]);
                    tNear = std::fmax(tNear, std::fma(qn, B, std::fma(std::fabs(inv[a]), -margin, A)));
                    tFar = std::fmin(tFar, std::fma(qf, B, A));
                }
                const bool h = !(tFar < tNear) && tNear < inf && n.ref[ch] != kNone;     // as traverse_wide.hip slab4 + its +inf marker
                tn[ch] = h ? tNear : inf;
                rf[ch] = n.ref[ch];
                nHit += h ? 1 : 0;
            }
            auto cswap = [&](int i, int j) { if (tn[j] < tn[i]) { std::swap(tn[i], tn[j]); std::swap(rf[i], rf[j]); } };
            cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
            for (int k = nHit - 1; k >= 1; --k) stack[sp++] = rf[k];
            if (nHit > 0) { cur = rf[0]; continue; }
        } else {
            const uint32_t cnt = ((cur >> 27) & 7u) + 1u, first = cur & 0x07FFFFFFu;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float* A = rowsA + (size_t)(first + i) * 4;
                const float* BC = rowsBC + (size_t)(first + i) * 8;
                if (!(cur & kSph)) {
                    if (c) c->tris++;
                    uint32_t kw;
                    std::memcpy(&kw, A + 3, 4);
                    TriAccel ta;
                    ta.n_u = A[0]; ta.n_v = A[1]; ta.n_d = A[2]; ta.k = (int32_t)(kw & 3u);
                    ta.b_u = BC[0]; ta.b_v = BC[1]; ta.b_d = BC[2];
                    ta.c_u = BC[4]; ta.c_v = BC[5]; ta.c_d = BC[6];
                    const int32_t slot = (int32_t)((kw & 0x1FFFFFFCu) >> 2);
                    const uint32_t cls = kw >> 29;
                    // a hit within the triangle's error band of an edge counts only if the ray passes the box of the
                    // triangle's leaf in the reference's tree (traverse_wide.hip, cover.h)
                    auto counts = [&](float u, float v) {
                        if (cls == 0 || !aux.refNodes) return true;
                        if (!(fmin2(fmin2(v, u), 1.0f - (v + u)) < bandOfClass(cls))) return true;
                        return slab(aux.refNodes[aux.leafOfSlot[slot]].b, r, r.tmax);
                    };
                    float bound = shadow ? r.tmax : fmin2(tTri, F);
                    float u, v, t = bound;
                    if (wald(ta, r, t, u, v)) {
                        if (!counts(u, v)) continue;
                        if (shadow) return 1;
                        tTri = t; hu = u; hv = v; primTri = slot;
                    } else if (!shadow && primTri >= 0 && slot != primTri) {
                        // an exact-t tie with the triangle held: inside one leaf of the reference's tree the lower slot
                        // wins; across two of its leaves the one its walk reaches first (see traverse_wide.hip)
                        float t2 = std::nextafter(tTri, inf);
                        if (wald(ta, r, t2, u, v) && t2 == tTri && counts(u, v)) {
                            bool first = slot < primTri;
                            if (aux.refNodes && aux.leafOfSlot[slot] != aux.leafOfSlot[primTri])
                                first = tiedFirst(aux.refNodes, aux.leafOfSlot[slot], aux.leafOfSlot[primTri], r) == aux.leafOfSlot[slot];
                            if (first) { hu = u; hv = v; primTri = slot; }
                        }
                    }
                } else {
                    if (c) c->spheres++;
                    const float opx = A[0] - r.o[0], opy = A[1] - r.o[1], opz = A[2] - r.o[2];
                    const double b = opx * r.d[0] + opy * r.d[1] + opz * r.d[2];
                    double det = b * b - (opx * opx + opy * opy + opz * opz) + A[3] * A[3];
                    if (det < 0) continue;
                    det = std::sqrt(det);
                    const double eps = 1e-4f;
                    double t = b - det;
                    if (!(t > eps)) t = b + det;
                    if (!(t > eps && t < (double)r.tmax)) continue;
                    if (shadow) return 1;
                    const float f = (float)t;
                    if (f > fmin2(tTri, F)) continue;
                    uint32_t idxBits;
                    std::memcpy(&idxBits, BC, 4);
                    const int32_t idx = (int32_t)idxBits;
                    const bool below = t < (double)f;
                    if (minS < 0 || f < F) { F = f; minS = idx; maxL = below ? idx : -1; }
                    else { minS = std::min(minS, idx); if (below && idx > maxL) maxL = idx; }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (shadow) return 0;
    const bool sph = (minS >= 0) && !(tTri < F);
    out.t = sph ? F : tTri;
    out.prim = sph ? -(std::max(minS, maxL) + 2) : primTri;
    out.u = sph ? 0.0f : hu;
    out.v = sph ? 0.0f : hv;
    if (!(out.t < r.tmax)) out.prim = -1;
    if (out.prim == -1) { out.u = 0.0f; out.v = 0.0f; }
    return out.prim != -1 ? 1 : 0;
}

template <class F>
void parallelFor(size_t n, int nthreads, F&& f)
{
    if (nthreads <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    std::atomic<size_t> next{ 0 };
    const size_t chunk = 1024;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&]() {
            for (;;) {
                size_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk));
            }
        });
    }
    for (auto& t : th) t.join();
}

} // namespace

extern "C" {

void* orc_scene_create(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                       const float* spheres, size_t nSpheres)
{
    auto* sc = new Scene();
    sc->verts.resize(nVerts);
    if (nVerts) std::memcpy(sc->verts.data(), verts, nVerts * 12);
    sc->tris.assign(tris, tris + nTris * 5);
    sc->spheres.assign(spheres, spheres + nSpheres * 4);
    for (size_t i = 0; i < nTris * 5; i += 5)
        for (int c = 0; c < 3; ++c)
            if (tris[i + c] < 0 || (size_t)tris[i + c] >= nVerts) { delete sc; return nullptr; }
    buildScene(*sc);
    return sc;
}

void orc_scene_destroy(void* h) { delete static_cast<Scene*>(h); }
size_t orc_num_nodes(void* h) { return static_cast<Scene*>(h)->nodes.size(); }
void orc_get_nodes(void* h, void* out)
{
    auto* sc = static_cast<Scene*>(h);
    std::memcpy(out, sc->nodes.data(), sc->nodes.size() * sizeof(Node));
}
void orc_get_triaccel(void* h, void* out)
{
    auto* sc = static_cast<Scene*>(h);
    std::memcpy(out, sc->triaccel.data(), sc->triaccel.size() * sizeof(TriAccel));
}
void orc_get_leaf_order(void* h, uint32_t* out)
{
    auto* sc = static_cast<Scene*>(h);
    std::memcpy(out, sc->leafOrder.data(), sc->leafOrder.size() * 4);
}

// mode 0: reference rule (Scene::intersect exactly)
// mode 1: engine rule for the triangle BVH (pruned, pair-wise box tests),
//         spheres still the reference's linear loop
// counts (optional): 4 x uint32 per ray = triangle-tree node records fetched,
//         triangle records tested, 0 (the oracle has no sphere tree), spheres tested
// padMax: Cover::padMax of the scene (yart_scene_cover), mode 1 only
void orc_intersect(void* h, const float* rays, size_t n, void* hitsOut, uint32_t* counts,
                   int mode, int nthreads, float padMax)
{
    const Scene& sc = *static_cast<Scene*>(h);
    Hit* out = static_cast<Hit*>(hitsOut);
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray r;
            std::memcpy(&r, rays + i * 8, 32);
            Hit hit;
            Counters c;
            if (mode == 0) {
                sceneIntersect(sc, r, hit, counts ? &c : nullptr);
            } else {
                hit.t = r.tmax; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
                size_t nS = sc.spheres.size() / 4;
                for (size_t s = 0; s < nS; ++s) {
                    c.spheres++;
                    if (sphere(&sc.spheres[s * 4], r, hit.t)) { hit.prim = -(int32_t)s - 2; hit.u = hit.v = 0.0f; }
                }
                traverseEngine<false>(sc, r, hit, counts ? &c : nullptr, padMax);
            }
            if (hit.prim == -1) { hit.u = 0.0f; hit.v = 0.0f; }
            out[i] = hit;
            if (counts) { counts[i * 4] = c.nodes; counts[i * 4 + 1] = c.tris; counts[i * 4 + 2] = 0; counts[i * 4 + 3] = c.spheres; }
        }
    });
}

void orc_intersect_shadow(void* h, const float* rays, size_t n, uint8_t* out, uint32_t* counts,
                          int mode, int nthreads, float padMax)
{
    const Scene& sc = *static_cast<Scene*>(h);
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray r;
            std::memcpy(&r, rays + i * 8, 32);
            Counters c;
            bool occ;
            if (mode == 0) {
                occ = sceneIntersectShadow(sc, r, counts ? &c : nullptr);
            } else {
                occ = false;
                size_t nS = sc.spheres.size() / 4;
                float t = r.tmax;
                for (size_t s = 0; s < nS && !occ; ++s) { c.spheres++; occ = sphere(&sc.spheres[s * 4], r, t); }
                if (!occ) {
                    Hit hh; hh.t = r.tmax; hh.u = hh.v = 0.0f; hh.prim = -1;
                    occ = traverseEngine<true>(sc, r, hh, counts ? &c : nullptr, padMax);
                }
            }
            out[i] = occ ? 1 : 0;
            if (counts) { counts[i * 4] = c.nodes; counts[i * 4 + 1] = c.tris; counts[i * 4 + 2] = 0; counts[i * 4 + 3] = c.spheres; }
        }
    });
}

// Stand-alone pieces for unit tests
// The engine's wide tree walked on the CPU (see traverseWide).  Rays the kernel leaves to the fix-up launches on the
// reference's tree (not regular, outside the cover, tied across two reference leaves) come back marked: prim =
// INT32_MIN / occlusion byte 2.
// refNodes / nRefNodes (optional): the reference's flattened nodes (yart_accel_get_nodes); cover (optional): the ten
// words of yart_scene_cover.  Without them the walk runs without the near-edge check, the tie rule falls back to "lower
// slot", and every origin counts as covered.
// any != 0: occlusion query, out = uint8 per ray; else out = Hit per ray.
// counts (optional): 4 x uint32 per ray = child records fetched, triangles tested, 0, spheres tested.
void orc_wide_intersect(const void* nodes, const float* rowsA, const float* rowsBC, const float* rays, size_t n,
                        void* out, uint32_t* counts, int any, int nthreads, const void* refNodes, size_t nRefNodes,
                        const float* cover)
{
    const WideNodeRec* nd = static_cast<const WideNodeRec*>(nodes);
    WideAux aux;
    std::vector<uint32_t> leafOfSlot;
    if (refNodes && nRefNodes) {
        aux.refNodes = static_cast<const Node*>(refNodes);
        size_t nSlots = 0;
        for (size_t i = 0; i < nRefNodes; ++i)
            if (aux.refNodes[i].splitAxis == 3) nSlots = std::max(nSlots, (size_t)aux.refNodes[i].offset + aux.refNodes[i].numTriangles);
        leafOfSlot.assign(nSlots, 0u);
        for (size_t i = 0; i < nRefNodes; ++i)
            if (aux.refNodes[i].splitAxis == 3)
                for (uint32_t k = 0; k < aux.refNodes[i].numTriangles; ++k) leafOfSlot[aux.refNodes[i].offset + k] = (uint32_t)i;
        aux.leafOfSlot = leafOfSlot.data();
    }
    if (cover) {
        uint32_t valid;
        std::memcpy(&valid, cover + 9, 4);
        aux.coverValid = valid != 0;
        for (int a = 0; a < 3; ++a) { aux.coverLo[a] = cover[a]; aux.coverHi[a] = cover[3 + a]; }
    }
    {
        // bounds of the root's child boxes: every grid of the tree lies inside (the kernel takes the builders' figure;
        // the margin only has to be wide enough, the result does not depend on it)
        const float inf = std::numeric_limits<float>::infinity();
        float gridLo[3], gridHi[3];
        for (int a = 0; a < 3; ++a) { gridLo[a] = inf; gridHi[a] = -inf; }
        for (int ch = 0; ch < 4; ++ch) {
            if (nd[0].ref[ch] == 0xFFFFFFFFu) continue;
            for (int a = 0; a < 3; ++a) {
                gridLo[a] = std::fmin(gridLo[a], nd[0].origin[a] + nd[0].qlo[a][ch] * nd[0].scale[a]);
                gridHi[a] = std::fmax(gridHi[a], nd[0].origin[a] + nd[0].qhi[a][ch] * nd[0].scale[a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            if (!(gridLo[a] <= gridHi[a])) gridLo[a] = gridHi[a] = 0.0f;
            const float lo = aux.coverValid ? std::fmin(aux.coverLo[a], gridLo[a]) : gridLo[a];
            const float hi = aux.coverValid ? std::fmax(aux.coverHi[a], gridHi[a]) : gridHi[a];
            aux.cullReach = std::fmax(aux.cullReach, (hi - lo) + 1.25f * (gridHi[a] - gridLo[a]));
        }
    }
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray r;
            std::memcpy(&r, rays + i * 8, 32);
            Hit hit;
            hit.t = r.tmax; hit.u = hit.v = 0.0f; hit.prim = -1;
            Counters c;
            if (any) static_cast<uint8_t*>(out)[i] = (uint8_t)traverseWide<true>(nd, rowsA, rowsBC, aux, r, hit, counts ? &c : nullptr);
            else { traverseWide<false>(nd, rowsA, rowsBC, aux, r, hit, counts ? &c : nullptr); static_cast<Hit*>(out)[i] = hit; }
            if (counts) { counts[i * 4] = c.nodes; counts[i * 4 + 1] = c.tris; counts[i * 4 + 2] = 0; counts[i * 4 + 3] = c.spheres; }
        }
    });
}

// Tone map of the reference's bitmap writer (src/bitmap.cpp:8-12), one channel: the expression as
// the reference writes it.  Parity unpinned -- src/bitmap.cpp does not compile with g++ 11 (:64), so
// this restatement is all there is to check the product's film output against.
uint8_t orc_tonemap(float val)
{
    static const uint8_t max = 255;
    return std::min(max, (uint8_t)(std::pow(1 - std::exp(-val), 1.0f / 2.2f) * 255 + 0.5f));
}

void orc_project(const float* a, const float* b, const float* c, int32_t triIdx, int32_t meshIdx, void* out)
{
    project(static_cast<TriAccel*>(out), V3{ a[0], a[1], a[2] }, V3{ b[0], b[1], b[2] }, V3{ c[0], c[1], c[2] },
            triIdx, meshIdx);
}
int orc_slab(const float* box6, const float* ray8)
{
    Ray r; std::memcpy(&r, ray8, 32);
    return slab(box6, r, r.tmax) ? 1 : 0;
}
int orc_wald(const void* ta, const float* ray8, float* tuv)
{
    Ray r; std::memcpy(&r, ray8, 32);
    float t = tuv[0], u = tuv[1], v = tuv[2];
    bool ok = wald(*static_cast<const TriAccel*>(ta), r, t, u, v);
    tuv[0] = t; tuv[1] = u; tuv[2] = v;
    return ok ? 1 : 0;
}
int orc_sphere(const float* s4, const float* ray8, float* t)
{
    Ray r; std::memcpy(&r, ray8, 32);
    return sphere(s4, r, *t) ? 1 : 0;
}

} // extern "C"
