// Host build of the 4-wide triangle+sphere BVH (wide_build.h): binned-SAH binary build on a small
// thread pool, collapse to four children per node, depth-first emission of 64-byte quantised nodes and
// two-stage leaf rows.
#include "wide_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>

namespace yart {
namespace {

constexpr uint32_t kSphereId = 0x80000000u;
constexpr int kBins = 16;
constexpr uint32_t kSpawnAbove = 1u << 15;      // ranges larger than this are split off as pool jobs
constexpr uint32_t kOneAxisAbove = 1u << 20;    // very large ranges: bin the widest axis only

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        const float inf = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) { lo[a] = inf; hi[a] = -inf; }
    }
    void grow(const float* l, const float* h)
    {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    float halfArea() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f)) return 0.0f;      // empty
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BPrim {
    float lo[3], hi[3], c[3];
    uint32_t id;                // input triangle number, or kSphereId | sphere number
};

struct BNode {
    Box box;
    uint32_t left = 0, right = 0;       // interior (count == 0)
    uint32_t begin = 0, count = 0;      // leaf: prims[begin, begin + count), one kind
};

class BinaryBuilder {
public:
    BinaryBuilder(std::vector<BPrim>& prims, int maxLeaf, bool medianOnly)
        : prims_(prims), maxLeaf_(maxLeaf), medianOnly_(medianOnly)
    {
        nodes_.resize(std::max<size_t>(1, 2 * prims.size() + 2));
    }

    void run()
    {
        next_ = 1;
        pending_ = 1;
        jobs_.push_back(Job{ 0u, 0u, static_cast<uint32_t>(prims_.size()) });
        unsigned nThreads = std::thread::hardware_concurrency();
        if (const char* v = std::getenv("OMP_NUM_THREADS")) { const int n = std::atoi(v); if (n > 0) nThreads = std::min<unsigned>(nThreads, n); }
        nThreads = std::max(1u, std::min(nThreads, 32u));
        if (prims_.size() < kSpawnAbove) nThreads = 1;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back([this]() { work(); });
        work();
        for (auto& t : pool) t.join();
        nodes_.resize(next_);
    }

    std::vector<BNode>& nodes() { return nodes_; }

private:
    struct Job { uint32_t node, begin, end; };

    void work()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this]() { return !jobs_.empty() || pending_ == 0; });
                if (jobs_.empty()) return;
                j = jobs_.back();
                jobs_.pop_back();
            }
            build(j.node, j.begin, j.end, true);
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) cv_.notify_all();
            }
        }
    }

    void spawn(uint32_t node, uint32_t begin, uint32_t end)
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            ++pending_;
            jobs_.push_back(Job{ node, begin, end });
        }
        cv_.notify_one();
    }

    void build(uint32_t node, uint32_t begin, uint32_t end, bool maySpawn)
    {
        Box box, cbox;
        box.reset();
        cbox.reset();
        for (uint32_t i = begin; i < end; ++i) {
            box.grow(prims_[i].lo, prims_[i].hi);
            cbox.grow(prims_[i].c, prims_[i].c);
        }
        BNode& n = nodes_[node];
        n.box = box;
        const uint32_t count = end - begin;
        uint32_t mid;
        if (count <= static_cast<uint32_t>(maxLeaf_)) {
            // a leaf holds one kind of primitive; a mixed remainder becomes two leaves
            auto first = prims_.begin() + begin, last = prims_.begin() + end;
            auto m = std::partition(first, last, [](const BPrim& p) { return (p.id & kSphereId) == 0; });
            if (m == first || m == last) {
                n.begin = begin;
                n.count = count;
                return;
            }
            mid = static_cast<uint32_t>(m - prims_.begin());
        } else {
            mid = split(begin, end, cbox);
        }
        const uint32_t l = next_.fetch_add(2);
        nodes_[node].left = l;
        nodes_[node].right = l + 1;
        nodes_[node].count = 0;
        if (maySpawn && count > kSpawnAbove) {
            spawn(l, begin, mid);
            spawn(l + 1, mid, end);
        } else {
            build(l, begin, mid, false);
            build(l + 1, mid, end, false);
        }
    }

    uint32_t median(uint32_t begin, uint32_t end, int axis)
    {
        const uint32_t mid = begin + (end - begin) / 2;
        std::nth_element(prims_.begin() + begin, prims_.begin() + mid, prims_.begin() + end,
                         [axis](const BPrim& a, const BPrim& b) { return a.c[axis] < b.c[axis]; });
        return mid;
    }

    uint32_t split(uint32_t begin, uint32_t end, const Box& cbox)
    {
        const uint32_t count = end - begin;
        float ext[3];
        int widest = 0;
        for (int a = 0; a < 3; ++a) {
            ext[a] = cbox.hi[a] - cbox.lo[a];
            if (ext[a] > ext[widest]) widest = a;
        }
        if (!(ext[widest] > 0.0f) || !(ext[widest] < std::numeric_limits<float>::infinity()))
            return begin + count / 2;                    // all centroids coincide (or are not finite)
        if (medianOnly_) return median(begin, end, widest);

        int bestAxis = -1, bestSplit = 0;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
            if (count > kOneAxisAbove && a != widest) continue;
            if (!(ext[a] > 0.0f)) continue;
            Box bins[kBins];
            uint32_t cnt[kBins] = { 0 };
            for (auto& b : bins) b.reset();
            const float scale = static_cast<float>(kBins) / ext[a];
            for (uint32_t i = begin; i < end; ++i) {
                int b = static_cast<int>((prims_[i].c[a] - cbox.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bins[b].grow(prims_[i].lo, prims_[i].hi);
                ++cnt[b];
            }
            float rightArea[kBins];
            uint32_t rightCnt[kBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bins[b]);
                c += cnt[b];
                rightArea[b] = acc.halfArea();
                rightCnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 1; b < kBins; ++b) {             // split s: bins [0,s) left, [s,kBins) right
                acc.grow(bins[b - 1]);
                c += cnt[b - 1];
                if (c == 0 || rightCnt[b] == 0) continue;
                const float cost = acc.halfArea() * static_cast<float>(c) + rightArea[b] * static_cast<float>(rightCnt[b]);
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
            }
        }
        if (bestAxis < 0) return median(begin, end, widest);
        const float lo = cbox.lo[bestAxis], scale = static_cast<float>(kBins) / ext[bestAxis];
        const int s = bestSplit, axis = bestAxis;
        auto m = std::partition(prims_.begin() + begin, prims_.begin() + end, [=](const BPrim& p) {
            int b = static_cast<int>((p.c[axis] - lo) * scale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            return b < s;
        });
        const uint32_t mid = static_cast<uint32_t>(m - prims_.begin());
        if (mid == begin || mid == end) return median(begin, end, widest);
        return mid;
    }

    std::vector<BPrim>& prims_;
    std::vector<BNode> nodes_;
    std::atomic<uint32_t> next_{ 1 };
    int maxLeaf_;
    bool medianOnly_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Job> jobs_;
    int pending_ = 0;
};

float bitsFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

class Emitter {
public:
    Emitter(const std::vector<BNode>& nodes, const std::vector<BPrim>& prims, const TriangleBvh& tri,
            const float* spheres, WideBvh& out)
        : nodes_(nodes), prims_(prims), tri_(tri), spheres_(spheres), out_(out) { }

    bool tooManyPrims = false;

    void run()
    {
        out_.nodes.clear();
        out_.rowsA.clear();
        out_.rowsBC.clear();
        out_.depth = 0;
        out_.numLeaves = 0;
        size_t nTri = 0, nSph = 0;
        for (const auto& p : prims_) ((p.id & kSphereId) ? nSph : nTri)++;
        out_.rowsA.reserve((nTri + nSph) * 4);
        out_.rowsBC.reserve((nTri + nSph) * 8);
        out_.nodes.reserve(prims_.size() / 2 + 16);
        if (prims_.empty()) {
            out_.nodes.emplace_back();
            clear(out_.nodes[0]);
            out_.depth = 1;
            return;
        }
        if (nodes_[0].count > 0) {                       // the whole scene fits one leaf
            out_.nodes.emplace_back();
            const uint32_t ref = leaf(nodes_[0]);
            setChildren(0, &nodes_[0].box, &ref, 1);
            out_.depth = 1;
            return;
        }
        emit(0, 1);
    }

private:
    static void clear(WideNode& n)
    {
        std::memset(&n, 0, sizeof n);
        for (int a = 0; a < 3; ++a) {
            n.scale[a] = 1.0f;
            for (int c = 0; c < 4; ++c) { n.qlo[a][c] = 255; n.qhi[a][c] = 0; }
        }
        for (auto& r : n.ref) r = kWideNone;
    }

    // Quantise up to four child boxes on the grid of their union.  Containment is decided in double
    // on the float values the kernel will read: origin + qlo * scale <= lo and origin + qhi * scale >= hi.
    void setChildren(uint32_t node, const Box* boxesIn, const uint32_t* refs, int n)
    {
        WideNode& w = out_.nodes[node];
        clear(w);
        Box all, clamped[4];
        all.reset();
        for (int i = 0; i < n; ++i) {
            clamped[i] = boxesIn[i];                    // keep the grid arithmetic finite
            for (int a = 0; a < 3; ++a) {
                clamped[i].lo[a] = std::max(clamped[i].lo[a], -8.0e37f);
                clamped[i].hi[a] = std::min(clamped[i].hi[a], 8.0e37f);
            }
            all.grow(clamped[i]);
        }
        const Box* boxes = clamped;
        for (int a = 0; a < 3; ++a) {
            const float origin = all.lo[a];
            float scale = (all.hi[a] - all.lo[a]) / 255.0f;
            if (!(scale > 0.0f) || !(scale < std::numeric_limits<float>::infinity())) scale = 1.0f;   // flat or unbounded axis
            for (;;) {
                bool ok = true;
                for (int i = 0; i < n && ok; ++i) {
                    const double lo = boxes[i].lo[a], hi = boxes[i].hi[a];
                    double ql = std::floor((lo - origin) / scale), qh = std::ceil((hi - origin) / scale);
                    while (ql > 0.0 && origin + ql * static_cast<double>(scale) > lo) ql -= 1.0;
                    while (origin + qh * static_cast<double>(scale) < hi) qh += 1.0;
                    if (ql < 0.0) ql = 0.0;
                    if (qh > 255.0) { ok = false; break; }
                    w.qlo[a][i] = static_cast<uint8_t>(ql);
                    w.qhi[a][i] = static_cast<uint8_t>(qh);
                }
                if (ok) break;
                scale = std::nextafter(scale * 1.000001f, std::numeric_limits<float>::infinity());
            }
            w.origin[a] = origin;
            w.scale[a] = scale;
        }
        for (int i = 0; i < n; ++i) w.ref[i] = refs[i];
    }

    uint32_t emit(uint32_t bnode, int depth)
    {
        out_.depth = std::max(out_.depth, depth);
        uint32_t kids[4] = { nodes_[bnode].left, nodes_[bnode].right, 0, 0 };
        int n = 2;
        while (n < 4) {                                  // open the largest interior child
            int best = -1;
            float bestArea = -1.0f;
            for (int i = 0; i < n; ++i) {
                if (nodes_[kids[i]].count > 0) continue;
                const float a = nodes_[kids[i]].box.halfArea();
                if (a > bestArea) { bestArea = a; best = i; }
            }
            if (best < 0) break;
            const uint32_t open = kids[best];
            kids[best] = nodes_[open].left;
            kids[n++] = nodes_[open].right;
        }
        const uint32_t idx = static_cast<uint32_t>(out_.nodes.size());
        out_.nodes.emplace_back();
        Box boxes[4];
        uint32_t refs[4];
        for (int i = 0; i < n; ++i) {
            const BNode& c = nodes_[kids[i]];
            boxes[i] = c.box;
            refs[i] = (c.count > 0) ? leaf(c) : emit(kids[i], depth + 1);
        }
        setChildren(idx, boxes, refs, n);
        return idx;
    }

    uint32_t leaf(const BNode& n)
    {
        const size_t first = out_.rowsA.size() / 4;
        if (first + n.count > kWideMaxPrims) { tooManyPrims = true; return kWideNone; }
        const uint32_t cnt = n.count;
        const bool sph = (prims_[n.begin].id & kSphereId) != 0;
        uint32_t ids[kWideMaxLeaf];
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t id = prims_[n.begin + i].id;
            ids[i] = sph ? (id & ~kSphereId) : tri_.slotOfInput[id];
        }
        std::sort(ids, ids + cnt);                       // in-leaf test order: lowest slot / sphere number first
        out_.rowsA.resize((first + cnt) * 4, 0.0f);
        out_.rowsBC.resize((first + cnt) * 8, 0.0f);
        for (uint32_t i = 0; i < cnt; ++i) {
            float* A = out_.rowsA.data() + (first + i) * 4;
            float* B = out_.rowsBC.data() + (first + i) * 8;
            float* C = B + 4;
            if (sph) {
                std::memcpy(A, spheres_ + static_cast<size_t>(ids[i]) * 4, 16);
                B[0] = bitsFloat(ids[i]);
            } else {
                const YartTriAccel& ta = tri_.triaccel[ids[i]];
                A[0] = ta.n_u; A[1] = ta.n_v; A[2] = ta.n_d; A[3] = bitsFloat(static_cast<uint32_t>(ta.k) | (ids[i] << 2));
                B[0] = ta.b_u; B[1] = ta.b_v; B[2] = ta.b_d;
                C[0] = ta.c_u; C[1] = ta.c_v; C[2] = ta.c_d;
            }
        }
        ++out_.numLeaves;
        return kWideLeafBit | (sph ? kWideSphereBit : 0u) | ((cnt - 1) << 27) | static_cast<uint32_t>(first);
    }

    const std::vector<BNode>& nodes_;
    const std::vector<BPrim>& prims_;
    const TriangleBvh& tri_;
    const float* spheres_;
    WideBvh& out_;
};

} // namespace

int buildWideBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                 const TriangleBvh& tri, const float* spheres, size_t nSpheres,
                 int maxLeaf, WideBvh& out, std::string& err)
{
    (void)nVerts;
    if (maxLeaf < 1) maxLeaf = 1;
    if (maxLeaf > kWideMaxLeaf) maxLeaf = kWideMaxLeaf;
    if (nTris >= (1u << 30) || nSpheres >= (1u << 30)) {
        err = "wide BVH: more than 2^30 primitives of one kind";
        return YART_ERR_INVALID_ARGUMENT;
    }
    std::vector<BPrim> prims(nTris + nSpheres);
    for (size_t i = 0; i < nTris; ++i) {
        const int32_t* t = tris + i * 5;
        BPrim& p = prims[i];
        const float* v0 = verts + static_cast<size_t>(t[0]) * 3;
        std::memcpy(p.lo, v0, 12);
        std::memcpy(p.hi, v0, 12);
        for (int k = 1; k < 3; ++k) {
            const float* v = verts + static_cast<size_t>(t[k]) * 3;
            for (int a = 0; a < 3; ++a) { p.lo[a] = std::min(p.lo[a], v[a]); p.hi[a] = std::max(p.hi[a], v[a]); }
        }
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.lo[a] + 0.5f * p.hi[a];
        p.id = static_cast<uint32_t>(i);
    }
    for (size_t i = 0; i < nSpheres; ++i) {
        // extents widened by a relative margin: the fp32 slab test must never reject a ray the
        // fp64 sphere test (include/sphere.h:63-97) accepts
        const float* s = spheres + i * 4;
        const float r = std::fabs(s[3]);
        BPrim& p = prims[nTris + i];
        for (int a = 0; a < 3; ++a) {
            const float margin = 1e-5f * (std::fabs(s[a]) + r) + 1e-30f;
            p.lo[a] = s[a] - r - margin;
            p.hi[a] = s[a] + r + margin;
            p.c[a] = s[a];
        }
        p.id = kSphereId | static_cast<uint32_t>(i);
    }
    // NaN coordinates would poison the binning arithmetic; such primitives can never be hit
    // (every comparison of their tests fails), so they are left out of the tree.
    prims.erase(std::remove_if(prims.begin(), prims.end(), [](const BPrim& p) {
        for (int a = 0; a < 3; ++a) if (!(p.lo[a] <= p.hi[a])) return true;
        return false;
    }), prims.end());

    const bool timing = std::getenv("YART_BUILD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int attempt = 0; attempt < 2; ++attempt) {
        const double t0 = now();
        BinaryBuilder b(prims, maxLeaf, attempt == 1);
        if (!prims.empty()) b.run();
        const double t1 = now();
        Emitter e(b.nodes(), prims, tri, spheres, out);
        e.run();
        if (timing) fprintf(stderr, "wide build: binary SAH %.2fs, collapse+emit %.2fs (%zu nodes, depth %d)\n", t1 - t0, now() - t1, out.nodes.size(), out.depth);
        if (e.tooManyPrims) {
            err = "wide BVH: more than 2^27 - 1 primitives";
            return YART_ERR_INVALID_ARGUMENT;
        }
        if (3 * out.depth + 1 <= kWideStack) return YART_OK;
        // SAH produced a tree too deep for the traversal stack: rebuild balanced (object median)
    }
    err = "wide BVH deeper than the traversal stack allows";
    return YART_ERR_TREE_TOO_DEEP;
}

} // namespace yart
