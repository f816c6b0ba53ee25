// Host build of the 4-wide triangle+sphere BVH (wide_build.h): binned-SAH binary build on a small
// thread pool, collapse to four children per node, depth-first emission of 64-byte quantised nodes and
// two-stage leaf rows.
#include "wide_build.h"
#include "cover.h"
#include "task_pool.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>

namespace yart {
namespace {

constexpr uint32_t kSphereId = 0x80000000u;
constexpr int kMaxBins = 64;
int envInt(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
constexpr uint32_t kSpawnAbove = 1u << 15;      // ranges larger than this are split off as pool jobs

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
    uint64_t code;              // Morton code of the centre (YART_WIDE_BUILDER=lbvh only)
};

struct BNode {                          // trivially constructible: the pool below is never value-initialised
    Box box;
    uint32_t left, right;               // interior (count == 0)
    uint32_t begin, count;              // leaf: prims[begin, begin + count), one kind
    uint32_t span;                      // primitives below this node (any node)
    // collapse plan (planCollapse): cost[i-1] = least total area of the wide nodes needed to hang this
    // subtree into at most i slots of a parent; take[i-1] = how many of those slots the left child
    // gets, 0 = the subtree becomes one wide node of its own; take[0] = the split when it does
    float cost[4];
    uint8_t take[4];
    uint32_t parent;                    // set by the parent when it allocates the pair
    uint32_t ready;                     // children whose plans are complete (atomic)
};

struct BNodePool {                      // 2N+2 nodes of address space; only the pages a build touches are committed
    std::unique_ptr<BNode[]> mem;
    size_t used = 0;
    const BNode& operator[](size_t i) const { return mem[i]; }
    BNode& operator[](size_t i) { return mem[i]; }
};

class BinaryBuilder {
public:
    BinaryBuilder(std::vector<BPrim>& prims, int maxLeaf, bool medianOnly)
        : prims_(prims), maxLeaf_(maxLeaf), medianOnly_(medianOnly)
    {
        nodes_.mem.reset(new BNode[std::max<size_t>(1, 2 * prims.size() + 2)]);
        nodes_[0] = BNode{};
    }

    // YART_WIDE_BUILDER=lbvh: the topology of a linear BVH (primitives in Morton order of their centres, every
    // range cut where the highest differing code bit flips) instead of binned SAH -- what the device builder
    // (wide_build_dev.hip) produces; kept on the host to compare tree quality without a GPU.
    void sortMorton()
    {
        Box cb;
        cb.reset();
        for (const BPrim& p : prims_) cb.grow(p.c, p.c);
        float inv[3];
        for (int a = 0; a < 3; ++a) {
            const float e = cb.hi[a] - cb.lo[a];
            inv[a] = (e > 0.0f && e < std::numeric_limits<float>::infinity()) ? 2097151.0f / e : 0.0f;
        }
        auto spread = [](uint64_t v) {
            v &= 0x1FFFFFull;
            v = (v | (v << 32)) & 0x1F00000000FFFFull;
            v = (v | (v << 16)) & 0x1F0000FF0000FFull;
            v = (v | (v << 8)) & 0x100F00F00F00F00Full;
            v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
            v = (v | (v << 2)) & 0x1249249249249249ull;
            return v;
        };
        for (BPrim& p : prims_) {
            uint64_t q[3];
            for (int a = 0; a < 3; ++a) {
                float f = (p.c[a] - cb.lo[a]) * inv[a];
                f = f < 0.0f ? 0.0f : (f > 2097151.0f ? 2097151.0f : f);
                q[a] = static_cast<uint64_t>(f);
            }
            p.code = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        }
        std::sort(prims_.begin(), prims_.end(), [](const BPrim& a, const BPrim& b) { return a.code < b.code || (a.code == b.code && a.id < b.id); });
    }

    uint32_t lbvhSplit(uint32_t begin, uint32_t end) const
    {
        const uint64_t first = prims_[begin].code, last = prims_[end - 1].code;
        if (first == last) return begin + (end - begin) / 2;
        const int bit = 63 - __builtin_clzll(first ^ last);          // highest bit in which the range's codes differ
        uint32_t lo = begin, hi = end - 1;                            // first index whose code has that bit set
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if ((prims_[mid].code >> bit) & 1ull) hi = mid; else lo = mid;
        }
        return hi;
    }

    void run()
    {
        next_ = 1;
        if (lbvh_) sortMorton();
        unsigned nThreads = buildThreads();
        if (prims_.size() < kSpawnAbove) nThreads = 1;
        pool_.run(nThreads, Job{ 0u, 0u, static_cast<uint32_t>(prims_.size()) },
                  [this](const Job& j) { build(j.node, j.begin, j.end, true); });
        nodes_.used = next_;
    }

    BNodePool& nodes() { return nodes_; }

private:
    // Which binary nodes become wide nodes.  Opening the child of largest area until four slots are
    // full (the usual greedy collapse) leaves 36 % of the nodes of a single-primitive-leaf tree with
    // two children; the cost of a traversal step does not depend on how many slots are used, so
    // the plan minimises the summed area (= expected visits) of the wide nodes instead: a dynamic
    // programme over (node, slots available), bottom-up, done while the tree is built.  The plan of a
    // node needs its children's: a leaf plans itself when it is made and reports to its parent; the
    // second child to report plans the parent and climbs on, whichever thread that is.
    void planned(uint32_t idx)
    {
        for (;;) {
            planNode(idx);
            if (idx == 0) return;
            const uint32_t p = nodes_[idx].parent;
            if (__atomic_fetch_add(&nodes_[p].ready, 1u, __ATOMIC_ACQ_REL) == 0u) return;     // the sibling is still being built
            idx = p;
        }
    }

    void planNode(uint32_t idx)
    {
        BNode& n = nodes_[idx];
        if (n.count > 0) {
            for (int i = 0; i < 4; ++i) { n.cost[i] = 0.0f; n.take[i] = 0; }
            return;
        }
        const BNode& l = nodes_[n.left];
        const BNode& r = nodes_[n.right];
        float forest[5];
        uint8_t split[5];
        for (int k = 2; k <= 4; ++k) {
            forest[k] = std::numeric_limits<float>::infinity();
            split[k] = 1;
            for (int j = 1; j < k; ++j) {
                const float c = l.cost[j - 1] + r.cost[k - j - 1];
                if (c < forest[k]) { forest[k] = c; split[k] = static_cast<uint8_t>(j); }
            }
        }
        const float asNode = n.box.halfArea() + forest[4];
        n.cost[0] = asNode;
        n.take[0] = split[4];
        for (int k = 2; k <= 4; ++k) {
            if (forest[k] < asNode) { n.cost[k - 1] = forest[k]; n.take[k - 1] = split[k]; }
            else { n.cost[k - 1] = asNode; n.take[k - 1] = 0; }
        }
    }

    struct Job { uint32_t node, begin, end; };

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
        n.span = count;
        n.left = n.right = n.begin = n.count = 0;
        uint32_t mid;
        if (count <= static_cast<uint32_t>(maxLeaf_)) {
            // a leaf holds one kind of primitive; a mixed remainder becomes two leaves
            auto first = prims_.begin() + begin, last = prims_.begin() + end;
            auto m = std::partition(first, last, [](const BPrim& p) { return (p.id & kSphereId) == 0; });
            if (m == first || m == last) {
                n.begin = begin;
                n.count = count;
                planned(node);
                return;
            }
            mid = static_cast<uint32_t>(m - prims_.begin());
        } else {
            mid = split(begin, end, cbox);
        }
        const uint32_t l = next_.fetch_add(2);
        n.left = l;
        n.right = l + 1;
        n.ready = 0;
        nodes_[l].parent = nodes_[l + 1].parent = node;
        if (maySpawn && count > kSpawnAbove) {
            pool_.spawn(Job{ l, begin, mid });
            pool_.spawn(Job{ l + 1, mid, end });
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
        if (lbvh_) return lbvhSplit(begin, end);

        const int nBins = count > static_cast<uint32_t>(fineBinsBelow_) ? bins_ : fineBins_;
        int bestAxis = -1, bestSplit = 0;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
            if (count > oneAxisAbove_ && a != widest) continue;
            if (!(ext[a] > 0.0f)) continue;
            Box bins[kMaxBins];
            uint32_t cnt[kMaxBins] = { 0 };
            for (auto& b : bins) b.reset();
            const float scale = static_cast<float>(nBins) / ext[a];
            for (uint32_t i = begin; i < end; ++i) {
                int b = static_cast<int>((prims_[i].c[a] - cbox.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= nBins ? nBins - 1 : b);
                bins[b].grow(prims_[i].lo, prims_[i].hi);
                ++cnt[b];
            }
            float rightArea[kMaxBins];
            uint32_t rightCnt[kMaxBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = nBins - 1; b > 0; --b) {
                acc.grow(bins[b]);
                c += cnt[b];
                rightArea[b] = acc.halfArea();
                rightCnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 1; b < nBins; ++b) {             // split s: bins [0,s) left, [s,nBins) right
                acc.grow(bins[b - 1]);
                c += cnt[b - 1];
                if (c == 0 || rightCnt[b] == 0) continue;
                const float cost = acc.halfArea() * static_cast<float>(c) + rightArea[b] * static_cast<float>(rightCnt[b]);
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
            }
        }
        if (bestAxis < 0) return median(begin, end, widest);
        const float lo = cbox.lo[bestAxis], scale = static_cast<float>(nBins) / ext[bestAxis];
        const int s = bestSplit, axis = bestAxis;
        auto m = std::partition(prims_.begin() + begin, prims_.begin() + end, [=](const BPrim& p) {
            int b = static_cast<int>((p.c[axis] - lo) * scale);
            b = b < 0 ? 0 : (b >= nBins ? nBins - 1 : b);
            return b < s;
        });
        const uint32_t mid = static_cast<uint32_t>(m - prims_.begin());
        if (mid == begin || mid == end) return median(begin, end, widest);
        return mid;
    }

    std::vector<BPrim>& prims_;
    BNodePool nodes_;
    std::atomic<uint32_t> next_{ 1 };
    int maxLeaf_;
    bool medianOnly_;
    bool lbvh_ = [] { const char* v = std::getenv("YART_WIDE_BUILDER"); return v && std::string(v) == "lbvh"; }();
    // SAH binning: bins per axis, all three axes below oneAxisAbove_ primitives (the widest only above)
    int bins_ = std::max(2, std::min(kMaxBins, envInt("YART_SAH_BINS", 16)));
    int fineBins_ = std::max(2, std::min(kMaxBins, envInt("YART_SAH_FINE_BINS", 16)));
    int fineBinsBelow_ = envInt("YART_SAH_FINE_BELOW", 0);
    uint32_t oneAxisAbove_ = static_cast<uint32_t>(envInt("YART_SAH_ONE_AXIS_ABOVE", 1 << 20));
    TaskPool<Job> pool_;
};

float bitsFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
uint32_t floatBits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Collapse + emission.  The top of the tree is emitted by one thread into segment 0; every binary
// subtree below `jobSpan` primitives becomes a job emitted by the pool into a segment of its own
// (local node / primitive numbers); the segments are then laid end to end, in depth-first order of
// their roots, and their references rebased.
class Emitter {
public:
    Emitter(const BNodePool& nodes, const std::vector<BPrim>& prims, const TriangleBvh& tri,
            const float* spheres, const std::vector<uint8_t>& bandClass, WideBvh& out)
        : nodes_(nodes), prims_(prims), tri_(tri), spheres_(spheres), bandClass_(bandClass), out_(out) { }

    bool tooManyPrims = false;
    bool planned_ = false;              // follow BNode::take (planCollapse) instead of the greedy collapse

    void run(unsigned nThreads)
    {
        out_.nodes.clear();
        out_.rowsA.clear();
        out_.rowsBC.clear();
        out_.depth = 0;
        out_.numLeaves = 0;
        if (prims_.size() > kWideMaxPrims) { tooManyPrims = true; return; }
        if (prims_.empty()) {
            out_.nodes.emplace_back();
            clear(out_.nodes[0]);
            out_.depth = 1;
            return;
        }
        Segment top;
        if (nodes_[0].count > 0) {                       // the whole scene fits one leaf
            top.nodes.emplace_back();
            const uint32_t ref = leaf(nodes_[0], top);
            setChildren(top.nodes[0], &nodes_[0].box, &ref, 1);
            top.depth = 1;
        } else {
            // the cut does not depend on the number of threads: the arrays are the same on every machine
            jobSpan_ = std::max<uint32_t>(4096u, static_cast<uint32_t>(prims_.size() / 1024u));
            emit(0, 1, top, &jobs_);
        }
        // jobs, in parallel
        std::vector<Segment> segs(jobs_.size());
        if (!jobs_.empty()) {
            std::atomic<size_t> next{ 0 };
            auto body = [&]() {
                for (size_t j; (j = next.fetch_add(1)) < jobs_.size();) {
                    emit(jobs_[j].bnode, jobs_[j].depth, segs[j], nullptr);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(body);
            body();
            for (auto& t : pool) t.join();
        }
        // lay the segments end to end
        std::vector<size_t> nodeBase(jobs_.size()), primBase(jobs_.size());
        size_t nNodes = top.nodes.size(), nPrims = top.rowsA.size() / 4;
        out_.depth = top.depth;
        out_.numLeaves = top.leaves;
        for (size_t j = 0; j < jobs_.size(); ++j) {
            nodeBase[j] = nNodes;
            primBase[j] = nPrims;
            nNodes += segs[j].nodes.size();
            nPrims += segs[j].rowsA.size() / 4;
            out_.depth = std::max(out_.depth, segs[j].depth);
            out_.numLeaves += segs[j].leaves;
        }
        if (nNodes >= (1ull << 31)) { tooManyPrims = true; return; }
        for (size_t j = 0; j < jobs_.size(); ++j)        // the parents' references to the job roots
            top.nodes[jobs_[j].parent].ref[jobs_[j].slot] = static_cast<uint32_t>(nodeBase[j]);
        out_.nodes.resize(nNodes);
        out_.rowsA.resize(nPrims * 4);
        out_.rowsBC.resize(nPrims * 8);
        auto place = [&](const Segment& sg, size_t nb, size_t pb) {
            for (size_t i = 0; i < sg.nodes.size(); ++i) {
                WideNode w = sg.nodes[i];
                if (nb || pb) {
                    for (auto& r : w.ref) {
                        if (r == kWideNone) continue;
                        if (r & kWideLeafBit) r = (r & 0xF8000000u) | static_cast<uint32_t>((r & 0x07FFFFFFu) + pb);
                        else r = static_cast<uint32_t>(r + nb);
                    }
                }
                out_.nodes[nb + i] = w;
            }
            if (!sg.rowsA.empty()) {
                std::memcpy(out_.rowsA.data() + pb * 4, sg.rowsA.data(), sg.rowsA.size() * sizeof(float));
                std::memcpy(out_.rowsBC.data() + pb * 8, sg.rowsBC.data(), sg.rowsBC.size() * sizeof(float));
            }
        };
        place(top, 0, 0);
        {
            std::atomic<size_t> next{ 0 };
            auto body = [&]() { for (size_t j; (j = next.fetch_add(1)) < segs.size();) place(segs[j], nodeBase[j], primBase[j]); };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads && t < segs.size(); ++t) pool.emplace_back(body);
            body();
            for (auto& t : pool) t.join();
        }
    }

private:
    struct Segment {
        std::vector<WideNode> nodes;
        std::vector<float> rowsA, rowsBC;
        int depth = 0;
        size_t leaves = 0;
    };
    struct Job { uint32_t bnode, parent; int slot, depth; };

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
    static void setChildren(WideNode& w, const Box* boxesIn, const uint32_t* refs, int n)
    {
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
            scale = packedStep(scale);                   // the step the packed node can hold (wide_build.h)
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
                scale = packedStep(std::nextafter(scale, std::numeric_limits<float>::infinity()));
            }
            w.origin[a] = origin;
            w.scale[a] = scale;
        }
        for (int i = 0; i < n; ++i) w.ref[i] = refs[i];
    }

    // Emits the wide node rooted at binary node `bnode` (and everything below it) into `sg`; with
    // `jobs` set, interior children of at most jobSpan_ primitives are recorded as jobs instead.
    uint32_t emit(uint32_t bnode, int depth, Segment& sg, std::vector<Job>* jobs)
    {
        sg.depth = std::max(sg.depth, depth);
        uint32_t kids[4] = { nodes_[bnode].left, nodes_[bnode].right, 0, 0 };
        int n = 2;
        if (planned_) {                                  // the slots go where planCollapse put them
            n = 0;
            const int leftSlots = nodes_[bnode].take[0];
            gather(nodes_[bnode].left, leftSlots, kids, n);
            gather(nodes_[bnode].right, 4 - leftSlots, kids, n);
        } else {
            while (n < 4) {                              // open the largest interior child
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
        }
        const uint32_t idx = static_cast<uint32_t>(sg.nodes.size());
        sg.nodes.emplace_back();
        Box boxes[4];
        uint32_t refs[4];
        for (int i = 0; i < n; ++i) {
            const BNode& c = nodes_[kids[i]];
            boxes[i] = c.box;
            if (c.count > 0) refs[i] = leaf(c, sg);
            else if (jobs && c.span <= jobSpan_) {
                jobs->push_back(Job{ kids[i], idx, i, depth + 1 });
                refs[i] = 0u;                            // patched when the segments are laid out
            } else refs[i] = emit(kids[i], depth + 1, sg, jobs);
        }
        setChildren(sg.nodes[idx], boxes, refs, n);
        return idx;
    }

    // the binary nodes that fill `slots` slots of a parent for the subtree at `b`
    void gather(uint32_t b, int slots, uint32_t* kids, int& n) const
    {
        const BNode& node = nodes_[b];
        if (node.count > 0 || slots == 1 || node.take[slots - 1] == 0) { kids[n++] = b; return; }
        const int leftSlots = node.take[slots - 1];
        gather(node.left, leftSlots, kids, n);
        gather(node.right, slots - leftSlots, kids, n);
    }

    uint32_t leaf(const BNode& n, Segment& sg)
    {
        const size_t first = sg.rowsA.size() / 4;
        const uint32_t cnt = n.count;
        const bool sph = (prims_[n.begin].id & kSphereId) != 0;
        uint32_t ids[kWideMaxLeaf];
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t id = prims_[n.begin + i].id;
            ids[i] = sph ? (id & ~kSphereId) : tri_.slotOfInput[id];
        }
        std::sort(ids, ids + cnt);                       // in-leaf test order: lowest slot / sphere number first
        sg.rowsA.resize((first + cnt) * 4, 0.0f);
        sg.rowsBC.resize((first + cnt) * 8, 0.0f);
        for (uint32_t i = 0; i < cnt; ++i) {
            float* A = sg.rowsA.data() + (first + i) * 4;
            float* B = sg.rowsBC.data() + (first + i) * 8;
            float* C = B + 4;
            if (sph) {
                std::memcpy(A, spheres_ + static_cast<size_t>(ids[i]) * 4, 16);
                B[0] = bitsFloat(ids[i]);
            } else {
                const YartTriAccel& ta = tri_.triaccel[ids[i]];
                A[0] = ta.n_u; A[1] = ta.n_v; A[2] = ta.n_d;
                A[3] = bitsFloat(static_cast<uint32_t>(ta.k) | (ids[i] << 2) | (static_cast<uint32_t>(bandClass_[ids[i]]) << kPackBandShift));
                B[0] = ta.b_u; B[1] = ta.b_v; B[2] = ta.b_d;
                C[0] = ta.c_u; C[1] = ta.c_v; C[2] = ta.c_d;
            }
        }
        ++sg.leaves;
        return kWideLeafBit | (sph ? kWideSphereBit : 0u) | ((cnt - 1) << 27) | static_cast<uint32_t>(first);
    }

    const BNodePool& nodes_;
    const std::vector<BPrim>& prims_;
    const TriangleBvh& tri_;
    const float* spheres_;
    const std::vector<uint8_t>& bandClass_;          // per leaf slot: class of the triangle's error band (cover.h)
    WideBvh& out_;
    std::vector<Job> jobs_;
    uint32_t jobSpan_ = 0;
};

// Node order in memory.  The emitter writes nodes depth first (a node, then the subtrees of its
// children one after the other).  What a fetch beyond L2 moves on MI355X is a 128-byte line = TWO nodes
// (tools/calib_fetch.hip: a scattered 64-byte read is a 128-byte request to memory), so which node shares
// a node's line decides whether the second half of every miss is useful:
//   mode 0  depth first as emitted (a node's line mate is its first child, or an unrelated subtree's end)
//   mode 1  a node shares its line with its largest interior child (the one a ray that visits the node
//           most likely visits next): the dependent fetch after a miss is already in L2
//   mode 2  the interior children of a node are stored side by side (siblings share lines)
// Lines follow one another depth first in every mode.  nodes[0] stays the root.
float childArea(const WideNode& w, int c)
{
    const float dx = (float(w.qhi[0][c]) - float(w.qlo[0][c])) * w.scale[0];
    const float dy = (float(w.qhi[1][c]) - float(w.qlo[1][c])) * w.scale[1];
    const float dz = (float(w.qhi[2][c]) - float(w.qlo[2][c])) * w.scale[2];
    if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f)) return -1.0f;
    return dx * dy + dy * dz + dz * dx;
}

void relayout(WideBvh& bvh, int mode)
{
    const size_t n = bvh.nodes.size();
    if (mode == 0 || n < 3) return;
    constexpr uint32_t kUnset = 0xFFFFFFFFu;
    std::vector<uint32_t> newOf(n, kUnset), oldAt;
    oldAt.reserve(n);
    auto interiorKids = [&](uint32_t x, uint32_t* kids, float* area) {      // sorted by area, largest first
        int k = 0;
        const WideNode& w = bvh.nodes[x];
        for (int c = 0; c < 4; ++c) {
            const uint32_t r = w.ref[c];
            if (r == kWideNone || (r & kWideLeafBit)) continue;
            kids[k] = r;
            area[k] = childArea(w, c);
            ++k;
        }
        for (int i = 1; i < k; ++i)
            for (int j = i; j > 0 && area[j] > area[j - 1]; --j) { std::swap(area[j], area[j - 1]); std::swap(kids[j], kids[j - 1]); }
        return k;
    };
    auto place = [&](uint32_t x) { newOf[x] = static_cast<uint32_t>(oldAt.size()); oldAt.push_back(x); };
    std::vector<uint32_t> stack;
    stack.push_back(0u);
    uint32_t kids[4];
    float area[4];
    if (mode == 1) {
        while (!stack.empty()) {
            const uint32_t x = stack.back();
            stack.pop_back();
            place(x);
            int k = interiorKids(x, kids, area);
            if ((oldAt.size() & 1u) == 0u || k == 0) {           // x took the second half of a line, or has no interior child
                for (int i = k - 1; i >= 0; --i) stack.push_back(kids[i]);
                continue;
            }
            const uint32_t heavy = kids[0];
            place(heavy);                                        // second half of x's line
            for (int i = k - 1; i >= 1; --i) stack.push_back(kids[i]);
            uint32_t hk[4];
            float ha[4];
            const int m = interiorKids(heavy, hk, ha);
            for (int i = m - 1; i >= 0; --i) stack.push_back(hk[i]);       // the heavy child's subtree comes first
        }
    } else {
        place(0u);
        while (!stack.empty()) {
            const uint32_t x = stack.back();
            stack.pop_back();
            const int k = interiorKids(x, kids, area);
            for (int i = 0; i < k; ++i) place(kids[i]);          // siblings side by side
            for (int i = k - 1; i >= 0; --i) stack.push_back(kids[i]);
        }
    }
    if (oldAt.size() != n) return;                               // unreachable nodes: keep the emitted order
    std::vector<WideNode> out(n);
    parallelChunks(n, buildThreads(), size_t(1) << 14, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            WideNode w = bvh.nodes[oldAt[i]];
            for (auto& r : w.ref)
                if (r != kWideNone && !(r & kWideLeafBit)) r = newOf[r];
            out[i] = w;
        }
    });
    bvh.nodes.swap(out);
}

constexpr uint32_t kStepMask = 0x7FE00000u;             // exponent + two mantissa bits

} // namespace

float packedStep(float s)
{
    uint32_t u = floatBits(s);
    if (u & ~kStepMask) u = (u & kStepMask) + 0x00200000u;          // round up; a mantissa carry bumps the exponent
    if (u < 0x00800000u) u = 0x00800000u;                            // no denormal steps
    if (u >= 0x7F800000u) u = 0x7F600000u;                           // nor infinite ones: the largest step (1.75 * 2^127)
    return bitsFloat(u);
}

bool packWide(const WideBvh& bvh, PackedWide& out)
{
    const size_t n = bvh.nodes.size();
    out.words.clear();
    out.numUnits = 0;
    if (n == 0) return false;
    // Children blocks one after the other, in node order, in two regions: blocks with interior children (64-byte
    // aligned; the leaf rows of such a node follow its interior children, padded to 64 bytes) behind the root, then
    // the blocks of the nodes that have leaf children only -- nothing but rows, packed densely, so that the rows
    // of neighbouring leaves share cache lines.
    std::vector<uint32_t> block(n), unitOf(n, 0u);
    size_t totalNodes = 4, totalRows = 0;
    for (size_t i = 0; i < n; ++i) {
        size_t interior = 0, leafUnits = 0;
        for (const uint32_t r : bvh.nodes[i].ref) {
            if (r == kWideNone) continue;
            if (!(r & kWideLeafBit)) ++interior;
            else {
                if ((r >> 27) & 7u) return false;                    // single-primitive leaves only
                leafUnits += (r & kWideSphereBit) ? 2 : 1;
            }
        }
        if (interior) {
            block[i] = static_cast<uint32_t>(totalNodes);
            totalNodes += 4 * interior + (leafUnits + 3) / 4 * 4;
        } else {
            block[i] = static_cast<uint32_t>(totalRows) | 0x80000000u;       // rebased below
            totalRows += leafUnits;
        }
        if (totalNodes + totalRows > kPackMaxUnits) return false;
    }
    for (size_t i = 0; i < n; ++i)
        if (block[i] & 0x80000000u) block[i] = static_cast<uint32_t>(totalNodes + (block[i] & 0x7FFFFFFFu));
    const size_t total = totalNodes + totalRows;
    for (size_t i = 0; i < n; ++i) {
        uint32_t j = 0;
        for (const uint32_t r : bvh.nodes[i].ref)
            if (r != kWideNone && !(r & kWideLeafBit)) {
                if (r >= n) return false;
                unitOf[r] = block[i] + 4 * j++;
            }
    }
    out.numUnits = total;
    out.words.assign(total * 4, 0u);
    std::atomic<bool> ok{ true };
    parallelChunks(n, buildThreads(), size_t(1) << 14, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const WideNode& w = bvh.nodes[i];
            uint32_t* u = &out.words[static_cast<size_t>(unitOf[i]) * 4];
            uint32_t steps = 0;
            for (int a = 0; a < 3; ++a) {
                u[a] = floatBits(w.origin[a]);
                const uint32_t sb = floatBits(w.scale[a]);
                if (sb & ~kStepMask) ok = false;                     // not a step of the packed format
                steps |= sb >> (10 * a);
            }
            u[3] = steps;
            for (int a = 0; a < 3; ++a) std::memcpy(&u[4 + a], w.qlo[a], 4);
            for (int a = 0; a < 3; ++a) std::memcpy(&u[7 + a], w.qhi[a], 4);
            uint32_t interior = 0;
            for (const uint32_t r : w.ref) interior += (r != kWideNone && !(r & kWideLeafBit)) ? 1u : 0u;
            uint32_t offs = 0, nextInterior = 0, nextLeaf = 4 * interior;
            for (int c = 0; c < 4; ++c) {
                const uint32_t r = w.ref[c];
                if (r == kWideNone) continue;
                if (!(r & kWideLeafBit)) {
                    offs |= ((4u * nextInterior++) << 2) << (8 * c);
                    continue;
                }
                const bool sph = (r & kWideSphereBit) != 0;
                const size_t first = r & 0x07FFFFFFu;
                offs |= ((nextLeaf << 2) | kPackLeaf | (sph ? kPackSphere : 0u)) << (8 * c);
                uint32_t* row = &out.words[(static_cast<size_t>(block[i]) + nextLeaf) * 4];
                std::memcpy(row, &bvh.rowsA[first * 4], 16);
                if (sph) std::memcpy(row + 4, &bvh.rowsBC[first * 8], 4);      // the sphere's number
                nextLeaf += sph ? 2u : 1u;
            }
            u[10] = block[i] << 2;
            u[11] = offs;
        }
    });
    return ok;
}

bool unpackWide(const uint32_t* words, size_t numUnits, const YartTriAccel* triaccel, size_t nTriangles, WideBvh& out)
{
    out = WideBvh{};
    if (numUnits < 4) return false;
    std::vector<uint32_t> level{ 0u }, next;             // unit indices of the nodes of one level
    std::vector<uint32_t> unitOfNode{ 0u };
    out.nodes.emplace_back();
    size_t done = 0;
    while (!level.empty()) {
        ++out.depth;
        next.clear();
        for (const uint32_t unit : level) {
            if (static_cast<size_t>(unit) + 4 > numUnits) return false;
            const uint32_t* u = words + static_cast<size_t>(unit) * 4;
            WideNode w;
            std::memset(&w, 0, sizeof w);
            for (int a = 0; a < 3; ++a) {
                w.origin[a] = bitsFloat(u[a]);
                w.scale[a] = bitsFloat((u[3] << (10 * a)) & kStepMask);
                std::memcpy(w.qlo[a], &u[4 + a], 4);
                std::memcpy(w.qhi[a], &u[7 + a], 4);
            }
            const size_t blockUnit = u[10] >> 2;
            for (int c = 0; c < 4; ++c) {
                w.ref[c] = kWideNone;
                if (w.qlo[0][c] > w.qhi[0][c]) continue;             // unused slot: inverted box
                const uint32_t ob = (u[11] >> (8 * c)) & 255u;
                const size_t child = blockUnit + (ob >> 2);
                if (!(ob & kPackLeaf)) {
                    if (out.nodes.size() + next.size() >= (size_t(1) << 31)) return false;
                    w.ref[c] = static_cast<uint32_t>(unitOfNode.size());
                    unitOfNode.push_back(static_cast<uint32_t>(child));
                    next.push_back(static_cast<uint32_t>(child));
                    continue;
                }
                const bool sph = (ob & kPackSphere) != 0;
                if (child + (sph ? 2 : 1) > numUnits) return false;
                const uint32_t* row = words + child * 4;
                const size_t first = out.rowsA.size() / 4;
                if (first > kWideMaxPrims) return false;
                out.rowsA.resize((first + 1) * 4);
                out.rowsBC.resize((first + 1) * 8, 0.0f);
                std::memcpy(&out.rowsA[first * 4], row, 16);
                if (sph) std::memcpy(&out.rowsBC[first * 8], row + 4, 4);
                else {
                    const size_t slot = (row[3] & kPackSlotMask) >> 2;
                    if (slot >= nTriangles) return false;
                    const YartTriAccel& ta = triaccel[slot];
                    float* B = &out.rowsBC[first * 8];
                    B[0] = ta.b_u; B[1] = ta.b_v; B[2] = ta.b_d;
                    B[4] = ta.c_u; B[5] = ta.c_v; B[6] = ta.c_d;
                }
                w.ref[c] = kWideLeafBit | (sph ? kWideSphereBit : 0u) | static_cast<uint32_t>(first);
                ++out.numLeaves;
            }
            if (done >= out.nodes.size()) return false;
            out.nodes[done++] = w;
        }
        out.nodes.resize(unitOfNode.size());
        level.swap(next);
        if (out.depth > kWideStack) return false;
    }
    return done == out.nodes.size();
}

Cover sceneCover(const float* verts, size_t nVerts, const float* spheres, size_t nSpheres)
{
    // bounds of the finite vertex coordinates and sphere extents, smallest sphere radius
    struct Acc { float lo[3], hi[3]; float rMin; bool any; };
    const float inf = std::numeric_limits<float>::infinity();
    const Acc empty{ { inf, inf, inf }, { -inf, -inf, -inf }, inf, false };
    std::mutex mu;
    Acc all = empty;
    auto merge = [&](const Acc& a) {
        std::lock_guard<std::mutex> lock(mu);
        for (int c = 0; c < 3; ++c) { all.lo[c] = std::min(all.lo[c], a.lo[c]); all.hi[c] = std::max(all.hi[c], a.hi[c]); }
        all.rMin = std::min(all.rMin, a.rMin);
        all.any = all.any || a.any;
    };
    parallelChunks(nVerts, buildThreads(), size_t(1) << 18, [&](size_t b, size_t e) {
        Acc a = empty;
        for (size_t i = b; i < e; ++i)
            for (int c = 0; c < 3; ++c) {
                const float x = verts[i * 3 + c];
                if (!(std::fabs(x) < inf)) continue;
                a.lo[c] = std::min(a.lo[c], x); a.hi[c] = std::max(a.hi[c], x); a.any = true;
            }
        merge(a);
    });
    parallelChunks(nSpheres, buildThreads(), size_t(1) << 18, [&](size_t b, size_t e) {
        Acc a = empty;
        for (size_t i = b; i < e; ++i) {
            const float* s = spheres + i * 4;
            const float r = std::fabs(s[3]);
            if (!(r < inf)) continue;
            a.rMin = std::min(a.rMin, r);
            for (int c = 0; c < 3; ++c) {
                if (!(std::fabs(s[c]) < inf)) continue;
                a.lo[c] = std::min(a.lo[c], s[c] - r); a.hi[c] = std::max(a.hi[c], s[c] + r); a.any = true;
            }
        }
        merge(a);
    });
    bool have = all.any;
    for (int c = 0; c < 3; ++c) {
        if (!(all.lo[c] <= all.hi[c])) { all.lo[c] = all.hi[c] = 0.0f; }      // an axis without a finite coordinate
        if (!(std::fabs(all.lo[c]) < inf && std::fabs(all.hi[c]) < inf)) have = false;
    }
    // Outliers: the error bounds grow with the size of the cover (cover.h), so one vertex far from everything else
    // (a sky dome, a stray coordinate of 1e38) must not set it.  The cover is built around where the primitives ARE: per
    // axis the median of a sample of the vertices and sphere centres, +- 64 x the widest median absolute deviation of
    // the three axes (a uniformly filled box has a deviation of a quarter of its side: 16 sides each way), and never
    // beyond the true bounds.  Rays that start outside it take the fix-up path (traverse.hip).
    const char* trim = std::getenv("YART_COVER_TRIM");
    if (have && !(trim && trim[0] == '0')) {
        constexpr size_t kSample = 4096;
        std::vector<float> q[3];
        auto take = [&](const float* p) { for (int c = 0; c < 3; ++c) if (std::fabs(p[c]) < inf) q[c].push_back(p[c]); };
        const size_t sv = std::max<size_t>(1, nVerts / kSample), ss = std::max<size_t>(1, nSpheres / kSample);
        for (size_t i = 0; i < nVerts; i += sv) take(verts + i * 3);
        for (size_t i = 0; i < nSpheres; i += ss) take(spheres + i * 4);
        float med[3], spread = 0.0f;
        bool ok = true;
        for (int c = 0; c < 3; ++c) {
            if (q[c].empty()) { ok = false; break; }
            std::sort(q[c].begin(), q[c].end());
            med[c] = q[c][q[c].size() / 2];
            for (float& x : q[c]) x = std::fabs(x - med[c]);
            std::sort(q[c].begin(), q[c].end());
            spread = std::max(spread, q[c][q[c].size() / 2]);
        }
        if (ok && spread > 0.0f && spread < inf)
            for (int c = 0; c < 3; ++c) {
                all.lo[c] = std::max(all.lo[c], med[c] - 64.0f * spread);
                all.hi[c] = std::min(all.hi[c], med[c] + 64.0f * spread);
            }
    }
    return makeCover(all.lo, all.hi, have, nSpheres, all.rMin < inf ? all.rMin : 0.0f);
}

float trianglePadMax(const float* verts, const int32_t* tris, size_t nTris, const Cover& cover)
{
    std::atomic<uint32_t> padMaxBits{ 0u };
    parallelChunks(nTris, buildThreads(), size_t(1) << 16, [&](size_t b, size_t e) {
        float chunkPad = 0.0f;
        for (size_t i = b; i < e; ++i) {
            const int32_t* t = tris + i * 5;
            const float *va = verts + static_cast<size_t>(t[0]) * 3, *vb = verts + static_cast<size_t>(t[1]) * 3, *vc = verts + static_cast<size_t>(t[2]) * 3;
            float rec[12], lo[3], hi[3];
            uint32_t cls;
            projectRecord(rec, va, vb, vc, t[4], t[3]);
            triBand(rec, va, vb, vc, cover, lo, hi, cls);
            chunkPad = std::max(chunkPad, bandExcess(lo, hi, va, vb, vc));
        }
        uint32_t bits = floatBits(chunkPad), seen = padMaxBits.load();
        while (bits > seen && !padMaxBits.compare_exchange_weak(seen, bits)) { }
    });
    return bitsFloat(padMaxBits.load());
}

int buildWideBvh(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                 const TriangleBvh& tri, const float* spheres, size_t nSpheres,
                 int maxLeaf, const Cover& cover, WideBvh& out, std::string& err)
{
    (void)nVerts;
    maxLeaf = 1;            // one primitive per leaf: what the packed format holds (wide_build.h)
    if (nTris >= (1u << 30) || nSpheres >= (1u << 30)) {
        err = "wide BVH: more than 2^30 primitives of one kind";
        return YART_ERR_INVALID_ARGUMENT;
    }
    // Pre-splitting (YART_WIDE_PRESPLIT=k): the box of a triangle is mostly empty space, and every ray
    // through that space pays a node's slot and a plane-row test.  A triangle is cut into up to k slabs
    // across the longest axis of its box and enters the build once per slab, with the box of its part
    // inside the slab; each part's leaf tests the whole triangle (same rows, same result), so a ray
    // that meets two parts finds the same hit twice.
    const int presplit = std::max(1, std::min(8, envInt("YART_WIDE_PRESPLIT", 1)));
    const size_t perTri = static_cast<size_t>(presplit);
    std::vector<BPrim> prims(nTris * perTri + nSpheres);
    std::vector<uint8_t> bandClass(nTris, 0);           // by leaf slot
    std::atomic<uint32_t> padMaxBits{ 0u };             // non-negative floats order like their bits
    parallelChunks(nTris, buildThreads(), size_t(1) << 16, [&](size_t b, size_t e) {
        float chunkPad = 0.0f;
        for (size_t i = b; i < e; ++i) {
            const int32_t* t = tris + i * 5;
            const float* vv[3] = { verts + static_cast<size_t>(t[0]) * 3, verts + static_cast<size_t>(t[1]) * 3,
                                   verts + static_cast<size_t>(t[2]) * 3 };
            BPrim whole;
            // the triangle's box grown by the error band of its Wald test (cover.h): every point the test can accept
            uint32_t cls = 0;
            const uint32_t slot = tri.slotOfInput[i];
            triBand(reinterpret_cast<const float*>(&tri.triaccel[slot]), vv[0], vv[1], vv[2], cover, whole.lo, whole.hi, cls);
            bandClass[slot] = static_cast<uint8_t>(cls);
            chunkPad = std::max(chunkPad, bandExcess(whole.lo, whole.hi, vv[0], vv[1], vv[2]));
            whole.id = static_cast<uint32_t>(i);
            BPrim* out = &prims[i * perTri];
            int made = 0;
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (whole.hi[a] - whole.lo[a] > whole.hi[axis] - whole.lo[axis]) axis = a;
            const double ext = static_cast<double>(whole.hi[axis]) - whole.lo[axis];
            if (presplit > 1 && ext > 0.0 && ext < 1e30) {
                for (int sidx = 0; sidx < presplit; ++sidx) {
                    // slab [p0, p1] widened a little: parts overlap, so no point of the triangle falls between two boxes
                    const double margin = ext * 1e-5;
                    const double p0 = whole.lo[axis] + ext * sidx / presplit - margin;
                    const double p1 = whole.lo[axis] + ext * (sidx + 1) / presplit + margin;
                    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
                    bool any = false;
                    auto add = [&](const double* q) {
                        any = true;
                        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
                    };
                    for (int k = 0; k < 3; ++k) {                 // every edge: end points inside the slab, crossings of its planes
                        const float* A = vv[k];
                        const float* B = vv[(k + 1) % 3];
                        const double a0 = A[axis], b0 = B[axis];
                        if (a0 >= p0 && a0 <= p1) { const double q[3] = { A[0], A[1], A[2] }; add(q); }
                        for (const double pl : { p0, p1 }) {
                            if ((a0 < pl && b0 > pl) || (a0 > pl && b0 < pl)) {
                                const double w = (pl - a0) / (b0 - a0);
                                double q[3];
                                for (int a = 0; a < 3; ++a) q[a] = A[a] + (static_cast<double>(B[a]) - A[a]) * w;
                                q[axis] = pl;
                                add(q);
                            }
                        }
                    }
                    if (!any) continue;
                    BPrim& p = out[made++];
                    for (int a = 0; a < 3; ++a) {                 // outward to float, a relative hair wider, never beyond the whole box
                        const double m = 1e-6 * (std::fabs(lo[a]) + std::fabs(hi[a])) + 1e-30;
                        float l = static_cast<float>(lo[a] - m), h = static_cast<float>(hi[a] + m);
                        if (static_cast<double>(l) > lo[a] - m) l = std::nextafter(l, -std::numeric_limits<float>::infinity());
                        if (static_cast<double>(h) < hi[a] + m) h = std::nextafter(h, std::numeric_limits<float>::infinity());
                        p.lo[a] = std::max(l, whole.lo[a]);
                        p.hi[a] = std::min(h, whole.hi[a]);
                    }
                    p.id = whole.id;
                }
            }
            if (made == 0) out[made++] = whole;
            for (int f = 0; f < made; ++f)
                for (int a = 0; a < 3; ++a) out[f].c[a] = 0.5f * out[f].lo[a] + 0.5f * out[f].hi[a];
            for (int f = made; f < presplit; ++f) {              // unused part slots: removed below
                out[f].lo[0] = std::numeric_limits<float>::quiet_NaN();
                out[f].hi[0] = 0.0f;
            }
        }
        uint32_t bits = floatBits(chunkPad), seen = padMaxBits.load();
        while (bits > seen && !padMaxBits.compare_exchange_weak(seen, bits)) { }
    });
    out.padMax = bitsFloat(padMaxBits.load());
    parallelChunks(nSpheres, buildThreads(), size_t(1) << 16, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            // the box of everything the fp64 sphere test (include/sphere.h:63-97) can accept from a covered ray: radius
            // sqrt(r^2 + 2^-20 D^2) (cover.h), rounded outward
            const float* s = spheres + i * 4;
            BPrim& p = prims[nTris * perTri + i];
            sphereBox(s, std::sqrt(sphereReff2(s, cover)), p.lo, p.hi);
            for (int a = 0; a < 3; ++a) p.c[a] = s[a];
            if (!sphereFinite(s)) p.lo[0] = std::numeric_limits<float>::quiet_NaN();       // never accepted: left out below
            p.id = kSphereId | static_cast<uint32_t>(i);
        }
    });
    // NaN and infinite coordinates would poison the binning arithmetic; such primitives can never be hit
    // (cover.h: triBand, sphereFinite), so they are left out of the tree.
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
        const char* greedy = std::getenv("YART_WIDE_GREEDY_COLLAPSE");
        const bool plan = !(greedy && greedy[0] == '1');
        const double t1 = now();
        Emitter e(b.nodes(), prims, tri, spheres, bandClass, out);
        e.planned_ = plan;
        e.run(prims.size() < kSpawnAbove ? 1u : buildThreads());
        if (timing) fprintf(stderr, "wide build: binary SAH %.2fs, collapse+emit %.2fs (%zu nodes, depth %d)\n", t1 - t0, now() - t1, out.nodes.size(), out.depth);
        if (e.tooManyPrims) {
            err = "wide BVH: more than 2^27 - 1 primitives";
            return YART_ERR_INVALID_ARGUMENT;
        }
        if (!prims.empty()) {
            std::memcpy(out.rootLo, b.nodes()[0].box.lo, 12);
            std::memcpy(out.rootHi, b.nodes()[0].box.hi, 12);
        }
        if (3 * out.depth + 1 <= kWideStack) {
            relayout(out, envInt("YART_WIDE_LAYOUT", 0));
            return YART_OK;
        }
        // SAH produced a tree too deep for the traversal stack: rebuild balanced (object median)
    }
    err = "wide BVH deeper than the traversal stack allows";
    return YART_ERR_TREE_TOO_DEEP;
}

} // namespace yart
