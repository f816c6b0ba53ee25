// wide_build_dev.hip -- the engine's 4-wide tree (wide_build.h) built ON THE GPU.
//
// The reference builds its tree on one host thread (src/bvhaccel.cpp:100-151, :309-363); the library's
// host builder (wide_build.cpp, binned SAH on a thread pool) needs 3 s for the 11 M primitives of BASELINE
// configs[3].  Here the same arrays -- 64-byte quantised nodes, first- and second-stage leaf rows, one
// primitive per leaf -- come out of a chain of data-parallel kernels in tens of milliseconds:
//
//   primBoxes   box + centre per primitive (triangles from the vertices, spheres widened as on the host)
//   codes       63-bit Morton code of the centre inside the bounds of all centres; radix sort (hipcub)
//   hierarchy   the linear-BVH topology over the sorted codes: every internal node finds its range and
//               split from code prefixes alone (Karras 2012), no dependence between nodes
//   levels      depth of every node, top-down, one launch per level
//   fit + plan  boxes bottom-up and, with them, the collapse plan of wide_build.cpp (least summed area
//               of the wide nodes, a dynamic programme over (node, slots)), one launch per level: a
//               kernel boundary is the only ordering between a node and its children
//   emit        wide nodes top-down, one launch per wide level, children numbered by one atomic per wave
//   rows        leaf rows in Morton order of the primitives (neighbours in space = neighbours in memory)
//
// Tree quality: spatial-median splits instead of binned SAH: +6 % node visits per ray on the uniform soup,
// +9..17 % on the architectural stand-in (tools/tree_quality.py, YART_WIDE_BUILDER=lbvh is the same
// topology built on the host).  The traversal kernel, the node format and every primitive test are the
// same, so results are identical to the host-built tree's up to exact-t ties.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"
#include "wide_build.h"
#include "wide_build_dev.h"
#include "cover.h"

namespace yart {
namespace {

constexpr uint32_t kNoParent = 0xFFFFFFFFu;

#define YART_HIP(call)                                                                              \
    do {                                                                                            \
        const hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return YART_ERR_HIP; } \
    } while (0)

struct Plan { float cost[4]; uint8_t take[4]; };

__device__ inline uint32_t orderedBits(float f)           // monotone float -> uint
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float fromOrdered(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// bounds[0..2] = min of centres (ordered bits), [3..5] = max, [6] = valid primitives
__global__ void __launch_bounds__(256) primBoxes(const float* __restrict__ verts, const int32_t* __restrict__ tris, uint32_t nTris,
                                                 const float* __restrict__ spheres, uint32_t nSpheres, uint32_t nVerts, Cover cover,
                                                 float4* __restrict__ lo, float4* __restrict__ hi, uint32_t* __restrict__ bounds)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = nTris + nSpheres;
    float l[3] = { 0, 0, 0 }, h[3] = { 0, 0, 0 };
    bool valid = false;
    if (i < nTris) {
        const int32_t* t = tris + (size_t)i * 5;
        valid = true;
        for (int k = 0; k < 3; ++k) {
            const uint32_t vi = (uint32_t)t[k];
            if (vi >= nVerts) { valid = false; break; }          // cannot happen: the reference-tree build rejected it
            const float* v = verts + (size_t)vi * 3;
            for (int a = 0; a < 3; ++a) {
                l[a] = k == 0 ? v[a] : fminf(l[a], v[a]);
                h[a] = k == 0 ? v[a] : fmaxf(h[a], v[a]);
            }
        }
        // fminf / fmaxf drop a NaN operand where std::min / std::max keep the first: test the vertices themselves
        if (valid)
            for (int k = 0; k < 3; ++k) {
                const float* v = verts + (size_t)(uint32_t)t[k] * 3;
                if (!(v[0] == v[0] && v[1] == v[1] && v[2] == v[2])) valid = false;
            }
        if (valid) {
            // grown by the error band of the triangle's Wald test, as wide_build.cpp does (cover.h)
            const float *va = verts + (size_t)(uint32_t)t[0] * 3, *vb = verts + (size_t)(uint32_t)t[1] * 3, *vc = verts + (size_t)(uint32_t)t[2] * 3;
            float rec[12];
            uint32_t cls;
            projectRecord(rec, va, vb, vc, t[4], t[3]);
            triBand(rec, va, vb, vc, cover, l, h, cls);
            const float ex = bandExcess(l, h, va, vb, vc);              // Cover::padMax
            if (ex > 0.0f) atomicMax(bounds + 7, __float_as_uint(ex));
        }
    } else if (i < n) {
        const float* s = spheres + (size_t)(i - nTris) * 4;
        sphereBox(s, sqrt(sphereReff2(s, cover)), l, h);        // as wide_build.cpp
        valid = sphereFinite(s);
    }
    if (valid) for (int a = 0; a < 3; ++a) if (!(l[a] <= h[a])) valid = false;      // NaN: can never be hit, left out
    if (i < n) {
        lo[i] = make_float4(l[0], l[1], l[2], valid ? 1.0f : 0.0f);
        hi[i] = make_float4(h[0], h[1], h[2], 0.0f);
    }
    // centre bounds: wave reduction, one atomic per wave and component
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * l[a] + 0.5f * h[a];
        uint32_t mn = valid ? orderedBits(c) : 0xFFFFFFFFu, mx = valid ? orderedBits(c) : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, (uint32_t)__shfl_down(mn, o));
            mx = max(mx, (uint32_t)__shfl_down(mx, o));
        }
        if ((threadIdx.x & 63u) == 0) { atomicMin(bounds + a, mn); atomicMax(bounds + 3 + a, mx); }
    }
    const uint64_t m = __ballot(valid);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(bounds + 6, (uint32_t)__popcll(m));
}

__device__ inline uint64_t spread21(uint64_t v)
{
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(256) mortonCodes(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n,
                                                   const uint32_t* __restrict__ bounds, uint64_t* __restrict__ codes, uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 l = lo[i], h = hi[i];
    ids[i] = i;
    if (l.w == 0.0f) { codes[i] = ~0ull; return; }          // invalid primitives sort to the end
    const float c[3] = { 0.5f * l.x + 0.5f * h.x, 0.5f * l.y + 0.5f * h.y, 0.5f * l.z + 0.5f * h.z };
    uint64_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float mn = fromOrdered(bounds[a]), mx = fromOrdered(bounds[3 + a]);
        const float e = mx - mn;
        const float inv = (e > 0.0f && e < __builtin_inff()) ? 2097151.0f / e : 0.0f;
        float f = (c[a] - mn) * inv;
        f = f < 0.0f ? 0.0f : (f > 2097151.0f ? 2097151.0f : f);
        q[a] = (uint64_t)f;
    }
    codes[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
}

// length of the common prefix of the keys (code, position) of sorted positions i and j; -1 outside the array
__device__ inline int prefix(const uint64_t* codes, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t a = codes[i], b = codes[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Internal node i of n - 1; leaf j is node (n - 1) + j.  left/right: child node numbers.
__global__ void __launch_bounds__(256) hierarchy(const uint64_t* __restrict__ codes, int n, uint32_t* __restrict__ left,
                                                 uint32_t* __restrict__ right, uint32_t* __restrict__ parent,
                                                 uint32_t* __restrict__ rangeLo, uint32_t* __restrict__ rangeHi)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = prefix(codes, n, i, i + 1) - prefix(codes, n, i, i - 1) > 0 ? 1 : -1;
    const int dMin = prefix(codes, n, i, i - d);
    int lMax = 2;
    while (prefix(codes, n, i, i + lMax * d) > dMin) lMax *= 2;
    int l = 0;
    for (int t = lMax / 2; t >= 1; t /= 2)
        if (prefix(codes, n, i, i + (l + t) * d) > dMin) l += t;
    const int j = i + l * d;
    const int dNode = prefix(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (t < 1) break;
        if (prefix(codes, n, i, i + (s + t) * d) > dNode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t lc = (first == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    const uint32_t rc = (last == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    left[i] = lc;
    right[i] = rc;
    rangeLo[i] = (uint32_t)first;
    rangeHi[i] = (uint32_t)last;
    parent[lc] = (uint32_t)i;
    parent[rc] = (uint32_t)i;
    if (i == 0) parent[0] = kNoParent;
}

// ---- SAH below the Morton order.  Where a linear BVH loses against binned SAH is in its subtrees of a few hundred
// primitives (profiles/r02_exp_hybrid_tree_quality.txt: Morton splits down to 256 primitives and SAH below give the
// SAH tree's node visits within 1 %, Morton splits all the way 6 % more).  So every maximal subtree of at most T
// primitives is split again, top-down with binned SAH (16 bins, three axes, as wide_build.cpp), by ONE thread: the
// subtrees are independent, there are tens of thousands of them, and each is a few thousand operations on 12 KB.
// The subtree over sorted positions [f, l] keeps its root; its other l - f - 1 internal nodes are exactly the
// numbers f + 1 .. l - 1 (every internal node of the radix tree is numbered by an end of its range, a left child
// by the last position, a right child by the first), which the new topology reuses; the primitives are permuted
// inside [f, l] only.
__global__ void __launch_bounds__(256) treeletRoots(const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi,
                                                    const uint32_t* __restrict__ parent, uint32_t nInternal, uint32_t maxSize,
                                                    uint32_t* __restrict__ roots, uint32_t* __restrict__ count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool root = false;
    if (i < nInternal) {
        const uint32_t size = rangeHi[i] - rangeLo[i] + 1u;
        if (size >= 3u && size <= maxSize) {
            const uint32_t p = parent[i];
            root = p == kNoParent || rangeHi[p] - rangeLo[p] + 1u > maxSize;
        }
    }
    const uint64_t m = __ballot(root);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (root) roots[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

struct SahBins { float lo[16][3], hi[16][3]; uint32_t cnt[16]; };

__device__ inline float boxHalfArea(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(64) treeletSah(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ nRoots,
                                                 const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi, uint32_t nLeaves,
                                                 uint32_t* __restrict__ sortedIds, const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                                 uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ parent)
{
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= *nRoots) return;
    const uint32_t rootNode = roots[t];
    const uint32_t f = rangeLo[rootNode], l = rangeHi[rootNode];
    uint32_t nextId = f + 1u;                       // internal nodes f + 1 .. l - 1 are this subtree's to reuse
    constexpr int kBins = 16;
    struct Todo { uint32_t node, b, e; };
    Todo stack[12];                                 // the larger part of every split waits here: at most log2(T) + 1 entries
    int sp = 0;
    Todo cur{ rootNode, f, l + 1u };
    for (;;) {
        const uint32_t count = cur.e - cur.b;       // >= 2
        float cLo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, cHi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
        for (uint32_t i = cur.b; i < cur.e; ++i) {
            const uint32_t id = sortedIds[i];
            const float4 pl = primLo[id], ph = primHi[id];
            const float c[3] = { 0.5f * pl.x + 0.5f * ph.x, 0.5f * pl.y + 0.5f * ph.y, 0.5f * pl.z + 0.5f * ph.z };
            for (int a = 0; a < 3; ++a) { cLo[a] = fminf(cLo[a], c[a]); cHi[a] = fmaxf(cHi[a], c[a]); }
        }
        int bestAxis = -1, bestSplit = 0;
        float bestCost = __builtin_inff(), bestLo = 0.0f, bestScale = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float ext = cHi[a] - cLo[a];
            if (!(ext > 0.0f) || !(ext < __builtin_inff())) continue;
            SahBins bins;
            for (int k = 0; k < kBins; ++k) {
                bins.cnt[k] = 0;
                for (int d = 0; d < 3; ++d) { bins.lo[k][d] = __builtin_inff(); bins.hi[k][d] = -__builtin_inff(); }
            }
            const float scale = (float)kBins / ext;
            for (uint32_t i = cur.b; i < cur.e; ++i) {
                const uint32_t id = sortedIds[i];
                const float4 pl = primLo[id], ph = primHi[id];
                const float pc = a == 0 ? 0.5f * pl.x + 0.5f * ph.x : (a == 1 ? 0.5f * pl.y + 0.5f * ph.y : 0.5f * pl.z + 0.5f * ph.z);
                int k = (int)((pc - cLo[a]) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                bins.lo[k][0] = fminf(bins.lo[k][0], pl.x); bins.lo[k][1] = fminf(bins.lo[k][1], pl.y); bins.lo[k][2] = fminf(bins.lo[k][2], pl.z);
                bins.hi[k][0] = fmaxf(bins.hi[k][0], ph.x); bins.hi[k][1] = fmaxf(bins.hi[k][1], ph.y); bins.hi[k][2] = fmaxf(bins.hi[k][2], ph.z);
                ++bins.cnt[k];
            }
            float rightArea[kBins];
            uint32_t rightCnt[kBins];
            float accLo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, accHi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
            uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) {
                for (int d = 0; d < 3; ++d) { accLo[d] = fminf(accLo[d], bins.lo[k][d]); accHi[d] = fmaxf(accHi[d], bins.hi[k][d]); }
                c += bins.cnt[k];
                rightArea[k] = boxHalfArea(accLo, accHi);
                rightCnt[k] = c;
            }
            for (int d = 0; d < 3; ++d) { accLo[d] = __builtin_inff(); accHi[d] = -__builtin_inff(); }
            c = 0;
            for (int k = 1; k < kBins; ++k) {              // split k: bins [0, k) left, [k, kBins) right
                for (int d = 0; d < 3; ++d) { accLo[d] = fminf(accLo[d], bins.lo[k - 1][d]); accHi[d] = fmaxf(accHi[d], bins.hi[k - 1][d]); }
                c += bins.cnt[k - 1];
                if (c == 0 || rightCnt[k] == 0) continue;
                const float cost = boxHalfArea(accLo, accHi) * (float)c + rightArea[k] * (float)rightCnt[k];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; bestLo = cLo[a]; bestScale = scale; }
            }
        }
        uint32_t mid = cur.b + count / 2u;                  // all centres coincide: halve the range as it stands
        if (bestAxis >= 0) {
            uint32_t i = cur.b, j = cur.e;                  // in-place partition: bin < bestSplit to the front
            while (i < j) {
                const uint32_t id = sortedIds[i];
                const float4 pl = primLo[id], ph = primHi[id];
                const float pc = bestAxis == 0 ? 0.5f * pl.x + 0.5f * ph.x : (bestAxis == 1 ? 0.5f * pl.y + 0.5f * ph.y : 0.5f * pl.z + 0.5f * ph.z);
                int k = (int)((pc - bestLo) * bestScale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                if (k < bestSplit) ++i;
                else { --j; sortedIds[i] = sortedIds[j]; sortedIds[j] = id; }
            }
            if (i > cur.b && i < cur.e) mid = i;
        }
        const uint32_t nl = mid - cur.b, nr = cur.e - mid;
        const uint32_t childL = nl == 1u ? nLeaves - 1u + cur.b : nextId++;
        const uint32_t childR = nr == 1u ? nLeaves - 1u + mid : nextId++;
        left[cur.node] = childL;
        right[cur.node] = childR;
        parent[childL] = cur.node;
        parent[childR] = cur.node;
        const Todo tl{ childL, cur.b, mid }, tr{ childR, mid, cur.e };
        // the smaller part next, the larger waits: the stack stays logarithmic
        const bool leftFirst = nl <= nr;
        const Todo& first = leftFirst ? tl : tr;
        const Todo& second = leftFirst ? tr : tl;
        const uint32_t nFirst = leftFirst ? nl : nr, nSecond = leftFirst ? nr : nl;
        if (nSecond >= 2u && nFirst >= 2u) { stack[sp++] = second; cur = first; continue; }
        if (nFirst >= 2u) { cur = first; continue; }
        if (nSecond >= 2u) { cur = second; continue; }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

// The same splits by ONE WAVE per subtree (round 3; the one-thread form above took 60 ms of C4's build and stays for
// treelets above kWaveTreelet).  The subtree's boxes sit in LDS; per range the wave reduces the bounds of the centres,
// bins all three axes in one pass with LDS integer min / max on order-preserving keys (a box union does not depend on
// the order of its operands, so the bins are the one-thread form's), evaluates the 45 candidate planes one per lane
// with the one-thread form's expressions and takes the cheapest, ties to the earlier (axis, plane) like its loop, and
// partitions stably by ballot ranks.  Same subtrees as sets of primitives; the order of the primitives inside a part
// (and so the node numbers) is this kernel's own -- deterministic, and nothing downstream depends on it.
constexpr int kWaveTreelet = 256;

__device__ inline uint32_t orderedKey(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float fromOrderedKey(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ inline float waveMinF(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ inline float waveMaxF(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

__global__ void __launch_bounds__(64) treeletSahWave(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ nRoots,
                                                     const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi, uint32_t nLeaves,
                                                     uint32_t* __restrict__ sortedIds, const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                                     uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ parent)
{
    constexpr int T = kWaveTreelet, kBins = 16;
    __shared__ float bl[3][T], bh[3][T];
    __shared__ uint32_t ids[T], idx[T], tmp[T];
    __shared__ uint32_t binLo[3][kBins][3], binHi[3][kBins][3], binCnt[3][kBins];
    const uint32_t lane = threadIdx.x;
    const uint32_t total = *nRoots;
    for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint32_t rootNode = roots[t];
        const uint32_t f = rangeLo[rootNode], l = rangeHi[rootNode];
        const uint32_t size = l - f + 1u;               // 3 .. T
        __syncthreads();                                // one wave: orders this subtree's LDS writes after the last one's reads
        for (uint32_t i = lane; i < size; i += 64u) {
            const uint32_t id = sortedIds[f + i];
            const float4 pl = primLo[id], ph = primHi[id];
            bl[0][i] = pl.x; bl[1][i] = pl.y; bl[2][i] = pl.z;
            bh[0][i] = ph.x; bh[1][i] = ph.y; bh[2][i] = ph.z;
            ids[i] = id;
            idx[i] = i;
        }
        __syncthreads();
        uint32_t nextId = f + 1u;
        struct Todo { uint32_t node, b, e; };
        Todo stack[10];
        int sp = 0;
        Todo cur{ rootNode, 0u, size };
        for (;;) {
            const uint32_t count = cur.e - cur.b;       // >= 2
            uint32_t mid = cur.b + count / 2u;
            if (count > 2u) {
                float cLo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, cHi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
                for (uint32_t i = cur.b + lane; i < cur.e; i += 64u) {
                    const uint32_t j = idx[i];
                    for (int a = 0; a < 3; ++a) { const float c = 0.5f * bl[a][j] + 0.5f * bh[a][j]; cLo[a] = fminf(cLo[a], c); cHi[a] = fmaxf(cHi[a], c); }
                }
                float scale[3];
                bool use[3];
                for (int a = 0; a < 3; ++a) {
                    cLo[a] = waveMinF(cLo[a]); cHi[a] = waveMaxF(cHi[a]);
                    const float ext = cHi[a] - cLo[a];
                    use[a] = (ext > 0.0f) && (ext < __builtin_inff());
                    scale[a] = use[a] ? (float)kBins / ext : 0.0f;
                }
                for (uint32_t i = lane; i < 3u * kBins; i += 64u) {
                    const uint32_t a = i / kBins, k = i % kBins;
                    for (int d = 0; d < 3; ++d) { binLo[a][k][d] = orderedKey(__builtin_inff()); binHi[a][k][d] = orderedKey(-__builtin_inff()); }
                    binCnt[a][k] = 0u;
                }
                __syncthreads();
                for (uint32_t i = cur.b + lane; i < cur.e; i += 64u) {
                    const uint32_t j = idx[i];
                    uint32_t kl[3], kh[3];
                    for (int d = 0; d < 3; ++d) { kl[d] = orderedKey(bl[d][j]); kh[d] = orderedKey(bh[d][j]); }
                    for (int a = 0; a < 3; ++a) {
                        if (!use[a]) continue;
                        const float pc = 0.5f * bl[a][j] + 0.5f * bh[a][j];
                        int k = (int)((pc - cLo[a]) * scale[a]);
                        k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                        for (int d = 0; d < 3; ++d) { atomicMin(&binLo[a][k][d], kl[d]); atomicMax(&binHi[a][k][d], kh[d]); }
                        atomicAdd(&binCnt[a][k], 1u);
                    }
                }
                __syncthreads();
                // candidate plane (a, k): bins [0, k) left, [k, kBins) right
                float cost = __builtin_inff();
                uint32_t leftCnt = 0;
                if (lane < 45u) {
                    const int a = (int)lane / 15, k = (int)lane % 15 + 1;
                    if (use[a]) {
                        float aLo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, aHi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
                        float rLo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, rHi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
                        uint32_t c = 0, rc = 0;
                        for (int q = 0; q < kBins; ++q) {
                            const uint32_t n = binCnt[a][q];
                            if (q < k) { c += n; for (int d = 0; d < 3; ++d) { aLo[d] = fminf(aLo[d], fromOrderedKey(binLo[a][q][d])); aHi[d] = fmaxf(aHi[d], fromOrderedKey(binHi[a][q][d])); } }
                            else { rc += n; for (int d = 0; d < 3; ++d) { rLo[d] = fminf(rLo[d], fromOrderedKey(binLo[a][q][d])); rHi[d] = fmaxf(rHi[d], fromOrderedKey(binHi[a][q][d])); } }
                        }
                        if (c != 0 && rc != 0) { cost = boxHalfArea(aLo, aHi) * (float)c + boxHalfArea(rLo, rHi) * (float)rc; leftCnt = c; }
                    }
                }
                const float best = waveMinF(cost);
                const uint64_t who = __ballot(cost == best && cost < __builtin_inff());
                if (who != 0) {
                    const int w = __builtin_ctzll(who);                 // the earliest (axis, plane) among the cheapest
                    const int bestAxis = w / 15, bestSplit = w % 15 + 1;
                    const uint32_t nLeft = (uint32_t)__shfl((int)leftCnt, w);
                    const float bLo = cLo[bestAxis], bScale = scale[bestAxis];
                    uint32_t lp = cur.b, rp = cur.b + nLeft;
                    for (uint32_t base = cur.b; base < cur.e; base += 64u) {
                        const uint32_t i = base + lane;
                        bool in = i < cur.e, goLeft = false;
                        uint32_t j = 0;
                        if (in) {
                            j = idx[i];
                            const float pc = 0.5f * bl[bestAxis][j] + 0.5f * bh[bestAxis][j];
                            int k = (int)((pc - bLo) * bScale);
                            k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                            goLeft = k < bestSplit;
                        }
                        const uint64_t lm = __ballot(in && goLeft), rm = __ballot(in && !goLeft);
                        const uint64_t below = (1ull << lane) - 1ull;
                        if (in) tmp[goLeft ? lp + (uint32_t)__popcll(lm & below) : rp + (uint32_t)__popcll(rm & below)] = j;
                        lp += (uint32_t)__popcll(lm);
                        rp += (uint32_t)__popcll(rm);
                    }
                    __syncthreads();
                    for (uint32_t i = cur.b + lane; i < cur.e; i += 64u) idx[i] = tmp[i];
                    __syncthreads();
                    mid = cur.b + nLeft;
                }
            }
            const uint32_t nl = mid - cur.b, nr = cur.e - mid;
            const uint32_t childL = nl == 1u ? nLeaves - 1u + f + cur.b : nextId++;
            const uint32_t childR = nr == 1u ? nLeaves - 1u + f + mid : nextId++;
            if (lane == 0) {
                left[cur.node] = childL;
                right[cur.node] = childR;
                parent[childL] = cur.node;
                parent[childR] = cur.node;
            }
            const Todo tl{ childL, cur.b, mid }, tr{ childR, mid, cur.e };
            const bool leftFirst = nl <= nr;
            const Todo first = leftFirst ? tl : tr, second = leftFirst ? tr : tl;
            const uint32_t nFirst = leftFirst ? nl : nr, nSecond = leftFirst ? nr : nl;
            if (nSecond >= 2u && nFirst >= 2u) { stack[sp++] = second; cur = first; continue; }
            if (nFirst >= 2u) { cur = first; continue; }
            if (nSecond >= 2u) { cur = second; continue; }
            if (sp == 0) break;
            cur = stack[--sp];
        }
        for (uint32_t i = lane; i < size; i += 64u) sortedIds[f + i] = ids[idx[i]];
    }
}

// depth of the children of every internal node at depth `level`; *any = 1 if there was such a node
__global__ void __launch_bounds__(256) assignLevel(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t nInternal,
                                                   uint32_t* __restrict__ depth, uint32_t level, uint32_t* __restrict__ any)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nInternal || depth[i] != level) return;
    depth[left[i]] = level + 1;
    depth[right[i]] = level + 1;
    *any = 1u;
}

__device__ inline float halfArea(const float4& l, const float4& h)
{
    const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
    if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

// leaves: node box = box of the primitive at that sorted position, plan = none (wide_build.cpp planNode)
__global__ void __launch_bounds__(256) leafNodes(const float4* __restrict__ primLo, const float4* __restrict__ primHi,
                                                 const uint32_t* __restrict__ sortedIds, uint32_t n, float4* __restrict__ nodeLo,
                                                 float4* __restrict__ nodeHi, Plan* __restrict__ plan)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint32_t node = n - 1 + j, p = sortedIds[j];
    nodeLo[node] = primLo[p];
    nodeHi[node] = primHi[p];
    Plan pl;
    for (int k = 0; k < 4; ++k) { pl.cost[k] = 0.0f; pl.take[k] = 0; }
    plan[node] = pl;
}

// internal nodes of one level: box = union of the children's, collapse plan from theirs (wide_build.cpp planNode)
__global__ void __launch_bounds__(256) fitLevel(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t nInternal,
                                                const uint32_t* __restrict__ depth, uint32_t level, float4* __restrict__ nodeLo,
                                                float4* __restrict__ nodeHi, Plan* __restrict__ plan)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nInternal || depth[i] != level) return;
    const uint32_t a = left[i], b = right[i];
    const float4 al = nodeLo[a], ah = nodeHi[a], bl = nodeLo[b], bh = nodeHi[b];
    const float4 l = make_float4(fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), 1.0f);
    const float4 h = make_float4(fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), 0.0f);
    nodeLo[i] = l;
    nodeHi[i] = h;
    const Plan pa = plan[a], pb = plan[b];
    float forest[5];
    uint8_t split[5];
    for (int k = 2; k <= 4; ++k) {
        forest[k] = __builtin_inff();
        split[k] = 1;
        for (int j = 1; j < k; ++j) {
            const float c = pa.cost[j - 1] + pb.cost[k - j - 1];
            if (c < forest[k]) { forest[k] = c; split[k] = (uint8_t)j; }
        }
    }
    Plan p;
    const float asNode = halfArea(l, h) + forest[4];
    p.cost[0] = asNode;
    p.take[0] = split[4];
    for (int k = 2; k <= 4; ++k) {
        if (forest[k] < asNode) { p.cost[k - 1] = forest[k]; p.take[k - 1] = split[k]; }
        else { p.cost[k - 1] = asNode; p.take[k - 1] = 0; }
    }
    plan[i] = p;
}

struct QueueEntry { uint32_t bnode, wide; };

// the binary nodes that fill `slots` slots of a parent for the subtree at b (wide_build.cpp gather), iteratively
__device__ inline void gatherKids(const uint32_t* left, const uint32_t* right, const Plan* plan, uint32_t nInternal,
                                  uint32_t b, int slots, uint32_t* kids, int& n)
{
    uint32_t stackNode[4];
    int stackSlots[4];
    int sp = 0;
    stackNode[0] = b; stackSlots[0] = slots; sp = 1;
    while (sp > 0) {
        --sp;
        const uint32_t x = stackNode[sp];
        const int s = stackSlots[sp];
        const bool leaf = x >= nInternal;
        const uint8_t tk = (leaf || s == 1) ? 0 : plan[x].take[s - 1];
        if (leaf || s == 1 || tk == 0) { kids[n++] = x; continue; }
        // right part first onto the stack so that the left part is emitted first
        stackNode[sp] = right[x]; stackSlots[sp] = s - tk; ++sp;
        stackNode[sp] = left[x]; stackSlots[sp] = tk; ++sp;
    }
}

// wide_build.cpp packedStep: the smallest step the packed node can hold (8 exponent + 2 mantissa bits) that is >= s
__device__ inline float packedStepDev(float s)
{
    uint32_t u = __float_as_uint(s);
    if (u & ~0x7FE00000u) u = (u & 0x7FE00000u) + 0x00200000u;
    if (u < 0x00800000u) u = 0x00800000u;
    if (u >= 0x7F800000u) u = 0x7F600000u;
    return __uint_as_float(u);
}

// Quantise up to four child boxes on the grid of their union (wide_build.cpp setChildren): containment is
// decided in double on the float values the traversal kernel will read.
__device__ inline void setChildren(WideNode& w, const float4* lo, const float4* hi, const uint32_t* refs, int n)
{
    for (int a = 0; a < 3; ++a) {
        w.origin[a] = 0.0f;
        w.scale[a] = 1.0f;
        for (int c = 0; c < 4; ++c) { w.qlo[a][c] = 255; w.qhi[a][c] = 0; }
    }
    for (int c = 0; c < 4; ++c) w.ref[c] = kWideNone;
    float cl[4][3], ch[4][3];
    for (int i = 0; i < n; ++i) {
        const float l3[3] = { lo[i].x, lo[i].y, lo[i].z }, h3[3] = { hi[i].x, hi[i].y, hi[i].z };
        for (int a = 0; a < 3; ++a) {                     // keep the grid arithmetic finite
            cl[i][a] = fmaxf(l3[a], -8.0e37f);
            ch[i][a] = fminf(h3[a], 8.0e37f);
        }
    }
    for (int a = 0; a < 3; ++a) {
        float allLo = __builtin_inff(), allHi = -__builtin_inff();
        for (int i = 0; i < n; ++i) { allLo = fminf(allLo, cl[i][a]); allHi = fmaxf(allHi, ch[i][a]); }
        const float origin = allLo;
        float scale = (allHi - allLo) / 255.0f;
        if (!(scale > 0.0f) || !(scale < __builtin_inff())) scale = 1.0f;      // flat or unbounded axis
        scale = packedStepDev(scale);
        for (int guard = 0; guard < 64; ++guard) {
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) {
                const double l = cl[i][a], h = ch[i][a];
                double ql = floor((l - (double)origin) / (double)scale), qh = ceil((h - (double)origin) / (double)scale);
                while (ql > 0.0 && (double)origin + ql * (double)scale > l) ql -= 1.0;
                while ((double)origin + qh * (double)scale < h) qh += 1.0;
                if (ql < 0.0) ql = 0.0;
                if (qh > 255.0) { ok = false; break; }
                w.qlo[a][i] = (uint8_t)ql;
                w.qhi[a][i] = (uint8_t)qh;
            }
            if (ok) break;
            scale = packedStepDev(nextafterf(scale, __builtin_inff()));
        }
        w.origin[a] = origin;
        w.scale[a] = scale;
    }
    for (int i = 0; i < n; ++i) w.ref[i] = refs[i];
}

// One wide level: every queue entry becomes a wide node; its interior children get node numbers (one atomic
// per wave) and go onto the next level's queue.
__global__ void __launch_bounds__(256) emitLevel(const QueueEntry* __restrict__ queue, uint32_t nQueue, QueueEntry* __restrict__ nextQueue,
                                                 uint32_t* __restrict__ counters /* [0] wide nodes so far, [1] next queue length */,
                                                 const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const Plan* __restrict__ plan,
                                                 uint32_t nInternal, uint32_t nTris, const uint32_t* __restrict__ sortedIds,
                                                 const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi, WideNode* __restrict__ out)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool active = q < nQueue;
    uint32_t kids[4];
    int n = 0;
    QueueEntry e{ 0, 0 };
    if (active) {
        e = queue[q];
        const int leftSlots = plan[e.bnode].take[0];
        gatherKids(left, right, plan, nInternal, left[e.bnode], leftSlots, kids, n);
        gatherKids(left, right, plan, nInternal, right[e.bnode], 4 - leftSlots, kids, n);
    }
    uint32_t nInterior = 0;
    for (int i = 0; i < n; ++i) nInterior += kids[i] < nInternal ? 1u : 0u;
    // wave prefix sum of nInterior -> node numbers and queue slots with one atomic pair per wave
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = nInterior;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t baseNode = 0, baseQueue = 0;
    if (lane == 63 && total) { baseNode = atomicAdd(counters + 0, total); baseQueue = atomicAdd(counters + 1, total); }
    baseNode = __shfl(baseNode, 63);
    baseQueue = __shfl(baseQueue, 63);
    if (!active) return;
    uint32_t k = incl - nInterior;
    float4 lo[4], hi[4];
    uint32_t refs[4];
    for (int i = 0; i < n; ++i) {
        lo[i] = nodeLo[kids[i]];
        hi[i] = nodeHi[kids[i]];
        if (kids[i] < nInternal) {
            refs[i] = baseNode + k;
            nextQueue[baseQueue + k] = QueueEntry{ kids[i], baseNode + k };
            ++k;
        } else {
            const uint32_t pos = kids[i] - nInternal;                 // sorted position = leaf row
            const bool sph = sortedIds[pos] >= nTris;
            refs[i] = kWideLeafBit | (sph ? kWideSphereBit : 0u) | pos;      // one primitive per leaf: count - 1 = 0
        }
    }
    WideNode w;
    setChildren(w, lo, hi, refs, n);
    out[e.wide] = w;
}

// ---- packing (wide_build.h: PackedWide).  The emitted nodes are in breadth-first order with the interior children
// of a node side by side; the block of node i's children follows the block of node i - 1's.
__device__ inline void childCounts(const WideNode& w, uint32_t& interior, uint32_t& leafUnits)
{
    interior = 0; leafUnits = 0;
    for (int c = 0; c < 4; ++c) {
        const uint32_t r = w.ref[c];
        if (r == kWideNone) continue;
        if (!(r & kWideLeafBit)) ++interior;
        else leafUnits += (r & kWideSphereBit) ? 2u : 1u;
    }
}

// Two regions (wide_build.cpp packWide): blocks with interior children, 64-byte aligned, and behind them the rows of
// the nodes with leaf children only.  Those rows keep the Morton order of their primitives (the leaves below one node
// are neighbours in that order), so neighbours in space share cache lines; the rows of leaves that live in a block of
// the first region leave their units of the second unused.
__global__ void __launch_bounds__(256) packSizes(const WideNode* __restrict__ nodes, uint32_t n, uint32_t* __restrict__ sizesN,
                                                 uint32_t* __restrict__ firstPos)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const WideNode w = nodes[i];
    uint32_t interior, leafUnits;
    childCounts(w, interior, leafUnits);
    sizesN[i] = interior ? 4u * interior + (leafUnits + 3u) / 4u * 4u : 0u;
    uint32_t first = 0;
    for (int c = 3; c >= 0; --c)
        if (w.ref[c] != kWideNone && (w.ref[c] & kWideLeafBit)) first = w.ref[c] & 0x07FFFFFFu;
    firstPos[i] = first;
}

__global__ void __launch_bounds__(256) rowUnits(const uint32_t* __restrict__ sortedIds, uint32_t n, uint32_t nTris, uint32_t* __restrict__ units)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < n) units[r] = sortedIds[r] < nTris ? 1u : 2u;
}

// block[i]: unit index of node i's children block
__global__ void __launch_bounds__(256) packBlocks(const uint32_t* __restrict__ sizesN, const uint32_t* __restrict__ sumN,
                                                  const uint32_t* __restrict__ firstPos, const uint32_t* __restrict__ rowSum, uint32_t n,
                                                  uint32_t rowsBase, uint32_t* __restrict__ block)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    block[i] = sizesN[i] ? 4u + sumN[i] : rowsBase + rowSum[firstPos[i]];
}

// where every node and every leaf row goes
__global__ void __launch_bounds__(256) packAssign(const WideNode* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ block,
                                                  uint32_t* __restrict__ unitOf, uint32_t* __restrict__ leafUnit)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const WideNode w = nodes[i];
    uint32_t interior, leafUnits;
    childCounts(w, interior, leafUnits);
    const uint32_t base = block[i];
    uint32_t nextInterior = 0, nextLeaf = 4u * interior;
    for (int c = 0; c < 4; ++c) {
        const uint32_t r = w.ref[c];
        if (r == kWideNone) continue;
        if (!(r & kWideLeafBit)) { unitOf[r] = base + 4u * nextInterior++; continue; }
        leafUnit[r & 0x07FFFFFFu] = base + nextLeaf;
        nextLeaf += (r & kWideSphereBit) ? 2u : 1u;
    }
    if (i == 0) unitOf[0] = 0u;
}

__global__ void __launch_bounds__(256) packNodes(const WideNode* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ block,
                                                 const uint32_t* __restrict__ unitOf, uint4* __restrict__ units)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const WideNode w = nodes[i];
    uint32_t interior, leafUnits;
    childCounts(w, interior, leafUnits);
    uint32_t steps = 0, lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        steps |= (__float_as_uint(w.scale[a]) & 0x7FE00000u) >> (10 * a);
        lo[a] = (uint32_t)w.qlo[a][0] | ((uint32_t)w.qlo[a][1] << 8) | ((uint32_t)w.qlo[a][2] << 16) | ((uint32_t)w.qlo[a][3] << 24);
        hi[a] = (uint32_t)w.qhi[a][0] | ((uint32_t)w.qhi[a][1] << 8) | ((uint32_t)w.qhi[a][2] << 16) | ((uint32_t)w.qhi[a][3] << 24);
    }
    uint32_t offs = 0, nextInterior = 0, nextLeaf = 4u * interior;
    for (int c = 0; c < 4; ++c) {
        const uint32_t r = w.ref[c];
        if (r == kWideNone) continue;
        if (!(r & kWideLeafBit)) { offs |= ((4u * nextInterior++) << 2) << (8 * c); continue; }
        const bool sph = (r & kWideSphereBit) != 0;
        offs |= ((nextLeaf << 2) | kPackLeaf | (sph ? kPackSphere : 0u)) << (8 * c);
        nextLeaf += sph ? 2u : 1u;
    }
    uint4* u = units + unitOf[i];
    u[0] = make_uint4(__float_as_uint(w.origin[0]), __float_as_uint(w.origin[1]), __float_as_uint(w.origin[2]), steps);
    u[1] = make_uint4(lo[0], lo[1], lo[2], hi[0]);
    u[2] = make_uint4(hi[1], hi[2], block[i] << 2, offs);
}

// leaf rows: the reference's TriAccel bits of the triangle's slot / the sphere record, at the unit packAssign chose
__global__ void __launch_bounds__(256) leafRows(const uint32_t* __restrict__ sortedIds, uint32_t n, uint32_t nTris,
                                                const uint32_t* __restrict__ slotOfInput, const float4* __restrict__ triaccel,
                                                const float4* __restrict__ spheres, const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                Cover cover, const uint32_t* __restrict__ leafUnit, float4* __restrict__ units)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t p = sortedIds[r];
    float4* row = units + leafUnit[r];
    if (p < nTris) {
        const uint32_t slot = slotOfInput[p];
        const float4 a = triaccel[(size_t)slot * 3 + 0], b = triaccel[(size_t)slot * 3 + 1], c = triaccel[(size_t)slot * 3 + 2];
        const uint32_t k = __float_as_uint(a.w);                        // TriAccel::k (0..2)
        // class of the triangle's error band (cover.h), from the record itself
        const float rec[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
        const int32_t* t = tris + (size_t)p * 5;
        float l[3], h[3];
        uint32_t cls;
        triBand(rec, verts + (size_t)(uint32_t)t[0] * 3, verts + (size_t)(uint32_t)t[1] * 3, verts + (size_t)(uint32_t)t[2] * 3, cover, l, h, cls);
        row[0] = make_float4(a.x, a.y, a.z, __uint_as_float((k & 3u) | (slot << 2) | (cls << kPackBandShift)));
    } else {
        const uint32_t s = p - nTris;
        row[0] = spheres[s];
        row[1] = make_float4(__uint_as_float(s), 0.0f, 0.0f, 0.0f);
    }
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

uint32_t blocks(size_t n) { return static_cast<uint32_t>((n + 255) / 256); }

} // namespace

struct DeviceWideBuilder::State {
    Scratch S;
    size_t nTris = 0, nSpheres = 0, n = 0, wideCount = 0;
    int wideDepth = 0;
    const uint32_t* sortedIds = nullptr;
    float* dSpheres = nullptr;
    float* dVerts = nullptr;
    int32_t* dTris = nullptr;
    WideNode* wide = nullptr;
    Cover cover;
    float padMax = 0.0f;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };     // bounds of the padded primitive boxes = of every node
    std::chrono::steady_clock::time_point t0;
    double msTopology = 0.0;
};

DeviceWideBuilder::DeviceWideBuilder() = default;
DeviceWideBuilder::~DeviceWideBuilder() { delete state_; }

// Phase 1: everything that needs only the scene -- boxes, sort, topology, fit, plan, wide nodes.
int DeviceWideBuilder::topology(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris, const float* spheres,
                                size_t nSpheres, const Cover& cover, std::string& err)
{
    delete state_;
    state_ = nullptr;
    const size_t nP = nTris + nSpheres;
    if (nP < 2 || nP > kWideMaxPrims || nVerts > 0xFFFFFFFFull) { err = "device builder: primitive count out of its range"; return YART_ERR_INVALID_ARGUMENT; }
    State* st = new State();
    state_ = st;
    st->t0 = std::chrono::steady_clock::now();
    st->nTris = nTris;
    st->nSpheres = nSpheres;
    st->cover = cover;
    Scratch& S = st->S;
    float* dVerts; int32_t* dTris; float* dSpheres;
    float4 *primLo, *primHi;
    uint32_t* dBounds;
    YART_HIP(S.alloc(&dVerts, nVerts * 3));
    YART_HIP(S.alloc(&dTris, nTris * 5));
    YART_HIP(S.alloc(&dSpheres, nSpheres * 4));
    YART_HIP(S.alloc(&primLo, nP));
    YART_HIP(S.alloc(&primHi, nP));
    YART_HIP(S.alloc(&dBounds, 8));
    st->dSpheres = dSpheres;
    st->dVerts = dVerts;
    st->dTris = dTris;
    if (nVerts) YART_HIP(hipMemcpyAsync(dVerts, verts, nVerts * 12, hipMemcpyHostToDevice, nullptr));
    if (nTris) YART_HIP(hipMemcpyAsync(dTris, tris, nTris * 20, hipMemcpyHostToDevice, nullptr));
    if (nSpheres) YART_HIP(hipMemcpyAsync(dSpheres, spheres, nSpheres * 16, hipMemcpyHostToDevice, nullptr));
    const uint32_t initBounds[8] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u };
    YART_HIP(hipMemcpyAsync(dBounds, initBounds, sizeof initBounds, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(primBoxes, dim3(blocks(nP)), dim3(256), 0, nullptr, dVerts, dTris, (uint32_t)nTris, dSpheres, (uint32_t)nSpheres,
                       (uint32_t)nVerts, cover, primLo, primHi, dBounds);
    YART_HIP(hipGetLastError());

    // Morton codes, sorted
    uint64_t *codesA, *codesB;
    uint32_t *idsA, *idsB;
    YART_HIP(S.alloc(&codesA, nP));
    YART_HIP(S.alloc(&codesB, nP));
    YART_HIP(S.alloc(&idsA, nP));
    YART_HIP(S.alloc(&idsB, nP));
    hipLaunchKernelGGL(mortonCodes, dim3(blocks(nP)), dim3(256), 0, nullptr, primLo, primHi, (uint32_t)nP, dBounds, codesA, idsA);
    YART_HIP(hipGetLastError());
    hipcub::DoubleBuffer<uint64_t> keys(codesA, codesB);
    hipcub::DoubleBuffer<uint32_t> vals(idsA, idsB);
    size_t tempBytes = 0;
    YART_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keys, vals, (int)nP, 0, 64, nullptr));
    char* temp;
    YART_HIP(S.alloc(&temp, tempBytes));
    YART_HIP(hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keys, vals, (int)nP, 0, 64, nullptr));
    const uint64_t* codes = keys.Current();
    const uint32_t* sortedIds = vals.Current();
    st->sortedIds = sortedIds;
    uint32_t hostBounds[8];
    YART_HIP(hipMemcpy(hostBounds, dBounds, sizeof hostBounds, hipMemcpyDeviceToHost));
    std::memcpy(&st->padMax, &hostBounds[7], 4);
    const size_t n = hostBounds[6];                       // valid primitives: the first n sorted positions
    if (n < 2) { err = "device builder: fewer than two valid primitives"; return YART_ERR_INVALID_ARGUMENT; }
    st->n = n;

    // topology
    const size_t nInternal = n - 1, nNodes = 2 * n - 1;
    uint32_t *left, *right, *parent, *depth, *dAny, *rangeLo, *rangeHi, *roots;
    YART_HIP(S.alloc(&left, nInternal));
    YART_HIP(S.alloc(&right, nInternal));
    YART_HIP(S.alloc(&rangeLo, nInternal));
    YART_HIP(S.alloc(&rangeHi, nInternal));
    YART_HIP(S.alloc(&roots, nInternal));
    YART_HIP(S.alloc(&parent, nNodes));
    YART_HIP(S.alloc(&depth, nNodes));
    YART_HIP(S.alloc(&dAny, 64));
    hipLaunchKernelGGL(hierarchy, dim3(blocks(nInternal)), dim3(256), 0, nullptr, codes, (int)n, left, right, parent, rangeLo, rangeHi);
    YART_HIP(hipGetLastError());
    YART_HIP(hipMemsetAsync(dAny, 0, 256, nullptr));
    {
        // binned SAH inside the subtrees of at most `treelet` primitives (YART_DEVICE_TREELET, 0 = keep the Morton splits)
        uint32_t treelet = 256;
        if (const char* v = std::getenv("YART_DEVICE_TREELET")) treelet = static_cast<uint32_t>(std::atoi(v));
        if (treelet > 2048) treelet = 2048;             // treeletSah's stack: log2 of it + 1 entries
        if (treelet >= 3) {
            hipLaunchKernelGGL(treeletRoots, dim3(blocks(nInternal)), dim3(256), 0, nullptr, rangeLo, rangeHi, parent, (uint32_t)nInternal, treelet, roots, dAny + 32);
            // one thread per root: the count stays on the device, the grid covers the most there can be
            const size_t maxRoots = nInternal / 2 + 1;
            const char* oneThread = std::getenv("YART_DEVICE_TREELET_THREAD");
            if (treelet <= (uint32_t)kWaveTreelet && !(oneThread && oneThread[0] == '1')) {
                // one wave per root, the waves of a fixed grid taking roots in turn (the count stays on the device)
                const size_t want = maxRoots < 65536 ? maxRoots : 65536;
                hipLaunchKernelGGL(treeletSahWave, dim3(static_cast<uint32_t>(want)), dim3(64), 0, nullptr, roots, dAny + 32, rangeLo, rangeHi, (uint32_t)n,
                                   const_cast<uint32_t*>(sortedIds), primLo, primHi, left, right, parent);
            } else
            hipLaunchKernelGGL(treeletSah, dim3(static_cast<uint32_t>((maxRoots + 63) / 64)), dim3(64), 0, nullptr, roots, dAny + 32, rangeLo, rangeHi, (uint32_t)n,
                               const_cast<uint32_t*>(sortedIds), primLo, primHi, left, right, parent);
            YART_HIP(hipGetLastError());
        }
    }
    YART_HIP(hipMemsetAsync(depth, 0xFF, nNodes * 4, nullptr));
    YART_HIP(hipMemsetAsync(depth, 0, 4, nullptr));        // the root, internal node 0
    YART_HIP(hipMemsetAsync(dAny, 0, 256, nullptr));
    // levels, eight launches between two looks at the flags
    uint32_t levels = 0;
    for (bool more = true; more;) {
        if (levels > 4096) { err = "device builder: hierarchy deeper than 4096 levels"; return YART_ERR_TREE_TOO_DEEP; }
        for (uint32_t k = 0; k < 8; ++k)
            hipLaunchKernelGGL(assignLevel, dim3(blocks(nInternal)), dim3(256), 0, nullptr, left, right, (uint32_t)nInternal, depth, levels + k, dAny + k);
        YART_HIP(hipGetLastError());
        uint32_t any[8];
        YART_HIP(hipMemcpy(any, dAny, sizeof any, hipMemcpyDeviceToHost));
        YART_HIP(hipMemsetAsync(dAny, 0, 256, nullptr));
        uint32_t k = 0;
        while (k < 8 && any[k]) ++k;
        levels += k;
        more = k == 8;
    }

    // boxes + collapse plan, deepest level first
    float4 *nodeLo, *nodeHi;
    Plan* plan;
    YART_HIP(S.alloc(&nodeLo, nNodes));
    YART_HIP(S.alloc(&nodeHi, nNodes));
    YART_HIP(S.alloc(&plan, nNodes));
    hipLaunchKernelGGL(leafNodes, dim3(blocks(n)), dim3(256), 0, nullptr, primLo, primHi, sortedIds, (uint32_t)n, nodeLo, nodeHi, plan);
    YART_HIP(hipGetLastError());
    for (uint32_t lv = levels; lv-- > 0;)
        hipLaunchKernelGGL(fitLevel, dim3(blocks(nInternal)), dim3(256), 0, nullptr, left, right, (uint32_t)nInternal, depth, lv, nodeLo, nodeHi, plan);
    YART_HIP(hipGetLastError());
    {
        float4 rl, rh;
        YART_HIP(hipMemcpy(&rl, nodeLo, sizeof rl, hipMemcpyDeviceToHost));
        YART_HIP(hipMemcpy(&rh, nodeHi, sizeof rh, hipMemcpyDeviceToHost));
        st->rootLo[0] = rl.x; st->rootLo[1] = rl.y; st->rootLo[2] = rl.z;
        st->rootHi[0] = rh.x; st->rootHi[1] = rh.y; st->rootHi[2] = rh.z;
    }

    // wide nodes, top-down
    WideNode* wide;
    QueueEntry *queueA, *queueB;
    uint32_t* counters;
    YART_HIP(S.alloc(&wide, nInternal));                  // a wide node swallows at least one binary node
    YART_HIP(S.alloc(&queueA, nInternal));
    YART_HIP(S.alloc(&queueB, nInternal));
    YART_HIP(S.alloc(&counters, 64));
    st->wide = wide;
    const QueueEntry root{ 0u, 0u };
    YART_HIP(hipMemcpyAsync(queueA, &root, sizeof root, hipMemcpyHostToDevice, nullptr));
    const uint32_t initCounters[2] = { 1u, 0u };
    YART_HIP(hipMemcpyAsync(counters, initCounters, sizeof initCounters, hipMemcpyHostToDevice, nullptr));
    uint32_t nQueue = 1;
    st->wideCount = 1;
    while (nQueue > 0) {
        ++st->wideDepth;
        if (3 * st->wideDepth + 1 > kWideStack) { err = "device builder: tree deeper than the traversal stack allows"; return YART_ERR_TREE_TOO_DEEP; }
        hipLaunchKernelGGL(emitLevel, dim3(blocks(nQueue)), dim3(256), 0, nullptr, queueA, nQueue, queueB, counters, left, right, plan,
                           (uint32_t)nInternal, (uint32_t)nTris, sortedIds, nodeLo, nodeHi, wide);
        YART_HIP(hipGetLastError());
        uint32_t c[2];
        YART_HIP(hipMemcpy(c, counters, sizeof c, hipMemcpyDeviceToHost));
        st->wideCount = c[0];
        nQueue = c[1];
        YART_HIP(hipMemsetAsync(counters + 1, 0, 4, nullptr));
        std::swap(queueA, queueB);
    }
    YART_HIP(hipDeviceSynchronize());
    st->msTopology = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st->t0).count();
    return YART_OK;
}

// Phase 2: what needs the reference's leaf order -- the leaf rows -- and the exact-size result arrays.
const float* DeviceWideBuilder::deviceVerts() const { return state_ ? state_->dVerts : nullptr; }
const int32_t* DeviceWideBuilder::deviceTris() const { return state_ ? state_->dTris : nullptr; }
const float* DeviceWideBuilder::deviceSpheres() const { return state_ ? state_->dSpheres : nullptr; }

int DeviceWideBuilder::finish(const void* dTriAccel, const uint32_t* slotOfInput, bool slotsOnDevice, DeviceWide& out, std::string& err)
{
    out = DeviceWide{};
    State* st = state_;
    if (!st || st->n < 2) { err = "device builder: finish() without a topology"; return YART_ERR_INVALID_ARGUMENT; }
    const auto tFinish0 = std::chrono::steady_clock::now();
    const uint32_t* dSlot = slotOfInput;
    if (!slotsOnDevice) {
        uint32_t* up;
        YART_HIP(st->S.alloc(&up, st->nTris));
        if (st->nTris) YART_HIP(hipMemcpyAsync(up, slotOfInput, st->nTris * 4, hipMemcpyHostToDevice, nullptr));
        dSlot = up;
    }
    // pack: block sizes, their running sum, the place of every node and leaf row, the packed units
    const uint32_t nNodes = static_cast<uint32_t>(st->wideCount), nPrims = static_cast<uint32_t>(st->n);
    uint32_t *sizesN, *sumN, *firstPos, *rowSize, *rowSum, *block, *unitOf, *leafUnit;
    YART_HIP(st->S.alloc(&sizesN, nNodes));
    YART_HIP(st->S.alloc(&sumN, nNodes));
    YART_HIP(st->S.alloc(&firstPos, nNodes));
    YART_HIP(st->S.alloc(&rowSize, nPrims));
    YART_HIP(st->S.alloc(&rowSum, nPrims));
    YART_HIP(st->S.alloc(&block, nNodes));
    YART_HIP(st->S.alloc(&unitOf, nNodes));
    YART_HIP(st->S.alloc(&leafUnit, nPrims));
    hipLaunchKernelGGL(packSizes, dim3(blocks(nNodes)), dim3(256), 0, nullptr, st->wide, nNodes, sizesN, firstPos);
    hipLaunchKernelGGL(rowUnits, dim3(blocks(nPrims)), dim3(256), 0, nullptr, st->sortedIds, nPrims, (uint32_t)st->nTris, rowSize);
    YART_HIP(hipGetLastError());
    size_t scanBytes = 0, scanBytes2 = 0;
    YART_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, sizesN, sumN, (int)nNodes, nullptr));
    YART_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes2, rowSize, rowSum, (int)nPrims, nullptr));
    scanBytes = std::max(scanBytes, scanBytes2);
    char* scanTemp;
    YART_HIP(st->S.alloc(&scanTemp, scanBytes));
    YART_HIP(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, sizesN, sumN, (int)nNodes, nullptr));
    YART_HIP(hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, rowSize, rowSum, (int)nPrims, nullptr));
    uint32_t last[4];
    YART_HIP(hipMemcpy(&last[0], sumN + (nNodes - 1), 4, hipMemcpyDeviceToHost));
    YART_HIP(hipMemcpy(&last[1], sizesN + (nNodes - 1), 4, hipMemcpyDeviceToHost));
    YART_HIP(hipMemcpy(&last[2], rowSum + (nPrims - 1), 4, hipMemcpyDeviceToHost));
    YART_HIP(hipMemcpy(&last[3], rowSize + (nPrims - 1), 4, hipMemcpyDeviceToHost));
    const size_t rowsBase = size_t(4) + last[0] + last[1];
    const size_t numUnits = rowsBase + last[2] + last[3];
    if (numUnits > kPackMaxUnits) { err = "device builder: tree too large for the packed format"; return YART_ERR_INVALID_ARGUMENT; }
    void* units = nullptr;
    hipError_t e = hipMalloc(&units, numUnits * 16);
    if (e == hipSuccess) e = hipMemsetAsync(units, 0, numUnits * 16, nullptr);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(packBlocks, dim3(blocks(nNodes)), dim3(256), 0, nullptr, sizesN, sumN, firstPos, rowSum, nNodes, (uint32_t)rowsBase, block);
        hipLaunchKernelGGL(packAssign, dim3(blocks(nNodes)), dim3(256), 0, nullptr, st->wide, nNodes, block, unitOf, leafUnit);
        hipLaunchKernelGGL(packNodes, dim3(blocks(nNodes)), dim3(256), 0, nullptr, st->wide, nNodes, block, unitOf, static_cast<uint4*>(units));
        hipLaunchKernelGGL(leafRows, dim3(blocks(st->n)), dim3(256), 0, nullptr, st->sortedIds, (uint32_t)st->n, (uint32_t)st->nTris, dSlot,
                           static_cast<const float4*>(dTriAccel), reinterpret_cast<const float4*>(st->dSpheres), st->dVerts, st->dTris, st->cover,
                           leafUnit, static_cast<float4*>(units));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(units);
        err = std::string("device builder: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? YART_ERR_OUT_OF_MEMORY : YART_ERR_HIP;
    }
    out.units = units; out.numUnits = numUnits;
    out.numNodes = st->wideCount; out.numPrims = st->n; out.depth = st->wideDepth;
    std::memcpy(out.rootLo, st->rootLo, 12);
    std::memcpy(out.rootHi, st->rootHi, 12);
    out.padMax = st->padMax;
    out.msBuild = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st->t0).count();
    out.msTopology = st->msTopology;
    out.msFinish = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tFinish0).count();
    delete state_;
    state_ = nullptr;
    return YART_OK;
}

} // namespace yart
