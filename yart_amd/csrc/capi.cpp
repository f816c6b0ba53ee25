// capi.cpp -- the C-ABI of include/yart_amd.h: accel lifetime, launches, host-buffer
// convenience calls.  All device work is in traverse.hip / sort.hip; there is no CPU
// intersection path in this library -- without a HIP device every intersect call fails.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/yart_amd.h"
#include "bvh_build.h"
#include "accel.h"
#include "traverse.h"
#include "wide_build.h"
#include "wide_build_dev.h"
#include "ref_build_dev.h"
#include "obj_ingest.h"
#include "task_pool.h"

namespace yart {
void toneMapBgr(const float* rgb, size_t nPixels, uint8_t* bgr);
bool writeBmp(const char* path, const float* rgb, int32_t width, int32_t height, std::string& err);
}

namespace {

thread_local std::string g_lastError;

int fail(int status, const std::string& msg)
{
    g_lastError = msg;
    return status;
}

int failHip(hipError_t e, const char* what)
{
    g_lastError = std::string(what) + ": " + hipGetErrorString(e);
    return YART_ERR_HIP;
}

} // namespace

namespace yart {
int failStatus(int status, const std::string& msg) { return fail(status, msg); }
}

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int upload(void** dst, const void* src, size_t bytes, size_t& total)
{
    *dst = nullptr;
    if (bytes == 0) return YART_OK;
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return failHip(e, "hipMemcpy H2D");
    total += bytes;
    return YART_OK;
}

// Brackets one launch with events when profiling is on.
struct ScopedSpan {
    YartAccel* a;
    hipStream_t stream;
    YartAccel::Span span;
    bool on;
    // timed = false: a call that must not allocate (the no-alloc entry point: events and the span list are allocations)
    ScopedSpan(YartAccel* accel, int kind, hipStream_t s, bool timed = true) : a(accel), stream(s), on(accel->profile && timed)
    {
        span.kind = kind;
        if (!on) return;
        if (hipEventCreate(&span.start) != hipSuccess) { on = false; return; }
        if (hipEventCreate(&span.stop) != hipSuccess) { (void)hipEventDestroy(span.start); on = false; return; }
        if (hipEventRecord(span.start, stream) != hipSuccess) { drop(); on = false; }
    }
    ~ScopedSpan()
    {
        if (!on) return;
        if (hipEventRecord(span.stop, stream) == hipSuccess) {
            std::lock_guard<std::mutex> lock(a->profileMutex);
            a->spans.push_back(span);
        }
        else drop();
    }
    void drop()
    {
        (void)hipEventDestroy(span.start);
        (void)hipEventDestroy(span.stop);
    }
};

// The calling stream's context (work-counter slots + sort workspace), created at the stream's
// first call.  Contexts live as long as the accel.
YartAccel::StreamCtx* streamContext(YartAccel* a, hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(a->ctxMutex);
    auto it = a->ctx.find(static_cast<void*>(stream));
    if (it != a->ctx.end()) return it->second;
    auto c = std::make_unique<YartAccel::StreamCtx>();
    hipError_t e = hipMalloc(&c->counters, yart::kCounterBlockBytes);
    if (e != hipSuccess) { fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(work counters): ") + hipGetErrorString(e)); return nullptr; }
    YartAccel::StreamCtx* raw = c.release();
    a->ctx.emplace(static_cast<void*>(stream), raw);
    return raw;
}

} // namespace

namespace yart {

int runIntersect(YartAccel* a, const Launch& L, const CallScratch* given)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "null accel");
    if (L.n == 0) return YART_OK;
    if (!L.rays || (!L.any && !L.hits) || (L.any && !L.occluded) || (L.count && !L.counts))
        return fail(YART_ERR_INVALID_ARGUMENT, "null ray / result pointer");
    if (L.n > 0x7FFFFFFFu) return fail(YART_ERR_INVALID_ARGUMENT, "batch larger than 2^31-1 rays");
    if ((reinterpret_cast<uintptr_t>(L.rays) & 15u) || (!L.any && (reinterpret_cast<uintptr_t>(L.hits) & 15u)))
        return fail(YART_ERR_INVALID_ARGUMENT, "ray / hit arrays must be 16-byte aligned");
    DeviceGuard guard(a->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");

    const bool prune = !(L.flags & YART_REFERENCE_RULE);
    const bool persistent = !(L.flags & YART_STATIC_LAUNCH);
    const uint32_t n = static_cast<uint32_t>(L.n);
    const bool sorted = (L.flags & YART_SORT_RAYS) && n >= a->sortMinRays;

    // scratch of this call: the caller's (nothing is allocated) or the stream's own context
    CallScratch sc;
    std::unique_lock<std::mutex> submit;
    if (given) {
        sc = *given;
        if (sorted && yart::sortWorkspaceBytes(n) > sc.sortWsBytes)
            return fail(YART_ERR_INVALID_ARGUMENT, "batch too large for the given sort workspace");
    } else {
        YartAccel::StreamCtx* c = streamContext(a, L.stream);
        if (!c) return YART_ERR_OUT_OF_MEMORY;
        submit = std::unique_lock<std::mutex>(c->submit);
        if (sorted) {
            const size_t need = yart::sortWorkspaceBytes(n);
            if (need > c->sortWsBytes) {
                (void)hipFree(c->sortWs);          // synchronises: no earlier launch still reads it
                c->sortWs = nullptr;
                c->sortWsBytes = 0;
                const hipError_t e = hipMalloc(&c->sortWs, need);
                if (e != hipSuccess) return fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(sort workspace): ") + hipGetErrorString(e));
                c->sortWsBytes = need;
            }
        }
        sc.counters = c->counters;
        sc.sortWs = c->sortWs;
        sc.sortWsBytes = c->sortWsBytes;
    }
    // one 256-byte slot per launch of this call; each launcher zeroes its own on the stream
    uint32_t* const slot0 = static_cast<uint32_t*>(sc.counters);
    uint32_t* const slot1 = slot0 + 64;
    uint32_t* const slot2 = slot0 + 128;

    const uint32_t* perm = nullptr;
    if (sorted) {
        hipError_t e;
        {
            ScopedSpan span(a, 0, L.stream, !given);
            e = yart::sortRays(reinterpret_cast<const float4*>(L.rays), n, a->sceneLo, a->sceneHi, a->sortOctShift, a->sortGridBits,
                               &perm, sc.sortWs, sc.sortWsBytes, L.stream);
        }
        if (e != hipSuccess) return failHip(e, "sortRays");
    }

    yart::TraceParams P;
    std::memset(&P, 0, sizeof P);
    P.rays = reinterpret_cast<const float4*>(L.rays);
    P.perm = perm;
    P.n = n;
    P.hits = reinterpret_cast<float4*>(L.hits);
    P.occluded = L.occluded;
    P.counts = L.counts;
    P.refillMin = a->refillMin;
    P.nodeMin = a->nodeMin;
    P.cover = a->cover;
    std::memcpy(P.sceneLo, a->sceneLo, 12);
    std::memcpy(P.sceneHi, a->sceneHi, 12);
    hipError_t e = hipSuccess;
    const bool useWide = a->hasWide && !(L.flags & (YART_REFERENCE_RULE | YART_REFERENCE_TREE));
    P.deferFlag = nullptr;
    P.onlyDeferred = 0;
    bool fixupOnly = false;
    if (useWide) {
        yart::WideParams W;
        std::memset(&W, 0, sizeof W);
        W.units = static_cast<const float4*>(a->dWideUnits);
        W.triaccel = static_cast<const float4*>(a->dTriAccel);
        W.rays = P.rays; W.perm = perm; W.n = n;
        W.hits = P.hits; W.occluded = P.occluded; W.counts = P.counts;
        W.workCounter = slot0;
        W.deferFlag = W.workCounter + 1;
        W.deferList = slot0 + 256;                  // behind the four counter slots (accel.h: kCounterBlockBytes)
        W.deferCap = yart::kDeferCap;
        W.refNodes = static_cast<const float4*>(a->dRefNodes);
        W.leafOfSlot = static_cast<const uint32_t*>(a->dLeafOfSlot);
        W.cover = a->cover;
        W.cullReach = a->cullReach;
        W.cullMargin = 0x1p-21f * a->cullReach;
        W.refillMin = a->wideRefillMin; W.nodeMin = a->wideNodeMin; W.sphereMin = a->wideSphereMin; W.chunk = a->wideChunk; W.maxBlocksPerCU = a->wideBlocksPerCU; W.laneCap = a->wideLaneCap; W.hasSpheres = a->numSpheres > 0 ? 1u : 0u;
        if (!given) {
            // experiment flags, read per call
            static const auto envU = [](const char* name, uint32_t dflt) { const char* v = std::getenv(name); return v ? static_cast<uint32_t>(std::atoi(v)) : dflt; };
            W.poolSlots = envU("YART_WIDE_POOL", 0);
            if (W.poolSlots) {
                W.poolWindow = envU("YART_WIDE_POOL_WINDOW", 10);
                W.poolBlock = envU("YART_WIDE_POOL_BLOCK", 512);
                W.poolKeep = envU("YART_WIDE_POOL_KEEP", 48);
                W.poolTriMin = envU("YART_WIDE_POOL_TRI", 32);
                W.poolSphMin = envU("YART_WIDE_POOL_SPH", 24);
                if (W.poolBlock != 256 && W.poolBlock != 512) W.poolBlock = 512;
                if (W.poolSlots > 32) W.poolSlots = 32;
                if (W.poolSlots <= W.poolBlock / 64) W.poolSlots = W.poolBlock / 64 + 1;      // a wave must always find a slot
                if (W.poolWindow < 2) W.poolWindow = 2;
                if (W.poolWindow > 32) W.poolWindow = 32;
            }
        }
        {
            ScopedSpan span(a, 2, L.stream, !given);
            e = yart::launchTraceWide(W, L.any, L.count, persistent, a->numCUs, L.stream);
        }
        // rays the wide kernel does not take (irregular, outside the cover, too nearly axis-parallel for the box arithmetic) are answered on
        // the reference tree
        P.deferFlag = W.deferFlag;
        P.deferList = W.deferList;
        P.deferCap = W.deferCap;
        P.onlyDeferred = 1;
        fixupOnly = true;
    }
    const bool skipRefTree = fixupOnly && (L.flags & YART_RAYS_REGULAR);
    if (e == hipSuccess && !skipRefTree && a->numSpheres > 0) {
        // Scene::intersect / intersectShadow visit the spheres before the triangle tree
        // (include/scene.h:142-146, :176-182); the triangle pass starts from the sphere hit.
        P.pairs = static_cast<const float4*>(a->dSphPairs);
        P.prims = static_cast<const float4*>(a->dSphRecs);
        P.primIndex = static_cast<const uint32_t*>(a->dSphIndex);
        P.workCounter = slot1;
        P.chained = 0;
        ScopedSpan span(a, 1, L.stream, !given);
        e = yart::launchTrace(P, 1, L.any, prune, L.count, persistent, a->numCUs, L.stream);
    }
    if (e == hipSuccess && !skipRefTree) {
        P.pairs = static_cast<const float4*>(a->dTriPairs);
        P.prims = static_cast<const float4*>(a->dTriAccel);
        P.primIndex = nullptr;
        P.workCounter = slot2;
        P.chained = a->numSpheres > 0 ? 1u : 0u;
        ScopedSpan span(a, fixupOnly ? 1 : 2, L.stream, !given);
        e = yart::launchTrace(P, 0, L.any, prune, L.count, persistent, a->numCUs, L.stream);
    }
    if (a->profile && !given) {
        std::lock_guard<std::mutex> lock(a->profileMutex);
        a->profiledCalls++;
    }
    if (e != hipSuccess) return failHip(e, "traversal launch");
    return YART_OK;
}

} // namespace yart

namespace {

using yart::Launch;

int run(YartAccel* a, const Launch& L) { return yart::runIntersect(a, L, nullptr); }

} // namespace

extern "C" {

int yart_accel_create(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                      const float* spheres, size_t n_spheres, int device, YartAccel** out)
{
    if (!out) return fail(YART_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
        return fail(YART_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= nDev) return fail(YART_ERR_INVALID_ARGUMENT, "device ordinal out of range");

    YartAccel* a = new (std::nothrow) YartAccel();
    if (!a) return fail(YART_ERR_OUT_OF_MEMORY, "host allocation failed");
    a->device = device;
    if (const char* v = std::getenv("YART_REFILL_MIN")) a->refillMin = static_cast<uint32_t>(std::atoi(v));
    if (const char* v = std::getenv("YART_NODE_MIN")) a->nodeMin = static_cast<uint32_t>(std::atoi(v));
    if (a->refillMin < 1) a->refillMin = 1;
    if (a->refillMin > 64) a->refillMin = 64;
    if (a->nodeMin < 1) a->nodeMin = 1;
    if (a->nodeMin > 64) a->nodeMin = 64;
    if (const char* v = std::getenv("YART_WIDE_REFILL_MIN")) a->wideRefillMin = static_cast<uint32_t>(std::atoi(v));
    if (a->wideRefillMin < 1) a->wideRefillMin = 1;
    if (a->wideRefillMin > 64) a->wideRefillMin = 64;
    if (const char* v = std::getenv("YART_WIDE_NODE_MIN")) a->wideNodeMin = static_cast<uint32_t>(std::atoi(v));
    if (a->wideNodeMin < 1) a->wideNodeMin = 1;
    if (a->wideNodeMin > 64) a->wideNodeMin = 64;
    if (const char* v = std::getenv("YART_WIDE_SPHERE_MIN")) a->wideSphereMin = static_cast<uint32_t>(std::atoi(v));
    if (a->wideSphereMin < 1) a->wideSphereMin = 1;
    if (a->wideSphereMin > 64) a->wideSphereMin = 64;
    if (const char* v = std::getenv("YART_SORT_GRID_BITS")) a->sortGridBits = static_cast<uint32_t>(std::atoi(v));
    if (const char* v = std::getenv("YART_SORT_MIN_RAYS")) a->sortMinRays = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
    if (const char* v = std::getenv("YART_SORT_OCT_SHIFT")) a->sortOctShift = static_cast<uint32_t>(std::atoi(v));
    if (const char* v = std::getenv("YART_WIDE_BLOCKS_PER_CU")) a->wideBlocksPerCU = static_cast<uint32_t>(std::atoi(v));
    if (const char* v = std::getenv("YART_WIDE_LANE_CAP")) a->wideLaneCap = static_cast<uint32_t>(std::atoi(v));
    if (const char* v = std::getenv("YART_WIDE_CHUNK")) a->wideChunk = static_cast<uint32_t>(std::atoi(v));
    a->wideChunk = (a->wideChunk + 63) / 64 * 64;
    if (a->wideChunk < 64) a->wideChunk = 64;
    if (a->wideChunk > 4096) a->wideChunk = 4096;
    std::string err;
    const char* noWide = std::getenv("YART_NO_WIDE");
    const bool wantWide = !(noWide && noWide[0] == '1');
    // The 4-wide tree is built on the GPU (wide_build_dev.hip) for scenes of 2^18 primitives and more -- the host builder
    // (binned SAH on a thread pool) needs 3.8 s for the 11 M primitives of BASELINE configs[3], the device 0.1 s for a
    // tree the traversal kernel walks 1-2 % slower -- and on the host below that, where it takes no time;
    // YART_BUILD_DEVICE=1 / 0 decide for every size.  The builder's first phase needs only the scene and runs in a
    // thread of its own.
    const char* devBuild = std::getenv("YART_BUILD_DEVICE");
    const bool deviceByDefault = n_tris + n_spheres >= (size_t(1) << 18);
    const bool wantDeviceBuild = wantWide && (devBuild ? devBuild[0] == '1' : deviceByDefault) && n_tris + n_spheres >= 2;
    const auto tBuild0 = std::chrono::steady_clock::now();
    if ((n_tris && (!verts || !tris)) || n_tris >= yart::kMaxLeafSlot || (n_spheres && !spheres) || n_spheres >= yart::kMaxLeafSlot) {
        delete a;
        return fail(YART_ERR_INVALID_ARGUMENT, "yart_accel_create: null scene array or too many primitives");
    }
    a->cover = yart::sceneCover(verts, n_verts, spheres, n_spheres);
    const yart::Cover cover = a->cover;
    yart::DeviceWideBuilder devBuilder;
    std::thread devThread;
    int devStatus = YART_ERR_HIP;
    std::string devErr;
    if (wantDeviceBuild) {
        devThread = std::thread([&]() {
            if (hipSetDevice(device) != hipSuccess) { devErr = "hipSetDevice failed"; return; }
            devStatus = devBuilder.topology(verts, n_verts, tris, n_tris, spheres, n_spheres, cover, devErr);
        });
    }
    // ... and so is the reference's own tree (ref_build_dev.hip), bit for bit, once the scene is on the device;
    // YART_BUILD_DEVICE_REF=0 keeps that one on the host (the two builds then overlap as before)
    const char* devRefEnv = std::getenv("YART_BUILD_DEVICE_REF");
    const bool wantDeviceRef = wantDeviceBuild && !(devRefEnv && devRefEnv[0] == '0') && n_tris >= 1024;
    DeviceGuard guard(device);
    if (!guard.ok) {
        if (devThread.joinable()) devThread.join();
        delete a;
        return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    }
    yart::SphereBvh sph;
    yart::DeviceRefTree devRef;
    yart::DeviceSphereTree devSph;
    bool refOnDevice = false, sphOnDevice = false;
    int st = YART_OK;
    if (wantDeviceRef) {
        // the sphere tree of the reference-tree path: on the host while the device sorts, or -- from 1024 spheres --
        // by the same device builder afterwards
        const bool wantDeviceSph = n_spheres >= 1024;
        if (!wantDeviceSph) st = yart::buildSphereBvh(spheres, n_spheres, cover, sph, err);
        devThread.join();
        if (st == YART_OK && devStatus == YART_OK) {
            bool useHost = false;
            st = yart::buildReferenceTreeDevice(devBuilder.deviceVerts(), verts, n_verts, devBuilder.deviceTris(), tris, n_tris, devRef, &useHost, err);
            refOnDevice = st == YART_OK && !useHost;
            if (st == YART_OK && wantDeviceSph) {
                useHost = false;
                st = yart::buildSphereTreeDevice(devBuilder.deviceSpheres(), n_spheres, cover, devSph, &useHost, err);
                sphOnDevice = st == YART_OK && !useHost;
            }
        }
        if (st == YART_OK && !refOnDevice) st = yart::buildTriangleBvh(verts, n_verts, tris, n_tris, a->tri, err);
        if (st == YART_OK && wantDeviceSph && !sphOnDevice) st = yart::buildSphereBvh(spheres, n_spheres, cover, sph, err);
    } else {
        st = yart::buildTriangleBvh(verts, n_verts, tris, n_tris, a->tri, err);
        if (st == YART_OK) st = yart::buildSphereBvh(spheres, n_spheres, cover, sph, err);
        if (devThread.joinable()) devThread.join();
    }
    const double msRefTree = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tBuild0).count();
    const bool useDeviceBuild = wantDeviceBuild && devStatus == YART_OK;
    yart::WideBvh wide;
    if (st == YART_OK && wantWide && !useDeviceBuild) {
        const int maxLeaf = 1;      // measured best on MI355X (tools/exp_wide.py): the box of a single primitive is a
                                    // cheaper rejection than its plane row, and rows of different leaves are not tested
        st = yart::buildWideBvh(verts, n_verts, tris, n_tris, a->tri, spheres, n_spheres, maxLeaf, cover, wide, err);
    }
    if (st != YART_OK) { devRef.release(); devSph.release(); delete a; return fail(st, err); }
    a->numTriPairs = refOnDevice ? devRef.numPairs : a->tri.pairs.size();
    a->numTriangles = n_tris;
    a->numNodes = refOnDevice ? devRef.numNodes : a->tri.nodes.size();
    if (refOnDevice) a->tri.depth = devRef.depth;
    a->numSphPairs = sphOnDevice ? devSph.numPairs : sph.pairs.size();
    a->numSpheres = n_spheres;
    a->sphDepth = sphOnDevice ? devSph.depth : sph.depth;

    const float inf = std::numeric_limits<float>::infinity();
    for (int c = 0; c < 3; ++c) { a->sceneLo[c] = inf; a->sceneHi[c] = -inf; }
    auto growBy = [&](const yart::DevNodeRec& r) {
        for (int c = 0; c < 3; ++c) {
            if (r.lo[c] < a->sceneLo[c]) a->sceneLo[c] = r.lo[c];
            if (r.hi[c] > a->sceneHi[c]) a->sceneHi[c] = r.hi[c];
        }
    };
    if (refOnDevice) {
        yart::DevNodeRec root;
        std::memcpy(root.lo, devRef.rootLo, 12);
        std::memcpy(root.hi, devRef.rootHi, 12);
        growBy(root);
    } else if (n_tris) growBy(a->tri.pairs[0].child[0]);
    if (sphOnDevice) {
        yart::DevNodeRec root;
        std::memcpy(root.lo, devSph.rootLo, 12);
        std::memcpy(root.hi, devSph.rootHi, 12);
        growBy(root);
    } else if (n_spheres) growBy(sph.pairs[0].child[0]);

    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { devRef.release(); devSph.release(); delete a; return fail(YART_ERR_NO_DEVICE, "hipGetDeviceProperties failed"); }
    a->numCUs = prop.multiProcessorCount;

    if (refOnDevice) {
        // the builder's arrays are the accel's: nothing is uploaded, nothing exists on the host
        a->dTriPairs = devRef.pairs;
        a->dTriAccel = devRef.triaccel;
        a->dRefNodes = devRef.nodes;
        a->dSlotOfInput = devRef.slotOfInput;
        a->deviceBytes += devRef.numPairs * sizeof(yart::DevNodePair) + n_tris * (sizeof(YartTriAccel) + 4) + devRef.numNodes * sizeof(YartBvhNode);
        a->refBuiltOnDevice = true;
        a->msRefDeviceBuild = devRef.ms;
        a->refHostMedianSplits = devRef.medianSplitsOnHost;
        a->refHostProjections = devRef.projectionsOnHost;
        if (std::getenv("YART_BUILD_TIMING"))
            fprintf(stderr, "device reference-tree build: %.1f ms (%zu nodes, depth %d; on the host: %zu median splits, %zu degenerate projections)\n",
                    devRef.ms, devRef.numNodes, devRef.depth, devRef.medianSplitsOnHost, devRef.projectionsOnHost);
    } else {
        st = upload(&a->dTriPairs, a->tri.pairs.data(), a->tri.pairs.size() * sizeof(yart::DevNodePair), a->deviceBytes);
        if (st == YART_OK) st = upload(&a->dTriAccel, a->tri.triaccel.data(), a->tri.triaccel.size() * sizeof(YartTriAccel), a->deviceBytes);
    }
    if (sphOnDevice) {
        a->dSphPairs = devSph.pairs;
        a->dSphRecs = devSph.records;
        a->dSphIndex = devSph.index;
        a->deviceBytes += devSph.numPairs * sizeof(yart::DevNodePair) + n_spheres * 20;
        if (std::getenv("YART_BUILD_TIMING"))
            fprintf(stderr, "device sphere-tree build: %.1f ms (%zu pairs, depth %d)\n", devSph.ms, devSph.numPairs, devSph.depth);
    } else {
        if (st == YART_OK) st = upload(&a->dSphPairs, sph.pairs.data(), sph.pairs.size() * sizeof(yart::DevNodePair), a->deviceBytes);
        if (st == YART_OK) st = upload(&a->dSphRecs, sph.records.data(), sph.records.size() * sizeof(float), a->deviceBytes);
        if (st == YART_OK) st = upload(&a->dSphIndex, sph.index.data(), sph.index.size() * sizeof(uint32_t), a->deviceBytes);
    }
    if (st == YART_OK && useDeviceBuild) {
        // second phase of the device build: leaf rows from the reference's TriAccel records and leaf order
        yart::DeviceWide dw;
        st = devBuilder.finish(a->dTriAccel, refOnDevice ? static_cast<const uint32_t*>(a->dSlotOfInput) : a->tri.slotOfInput.data(), refOnDevice, dw, err);
        if (st != YART_OK) st = fail(st, err);
        else {
            a->dWideUnits = dw.units; a->wideUnits = dw.numUnits;
            std::memcpy(a->gridLo, dw.rootLo, 12);
            std::memcpy(a->gridHi, dw.rootHi, 12);
            a->cover.padMax = dw.padMax;
            a->wideNodes = dw.numNodes; a->widePrims = dw.numPrims; a->wideDepth = dw.depth;
            a->deviceBytes += dw.numUnits * 16;
            a->hasWide = true;
            a->wideBuiltOnDevice = true;
            a->msWideDeviceBuild = dw.msTopology + dw.msFinish;
            if (std::getenv("YART_BUILD_TIMING"))
                fprintf(stderr, "device wide build: %.1f ms (scene upload, sort, topology, fit, plan, %zu wide nodes, depth %d) + %.1f ms (leaf rows); "
                                "reference tree on the host meanwhile: %.1f ms\n", dw.msTopology, dw.numNodes, dw.depth, dw.msFinish, msRefTree);
        }
    } else if (st == YART_OK && wantWide) {
        yart::PackedWide packed;
        if (!yart::packWide(wide, packed)) { err = "wide BVH does not fit the packed format"; st = fail(YART_ERR_INVALID_ARGUMENT, err); }
        else st = upload(&a->dWideUnits, packed.words.data(), packed.numUnits * 16, a->deviceBytes);
        a->wideUnits = packed.numUnits;
        std::memcpy(a->gridLo, wide.rootLo, 12);
        std::memcpy(a->gridHi, wide.rootHi, 12);
        a->cover.padMax = wide.padMax;
        a->hasWide = (st == YART_OK);
        a->wideNodes = wide.nodes.size();
        a->widePrims = wide.rowsA.size() / 4;
        a->wideDepth = wide.depth;
    }
    if (st == YART_OK && !a->hasWide) a->cover.padMax = yart::trianglePadMax(verts, tris, n_tris, a->cover);
    if (st == YART_OK && a->hasWide) {
        // |A| + 255 |B| of the wide kernel's box test <= cullReach x the largest |1 / d| of a covered ray (traverse_wide.hip)
        float reach = 0.0f;
        for (int c = 0; c < 3; ++c) {
            const float lo = a->cover.valid ? std::min(a->cover.lo[c], a->gridLo[c]) : a->gridLo[c];
            const float hi = a->cover.valid ? std::max(a->cover.hi[c], a->gridHi[c]) : a->gridHi[c];
            reach = std::max(reach, (hi - lo) + 1.25f * (a->gridHi[c] - a->gridLo[c]));
        }
        a->cullReach = reach == reach ? reach : std::numeric_limits<float>::infinity();
    }
    if (st == YART_OK && a->hasWide && n_tris) {
        // the reference's flattened nodes on the device (a device-built accel has them already) and, per triangle slot,
        // its leaf there (traverse_wide.hip: near-edge hits, ties)
        if (!a->dRefNodes) st = upload(&a->dRefNodes, a->tri.nodes.data(), a->tri.nodes.size() * sizeof(YartBvhNode), a->deviceBytes);
    }
    if (st == YART_OK && a->hasWide && n_tris) {
        hipError_t e = hipMalloc(&a->dLeafOfSlot, n_tris * 4);
        if (e == hipSuccess) e = yart::launchLeafOfSlot(a->dRefNodes, a->numNodes, static_cast<uint32_t*>(a->dLeafOfSlot), nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) st = failHip(e, "leaf table of the reference tree");
        else a->deviceBytes += n_tris * 4;
    }
    // this device's copy of the static scratch of yart_intersect_closest_noalloc_dev (looked up once per
    // device, with that device current)
    // (a device the table does not reach, or a failed lookup, leaves it null: only the no-alloc call then fails)
    if (st == YART_OK && yart::noallocScratchOfDevice(device, &a->noallocScratch, &a->noallocBytes) != YART_OK) {
        a->noallocScratch = nullptr;
        a->noallocBytes = 0;
    }
    if (st != YART_OK) { yart_accel_destroy(a); return st; }
    // the device copy is authoritative from here on; the pair array is not needed on the host
    a->tri.pairs.clear();
    a->tri.pairs.shrink_to_fit();
    *out = a;
    return YART_OK;
}

void yart_accel_destroy(YartAccel* a)
{
    if (!a) return;
    {
        DeviceGuard guard(a->device);
        (void)hipFree(a->dTriPairs);
        (void)hipFree(a->dTriAccel);
        (void)hipFree(a->dRefNodes);
        (void)hipFree(a->dSlotOfInput);
        (void)hipFree(a->dSphPairs);
        (void)hipFree(a->dSphRecs);
        (void)hipFree(a->dSphIndex);
        for (auto& kv : a->ctx) {
            (void)hipFree(kv.second->counters);
            (void)hipFree(kv.second->sortWs);
            delete kv.second;
        }
        (void)hipFree(a->dWideUnits);
        (void)hipFree(a->dLeafOfSlot);
        (void)hipFree(a->dSurfRows);
        (void)hipFree(a->dSurfSpheres);
        for (auto& c : a->stagingFree) {
            (void)hipHostFree(c.host);
            (void)hipFree(c.big);
            (void)hipStreamDestroy(c.stream);
        }
    }
    delete a;
}

int yart_intersect_closest_dev(YartAccel* accel, const YartRay* d_rays, size_t n, YartHit* d_hits,
                               unsigned flags, void* stream)
{
    return run(accel, Launch{ false, false, d_rays, n, d_hits, nullptr, nullptr, flags, static_cast<hipStream_t>(stream) });
}

int yart_intersect_any_dev(YartAccel* accel, const YartRay* d_rays, size_t n, uint8_t* d_occluded,
                           unsigned flags, void* stream)
{
    return run(accel, Launch{ true, false, d_rays, n, nullptr, d_occluded, nullptr, flags, static_cast<hipStream_t>(stream) });
}

int yart_intersect_closest_counted_dev(YartAccel* accel, const YartRay* d_rays, size_t n, YartHit* d_hits,
                                       uint32_t* d_counts, unsigned flags, void* stream)
{
    return run(accel, Launch{ false, true, d_rays, n, d_hits, nullptr, d_counts, flags, static_cast<hipStream_t>(stream) });
}

int yart_intersect_any_counted_dev(YartAccel* accel, const YartRay* d_rays, size_t n, uint8_t* d_occluded,
                                   uint32_t* d_counts, unsigned flags, void* stream)
{
    return run(accel, Launch{ true, true, d_rays, n, nullptr, d_occluded, d_counts, flags, static_cast<hipStream_t>(stream) });
}

// Host-buffer calls take a pooled context (see YartAccel::Staging): batches of up to kStagingRays
// rays go through its pinned block, larger ones through its device block.
constexpr size_t kStagingRays = 4096;
constexpr size_t kStagingBytes = kStagingRays * (sizeof(YartRay) + sizeof(YartHit));

static int pooledHostCall(YartAccel* accel, const YartRay* rays, size_t n, YartHit* hits, uint8_t* occluded, unsigned flags)
{
    YartAccel::Staging c;
    {
        std::lock_guard<std::mutex> lock(accel->stagingMutex);
        if (!accel->stagingFree.empty()) { c = accel->stagingFree.back(); accel->stagingFree.pop_back(); }
    }
    if (!c.host) {
        hipError_t e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
        if (e != hipSuccess) return failHip(e, "hipStreamCreate");
        e = hipHostMalloc(&c.host, kStagingBytes, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&c.dev, c.host, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(c.host);
            (void)hipStreamDestroy(c.stream);
            return fail(YART_ERR_OUT_OF_MEMORY, std::string("pinned staging block: ") + hipGetErrorString(e));
        }
    }
    int st = YART_OK;
    const size_t outBytes = hits ? n * sizeof(YartHit) : n;
    if (n <= kStagingRays) {
        // the kernels read and write the pinned block in place: launches + one synchronise
        std::memcpy(c.host, rays, n * sizeof(YartRay));
        char* dOut = static_cast<char*>(c.dev) + kStagingRays * sizeof(YartRay);
        const char* hOut = static_cast<char*>(c.host) + kStagingRays * sizeof(YartRay);
        // a handful of rays cannot fill persistent workgroups: plain grid.  The default kernel alone first (no fix-up
        // launches: one-ray calls are launch bound); the block is host memory, so whether it left a ray unresolved
        // (irregular, outside the cover, too nearly axis-parallel for the box arithmetic) shows in the results, and only then is the
        // batch traced again on the reference's tree.
        const bool wideOnly = !(flags & (YART_REFERENCE_RULE | YART_REFERENCE_TREE)) && accel->hasWide;
        for (int pass = 0; pass < 2 && st == YART_OK; ++pass) {
            const unsigned f = flags | YART_STATIC_LAUNCH | (wideOnly ? (pass == 0 ? YART_RAYS_REGULAR : YART_REFERENCE_TREE) : 0u);
            st = hits ? yart_intersect_closest_dev(accel, static_cast<YartRay*>(c.dev), n, reinterpret_cast<YartHit*>(dOut), f, c.stream)
                      : yart_intersect_any_dev(accel, static_cast<YartRay*>(c.dev), n, reinterpret_cast<uint8_t*>(dOut), f, c.stream);
            if (st == YART_OK) {
                const hipError_t e = hipStreamSynchronize(c.stream);
                if (e != hipSuccess) st = failHip(e, "hipStreamSynchronize");
            }
            if (!wideOnly || st != YART_OK) break;
            bool unresolved = false;
            if (hits) { const YartHit* h = reinterpret_cast<const YartHit*>(hOut); for (size_t i = 0; i < n && !unresolved; ++i) unresolved = h[i].prim == YART_PRIM_UNRESOLVED; }
            else { const uint8_t* o = reinterpret_cast<const uint8_t*>(hOut); for (size_t i = 0; i < n && !unresolved; ++i) unresolved = o[i] == YART_OCC_UNRESOLVED; }
            if (!unresolved) break;
        }
        if (st == YART_OK) std::memcpy(hits ? static_cast<void*>(hits) : static_cast<void*>(occluded), hOut, outBytes);
    } else {
        // larger batches: a device block kept with the context (grown on demand), copies on its stream
        const size_t rayBytes = (n * sizeof(YartRay) + 255) & ~size_t(255);
        const size_t need = rayBytes + outBytes;
        if (need > c.bigBytes) {
            (void)hipFree(c.big);
            c.big = nullptr;
            c.bigBytes = 0;
            const hipError_t e = hipMalloc(&c.big, need);
            if (e != hipSuccess) st = fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
            else c.bigBytes = need;
        }
        if (st == YART_OK) {
            char* dOut = static_cast<char*>(c.big) + rayBytes;
            hipError_t e = hipMemcpyAsync(c.big, rays, n * sizeof(YartRay), hipMemcpyHostToDevice, c.stream);
            if (e != hipSuccess) st = failHip(e, "hipMemcpy H2D");
            if (st == YART_OK)
                st = hits ? yart_intersect_closest_dev(accel, static_cast<YartRay*>(c.big), n, reinterpret_cast<YartHit*>(dOut), flags, c.stream)
                          : yart_intersect_any_dev(accel, static_cast<YartRay*>(c.big), n, reinterpret_cast<uint8_t*>(dOut), flags, c.stream);
            if (st == YART_OK) {
                e = hipMemcpyAsync(hits ? static_cast<void*>(hits) : static_cast<void*>(occluded), dOut, outBytes, hipMemcpyDeviceToHost, c.stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
                if (e != hipSuccess) st = failHip(e, "hipMemcpy D2H");
            }
        }
    }
    std::lock_guard<std::mutex> lock(accel->stagingMutex);
    accel->stagingFree.push_back(c);
    return st;
}

static int hostCall(YartAccel* accel, const YartRay* rays, size_t n, YartHit* hits, uint8_t* occluded, unsigned flags)
{
    if (!accel) return fail(YART_ERR_INVALID_ARGUMENT, "null accel");
    if (n == 0) return YART_OK;
    if (!rays || (!hits && !occluded)) return fail(YART_ERR_INVALID_ARGUMENT, "null ray / result pointer");
    DeviceGuard guard(accel->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    return pooledHostCall(accel, rays, n, hits, occluded, flags);
}

int yart_intersect_closest(YartAccel* accel, const YartRay* rays, size_t n, YartHit* hits, unsigned flags)
{
    if (!hits && n) return fail(YART_ERR_INVALID_ARGUMENT, "hits is null");
    return hostCall(accel, rays, n, hits, nullptr, flags);
}

int yart_intersect_any(YartAccel* accel, const YartRay* rays, size_t n, uint8_t* occluded, unsigned flags)
{
    if (!occluded && n) return fail(YART_ERR_INVALID_ARGUMENT, "occluded is null");
    return hostCall(accel, rays, n, nullptr, occluded, flags);
}

struct YartQueue {
    YartAccel* accel = nullptr;
    size_t capacity = 0;
    hipStream_t stream = nullptr;
    void* host = nullptr;       // one pinned block: rays | hits | surfaces | occluded
    void* dev = nullptr;        // the same layout on the device
    size_t offHits = 0, offSurf = 0, offOcc = 0, bytes = 0;
};

int yart_queue_create(YartAccel* accel, size_t capacity, YartQueue** out)
{
    if (!accel || !out || capacity == 0 || capacity > 0x7FFFFFFFu) return fail(YART_ERR_INVALID_ARGUMENT, "yart_queue_create: bad argument");
    *out = nullptr;
    DeviceGuard guard(accel->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    auto q = std::make_unique<YartQueue>();
    q->accel = accel;
    q->capacity = capacity;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    q->offHits = up(capacity * sizeof(YartRay));
    q->offSurf = q->offHits + up(capacity * sizeof(YartHit));
    q->offOcc = q->offSurf + up(capacity * sizeof(YartSurface));
    q->bytes = q->offOcc + up(capacity);
    hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(&q->host, q->bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&q->dev, q->bytes);
    if (e != hipSuccess) {
        (void)hipHostFree(q->host); (void)hipFree(q->dev);
        if (q->stream) (void)hipStreamDestroy(q->stream);
        return fail(YART_ERR_OUT_OF_MEMORY, std::string("yart_queue_create: ") + hipGetErrorString(e));
    }
    *out = q.release();
    return YART_OK;
}

void yart_queue_destroy(YartQueue* q)
{
    if (!q) return;
    {
        DeviceGuard guard(q->accel->device);
        (void)hipStreamSynchronize(q->stream);
        (void)hipHostFree(q->host);
        (void)hipFree(q->dev);
        (void)hipStreamDestroy(q->stream);
    }
    delete q;
}

YartRay* yart_queue_rays(YartQueue* q) { return q ? static_cast<YartRay*>(q->host) : nullptr; }
const YartHit* yart_queue_hits(const YartQueue* q) { return q ? reinterpret_cast<const YartHit*>(static_cast<const char*>(q->host) + q->offHits) : nullptr; }
const YartSurface* yart_queue_surfaces(const YartQueue* q) { return q ? reinterpret_cast<const YartSurface*>(static_cast<const char*>(q->host) + q->offSurf) : nullptr; }
const uint8_t* yart_queue_occluded(const YartQueue* q) { return q ? reinterpret_cast<const uint8_t*>(static_cast<const char*>(q->host) + q->offOcc) : nullptr; }

static int queueRun(YartQueue* q, size_t n, unsigned flags, bool any, bool surface)
{
    if (!q) return fail(YART_ERR_INVALID_ARGUMENT, "null queue");
    if (n == 0) return YART_OK;
    if (n > q->capacity) return fail(YART_ERR_INVALID_ARGUMENT, "yart_queue: batch larger than the queue's capacity");
    DeviceGuard guard(q->accel->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    char* h = static_cast<char*>(q->host);
    char* d = static_cast<char*>(q->dev);
    hipError_t e = hipMemcpyAsync(d, h, n * sizeof(YartRay), hipMemcpyHostToDevice, q->stream);
    if (e != hipSuccess) return failHip(e, "hipMemcpyAsync H2D");
    int st;
    if (any) {
        st = yart_intersect_any_dev(q->accel, reinterpret_cast<const YartRay*>(d), n, reinterpret_cast<uint8_t*>(d + q->offOcc), flags, q->stream);
        if (st != YART_OK) return st;
        e = hipMemcpyAsync(h + q->offOcc, d + q->offOcc, n, hipMemcpyDeviceToHost, q->stream);
    } else {
        st = yart_intersect_closest_dev(q->accel, reinterpret_cast<const YartRay*>(d), n, reinterpret_cast<YartHit*>(d + q->offHits), flags, q->stream);
        if (st != YART_OK) return st;
        e = hipMemcpyAsync(h + q->offHits, d + q->offHits, n * sizeof(YartHit), hipMemcpyDeviceToHost, q->stream);
        if (e == hipSuccess && surface) {
            st = yart_surface_dev(q->accel, reinterpret_cast<const YartRay*>(d), reinterpret_cast<const YartHit*>(d + q->offHits), n,
                                  reinterpret_cast<YartSurface*>(d + q->offSurf), q->stream);
            if (st != YART_OK) return st;
            e = hipMemcpyAsync(h + q->offSurf, d + q->offSurf, n * sizeof(YartSurface), hipMemcpyDeviceToHost, q->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(q->stream);
    return e == hipSuccess ? YART_OK : failHip(e, "yart_queue copy / synchronise");
}

int yart_queue_closest(YartQueue* q, size_t n, unsigned flags, int with_surface) { return queueRun(q, n, flags, false, with_surface != 0); }
int yart_queue_any(YartQueue* q, size_t n, unsigned flags) { return queueRun(q, n, flags, true, false); }

size_t yart_accel_num_nodes(const YartAccel* a) { return a ? a->numNodes : 0; }
size_t yart_accel_num_triangles(const YartAccel* a) { return a ? a->numTriangles : 0; }
size_t yart_accel_num_spheres(const YartAccel* a) { return a ? a->numSpheres : 0; }
int yart_accel_depth(const YartAccel* a) { return a ? a->tri.depth : 0; }
size_t yart_accel_device_bytes(const YartAccel* a) { return a ? a->deviceBytes : 0; }

int yart_accel_get_nodes(const YartAccel* a, YartBvhNode* out)
{
    if (!a || !out) return fail(YART_ERR_INVALID_ARGUMENT, "null argument");
    if (a->refBuiltOnDevice) {
        DeviceGuard guard(a->device);
        const hipError_t e = hipMemcpy(out, a->dRefNodes, a->numNodes * sizeof(YartBvhNode), hipMemcpyDeviceToHost);
        return e == hipSuccess ? YART_OK : failHip(e, "hipMemcpy D2H");
    }
    std::memcpy(out, a->tri.nodes.data(), a->tri.nodes.size() * sizeof(YartBvhNode));
    return YART_OK;
}

int yart_accel_get_triaccel(const YartAccel* a, YartTriAccel* out)
{
    if (!a || !out) return fail(YART_ERR_INVALID_ARGUMENT, "null argument");
    if (a->refBuiltOnDevice) {
        DeviceGuard guard(a->device);
        const hipError_t e = hipMemcpy(out, a->dTriAccel, a->numTriangles * sizeof(YartTriAccel), hipMemcpyDeviceToHost);
        return e == hipSuccess ? YART_OK : failHip(e, "hipMemcpy D2H");
    }
    std::memcpy(out, a->tri.triaccel.data(), a->tri.triaccel.size() * sizeof(YartTriAccel));
    return YART_OK;
}

size_t yart_accel_num_wide_nodes(const YartAccel* a) { return a ? a->wideNodes : 0; }
size_t yart_accel_num_wide_prims(const YartAccel* a) { return a ? a->widePrims : 0; }
int yart_accel_wide_depth(const YartAccel* a) { return a ? a->wideDepth : 0; }
int yart_accel_wide_built_on_device(const YartAccel* a, double* ms_out)
{
    if (ms_out) *ms_out = a ? a->msWideDeviceBuild : 0.0;
    return a && a->wideBuiltOnDevice ? 1 : 0;
}

int yart_accel_ref_built_on_device(const YartAccel* a, double* ms_out, size_t* host_median_splits, size_t* host_projections)
{
    if (ms_out) *ms_out = a ? a->msRefDeviceBuild : 0.0;
    if (host_median_splits) *host_median_splits = a ? a->refHostMedianSplits : 0;
    if (host_projections) *host_projections = a ? a->refHostProjections : 0;
    return a && a->refBuiltOnDevice ? 1 : 0;
}

int yart_accel_get_wide(const YartAccel* a, void* nodes_out, void* rows_a_out, void* rows_bc_out)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "null argument");
    if (!a->hasWide) return fail(YART_ERR_INVALID_ARGUMENT, "accel was created without the wide tree (YART_NO_WIDE)");
    DeviceGuard guard(a->device);
    // the device holds the packed tree only: fetched and unpacked into the node / row arrays (wide_build.h)
    std::vector<uint32_t> words(a->wideUnits * 4);
    std::vector<YartTriAccel> fetched;
    hipError_t e = hipMemcpy(words.data(), a->dWideUnits, a->wideUnits * 16, hipMemcpyDeviceToHost);
    const YartTriAccel* ta = a->tri.triaccel.data();
    if (e == hipSuccess && a->refBuiltOnDevice && a->numTriangles) {
        fetched.resize(a->numTriangles);
        e = hipMemcpy(fetched.data(), a->dTriAccel, a->numTriangles * sizeof(YartTriAccel), hipMemcpyDeviceToHost);
        ta = fetched.data();
    }
    if (e != hipSuccess) return failHip(e, "hipMemcpy D2H");
    yart::WideBvh wide;
    if (!yart::unpackWide(words.data(), a->wideUnits, ta, a->numTriangles, wide) || wide.nodes.size() != a->wideNodes ||
        wide.rowsA.size() != a->widePrims * 4)
        return fail(YART_ERR_HIP, "yart_accel_get_wide: the packed tree on the device is not well formed");
    if (nodes_out) std::memcpy(nodes_out, wide.nodes.data(), wide.nodes.size() * sizeof(yart::WideNode));
    if (rows_a_out && a->widePrims) std::memcpy(rows_a_out, wide.rowsA.data(), wide.rowsA.size() * sizeof(float));
    if (rows_bc_out && a->widePrims) std::memcpy(rows_bc_out, wide.rowsBC.data(), wide.rowsBC.size() * sizeof(float));
    return YART_OK;
}

int yart_build_host(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                    YartBvhNode* nodes_out, size_t* n_nodes, YartTriAccel* triaccel_out)
{
    yart::TriangleBvh bvh;
    std::string err;
    const int st = yart::buildTriangleBvh(verts, n_verts, tris, n_tris, bvh, err);
    if (st != YART_OK) return fail(st, err);
    if (n_nodes) *n_nodes = bvh.nodes.size();
    if (nodes_out) std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(YartBvhNode));
    if (triaccel_out) std::memcpy(triaccel_out, bvh.triaccel.data(), bvh.triaccel.size() * sizeof(YartTriAccel));
    return YART_OK;
}

int yart_camera_basis(const float pos[3], const float dir[3], int32_t width, int32_t height, float fov, float out[12])
{
    if (!pos || !dir || !out || width <= 0 || height <= 0) return fail(YART_ERR_INVALID_ARGUMENT, "yart_camera_basis: null argument or empty film");
    // normal(Vector3f): v * (1.0f / length()) (include/vector.h:136-167), as src/rt.cpp:22-28 passes the direction
    const float len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float inv = 1.0f / len;
    const float d[3] = { dir[0] * inv, dir[1] * inv, dir[2] * inv };
    // Camera::Camera (include/camera.h:9-22): right_ = cross(direction_, up) * (width * fov_ / height_), up = (0, 1, 0);
    // up_ = normal(cross(right_, direction_)) * fov_
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    const float k = static_cast<float>(width) * fov / static_cast<float>(height);
    const float r[3] = { (d[1] * up[2] - d[2] * up[1]) * k, (d[2] * up[0] - d[0] * up[2]) * k, (d[0] * up[1] - d[1] * up[0]) * k };
    const float c[3] = { r[1] * d[2] - r[2] * d[1], r[2] * d[0] - r[0] * d[2], r[0] * d[1] - r[1] * d[0] };
    const float cinv = 1.0f / std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int i = 0; i < 3; ++i) {
        out[i] = pos[i];
        out[3 + i] = d[i];
        out[6 + i] = r[i];
        out[9 + i] = c[i] * cinv * fov;
    }
    return YART_OK;
}

int yart_scene_cover(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, const float* spheres, size_t n_spheres, float out[10])
{
    if (!out || (n_verts && !verts) || (n_spheres && !spheres) || (n_tris && !tris)) return fail(YART_ERR_INVALID_ARGUMENT, "yart_scene_cover: null argument");
    for (size_t i = 0; i < n_tris; ++i)
        for (int c = 0; c < 3; ++c)
            if (tris[i * 5 + c] < 0 || static_cast<size_t>(tris[i * 5 + c]) >= n_verts) return fail(YART_ERR_INVALID_ARGUMENT, "yart_scene_cover: vertex index out of range");
    yart::Cover c = yart::sceneCover(verts, n_verts, spheres, n_spheres);
    c.padMax = yart::trianglePadMax(verts, tris, n_tris, c);
    static_assert(sizeof(yart::Cover) == 40, "ten words");
    std::memcpy(out, &c, sizeof c);
    return YART_OK;
}

int yart_build_wide_host(const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                         const float* spheres, size_t n_spheres, int max_leaf,
                         void* nodes_out, size_t* n_nodes, void* rows_a_out, void* rows_bc_out, size_t* n_prims,
                         int* depth)
{
    yart::TriangleBvh tri;
    yart::WideBvh wide;
    std::string err;
    if (n_tris >= (1u << 30) || n_spheres >= (1u << 30)) return fail(YART_ERR_INVALID_ARGUMENT, "wide BVH: more than 2^30 primitives of one kind");
    int st = yart::buildTriangleBvh(verts, n_verts, tris, n_tris, tri, err);
    const yart::Cover cover = st == YART_OK ? yart::sceneCover(verts, n_verts, spheres, n_spheres) : yart::Cover{};
    if (st == YART_OK) st = yart::buildWideBvh(verts, n_verts, tris, n_tris, tri, spheres, n_spheres, max_leaf, cover, wide, err);
    if (st != YART_OK) return fail(st, err);
    {
        // what is handed out is what the device would hold: packed (wide_build.h) and unpacked again
        yart::PackedWide packed;
        yart::WideBvh back;
        if (!yart::packWide(wide, packed) || !yart::unpackWide(packed.words.data(), packed.numUnits, tri.triaccel.data(), tri.triaccel.size(), back))
            return fail(YART_ERR_INVALID_ARGUMENT, "wide BVH does not fit the packed format");
        wide = std::move(back);
    }
    if (n_nodes) *n_nodes = wide.nodes.size();
    if (n_prims) *n_prims = wide.rowsA.size() / 4;
    if (depth) *depth = wide.depth;
    if (nodes_out) std::memcpy(nodes_out, wide.nodes.data(), wide.nodes.size() * sizeof(yart::WideNode));
    if (rows_a_out) std::memcpy(rows_a_out, wide.rowsA.data(), wide.rowsA.size() * sizeof(float));
    if (rows_bc_out) std::memcpy(rows_bc_out, wide.rowsBC.data(), wide.rowsBC.size() * sizeof(float));
    return YART_OK;
}

int yart_accel_profile_enable(YartAccel* a, int enable)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "null accel");
    a->profile = enable != 0;
    return YART_OK;
}

int yart_accel_profile_read(YartAccel* a, double* ms_sort, double* ms_spheres, double* ms_triangles,
                            uint64_t* n_calls)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "null accel");
    DeviceGuard guard(a->device);
    double ms[3] = { 0.0, 0.0, 0.0 };
    int st = YART_OK;
    std::vector<YartAccel::Span> spans;
    uint64_t calls = 0;
    {
        std::lock_guard<std::mutex> lock(a->profileMutex);
        spans.swap(a->spans);
        calls = a->profiledCalls;
        a->profiledCalls = 0;
    }
    for (auto& s : spans) {
        float t = 0.0f;
        hipError_t e = hipEventSynchronize(s.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, s.start, s.stop);
        if (e != hipSuccess) st = failHip(e, "profile event");
        else ms[s.kind] += t;
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    if (ms_sort) *ms_sort = ms[0];
    if (ms_spheres) *ms_spheres = ms[1];
    if (ms_triangles) *ms_triangles = ms[2];
    if (n_calls) *n_calls = calls;
    return st;
}

} // extern "C"

namespace yart {
hipError_t launchSurfaceRows(const float* verts, uint32_t nVerts, const int32_t* tris, uint32_t nTris, const float* normals,
                             const uint32_t* slotOfInput, void* rows, uint32_t* bad, hipStream_t stream);      // surface.hip
}

namespace {

// yart_accel_set_surface_data for an accel whose leaf order lives on the device: the scene goes up as it is and a kernel
// writes the rows (C4: 0.17 s of host loop + a 640 MB upload -> the uploads of the scene arrays + 2 ms).
int surfaceRowsOnDevice(YartAccel* a, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris, const float* normals,
                        const float* spheres, size_t n_spheres)
{
    DeviceGuard guard(a->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    char* block = nullptr;
    const size_t vb = n_verts * 12, tb = n_tris * 20, off1 = (vb + 255) / 256 * 256, off2 = off1 + (vb + 255) / 256 * 256, off3 = off2 + (tb + 255) / 256 * 256;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&block), off3 + 256);
    if (e != hipSuccess) return fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(surface staging): ") + hipGetErrorString(e));
    void* rows = nullptr;
    int st = YART_OK;
    if ((e = hipMalloc(&rows, n_tris * 64)) != hipSuccess) st = fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(surface rows): ") + hipGetErrorString(e));
    uint32_t bad = 0;
    if (st == YART_OK &&
        ((e = hipMemcpy(block, verts, vb, hipMemcpyHostToDevice)) != hipSuccess ||
         (e = hipMemcpy(block + off1, normals, vb, hipMemcpyHostToDevice)) != hipSuccess ||
         (e = hipMemcpy(block + off2, tris, tb, hipMemcpyHostToDevice)) != hipSuccess ||
         (e = hipMemset(block + off3, 0, 4)) != hipSuccess ||
         (e = yart::launchSurfaceRows(reinterpret_cast<const float*>(block), static_cast<uint32_t>(n_verts), reinterpret_cast<const int32_t*>(block + off2),
                                      static_cast<uint32_t>(n_tris), reinterpret_cast<const float*>(block + off1),
                                      static_cast<const uint32_t*>(a->dSlotOfInput), rows, reinterpret_cast<uint32_t*>(block + off3), nullptr)) != hipSuccess ||
         (e = hipMemcpy(&bad, block + off3, 4, hipMemcpyDeviceToHost)) != hipSuccess))
        st = failHip(e, "surface rows on the device");
    (void)hipFree(block);
    if (st == YART_OK && bad) st = fail(YART_ERR_INVALID_ARGUMENT, "yart_accel_set_surface_data: vertex index out of range");
    if (st != YART_OK) { (void)hipFree(rows); return st; }
    (void)hipFree(a->dSurfRows); (void)hipFree(a->dSurfSpheres);
    a->dSurfRows = rows;
    a->dSurfSpheres = nullptr;
    a->deviceBytes += n_tris * 64;
    return upload(&a->dSurfSpheres, spheres, n_spheres * 16, a->deviceBytes);
}

} // namespace

extern "C" {

int yart_accel_set_surface_data(YartAccel* a, const float* verts, size_t n_verts, const int32_t* tris, size_t n_tris,
                                const float* normals, const float* spheres, size_t n_spheres)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "accel is null");
    {
        const char* host = std::getenv("YART_SURFACE_ROWS_HOST");
        if (a->refBuiltOnDevice && a->dSlotOfInput && n_tris == a->numTriangles && n_spheres == a->numSpheres && n_tris > 0 &&
            n_tris <= 0xFFFFFFFFull / 5 && n_verts <= 0xFFFFFFFFull / 3 && verts && tris && normals && (!n_spheres || spheres) &&
            !(host && host[0] == '1'))
            return surfaceRowsOnDevice(a, verts, n_verts, tris, n_tris, normals, spheres, n_spheres);
    }
    if (a->refBuiltOnDevice && n_tris == a->numTriangles) {
        // the leaf order lives on the device: fetched once
        std::lock_guard<std::mutex> lock(a->hostCopyMutex);
        if (a->tri.slotOfInput.size() != n_tris) {
            DeviceGuard guard(a->device);
            a->tri.slotOfInput.resize(n_tris);
            const hipError_t e = hipMemcpy(a->tri.slotOfInput.data(), a->dSlotOfInput, n_tris * sizeof(uint32_t), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { a->tri.slotOfInput.clear(); return failHip(e, "hipMemcpy D2H"); }
        }
    }
    if (n_tris != a->numTriangles || n_spheres != a->numSpheres || a->tri.slotOfInput.size() != n_tris)
        return fail(YART_ERR_INVALID_ARGUMENT, "yart_accel_set_surface_data: not the scene the accel was created from");
    if ((n_tris && (!verts || !tris || !normals)) || (n_spheres && !spheres))
        return fail(YART_ERR_INVALID_ARGUMENT, "yart_accel_set_surface_data: null argument");
    std::vector<float> rows(n_tris * 16);       // one 64-byte line per triangle
    std::atomic<bool> bad{ false };
    yart::parallelChunks(n_tris, yart::buildThreads(), size_t(1) << 16, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const int32_t* t = tris + i * 5;
            for (int c = 0; c < 3; ++c)
                if (t[c] < 0 || static_cast<size_t>(t[c]) >= n_verts) { bad = true; return; }
            const float* p0 = verts + static_cast<size_t>(t[0]) * 3;
            const float* p1 = verts + static_cast<size_t>(t[1]) * 3;
            const float* p2 = verts + static_cast<size_t>(t[2]) * 3;
            const float e1[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };       // include/triangle.h:158-164
            const float e2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
            const float cx = e1[1] * e2[2] - e1[2] * e2[1];
            const float cy = e1[2] * e2[0] - e1[0] * e2[2];
            const float cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float inv = 1.0f / std::sqrt(cx * cx + cy * cy + cz * cz);           // include/vector.h:165-168
            const size_t slot = a->tri.slotOfInput[i];
            float* r = &rows[slot * 16];
            r[0] = cx * inv; r[1] = cy * inv; r[2] = cz * inv;
            for (int c = 0; c < 3; ++c) std::memcpy(r + 3 + c * 3, normals + static_cast<size_t>(t[c]) * 3, 12);
            std::memcpy(r + 12, t + 3, 8);      // meshIdx, triIdx
            r[14] = r[15] = 0.0f;
        }
    });
    if (bad) return fail(YART_ERR_INVALID_ARGUMENT, "yart_accel_set_surface_data: vertex index out of range");
    DeviceGuard guard(a->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    (void)hipFree(a->dSurfRows); (void)hipFree(a->dSurfSpheres);
    a->dSurfRows = a->dSurfSpheres = nullptr;
    int st = upload(&a->dSurfRows, rows.data(), rows.size() * 4, a->deviceBytes);
    if (st == YART_OK) st = upload(&a->dSurfSpheres, spheres, n_spheres * 16, a->deviceBytes);
    return st;
}

int yart_surface_dev(YartAccel* a, const YartRay* d_rays, const YartHit* d_hits, size_t n, YartSurface* d_out, void* stream)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "accel is null");
    if (n == 0) return YART_OK;
    if (!d_rays || !d_hits || !d_out) return fail(YART_ERR_INVALID_ARGUMENT, "yart_surface_dev: null argument");
    if (n > 0xFFFFFFFFull / 2) return fail(YART_ERR_INVALID_ARGUMENT, "yart_surface_dev: batch too large");
    if ((a->numTriangles && !a->dSurfRows) || (a->numSpheres && !a->dSurfSpheres))
        return fail(YART_ERR_INVALID_ARGUMENT, "yart_surface_dev: call yart_accel_set_surface_data first");
    DeviceGuard guard(a->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    const hipError_t e = yart::launchSurface(d_rays, d_hits, static_cast<uint32_t>(n), a->dSurfRows, a->dSurfSpheres,
                                             d_out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? YART_OK : failHip(e, "surface kernel");
}

int yart_surface(YartAccel* a, const YartRay* rays, const YartHit* hits, size_t n, YartSurface* out)
{
    if (!a) return fail(YART_ERR_INVALID_ARGUMENT, "accel is null");
    if (n == 0) return YART_OK;
    if (!rays || !hits || !out) return fail(YART_ERR_INVALID_ARGUMENT, "yart_surface: null argument");
    DeviceGuard guard(a->device);
    if (!guard.ok) return fail(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    char* block = nullptr;
    const size_t rayBytes = n * sizeof(YartRay), hitBytes = n * sizeof(YartHit), outBytes = n * sizeof(YartSurface);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&block), rayBytes + hitBytes + outBytes);
    if (e != hipSuccess) return fail(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    int st = YART_OK;
    if ((e = hipMemcpy(block, rays, rayBytes, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(block + rayBytes, hits, hitBytes, hipMemcpyHostToDevice)) != hipSuccess) st = failHip(e, "hipMemcpy");
    if (st == YART_OK)
        st = yart_surface_dev(a, reinterpret_cast<const YartRay*>(block), reinterpret_cast<const YartHit*>(block + rayBytes), n,
                              reinterpret_cast<YartSurface*>(block + rayBytes + hitBytes), nullptr);
    if (st == YART_OK && (e = hipMemcpy(out, block + rayBytes + hitBytes, outBytes, hipMemcpyDeviceToHost)) != hipSuccess)
        st = failHip(e, "hipMemcpy");
    (void)hipFree(block);
    return st;
}

struct YartObjScene {
    yart::ObjScene scene;
    size_t numVertices = 0, numTriangles = 0;
};

int yart_obj_load(const char* folder, const char* file, YartObjScene** out)
{
    if (!file || !out) return fail(YART_ERR_INVALID_ARGUMENT, "yart_obj_load: null argument");
    *out = nullptr;
    auto s = std::make_unique<YartObjScene>();
    std::string err;
    if (!yart::loadObjScene(folder ? folder : "", file, s->scene, err)) return fail(YART_ERR_INVALID_ARGUMENT, "yart_obj_load: " + err);
    for (const auto& m : s->scene.meshes) {
        s->numVertices += m.positions.size() / 3;
        s->numTriangles += m.indices.size() / 3;
    }
    if (s->numVertices > 0x7fffffffu || s->numTriangles > 0x7fffffffu) return fail(YART_ERR_INVALID_ARGUMENT, "yart_obj_load: scene too large");
    *out = s.release();
    return YART_OK;
}

void yart_obj_free(YartObjScene* scene) { delete scene; }
size_t yart_obj_num_meshes(const YartObjScene* s) { return s ? s->scene.meshes.size() : 0; }
size_t yart_obj_num_vertices(const YartObjScene* s) { return s ? s->numVertices : 0; }
size_t yart_obj_num_triangles(const YartObjScene* s) { return s ? s->numTriangles : 0; }
size_t yart_obj_num_materials(const YartObjScene* s) { return s ? s->scene.materials.size() : 0; }
size_t yart_obj_num_spheres(const YartObjScene* s) { return s ? s->scene.spheres.size() / 4 : 0; }

int yart_obj_get(const YartObjScene* s, float* verts, float* normals, int32_t* tris, float* spheres,
                 YartObjMesh* meshes, YartObjMaterial* materials)
{
    if (!s) return fail(YART_ERR_INVALID_ARGUMENT, "yart_obj_get: null scene");
    size_t v0 = 0, t0 = 0;
    for (size_t mi = 0; mi < s->scene.meshes.size(); ++mi) {
        const yart::ObjMesh& m = s->scene.meshes[mi];
        const size_t nv = m.positions.size() / 3, nt = m.indices.size() / 3;
        if (verts) std::memcpy(verts + v0 * 3, m.positions.data(), nv * 12);
        if (normals) std::memcpy(normals + v0 * 3, m.normals.data(), nv * 12);
        if (tris) {
            for (size_t t = 0; t < nt; ++t) {
                int32_t* r = tris + (t0 + t) * 5;
                for (int k = 0; k < 3; ++k) r[k] = m.indices[t * 3 + k] + static_cast<int32_t>(v0);
                r[3] = static_cast<int32_t>(mi);
                r[4] = static_cast<int32_t>(t);
            }
        }
        if (meshes) {
            meshes[mi] = YartObjMesh{ static_cast<int32_t>(v0), static_cast<int32_t>(nv), static_cast<int32_t>(t0),
                                      static_cast<int32_t>(nt), m.material, m.normalsFromFile ? 1 : 0 };
        }
        v0 += nv;
        t0 += nt;
    }
    if (spheres && !s->scene.spheres.empty()) std::memcpy(spheres, s->scene.spheres.data(), s->scene.spheres.size() * 4);
    if (materials) {
        for (size_t i = 0; i < s->scene.materials.size(); ++i) {
            const yart::ObjMaterial& m = s->scene.materials[i];
            YartObjMaterial& o = materials[i];
            std::memcpy(o.diffuse, m.diffuse, 12);
            std::memcpy(o.specular, m.specular, 12);
            o.shininess = m.shininess;
            o.ior = m.ior;
            o.illum = m.illum;
            o.bsdf = yart::objBsdfKind(m);
        }
    }
    return YART_OK;
}

const char* yart_obj_mesh_name(const YartObjScene* s, size_t i)
{
    return s && i < s->scene.meshes.size() ? s->scene.meshes[i].name.c_str() : "";
}
const char* yart_obj_material_name(const YartObjScene* s, size_t i)
{
    return s && i < s->scene.materials.size() ? s->scene.materials[i].name.c_str() : "";
}

int yart_film_tonemap(const float* rgb, size_t n_pixels, uint8_t* bgr_out)
{
    if (n_pixels && (!rgb || !bgr_out)) return fail(YART_ERR_INVALID_ARGUMENT, "yart_film_tonemap: null argument");
    yart::toneMapBgr(rgb, n_pixels, bgr_out);
    return YART_OK;
}

int yart_film_write_bmp(const char* path, const float* rgb, int32_t width, int32_t height)
{
    if (!path || !rgb || width <= 0 || height <= 0 || static_cast<int64_t>(width) * height > (int64_t(1) << 28))
        return fail(YART_ERR_INVALID_ARGUMENT, "yart_film_write_bmp: null argument or bad size");
    std::string err;
    if (!yart::writeBmp(path, rgb, width, height, err)) return fail(YART_ERR_INVALID_ARGUMENT, "yart_film_write_bmp: " + err);
    return YART_OK;
}

const char* yart_status_string(int status)
{
    switch (status) {
    case YART_OK: return "ok";
    case YART_ERR_INVALID_ARGUMENT: return "invalid argument";
    case YART_ERR_NO_DEVICE: return "no HIP device";
    case YART_ERR_HIP: return "HIP runtime error";
    case YART_ERR_OUT_OF_MEMORY: return "out of memory";
    case YART_ERR_TREE_TOO_DEEP: return "BVH too deep";
    default: return "unknown status";
    }
}

const char* yart_last_error(void) { return g_lastError.c_str(); }

} // extern "C"
