// accel.h -- the accel object behind the opaque YartAccel handle of include/yart_amd.h.
// Private to the library: shared by the C-ABI layer (capi.cpp) and the entry point that allocates
// nothing (noalloc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/yart_amd.h"
#include "bvh_build.h"
#include "cover.h"

struct YartAccel {
    int device = 0;
    int numCUs = 0;
    yart::TriangleBvh tri;            // host copies (nodes / triaccel kept for introspection)
    size_t numTriPairs = 0, numSphPairs = 0, numSpheres = 0;
    size_t numTriangles = 0, numNodes = 0;   // sizes of the reference's arrays, wherever they were built
    // YART_BUILD_DEVICE=1 and ref_build_dev.hip took the scene: the reference's arrays exist on the device only
    // (dRefNodes, dTriAccel, dSlotOfInput); tri.nodes / tri.triaccel stay empty, tri.slotOfInput is fetched on
    // first use (yart_accel_set_surface_data) under hostCopyMutex.
    bool refBuiltOnDevice = false;
    double msRefDeviceBuild = 0.0;
    size_t refHostMedianSplits = 0, refHostProjections = 0;
    void* dRefNodes = nullptr;
    void* dSlotOfInput = nullptr;
    std::mutex hostCopyMutex;
    int sphDepth = 0;
    float sceneLo[3], sceneHi[3];     // bounds of everything, for the sort key grid
    yart::Cover cover;                // the rays the primitives' boxes are padded for (cover.h)
    float gridLo[3] = { 0, 0, 0 }, gridHi[3] = { 0, 0, 0 };   // bounds of the wide tree's (padded) primitive boxes
    float cullReach = 0.0f;           // traverse.h: WideParams::cullReach
    // device memory
    void* dTriPairs = nullptr;
    void* dTriAccel = nullptr;
    void* dSphPairs = nullptr;
    void* dSphRecs = nullptr;
    void* dSphIndex = nullptr;
    void* dLeafOfSlot = nullptr;      // per triangle slot: where its leaf's record sits in dTriPairs (traverse.h: launchLeafOfSlot)
    void* dWideUnits = nullptr;       // the engine's own 4-wide triangle+sphere tree, packed (wide_build.h: PackedWide)
    size_t wideUnits = 0;
    void* dSurfRows = nullptr;        // hit post-processing (yart_accel_set_surface_data)
    void* dSurfSpheres = nullptr;
    bool hasWide = false;
    bool wideBuiltOnDevice = false;    // YART_BUILD_DEVICE=1 and the device builder took the scene
    double msWideDeviceBuild = 0.0;
    size_t wideNodes = 0, widePrims = 0;   // array sizes; the device copy is the only one kept
    int wideDepth = 0;
    size_t deviceBytes = 0;
    // per-kernel timing (yart_accel_profile_*): kind 0 = sort, 1 = other passes, 2 = main traversal kernel
    struct Span { hipEvent_t start, stop; int kind; };
    bool profile = false;
    std::mutex profileMutex;          // guards spans / profiledCalls: intersect calls come from several host threads
    std::vector<Span> spans;
    uint64_t profiledCalls = 0;
    // scheduling knobs of the persistent kernel (YART_REFILL_MIN / YART_NODE_MIN override)
    uint32_t refillMin = 16, nodeMin = 24;                    // reference-tree kernels
    uint32_t wideRefillMin = 16, wideNodeMin = 40, wideSphereMin = 8, wideChunk = 64, wideBlocksPerCU = 0, wideLaneCap = 0;   // wide-tree kernel
    uint32_t sortMinRays = 3u << 19;   // below ~1.5 M rays the sort costs more than it saves (profiles/r01_batch_sweep.jsonl)
    uint32_t sortOctShift = 0, sortGridBits = 7;      // 7: 24-bit keys = 3 radix passes; finer grids measured no better
    // Small host-buffer calls (the one-ray-per-call pattern of the reference's renderer,
    // src/renderer.cpp:59,104): a pool of { stream, pinned host block the kernels read and
    // write in place } so that such a call costs launches + one stream synchronise, no
    // allocation and no copy.  A context is taken for the duration of one call.
    struct Staging { hipStream_t stream = nullptr; void* host = nullptr; void* dev = nullptr;
                     void* big = nullptr; size_t bigBytes = 0; };     // + a device block kept for larger batches
    std::mutex stagingMutex;
    std::vector<Staging> stagingFree;

    // What the launches of ONE stream need beside the caller's arrays: the work-counter slots of the
    // persistent kernels (4 x 256 bytes, one per launch of a call) and the coherence sort's workspace,
    // kept and grown on demand.  Launches on a stream run in order, so a stream's calls can share
    // them; calls on different streams never do.  `submit` is held while a call enqueues its
    // launches (two host threads feeding one stream would otherwise interleave sort and traversal).
    struct StreamCtx {
        std::mutex submit;
        void* counters = nullptr;
        void* sortWs = nullptr;
        size_t sortWsBytes = 0;
    };
    std::mutex ctxMutex;
    std::unordered_map<void*, StreamCtx*> ctx;

    // the static scratch of the entry point that allocates nothing (noalloc.hip), on this accel's device
    void* noallocScratch = nullptr;
    size_t noallocBytes = 0;
};

namespace yart {

constexpr uint32_t kDeferCap = 16384;             // rays the wide kernel can list for the fix-up launches of one call
constexpr size_t kCounterBlockBytes = 1024 + size_t(kDeferCap) * 4;   // 4 work-counter slots of 256 bytes + that list

// Scratch of one intersect call.  counters: kCounterBlockBytes of device memory.
struct CallScratch {
    void* counters = nullptr;
    void* sortWs = nullptr;
    size_t sortWsBytes = 0;
};

struct Launch {
    bool any, count;
    const YartRay* rays;
    size_t n;
    YartHit* hits;
    uint8_t* occluded;
    uint32_t* counts;
    unsigned flags;
    hipStream_t stream;
};

// Enqueues the launches of one intersect call (capi.cpp).  scratch == nullptr: the stream's own
// context is used and grown as needed (may allocate); otherwise nothing is allocated and a batch
// the given workspace cannot sort is an error.
int runIntersect(YartAccel* a, const Launch& L, const CallScratch* scratch);

// noalloc.hip: address and size of this device's copy of the static scratch array (device current).
int noallocScratchOfDevice(int device, void** ptr, size_t* bytes);

int failStatus(int status, const std::string& msg);

} // namespace yart
