// noalloc.hip -- yart_intersect_closest_noalloc_dev: the closest-hit call for callers that cannot
// allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
// No counterpart in the reference.
//
// yart_intersect_closest_dev needs device memory beyond the caller's rays and hits: the work-counter
// slots of the persistent kernels (ranges of the batch handed out by atomic adds, the wide kernel's
// "rays left for the fix-up launch" flag) and, for YART_SORT_RAYS, the coherence sort's keys,
// permutation and histograms.  There the library keeps them per stream and grows them with hipMalloc.
// Here they live in ONE static device array of kNoallocBytes per GPU, so this entry point calls no
// hipMalloc / hipFree / allocating routine; the same kernels run, with the same results.
//
// The array's NAME is a host-side stand-in, not a device address: the address of each device's copy
// comes from hipGetSymbolAddress, asked once per device with that device current (noallocScratchOfDevice,
// called by yart_accel_create), and only that pointer is ever handed to kernels, memsets and the sort.
//
// Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
// threads) race.  What must start at zero (the counter slots) is cleared by each launcher on the
// stream; the sort workspace is fully rewritten before it is read.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/yart_amd.h"
#include "accel.h"
#include "traverse.h"

namespace yart {
namespace {

constexpr size_t kNoallocBytes = size_t(40) << 20;       // 1 KiB of counter slots + sort workspace of YART_NOALLOC_MAX_RAYS rays
static_assert(kNoallocBytes >= kCounterBlockBytes + 4 * (size_t(YART_NOALLOC_MAX_RAYS) * 4 + 256) + (size_t(4) << 20),
              "scratch holds the counter slots, four n-element sort buffers and 4 MiB of radix-sort histograms");

static __device__ __align__(256) unsigned char scratch[kNoallocBytes];

constexpr int kMaxDevices = 64;
std::mutex g_mutex;
void* g_ptr[kMaxDevices];        // per device: each GPU holds its own copy of the array at its own address
bool g_known[kMaxDevices];

} // namespace

int noallocScratchOfDevice(int device, void** ptr, size_t* bytes)
{
    *ptr = nullptr;
    *bytes = 0;
    if (device < 0 || device >= kMaxDevices) return failStatus(YART_ERR_INVALID_ARGUMENT, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_known[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess || cur != device)
            return failStatus(YART_ERR_HIP, "noalloc scratch: the accel's device must be current when its copy of the scratch array is looked up");
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return failStatus(YART_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_ptr[device] = p;
        g_known[device] = true;
    }
    *ptr = g_ptr[device];
    *bytes = kNoallocBytes;
    return YART_OK;
}

} // namespace yart

extern "C" int yart_intersect_closest_noalloc_dev(YartAccel* accel, const YartRay* d_rays, size_t n, YartHit* d_hits,
                                                  unsigned flags, void* stream)
{
    if (!accel) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "null accel");
    if (n > YART_NOALLOC_MAX_RAYS)
        return yart::failStatus(YART_ERR_INVALID_ARGUMENT,
                                "yart_intersect_closest_noalloc_dev: batch larger than YART_NOALLOC_MAX_RAYS (2097152) rays; "
                                "split it or use yart_intersect_closest_dev");
    if (!accel->noallocScratch || accel->noallocBytes < yart::kCounterBlockBytes)
        return yart::failStatus(YART_ERR_HIP, "yart_intersect_closest_noalloc_dev: the scratch array was not set up at yart_accel_create");
    yart::CallScratch sc;
    unsigned char* base = static_cast<unsigned char*>(accel->noallocScratch);       // the device pointer, never the symbol
    sc.counters = base;
    sc.sortWs = base + yart::kCounterBlockBytes;
    sc.sortWsBytes = accel->noallocBytes - yart::kCounterBlockBytes;
    return yart::runIntersect(accel, yart::Launch{ false, false, d_rays, n, d_hits, nullptr, nullptr, flags,
                                                   static_cast<hipStream_t>(stream) }, &sc);
}
