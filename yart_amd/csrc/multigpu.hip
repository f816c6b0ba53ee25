// multigpu.hip -- image mode across the GPUs of a node behind the C-ABI (include/yart_amd.h, "Multi-GPU
// image mode"): the reference's tile list dealt round-robin to ranks, and the framebuffer gather
// over RCCL.  RCCL is reached through dlopen so that the library itself has no hard dependency on it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"
#include "accel.h"

namespace yart {
namespace {

struct TileRect { int32_t x0, y0, x1, y1; };

// Renderer::render's enumeration (src/renderer.cpp:163-217)
std::vector<TileRect> tileList(int32_t width, int32_t height, int32_t tile)
{
    std::vector<TileRect> out;
    if (width <= 0 || height <= 0 || tile <= 0) return out;
    const int32_t fx = width / tile, fy = height / tile;
    out.reserve(static_cast<size_t>(fx + 1) * (fy + 1));
    for (int32_t i = 0; i < fx; ++i)
        for (int32_t j = 0; j < fy; ++j) out.push_back({ i * tile, j * tile, (i + 1) * tile, (j + 1) * tile });
    const int32_t leftW = width - fx * tile, leftH = height - fy * tile;
    if (leftW > 0) for (int32_t j = 0; j < fy; ++j) out.push_back({ fx * tile, j * tile, width, (j + 1) * tile });
    if (leftH > 0) for (int32_t i = 0; i < fx; ++i) out.push_back({ i * tile, fy * tile, (i + 1) * tile, height });
    if (leftW > 0 && leftH > 0) out.push_back({ fx * tile, fy * tile, width, height });
    return out;
}

size_t rankPixelCount(const std::vector<TileRect>& tiles, int32_t rank, int32_t world)
{
    size_t n = 0;
    for (size_t t = static_cast<size_t>(rank); t < tiles.size(); t += static_cast<size_t>(world))
        n += static_cast<size_t>(tiles[t].x1 - tiles[t].x0) * static_cast<size_t>(tiles[t].y1 - tiles[t].y0);
    return n;
}

// frame[pixel(p)] = part[p] for the pixels of one rank's tiles.  One thread per pixel of the part; the tile a
// pixel belongs to is found by binary search over the prefix sums of the rank's tile sizes.
__global__ void __launch_bounds__(256) scatterTiles(const float* __restrict__ part, uint32_t nPixels, const int4* __restrict__ rects,
                                                    const uint32_t* __restrict__ firstPixel, uint32_t nTiles, int32_t width,
                                                    float* __restrict__ frame)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= nPixels) return;
    uint32_t lo = 0, hi = nTiles;                       // last tile whose first pixel is <= p
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (firstPixel[mid] <= p) lo = mid; else hi = mid;
    }
    const int4 r = rects[lo];
    const uint32_t local = p - firstPixel[lo];
    const uint32_t w = (uint32_t)(r.z - r.x);
    const size_t pixel = (size_t)(r.y + (int32_t)(local / w)) * (size_t)width + (size_t)(r.x + (int32_t)(local % w));
    frame[pixel * 3 + 0] = part[(size_t)p * 3 + 0];
    frame[pixel * 3 + 1] = part[(size_t)p * 3 + 1];
    frame[pixel * 3 + 2] = part[(size_t)p * 3 + 2];
}

// ---- RCCL through dlopen --------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // a copy already in the process (PyTorch brings its own) is used first
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("librccl not found: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int ncclFail(ncclResult_t e, const char* what)
{
    Rccl* r = rccl();
    return failStatus(YART_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

static_assert(sizeof(ncclUniqueId) == 128, "yart_comm_unique_id hands out 128 bytes");

} // namespace
} // namespace yart

struct YartComm {
    int32_t rank = 0, world = 1;
    int device = 0;
    ncclComm_t comm = nullptr;
    // rank 0: staging for the other ranks' parts and the tile tables of the last frame shape
    void* staging = nullptr;
    size_t stagingBytes = 0;
    void* tables = nullptr;
    size_t tablesBytes = 0;
    int32_t tabWidth = 0, tabHeight = 0, tabTile = 0;
    hipStream_t lastStream = nullptr;   // the stream of the last gather: ONE gather may be in flight per communicator; its
    bool used = false;                  // scatter kernels read `tables` and `staging`, which a later call may replace
    std::vector<size_t> tabRectOffset, tabFirstOffset, tabTiles;     // per rank: byte offsets into `tables`, tile counts
};

extern "C" {

size_t yart_tiles_count(int32_t width, int32_t height, int32_t tile)
{
    return yart::tileList(width, height, tile).size();
}

int yart_tiles_list(int32_t width, int32_t height, int32_t tile, int32_t* out)
{
    if (width <= 0 || height <= 0 || tile <= 0 || !out) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_tiles_list: bad argument");
    const auto tiles = yart::tileList(width, height, tile);
    std::memcpy(out, tiles.data(), tiles.size() * sizeof(yart::TileRect));
    return YART_OK;
}

size_t yart_tiles_rank_pixels(int32_t width, int32_t height, int32_t tile, int32_t rank, int32_t world, uint32_t* out)
{
    if (width <= 0 || height <= 0 || tile <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    if (static_cast<int64_t>(width) * height > (int64_t(1) << 31)) return 0;
    const auto tiles = yart::tileList(width, height, tile);
    if (!out) return yart::rankPixelCount(tiles, rank, world);
    size_t n = 0;
    for (size_t t = static_cast<size_t>(rank); t < tiles.size(); t += static_cast<size_t>(world)) {
        const yart::TileRect& r = tiles[t];
        for (int32_t y = r.y0; y < r.y1; ++y)
            for (int32_t x = r.x0; x < r.x1; ++x) out[n++] = static_cast<uint32_t>(y) * static_cast<uint32_t>(width) + static_cast<uint32_t>(x);
    }
    return n;
}

int yart_comm_unique_id(void* id128)
{
    if (!id128) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_comm_unique_id: null argument");
    yart::Rccl* r = yart::rccl();
    if (!r->error.empty()) return yart::failStatus(YART_ERR_NO_DEVICE, r->error);
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return yart::ncclFail(e, "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof id);
    return YART_OK;
}

int yart_comm_create(const void* id128, int32_t rank, int32_t world, int device, YartComm** out)
{
    if (!id128 || !out || world <= 0 || rank < 0 || rank >= world) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_comm_create: bad argument");
    *out = nullptr;
    yart::Rccl* r = yart::rccl();
    if (!r->error.empty()) return yart::failStatus(YART_ERR_NO_DEVICE, r->error);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return yart::failStatus(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    YartComm* c = new (std::nothrow) YartComm();
    if (!c) return yart::failStatus(YART_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    const ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != ncclSuccess) { delete c; return yart::ncclFail(e, "ncclCommInitRank"); }
    *out = c;
    return YART_OK;
}

void yart_comm_destroy(YartComm* c)
{
    if (!c) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    if (c->comm) (void)yart::rccl()->CommDestroy(c->comm);
    (void)hipFree(c->staging);
    (void)hipFree(c->tables);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
}

int yart_frame_gather(YartComm* c, const float* d_local, int32_t width, int32_t height, int32_t tile, float* d_frame, void* streamPtr)
{
    using namespace yart;
    if (!c || width <= 0 || height <= 0 || tile <= 0) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_frame_gather: bad argument");
    if (static_cast<int64_t>(width) * height > (int64_t(1) << 30)) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_frame_gather: frame too large");
    hipStream_t stream = static_cast<hipStream_t>(streamPtr);
    Rccl* r = rccl();
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) != hipSuccess) return failStatus(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{ prev };
    const auto tiles = tileList(width, height, tile);
    const size_t mine = rankPixelCount(tiles, c->rank, c->world);
    if (mine && !d_local) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_frame_gather: d_local_rgb is null");
    if (c->rank != 0) {
        if (mine == 0) return YART_OK;
        const ncclResult_t e = r->Send(d_local, mine * 3, ncclFloat, 0, c->comm, stream);
        return e == ncclSuccess ? YART_OK : ncclFail(e, "ncclSend");
    }
    if (!d_frame) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_frame_gather: d_frame_rgb is null on rank 0");

    // rank 0: tile tables of this frame shape on the device (kept until the shape changes)
    hipError_t he = hipSuccess;
    // An earlier gather's scatter kernels may still be reading the tables and the staging block (streams created
    // non-blocking do not wait for the copies and frees below): wait for it before either is replaced.  One gather
    // may be in flight per communicator; a caller that needs two at once creates two.
    auto drainEarlier = [&]() -> hipError_t {
        if (!c->used) return hipSuccess;
        const hipError_t e = hipStreamSynchronize(c->lastStream);
        c->used = false;
        return e;
    };
    if (c->used && c->lastStream != stream && (he = drainEarlier()) != hipSuccess) return failStatus(YART_ERR_HIP, std::string("hipStreamSynchronize(earlier gather): ") + hipGetErrorString(he));
    if (c->tabWidth != width || c->tabHeight != height || c->tabTile != tile) {
        if ((he = drainEarlier()) != hipSuccess) return failStatus(YART_ERR_HIP, std::string("hipStreamSynchronize(earlier gather): ") + hipGetErrorString(he));
        std::vector<char> host;
        c->tabRectOffset.assign(c->world, 0); c->tabFirstOffset.assign(c->world, 0); c->tabTiles.assign(c->world, 0);
        for (int32_t rk = 0; rk < c->world; ++rk) {
            std::vector<TileRect> rects;
            std::vector<uint32_t> first;
            uint32_t acc = 0;
            for (size_t t = static_cast<size_t>(rk); t < tiles.size(); t += static_cast<size_t>(c->world)) {
                rects.push_back(tiles[t]);
                first.push_back(acc);
                acc += static_cast<uint32_t>(tiles[t].x1 - tiles[t].x0) * static_cast<uint32_t>(tiles[t].y1 - tiles[t].y0);
            }
            auto append = [&](const void* p, size_t bytes) {
                const size_t at = (host.size() + 255) & ~size_t(255);
                host.resize(at + bytes);
                if (bytes) std::memcpy(host.data() + at, p, bytes);
                return at;
            };
            c->tabTiles[rk] = rects.size();
            c->tabRectOffset[rk] = append(rects.data(), rects.size() * sizeof(TileRect));
            c->tabFirstOffset[rk] = append(first.data(), first.size() * 4);
        }
        if (host.size() > c->tablesBytes) {
            (void)hipFree(c->tables);
            c->tables = nullptr; c->tablesBytes = 0;
            if ((he = hipMalloc(&c->tables, host.size())) != hipSuccess) return failStatus(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(tile tables): ") + hipGetErrorString(he));
            c->tablesBytes = host.size();
        }
        c->tabWidth = c->tabHeight = c->tabTile = 0;           // not valid until the copy below has succeeded
        if ((he = hipMemcpy(c->tables, host.data(), host.size(), hipMemcpyHostToDevice)) != hipSuccess) return failStatus(YART_ERR_HIP, std::string("hipMemcpy(tile tables): ") + hipGetErrorString(he));
        c->tabWidth = width; c->tabHeight = height; c->tabTile = tile;
    }
    // staging for the other ranks' parts
    std::vector<size_t> count(c->world), offset(c->world, 0);
    size_t others = 0;
    for (int32_t rk = 0; rk < c->world; ++rk) {
        count[rk] = rankPixelCount(tiles, rk, c->world);
        if (rk > 0) { offset[rk] = others; others += (count[rk] * 12 + 255) & ~size_t(255); }
    }
    if (others > c->stagingBytes) {
        if ((he = drainEarlier()) != hipSuccess) return failStatus(YART_ERR_HIP, std::string("hipStreamSynchronize(earlier gather): ") + hipGetErrorString(he));
        (void)hipFree(c->staging);
        c->staging = nullptr; c->stagingBytes = 0;
        if ((he = hipMalloc(&c->staging, others)) != hipSuccess) return failStatus(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(gather staging): ") + hipGetErrorString(he));
        c->stagingBytes = others;
    }
    c->lastStream = stream;
    c->used = true;
    if (c->world > 1) {
        ncclResult_t e = r->GroupStart();
        for (int32_t rk = 1; rk < c->world && e == ncclSuccess; ++rk)
            if (count[rk]) e = r->Recv(static_cast<char*>(c->staging) + offset[rk], count[rk] * 3, ncclFloat, rk, c->comm, stream);
        const ncclResult_t e2 = r->GroupEnd();
        if (e != ncclSuccess) return ncclFail(e, "ncclRecv");
        if (e2 != ncclSuccess) return ncclFail(e2, "ncclGroupEnd");
    }
    for (int32_t rk = 0; rk < c->world; ++rk) {
        if (!count[rk]) continue;
        const float* part = rk == 0 ? d_local : reinterpret_cast<const float*>(static_cast<char*>(c->staging) + offset[rk]);
        const char* tab = static_cast<const char*>(c->tables);
        hipLaunchKernelGGL(scatterTiles, dim3((static_cast<uint32_t>(count[rk]) + 255u) / 256u), dim3(256), 0, stream, part,
                           static_cast<uint32_t>(count[rk]), reinterpret_cast<const int4*>(tab + c->tabRectOffset[rk]),
                           reinterpret_cast<const uint32_t*>(tab + c->tabFirstOffset[rk]), static_cast<uint32_t>(c->tabTiles[rk]), width, d_frame);
        if ((he = hipGetLastError()) != hipSuccess) return failStatus(YART_ERR_HIP, std::string("scatterTiles: ") + hipGetErrorString(he));
    }
    return YART_OK;
}

int yart_comm_allreduce_u64(YartComm* c, uint64_t* d_values, size_t n, void* streamPtr)
{
    using namespace yart;
    if (!c || (n && !d_values)) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_comm_allreduce_u64: null argument");
    if (n == 0 || c->world == 1) return YART_OK;             // the sum over one rank is the value itself
    Rccl* r = rccl();
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) != hipSuccess) return failStatus(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    const ncclResult_t e = r->AllReduce(d_values, d_values, n, ncclUint64, ncclSum, c->comm, static_cast<hipStream_t>(streamPtr));
    if (prev >= 0) (void)hipSetDevice(prev);
    return e == ncclSuccess ? YART_OK : ncclFail(e, "ncclAllReduce");
}

} // extern "C"
