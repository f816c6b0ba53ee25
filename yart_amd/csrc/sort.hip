// sort.hip -- coherence sort of a ray batch (YART_SORT_RAYS).
//
// Key = Morton code of the ray origin quantised to (2^gridBits)^3 cells of the scene
// bounds (default 7 bits per axis: 24-bit key, three radix passes), followed by the 3-bit direction octant; rays that start in the same cell
// and head the same way become neighbours, so the lanes of a wave walk the same
// upper tree levels and touch the same cache lines.  The reference has no
// counterpart: its renderer issues one ray at a time (src/renderer.cpp:59,104).
// Only a permutation is produced; rays and hits stay where the caller put them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "traverse.h"

namespace yart {
namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v)      // 9 bits -> every third bit
{
    v &= 0x1FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256) rayKeys(const float4* __restrict__ rays, uint32_t n,
                                               float lox, float loy, float loz,
                                               float sx, float sy, float sz,
                                               uint32_t octShift, float cellMax,
                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 o = rays[(size_t)i * 2];
    const float4 d = rays[(size_t)i * 2 + 1];
    const float fx = fminf(fmaxf((o.x - lox) * sx, 0.0f), cellMax);
    const float fy = fminf(fmaxf((o.y - loy) * sy, 0.0f), cellMax);
    const float fz = fminf(fmaxf((o.z - loz) * sz, 0.0f), cellMax);
    const uint32_t m = spread3((uint32_t)fx) | (spread3((uint32_t)fy) << 1) | (spread3((uint32_t)fz) << 2);
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    // the octant sits above the lowest `octShift` Morton bits: rays of one (2^-(9-octShift/3))^3
    // cell are grouped by heading before they are ordered by position inside the cell
    keys[i] = ((m >> octShift) << (octShift + 3)) | (oct << octShift) | (m & ((1u << octShift) - 1u));
    ids[i] = i;
}

size_t alignUp(size_t v) { return (v + 255) & ~size_t(255); }

size_t cubTempBytes(uint32_t n)
{
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, v, (int)n, 0, 30, 0);
    return bytes;
}

} // namespace

// Workspace: keys and ray numbers, two buffers each (the radix passes ping-pong between them, so the
// library needs no copies of its own), then hipcub's histograms.
size_t sortWorkspaceBytes(uint32_t n)
{
    return 4 * alignUp((size_t)n * 4) + alignUp(cubTempBytes(n));
}

hipError_t sortRays(const float4* rays, uint32_t n, const float lo[3], const float hi[3], uint32_t octShift,
                    uint32_t gridBits, const uint32_t** perm, void* workspace, size_t workspaceBytes, hipStream_t stream)
{
    *perm = nullptr;
    if (n == 0) return hipSuccess;
    const size_t seg = alignUp((size_t)n * 4);
    if (workspaceBytes < 4 * seg) return hipErrorInvalidValue;
    char* w = static_cast<char*>(workspace);
    hipcub::DoubleBuffer<uint32_t> keys(reinterpret_cast<uint32_t*>(w), reinterpret_cast<uint32_t*>(w + seg));
    hipcub::DoubleBuffer<uint32_t> ids(reinterpret_cast<uint32_t*>(w + 2 * seg), reinterpret_cast<uint32_t*>(w + 3 * seg));
    void* temp = w + 4 * seg;
    size_t tempBytes = workspaceBytes - 4 * seg;
    if (gridBits < 1) gridBits = 1;
    if (gridBits > 9) gridBits = 9;
    const float cells = static_cast<float>(1u << gridBits);
    const int keyBits = static_cast<int>(3 * gridBits + 3);
    if (octShift > 3 * gridBits) octShift = 3 * gridBits;
    float s[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = hi[a] - lo[a];
        s[a] = (ext > 0.0f && ext < 3.0e38f) ? cells / ext : 0.0f;
    }
    hipLaunchKernelGGL(rayKeys, dim3((n + 255) / 256), dim3(256), 0, stream, rays, n,
                       lo[0], lo[1], lo[2], s[0], s[1], s[2], octShift, cells - 1.0f, keys.Current(), ids.Current());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keys, ids, (int)n, 0, keyBits, stream);
    *perm = ids.Current();           // whichever buffer the last pass wrote (known on the host: it depends on the pass count)
    return e;
}

} // namespace yart
