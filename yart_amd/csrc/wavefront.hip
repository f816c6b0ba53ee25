// wavefront.hip -- device-resident wavefront driver and the diffuse path integrator that runs on it
// (include/yart_amd.h, "Wavefront driver").
//
// The reference renders one pixel at a time: trace() (src/renderer.cpp:29-136) loops over samples and
// bounces and calls Scene::intersect / intersectShadow for single rays.  Here every path in flight owns
// a SLOT of a pool kept in HBM; one iteration advances every slot by one bounce with two engine batches
// (closest hit over the slots' rays, any hit over their shadow rays) and three small kernels:
//
//   generate   free slots take the next path numbers from a counter and get their camera ray
//              (src/renderer.cpp:43-50, include/camera.h:24-31)
//   shade      hit -> normal and shading normal (surface_inl.h, the step yart_surface_dev does for other callers),
//              light sample, shadow ray, roulette, cosine-weighted bounce  (:59-102, :113-129)
//   apply      occlusion -> direct light added; finished paths write colour / spp   (:104-110, :132)
//
// A slot without a path carries a ray that misses the scene's bounds, so batches are always `slots`
// rays wide and the traversal kernel retires the fillers at the root.  Results are per PATH (a slab of
// 16 bytes per path, summed per pixel in sample order afterwards), so the image does not depend on
// which slot or iteration traced a path, and equals the synchronous per-pixel loop bit for bit
// (tests/test_wavefront_dev.py against oracle/diffuse_ref.py + the oracle's intersection).
// Arithmetic: +, -, *, /, sqrt only, -ffp-contract=off, so a CPU restatement rounds identically.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"
#include "accel.h"
#include "surface_inl.h"

namespace yart {
namespace {

constexpr uint32_t kNoPath = 0xFFFFFFFFu;
constexpr float kEps = 1e-4f;                      // include/constants.h:7
constexpr float kInvPi = 0.318309886f;
constexpr size_t kSlabBytes = size_t(2) << 30;     // per-path results of one sample slab

struct WfBuffers {
    float4* rays;        // 2 per slot: pending closest-hit ray (YartRay)
    float4* hits;        // 1 per slot (YartHit)
    float4* shadow;      // 2 per slot
    uint8_t* occ;        // 1 per slot
    uint32_t* path;      // slab path number, or kNoPath
    float4* weight;      // xyz = pathWeight (src/renderer.cpp:41), w = bounce number as int bits
    float4* color;       // xyz = colour of the sample so far (:40)
    float4* pend;        // xyz = direct-light term if the shadow ray is clear, w bits: 1 = the path goes on
    uint32_t* counters;  // 4 counters, one 128-byte line each (word 32 k): [0] next path to hand out  [1] paths finished (by a miss
                         // in the shade kernel, or in the apply kernel)  [2] closest rays this iteration  [3] shadow rays this iteration
};
constexpr uint32_t kCtr = 32;        // words between counters
constexpr uint32_t kMaxBlocks = 2048;  // kernels walk the slots with a grid stride: one atomic per block and counter, not per wave

struct DiffuseParams {
    float pos[3], dir[3], right[3], up[3];
    float light[3], intensity[3], albedo[3];
    int32_t width, height, maxDepth;
    float invSpp;
    uint32_t seed;
    const uint32_t* pixels;
    uint32_t slabFirstSample, slabSamples, slabPaths;
};

__host__ __device__ inline uint32_t pcg(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// random number `dim` of a path: a hash of (seed, pixel number, sample, dim) in [0, 1)
__device__ inline float rnd(uint32_t key, uint32_t dim)
{
    return (float)(pcg(key + dim * 0x632BE5ABu) >> 8) * 5.9604644775390625e-8f;      // 2^-24
}

__device__ inline uint32_t pathKey(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    return pcg(pcg(pcg(seed ^ 0x9E3779B9u) + pixel) + sample);
}

__device__ inline void deadRay(float4* rays, uint32_t i)
{
    // starts far outside any scene and points away from it: misses the root's boxes (a regular ray, traverse.h)
    rays[(size_t)i * 2 + 0] = make_float4(1e30f, 1e30f, 1e30f, 0.0f);
    rays[(size_t)i * 2 + 1] = make_float4(0.57735026f, 0.57735026f, 0.57735026f, 0.0f);
}

// adds the block's count of `n` to a counter with one atomic (call from every thread of the block)
__device__ inline void blockAdd(uint32_t* counter, uint32_t n, uint32_t* lds)
{
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
        if (total) atomicAdd(counter, total);
    }
}

__device__ inline void startPath(const WfBuffers& B, const DiffuseParams& J, uint32_t i, uint32_t p)
{
    if (p >= J.slabPaths) { deadRay(B.rays, i); return; }      // the slab's paths have all been handed out
    const uint32_t pixelIdx = p / J.slabSamples;
    const uint32_t sample = J.slabFirstSample + p % J.slabSamples;
    const uint32_t pixel = J.pixels ? J.pixels[pixelIdx] : pixelIdx;
    const uint32_t key = pathKey(J.seed, pixel, sample);
    const float x = (float)(pixel % (uint32_t)J.width) + rnd(key, 0) - 0.5f;     // src/renderer.cpp:47-50
    const float y = (float)(pixel / (uint32_t)J.width) + rnd(key, 1) - 0.5f;
    const float a = (x + 0.5f) / (float)J.width - 0.5f;                           // include/camera.h:26-28
    const float b = -(y + 0.5f) / (float)J.height + 0.5f;
    const float dx = J.right[0] * a + J.up[0] * b + J.dir[0];
    const float dy = J.right[1] * a + J.up[1] * b + J.dir[1];
    const float dz = J.right[2] * a + J.up[2] * b + J.dir[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    B.rays[(size_t)i * 2 + 0] = make_float4(J.pos[0], J.pos[1], J.pos[2], 0.0f);
    B.rays[(size_t)i * 2 + 1] = make_float4(dx * inv, dy * inv, dz * inv, __builtin_inff());
    B.path[i] = p;
    B.weight[i] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0));
    B.color[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Every block owns a contiguous range of slots: it counts the free ones, reserves that many path numbers
// with ONE atomic, and hands them to its free slots in slot order.
__global__ void __launch_bounds__(256) wfGenerate(WfBuffers B, DiffuseParams J, uint32_t nSlots, uint32_t perBlock)
{
    __shared__ uint32_t lds[8];
    const uint32_t begin = blockIdx.x * perBlock;
    const uint32_t end = begin + perBlock < nSlots ? begin + perBlock : nSlots;
    uint32_t mine = 0;
    for (uint32_t i = begin + threadIdx.x; i < end; i += 256u) mine += B.path[i] == kNoPath ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        // Nothing is reserved once the slab's paths are all handed out: the counter would otherwise grow by the free slots
        // of every drain iteration and, with many slots and a deep path limit, wrap around 2^32 and hand paths out twice.
        const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
        const uint32_t seen = __hip_atomic_load(B.counters + 0 * kCtr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[4] = (total && seen < J.slabPaths) ? atomicAdd(B.counters + 0 * kCtr, total) : J.slabPaths;
    }
    __syncthreads();
    uint32_t next = lds[4];                                  // first path number of the chunk being handed out
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t c = begin; c < end; c += 256u) {
        const uint32_t i = c + threadIdx.x;
        const bool want = i < end && B.path[i] == kNoPath;
        const uint64_t m = __ballot(want);
        __syncthreads();                                     // lds[0..3] of the previous round have been read
        if (lane == 0) lds[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t n = lds[k]; all += n; if (k < wave) before += n; }
        if (want) startPath(B, J, i, next + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
        next += all;
    }
}

// Hit post-processing (row K) is folded in: the surface point comes straight from the hit record and the accel's surface
// rows (surface_inl.h) instead of a YartSurface record written by one kernel and read by the next (96 bytes per slot
// and iteration less; round 3).
__global__ void __launch_bounds__(256) wfShade(WfBuffers B, DiffuseParams J, uint32_t nSlots, float4* slab,
                                               const float4* __restrict__ surfRows, const float4* __restrict__ surfSpheres)
{
    __shared__ uint32_t lds[4];
    uint32_t nLive = 0, nShadow = 0, nMissed = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nSlots; i += gridDim.x * 256u) {
    const bool live = B.path[i] != kNoPath;
    if (live) {
        ++nLive;
        const uint32_t p = B.path[i];
        const float4 hit = B.hits[i];                          // t, u, v, prim
        const int prim = __float_as_int(hit.w);
        if (prim == -1) {
            // src/renderer.cpp:59-60: nothing hit, the sample is what it has gathered so far
            const float4 c = B.color[i];
            slab[p] = make_float4(c.x * J.invSpp, c.y * J.invSpp, c.z * J.invSpp, 0.0f);
            B.path[i] = kNoPath;
            ++nMissed;
            deadRay(B.rays, i);
            deadRay(B.shadow, i);
        } else {
            ++nShadow;
            const float4 o = B.rays[(size_t)i * 2 + 0], d = B.rays[(size_t)i * 2 + 1];
            float4 w = B.weight[i];
            const int bounce = __float_as_int(w.w);
            const uint32_t pixelIdx = p / J.slabSamples;
            const uint32_t pixel = J.pixels ? J.pixels[pixelIdx] : pixelIdx;
            const uint32_t key = pathKey(J.seed, pixel, J.slabFirstSample + p % J.slabSamples);
            const uint32_t dim = 2u + (uint32_t)bounce * 32u;
            const SurfacePoint sp = surfaceOfHit(hit, surfRows, surfSpheres, B.rays + (size_t)i * 2);
            const float4 s0 = make_float4(sp.nx, sp.ny, sp.nz, hit.x);           // normal, t
            const float4 s1 = make_float4(sp.sx, sp.sy, sp.sz, hit.y);           // shading normal, u
            const float t = s0.w;
            const float Px = o.x + d.x * t, Py = o.y + d.y * t, Pz = o.z + d.z * t;            // :72
            const bool front = s0.x * d.x + s0.y * d.y + s0.z * d.z < 0.0f;                      // :74
            // :73-74: nl is the SHADING normal (not normalised, include/triangle.h:166-171), turned by the sign of the
            // geometric one.  A scene given without vertex normals (all zero) is flat shaded: the geometric normal.
            const bool flat = s1.x == 0.0f && s1.y == 0.0f && s1.z == 0.0f;
            const float snx = flat ? s0.x : s1.x, sny = flat ? s0.y : s1.y, snz = flat ? s0.z : s1.z;
            const float nlx = front ? snx : -snx, nly = front ? sny : -sny, nlz = front ? snz : -snz;
            // light sample (include/light.h:41-52), shadow ray (:101-102; the point light's eps is 0)
            const float tx = J.light[0] - Px, ty = J.light[1] - Py, tz = J.light[2] - Pz;
            const float dist2 = tx * tx + ty * ty + tz * tz;
            const float dist = sqrtf(dist2);
            const float invD = 1.0f / dist;
            const float wix = tx * invD, wiy = ty * invD, wiz = tz * invD;
            B.shadow[(size_t)i * 2 + 0] = make_float4(Px + wix * kEps, Py + wiy * kEps, Pz + wiz * kEps, 0.0f);
            B.shadow[(size_t)i * 2 + 1] = make_float4(wix, wiy, wiz, dist);
            const float cosL = fabsf(nlx * wix + nly * wiy + nlz * wiz);
            float4 pend;
            pend.x = w.x * (J.albedo[0] * kInvPi) * (J.intensity[0] / dist2) * cosL;            // :106-108
            pend.y = w.y * (J.albedo[1] * kInvPi) * (J.intensity[1] / dist2) * cosL;
            pend.z = w.z * (J.albedo[2] * kInvPi) * (J.intensity[2] / dist2) * cosL;
            // roulette (:113-117), then the bounce (:118-129)
            const float q = 0.2126f * w.x + 0.7152f * w.y + 0.0722f * w.z;                       // include/spectrum.h:41-44
            bool goesOn = !(rnd(key, dim + 3u) > q) && q > 0.0f && bounce + 1 < J.maxDepth;
            if (goesOn) {
                w.x = w.x / q; w.y = w.y / q; w.z = w.z / q;
                // cosine-weighted direction: a point of the unit disk by rejection (no trigonometry: only
                // operations that round the same everywhere), lifted to the hemisphere
                float lx = 0.0f, ly = 0.0f;
                for (uint32_t k = 0; k < 8u; ++k) {
                    const float ax = 2.0f * rnd(key, dim + 4u + 2u * k) - 1.0f;
                    const float ay = 2.0f * rnd(key, dim + 5u + 2u * k) - 1.0f;
                    if (ax * ax + ay * ay <= 1.0f) { lx = ax; ly = ay; break; }
                }
                float lz = sqrtf(fmaxf(0.0f, 1.0f - lx * lx - ly * ly));
                // include/frame.h:18-32
                const float invN = 1.0f / sqrtf(nlx * nlx + nly * nly + nlz * nlz);
                const float nx = nlx * invN, ny = nly * invN, nz = nlz * invN;
                float fx, fy, fz;
                if (fabsf(nx) > fabsf(ny)) {
                    const float il = 1.0f / sqrtf(nx * nx + nz * nz);
                    fx = -nz * il; fy = 0.0f; fz = nx * il;
                } else {
                    const float il = 1.0f / sqrtf(ny * ny + nz * nz);
                    fx = 0.0f; fy = nz * il; fz = -ny * il;
                }
                const float sx = ny * fz - nz * fy, sy = nz * fx - nx * fz, sz = nx * fy - ny * fx;   // s = cross(n, t)
                // Lambertian::sample (src/bsdf.cpp:86-95): the sampled direction lies on wo's side of the frame (wo.z =
                // dot(-dir, n) is negative only where the shading normal disagrees with the geometric one); a direction
                // within EPS of the horizon ends the path (:92-93 -> src/renderer.cpp:121-122)
                const float pdfL = lz * kInvPi;
                if (-(d.x * nx + d.y * ny + d.z * nz) < 0.0f) lz = -lz;
                const float ddx = lx * sx + ly * fx + lz * nx;                                         // toWorld
                const float ddy = lx * sy + ly * fy + lz * ny;
                const float ddz = lx * sz + ly * fz + lz * nz;
                if (fabsf(lz) < kEps) goesOn = false;
                else {
                    // src/renderer.cpp:128: pathWeight * refl * |dot(dir, nl)| / pdf, nl as it is (not normalised)
                    const float cosW = fabsf(ddx * nlx + ddy * nly + ddz * nlz);
                    w.x = w.x * (J.albedo[0] * kInvPi) * cosW / pdfL;
                    w.y = w.y * (J.albedo[1] * kInvPi) * cosW / pdfL;
                    w.z = w.z * (J.albedo[2] * kInvPi) * cosW / pdfL;
                    B.rays[(size_t)i * 2 + 0] = make_float4(Px + ddx * kEps, Py + ddy * kEps, Pz + ddz * kEps, 0.0f);
                    B.rays[(size_t)i * 2 + 1] = make_float4(ddx, ddy, ddz, __builtin_inff());
                }
            }
            if (!goesOn) deadRay(B.rays, i);
            w.w = __int_as_float(bounce + 1);
            B.weight[i] = w;
            pend.w = __int_as_float(goesOn ? 1 : 0);
            B.pend[i] = pend;
        }
    } else {
        deadRay(B.shadow, i);
    }
    }
    blockAdd(B.counters + 2 * kCtr, nLive, lds);
    blockAdd(B.counters + 3 * kCtr, nShadow, lds);
    blockAdd(B.counters + 1 * kCtr, nMissed, lds);
}

__global__ void __launch_bounds__(256) wfApply(WfBuffers B, DiffuseParams J, uint32_t nSlots, float4* slab)
{
    __shared__ uint32_t lds[4];
    uint32_t nFinished = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nSlots; i += gridDim.x * 256u) {
        const uint32_t p = B.path[i];
        if (p == kNoPath) continue;
        const float4 pend = B.pend[i];
        float4 c = B.color[i];
        if (B.occ[i] == 0) { c.x = c.x + pend.x; c.y = c.y + pend.y; c.z = c.z + pend.z; }    // :104-109
        if (__float_as_int(pend.w) == 0) {
            slab[p] = make_float4(c.x * J.invSpp, c.y * J.invSpp, c.z * J.invSpp, 0.0f);      // :132
            B.path[i] = kNoPath;
            ++nFinished;
        } else B.color[i] = c;
    }
    blockAdd(B.counters + 1 * kCtr, nFinished, lds);
}

__global__ void __launch_bounds__(256) wfReduce(const float4* slab, uint32_t nPixels, uint32_t slabSamples, float* rgb, int first)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nPixels) return;
    float r = first ? 0.0f : rgb[(size_t)i * 3 + 0], g = first ? 0.0f : rgb[(size_t)i * 3 + 1], b = first ? 0.0f : rgb[(size_t)i * 3 + 2];
    for (uint32_t s = 0; s < slabSamples; ++s) {           // finalColor = finalColor + color * invMaxIter, in sample order
        const float4 c = slab[(size_t)i * slabSamples + s];
        r = r + c.x; g = g + c.y; b = b + c.z;
    }
    rgb[(size_t)i * 3 + 0] = r; rgb[(size_t)i * 3 + 1] = g; rgb[(size_t)i * 3 + 2] = b;
}

} // namespace
} // namespace yart

struct YartWavefront {
    YartAccel* accel = nullptr;
    size_t slots = 0;
    void* block = nullptr;          // one allocation for every per-slot array
    yart::WfBuffers buf{};
    float4* slab = nullptr;
    size_t slabPaths = 0;
    std::vector<hipEvent_t> events;
    uint32_t* hostCounters = nullptr;   // pinned: two copies of the counter block, read back while the next iteration runs
};

namespace {

struct DevGuard {
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int hipFail(hipError_t e, const char* what)
{
    return yart::failStatus(YART_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

size_t alignUp(size_t v) { return (v + 255) & ~size_t(255); }

} // namespace

extern "C" {

int yart_wavefront_create(YartAccel* accel, size_t slots, YartWavefront** out)
{
    if (!accel || !out) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_create: null argument");
    *out = nullptr;
    if (slots < 1 || slots > (size_t(1) << 27)) return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_create: slots out of range (1 .. 2^27)");
    if ((accel->numTriangles && !accel->dSurfRows) || (accel->numSpheres && !accel->dSurfSpheres))
        return yart::failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_create: call yart_accel_set_surface_data first");
    DevGuard guard(accel->device);
    if (!guard.ok) return yart::failStatus(YART_ERR_NO_DEVICE, "hipSetDevice failed");
    YartWavefront* w = new (std::nothrow) YartWavefront();
    if (!w) return yart::failStatus(YART_ERR_OUT_OF_MEMORY, "host allocation failed");
    w->accel = accel;
    w->slots = slots;
    const size_t n = slots;
    const size_t sizes[9] = { n * 32, n * 16, n * 32, n, n * 4, n * 16, n * 16, n * 16, 512 };
    size_t total = 0;
    for (size_t s : sizes) total += alignUp(s);
    hipError_t e = hipMalloc(&w->block, total);
    if (e != hipSuccess) { delete w; return yart::failStatus(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(wavefront pool): ") + hipGetErrorString(e)); }
    char* p = static_cast<char*>(w->block);
    auto take = [&](size_t bytes) { char* q = p; p += alignUp(bytes); return q; };
    w->buf.rays = reinterpret_cast<float4*>(take(sizes[0]));
    w->buf.hits = reinterpret_cast<float4*>(take(sizes[1]));
    w->buf.shadow = reinterpret_cast<float4*>(take(sizes[2]));
    w->buf.occ = reinterpret_cast<uint8_t*>(take(sizes[3]));
    w->buf.path = reinterpret_cast<uint32_t*>(take(sizes[4]));
    w->buf.weight = reinterpret_cast<float4*>(take(sizes[5]));
    w->buf.color = reinterpret_cast<float4*>(take(sizes[6]));
    w->buf.pend = reinterpret_cast<float4*>(take(sizes[7]));
    w->buf.counters = reinterpret_cast<uint32_t*>(take(sizes[8]));
    *out = w;
    return YART_OK;
}

void yart_wavefront_destroy(YartWavefront* w)
{
    if (!w) return;
    {
        DevGuard guard(w->accel->device);
        (void)hipFree(w->block);
        (void)hipFree(w->slab);
        for (hipEvent_t ev : w->events) (void)hipEventDestroy(ev);
        (void)hipHostFree(w->hostCounters);
    }
    delete w;
}

int yart_wavefront_render_diffuse(YartWavefront* w, const YartDiffuseJob* job, float* d_rgb, YartWavefrontStats* stats,
                                  YartWavefrontIter* iters, size_t max_iters, void* streamPtr)
{
    using namespace yart;
    if (!w || !job || !d_rgb) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_render_diffuse: null argument");
    if (job->width <= 0 || job->height <= 0 || job->spp <= 0 || job->max_depth <= 0 ||
        (int64_t)job->width * job->height > (int64_t(1) << 30))
        return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_render_diffuse: bad image size, spp or depth");
    const size_t nPixels = job->d_pixels ? job->n_pixels : (size_t)job->width * job->height;
    if (nPixels == 0) return YART_OK;
    if (nPixels > (size_t(1) << 30)) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_render_diffuse: too many pixels");
    hipStream_t stream = static_cast<hipStream_t>(streamPtr);
    DevGuard guard(w->accel->device);
    if (!guard.ok) return failStatus(YART_ERR_NO_DEVICE, "hipSetDevice failed");

    // samples are rendered in slabs so that the per-path results stay within kSlabBytes
    uint32_t slabSamples = (uint32_t)std::min<size_t>((size_t)job->spp, std::max<size_t>(1, kSlabBytes / 16 / nPixels));
    const size_t slabPathsMax = nPixels * slabSamples;
    if (slabPathsMax >= 0xFFFFFFF0ull) return failStatus(YART_ERR_INVALID_ARGUMENT, "yart_wavefront_render_diffuse: slab too large");
    if (slabPathsMax > w->slabPaths) {
        (void)hipFree(w->slab);
        w->slab = nullptr;
        w->slabPaths = 0;
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&w->slab), slabPathsMax * 16);
        if (e != hipSuccess) return failStatus(YART_ERR_OUT_OF_MEMORY, std::string("hipMalloc(sample slab): ") + hipGetErrorString(e));
        w->slabPaths = slabPathsMax;
    }
    if (!w->hostCounters) {
        const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&w->hostCounters), 2 * 4 * kCtr * sizeof(uint32_t));
        if (e != hipSuccess) { w->hostCounters = nullptr; return hipFail(e, "hipHostMalloc"); }
    }
    while (w->events.size() < 16) {
        hipEvent_t ev;
        const hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) return hipFail(e, "hipEventCreate");
        w->events.push_back(ev);
    }

    DiffuseParams J;
    std::memset(&J, 0, sizeof J);
    for (int c = 0; c < 3; ++c) {
        J.pos[c] = job->cam_pos[c]; J.dir[c] = job->cam_dir[c]; J.right[c] = job->cam_right[c]; J.up[c] = job->cam_up[c];
        J.light[c] = job->light_pos[c]; J.intensity[c] = job->light_intensity[c]; J.albedo[c] = job->albedo[c];
    }
    J.width = job->width; J.height = job->height; J.maxDepth = job->max_depth;
    J.invSpp = 1.0f / (float)job->spp;
    J.seed = job->seed;
    J.pixels = job->d_pixels;

    YartWavefrontStats st;
    std::memset(&st, 0, sizeof st);
    const auto wall0 = std::chrono::steady_clock::now();
    const uint32_t nSlots = (uint32_t)w->slots;
    const uint32_t grid = std::min<uint32_t>((nSlots + 255u) / 256u, kMaxBlocks);
    const uint32_t perBlock = ((nSlots + grid - 1u) / grid + 255u) / 256u * 256u;      // wfGenerate: contiguous slots per block
    const YartRay* dRays = reinterpret_cast<const YartRay*>(w->buf.rays);
    const YartRay* dShadow = reinterpret_cast<const YartRay*>(w->buf.shadow);
    YartHit* dHits = reinterpret_cast<YartHit*>(w->buf.hits);
    hipError_t e = hipSuccess;
    int status = YART_OK;
    size_t iterNo = 0;

    for (uint32_t firstSample = 0; firstSample < (uint32_t)job->spp && status == YART_OK; firstSample += slabSamples) {
        const uint32_t ss = std::min<uint32_t>(slabSamples, (uint32_t)job->spp - firstSample);
        J.slabFirstSample = firstSample;
        J.slabSamples = ss;
        J.slabPaths = (uint32_t)(nPixels * ss);
        if ((e = hipMemsetAsync(w->buf.counters, 0, 512, stream)) != hipSuccess) return hipFail(e, "hipMemsetAsync");
        if ((e = hipMemsetAsync(w->buf.path, 0xFF, (size_t)nSlots * 4, stream)) != hipSuccess) return hipFail(e, "hipMemsetAsync");
        uint32_t finished = 0, handedOut = 0;
        // every iteration ends at least one bounce of every live path, and a path has at most max_depth
        const size_t guardIters = ((size_t)J.slabPaths / nSlots + 2) * ((size_t)job->max_depth + 1) + 8;
        // One iteration's launches; its counters are copied to pinned copy `k` and event 7 of set `k` marks the end.
        auto enqueue = [&](int k, bool generate) -> int {
            hipEvent_t* ev = w->events.data() + 8 * k;
            (void)hipEventRecord(ev[0], stream);
            if (generate) {
                hipLaunchKernelGGL(wfGenerate, dim3(grid), dim3(256), 0, stream, w->buf, J, nSlots, perBlock);
                if ((e = hipGetLastError()) != hipSuccess) return hipFail(e, "wfGenerate");
            }
            if ((e = hipMemsetAsync(w->buf.counters + 2 * kCtr, 0, 2 * kCtr * 4, stream)) != hipSuccess) return hipFail(e, "hipMemsetAsync");
            (void)hipEventRecord(ev[1], stream);
            int stt = yart_intersect_closest_dev(w->accel, dRays, nSlots, dHits, job->flags, stream);
            if (stt != YART_OK) return stt;
            (void)hipEventRecord(ev[2], stream);           // (the surface step is part of the shade kernel)
            hipLaunchKernelGGL(wfShade, dim3(grid), dim3(256), 0, stream, w->buf, J, nSlots, w->slab,
                               static_cast<const float4*>(w->accel->dSurfRows), static_cast<const float4*>(w->accel->dSurfSpheres));
            if ((e = hipGetLastError()) != hipSuccess) return hipFail(e, "wfShade");
            (void)hipEventRecord(ev[3], stream);
            stt = yart_intersect_any_dev(w->accel, dShadow, nSlots, w->buf.occ, job->flags, stream);
            if (stt != YART_OK) return stt;
            (void)hipEventRecord(ev[4], stream);
            hipLaunchKernelGGL(wfApply, dim3(grid), dim3(256), 0, stream, w->buf, J, nSlots, w->slab);
            if ((e = hipGetLastError()) != hipSuccess) return hipFail(e, "wfApply");
            (void)hipEventRecord(ev[5], stream);
            if ((e = hipMemcpyAsync(w->hostCounters + (size_t)k * 4 * kCtr, w->buf.counters, 4 * kCtr * sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
                return hipFail(e, "hipMemcpyAsync");
            (void)hipEventRecord(ev[7], stream);
            return YART_OK;
        };
        // waits for iteration copy `k` and books it
        auto collect = [&](int k) -> int {
            hipEvent_t* ev = w->events.data() + 8 * k;
            if ((e = hipEventSynchronize(ev[7])) != hipSuccess) return hipFail(e, "hipEventSynchronize");
            const uint32_t* raw = w->hostCounters + (size_t)k * 4 * kCtr;
            const uint32_t c[4] = { raw[0], raw[kCtr], raw[2 * kCtr], raw[3 * kCtr] };
            handedOut = std::min<uint32_t>(c[0], J.slabPaths);
            const uint32_t live = c[2], senders = c[3];
            st.closest_rays += live;
            st.shadow_rays += senders;
            finished = c[1];
            float ms[5] = { 0, 0, 0, 0, 0 };
            for (int q = 0; q < 5; ++q) (void)hipEventElapsedTime(&ms[q], ev[q], ev[q + 1]);
            st.ms_closest += ms[1]; st.ms_shadow += ms[3]; st.ms_shade += ms[0] + ms[2] + ms[4];
            if (iters && iterNo < max_iters) {
                YartWavefrontIter& r = iters[iterNo];
                r.closest_rays = live; r.shadow_rays = senders;
                r.ms_closest = ms[1]; r.ms_surface = 0.0f; r.ms_shadow = ms[3]; r.ms_shade = ms[0] + ms[2] + ms[4];
            }
            ++iterNo;
            ++st.iterations;
            return YART_OK;
        };
        if (job->regenerate) {
            // Free slots take new paths in every iteration, so nothing an iteration does depends on the host having seen
            // the counters of the one before: iteration it + 1 is enqueued before the host waits for iteration it, and the
            // GPU never idles over the read-back.  The iteration enqueued after the last paths ended finds every slot
            // free and no path left: its rays are the fillers, it changes nothing and is not booked.
            if ((status = enqueue(0, true)) != YART_OK) return status;
            for (size_t it = 0;; ++it) {
                if (it > guardIters) return failStatus(YART_ERR_HIP, "yart_wavefront_render_diffuse: paths did not finish (internal error)");
                if ((status = enqueue((int)((it + 1) & 1), true)) != YART_OK) return status;
                if ((status = collect((int)(it & 1))) != YART_OK) return status;
                if (finished >= J.slabPaths) {
                    if ((e = hipEventSynchronize(w->events[8 * ((it + 1) & 1) + 7])) != hipSuccess) return hipFail(e, "hipEventSynchronize");
                    break;
                }
            }
        } else {
            for (size_t it = 0; finished < J.slabPaths; ++it) {
                if (it > guardIters) return failStatus(YART_ERR_HIP, "yart_wavefront_render_diffuse: paths did not finish (internal error)");
                if ((status = enqueue(0, finished == handedOut)) != YART_OK) return status;      // pool mode: only when every slot is free
                if ((status = collect(0)) != YART_OK) return status;
            }
        }
        st.paths += J.slabPaths;
        hipLaunchKernelGGL(wfReduce, dim3(((uint32_t)nPixels + 255u) / 256u), dim3(256), 0, stream, w->slab, (uint32_t)nPixels, ss, d_rgb,
                           firstSample == 0 ? 1 : 0);
        if ((e = hipGetLastError()) != hipSuccess) return hipFail(e, "wfReduce");
    }
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return hipFail(e, "hipStreamSynchronize");
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = st;
    return YART_OK;
}

} // extern "C"
