// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// ref_shim.cpp: thin C-ABI wrapper that compiles the REAL reference
// (elaic/yart, read-only at /root/reference) from the sources where they lie
// and exposes its Scene::intersect / Scene::intersectShadow / BvhAccel to
// Python (ctypes).  Built only in the authoring container by oracle/Makefile
// into oracle/_ref/libyart_ref.so (git-ignored; travels to the GPU box as a
// built binary).  Uses:
//   * oracle/make_golden.py  -> golden vectors under tests/golden/
//   * tests/                 -> pin the CPU restatement (oracle/cpu_ref.cpp)
//   * bench.py cpu_baseline  -> kind "reference"
//
// No reference source is copied here: this file only #includes the reference
// headers/sources by path and calls their public entry points.  Private
// members are reached with the usual test-only `#define private public`
// (needed because Scene::preprocess() overflows in loadTriaccel8 when
// triangleCount % 8 == 0 -- SURVEY.md 8c -- so the BvhAccel is constructed
// directly, exactly as preprocess() does at include/scene.h:73).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <immintrin.h>

#define private public
#include "scene.h"
#include "camera.h"
#include "tiny_obj_loader.h"
// FlattenedBvhNode is only defined inside the .cpp (src/bvhaccel.cpp:64), so
// this TU *is* the bvhaccel translation unit of the reference build.
#include "../src/bvhaccel.cpp"
#undef private
// The reference's renderer: trace() (src/renderer.cpp:29-136) is driven directly, pixel by
// pixel, for the caller-harness fixtures.  Renderer::render / TileTask come
// along and need the reference's scheduler at link time (src/scheduler.cpp, built beside this).
#include "../src/renderer.cpp"
// product header under test: the batched driver the restated integrator below plugs into
#include "../yart_amd/cxx/wavefront.hpp"

namespace {

struct RefScene {
    std::unique_ptr<Scene> scene;
    // per mesh: the mesh's Bsdf* -> mesh index (each mesh gets its own object)
    std::unordered_map<const Bsdf*, int32_t> bsdfToMesh;
    // per mesh: positions (in leaf order) of that mesh's TriAccel records
    std::vector<std::vector<int32_t>> meshLeafSlots;
    // per sphere: the fake AreaLight* tag identifies the sphere
    size_t numSpheres = 0;
    size_t numTriangles = 0;
};

struct NodeOut {            // normalised 32-byte dump of FlattenedBvhNode
    float bounds[6];
    uint32_t offset;        // childOffset | triangleOffset
    uint8_t numTriangles;   // 0 for interior nodes (uninitialised in the reference)
    uint8_t splitAxis;      // 0,1,2 ; 3 = leaf
    uint8_t pad[2];
};
static_assert(sizeof(NodeOut) == 32, "NodeOut must be 32 bytes");

struct HitOut {
    int32_t hit;            // return value of Scene::intersect
    float t, u, v;
    int32_t prim;           // >=0: leaf-order TriAccel index; -(i+2): sphere i; -1 miss
    int32_t ambiguous;      // >1 if several triangles of the mesh reproduce (t,u,v) bit-for-bit
    float normal[3];
    float shadingNormal[3];
};

// A tag that is never dereferenced: Scene::intersect only copies getLight()
// into RayHitInfo::areaLight (include/sphere.h:83).
AreaLight* sphereTag(size_t i) { return reinterpret_cast<AreaLight*>(uintptr_t(0x1000 + 16 * i)); }

RefScene* buildScene(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                     const float* spheres, size_t nSpheres, int genNormals)
{
    auto rs = new RefScene();
    std::vector<TriangleMesh> meshes;
    // group triangles by meshId (must be contiguous, ascending from 0)
    size_t i = 0;
    while (i < nTris) {
        int32_t mid = tris[i * 5 + 3];
        size_t j = i;
        int32_t vmin = std::numeric_limits<int32_t>::max(), vmax = -1;
        while (j < nTris && tris[j * 5 + 3] == mid) {
            for (int c = 0; c < 3; ++c) {
                vmin = std::min(vmin, tris[j * 5 + c]);
                vmax = std::max(vmax, tris[j * 5 + c]);
            }
            ++j;
        }
        if (mid != (int32_t)meshes.size() || vmax >= (int64_t)nVerts) {
            fprintf(stderr, "ref_shim: mesh ids must be contiguous from 0 / bad vertex index\n");
            delete rs;
            return nullptr;
        }
        std::vector<Vector3f> v;
        v.reserve(vmax - vmin + 1);
        for (int32_t k = vmin; k <= vmax; ++k)
            v.emplace_back(verts[k * 3 + 0], verts[k * 3 + 1], verts[k * 3 + 2]);
        std::vector<Triangle> t;
        t.reserve(j - i);
        for (size_t k = i; k < j; ++k) {
            if (tris[k * 5 + 4] != (int32_t)(k - i)) {
                fprintf(stderr, "ref_shim: triIds must count from 0 inside each mesh\n");
                delete rs;
                return nullptr;
            }
            t.emplace_back(tris[k * 5 + 0] - vmin, tris[k * 5 + 1] - vmin, tris[k * 5 + 2] - vmin);
        }
        auto bsdf = std::make_shared<Lambertian>(Spectrum(0.5f, 0.5f, 0.5f));
        if (genNormals) {
            meshes.emplace_back(v, t, bsdf);     // include/triangle.h:76 (generates vertex normals)
        } else {
            std::vector<Vector3f> n(v.size(), Vector3f(0.0f, 0.0f, 1.0f));
            meshes.emplace_back(v, n, t, bsdf);  // include/triangle.h:113
        }
        i = j;
    }

    std::vector<std::shared_ptr<Shape>> shapes;
    for (size_t s = 0; s < nSpheres; ++s) {
        auto sp = std::make_shared<Sphere>(spheres[s * 4 + 3],
            Vector3f(spheres[s * 4 + 0], spheres[s * 4 + 1], spheres[s * 4 + 2]),
            Spectrum(0.5f, 0.5f, 0.5f), Bxdf::Diff);
        sp->setLight(sphereTag(s));
        shapes.push_back(sp);
    }
    std::vector<std::shared_ptr<Light>> lights;

    rs->scene = std::make_unique<Scene>(meshes, shapes, lights);
    // what Scene::preprocess() does at include/scene.h:73, minus the TriAccel8 path
    rs->scene->accel_ = std::make_shared<BvhAccel>(*rs->scene);
    rs->numSpheres = nSpheres;
    rs->numTriangles = nTris;

    const auto& ms = rs->scene->meshes_;
    rs->meshLeafSlots.resize(ms.size());
    for (size_t m = 0; m < ms.size(); ++m) rs->bsdfToMesh[ms[m].getBsdf()] = (int32_t)m;
    const TriAccel* ta = rs->scene->accel_->triangles_;
    for (size_t k = 0; k < nTris; ++k) rs->meshLeafSlots[ta[k].meshIdx].push_back((int32_t)k);
    return rs;
}

template <class F>
void parallelFor(size_t n, int nthreads, F&& f)
{
    if (nthreads <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
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

template <class F>
void parallelFor1(size_t n, int nthreads, F&& f)
{
    if (nthreads <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto& t : th) t.join();
}

Ray makeRay(const float* r)
{
    Ray ray(Vector3f(r[0], r[1], r[2]), Vector3f(r[4], r[5], r[6]));
    ray.minT = r[3];
    ray.maxT = r[7];
    return ray;
}

// ---- caller harness: the reference's own trace() as the driver ------------
//
// Scene::intersect and Scene::intersectShadow call every Shape's virtual intersect() before the
// BVH (include/scene.h:142-144, :176-180).  Two Shape subclasses use that hook, without touching
// any reference source:
//   SpyShape    returns "no hit" and records the ray -> the exact ray stream trace() issues
//   EngineShape forwards the query to an external intersection engine through two C function
//               pointers (the C-ABI of include/yart_amd.h) and fills RayHitInfo the way the
//               reference does after a hit (src/bvhaccel.cpp:283-292, include/sphere.h:79-83)
//               -> the reference's renderer, BSDFs, lights and RNG drive the engine as a drop-in.
struct RayLog { std::vector<float> closest, shadow; };
thread_local RayLog* tlsLog = nullptr;

void logRay(const Ray& ray)
{
    // trace() builds closest-hit rays with maxT = +inf (include/vector.h:244-249) and shadow rays
    // with a finite maxT (src/renderer.cpp:101-102): that is how the two kinds are told apart.
    auto& v = std::isinf(ray.maxT) ? tlsLog->closest : tlsLog->shadow;
    const float r[8] = { ray.orig.x, ray.orig.y, ray.orig.z, ray.minT, ray.dir.x, ray.dir.y, ray.dir.z, ray.maxT };
    v.insert(v.end(), r, r + 8);
}

class SpyShape : public Shape {
public:
    bool intersect(const Ray& ray, RayHitInfo* const) const override { if (tlsLog) logRay(ray); return false; }
    Vector3f sample(float, float, float* pdf) const override { *pdf = 0.0f; return Vector3f(0.0f, 0.0f, 0.0f); }
    float area() const override { return 0.0f; }
};

struct EngineHit { float t, u, v; int32_t prim; };          // = YartHit

// What the reference derives after a hit (src/bvhaccel.cpp:283-292, include/sphere.h:79-83), from the
// engine's (t, u, v, prim) with the reference's own mesh / sphere accessors.
void fillHitInfo(const Scene& geometry, const TriAccel* leafOrder, const Ray& ray, const EngineHit& h, RayHitInfo* hitInfo)
{
    hitInfo->t = h.t;
    hitInfo->u = h.u;
    hitInfo->v = h.v;
    if (h.prim >= 0) {
        const TriAccel& ta = leafOrder[h.prim];
        const auto& mesh = geometry.meshes_[ta.meshIdx];
        hitInfo->normal = mesh.getNormal(ta.triIdx);
        hitInfo->shadingNormal = mesh.getShadingNormal(ta.triIdx, h.u, h.v);
        hitInfo->bsdf = mesh.getBsdf();
        hitInfo->areaLight = nullptr;
    } else {
        const auto* sp = static_cast<const Sphere*>(geometry.shapes_[-(h.prim + 2)].get());
        hitInfo->normal = normal(ray.orig + ray.dir * hitInfo->t - sp->position_);
        hitInfo->shadingNormal = hitInfo->normal;
        hitInfo->bsdf = sp->bsdf_.get();
        hitInfo->areaLight = sp->getLight();
    }
}
typedef int (*EngineClosestFn)(void* accel, const void* rays, size_t n, void* hits, unsigned flags);
typedef int (*EngineAnyFn)(void* accel, const void* rays, size_t n, uint8_t* occluded, unsigned flags);

class EngineShape : public Shape {
public:
    EngineShape(const Scene& geometry, const TriAccel* leafOrder, EngineClosestFn c, EngineAnyFn a, void* accel,
                unsigned flags)
        : geometry_(geometry), leafOrder_(leafOrder), closest_(c), any_(a), accel_(accel), flags_(flags) { }

    bool intersect(const Ray& ray, RayHitInfo* const hitInfo) const override
    {
        static_assert(sizeof(Ray) == 32, "Ray must be the 32-byte record the engine takes");
        if (!std::isinf(ray.maxT)) {                        // Scene::intersectShadow
            uint8_t occ = 0;
            if (any_(accel_, &ray, 1, &occ, flags_) != 0) { failed = true; return false; }
            ++numAny;
            return occ != 0;
        }
        EngineHit h;
        if (closest_(accel_, &ray, 1, &h, flags_) != 0) { failed = true; return false; }
        ++numClosest;
        if (h.prim == -1 || !(h.t < hitInfo->t)) return false;
        fillHitInfo(geometry_, leafOrder_, ray, h, hitInfo);
        return true;
    }
    Vector3f sample(float, float, float* pdf) const override { *pdf = 0.0f; return Vector3f(0.0f, 0.0f, 0.0f); }
    float area() const override { return 0.0f; }

    mutable std::atomic<uint64_t> numClosest{0}, numAny{0};
    mutable std::atomic<bool> failed{false};

private:
    const Scene& geometry_;
    const TriAccel* leafOrder_;
    EngineClosestFn closest_;
    EngineAnyFn any_;
    void* accel_;
    unsigned flags_;
};


// trace() (src/renderer.cpp:29-136) restated as the three callbacks of yart_amd::PathIntegrator
// (yart_amd/cxx/wavefront.hpp): the same statements on the reference's own Rng / Camera / Light / Bsdf /
// Frame / Spectrum objects, cut at the two intersection calls, state kept per pixel.  Every pixel
// consumes its random numbers in trace()'s order (this TU is compiled by the same g++, so the
// unspecified evaluation order of the paired randomFloat() arguments, :47-50, :98-99, :118-119, is the
// one trace() gets), hence a batched run must reproduce trace()'s pixels bit for bit.
class RefTraceIntegrator : public yart_amd::PathIntegrator {
public:
    RefTraceIntegrator(const Scene& geometry, const TriAccel* leafOrder, Camera& camera, const int32_t* pixels, size_t n)
        : geometry_(geometry), leafOrder_(leafOrder), camera_(camera), slots_(n)
    {
        for (size_t i = 0; i < n; ++i) {
            Slot& s = slots_[i];
            s.x = pixels[i * 2 + 0];
            s.y = pixels[i * 2 + 1];
            s.rng = Rng(s.y * camera.getWidth() + s.x);                                   // :33
            s.finalColor = Spectrum(0.0f);
        }
    }

    bool nextPath(size_t slot, YartRay* ray) override
    {
        Slot& s = slots_[slot];
        if (s.k == kMaxIter) {
            if (!s.written) { camera_.accumulate(s.x, s.y, s.finalColor.toRGB()); s.written = true; }   // :135
            return false;
        }
        s.color = Spectrum(0.0f);                                                         // :44-45
        s.pathWeight = Spectrum(1.0f);
        s.currentRay = camera_.sample(s.x + s.rng.randomFloat() - 0.5f, s.y + s.rng.randomFloat() - 0.5f);   // :47-50
        s.evaluateDirectLightHit = true;                                                  // :52
        s.bounce = 0;
        std::memcpy(ray, &s.currentRay, sizeof(Ray));
        return true;
    }

    bool onClosest(size_t slot, const YartRay&, const YartHit& hit, YartRay* shadowRay) override
    {
        using std::abs;
        Slot& s = slots_[slot];
        RayHitInfo& isect = s.isect;
        isect.t = s.currentRay.maxT;                                                      // include/scene.h:139-140
        isect.areaLight = nullptr;
        if (hit.prim == -1) return endSample(s);                                          // :59-60
        EngineHit h{ hit.t, hit.u, hit.v, hit.prim };
        fillHitInfo(geometry_, leafOrder_, s.currentRay, h, &isect);
        if (s.evaluateDirectLightHit && isect.areaLight) s.color += s.pathWeight * isect.areaLight->intensity();   // :62-64
        if (!isect.bsdf) return endSample(s);                                             // :66-67
        s.intersection = s.currentRay.orig + (s.currentRay.dir * isect.t);                // :74
        auto norm = isect.shadingNormal;
        s.nl = dot(isect.normal, s.currentRay.dir) < 0.0f ? norm : norm * -1.0f;          // :76
        s.hitFrame = Frame(s.nl);
        s.wo = s.hitFrame.toLocal(-s.currentRay.dir);
        const auto& lights = geometry_.getLights();
        s.numLights = static_cast<int32_t>(lights.size());
        int32_t lightIdx = std::min(static_cast<int32_t>(s.rng.randomFloat() * s.numLights), s.numLights - 1);   // :90-93
        const auto& light = lights[lightIdx];
        float eps;
        Vector3f sampledPosition;
        s.lightEmission = light->sample(s.intersection, &s.wi, &s.pdf, &sampledPosition, &eps,
            s.rng.randomFloat(), s.rng.randomFloat());                                    // :98-99
        auto lightRay = Ray(s.intersection + s.wi * EPS, s.wi);                           // :101-102
        lightRay.maxT = length(s.intersection - sampledPosition) - eps;
        std::memcpy(shadowRay, &lightRay, sizeof(Ray));
        return true;
    }

    bool onShadow(size_t slot, bool occluded, YartRay* nextRay) override
    {
        using std::abs;
        Slot& s = slots_[slot];
        RayHitInfo& isect = s.isect;
        if (!occluded) {                                                                  // :104-108
            Spectrum f = isect.bsdf->f(s.wo, s.wi);
            s.color = s.color + (s.pathWeight * f * s.lightEmission
                * (abs(dot(s.nl, s.wi)) / s.pdf) * (float)s.numLights);
        }
        float continueProbability = s.pathWeight.y();                                     // :113-115
        if (s.rng.randomFloat() > continueProbability) return endSample(s);
        s.pathWeight /= continueProbability;                                              // :117
        Vector3f wi;
        float pdf;
        Spectrum refl = isect.bsdf->sample(s.wo, &wi, s.rng.randomFloat(), s.rng.randomFloat(), &pdf);   // :118-119
        if (refl.y() == 0.0f) return endSample(s);                                        // :121-122
        s.evaluateDirectLightHit = isect.bsdf->isDelta();                                 // :124
        Vector3f dir = s.hitFrame.toWorld(wi);                                            // :126
        s.pathWeight = s.pathWeight * refl * abs(dot(dir, s.nl)) / pdf;                   // :128
        s.currentRay = { s.intersection + dir * EPS, dir };                               // :129
        if (++s.bounce == 10) return endSample(s);                                        // :58
        std::memcpy(nextRay, &s.currentRay, sizeof(Ray));
        return true;
    }

private:
    static constexpr int kMaxIter = 4096;                                                 // :36
    struct Slot {
        Rng rng;
        int32_t x = 0, y = 0;
        int k = 0, bounce = 0;
        bool written = false, evaluateDirectLightHit = true;
        Spectrum finalColor, color, pathWeight, lightEmission;
        Ray currentRay{ Vector3f(0.0f, 0.0f, 0.0f), Vector3f(0.0f, 0.0f, 1.0f) };
        RayHitInfo isect;
        Vector3f intersection, nl, wo, wi;
        Frame hitFrame;
        float pdf = 0.0f;
        int32_t numLights = 0;
    };
    static bool endSample(Slot& s)
    {
        static const float invMaxIter = 1.0f / kMaxIter;                                  // :37
        s.finalColor = s.finalColor + (s.color * invMaxIter);                             // :132
        ++s.k;
        return false;
    }

    const Scene& geometry_;
    const TriAccel* leafOrder_;
    Camera& camera_;
    std::vector<Slot> slots_;
};

// The camera of src/rt.cpp:22-28 at the given resolution.
Camera cornellCamera(int width, int height)
{
    return Camera(Vector3f(50.0f, 48.0f, 220.0f), normal(Vector3f(0.0f, -0.042612f, -1.0f)), width, height, 0.785398f);
}

} // namespace

extern "C" {

// trace() (src/renderer.cpp:29-136: 4096 samples per pixel, <= 10 bounces, Rng(y * width + x)) on
// the listed pixels of a width x height Cornell-box frame (Scene::makeCornellBox, camera of
// src/rt.cpp:22-28).  rgbOut[i] = the pixel trace() accumulated (Camera::accumulate).
//   mode 0  the reference's own intersection code
//   mode 1  as 0, plus the ray stream: every ray handed to Scene::intersect / intersectShadow is
//           appended to *closestLog / *shadowLog (8 floats each; malloc'ed here, release with
//           ref_free); pixel order, single thread, so the order is trace()'s own
//   mode 2  intersection through the engine callbacks (EngineShape); accel / flags are passed on.
//           Returns the number of engine calls made, or -1 if an engine call failed.
//   mode 3  trace() restated as a yart_amd::PathIntegrator, driven by yart_amd::WavefrontDriver: all
//           listed pixels in flight at once, one closest-hit batch + one shadow batch per round.
//           Returns the number of rays traced; *nClosest = closest-hit batches, *nShadow = largest batch.
//   mode 4  as 3, with the batches going through a YartQueue (pinned mirrors of device arrays; the
//           yart_queue_* entry points are handed over beforehand with ref_set_queue_api).
static yart_amd::QueueApi g_queueApi;
static bool g_queueApiSet = false;

void ref_set_queue_api(void* const* fns)      // create, destroy, rays, hits, surfaces, occluded, closest, any
{
    g_queueApi.create = reinterpret_cast<decltype(g_queueApi.create)>(fns[0]);
    g_queueApi.destroy = reinterpret_cast<decltype(g_queueApi.destroy)>(fns[1]);
    g_queueApi.rays = reinterpret_cast<decltype(g_queueApi.rays)>(fns[2]);
    g_queueApi.hits = reinterpret_cast<decltype(g_queueApi.hits)>(fns[3]);
    g_queueApi.surfaces = reinterpret_cast<decltype(g_queueApi.surfaces)>(fns[4]);
    g_queueApi.occluded = reinterpret_cast<decltype(g_queueApi.occluded)>(fns[5]);
    g_queueApi.closest = reinterpret_cast<decltype(g_queueApi.closest)>(fns[6]);
    g_queueApi.any = reinterpret_cast<decltype(g_queueApi.any)>(fns[7]);
    g_queueApiSet = true;
}

long ref_trace_cornell(int width, int height, const int32_t* pixels, size_t nPixels, int mode, int nthreads,
                       float* rgbOut, float** closestLog, size_t* nClosest, float** shadowLog, size_t* nShadow,
                       EngineClosestFn closestFn, EngineAnyFn anyFn, void* accel, unsigned flags)
{
    Scene cornell = Scene::makeCornellBox();
    Camera camera = cornellCamera(width, height);
    std::unique_ptr<Scene> scene;
    std::shared_ptr<EngineShape> engine;
    std::unique_ptr<BvhAccel> leafOrderSource;
    if (mode == 3 || mode == 4) {
        if (mode == 4 && !g_queueApiSet) { fprintf(stderr, "ref_trace_cornell: mode 4 needs ref_set_queue_api\n"); return -1; }
        // the batched driver (yart_amd/cxx/wavefront.hpp) with trace() restated as its integrator
        leafOrderSource = std::make_unique<BvhAccel>(cornell);
        RefTraceIntegrator integ(cornell, leafOrderSource->triangles_, camera, pixels, nPixels);
        yart_amd::Engine eng;
        eng.closest = reinterpret_cast<int (*)(YartAccel*, const YartRay*, size_t, YartHit*, unsigned)>(closestFn);
        eng.any = reinterpret_cast<int (*)(YartAccel*, const YartRay*, size_t, uint8_t*, unsigned)>(anyFn);
        eng.accel = static_cast<YartAccel*>(accel);
        eng.flags = flags;
        if (mode == 4) eng.queue = &g_queueApi;
        yart_amd::WavefrontDriver driver(eng, nPixels, nthreads);
        yart_amd::WavefrontStats st;
        try {
            st = driver.run(integ);
        } catch (const std::exception& e) {
            fprintf(stderr, "ref_trace_cornell: %s\n", e.what());
            return -1;
        }
        for (size_t i = 0; i < nPixels; ++i) {
            const RGBColor& c = camera.sensor_.buffer_[(size_t)pixels[i * 2 + 1] * width + pixels[i * 2 + 0]];
            rgbOut[i * 3 + 0] = c.r; rgbOut[i * 3 + 1] = c.g; rgbOut[i * 3 + 2] = c.b;
        }
        if (nClosest) *nClosest = st.closestBatches;
        if (nShadow) *nShadow = st.largestBatch;
        return (long)(st.closestRays + st.shadowRays);
    }
    if (mode == 2) {
        // prim ids of the engine index the reference's leaf-ordered TriAccel array
        leafOrderSource = std::make_unique<BvhAccel>(cornell);
        engine = std::make_shared<EngineShape>(cornell, leafOrderSource->triangles_, closestFn, anyFn, accel, flags);
        scene = std::make_unique<Scene>(std::vector<TriangleMesh>(), std::vector<std::shared_ptr<Shape>>{ engine },
                                        cornell.lights_);
    } else {
        auto shapes = cornell.shapes_;
        if (mode == 1) shapes.insert(shapes.begin(), std::make_shared<SpyShape>());
        scene = std::make_unique<Scene>(cornell.meshes_, shapes, cornell.lights_);
    }
    scene->accel_ = std::make_shared<BvhAccel>(*scene);      // include/scene.h:73 without the TriAccel8 path

    RayLog log;
    if (mode == 1) nthreads = 1;
    parallelFor1(nPixels, nthreads, [&](size_t i) {
        tlsLog = (mode == 1) ? &log : nullptr;
        trace(*scene, camera, pixels[i * 2 + 0], pixels[i * 2 + 1]);
        tlsLog = nullptr;
    });
    for (size_t i = 0; i < nPixels; ++i) {
        const RGBColor& c = camera.sensor_.buffer_[(size_t)pixels[i * 2 + 1] * width + pixels[i * 2 + 0]];
        rgbOut[i * 3 + 0] = c.r; rgbOut[i * 3 + 1] = c.g; rgbOut[i * 3 + 2] = c.b;
    }
    if (mode == 1) {
        *nClosest = log.closest.size() / 8;
        *nShadow = log.shadow.size() / 8;
        *closestLog = static_cast<float*>(std::malloc(log.closest.size() * 4 + 4));
        *shadowLog = static_cast<float*>(std::malloc(log.shadow.size() * 4 + 4));
        std::memcpy(*closestLog, log.closest.data(), log.closest.size() * 4);
        std::memcpy(*shadowLog, log.shadow.data(), log.shadow.size() * 4);
    }
    if (mode == 2) return engine->failed ? -1 : (long)(engine->numClosest + engine->numAny);
    return 0;
}

void ref_free(void* p) { std::free(p); }

// The reference's OWN integrator on a caller's scene: trace() (src/renderer.cpp:29-136, 4096 samples per pixel, <= 10
// bounces, Rng(y * width + x)) with the reference's Lambertian (src/bsdf.cpp:79-95) on every mesh and sphere and one
// PointLight (include/light.h:33-58), camera as src/rt.cpp:22-28 builds it (normal(dir)).  The fixture the DEVICE
// integrator (yart_amd/csrc/wavefront.hip) is pinned to: tests/golden/g6_trace_diffuse.npz.
//   pixels: x, y pairs; rgbOut: 3 floats per listed pixel; vertexNormalsOut (optional): the normals the reference's
//   TriangleMesh constructor generated (include/triangle.h:76-111), one per vertex of the input array.
int ref_trace_diffuse(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris, const float* spheres, size_t nSpheres,
                      const float* albedo, const float* lightPos, const float* lightIntensity,
                      const float* camPos, const float* camDir, float fov, int width, int height,
                      const int32_t* pixels, size_t nPixels, int nthreads, float* rgbOut, float* vertexNormalsOut)
{
    std::vector<TriangleMesh> meshes;
    const Spectrum alb(albedo[0], albedo[1], albedo[2]);
    std::vector<int32_t> vertexBase;
    size_t i = 0;
    while (i < nTris) {                                       // as buildScene: one mesh per run of equal mesh ids
        const int32_t mid = tris[i * 5 + 3];
        size_t j = i;
        int32_t vmin = std::numeric_limits<int32_t>::max(), vmax = -1;
        while (j < nTris && tris[j * 5 + 3] == mid) {
            for (int c = 0; c < 3; ++c) { vmin = std::min(vmin, tris[j * 5 + c]); vmax = std::max(vmax, tris[j * 5 + c]); }
            ++j;
        }
        if (mid != (int32_t)meshes.size() || vmin < 0 || vmax >= (int64_t)nVerts) return -1;
        std::vector<Vector3f> v;
        for (int32_t k = vmin; k <= vmax; ++k) v.emplace_back(verts[k * 3 + 0], verts[k * 3 + 1], verts[k * 3 + 2]);
        std::vector<Triangle> t;
        for (size_t k = i; k < j; ++k) t.emplace_back(tris[k * 5 + 0] - vmin, tris[k * 5 + 1] - vmin, tris[k * 5 + 2] - vmin);
        meshes.emplace_back(v, t, std::make_shared<Lambertian>(alb));
        vertexBase.push_back(vmin);
        i = j;
    }
    std::vector<std::shared_ptr<Shape>> shapes;
    for (size_t k = 0; k < nSpheres; ++k)
        shapes.push_back(std::make_shared<Sphere>(spheres[k * 4 + 3], Vector3f(spheres[k * 4 + 0], spheres[k * 4 + 1], spheres[k * 4 + 2]), alb, Bxdf::Diff));
    std::vector<std::shared_ptr<Light>> lights{ std::make_shared<PointLight>(
        Vector3f(lightPos[0], lightPos[1], lightPos[2]), Spectrum(lightIntensity[0], lightIntensity[1], lightIntensity[2])) };
    Scene scene(meshes, shapes, lights);
    scene.accel_ = std::make_shared<BvhAccel>(scene);        // include/scene.h:73 without the TriAccel8 path
    if (vertexNormalsOut) {
        std::memset(vertexNormalsOut, 0, nVerts * 12);
        for (size_t m = 0; m < scene.meshes_.size(); ++m) {
            const auto& nrm = scene.meshes_[m].normals_;
            for (size_t k = 0; k < nrm.size(); ++k) {
                float* o = vertexNormalsOut + (size_t)(vertexBase[m] + (int32_t)k) * 3;
                o[0] = nrm[k].x; o[1] = nrm[k].y; o[2] = nrm[k].z;
            }
        }
    }
    Camera camera(Vector3f(camPos[0], camPos[1], camPos[2]), normal(Vector3f(camDir[0], camDir[1], camDir[2])), width, height, fov);
    parallelFor1(nPixels, nthreads, [&](size_t p) { trace(scene, camera, pixels[p * 2 + 0], pixels[p * 2 + 1]); });
    for (size_t p = 0; p < nPixels; ++p) {
        const RGBColor& c = camera.sensor_.buffer_[(size_t)pixels[p * 2 + 1] * width + pixels[p * 2 + 0]];
        rgbOut[p * 3 + 0] = c.r; rgbOut[p * 3 + 1] = c.g; rgbOut[p * 3 + 2] = c.b;
    }
    return 0;
}

void* ref_scene_create(const float* verts, size_t nVerts, const int32_t* tris, size_t nTris,
                       const float* spheres, size_t nSpheres, int genNormals)
{
    return buildScene(verts, nVerts, tris, nTris, spheres, nSpheres, genNormals);
}

void ref_scene_destroy(void* h) { delete static_cast<RefScene*>(h); }

size_t ref_num_nodes(void* h)
{
    return static_cast<RefScene*>(h)->scene->accel_->optimizedAccel_.size();
}

void ref_get_nodes(void* h, void* out)
{
    const auto& nodes = static_cast<RefScene*>(h)->scene->accel_->optimizedAccel_;
    auto* o = static_cast<NodeOut*>(out);
    for (size_t i = 0; i < nodes.size(); ++i) {
        const auto& n = nodes[i];
        NodeOut r;
        std::memset(&r, 0, sizeof r);
        r.bounds[0] = n.bounds.min.x; r.bounds[1] = n.bounds.min.y; r.bounds[2] = n.bounds.min.z;
        r.bounds[3] = n.bounds.max.x; r.bounds[4] = n.bounds.max.y; r.bounds[5] = n.bounds.max.z;
        r.splitAxis = (uint8_t)n.splitAxis;
        r.offset = n.childOffset;
        r.numTriangles = (n.splitAxis == SplitAxis::None) ? n.numTriangles : 0;
        o[i] = r;
    }
}

void ref_get_triaccel(void* h, void* out)
{
    auto* rs = static_cast<RefScene*>(h);
    static_assert(sizeof(TriAccel) == 48, "TriAccel must be 48 bytes");
    std::memcpy(out, rs->scene->accel_->triangles_, rs->numTriangles * sizeof(TriAccel));
}

// Per-vertex normals the reference generated in TriangleMesh's constructor
// (include/triangle.h:76-111), scattered back to the caller's global vertex numbering
// (vertexBase[m] = first global vertex of mesh m).  Scene-ingestion data, exported so that
// fixtures can carry it as INPUT for the hit post-processing check.
void ref_get_vertex_normals(void* h, const int32_t* vertexBase, float* out)
{
    const auto& ms = static_cast<RefScene*>(h)->scene->meshes_;
    for (size_t m = 0; m < ms.size(); ++m) {
        const auto& n = ms[m].normals_;
        for (size_t k = 0; k < n.size(); ++k) {
            float* o = out + (size_t)(vertexBase[m] + (int32_t)k) * 3;
            o[0] = n[k].x; o[1] = n[k].y; o[2] = n[k].z;
        }
    }
}

// Scene::intersect (include/scene.h:136) on every ray; the primitive identity,
// which the reference API does not return, is recovered from the borrowed
// pointers in RayHitInfo (sphere: areaLight tag; triangle: bsdf -> mesh) and,
// inside the mesh, by replaying the reference's own public
// intersect(TriAccel, Ray, RayHitInfo*) (include/triaccel.h:170) until one
// record reproduces (t,u,v) bit-for-bit.
void ref_intersect(void* h, const float* rays, size_t n, void* out, int nthreads)
{
    auto* rs = static_cast<RefScene*>(h);
    auto* o = static_cast<HitOut*>(out);
    const Scene& scene = *rs->scene;
    const TriAccel* ta = scene.accel_->triangles_;
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray ray = makeRay(rays + i * 8);
            RayHitInfo isect;
            std::memset(&isect, 0, sizeof isect);
            bool hit = scene.intersect(ray, &isect);
            HitOut r;
            std::memset(&r, 0, sizeof r);
            r.hit = hit ? 1 : 0;
            r.t = isect.t;
            r.prim = -1;
            if (hit) {
                r.u = isect.u; r.v = isect.v;
                r.normal[0] = isect.normal.x; r.normal[1] = isect.normal.y; r.normal[2] = isect.normal.z;
                r.shadingNormal[0] = isect.shadingNormal.x;
                r.shadingNormal[1] = isect.shadingNormal.y;
                r.shadingNormal[2] = isect.shadingNormal.z;
                if (isect.areaLight != nullptr) {
                    size_t s = (reinterpret_cast<uintptr_t>(isect.areaLight) - 0x1000) / 16;
                    r.prim = -(int32_t)s - 2;
                    r.ambiguous = 1;
                } else {
                    int32_t mesh = rs->bsdfToMesh.at(isect.bsdf);
                    for (int32_t slot : rs->meshLeafSlots[mesh]) {
                        RayHitInfo probe;
                        probe.t = std::numeric_limits<float>::infinity();
                        if (::intersect(ta[slot], ray, &probe) &&
                            std::memcmp(&probe.t, &isect.t, 4) == 0 &&
                            std::memcmp(&probe.u, &isect.u, 4) == 0 &&
                            std::memcmp(&probe.v, &isect.v, 4) == 0) {
                            if (r.ambiguous == 0) r.prim = slot;
                            r.ambiguous++;
                        }
                    }
                }
            }
            o[i] = r;
        }
    });
}

// Scene::intersectShadow (include/scene.h:170)
void ref_intersect_shadow(void* h, const float* rays, size_t n, uint8_t* out, int nthreads)
{
    const Scene& scene = *static_cast<RefScene*>(h)->scene;
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray ray = makeRay(rays + i * 8);
            out[i] = scene.intersectShadow(ray) ? 1 : 0;
        }
    });
}

// The reference itself behind the two engine signatures of include/yart_amd.h (accel = a RefScene*):
// lets the CPU tests run everything that is written against those signatures -- the batched driver,
// the EngineShape hook -- without a GPU.
int ref_engine_closest(void* h, const void* rays, size_t n, void* hits, unsigned)
{
    std::vector<HitOut> tmp(n);
    ref_intersect(h, static_cast<const float*>(rays), n, tmp.data(), 1);
    auto* out = static_cast<EngineHit*>(hits);
    for (size_t i = 0; i < n; ++i) {
        out[i].t = tmp[i].t;
        out[i].u = tmp[i].hit ? tmp[i].u : 0.0f;
        out[i].v = tmp[i].hit ? tmp[i].v : 0.0f;
        out[i].prim = tmp[i].hit ? tmp[i].prim : -1;
    }
    return 0;
}

int ref_engine_any(void* h, const void* rays, size_t n, uint8_t* occluded, unsigned)
{
    ref_intersect_shadow(h, static_cast<const float*>(rays), n, occluded, 1);
    return 0;
}

// Throughput loop for the CPU baseline: nothing but Scene::intersect per ray,
// striding worker threads over the ray array the way the reference's tile
// workers would (src/scheduler.cpp:46-72).  Writes t only.
void ref_intersect_throughput(void* h, const float* rays, size_t n, float* tOut, int nthreads)
{
    const Scene& scene = *static_cast<RefScene*>(h)->scene;
    parallelFor(n, nthreads, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            Ray ray = makeRay(rays + i * 8);
            RayHitInfo isect;
            scene.intersect(ray, &isect);
            tOut[i] = isect.t;
        }
    });
}

// Geometry of Scene::makeCornellBox() (src/scene.cpp:4-157) flattened into the
// exchange format.  Call with null pointers to get the counts.
void ref_cornell(float* verts, size_t* nVerts, int32_t* tris, size_t* nTris,
                 float* spheres, size_t* nSpheres)
{
    Scene s = Scene::makeCornellBox();
    size_t nv = 0, nt = 0;
    for (size_t m = 0; m < s.meshes_.size(); ++m) {
        const auto& mesh = s.meshes_[m];
        const auto& v = mesh.vertices_;
        const auto& t = mesh.triangles_;
        for (size_t k = 0; k < t.size(); ++k) {
            if (tris) {
                tris[(nt + k) * 5 + 0] = t[k].idx0 + (int32_t)nv;
                tris[(nt + k) * 5 + 1] = t[k].idx1 + (int32_t)nv;
                tris[(nt + k) * 5 + 2] = t[k].idx2 + (int32_t)nv;
                tris[(nt + k) * 5 + 3] = (int32_t)m;
                tris[(nt + k) * 5 + 4] = (int32_t)k;
            }
        }
        for (size_t k = 0; k < v.size(); ++k) {
            if (verts) {
                verts[(nv + k) * 3 + 0] = v[k].x;
                verts[(nv + k) * 3 + 1] = v[k].y;
                verts[(nv + k) * 3 + 2] = v[k].z;
            }
        }
        nv += v.size();
        nt += t.size();
    }
    for (size_t i = 0; i < s.shapes_.size(); ++i) {
        auto* sp = static_cast<Sphere*>(s.shapes_[i].get());
        if (spheres) {
            spheres[i * 4 + 0] = sp->position_.x;
            spheres[i * 4 + 1] = sp->position_.y;
            spheres[i * 4 + 2] = sp->position_.z;
            spheres[i * 4 + 3] = sp->radius_;
        }
    }
    *nVerts = nv; *nTris = nt; *nSpheres = s.shapes_.size();
}

// Camera::sample (include/camera.h:24-31) for an array of film positions.
// Only the inline constructor and sample() are used; src/bitmap.cpp (which
// does not compile on g++ 11) is neither built nor needed.
void ref_camera_rays(const float* pos, const float* dir, int width, int height, float fov,
                     const float* xy, size_t n, float* raysOut)
{
    Camera cam(Vector3f(pos[0], pos[1], pos[2]), normal(Vector3f(dir[0], dir[1], dir[2])),
               width, height, fov);
    for (size_t i = 0; i < n; ++i) {
        Ray r = cam.sample(xy[i * 2 + 0], xy[i * 2 + 1]);
        float* o = raysOut + i * 8;
        o[0] = r.orig.x; o[1] = r.orig.y; o[2] = r.orig.z; o[3] = r.minT;
        o[4] = r.dir.x;  o[5] = r.dir.y;  o[6] = r.dir.z;  o[7] = r.maxT;
    }
}

// uniformSphereSample (include/qmc.h:82-94), used for the "incoherent" ray set
void ref_uniform_sphere(const float* u12, size_t n, float* dirOut)
{
    for (size_t i = 0; i < n; ++i) {
        Vector3f d = uniformSphereSample(u12[i * 2], u12[i * 2 + 1]);
        dirOut[i * 3 + 0] = d.x; dirOut[i * 3 + 1] = d.y; dirOut[i * 3 + 2] = d.z;
    }
}

// ---- Scene ingestion: the reference's own Scene::loadFromObj (src/scene.cpp:159-298)
// and, for the material table it does not keep, its bundled tinyobjloader called directly.
struct RefObj {
    std::vector<float> verts, normals, spheres, materials;     // materials: 9 floats each
    std::vector<int32_t> tris, meshes;                          // meshes: 6 ints each (= YartObjMesh)
};

void* ref_obj_load(const char* folder, const char* file)
{
    auto* o = new RefObj();
    Scene s = Scene::loadFromObj(folder, file);
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    tinyobj::LoadObj(shapes, materials, err, (std::string(folder) + file).c_str(), folder);
    int32_t v0 = 0, t0 = 0;
    for (size_t m = 0; m < s.meshes_.size(); ++m) {
        const auto& mesh = s.meshes_[m];
        for (const auto& v : mesh.vertices_) { o->verts.push_back(v.x); o->verts.push_back(v.y); o->verts.push_back(v.z); }
        for (const auto& n : mesh.normals_) { o->normals.push_back(n.x); o->normals.push_back(n.y); o->normals.push_back(n.z); }
        int32_t k = 0;
        for (const auto& t : mesh.triangles_) {
            const int32_t rec[5] = { t.idx0 + v0, t.idx1 + v0, t.idx2 + v0, (int32_t)m, k++ };
            o->tris.insert(o->tris.end(), rec, rec + 5);
        }
        const auto& tm = shapes[m].mesh;
        const int32_t rec[6] = { v0, (int32_t)mesh.vertices_.size(), t0, (int32_t)mesh.triangles_.size(),
                                 tm.material_ids[0], tm.positions.size() == tm.normals.size() ? 1 : 0 };
        o->meshes.insert(o->meshes.end(), rec, rec + 6);
        v0 += (int32_t)mesh.vertices_.size();
        t0 += (int32_t)mesh.triangles_.size();
    }
    for (const auto& sh : s.shapes_) {
        auto* sp = static_cast<Sphere*>(sh.get());
        o->spheres.insert(o->spheres.end(), { sp->position_.x, sp->position_.y, sp->position_.z, sp->radius_ });
    }
    for (const auto& m : materials) {
        o->materials.insert(o->materials.end(), { m.diffuse[0], m.diffuse[1], m.diffuse[2], m.specular[0], m.specular[1],
                                                  m.specular[2], m.shininess, m.ior, (float)m.illum });
    }
    return o;
}

void ref_obj_counts(void* h, size_t* nMeshes, size_t* nVerts, size_t* nTris, size_t* nSpheres, size_t* nMaterials)
{
    auto* o = static_cast<RefObj*>(h);
    *nMeshes = o->meshes.size() / 6; *nVerts = o->verts.size() / 3; *nTris = o->tris.size() / 5;
    *nSpheres = o->spheres.size() / 4; *nMaterials = o->materials.size() / 9;
}

void ref_obj_get(void* h, float* verts, float* normals, int32_t* tris, float* spheres, int32_t* meshes, float* materials)
{
    auto* o = static_cast<RefObj*>(h);
    std::memcpy(verts, o->verts.data(), o->verts.size() * 4);
    std::memcpy(normals, o->normals.data(), o->normals.size() * 4);
    std::memcpy(tris, o->tris.data(), o->tris.size() * 4);
    std::memcpy(spheres, o->spheres.data(), o->spheres.size() * 4);
    std::memcpy(meshes, o->meshes.data(), o->meshes.size() * 4);
    std::memcpy(materials, o->materials.data(), o->materials.size() * 4);
}

void ref_obj_free(void* h) { delete static_cast<RefObj*>(h); }

} // extern "C"
