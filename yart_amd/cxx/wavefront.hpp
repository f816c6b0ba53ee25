// wavefront.hpp -- batched driver for a path tracer built like yart's.  Header-only C++17.
//
// The reference's integrator is synchronous and per pixel: trace() (src/renderer.cpp:29-136) calls
// Scene::intersect, shades, calls Scene::intersectShadow, samples the BSDF, and loops -- one ray per
// call, from 8 threads (src/scheduler.cpp:20).  The engine wants 10^5..10^8 rays per call.  This
// driver turns the loop inside out WITHOUT touching the integrator's arithmetic or the order in which
// a pixel consumes its random numbers: the integrator is cut at its two intersection calls into three
// callbacks, every pixel in flight ("slot") advances through them in its own order, and all slots
// that wait for the same kind of answer are traced as one batch.
//
//   nextPath   src/renderer.cpp:43-52   start the slot's next sample: camera ray (false: slot done)
//   onClosest  src/renderer.cpp:59-102  closest-hit answer -> shading set-up, light sample, shadow ray
//   onShadow   src/renderer.cpp:104-129 occlusion answer -> direct light, roulette, BSDF sample, next ray
//
// Callbacks of DIFFERENT slots run concurrently (host threads); a slot's callbacks run in order.
// The engine is reached through two function pointers with the signatures of
// yart_intersect_closest / yart_intersect_any (include/yart_amd.h); linkedEngine() binds the
// library's own.  With the reference's trace() restated as such an integrator the driver reproduces
// the reference's image bit for bit (tests/test_trace_dropin.py).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/yart_amd.h"

namespace yart_amd {

class PathIntegrator {
public:
    virtual ~PathIntegrator() = default;
    virtual bool nextPath(size_t slot, YartRay* ray) = 0;
    virtual bool onClosest(size_t slot, const YartRay& ray, const YartHit& hit, YartRay* shadowRay) = 0;
    // queue mode with Engine::surfaces: the hit with its normals (row K computed on the device); default: drop them
    virtual bool onClosestSurface(size_t slot, const YartRay& ray, const YartSurface& s, YartRay* shadowRay)
    {
        const YartHit hit = { s.t, s.u, s.v, s.prim };
        return onClosest(slot, ray, hit, shadowRay);
    }
    virtual bool onShadow(size_t slot, bool occluded, YartRay* nextRay) = 0;
};

// The yart_queue_* entry points (include/yart_amd.h): device arrays mirrored in pinned host memory.  With
// them the driver writes every batch straight into the pinned mirror and reads the results from there:
// no staging copies on the host, DMA both ways, optionally the hit post-processing on the device.
struct QueueApi {
    int (*create)(YartAccel*, size_t, YartQueue**) = nullptr;
    void (*destroy)(YartQueue*) = nullptr;
    YartRay* (*rays)(YartQueue*) = nullptr;
    const YartHit* (*hits)(const YartQueue*) = nullptr;
    const YartSurface* (*surfaces)(const YartQueue*) = nullptr;
    const uint8_t* (*occluded)(const YartQueue*) = nullptr;
    int (*closest)(YartQueue*, size_t, unsigned, int) = nullptr;
    int (*any)(YartQueue*, size_t, unsigned) = nullptr;
};

struct Engine {
    int (*closest)(YartAccel*, const YartRay*, size_t, YartHit*, unsigned) = nullptr;
    int (*any)(YartAccel*, const YartRay*, size_t, uint8_t*, unsigned) = nullptr;
    YartAccel* accel = nullptr;
    unsigned flags = YART_DEFAULT;
    const QueueApi* queue = nullptr;      // set: batches go through a YartQueue instead of the host-buffer calls
    bool surfaces = false;                // queue mode: also fetch YartSurface records (onClosestSurface)
};

inline Engine linkedEngine(YartAccel* accel, unsigned flags = YART_DEFAULT)
{
    Engine e;
    e.closest = &yart_intersect_closest;
    e.any = &yart_intersect_any;
    e.accel = accel;
    e.flags = flags;
    return e;
}

inline const QueueApi* linkedQueueApi()
{
    static const QueueApi api = { &yart_queue_create, &yart_queue_destroy, &yart_queue_rays, &yart_queue_hits,
                                  &yart_queue_surfaces, &yart_queue_occluded, &yart_queue_closest, &yart_queue_any };
    return &api;
}

struct WavefrontStats {
    uint64_t closestRays = 0, shadowRays = 0, closestBatches = 0, shadowBatches = 0, paths = 0;
    size_t largestBatch = 0;
};

// Host threads that live as long as the driver: a path-traced frame makes 10^4..10^5 batches, each
// followed by a parallel loop over its slots, so the threads are started once and woken per loop
// (the reference starts its 8 workers once per frame as well, src/scheduler.cpp:20-44).
class WorkerPool {
public:
    explicit WorkerPool(int nThreads)
    {
        for (int t = 1; t < nThreads; ++t) threads_.emplace_back([this]() { serve(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
            ++generation_;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

    // f(i) for every i in [0, n); returns when all calls have returned.  One loop at a time.
    template <class F>
    void parallelFor(size_t n, F&& f)
    {
        if (threads_.empty() || n < 64) { for (size_t i = 0; i < n; ++i) f(i); return; }
        const size_t chunk = n / ((threads_.size() + 1) * 8) + 1;
        std::atomic<size_t> next{ 0 };
        std::function<void()> body = [&]() {
            for (;;) {
                const size_t b = next.fetch_add(chunk);
                if (b >= n) return;
                const size_t e = b + chunk < n ? b + chunk : n;
                for (size_t i = b; i < e; ++i) f(i);
            }
        };
        {
            std::lock_guard<std::mutex> lock(m_);
            body_ = &body;
            busy_ = threads_.size();
            ++generation_;
        }
        wake_.notify_all();
        body();
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this]() { return busy_ == 0; });
        body_ = nullptr;
    }

private:
    void serve()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()>* body;
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&]() { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                body = body_;
            }
            (*body)();
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--busy_ == 0) done_.notify_one();
            }
        }
    }

    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void()>* body_ = nullptr;
    uint64_t generation_ = 0;
    size_t busy_ = 0;
    bool stop_ = false;
};

class WavefrontDriver {
public:
    WavefrontDriver(Engine engine, size_t slots, int hostThreads = 8)
        : engine_(engine), nSlots_(slots), pool_(hostThreads < 1 ? 1 : hostThreads) { }

    WavefrontStats run(PathIntegrator& integ)
    {
        enum : uint8_t { kIdle, kClosest, kShadow, kDone };
        std::vector<uint8_t> state(nSlots_, kIdle);
        std::vector<YartRay> ray(nSlots_);              // the slot's pending ray
        std::vector<uint32_t> ids;
        std::vector<YartRay> batch;
        std::vector<YartHit> hits;
        std::vector<uint8_t> occ;
        WavefrontStats st;
        ids.reserve(nSlots_);
        struct QueueHolder {
            const QueueApi* api = nullptr;
            YartQueue* q = nullptr;
            ~QueueHolder() { if (q) api->destroy(q); }
        } qh;
        if (engine_.queue) {
            qh.api = engine_.queue;
            check(qh.api->create(engine_.accel, nSlots_, &qh.q));
        }
        for (;;) {
            // slots without a path start their next one (sequential per slot, parallel across slots)
            std::atomic<uint64_t> started{ 0 };
            parallelFor(nSlots_, [&](size_t s) {
                if (state[s] != kIdle) return;
                if (integ.nextPath(s, &ray[s])) { state[s] = kClosest; started.fetch_add(1, std::memory_order_relaxed); }
                else state[s] = kDone;
            });
            st.paths += started.load();

            collect(state, kClosest, ids);
            if (!ids.empty()) {
                st.closestRays += ids.size();
                st.closestBatches++;
                if (ids.size() > st.largestBatch) st.largestBatch = ids.size();
                if (qh.q) {
                    // straight into the pinned mirror of the device queue, results read from the mirrors
                    YartRay* qr = qh.api->rays(qh.q);
                    parallelFor(ids.size(), [&](size_t i) { qr[i] = ray[ids[i]]; });
                    check(qh.api->closest(qh.q, ids.size(), engine_.flags, engine_.surfaces ? 1 : 0));
                    const YartHit* qhits = qh.api->hits(qh.q);
                    const YartSurface* qs = qh.api->surfaces(qh.q);
                    parallelFor(ids.size(), [&](size_t i) {
                        const uint32_t s = ids[i];
                        YartRay shadow;
                        const bool more = engine_.surfaces ? integ.onClosestSurface(s, qr[i], qs[i], &shadow)
                                                           : integ.onClosest(s, qr[i], qhits[i], &shadow);
                        if (more) { ray[s] = shadow; state[s] = kShadow; }
                        else state[s] = kIdle;
                    });
                } else {
                    gather(ids, ray, batch);
                    hits.resize(ids.size());
                    check(engine_.closest(engine_.accel, batch.data(), batch.size(), hits.data(), engine_.flags));
                    parallelFor(ids.size(), [&](size_t i) {
                        const uint32_t s = ids[i];
                        YartRay shadow;
                        if (integ.onClosest(s, batch[i], hits[i], &shadow)) { ray[s] = shadow; state[s] = kShadow; }
                        else state[s] = kIdle;
                    });
                }
            }

            collect(state, kShadow, ids);
            if (!ids.empty()) {
                st.shadowRays += ids.size();
                st.shadowBatches++;
                const uint8_t* occluded;
                if (qh.q) {
                    YartRay* qr = qh.api->rays(qh.q);
                    parallelFor(ids.size(), [&](size_t i) { qr[i] = ray[ids[i]]; });
                    check(qh.api->any(qh.q, ids.size(), engine_.flags));
                    occluded = qh.api->occluded(qh.q);
                } else {
                    gather(ids, ray, batch);
                    occ.resize(ids.size());
                    check(engine_.any(engine_.accel, batch.data(), batch.size(), occ.data(), engine_.flags));
                    occluded = occ.data();
                }
                parallelFor(ids.size(), [&](size_t i) {
                    const uint32_t s = ids[i];
                    YartRay next;
                    if (integ.onShadow(s, occluded[i] != 0, &next)) { ray[s] = next; state[s] = kClosest; }
                    else state[s] = kIdle;
                });
            }

            bool allDone = true;
            for (size_t s = 0; s < nSlots_ && allDone; ++s) allDone = state[s] == kDone;
            if (allDone) return st;
        }
    }

private:
    static void collect(const std::vector<uint8_t>& state, uint8_t want, std::vector<uint32_t>& ids)
    {
        ids.clear();
        for (size_t s = 0; s < state.size(); ++s)
            if (state[s] == want) ids.push_back(static_cast<uint32_t>(s));
    }
    static void gather(const std::vector<uint32_t>& ids, const std::vector<YartRay>& ray, std::vector<YartRay>& batch)
    {
        batch.resize(ids.size());
        for (size_t i = 0; i < ids.size(); ++i) batch[i] = ray[ids[i]];
    }
    static void check(int st)
    {
        if (st != YART_OK) throw std::runtime_error("yart_amd::WavefrontDriver: engine call failed, status " + std::to_string(st));
    }
    template <class F>
    void parallelFor(size_t n, F&& f) { pool_.parallelFor(n, std::forward<F>(f)); }

    Engine engine_;
    size_t nSlots_;
    WorkerPool pool_;
};

} // namespace yart_amd
