// A minimal fork-join pool for the host builders: jobs may spawn jobs; run() returns when none is
// queued or running.  Every range of a top-down tree build is handled by exactly one job, so the
// result does not depend on how the jobs are scheduled.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace yart {

// host threads for the build: the machine's, capped by OMP_NUM_THREADS (the GPU boxes set it to
// their CPU share) and by 32
inline unsigned buildThreads()
{
    unsigned n = std::thread::hardware_concurrency();
    if (const char* v = std::getenv("OMP_NUM_THREADS")) { const int m = std::atoi(v); if (m > 0) n = std::min<unsigned>(n, static_cast<unsigned>(m)); }
    return std::max(1u, std::min(n, 32u));
}

template <class Job>
class TaskPool {
public:
    using Fn = std::function<void(const Job&)>;

    // process(job) is called once per job, possibly concurrently for different jobs; it may call spawn()
    void run(unsigned nThreads, const Job& root, Fn process)
    {
        process_ = std::move(process);
        jobs_.clear();
        jobs_.push_back(root);
        pending_ = 1;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back([this]() { work(); });
        work();
        for (auto& t : pool) t.join();
    }

    void spawn(const Job& j)
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            ++pending_;
            jobs_.push_back(j);
        }
        cv_.notify_one();
    }

private:
    void work()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this]() { return !jobs_.empty() || pending_ == 0; });
                if (jobs_.empty()) return;
                j = jobs_.back();
                jobs_.pop_back();
            }
            process_(j);
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) cv_.notify_all();
            }
        }
    }

    Fn process_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Job> jobs_;
    int pending_ = 0;
};

// parallel loop over [0, n) in chunks
template <class F>
void parallelChunks(size_t n, unsigned nThreads, size_t chunk, F&& f)
{
    if (nThreads <= 1 || n <= chunk) { f(size_t(0), n); return; }
    std::atomic<size_t> next{ 0 };
    auto body = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(chunk);
            if (b >= n) return;
            f(b, std::min(n, b + chunk));
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(body);
    body();
    for (auto& t : pool) t.join();
}

} // namespace yart
