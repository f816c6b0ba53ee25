// WorkerPool of yart_amd/cxx/wavefront.hpp under ThreadSanitizer: many short loops of varying
// length, every index visited exactly once, no data race reported.
#include <cstdio>
#include <numeric>
#include "../../yart_amd/cxx/wavefront.hpp"

int main()
{
    yart_amd::WorkerPool pool(8);
    std::vector<uint32_t> visits(5000);
    uint64_t total = 0, expect = 0;
    for (int round = 0; round < 4000; ++round) {
        const size_t n = static_cast<size_t>((round * 7919) % 5000);
        std::fill(visits.begin(), visits.end(), 0u);
        pool.parallelFor(n, [&](size_t i) { visits[i] += 1; });
        for (size_t i = 0; i < visits.size(); ++i) {
            if (visits[i] != (i < n ? 1u : 0u)) { fprintf(stderr, "round %d index %zu visited %u times\n", round, i, visits[i]); return 1; }
        }
        total += std::accumulate(visits.begin(), visits.end(), uint64_t(0));
        expect += n;
    }
    printf("%llu %llu\n", (unsigned long long)total, (unsigned long long)expect);
    return total == expect ? 0 : 1;
}
