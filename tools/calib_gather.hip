// Calibration microbenchmark (diagnostic tool, not part of the library):
// per-lane random gathers of 64-byte records as 4 x dwordx4 -- the access pattern of the
// traversal kernels' node fetch -- from tables of several sizes.  Reports records/s and
// GB/s of useful bytes; run under rocprofv3 --pmc to calibrate FETCH_SIZE / TCC_EA0_RDREQ
// against the known byte count (every record of the table is read exactly once per pass
// when `unique` is set: idx = (i * odd_multiplier) mod n).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LOADS>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ tab, uint64_t nRec, uint64_t mult,
                                             uint64_t total, float* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i < total; i += stride) {
        const uint64_t rec = (i * mult) % nRec;
        const float4* p = tab + rec * 4;
        float4 a = p[0];
        acc += a.x;
        if (LOADS >= 2) { float4 b = p[1]; acc += b.y; }
        if (LOADS >= 4) { float4 c = p[2], d = p[3]; acc += c.z + d.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// TEAM lanes share one record: lane j of a team reads the j-th (64/TEAM)-byte part.
template <int TEAM>
__global__ void __launch_bounds__(256) gatherTeam(const float4* __restrict__ tab, uint64_t nRec, uint64_t mult,
                                                 uint64_t total, float* __restrict__ out)
{
    const uint64_t gtid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256 / TEAM;
    const int part = threadIdx.x % TEAM;
    float acc = 0.f;
    for (uint64_t i = gtid / TEAM; i < total; i += stride) {
        const uint64_t rec = (i * mult) % nRec;
        const float4* p = tab + rec * 4 + part * (4 / TEAM);
        for (int k = 0; k < 4 / TEAM; ++k) { float4 a = p[k]; acc += a.x + a.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv)
{
    const int reps = 3;
    float* out; CK(hipMalloc(&out, 4));
    const uint64_t sizesMB[] = { 1, 16, 128, 1024 };
    for (uint64_t mb : sizesMB) {
        const uint64_t bytes = mb << 20, nRec = bytes / 64;
        float4* tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0, bytes));
        const uint64_t total = nRec < (1ull << 26) ? (1ull << 26) : nRec;   // >= one full pass
        const uint64_t mult = 2654435761ull | 1ull;                          // odd -> bijection mod 2^k
        for (int loads : { 4, 1 }) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0));
                if (loads == 4) hipLaunchKernelGGL(gather<4>, dim3(256 * 8), dim3(256), 0, 0, tab, nRec, mult, total, out);
                else hipLaunchKernelGGL(gather<1>, dim3(256 * 8), dim3(256), 0, 0, tab, nRec, mult, total, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            const double recPerS = total / (best * 1e-3);
            printf("table %5llu MB  %d x dwordx4 per record: %8.3f ms  %7.2f Grec/s  %8.1f GB/s useful  (%.3f rec/clk/CU @2.4GHz)\n",
                   (unsigned long long)mb, loads, best, recPerS / 1e9, recPerS * 16.0 * loads / 1e9, recPerS / 256 / 2.4e9);
        }
        for (int team : { 2, 4 }) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0));
                if (team == 2) hipLaunchKernelGGL(gatherTeam<2>, dim3(256 * 8), dim3(256), 0, 0, tab, nRec, mult, total, out);
                else hipLaunchKernelGGL(gatherTeam<4>, dim3(256 * 8), dim3(256), 0, 0, tab, nRec, mult, total, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            const double recPerS = total / (best * 1e-3);
            printf("table %5llu MB  %d lanes per 64-B record        : %8.3f ms  %7.2f Grec/s  %8.1f GB/s useful  (%.3f rec/clk/CU @2.4GHz)\n",
                   (unsigned long long)mb, team, best, recPerS / 1e9, recPerS * 64.0 / 1e9, recPerS / 256 / 2.4e9);
        }
        CK(hipFree(tab));
    }
    return 0;
}
