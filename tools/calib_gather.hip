// Calibration microbenchmark (diagnostic tool, not part of the library): random gathers of
// RECORD-byte records (64 or 128, aligned) from tables of several sizes, with TEAM lanes sharing a
// record (TEAM = 1: every lane fetches a whole record of its own with RECORD/16 dwordx4 loads --
// the access pattern of a one-ray-per-lane node fetch; TEAM = 4: a quad of adjacent lanes fetches
// one record, each lane RECORD/64 dwordx4 loads 64 bytes apart -- a quad-cooperative node fetch).
// Index arithmetic is a 32-bit multiply + mask so the loop is not ALU-bound.  Reports records/s
// and useful GB/s; every record of the table is read exactly once per pass (odd multiplier =
// bijection mod 2^k).  Run under rocprofv3 --pmc to calibrate FETCH_SIZE against known bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RECORD, int TEAM>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ tab, uint32_t recMask, uint32_t mult,
                                             uint32_t total, float* __restrict__ out)
{
    constexpr int kVec = RECORD / 16;             // float4 per record
    constexpr int kPerLane = kVec / TEAM;         // loads per lane per record
    const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stride = gridDim.x * 256 / TEAM;
    const int part = threadIdx.x % TEAM;
    float acc = 0.f;
    for (uint32_t i = gtid / TEAM; i < total; i += stride) {
        const uint32_t rec = (i * mult) & recMask;
        // TEAM lanes read 16-byte parts part, part+TEAM, ...: each load instruction covers TEAM*16
        // contiguous bytes per team
        const float4* p = tab + (size_t)rec * kVec + part;
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) { float4 a = p[k * TEAM]; acc += a.x + a.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int RECORD, int TEAM>
void run(const float4* tab, uint64_t bytes, float* out)
{
    const uint32_t nRec = (uint32_t)(bytes / RECORD);
    const uint32_t total = nRec < (1u << 25) ? (1u << 25) : nRec;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather<RECORD, TEAM>), dim3(256 * 8), dim3(256), 0, 0, tab, nRec - 1, 2654435761u, total, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double recPerS = total / (best * 1e-3);
    printf("table %5llu MB  record %3d B  %d lane(s)/record: %8.3f ms  %7.2f Grec/s  %8.1f GB/s useful  %.3f rec/clk/CU  %.3f 16B-lane-req/clk/CU\n",
           (unsigned long long)(bytes >> 20), RECORD, TEAM, best, recPerS / 1e9, recPerS * RECORD / 1e9,
           recPerS / 256 / 2.4e9, recPerS * (RECORD / 16) / 256 / 2.4e9);
}

int main()
{
    float* out; CK(hipMalloc(&out, 4));
    const uint64_t sizesMB[] = { 2, 16, 128, 1024 };
    for (uint64_t mb : sizesMB) {
        const uint64_t bytes = mb << 20;
        float4* tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0, bytes));
        run<16, 1>(tab, bytes, out);
        run<64, 1>(tab, bytes, out);
        run<64, 2>(tab, bytes, out);
        run<64, 4>(tab, bytes, out);
        run<128, 1>(tab, bytes, out);
        run<128, 4>(tab, bytes, out);
        run<128, 8>(tab, bytes, out);
        CK(hipFree(tab));
    }
    return 0;
}
