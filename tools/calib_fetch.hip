// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on the traversal kernel's own access shape
// (diagnostic tool, not part of the library).  Every kernel reads a table far
// larger than L2 + Infinity Cache exactly once per launch, so the bytes that must cross to HBM are
// known: table bytes.
//   calibGather64  one lane fetches a whole 64-byte record of its own with four dwordx4 loads, records in
//                  pseudo-random order (odd multiplier = bijection): the node fetch of traceWide
//   calibGather16  one lane fetches 16 bytes of a pseudo-random 64-byte line (a leaf plane row)
//   calibStream16  the guide's control: wide coalesced streaming read, 16 bytes per lane
// Run each under `rocprofv3 --pmc FETCH_SIZE` and `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`;
// factor = known bytes / (FETCH_SIZE x 1024).
//   usage: calib_fetch [table MiB, default 1024] [launches, default 3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) calibGather64(const float4* __restrict__ tab, uint32_t recMask, uint32_t mult,
                                                      uint32_t total, float* __restrict__ out)
{
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint32_t rec = (i * mult) & recMask;
        const float4* p = tab + (size_t)rec * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(256) calibGather16(const float4* __restrict__ tab, uint32_t recMask, uint32_t mult,
                                                      uint32_t total, float* __restrict__ out)
{
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint32_t rec = (i * mult) & recMask;
        const float4 a = tab[(size_t)rec * 4 + (i & 3u)];
        acc += a.x + a.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(256) calibStream16(const float4* __restrict__ tab, uint32_t totalVec, float* __restrict__ out)
{
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < totalVec; i += gridDim.x * 256) {
        const float4 a = tab[i];
        acc += a.x + a.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv)
{
    const uint64_t mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
    const int launches = argc > 2 ? atoi(argv[2]) : 3;
    const uint64_t bytes = mb << 20;
    const uint32_t nRec = (uint32_t)(bytes / 64);
    if ((nRec & (nRec - 1)) != 0) { printf("table size must be a power of two MiB\n"); return 1; }
    float* out; CK(hipMalloc(&out, 4));
    float4* tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(256 * 8), block(256);
    for (int k = 0; k < 3; ++k) {
        float best = 1e30f;
        for (int r = 0; r < launches; ++r) {
            CK(hipEventRecord(e0));
            if (k == 0) hipLaunchKernelGGL(calibGather64, grid, block, 0, 0, tab, nRec - 1, 2654435761u, nRec, out);
            if (k == 1) hipLaunchKernelGGL(calibGather16, grid, block, 0, 0, tab, nRec - 1, 2654435761u, nRec, out);
            if (k == 2) hipLaunchKernelGGL(calibStream16, grid, block, 0, 0, tab, (uint32_t)(bytes / 16), out);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const char* names[3] = { "calibGather64", "calibGather16", "calibStream16" };
        const double useful = (k == 1) ? (double)nRec * 16 : (double)bytes;
        printf("%s: table %llu MiB, %d launches, best %.3f ms; lines touched per launch %u x 64 B = %.0f bytes; bytes the lanes use %.0f (%.1f GB/s)\n",
               names[k], (unsigned long long)mb, launches, best, nRec, (double)nRec * 64, useful, useful / (best * 1e-3) / 1e9);
    }
    return 0;
}
