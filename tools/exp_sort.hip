// Experiment (not part of the library): the ray-key sort (2^24 pairs of 24-bit key + ray number) with rocPRIM's
// onesweep under other digit widths / block shapes than the library default that hipcub::DeviceRadixSort picks.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class Config>
void run(const char* name, uint32_t n, int keyBits, const uint32_t* dKeys0, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1,
         const std::vector<uint32_t>& want)
{
    size_t bytes = 0;
    rocprim::double_buffer<uint32_t> keys(k0, k1), vals(v0, v1);
    CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, keys, vals, n, 0, keyBits, 0)));
    void* temp; CK(hipMalloc(&temp, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipMemcpy(k0, dKeys0, (size_t)n * 4, hipMemcpyDeviceToDevice));
        keys = rocprim::double_buffer<uint32_t>(k0, k1); vals = rocprim::double_buffer<uint32_t>(v0, v1);
        CK(hipEventRecord(e0));
        CK((rocprim::radix_sort_pairs<Config>(temp, bytes, keys, vals, n, 0, keyBits, 0)));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<uint32_t> got(n);
    CK(hipMemcpy(got.data(), keys.current(), (size_t)n * 4, hipMemcpyDeviceToHost));
    printf("%-44s %.3f ms  temp %zu KB  %s\n", name, best, bytes >> 10, got == want ? "sorted" : "WRONG");
    CK(hipFree(temp));
}

int main()
{
    const uint32_t n = 1u << 24; const int keyBits = 24;
    std::vector<uint32_t> h(n);
    uint32_t x = 12345u;
    for (auto& k : h) { x = x * 1664525u + 1013904223u; k = x >> 8; }
    std::vector<uint32_t> want = h; std::sort(want.begin(), want.end());
    // keys sorted on their low 21 bits only (stable): what a 21-bit sort must return
    std::vector<uint32_t> want21 = h; std::stable_sort(want21.begin(), want21.end(), [](uint32_t a, uint32_t b) { return (a & 0x1FFFFFu) < (b & 0x1FFFFFu); });
    uint32_t *d0, *k0, *k1, *v0, *v1;
    CK(hipMalloc(&d0, (size_t)n * 4)); CK(hipMalloc(&k0, (size_t)n * 4)); CK(hipMalloc(&k1, (size_t)n * 4));
    CK(hipMalloc(&v0, (size_t)n * 4)); CK(hipMalloc(&v1, (size_t)n * 4));
    CK(hipMemcpy(d0, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(v0, 0, (size_t)n * 4));
    using namespace rocprim;
    run<default_config>("default", n, keyBits, d0, k0, k1, v0, v1, want);
#define OS(hb, hi, sb, si, bits) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<hb, hi>, kernel_config<sb, si>, bits, block_radix_rank_algorithm::match>, 0>
#if VARIANT == 0
    run<OS(1024, 16, 1024, 16, 8)>("onesweep 8 bits, 1024 x 16, match", n, keyBits, d0, k0, k1, v0, v1, want);
    run<OS(512, 16, 512, 16, 8)>("onesweep 8 bits, 512 x 16, match", n, keyBits, d0, k0, k1, v0, v1, want);
    run<OS(512, 24, 512, 24, 8)>("onesweep 8 bits, 512 x 24, match", n, keyBits, d0, k0, k1, v0, v1, want);
    run<OS(256, 24, 256, 24, 8)>("onesweep 8 bits, 256 x 24, match", n, keyBits, d0, k0, k1, v0, v1, want);
    run<OS(1024, 24, 1024, 24, 8)>("onesweep 8 bits, 1024 x 24, match", n, keyBits, d0, k0, k1, v0, v1, want);
#elif VARIANT == 3
    run<OS(256, 16, 256, 16, 11)>("21-bit keys: onesweep 11 bits, 256 x 16", n, 21, d0, k0, k1, v0, v1, want21);
    run<OS(256, 32, 256, 32, 11)>("21-bit keys: onesweep 11 bits, 256 x 32", n, 21, d0, k0, k1, v0, v1, want21);
    run<OS(1024, 16, 1024, 16, 8)>("21-bit keys: onesweep 8 bits, 1024 x 16", n, 21, d0, k0, k1, v0, v1, want21);
    run<OS(1024, 16, 1024, 16, 7)>("21-bit keys: onesweep 7 bits, 1024 x 16", n, 21, d0, k0, k1, v0, v1, want21);
#elif VARIANT == 1
    run<OS(256, 16, 256, 16, 12)>("onesweep 12 bits, 256 x 16, match", n, keyBits, d0, k0, k1, v0, v1, want);
    run<OS(256, 24, 256, 24, 12)>("onesweep 12 bits, 256 x 24, match", n, keyBits, d0, k0, k1, v0, v1, want);
#elif VARIANT == 2
    run<OS(512, 16, 512, 16, 12)>("onesweep 12 bits, 512 x 16, match", n, keyBits, d0, k0, k1, v0, v1, want);
#endif
    return 0;
}
