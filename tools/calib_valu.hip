// VALU issue-rate calibration (diagnostic tool, not part of the library): how many cycles does a SIMD
// spend per wave64 vector instruction of the kinds the traversal's node step is made of, with 1, 2, 4
// and 8 waves resident per SIMD?  Answers whether SQ_INSTS_VALU x 4 clk (one quad-cycle each) or x 2 clk
// is the right price when reading the kernel's counters.  Cycles from s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;
constexpr int kUnroll = 16;         // independent instructions per iteration

template <int KIND>
__global__ void __launch_bounds__(256) valu(float* out, uint64_t* cycles, float seed)
{
    float a[kUnroll];
    f32x2 p[kUnroll / 2];
    uint32_t w = __float_as_uint(seed) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < kUnroll / 2; ++i) p[i] = (f32x2){ seed + i, seed - i };
    const f32x2 pm = { seed, seed * 0.5f }, pa = { 1.0f, seed };
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        // inline asm: the compiler would otherwise pack scalar fmas into v_pk_fma_f32 and hoist the conversions
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[0]));
            if (KIND == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 2) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(w));
            if (KIND == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(seed), "v"(pa[1]) : "vcc");
            if (KIND == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed) : );
            if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(seed) : "vcc");
            if (KIND == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 11) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(seed));
            if (KIND == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 15) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(a[i]) : "v"(w));
            if (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 18) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (KIND == 19) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(pa[1]));
            if (KIND == 20) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 21) asm volatile("v_min_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 22) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a[i]) : "v"(w));
            // src0 read as the f16 in the low / high half of the register, src1 and src2 as f32
            if (KIND == 25) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(w), "v"(seed));
            if (KIND == 26) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(w), "v"(seed));
            if (KIND == 27) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(w));
            if (KIND == 28) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(w));
            if (KIND == 29) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 30) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 31) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (KIND == 32) asm volatile("v_lshlrev_b32 %0, 10, %1" : "=v"(a[i]) : "v"(w));
            if (KIND == 33) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(w), "v"(seed));
            if (KIND == 34) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(w), "v"(seed));
            if (KIND == 35) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(w), "v"(seed));
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pa));
        }
        if (KIND == 36) {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0] clamp" : "+v"(p[i]) : "v"(pm), "v"(pa));
        }
        if (KIND == 37) {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(pm), "v"(pa));
        }
        if (KIND >= 40 && KIND <= 47) {
            // operand classes: does the packed fma (with and without clamp) slow down on denormal, huge or NaN values?
            const float x = KIND == 40 ? 3.0f : KIND == 41 ? -3.0f : KIND == 42 ? 1e-41f : KIND == 43 ? 1e30f :
                            KIND == 44 ? __builtin_inff() : KIND == 45 ? __builtin_nanf("") : KIND == 46 ? 1e-41f : 0.25f;
            const f32x2 src = { x * seed, x * seed };
            const f32x2 one = { seed, seed }, zero = { 0.0f, 0.0f };
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) {
                if (KIND == 46) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p[i]) : "v"(src), "v"(one), "v"(zero));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(p[i]) : "v"(src), "v"(one), "v"(zero));
            }
        }
        if (KIND == 23) {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
        }
        if (KIND == 24) {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < kUnroll / 2; ++i) s += p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int instrPerIter, int blocksPerCU, float* out, uint64_t* dCycles, int numCUs)
{
    const int grid = numCUs * blocksPerCU;
    const int waves = grid * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((valu<KIND>), dim3(grid), dim3(256), 0, 0, out, dCycles, 1.0001f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((valu<KIND>), dim3(grid), dim3(256), 0, 0, out, dCycles, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> c(waves);
    CK(hipMemcpy(c.data(), dCycles, waves * 8, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double med = (double)c[waves / 2];
    const double instr = (double)kIters * instrPerIter;
    // waves per SIMD = blocksPerCU (a 256-thread block puts one wave on each of the CU's 4 SIMDs)
    printf("%-28s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, kernel %.3f ms -> %.2f ns per instruction per SIMD\n",
           name, blocksPerCU, med / instr, ms, ms * 1e6 / (instr * blocksPerCU));
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int numCUs = prop.multiProcessorCount;
    float* out; CK(hipMalloc(&out, 4));
    uint64_t* dCycles; CK(hipMalloc(&dCycles, (size_t)numCUs * 8 * 4 * 8));
    printf("CUs %d, clock %d kHz\n", numCUs, prop.clockRate);
    for (int b : { 1, 8 }) {
        run<0>("v_fma_f32", kUnroll, b, out, dCycles, numCUs);
        run<1>("v_min_f32 (e32)", kUnroll, b, out, dCycles, numCUs);
        run<21>("v_min_f32 (e64)", kUnroll, b, out, dCycles, numCUs);
        run<2>("v_cvt_f32_ubyte1", kUnroll, b, out, dCycles, numCUs);
        run<15>("v_cvt_f32_u32_sdwa byte2", kUnroll, b, out, dCycles, numCUs);
        run<22>("v_bfe_u32", kUnroll, b, out, dCycles, numCUs);
        run<3>("v_cmp + v_cndmask", 2 * kUnroll, b, out, dCycles, numCUs);
        run<8>("v_cndmask_b32", kUnroll, b, out, dCycles, numCUs);
        run<9>("v_cmp_lt_f32", kUnroll, b, out, dCycles, numCUs);
        run<5>("v_max3_f32", kUnroll, b, out, dCycles, numCUs);
        run<17>("v_med3_f32", kUnroll, b, out, dCycles, numCUs);
        run<6>("v_mul_f32", kUnroll, b, out, dCycles, numCUs);
        run<18>("v_mul_f32 x,x,x", kUnroll, b, out, dCycles, numCUs);
        run<7>("v_add_f32", kUnroll, b, out, dCycles, numCUs);
        run<19>("v_fmac_f32", kUnroll, b, out, dCycles, numCUs);
        run<20>("v_rcp_f32", kUnroll, b, out, dCycles, numCUs);
        run<10>("v_and_or_b32", kUnroll, b, out, dCycles, numCUs);
        run<11>("v_min_u32", kUnroll, b, out, dCycles, numCUs);
        run<12>("v_mov_b32", kUnroll, b, out, dCycles, numCUs);
        run<13>("v_mad_u32_u24", kUnroll, b, out, dCycles, numCUs);
        run<14>("v_lshl_add_u32", kUnroll, b, out, dCycles, numCUs);
        run<16>("v_perm_b32", kUnroll, b, out, dCycles, numCUs);
        run<25>("v_fma_mix_f32 (f16 lo src0)", kUnroll, b, out, dCycles, numCUs);
        run<26>("v_fma_mix_f32 (f16 hi src0)", kUnroll, b, out, dCycles, numCUs);
        run<27>("v_cvt_f32_f16", kUnroll, b, out, dCycles, numCUs);
        run<28>("v_cvt_f32_f16_sdwa word1", kUnroll, b, out, dCycles, numCUs);
        run<29>("v_max_f32", kUnroll, b, out, dCycles, numCUs);
        run<30>("v_pk_max_f16", kUnroll, b, out, dCycles, numCUs);
        run<31>("v_sub_f32", kUnroll, b, out, dCycles, numCUs);
        run<32>("v_lshlrev_b32", kUnroll, b, out, dCycles, numCUs);
        run<33>("v_and_b32", kUnroll, b, out, dCycles, numCUs);
        run<34>("v_add_u32", kUnroll, b, out, dCycles, numCUs);
        run<35>("v_add_u32_sdwa byte1", kUnroll, b, out, dCycles, numCUs);
        run<4>("v_pk_fma_f32", kUnroll / 2, b, out, dCycles, numCUs);
        run<37>("v_pk_fma_f32 op_sel_hi:[1,0,0]", kUnroll / 2, b, out, dCycles, numCUs);
        run<36>("v_pk_fma_f32 op_sel_hi clamp", kUnroll / 2, b, out, dCycles, numCUs);
        run<47>("pk_fma clamp, value 0.25", kUnroll / 2, b, out, dCycles, numCUs);
        run<40>("pk_fma clamp, value 3 (clamped)", kUnroll / 2, b, out, dCycles, numCUs);
        run<41>("pk_fma clamp, value -3 (clamped)", kUnroll / 2, b, out, dCycles, numCUs);
        run<42>("pk_fma clamp, denormal", kUnroll / 2, b, out, dCycles, numCUs);
        run<46>("pk_fma no clamp, denormal", kUnroll / 2, b, out, dCycles, numCUs);
        run<43>("pk_fma clamp, 1e30", kUnroll / 2, b, out, dCycles, numCUs);
        run<44>("pk_fma clamp, inf", kUnroll / 2, b, out, dCycles, numCUs);
        run<45>("pk_fma clamp, NaN", kUnroll / 2, b, out, dCycles, numCUs);
        run<23>("v_pk_mul_f32", kUnroll / 2, b, out, dCycles, numCUs);
        run<24>("v_pk_add_f32", kUnroll / 2, b, out, dCycles, numCUs);
    }
    return 0;
}
