// VALU issue-rate probe for gfx950 (MI355X): cycles per wave64 instruction of the instruction kinds the blend and calc_view
// kernels are made of -- v_fma_f32, v_pk_fma_f32, v_exp_f32, v_fma_mix_f32, v_fma_mixlo_f16 -- as INDEPENDENT streams (8 accumulators:
// issue rate) and as a DEPENDENT chain (1 accumulator: latency), at 1, 2, 4 and 8 waves per SIMD.  The guide
// (MI355X_MICROARCH.md "Wave scheduling") states 2 cycles per wave64 VALU instruction; rounds 3-4 FITTED 4.8 cycles from SQ counters.
// This measures it: every wave times its own loop with s_memtime (shader clock) and the launch is timed with hipEvents; the
// constant 100 MHz s_memrealtime gives the shader clock the run had.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
// Output: one line per (kind, dep, waves/SIMD): cycles per wave-instruction seen by ONE wave, and per SIMD (elapsed cycles x SIMDs /
// wave-instructions issued on them) -- the second is the issue roof a VALU-bound kernel is priced against.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, PKFMA = 1, EXP = 2, MIX = 3, MIXLO = 4, NKIND = 5 };
static const char* kKindName[NKIND] = { "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_fma_mix_f32", "v_fma_mixlo_f16" };
constexpr int UNROLL = 8;              // instructions per loop iteration (8 accumulators when independent)

// one instruction of KIND on accumulator a (b, c: loop-invariant operands)
template <int KIND> __device__ __forceinline__ void op(float& a, float b, float c) {
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else if (KIND == MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(a) : "v"(b), "v"(c));
    else if (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c));
}
__device__ __forceinline__ void op_pk(v2f& a, v2f b, v2f c) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }

template <int KIND, bool DEP>
__global__ void probe(float* __restrict__ sink, int iters, unsigned long long* __restrict__ cycles, unsigned long long* __restrict__ realtime) {
    const float b = 1.0000001f, c = 1e-9f;
    const float seed = (float)(threadIdx.x & 63) * 1e-3f + 0.5f;
    unsigned long long t0, t1, r0, r1;
    float acc = 0.f;
    if (KIND == PKFMA) {
        v2f a[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) a[k] = v2f{ seed + (float)k, seed - (float)k };
        const v2f b2 = { b, b }, c2 = { c, c };
        __syncthreads();
        r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) op_pk(a[DEP ? 0 : k], b2, c2);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc += a[k].x + a[k].y;
    } else {
        float a[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) a[k] = seed + (float)k * 0.01f;
        __syncthreads();
        r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) op<KIND>(a[DEP ? 0 : k], b, c);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc += a[k];
    }
    if (acc == 123.456f) sink[0] = acc;                          // keeps the chains alive
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cycles[w] = t1 - t0; realtime[w] = r1 - r0;
    }
}

template <int KIND, bool DEP> static void launch(int grid, int block, float* sink, int iters, unsigned long long* cyc, unsigned long long* rt) {
    hipLaunchKernelGGL((probe<KIND, DEP>), dim3(grid), dim3(block), 0, 0, sink, iters, cyc, rt);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s (%s), %d CUs, clockRate %d kHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
    const int iters = 20000;                                      // x UNROLL = 160,000 instructions per wave
    const size_t maxWaves = (size_t)cus * 32;
    float* sink; unsigned long long *cyc, *rt;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, maxWaves * 8)); CK(hipMalloc(&rt, maxWaves * 8));
    std::vector<unsigned long long> hc(maxWaves), hr(maxWaves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# kind dep waves/SIMD | per-wave cycles/instr (median wave) | per-SIMD cycles/instr (s_memtime) | shader MHz (memtime/realtime) | launch us | Gwave-instr/s whole chip\n");
    for (int kind = 0; kind < NKIND; ++kind)
        for (int dep = 0; dep < 2; ++dep)
            for (int wps : { 1, 2, 4, 8 }) {
                // wps waves per SIMD = 4 wps waves per CU: blocks of min(1024, 256 wps) threads, 1 or 2 per CU
                const int block = std::min(1024, 256 * wps), perCu = (256 * wps) / block, grid = cus * perCu;
                for (int rep = 0; rep < 2; ++rep) {               // rep 0 warms up (clocks, code)
                    CK(hipEventRecord(e0, 0));
#define L(K) do { if (dep) launch<K, true>(grid, block, sink, iters, cyc, rt); else launch<K, false>(grid, block, sink, iters, cyc, rt); } while (0)
                    switch (kind) { case FMA: L(FMA); break; case PKFMA: L(PKFMA); break; case EXP: L(EXP); break; case MIX: L(MIX); break; default: L(MIXLO); break; }
#undef L
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                }
                float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
                const size_t waves = (size_t)grid * (block / 64);
                CK(hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hr.data(), rt, waves * 8, hipMemcpyDeviceToHost));
                std::vector<unsigned long long> s(hc.begin(), hc.begin() + waves);
                std::sort(s.begin(), s.end());
                const double med = (double)s[waves / 2], instr = (double)iters * UNROLL;
                std::vector<unsigned long long> sr(hr.begin(), hr.begin() + waves);
                std::sort(sr.begin(), sr.end());
                const double mhz = med / ((double)sr[waves / 2] / 100.0);      // s_memrealtime ticks at 100 MHz
                printf("%-16s %s %d | %7.3f | %7.3f | %7.0f | %8.1f | %8.1f\n", kKindName[kind], dep ? "dep" : "ind", wps, med / instr, med / (instr * wps), mhz,
                       ms * 1e3, (double)waves * instr / (ms * 1e-3) / 1e9);
            }
    printf("# reading: 'per-SIMD cycles/instr' at 4-8 waves, independent = the issue roof (guide: 2.0; a SIMD-16 pipe: 4.0); dep @ 1 wave = the instruction's latency\n");
    return 0;
}
