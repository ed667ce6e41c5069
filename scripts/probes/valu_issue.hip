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
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, PKFMA = 1, EXP = 2, MIX = 3, MIXLO = 4, CVTPK = 5, CVT16 = 6, CVT16HI = 7, CVT32 = 8, NKIND = 9 };
static const char* kKindName[NKIND] = { "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_cvt_pk_f16_f32", "v_cvt_f16_f32", "v_cvt_f16_f32_sdwa", "v_cvt_f32_f16" };
constexpr int UNROLL = 8;              // instructions per loop iteration (8 accumulators when independent)

// one instruction of KIND on accumulator a (b, c: loop-invariant operands)
template <int KIND> __device__ __forceinline__ void op(float& a, float b, float c) {
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else if (KIND == MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(a) : "v"(b), "v"(c));
    else if (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c));
    else if (KIND == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    else if (KIND == CVT16) asm volatile("v_cvt_f16_f32 %0, %1" : "+v"(a) : "v"(b));                     // (the plain conversion: RTNE, what __float2half compiles to)
    else if (KIND == CVT16HI) asm volatile("v_cvt_f16_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(a) : "v"(b));   // into the high half
    else if (KIND == CVT32) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
}
__device__ __forceinline__ void op_pk(v2f& a, v2f b, v2f c) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }

template <int KIND, bool DEP>
__global__ void probe(float* __restrict__ sink, int iters, unsigned long long* __restrict__ cycles, unsigned long long* __restrict__ realtime, unsigned long long* __restrict__ rstart) {
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
        cycles[w] = t1 - t0; realtime[w] = r1 - r0; rstart[w] = r0;
    }
}

template <int KIND, bool DEP> static void launch(int grid, int block, float* sink, int iters, unsigned long long* cyc, unsigned long long* rt, unsigned long long* rs) {
    hipLaunchKernelGGL((probe<KIND, DEP>), dim3(grid), dim3(block), 0, 0, sink, iters, cyc, rt, rs);
}

struct Row { double cycMin, cycMed, cycMax, durMinUs, durMedUs, durMaxUs, spreadUs, spanUs, launchUs, mhz, gwi, cycPerInstrSimdByTime; };

// One configuration: `wps` waves per SIMD on every SIMD of the chip (512-thread blocks: 2 waves per SIMD each, so a CU holds wps / 2 of them --
// or one 256-thread block for wps = 1), every wave timing its own loop.
static int run(int kind, int dep, int wps, int cus, int iters, float* sink, unsigned long long* cyc, unsigned long long* rt, unsigned long long* rs,
               std::vector<unsigned long long>& hc, std::vector<unsigned long long>& hr, std::vector<unsigned long long>& hs, hipEvent_t e0, hipEvent_t e1, Row& R) {
    const int block = wps == 1 ? 256 : 512, perCu = (256 * wps) / block, grid = cus * perCu;
    for (int rep = 0; rep < 2; ++rep) {               // rep 0 warms up (clocks, code)
        CK(hipEventRecord(e0, 0));
#define L(K) do { if (dep) launch<K, true>(grid, block, sink, iters, cyc, rt, rs); else launch<K, false>(grid, block, sink, iters, cyc, rt, rs); } while (0)
        switch (kind) { case FMA: L(FMA); break; case PKFMA: L(PKFMA); break; case EXP: L(EXP); break; case MIX: L(MIX); break; case CVTPK: L(CVTPK); break; case CVT16: L(CVT16); break; case CVT16HI: L(CVT16HI); break; case CVT32: L(CVT32); break; default: L(MIXLO); break; }
#undef L
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t waves = (size_t)grid * (block / 64);
    CK(hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, waves * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), rs, waves * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> s(hc.begin(), hc.begin() + waves), sr(hr.begin(), hr.begin() + waves);
    std::sort(s.begin(), s.end()); std::sort(sr.begin(), sr.end());
    unsigned long long first = ~0ull, lastStart = 0, lastEnd = 0;
    for (size_t w = 0; w < waves; ++w) { first = std::min(first, hs[w]); lastStart = std::max(lastStart, hs[w]); lastEnd = std::max(lastEnd, hs[w] + hr[w]); }
    const double instr = (double)iters * UNROLL;
    R.cycMin = s.front() / instr; R.cycMed = s[waves / 2] / instr; R.cycMax = s.back() / instr;
    R.durMinUs = sr.front() / 100.0; R.durMedUs = sr[waves / 2] / 100.0; R.durMaxUs = sr.back() / 100.0;     // s_memrealtime ticks at 100 MHz
    R.spreadUs = (lastStart - first) / 100.0; R.spanUs = (lastEnd - first) / 100.0; R.launchUs = ms * 1e3;
    R.mhz = (double)s[waves / 2] / ((double)sr[waves / 2] / 100.0);
    R.gwi = (double)waves * instr / (R.spanUs * 1e-6) / 1e9;                 // by the waves' own clock: first start to last end
    R.cycPerInstrSimdByTime = R.spanUs * R.mhz * ((double)cus * 4.0) / ((double)waves * instr);
    return 0;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "--quick";
    const int firstKind = (argc > 2 && std::string(argv[1]) == "--from") ? atoi(argv[2]) : 0;       // --from 6: the conversion kinds only
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = quick ? 10000 : 20000;                      // x UNROLL instructions per wave
    const size_t maxWaves = (size_t)cus * 32;
    float* sink; unsigned long long *cyc, *rt, *rs;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, maxWaves * 8)); CK(hipMalloc(&rt, maxWaves * 8)); CK(hipMalloc(&rs, maxWaves * 8));
    std::vector<unsigned long long> hc(maxWaves), hr(maxWaves), hs(maxWaves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Row R;
    if (quick) {
        // what bench.py prices a VALU-bound kernel against: plain independent v_fma_f32 streams, 8 waves per SIMD on the whole chip, timed by the
        // waves' own clocks (first start to last end); the shader clock under that load next to the clock one wave per SIMD runs at
        Row L;
        if (run(FMA, 0, 1, cus, iters, sink, cyc, rt, rs, hc, hr, hs, e0, e1, L)) return 1;
        if (run(FMA, 0, 8, cus, iters, sink, cyc, rt, rs, hc, hr, hs, e0, e1, R)) return 1;
        if (run(FMA, 0, 8, cus, iters, sink, cyc, rt, rs, hc, hr, hs, e0, e1, R)) return 1;
        printf("{\"gwi_per_s\": %.1f, \"waves_per_simd\": 8, \"cus\": %d, \"mhz_under_load\": %.0f, \"mhz_light\": %.0f, \"cycles_per_instr_per_simd\": %.3f, \"start_spread_us\": %.1f, "
               "\"span_us\": %.1f, \"launch_us\": %.1f, \"spec_gwi_per_s\": %.1f}\n", R.gwi, cus, R.mhz, L.mhz, R.cycPerInstrSimdByTime, R.spreadUs, R.spanUs, R.launchUs, cus * 4 * 2.4 / 2.0);
        return 0;
    }
    printf("# %s (%s), %d CUs, clockRate %d kHz; 512-thread blocks (2 waves per SIMD each), %d instructions per wave\n", prop.name, prop.gcnArchName, cus, prop.clockRate, iters * UNROLL);
    printf("# cyc/instr = s_memtime cycles per wave-instruction seen by a wave; dur = a wave's own loop by s_memrealtime (100 MHz); spread = last wave start - first wave start;\n"
           "# span = first start .. last end; launch = hipEvent bracket; MHz = median wave's cycles / its realtime; Gwi/s = wave-instr of the whole chip / span;\n"
           "# cyc/instr/SIMD = span x MHz x SIMDs / wave-instr (what the issue roof is at THAT clock; the guide: 2.0)\n");
    printf("# kind              dep w/SIMD | cyc/instr min med max     | dur us min med max          | spread us | span us | launch us |  MHz | Gwi/s | cyc/instr/SIMD\n");
    for (int kind = firstKind; kind < NKIND; ++kind)
        for (int dep = 0; dep < 2; ++dep)
            for (int wps : { 1, 2, 4, 6, 8 }) {
                if (run(kind, dep, wps, cus, iters, sink, cyc, rt, rs, hc, hr, hs, e0, e1, R)) return 1;
                printf("%-16s %s %d | %7.3f %7.3f %7.3f | %8.1f %8.1f %8.1f | %8.1f | %8.1f | %8.1f | %5.0f | %7.1f | %6.3f\n", kKindName[kind], dep ? "dep" : "ind", wps,
                       R.cycMin, R.cycMed, R.cycMax, R.durMinUs, R.durMedUs, R.durMaxUs, R.spreadUs, R.spanUs, R.launchUs, R.mhz, R.gwi, R.cycPerInstrSimdByTime);
            }
    printf("# spec: %d CUs x 4 SIMDs x 2.4 GHz / 2 cycles = %.1f Gwi/s\n", cus, cus * 4 * 2.4 / 2.0);
    return 0;
}
