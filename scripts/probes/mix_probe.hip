// Checks the v_fma_mix* helpers of gs_raster.hip against plain C++ on random inputs (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ float mix_mul_lo(uint32_t h, float x) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x)); return d; }
__device__ __forceinline__ float mix_mul_hi(uint32_t h, float x) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x)); return d; }
__device__ __forceinline__ float mix_mul_lo_sat(uint32_t h, float x) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(h), "v"(x)); return d; }
__device__ __forceinline__ float mix_one_minus_hi(uint32_t h) { float d; asm("v_fma_mix_f32 %0, %1, -1.0, 1.0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h)); return d; }
__device__ __forceinline__ float mix_one_minus_hi_v(uint32_t h) { float d; const float m1 = -1.0f, p1 = 1.0f; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(m1), "v"(p1)); return d; }
__device__ __forceinline__ void mix_blend_pair(uint32_t& acc, float pLo, float pHi, float t) {
    asm("v_fma_mixlo_f16 %0, %1, %3, %0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, %0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(pLo), "v"(pHi), "v"(t));
}
__device__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu)); }
__device__ uint32_t f2h(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f); }

__global__ void probe(const uint32_t* in, const float* fin, uint32_t n, unsigned long long* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = in[i], acc0 = in[i + n];
    const float x = fin[i], y = fin[i + n], t = fin[i + 2 * n];
    if (__float_as_uint(mix_mul_lo(h, x)) != __float_as_uint(h2f(h) * x)) atomicAdd(&bad[0], 1ull);
    if (__float_as_uint(mix_mul_hi(h, x)) != __float_as_uint(h2f(h >> 16) * x)) atomicAdd(&bad[1], 1ull);
    const float s = fminf(fmaxf(h2f(h) * x, 0.0f), 1.0f);
    if (__float_as_uint(mix_mul_lo_sat(h, x)) != __float_as_uint(s) && s == s) atomicAdd(&bad[2], 1ull);
    if (__float_as_uint(mix_one_minus_hi(h)) != __float_as_uint(1.0f - h2f(h >> 16))) atomicAdd(&bad[3], 1ull);
    if (__float_as_uint(mix_one_minus_hi_v(h)) != __float_as_uint(1.0f - h2f(h >> 16))) atomicAdd(&bad[4], 1ull);
    uint32_t acc = acc0;
    mix_blend_pair(acc, x, y, t);
    const uint32_t want = f2h(fmaf(x, t, h2f(acc0))) | (f2h(fmaf(y, t, h2f(acc0 >> 16))) << 16);
    if ((acc & 0xffffu) != (want & 0xffffu)) atomicAdd(&bad[5], 1ull);
    if ((acc >> 16) != (want >> 16)) atomicAdd(&bad[6], 1ull);
    if (i < 4) printf("i=%u h=%08x x=%g  one_minus asm %g / v %g / want %g   pair acc0=%08x got %08x want %08x\n", i, h, x, mix_one_minus_hi(h), mix_one_minus_hi_v(h), 1.0f - h2f(h >> 16), acc0, acc, want);
}

int main() {
    const uint32_t n = 1 << 22;
    std::vector<uint32_t> in(2 * n); std::vector<float> fin(3 * n);
    srand(1);
    auto rh = []() { // a finite non-negative half in [0, 2): what colours / alphas look like, plus a few specials
        uint32_t r = (uint32_t)rand();
        return (r % 0x4000u);
    };
    for (uint32_t i = 0; i < n; ++i) {
        in[i] = rh() | (rh() << 16); in[i + n] = rh() | (rh() << 16);
        fin[i] = (float)rand() / RAND_MAX; fin[i + n] = (float)rand() / RAND_MAX * 0.7f; fin[i + 2 * n] = (float)rand() / RAND_MAX;
        if (i % 97 == 0) fin[i] = 1e-40f * (rand() % 100);          // fp32 denormals
        if (i % 89 == 0) fin[i + 2 * n] = 6e-8f * (rand() % 100);   // results in the fp16 denormal range
    }
    uint32_t* din; float* dfin; unsigned long long* dbad;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dfin, fin.size() * 4); hipMalloc(&dbad, 64);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dfin, fin.data(), fin.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 64);
    probe<<<n / 256, 256>>>(din, dfin, n, dbad);
    unsigned long long bad[8];
    hipMemcpy(bad, dbad, 64, hipMemcpyDeviceToHost);
    printf("mismatches of %u: mul_lo %llu mul_hi %llu mul_lo_sat %llu one_minus(inline) %llu one_minus(vgpr) %llu pair_lo %llu pair_hi %llu\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6]);
    return 0;
}
