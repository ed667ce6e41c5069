// Known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// Onesweep kernel uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 for 16-B/lane streams; other widths uncalibrated).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void calib_read_b32(const uint32_t* __restrict__ in, uint32_t* out, size_t n) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read_b128(const uint4* __restrict__ in, uint32_t* out, size_t n) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write_b32(uint32_t* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}
__global__ void calib_write_b128(uint4* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// scatter of 4-byte words in runs of 32 (128 B), the granularity of an Onesweep pass on uniformly random digits
__global__ void calib_scatter_runs32(uint32_t* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t run = i >> 5, lane = i & 31;
        const size_t runs = n >> 5;
        const size_t dst = ((run * 2654435761ull) % runs) * 32 + lane;     // a permutation of the runs when `runs` is a power of two
        out[dst] = (uint32_t)i;
    }
}
// bin_emit's rectangle gather (and, with 4-byte words, the Onesweep gather pass): `m` random 8-byte gathers over an array of `n` uint2
// (6.13 M x 8 B = 49 MB like C2's rects[]; 2.2 M gathers like its visible splats).  What does FETCH_SIZE tally for a LONE request?
// requested bytes = 8 m; sectors touched <= m (64 B each).  index = a multiplicative hash: no two neighbouring lanes share a sector.
__global__ void calib_gather_b64(const uint2* __restrict__ in, uint32_t* out, uint32_t n, uint32_t m) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(((unsigned long long)i * 2654435761ull + 12345ull) % n);
        const uint2 v = in[j];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_gather_b32(const uint32_t* __restrict__ in, uint32_t* out, uint32_t n, uint32_t m) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) acc += in[(uint32_t)(((unsigned long long)i * 2654435761ull + 12345ull) % n)];
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30;    // 1 GiB: four times the 256 MiB Infinity Cache
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        calib_read_b32<<<4096, 256>>>((const uint32_t*)a, (uint32_t*)b, bytes / 4);
        calib_read_b128<<<4096, 256>>>((const uint4*)a, (uint32_t*)b, bytes / 16);
        calib_write_b32<<<4096, 256>>>((uint32_t*)b, bytes / 4);
        calib_write_b128<<<4096, 256>>>((uint4*)b, bytes / 16);
        calib_scatter_runs32<<<4096, 256>>>((uint32_t*)b, bytes / 4);
        // flush the caches between the gather probes with a 1-GiB read (the array would otherwise sit in L2 / Infinity Cache from the last rep)
        calib_gather_b64<<<2048, 256>>>((const uint2*)a, (uint32_t*)b, 6131954u, 2200000u);
        calib_read_b128<<<4096, 256>>>((const uint4*)a, (uint32_t*)b, bytes / 16);
        calib_gather_b32<<<2048, 256>>>((const uint32_t*)a, (uint32_t*)b, 6131954u, 6131954u);
    }
    CK(hipDeviceSynchronize());
    printf("calib done: every streaming kernel moves %zu bytes; calib_gather_b64 = 2,200,000 gathers of 8 B over 49 MB, calib_gather_b32 = 6,131,954 gathers of 4 B over 24.5 MB\n", bytes);
    return 0;
}
