// gs_sort.hip -- CSSetIndices, CSCalcDistances and the device radix sort, re-designed for gfx950.
//
// Replaces (semantics only -- the structure is new):
//   SplatUtilities.compute:59-82      CSSetIndices, CSCalcDistances
//   DeviceRadixSort.hlsl:42,163,428,451 + SortCommon.hlsl   InitDeviceRadixSort / Upsweep / Scan / Downsweep
//   GpuSorting.cs:142-198             the 13-dispatch reduce-then-scan driver
// with a Onesweep LSD radix sort (8-bit digits): ONE histogram sweep for all passes (fused into the key
// generation) + one chained-scan binning kernel per pass with a two-level decoupled look-back.  Contract kept:
// stable, ascending, (uint32 key, uint32 payload) pairs (KEY_UINT PAYLOAD_UINT SHOULD_ASCEND SORT_PAIRS).
//
// gfx950 specifics: 64-lane waves -- ranking is a wave-level multi-split from 8 __ballot()s per key (one per
// digit bit, folded with one v_bitop3 per 32-lane half) and v_mbcnt below the lane; per-wave digit histograms live in LDS; inter-workgroup
// look-back words are single 8-byte {epoch, flag, value} granules written/read with relaxed AGENT-scope atomics
// (the per-XCD L2s are not coherent, see MI355X_MICROARCH.md "inter-workgroup visibility"; the granule carries
// its own tag so no fence is needed and no status memset between passes: each pass uses a fresh epoch).
// Partitions are handed out by atomic tickets (16 counters in separate cache lines) inside a persistent grid, so a
// workgroup only ever waits on partitions that are running; all partitions of a pass are resident at once, so the
// look-back goes through per-group aggregates (32 partitions) instead of a chain of INCLUSIVE hand-offs; every spin is
// bounded and reports GS_ERR_SORT_TIMEOUT instead of hanging.  Design notes and measurements: DESIGN.md section 4.1.
#include "gs_common.h"

namespace gs {

namespace {

constexpr int RADIX = 256;
#ifndef GS_SORT_THREADS
#define GS_SORT_THREADS 512
#endif
#ifndef GS_SORT_KPT
#define GS_SORT_KPT 16
#endif
constexpr int THREADS = GS_SORT_THREADS;
constexpr int WAVES = THREADS / 64;
constexpr int KPT = GS_SORT_KPT;
constexpr int PART = THREADS * KPT;
constexpr uint32_t SPIN_LIMIT = 1u << 24;
constexpr uint32_t TICKET_CLASSES = 16;        // partition-ticket counters per pass (one 128-B line each)
#ifndef GS_SORT_GROUP
#define GS_SORT_GROUP 32
#endif
constexpr int GROUP = GS_SORT_GROUP;                     // partitions per look-back group (~sqrt of the partition count of a 6 M key sort)

constexpr unsigned long long FLAG_AGG = 1ull, FLAG_INCL = 2ull;

__device__ __forceinline__ unsigned long long pack_status(uint32_t epoch, unsigned long long flag, uint32_t value) {
    return ((unsigned long long)epoch << 34) | (flag << 32) | (unsigned long long)value;
}
__device__ __forceinline__ unsigned long long ld_status(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 32-bit element index on a wave-uniform base pointer: lets the compiler use the SGPR-base + 32-bit-VGPR-offset
// addressing form instead of a 64-bit address per access (arrays are < 4 GB: counts are capped at 2^30)
__device__ __forceinline__ uint32_t ldg32(const uint32_t* base, uint32_t idx) { return *(const uint32_t*)((const char*)base + (size_t)(idx << 2)); }
__device__ __forceinline__ void stg32(uint32_t* base, uint32_t idx, uint32_t v) { *(uint32_t*)((char*)base + (size_t)(idx << 2)) = v; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// add 1 to an LDS histogram bin; wave-aggregated when the whole wave hits one bin (the common case for the
// high digits of depth keys, where a per-lane ds_add would serialise 64 deep on one address)
__device__ __forceinline__ void lds_hist_add(uint32_t* h, uint32_t d) {
    const uint32_t first = __builtin_amdgcn_readfirstlane(d);
    const unsigned long long act = __ballot(1);
    if (__all(d == first)) {
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&h[first], (uint32_t)__popcll(act));
    } else {
        atomicAdd(&h[d], 1u);
    }
}

__global__ __launch_bounds__(1024) void set_indices_kernel(uint32_t* order, uint32_t n) {
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    if (i < n) order[i] = i;
}

// CSCalcDistances fused with the 4 digit histograms of the Onesweep sort.
// The kernel is a dependent chain  order[i] -> pos[order[i]] (random 4..12-B gather) -> key  per splat, i.e. pure memory
// latency: each thread carries GS_DIST_ILP independent chains (all index loads first, then all gathers), and the grid is
// sized for full occupancy, so that enough gathers are in flight to cover the ~2 us round trip of a miss.
#ifndef GS_DIST_ILP
#define GS_DIST_ILP 4
#endif
__global__ __launch_bounds__(256) void calc_distances_kernel(gsm::AssetView a, const uint32_t* __restrict__ order,
                                                             float m20, float m21, float m22, float m23,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ hist, uint32_t n,
                                                             unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords,
                                                             uint32_t* __restrict__ nextControl) {
    __shared__ uint32_t s_h[4 * RADIX];
    for (int j = threadIdx.x; j < 4 * RADIX; j += 256) s_h[j] = 0;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < groupAggWords; j += gridDim.x * 256u) groupAgg[j] = 0ull;   // the sort passes accumulate into it
    // the control block (histograms, tickets, error) of the NEXT sort: the two blocks alternate, so no memset launch per sort
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < (uint32_t)(sizeof(SortControl) / 4); j += gridDim.x * 256u) nextControl[j] = 0u;
    __syncthreads();
    constexpr uint32_t ILP = GS_DIST_ILP;
    constexpr uint32_t TILE = 256u * ILP;
    // XCD-aware traversal.  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), each XCD has its own 4 MB L2, and the
    // gather pos[order[i]] fetches a whole 64-byte sector (16 Norm11 positions) per splat.  The asset is in Morton order,
    // so the 16 splats of a sector are spatial neighbours and therefore close in DEPTH ORDER too: if one XCD walks a
    // contiguous eighth of the sorted positions front to back, the other 15 accesses of a sector arrive at the same L2
    // while the sector is still resident, instead of 16 fetches spread over 8 L2s.
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;     // gridDim.x is a multiple of 8
    const uint32_t tilesTotal = (n + TILE - 1u) / TILE;
    const uint32_t tilesPerXcd = (tilesTotal + 7u) / 8u;
    const uint32_t tileEnd = min(tilesTotal, (xcd + 1u) * tilesPerXcd);
    for (uint32_t tile = xcd * tilesPerXcd + slot; tile < tileEnd; tile += slots) {
        const uint32_t base = tile * TILE + threadIdx.x;
        uint32_t oi[ILP];
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            const uint32_t i = base + k * 256u;
            oi[k] = (i < n) ? order[i] : 0xffffffffu;
        }
        uint32_t key[ILP];
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            key[k] = (oi[k] != 0xffffffffu) ? gsm::SortKey(a, oi[k], m20, m21, m22, m23) : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            if (oi[k] == 0xffffffffu) continue;
            keys[base + k * 256u] = key[k];
            lds_hist_add(s_h, key[k] & 255u);
            lds_hist_add(s_h + RADIX, (key[k] >> 8) & 255u);
            lds_hist_add(s_h + 2 * RADIX, (key[k] >> 16) & 255u);
            lds_hist_add(s_h + 3 * RADIX, key[k] >> 24);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 4 * RADIX; j += 256) {
        const uint32_t c = s_h[j];
        if (c) atomicAdd(&hist[j], c);
    }
}

// stand-alone histogram (gs_sorter path): `passes` digit histograms of keys[0..n)
__global__ __launch_bounds__(256) void histogram_kernel(const uint32_t* __restrict__ keys, uint32_t nImm, const uint32_t* nPtr,
                                                        int passes, uint32_t lastMask, uint32_t* __restrict__ hist,
                                                        unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords) {
    __shared__ uint32_t s_h[4 * RADIX];
    for (int j = threadIdx.x; j < 4 * RADIX; j += 256) s_h[j] = 0;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < groupAggWords; j += gridDim.x * 256u) groupAgg[j] = 0ull;
    __syncthreads();
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t key = keys[i];
        for (int p = 0; p < passes; ++p) lds_hist_add(s_h + p * RADIX, (key >> (8 * p)) & (p == passes - 1 ? lastMask : 255u));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < passes * RADIX; j += 256) {
        const uint32_t c = s_h[j];
        if (c) atomicAdd(&hist[j], c);
    }
}

// One Onesweep pass: reads (keysIn, valsIn), writes (keysOut, valsOut) stably partitioned by digit (key>>shift)&mask.
// Register diet (the kernel is latency-bound, so resident waves matter): payloads are loaded only after the keys
// have left the registers for LDS, local positions overwrite the ranks, and the digit of each output slot is kept
// packed 4 per register instead of a 32-bit global index per slot.
#ifdef GS_EXP_SORT_TIMELINE       // experiment build: per-partition phase timestamps (100 MHz wall clock) of the LAST launch
__device__ unsigned long long g_timeline[16384 * 16];
#define GS_TL(k) do { if (threadIdx.x == 0 && part < 16384u) g_timeline[part * 16u + (k)] = wall_clock64(); } while (0)
#else
#define GS_TL(k) do { } while (0)
#endif
#ifndef GS_SORT_LOOKBACK_BATCH
#define GS_SORT_LOOKBACK_BATCH 8
#endif
#ifndef GS_SORT_MINWAVES
#define GS_SORT_MINWAVES 6      // <= 80 VGPRs: three 512-thread workgroups per CU (a handful of loop-invariant values spill to scratch)
#endif
__global__ __launch_bounds__(THREADS, GS_SORT_MINWAVES) void onesweep_kernel(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
                                                           uint32_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut,
                                                           const uint32_t* __restrict__ hist, unsigned long long* status,
                                                           unsigned long long* groupAgg, uint32_t* ticket, uint32_t* error, uint32_t nImm, const uint32_t* nPtr,
                                                           uint32_t shift, uint32_t epoch, uint32_t digitMask) {
    __shared__ uint32_t s_hist[WAVES * RADIX];   // per-wave digit counts -> wave-exclusive offsets
    __shared__ uint32_t s_lbase[RADIX];          // exclusive digit offsets inside the partition
    __shared__ uint32_t s_gbase[RADIX];          // global index of local slot j with digit d = s_gbase[d] + j
    __shared__ uint32_t s_buf[PART];
    __shared__ uint32_t s_wtot[WAVES];
    __shared__ uint32_t s_htot[RADIX / 64];
    __shared__ uint32_t s_part;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    const uint32_t numParts = (n + PART - 1) / PART;

    // global exclusive digit offsets = exclusive scan of this pass's 256-bin histogram (raw counts, accumulated by the key
    // generation / histogram kernel): every workgroup scans it for itself instead of a separate 1-workgroup launch
    uint32_t histExcl = 0;
    if (tid < RADIX) {
        const uint32_t c = hist[tid];
        const uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) s_htot[w] = incl;
        histExcl = incl - c;
    }
    __syncthreads();
    if (tid < RADIX)
        for (int k = 0; k < w; ++k) histExcl += s_htot[k];

    for (;;) {
        __syncthreads();                                    // previous partition's LDS reads are finished
#ifdef GS_EXP_SORT_TIMELINE
        const unsigned long long tl0 = wall_clock64();
#endif
        // Partition tickets.  One counter would serialise every workgroup of the grid on a single L2 channel (~12 ns per
        // same-address atomic: the 768th workgroup starts 9 us late in a 35 us pass), so there are TICKET_CLASSES counters
        // in separate 128-B lines; workgroup b draws from counter b % TICKET_CLASSES and ticket t of class c is partition
        // t * TICKET_CLASSES + c.  The grid is persistent and every class has resident workgroups, so the lowest partition
        // that is not finished has either been drawn or will be drawn by a workgroup of its class that is free to do so:
        // a workgroup still only waits on partitions that are running or will run without needing a new slot.
        if (tid == 0) {
            const uint32_t cls = blockIdx.x % TICKET_CLASSES;
            s_part = __hip_atomic_fetch_add(ticket + cls * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * TICKET_CLASSES + cls;
        }
        for (int k = tid; k < WAVES * RADIX; k += THREADS) s_hist[k] = 0;
        __syncthreads();
        const uint32_t part = s_part;
        if (part >= numParts) break;
#ifdef GS_EXP_SORT_TIMELINE
        if (tid == 0 && part < 16384u) g_timeline[part * 16u + 0] = tl0;
#endif
        GS_TL(1);                                           // ticket taken, histogram cleared

        const uint32_t partBase = part * (uint32_t)PART;
        const uint32_t valid = min((uint32_t)PART, n - partBase);
        const uint32_t waveBase = partBase + (uint32_t)w * (64u * KPT);

        // ---- load keys: wave-striped, item (w,k,lane) has global index waveBase + k*64 + lane ----------
        // full partitions (all but the last) take the unconditional path: wave-uniform base + lane*4 + immediate
        uint32_t key[KPT];
        const bool full = valid == (uint32_t)PART;
        if (full) {
            const uint32_t* kp = keysIn + waveBase;
#pragma unroll
            for (int k = 0; k < KPT; ++k) key[k] = ldg32(kp + k * 64, (uint32_t)lane);
        } else {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t gi = waveBase + (uint32_t)k * 64u + lane;
                key[k] = (gi < n) ? ldg32(keysIn, gi) : 0xffffffffu; // tail dummies sort last and are never written
            }
        }

#ifdef GS_EXP_SORT_TIMELINE
        { uint32_t x = 0;
#pragma unroll
          for (int k = 0; k < KPT; ++k) x |= key[k];
          asm volatile("" :: "v"(x)); }                     // wait for the key loads before the timestamp
#endif
        GS_TL(2);                                           // keys arrived
        // ---- rank inside the wave: multi-split by 8 ballots, running per-wave LDS histogram ------------
        uint32_t pos[KPT];                                   // rank now, local position later
        uint32_t* wh = s_hist + w * RADIX;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t d = (key[k] >> shift) & digitMask;
            // m = lanes of this wave holding the same digit.  Per digit bit: sb = 0 / ~0 (v_bfe_i32), one ballot, and
            // m &= ~(ballot ^ sb) as ONE three-input bit op per 32-lane half (v_bitop3_b32, truth table 0x90): 4 VALU
            // instructions per bit where the generic select/xor/and sequence the compiler emits takes 10.
            uint32_t mlo = ~0u, mhi = ~0u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t sb = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
                const unsigned long long bal = __ballot((int)sb < 0);
                mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bal, sb, 0x90);
                mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bal >> 32), sb, 0x90);
            }
            const uint32_t lower = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // same-digit lanes below this one
            const uint32_t cnt = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
            const uint32_t pre = wh[d];
            __builtin_amdgcn_wave_barrier();
            if (lower == 0) wh[d] = pre + cnt;
            __builtin_amdgcn_wave_barrier();
            pos[k] = pre + lower;
            // Opaque to the optimiser: without it the digit and its LDS address are kept in registers per key for the
            // scatter below (3 extra VGPRs x KPT, 150 VGPRs total = 3 waves/SIMD); recomputing them costs 2 VALU ops.
            asm volatile("" : "+v"(key[k]));
        }
        __syncthreads();
        GS_TL(3);                                           // ranked

        // ---- partition digit counts, wave-exclusive offsets, local exclusive scan over digits --------
        // (threads >= RADIX only help with loads/stores; digit `tid` is owned by thread tid < RADIX)
        uint32_t total = 0, lbase = 0;
        unsigned long long* myStatus = status + (size_t)part * RADIX + (tid & (RADIX - 1));
        if (tid < RADIX) {
#pragma unroll
            for (int k = 0; k < WAVES; ++k) {
                const uint32_t c = s_hist[k * RADIX + tid];
                s_hist[k * RADIX + tid] = total;
                total += c;
            }
            // publish this partition's digit count right away (decoupled look-back: successors need only this)
            st_status(myStatus, pack_status(epoch, part == 0 ? FLAG_INCL : FLAG_AGG, total));
            // ... and add it to the aggregate of this partition's group of GROUP consecutive partitions: one 64-bit word per
            // (group, digit) = members published << 40 | sum of their counts, so a reader sees a consistent pair
            __hip_atomic_fetch_add(groupAgg + (size_t)(part / GROUP) * RADIX + tid, (1ull << 40) | (unsigned long long)total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t incl = wave_incl_scan(total, lane);
            if (lane == 63) s_wtot[w] = incl;
            lbase = incl - total;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
#pragma unroll
            for (int k = 0; k < RADIX / 64; ++k) wbase += (k < w) ? s_wtot[k] : 0u;
            lbase += wbase;
            s_lbase[tid] = lbase;

        }
        __syncthreads();

        // ---- scatter keys through LDS so that global writes are runs of equal digits -------------------
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t d = (key[k] >> shift) & digitMask;
            pos[k] = s_lbase[d] + wh[d] + pos[k];
            s_buf[pos[k]] = key[k];
        }
        // payloads: issued now (the key registers are dead), consumed after the key write-out
        uint32_t val[KPT];
        if (full) {
            const uint32_t* vp = valsIn + waveBase;
#pragma unroll
            for (int k = 0; k < KPT; ++k) val[k] = ldg32(vp + k * 64, (uint32_t)lane);
        } else {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t gi = waveBase + (uint32_t)k * 64u + lane;
                val[k] = (gi < n) ? ldg32(valsIn, gi) : 0u;
            }
        }
        GS_TL(4);                                           // keys scattered to LDS, payload loads issued
        // ---- look back over earlier partitions for digit `tid` (keys are parked in LDS by now, so the batch of
        //      status words below replaces the key registers instead of adding to them) ------------------------
        if (tid < RADIX) {
            uint32_t exclPrefix = 0;
            // A digit this partition does not hold needs no base (s_gbase[d] is never read) and its INCLUSIVE word is never
            // required by anyone (successors pass over the AGGREGATE 0 / use the group aggregates), so its look-back is
            // skipped: in the passes over the high key bytes almost every digit is empty almost everywhere.
            if (part > 0 && total > 0) {
                // Every partition of a pass is resident at once (persistent grid) and all of them finish ranking at about the
                // same time, so a plain decoupled look-back degenerates into a serial chain of INCLUSIVE hand-offs sweeping
                // over the partitions (measured: 23 us of a 39 us pass for 749 partitions).  Two levels remove the chain:
                //   1. walk the earlier partitions of the own group (< GROUP words, AGGREGATE or INCLUSIVE),
                //   2. then whole groups: the group's last partition if it is already INCLUSIVE, else the group aggregate
                //      once all GROUP members have added to it -- which depends on ranking only, not on anyone's look-back.
                // LB words are requested per round and consumed in order.
                constexpr int LB = GS_SORT_LOOKBACK_BATCH;
                const int grp = (int)(part / GROUP), grpStart = grp * GROUP;
                int q = (int)part - 1;
                uint32_t spins = 0;
                bool done = false;
#ifdef GS_EXP_SORT_TIMELINE
                uint32_t tlRounds = 0;
#endif
                while (!done && q >= grpStart) {                                   // ---- level 1: own group
#ifdef GS_EXP_SORT_TIMELINE
                    ++tlRounds;
#endif
                    unsigned long long sv[LB];
#pragma unroll
                    for (int b = 0; b < LB; ++b) {
                        const int qi = q - b;
                        sv[b] = qi >= grpStart ? ld_status(status + (size_t)qi * RADIX + tid) : 0ull;
                    }
                    int consumed = 0;
#pragma unroll
                    for (int b = 0; b < LB; ++b) {
                        if (done || consumed != b) continue;                         // stop at the first word that is not ready
                        const uint32_t e = (uint32_t)(sv[b] >> 34);
                        const uint32_t f = (uint32_t)(sv[b] >> 32) & 3u;
                        if (e == epoch && f != 0) {
                            exclPrefix += (uint32_t)sv[b];
                            consumed = b + 1;
                            if (f == (uint32_t)FLAG_INCL) done = true;
                        }
                    }
                    q -= consumed;
                    if (!done && consumed == 0) {
                        if (++spins > SPIN_LIMIT) { atomicOr(error, 1u); done = true; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                int j = grp - 1;
                while (!done && j >= 0) {                                          // ---- level 2: whole groups (all of them full)
#ifdef GS_EXP_SORT_TIMELINE
                    ++tlRounds;
#endif
                    constexpr int GB = LB / 2 > 0 ? LB / 2 : 1;
                    unsigned long long last[GB], agg[GB];
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        const int jj = j - b;
                        last[b] = jj >= 0 ? ld_status(status + ((size_t)(jj + 1) * GROUP - 1) * RADIX + tid) : 0ull;
                        agg[b] = jj >= 0 ? __hip_atomic_load(groupAgg + (size_t)jj * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
                    int consumed = 0;
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        if (done || consumed != b || j - b < 0) continue;
                        const uint32_t e = (uint32_t)(last[b] >> 34);
                        const uint32_t f = (uint32_t)(last[b] >> 32) & 3u;
                        if (e == epoch && f == (uint32_t)FLAG_INCL) {              // everything up to the end of group j-b
                            exclPrefix += (uint32_t)last[b];
                            consumed = b + 1;
                            done = true;
                        } else if ((uint32_t)(agg[b] >> 40) == (uint32_t)GROUP) {  // all members of group j-b have ranked
                            exclPrefix += (uint32_t)agg[b];
                            consumed = b + 1;
                        }
                    }
                    j -= consumed;
                    if (!done && consumed == 0) {
                        if (++spins > SPIN_LIMIT) { atomicOr(error, 1u); done = true; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                st_status(myStatus, pack_status(epoch, FLAG_INCL, exclPrefix + total));
#ifdef GS_EXP_SORT_TIMELINE
                if (tid == 0 && part < 16384u) { g_timeline[part * 16u + 10] = tlRounds; g_timeline[part * 16u + 11] = (unsigned long long)((int)part - 1 - q) + (unsigned long long)(grp - 1 - j) * 1000ull; g_timeline[part * 16u + 12] = spins; }
#endif
            }
            s_gbase[tid] = histExcl + exclPrefix - lbase;
        }
        GS_TL(5);                                           // look-back done (thread 0 = digit 0)
        __syncthreads();
        GS_TL(6);                                           // everyone's look-back done
        uint32_t dpack[(KPT + 3) / 4];
#pragma unroll
        for (int k = 0; k < (KPT + 3) / 4; ++k) dpack[k] = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t j = (uint32_t)tid + (uint32_t)k * THREADS;
            if (j < valid) {
                const uint32_t kk = s_buf[j];
                const uint32_t d = (kk >> shift) & digitMask;
                stg32(keysOut, s_gbase[d] + j, kk);
                dpack[k >> 2] |= d << (8 * (k & 3));
            }
        }
        __syncthreads();
        GS_TL(7);                                           // keys written out
#pragma unroll
        for (int k = 0; k < KPT; ++k) s_buf[pos[k]] = val[k];
        __syncthreads();
        GS_TL(8);                                           // payloads in LDS
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t j = (uint32_t)tid + (uint32_t)k * THREADS;
            if (j < valid) stg32(valsOut, s_gbase[(dpack[k >> 2] >> (8 * (k & 3))) & 255u] + j, s_buf[j]);
        }
        GS_TL(9);                                           // payload stores issued
    }
}

__global__ void copy_pairs_kernel(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs, uint32_t* __restrict__ kd,
                                  uint32_t* __restrict__ vd, uint32_t nImm, const uint32_t* nPtr) {
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { kd[i] = ks[i]; vd[i] = vs[i]; }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

} // namespace

#ifdef GS_EXP_SORT_TIMELINE
extern "C" int32_t gs_debug_read_sort_timeline(void* out, size_t bytes) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), bytes) == hipSuccess ? 0 : -2;
}
#endif

int32_t sort_state_create(gs_context* ctx, SortState& st, uint32_t maxCount) {
    (void)ctx;
    st.maxCount = maxCount;
    st.maxParts = div_up(maxCount > 0 ? maxCount : 1, PART);
    GS_HIP(hipMalloc((void**)&st.altKeys, (size_t)(maxCount + 16) * 4));
    GS_HIP(hipMalloc((void**)&st.altVals, (size_t)(maxCount + 16) * 4));
    GS_HIP(hipMalloc((void**)&st.status, (size_t)st.maxParts * RADIX * 8));
    GS_HIP(hipMemsetAsync(st.status, 0, (size_t)st.maxParts * RADIX * 8, ctx->stream));
    st.maxGroups = div_up(st.maxParts, (uint32_t)GROUP);
    GS_HIP(hipMalloc((void**)&st.groupAgg, (size_t)4 * st.maxGroups * RADIX * 8));
    return GS_OK;
}

void sort_state_destroy(SortState& st) {
    if (st.altKeys) (void)hipFree(st.altKeys);
    if (st.altVals) (void)hipFree(st.altVals);
    if (st.status) (void)hipFree(st.status);
    if (st.groupAgg) (void)hipFree(st.groupAgg);
    st = SortState();
}

int32_t enqueue_set_indices(gs_context* ctx, uint32_t* order, uint32_t n) {
    hipLaunchKernelGGL(set_indices_kernel, dim3(div_up(n, 1024)), dim3(1024), 0, ctx->stream, order, n);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

uint32_t sort_group_words(uint32_t nUpper, int passes) { return (uint32_t)passes * div_up(div_up(max(nUpper, 1u), PART), (uint32_t)GROUP) * RADIX; }

int32_t enqueue_calc_distances(gs_context* ctx, hipStream_t stream, const gsm::AssetView& a, const uint32_t* order, const float* m, uint32_t* keys,
                               SortControl* control, SortControl* nextControl, uint32_t n, SortState& st) {
    // `control` was zeroed by the previous sort's launch of this kernel (or at creation); this launch zeroes `nextControl`
#ifndef GS_DIST_BLOCKS_PER_CU
#define GS_DIST_BLOCKS_PER_CU 2      // a narrow window of sorted positions per XCD keeps the gathered sectors in its L2 (measured: 2 beats 4 and 8)
#endif
    const uint32_t grid = (max(1u, min(div_up(n, 256u * GS_DIST_ILP), (uint32_t)ctx->cuCount * GS_DIST_BLOCKS_PER_CU)) + 7u) & ~7u;
    hipLaunchKernelGGL(calc_distances_kernel, dim3(grid), dim3(256), 0, stream, a, order, m[8], m[9], m[10], m[11], keys,
                       control->hist, n, st.groupAgg, sort_group_words(n, 4), (uint32_t*)nextControl);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_histogram(gs_context* ctx, hipStream_t stream, const uint32_t* keys, uint32_t n, const uint32_t* nPtr, int passes, uint32_t lastMask, SortControl* control,
                          SortState& st) {
    GS_HIP(hipMemsetAsync(control, 0, sizeof(SortControl), stream));
    const uint32_t grid = max(1u, min(div_up(n, 256), (uint32_t)ctx->cuCount * 4u));
    hipLaunchKernelGGL(histogram_kernel, dim3(grid), dim3(256), 0, stream, keys, n, nPtr, passes, lastMask, control->hist, st.groupAgg, sort_group_words(n, passes));
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_sort_passes(gs_context* ctx, hipStream_t stream, SortState& st, SortControl* control, uint32_t* keys, uint32_t* vals, uint32_t nUpper,
                            const uint32_t* nPtr, int passes, uint32_t lastMask, gs_renderer* profR, int evFirst) {
    if (passes < 1 || passes > 4) return fail(GS_ERR_INVALID_ARGUMENT, "sort passes");
    if (nUpper > st.maxCount) return fail(GS_ERR_INVALID_ARGUMENT, "sort count exceeds sorter capacity");
    if (nUpper == 0) return GS_OK;
    const uint32_t parts = div_up(nUpper, PART);
    // persistent grid: no more workgroups than are resident at once (3 per CU at <= 80 VGPRs / 43 KB LDS), a multiple of
    // the ticket classes so that every class is served
    const uint32_t capacity = max(((uint32_t)ctx->cuCount * 3u / TICKET_CLASSES) * TICKET_CLASSES, TICKET_CLASSES);
    const uint32_t grid = min(div_up(parts, TICKET_CLASSES) * TICKET_CLASSES, capacity);
    uint32_t *ks = keys, *vs = vals, *kd = st.altKeys, *vd = st.altVals;
    const uint32_t groups = div_up(parts, (uint32_t)GROUP);
    // st.groupAgg[passes][groups][256] accumulates: it was zeroed by the kernel that produced the keys / their histograms
    if (profR && evFirst >= 0) prof_record(profR, evFirst, stream);
    for (int p = 0; p < passes; ++p) {
        uint32_t epoch = (++st.epoch) & 0x3fffffffu;
        if (epoch == 0) {   // 30-bit epoch wrapped: wipe the status array (it may hold every old epoch), restart at 1
            GS_HIP(hipMemsetAsync(st.status, 0, (size_t)st.maxParts * RADIX * 8, stream));
            st.epoch = epoch = 1;
        }
        hipLaunchKernelGGL(onesweep_kernel, dim3(grid), dim3(THREADS), 0, stream, ks, vs, kd, vd, control->hist + RADIX * p,
                           st.status, st.groupAgg + (size_t)p * groups * RADIX, control->tickets[p], &control->error, nUpper, nPtr, (uint32_t)(8 * p), epoch, p == passes - 1 ? lastMask : 255u);
        uint32_t* t = ks; ks = kd; kd = t;
        t = vs; vs = vd; vd = t;
    }
    if (profR && evFirst >= 0) prof_record(profR, evFirst + 1, stream);
    if (ks != keys) {
        hipLaunchKernelGGL(copy_pairs_kernel, dim3(max(1u, min(div_up(nUpper, 256), (uint32_t)ctx->cuCount * 8u))), dim3(256), 0,
                           stream, ks, vs, keys, vals, nUpper, nPtr);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

} // namespace gs
