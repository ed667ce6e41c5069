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
// gfx950 specifics: 64-lane waves -- ranking is a wave-level multi-split from one __ballot() per digit bit (folded
// with one v_bitop3 per 32-lane half) and v_mbcnt below the lane; per-wave digit histograms live in LDS; inter-workgroup
// look-back words are self-tagged granules -- a 4-byte {epoch:18, count:14} word per (partition, digit), published four
// digits at a time with one 16-byte write-through (sc1) store, and 8-byte {epoch, value} / {members, sum} words per
// (group of 32 partitions, digit) -- written/read with AGENT-scope relaxed accesses (the per-XCD L2s are not coherent, see
// MI355X_MICROARCH.md "inter-workgroup visibility"; every granule carries its own tag so no fence is needed and no
// status memset between passes: each pass uses a fresh epoch).
// Partitions are handed out by atomic tickets (16 counters in separate cache lines) inside a persistent grid, so a
// workgroup only ever waits on partitions that are running; all partitions of a pass are resident at once, so the
// look-back goes through per-group aggregates (32 partitions) instead of a chain of INCLUSIVE hand-offs; every spin is
// bounded and reports GS_ERR_SORT_TIMEOUT instead of hanging.  Design notes and measurements: DESIGN.md section 4.1.
//
// Depth sort of a frame (gs_renderer_sort): sort_keys_kernel writes key[s] for every splat s in INDEX order (coalesced
// position reads; the four digit histograms do not depend on the order) and the first Onesweep pass gathers
// key[order[i]] itself (GATHER instantiation), so CSCalcDistances' random access is a single 4-byte gather hidden
// inside a pass that is resident anyway, instead of a kernel of its own that waits on a position + ChunkInfo gather.
#include "gs_common.h"
#include <cstdlib>

namespace gs {

namespace {

constexpr int RADIX = 256;
constexpr int THREADS = 512;
constexpr int WAVES = THREADS / 64;
// Two shapes of a pass, chosen per sort by the expected key count (enqueue_sort_passes):
//   A  16 keys per thread, 8,192-key partitions, <= 80 VGPRs: three workgroups per CU -- up to 768 partitions (6.3 M keys) in ONE round;
//   B  20 keys per thread, 10,240-key partitions, <= 128 VGPRs: two workgroups per CU -- longer runs per digit (160-byte stores), a fifth
//      fewer status words and look-back steps: 8 % faster on 50 M keys, 9 % slower on 6 M (599 partitions on 512 slots = two rounds).
//   C   8 keys per thread, 4,096-key partitions (round 5): for SMALL sorts -- the depth sort of the visible splats only (2.2 M keys at C2 =
//      267 partitions of shape A on 768 slots: a third of the chip busy, the pass as long as one partition's chain).  Half the keys per
//      partition halve that chain's load / rank / scatter phases and put two partitions on every CU.  Plain 8-bit passes only.
// Sizing (status words, group words) follows the smallest partition a sort may use (SortState::partMin: shape C for the depth sort, A otherwise).
constexpr int KPT_A = 16, KPT_B = 20, KPT_C = 8;
constexpr int PART_A = THREADS * KPT_A, PART_B = THREADS * KPT_B, PART_C = THREADS * KPT_C;
constexpr int PART_MIN = PART_C;
constexpr uint32_t kBigSortKeys = 24u << 20;          // expected keys above which shape B is used (four rounds of shape A)
constexpr uint32_t kSmallSortKeys = 7u << 19;         // expected keys (3.67 M) up to which shape C is used: its partitions then fit one round of the grid
constexpr uint32_t SPIN_LIMIT = 1u << 24;
constexpr uint32_t TICKET_CLASSES = 16;        // partition-ticket counters per pass (one 128-B line each)
constexpr int GROUP = 32;                     // partitions per look-back group (~sqrt of the partition count of a 6 M key sort)

// per-(partition, digit) status word: {epoch:18 | count:14}; valid for a pass iff its epoch field equals the pass's epoch
constexpr uint32_t COUNT_BITS = 14, EPOCH_MASK = (1u << 18) - 1u;
static_assert(PART_A < (1 << COUNT_BITS) && PART_B < (1 << COUNT_BITS), "a partition's digit count must fit the status word");
__device__ __forceinline__ uint32_t pack_status(uint32_t epoch, uint32_t count) { return (epoch << COUNT_BITS) | count; }
__device__ __forceinline__ uint32_t ld_status(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_word64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte write-through store (agent scope): four status words in ONE fabric write (scalar sc1 stores cost one fabric
// write each whatever their size, MI355X_MICROARCH.md).  The trailing s_nop keeps hipcc's hazard tracking honest.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sc1(uint32_t* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// 32-bit element index on a wave-uniform base pointer: lets the compiler use the SGPR-base + 32-bit-VGPR-offset
// addressing form instead of a 64-bit address per access (arrays are < 4 GB: counts are capped at 2^30)
__device__ __forceinline__ uint32_t ldg32(const uint32_t* base, uint32_t idx) { return *(const uint32_t*)((const char*)base + (size_t)(idx << 2)); }
// the random 4-byte gather of the first depth-sort pass: a plain (L1-allocating) load -- the `nt` form, which bypasses the
// CU's L1, was measured 45 us slower per pass: neighbours of a sector do meet in L1
__device__ __forceinline__ uint32_t gather32(const uint32_t* base, uint32_t idx) { return ldg32(base, idx); }
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

// stand-alone histogram (gs_sorter path): `passes` digit histograms of keys[0..n)
__global__ __launch_bounds__(256) void histogram_kernel(const uint32_t* __restrict__ keys, uint32_t nImm, const uint32_t* nPtr,
                                                        int passes, uint32_t lastMask, uint32_t* __restrict__ hist,
                                                        unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords, uint32_t copies) {
    GS_CHAIN_PRIORITY();
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
    uint32_t* myHist = hist + (blockIdx.x % copies) * (uint32_t)kHistStride;      // SortControl::hist: one of the copies
    for (int j = threadIdx.x; j < passes * RADIX; j += 256) {
        const uint32_t c = s_h[j];
        if (c) atomicAdd(&myHist[j], c);
    }
}

// Sort keys of all splats in INDEX order + the 4 digit histograms of the Onesweep sort (they do not depend on the order):
// key[s] = FloatToSortableUint(dot(matrix row 2, (pos[s], 1))), CSCalcDistances' arithmetic (SplatUtilities.compute:76-81)
// with the gather through _SplatSortKeys left to the first sort pass.  Streaming: 4..12 B in, 4 B out per splat.  One
// 1024-thread workgroup per CU (few workgroups = few flushes of the LDS histograms into the 1024 global bins, whose
// same-address atomics serialise), each thread keeps ILP chunks in flight; a quarter of a workgroup = one 256-splat chunk,
// so ChunkInfo is wave-uniform.
constexpr uint32_t kKeysIlp = 4;             // 256-splat chunks in flight per quarter workgroup of sort_keys_kernel
template <int POSFMT>
__global__ __launch_bounds__(1024) void sort_keys_kernel(gsm::AssetView a, float m20, float m21, float m22, float m23,
                                                         uint32_t* __restrict__ keyBySplat, uint32_t* __restrict__ hist, uint32_t n,
                                                         unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords,
                                                         uint32_t* __restrict__ nextControl, uint32_t copies) {
    __shared__ uint32_t s_h[4 * RADIX];
    for (int j = threadIdx.x; j < 4 * RADIX; j += 1024) s_h[j] = 0;
    for (uint32_t j = blockIdx.x * 1024u + threadIdx.x; j < groupAggWords; j += gridDim.x * 1024u) groupAgg[j] = 0ull;   // the sort passes accumulate into it
    // the control block (histograms, tickets, error) of the NEXT sort: the two blocks alternate, so no memset launch per sort
    for (uint32_t j = blockIdx.x * 1024u + threadIdx.x; j < (uint32_t)(sizeof(SortControl) / 4); j += gridDim.x * 1024u) nextControl[j] = 0u;
    __syncthreads();
    constexpr uint32_t ILP = kKeysIlp;
    const uint32_t chunks = (n + 255u) >> 8;
    const uint32_t sub = threadIdx.x >> 8, t = threadIdx.x & 255u;
    const uint32_t step = gridDim.x * (4u * ILP);
    // two-stage software pipeline: the positions of the next ILP chunks are in flight while the keys of the current ones go
    // through the LDS histograms (the LDS atomic unit and the memory pipe then work at the same time instead of in turns)
    // (raw dwords first, decoded only when they are consumed: every load of a stage is issued before anything waits)
    typedef gsm::RawVec<POSFMT> Raw;
    // The chunk's position bounds (ChunkInfo.posX/Y/Z: 24 bytes at +16) travel through the same pipeline as the positions, as
    // VECTOR loads of a wave-uniform address.  As scalar loads issued where they are used they were the kernel's critical path:
    // one exposed ~1 us round trip per chunk, four per iteration (the position loads were prefetched, these were not).
    struct Staged { Raw raw; uint4 bx; uint2 bz; };              // bx = posX.min, posX.max, posY.min, posY.max; bz = posZ.min, posZ.max
    const bool chunked = a.chunkCount != 0u;
    const uint8_t* cbase = chunked ? a.chunk : (const uint8_t*)keyBySplat;            // (no chunks: any readable 64 bytes, never used)
    const uint32_t lastChunk = chunked ? a.chunkCount - 1u : 0u;
    auto load = [&](uint32_t c0, Staged (&p)[ILP]) {
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            // unconditional loads with a clamped index: a load inside a divergent branch is waited for at the end of the branch
            const uint32_t ci = min(c0 + k * 4u + sub, chunks - 1u);
            const uint32_t idx = min(ci * 256u + t, n - 1u);
            p[k].raw = gsm::LoadRawT<POSFMT>(a.pos, (uint64_t)idx * gsm::vecStrideT<POSFMT>());
            const uint8_t* c = cbase + (size_t)min(ci, lastChunk) * 64u;
            p[k].bx = *(const uint4*)(c + 16);
            p[k].bz = *(const uint2*)(c + 32);
        }
    };
    Staged cur[ILP], nxt[ILP];
    uint32_t c0 = blockIdx.x * (4u * ILP);
    if (c0 < chunks) load(c0, cur);
    for (; c0 < chunks; c0 += step) {
        const bool more = c0 + step < chunks;                                            // workgroup-uniform
        if (more) load(c0 + step, nxt);
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            const uint32_t ci = c0 + k * 4u + sub;
            const uint32_t idx = ci * 256u + t;
            if (idx >= n) continue;
            gsm::V3 pos = gsm::DecodeRawT<POSFMT>(cur[k].raw, (uint64_t)idx * gsm::vecStrideT<POSFMT>());
            if (chunked && ci <= lastChunk) {                    // LoadSplatPos' chunk de-normalisation (ChunkLerpPos), same expressions
                pos.x = gsm::lerpf(gsm::u2f(cur[k].bx.x), gsm::u2f(cur[k].bx.y), pos.x);
                pos.y = gsm::lerpf(gsm::u2f(cur[k].bx.z), gsm::u2f(cur[k].bx.w), pos.y);
                pos.z = gsm::lerpf(gsm::u2f(cur[k].bz.x), gsm::u2f(cur[k].bz.y), pos.z);
            }
            const uint32_t key = gsm::SortKeyOf(pos, m20, m21, m22, m23);
            keyBySplat[idx] = key;
            lds_hist_add(s_h, key & 255u);
            lds_hist_add(s_h + RADIX, (key >> 8) & 255u);
            lds_hist_add(s_h + 2 * RADIX, (key >> 16) & 255u);
            lds_hist_add(s_h + 3 * RADIX, key >> 24);
        }
        if (more) {
#pragma unroll
            for (uint32_t k = 0; k < ILP; ++k) cur[k] = nxt[k];
        }
    }
    __syncthreads();
    uint32_t* myHist = hist + (blockIdx.x % copies) * (uint32_t)kHistStride;      // SortControl::hist: one of the copies
    for (int j = threadIdx.x; j < 4 * RADIX; j += 1024) {
        const uint32_t c = s_h[j];
        if (c) atomicAdd(&myHist[j], c);
    }
}

// One Onesweep pass: reads (keysIn, valsIn), writes (keysOut, valsOut) stably partitioned by digit (key>>shift)&mask.
// BITS = digit width (8 for the depth sort; 6..8 for the tile-pair sort, whose keys have 12..16 significant bits: fewer
// ballots per key, fewer status words per partition).  GATHER = first pass of a frame's depth sort: the key of position
// i is keysIn[valsIn[i]] (keysIn = keys by splat index, valsIn = the previous order) -- CSCalcDistances' gather.
// Register diet (the kernel is latency-bound, so resident waves matter): payloads are loaded only after the keys
// have left the registers for LDS, local positions overwrite the ranks, and the digit of each output slot is kept
// packed 4 per register instead of a 32-bit global index per slot.
constexpr int kLookbackBatch = 16;           // status words requested per look-back round
constexpr int kMinWavesA = 6;                // shape A: <= 80 VGPRs = three 512-thread workgroups per CU (a handful of loop-invariant values spill to scratch)
template <int BITS, bool GATHER, int KPT>
__global__ __launch_bounds__(THREADS, KPT == KPT_B ? 4 : kMinWavesA) void onesweep_kernel(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
                                                           uint32_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut,
                                                           const uint32_t* __restrict__ hist, uint32_t* status,
                                                           unsigned long long* groupAgg, unsigned long long* groupIncl, uint32_t* ticket, uint32_t* error,
                                                           uint32_t nImm, const uint32_t* nPtr, uint32_t shift, uint32_t epoch, uint32_t digitMask, uint32_t histCopies, uint32_t gatherXcd) {
    GS_CHAIN_PRIORITY();
    constexpr int PART = THREADS * KPT;              // keys per partition
    constexpr int RDX = 1 << BITS;                   // digits of this pass
    constexpr int DW = (RDX + 63) / 64;              // waves that own digits
    __shared__ uint32_t s_hist[WAVES * RDX];         // per-wave digit counts -> wave-exclusive offsets
    __shared__ uint32_t s_lbase[RDX];                // exclusive digit offsets inside the partition
    __shared__ uint32_t s_gbase[RDX];                // global index of local slot j with digit d = s_gbase[d] + j
    __shared__ __attribute__((aligned(16))) uint32_t s_pub[RDX];   // this partition's status words, published 4 per store
    __shared__ uint32_t s_buf[PART];
    __shared__ uint32_t s_wtot[DW];
    __shared__ uint32_t s_htot[DW];
    __shared__ uint32_t s_part;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    const uint32_t numParts = (n + PART - 1) / PART;
    // partitions per XCD block of the GATHER pass: an eighth of the input, at most a look-back group
    // partitions per XCD block of the GATHER pass: the input is cut into 8 r equal blocks, r = the fewest rounds for which a
    // block is no larger than what one XCD runs at once (32 CUs x 3 workgroups), so that every XCD gets the same number of
    // blocks (measured on C2, 749 partitions: blocks of 94 -> 156 us per depth sort, 32 -> 165, 64 (unbalanced) -> 179)
    constexpr uint32_t kXcdMaxBlock = KPT == KPT_A ? 96 : 64;
    const uint32_t xcdRounds = (numParts + 8u * kXcdMaxBlock - 1u) / (8u * kXcdMaxBlock);
    const uint32_t xcdBlock = max(1u, (numParts + 8u * max(xcdRounds, 1u) - 1u) / (8u * max(xcdRounds, 1u)));
    (void)xcdBlock;

    // global exclusive digit offsets = exclusive scan of this pass's histogram (raw counts, accumulated by the key
    // generation / binning / histogram kernel): every workgroup scans it for itself instead of a separate 1-workgroup launch
    uint32_t histExcl = 0;
    bool digitLive = false;                            // some key of the whole input holds this digit
    if (tid < RDX) {
        uint32_t c = 0;
#pragma unroll
        for (int rp = 0; rp < kHistReplicas; ++rp) c += (uint32_t)rp < histCopies ? hist[rp * kHistStride + tid] : 0u;      // the copies of SortControl::hist in use (independent loads)
        digitLive = c != 0u;
        const uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) s_htot[w] = incl;
        histExcl = incl - c;
    }
    bool quadLive = false;                             // threads < RDX/4 publish four digits' status words at a time
    if (tid < RDX / 4) {
        uint32_t any = 0;
#pragma unroll
        for (int rp = 0; rp < kHistReplicas; ++rp) if ((uint32_t)rp < histCopies) { const uint4 h4 = ((const uint4*)(hist + rp * kHistStride))[tid]; any |= h4.x | h4.y | h4.z | h4.w; }
        quadLive = any != 0u;
    }
    __syncthreads();
    if (tid < RDX)
        for (int k = 0; k < w; ++k) histExcl += s_htot[k];

    // Which partition a workgroup takes.  A partition waits (look-back) on EVERY partition before it, so forward progress needs the lowest unfinished
    // partition to be held by a workgroup that is RUNNING -- whatever else shares the GPU (another frame in flight on another stream, another process: kernels
    // that may themselves be spinning on workgroups of theirs that cannot be dispatched while ours hold the slots).
    //   The grid covers every partition (one round: a 6 M-key depth pass, the pair sort of a 6 M-pair frame): workgroup b takes partition b, no atomic, and
    //             exits when it is done.  The dispatcher starts workgroups in blockIdx order, so a workgroup only ever waits on workgroups started before it, and
    //             those need nothing from anyone to finish.
    //   Otherwise (a persistent grid walking more partitions than it has workgroups): EVERY partition, the first one too, comes from ONE counter -- whoever is
    //             running claims the lowest unclaimed partition.  (A static first round is not safe here: a workgroup that has not been dispatched yet owns a
    //             partition nobody else can take, the running ones never exit -- they claim further partitions and end up spinning on that one -- and two such
    //             kernels on two streams can hold each other's wave slots for good.  Found with frames in flight at C3 / C5 / C2d / C4, whose pair sorts walk
    //             more partitions than fit: bounded spins expired, GS_ERR_SORT_TIMEOUT.  Rounds 1-5 spread the ~768 simultaneous requests at the head of the
    //             kernel over 16 counters; per-class counters stall the same way when a class has no running workgroup -- found by two processes on one GPU.
    //             The requests are 12 ns apart on one address: the 768th workgroup starts 9 us late, behind a look-back chain that reaches it later still.)
    // The GATHER pass of a frame's depth sort may instead deal whole blocks of consecutive partitions to the XCDs (gatherXcd != 0: 16 counters, two per XCD) so
    // that the 16 keys of a gathered 64-byte sector meet in one L2 -- 25 us faster per sort at C2, but only deadlock-free while no other kernel that spins
    // shares the GPU; gs_context_set_shared_gpu / GSPLAT_SHARED_GPU=1 selects the dependency-ordered form for it too.
    //   gatherXcd bit 1 (the context shares the GPU with nobody: gs_shared_gpu() is false -- the condition the XCD deal has): the persistent grid's FIRST
    //             partitions are static after all (workgroup b takes partition b, the counter hands out the rest): without a second spinning kernel the
    //             not-yet-dispatched owners of first-round partitions get their slots as soon as anything exits, and the head of the kernel does not queue
    //             on one address (bin_emit at C2: 78 us against 94; the pair passes of C3 / C4).  One such kernel beside any number of one-counter kernels is
    //             still safe -- those finish with whatever workgroups they have and release their slots.
    const bool xcdDeal = GATHER && (gatherXcd & 1u);
    const bool staticFirst = (gatherXcd & 2u) != 0u;
    const bool oneRound = !xcdDeal && gridDim.x >= numParts;      // (the XCD deal is not the identity: its first tickets need not cover every partition)
    for (uint32_t round = 0;; ++round) {
        if (oneRound && round) break;
        __syncthreads();                                    // previous partition's LDS reads are finished
        if (tid == 0) {
            uint32_t p;
            if (xcdDeal) {
                // Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md) = cls % 8: every XCD takes whole blocks of `xb` consecutive partitions (block j belongs
                // to XCD j % 8; the two ticket classes of an XCD alternate inside its blocks): the other 15 requests for a sector then arrive at the L2 that
                // already holds it.  Monotonic per class, a bijection.
                const uint32_t cls = blockIdx.x % TICKET_CLASSES;
                const uint32_t t = __hip_atomic_fetch_add(ticket + cls * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t xb = xcdBlock;
                const uint32_t x = cls & 7u, u = t * 2u + (cls >> 3);       // u-th partition of XCD x
                p = ((u / xb) * 8u + x) * xb + (u % xb);
            } else {
                p = (oneRound || (staticFirst && round == 0u)) ? blockIdx.x
                                                               : (staticFirst ? gridDim.x : 0u) + __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_part = p;
        }
        for (int k = tid; k < WAVES * RDX; k += THREADS) s_hist[k] = 0;
        __syncthreads();
        const uint32_t part = s_part;
        if (part >= numParts) break;

        const uint32_t partBase = part * (uint32_t)PART;
        const uint32_t valid = min((uint32_t)PART, n - partBase);
        const uint32_t waveBase = partBase + (uint32_t)w * (64u * KPT);

        // ---- load keys: wave-striped, item (w,k,lane) has global index waveBase + k*64 + lane ----------
        // full partitions (all but the last) take the unconditional path: wave-uniform base + lane*4 + immediate
        uint32_t key[KPT];
        const bool full = valid == (uint32_t)PART;
        if (GATHER) {
            // the previous order first (coalesced), then one independent 4-byte gather per key: KPT of them in flight per thread
            if (full) {
                const uint32_t* vp = valsIn + waveBase;
#pragma unroll
                for (int k = 0; k < KPT; ++k) key[k] = ldg32(vp + k * 64, (uint32_t)lane);
#pragma unroll
                for (int k = 0; k < KPT; ++k) key[k] = gather32(keysIn, key[k]);
            } else {
#pragma unroll
                for (int k = 0; k < KPT; ++k) {
                    const uint32_t gi = waveBase + (uint32_t)k * 64u + lane;
                    key[k] = (gi < n) ? gather32(keysIn, ldg32(valsIn, gi)) : 0xffffffffu;
                }
            }
        } else if (full) {
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

        // ---- rank inside the wave: multi-split by one ballot per digit bit, running per-wave LDS histogram ------------
        uint32_t pos[KPT];                                   // rank now, local position later
        uint32_t* wh = s_hist + w * RDX;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t d = (key[k] >> shift) & digitMask;
            // m = lanes of this wave holding the same digit.  Per digit bit: sb = 0 / ~0 (v_bfe_i32), one ballot, and
            // m &= ~(ballot ^ sb) as ONE three-input bit op per 32-lane half (v_bitop3_b32, truth table 0x90): 4 VALU
            // instructions per bit where the generic select/xor/and sequence the compiler emits takes 10.
            uint32_t mlo = ~0u, mhi = ~0u;
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
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

        // ---- partition digit counts, wave-exclusive offsets, local exclusive scan over digits --------
        // (threads >= RDX only help with loads/stores; digit `tid` is owned by thread tid < RDX)
        uint32_t total = 0, lbase = 0;
        if (tid < RDX) {
#pragma unroll
            for (int k = 0; k < WAVES; ++k) {
                const uint32_t c = s_hist[k * RDX + tid];
                s_hist[k * RDX + tid] = total;
                total += c;
            }
            s_pub[tid] = pack_status(epoch, total);
            // add the count to the aggregate of this partition's group of GROUP consecutive partitions: one 64-bit word per
            // (group, digit) = members published << 40 | sum of their counts, so a reader sees a consistent pair.  A digit that
            // no key of the whole input holds is never looked up by anyone: nothing is published for it.
            if (digitLive)
                __hip_atomic_fetch_add(groupAgg + (size_t)(part / GROUP) * RADIX + tid, (1ull << 40) | (unsigned long long)total, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t incl = wave_incl_scan(total, lane);
            if (lane == 63) s_wtot[w] = incl;
            lbase = incl - total;
        }
        __syncthreads();
        if (tid < RDX) {
            uint32_t wbase = 0;
#pragma unroll
            for (int k = 0; k < DW; ++k) wbase += (k < w) ? s_wtot[k] : 0u;
            lbase += wbase;
            s_lbase[tid] = lbase;
        }
        // publish this partition's digit counts (decoupled look-back: successors need only this), four digits per 16-byte
        // write-through store; quads of digits that no key of the whole input holds are never looked up: skipped
        if (tid < RDX / 4 && quadLive) st16_sc1(status + (size_t)part * RDX + (size_t)tid * 4, *(const u32x4*)&s_pub[tid * 4]);
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
        // ---- look back over earlier partitions for digit `tid` (keys are parked in LDS by now, so the batch of
        //      status words below replaces the key registers instead of adding to them) ------------------------
        if (tid < RDX) {
            uint32_t exclPrefix = 0;
            // A digit this partition does not hold needs no base (s_gbase[d] is never read), so its look-back is skipped:
            // in the passes over the high key bytes almost every digit is empty almost everywhere.
            // (digitLive: the 0xffffffff dummies that pad the last partition are counted in `total` but are not part of the input;
            // nobody publishes a digit that only they hold.)
            if (part > 0 && total > 0 && digitLive) {
                // Every partition of a pass is resident at once (persistent grid) and all of them finish ranking at about the
                // same time, so a plain decoupled look-back degenerates into a serial chain of INCLUSIVE hand-offs sweeping
                // over the partitions (measured: 23 us of a 39 us pass for 749 partitions).  Two levels remove the chain:
                //   1. sum the counts of the earlier partitions of the own group (< GROUP status words),
                //   2. then whole groups, nearest first: the group's inclusive prefix if its last partition has published one,
                //      else the group aggregate once all GROUP members have added to it -- which depends on ranking only,
                //      not on anyone's look-back.
                // LB words are requested per round and consumed in order.
                constexpr int LB = kLookbackBatch;
                const int grp = (int)(part / GROUP), grpStart = grp * GROUP;
                int q = (int)part - 1;
                uint32_t spins = 0;
                bool done = false;
                while (!done && q >= grpStart) {                                   // ---- level 1: own group
                    uint32_t sv[LB];
#pragma unroll
                    for (int b = 0; b < LB; ++b) {
                        const int qi = q - b;
                        sv[b] = qi >= grpStart ? ld_status(status + (size_t)qi * RDX + tid) : 0u;
                    }
                    int consumed = 0;
#pragma unroll
                    for (int b = 0; b < LB; ++b) {
                        if (consumed != b || q - b < grpStart) continue;             // stop at the first word that is not ready
                        if ((sv[b] >> COUNT_BITS) == epoch) {
                            exclPrefix += sv[b] & ((1u << COUNT_BITS) - 1u);
                            consumed = b + 1;
                        }
                    }
                    q -= consumed;
                    if (consumed == 0) {
                        if (++spins > SPIN_LIMIT) { atomicOr(error, 1u); done = true; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                int j = grp - 1;
                while (!done && j >= 0) {                                          // ---- level 2: whole groups (all of them full)
                    constexpr int GB = 4;
                    unsigned long long incl[GB], agg[GB];
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        const int jj = j - b;
                        incl[b] = jj >= 0 ? ld_word64(groupIncl + (size_t)jj * RADIX + tid) : 0ull;
                        agg[b] = jj >= 0 ? ld_word64(groupAgg + (size_t)jj * RADIX + tid) : 0ull;
                    }
                    int consumed = 0;
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        if (done || consumed != b || j - b < 0) continue;
                        if ((uint32_t)(incl[b] >> 32) == epoch) {                  // everything up to the end of group j-b
                            exclPrefix += (uint32_t)incl[b];
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
            }
            // the last partition of a group that knows its prefix publishes the group's inclusive prefix: a shortcut for
            // every later group's level 2 (those that find it stop there; those that do not use the aggregates)
            if ((part % GROUP) == (uint32_t)(GROUP - 1) && total > 0 && digitLive)
                st_word64(groupIncl + (size_t)(part / GROUP) * RADIX + tid, ((unsigned long long)epoch << 32) | (unsigned long long)(exclPrefix + total));
            s_gbase[tid] = histExcl + exclPrefix - lbase;
        }
        __syncthreads();
        uint32_t dpack[(KPT + 3) / 4];
#pragma unroll
        for (int k = 0; k < (KPT + 3) / 4; ++k) dpack[k] = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t j = (uint32_t)tid + (uint32_t)k * THREADS;
            if (j < valid) {
                const uint32_t kk = s_buf[j];
                const uint32_t d = (kk >> shift) & digitMask;
                if (keysOut) stg32(keysOut, s_gbase[d] + j, kk);       // (null: the last pass of a depth sort whose sorted keys nobody reads)
                dpack[k >> 2] |= d << (8 * (k & 3));
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KPT; ++k) s_buf[pos[k]] = val[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t j = (uint32_t)tid + (uint32_t)k * THREADS;
            if (j < valid) stg32(valsOut, s_gbase[(dpack[k >> 2] >> (8 * (k & 3))) & 255u] + j, s_buf[j]);
        }
    }
}

__global__ void copy_pairs_kernel(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs, uint32_t* __restrict__ kd,
                                  uint32_t* __restrict__ vd, uint32_t nImm, const uint32_t* nPtr) {
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { kd[i] = ks[i]; vd[i] = vs[i]; }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

} // namespace


int32_t sort_state_create(gs_context* ctx, SortState& st, uint32_t maxCount, bool smallPartitions) {
    if (maxCount > kSortMaxCount) return fail(GS_ERR_INVALID_ARGUMENT, "sort capacity above 2^30 keys");
    st.maxCount = maxCount;
    st.partMin = smallPartitions ? (uint32_t)PART_MIN : (uint32_t)PART_A;       // the smallest partition a pass of this sort may use: sizes the status / group words
    st.maxParts = div_up(maxCount > 0 ? maxCount : 1, st.partMin);
    GS_HIP(hipMalloc((void**)&st.altKeys, ((size_t)maxCount + 16) * 4));
    GS_HIP(hipMalloc((void**)&st.altVals, ((size_t)maxCount + 16) * 4));
    GS_HIP(hipMalloc((void**)&st.status, (size_t)st.maxParts * RADIX * 4));
    GS_HIP(hipMemsetAsync(st.status, 0, (size_t)st.maxParts * RADIX * 4, ctx->stream));
    st.maxGroups = div_up(st.maxParts, (uint32_t)GROUP);
    GS_HIP(hipMalloc((void**)&st.groupAgg, (size_t)4 * st.maxGroups * RADIX * 8));
    GS_HIP(hipMalloc((void**)&st.groupIncl, (size_t)st.maxGroups * RADIX * 8));
    GS_HIP(hipMemsetAsync(st.groupIncl, 0, (size_t)st.maxGroups * RADIX * 8, ctx->stream));
    return GS_OK;
}

void sort_state_destroy(SortState& st) {
    if (st.altKeys) (void)hipFree(st.altKeys);
    if (st.altVals) (void)hipFree(st.altVals);
    if (st.status) (void)hipFree(st.status);
    if (st.groupAgg) (void)hipFree(st.groupAgg);
    if (st.groupIncl) (void)hipFree(st.groupIncl);
    st = SortState();
}

namespace { __global__ __launch_bounds__(256) void gather_keys_kernel(const uint32_t* __restrict__ keyBySplat, const uint32_t* __restrict__ order, uint32_t* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = keyBySplat[order[i]];
} }
// sorted keys on demand (distances[i] = key of the splat at sorted position i): the depth sort itself only delivers the order
int32_t enqueue_gather_keys(gs_context* ctx, const uint32_t* keyBySplat, const uint32_t* order, uint32_t* out, uint32_t n) {
    hipLaunchKernelGGL(gather_keys_kernel, dim3(max(1u, min(div_up(n, 256), (uint32_t)ctx->cuCount * 8u))), dim3(256), 0, ctx->stream, keyBySplat, order, out, n);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_set_indices(gs_context* ctx, uint32_t* order, uint32_t n) {
    hipLaunchKernelGGL(set_indices_kernel, dim3(div_up(n, 1024)), dim3(1024), 0, ctx->stream, order, n);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Copies of the digit histograms a producer with `producerBlocks` flushing workgroups spreads over: every copy is 32 lines that queue same-line
// atomics at the memory side, but every Onesweep workgroup has to sum the copies in use.  GSPLAT_HIST_COPIES pins it (A/B).
uint32_t hist_copies(int producerBlocks) {
    static const int forced = [] { const char* e = getenv("GSPLAT_HIST_COPIES"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= kHistReplicas) ? v : 0; }();
    if (forced) return (uint32_t)forced;
    return producerBlocks > 512 ? (uint32_t)kHistReplicas : (producerBlocks > 256 ? 4u : 1u);
}

uint32_t sort_group_words(const SortState& st, uint32_t nUpper, int passes) { return (uint32_t)passes * div_up(div_up(max(nUpper, 1u), st.partMin), (uint32_t)GROUP) * RADIX; }

int32_t enqueue_sort_keys(gs_context* ctx, hipStream_t stream, const gsm::AssetView& a, const float* m, uint32_t* keyBySplat,
                          SortControl* control, SortControl* nextControl, uint32_t n, SortState& st) {
    // `control` was zeroed by the previous sort's launch of this kernel (or at creation); this launch zeroes `nextControl`
    const uint32_t chunks = div_up(n, 256u);
    const uint32_t grid = max(1u, min(div_up(chunks, 4u * kKeysIlp), (uint32_t)ctx->cuCount));      // one 1024-thread workgroup per CU
    st.histCopies = hist_copies((int)grid);
#define GS_LAUNCH_KEYS(F) hipLaunchKernelGGL(sort_keys_kernel<F>, dim3(grid), dim3(1024), 0, stream, a, m[8], m[9], m[10], m[11], keyBySplat, \
                                             control->hist, n, st.groupAgg, sort_group_words(st, n, 4), (uint32_t*)nextControl, st.histCopies)
    switch (a.posFmt) { case 0: GS_LAUNCH_KEYS(0); break; case 1: GS_LAUNCH_KEYS(1); break; case 2: GS_LAUNCH_KEYS(2); break; default: GS_LAUNCH_KEYS(3); break; }
#undef GS_LAUNCH_KEYS
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_histogram(gs_context* ctx, hipStream_t stream, const uint32_t* keys, uint32_t n, const uint32_t* nPtr, int passes, uint32_t lastMask, SortControl* control,
                          SortState& st) {
    GS_HIP(hipMemsetAsync(control, 0, sizeof(SortControl), stream));
    const uint32_t grid = max(1u, min(div_up(n, 256), (uint32_t)ctx->cuCount * 4u));
    st.histCopies = hist_copies((int)grid);
    hipLaunchKernelGGL(histogram_kernel, dim3(grid), dim3(256), 0, stream, keys, n, nPtr, passes, lastMask, control->hist, st.groupAgg, sort_group_words(st, n, passes), st.histCopies);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_sort_passes(gs_context* ctx, hipStream_t stream, SortState& st, SortControl* control, uint32_t* keys, uint32_t* vals, uint32_t nUpper,
                            const uint32_t* nPtr, int passes, uint32_t lastMask, gs_renderer* profR, int evFirst, int bits, const uint32_t* gatherKeys,
                            bool skipLastKeys, uint32_t expected) {
    const uint32_t histCopies = min(max(st.histCopies, 1u), (uint32_t)kHistReplicas);
    if (passes < 1 || passes > 4) return fail(GS_ERR_INVALID_ARGUMENT, "sort passes");
    if (bits < 6 || bits > 8 || (gatherKeys && bits != 8)) return fail(GS_ERR_INVALID_ARGUMENT, "sort digit width");
    if (nUpper > st.maxCount) return fail(GS_ERR_INVALID_ARGUMENT, "sort count exceeds sorter capacity");
    if (nUpper == 0) return GS_OK;
    // shape by the expected key count (the pair sort knows only an upper bound on the host: the caller passes the last frame's count);
    // the order produced is the same either way.  GSPLAT_SORT_SHAPE=a|b pins it (tests run every size through both).
    static const int forcedShape = [] { const char* e = getenv("GSPLAT_SORT_SHAPE");
                                        return !e ? 0 : (e[0] == 'a' || e[0] == 'A') ? 1 : (e[0] == 'b' || e[0] == 'B') ? 2 : (e[0] == 'c' || e[0] == 'C') ? 3 : 0; }();
    const uint32_t expect = min(expected ? expected : nUpper, nUpper);
    const bool canC = bits == 8 && !gatherKeys && st.partMin <= (uint32_t)PART_C;      // shape C exists for the plain 8-bit passes of a sort sized for it
    const bool shapeC = canC && (forcedShape ? forcedShape == 3 : expect <= kSmallSortKeys);
    const bool shapeB = !shapeC && (forcedShape ? forcedShape == 2 : expect > kBigSortKeys);
    const uint32_t parts = div_up(nUpper, shapeC ? (uint32_t)PART_C : shapeB ? (uint32_t)PART_B : (uint32_t)PART_A);
    // persistent grid: no more workgroups than are resident at once (3 per CU at <= 80 VGPRs / 43 KB LDS, 2 at <= 128 / 52 KB), a
    // multiple of the ticket classes so that every class is served
    const uint32_t capacity = max(((uint32_t)ctx->cuCount * (shapeB ? 2u : 3u) / TICKET_CLASSES) * TICKET_CLASSES, TICKET_CLASSES);
    const uint32_t grid = min(div_up(parts, TICKET_CLASSES) * TICKET_CLASSES, capacity);
    uint32_t *ks = keys, *vs = vals, *kd = st.altKeys, *vd = st.altVals;
    const uint32_t groups = div_up(div_up(nUpper, st.partMin), (uint32_t)GROUP);             // per-pass stride of the group words: sort_group_words()
    const uint32_t fullMask = (1u << bits) - 1u;
    // st.groupAgg[passes][groups][256] accumulates: it was zeroed by the kernel that produced the keys / their histograms
    if (profR && evFirst >= 0) prof_record(profR, evFirst, stream);
    for (int p = 0; p < passes; ++p) {
        uint32_t epoch = (++st.epoch) & EPOCH_MASK;
        if (epoch == 0) {   // 18-bit epoch wrapped: wipe the tagged words (they may hold every old epoch), restart at 1
            GS_HIP(hipMemsetAsync(st.status, 0, (size_t)st.maxParts * RADIX * 4, stream));
            GS_HIP(hipMemsetAsync(st.groupIncl, 0, (size_t)st.maxGroups * RADIX * 8, stream));
            st.epoch = epoch = 1;
        }
        const uint32_t* hist = control->hist + RADIX * p;
        unsigned long long* agg = st.groupAgg + (size_t)p * groups * RADIX;
        const uint32_t mask = p == passes - 1 ? (lastMask & fullMask) : fullMask;
        const uint32_t shift = (uint32_t)(bits * p);
        uint32_t* kdst = (skipLastKeys && p == passes - 1) ? (uint32_t*)nullptr : kd;      // the payload (order) is all the caller wants
        // kernel timing (gs_renderer_set_kernel_timing): the launch's OWN start / stop timestamps (hipExtLaunchKernelGGL: taken from the
        // dispatch packet's completion signal, what rocprofv3 --kernel-trace reports), not a pair of hipEventRecord packets around the
        // launches -- those include the kernel boundaries and a barrier packet each (8-10 % above the kernel's own duration for a 31 us
        // pass).  The timestamped launches cost ~6 us each themselves, which is why this is a mode of its own and not part of the
        // stage brackets (with it on, sort_ms / pair_sort_ms read ~25 us high).
        hipEvent_t evStart = nullptr, evStop = nullptr;
        if (profR && profR->profiling && profR->kernelTiming && profR->ev && evFirst >= 0) {
            const int kb = profR->profCur * kEvPerFrame + (evFirst == 10 ? 14 : 22) + 2 * p;
            evStart = profR->ev[kb]; evStop = profR->ev[kb + 1];
            profR->evValid[kb] = profR->evValid[kb + 1] = 1;
        }
#define GS_LAUNCH_ONESWEEP_K(B, G, KIN, K) \
        hipExtLaunchKernelGGL((onesweep_kernel<B, G, K>), dim3(grid), dim3(THREADS), 0, stream, evStart, evStop, 0, (const uint32_t*)(KIN), (const uint32_t*)vs, kdst, vd, \
                              (const uint32_t*)hist, st.status, agg, st.groupIncl, (uint32_t*)control->tickets[p], &control->error, nUpper, nPtr, shift, epoch, mask, histCopies, gs_shared_gpu(ctx) ? 0u : ((gatherKeys ? 1u : 0u) | 2u))
#define GS_LAUNCH_ONESWEEP(B, G, KIN) do { if (shapeB) GS_LAUNCH_ONESWEEP_K(B, G, KIN, KPT_B); else GS_LAUNCH_ONESWEEP_K(B, G, KIN, KPT_A); } while (0)
        if (p == 0 && gatherKeys) GS_LAUNCH_ONESWEEP(8, true, gatherKeys);
        else if (shapeC) GS_LAUNCH_ONESWEEP_K(8, false, ks, KPT_C);
        else if (bits == 8) GS_LAUNCH_ONESWEEP(8, false, ks);
        else if (bits == 7) GS_LAUNCH_ONESWEEP(7, false, ks);
        else GS_LAUNCH_ONESWEEP(6, false, ks);
#undef GS_LAUNCH_ONESWEEP
#undef GS_LAUNCH_ONESWEEP_K
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
