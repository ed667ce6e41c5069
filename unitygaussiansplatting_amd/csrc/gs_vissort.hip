// gs_vissort.hip -- GS_SORT_VISIBLE: the depth sort of the splats that are DRAWN, instead of all N.
//
// The reference sorts every splat every frame because its sort precedes its cull (SortPoints, GaussianSplatRenderer.cs:612-639, is
// recorded before CalcViewData, :579-610, and CSCalcDistances -- SplatUtilities.compute:69-82 -- keys all of _SplatCount).  Only the
// order among the splats that reach the screen is observable, and calc_view (gs_view.hip) needs nothing from the sort.  So in this
// mode the frame runs calc_view first and then, over its visibility bits:
//   1. visible_keys_kernel   keys of the V visible splats (CSCalcDistances' arithmetic, the matrix of the last gs_renderer_sort) compacted
//                            in SPLAT-INDEX order into (visKeys, visIdx), the four digit histograms of those keys, V itself;
//   2. four plain Onesweep passes (gs_sort.hip) over V pairs -- no gather pass, no keys of culled splats;
//   3. tie_fix_kernel<TB>    the reference's order among EQUAL keys.  Its sort is stable and its input is the previous frame's order, so
//                            after sorts M_1 .. M_k of a base order B its buffer is sorted lexicographically by (key under M_k, ..., key under
//                            M_1, rank in B): tied splats keep what the earlier sort matrices gave them.  A stable sort of an index-ordered
//                            compaction leaves ties in index order; the fix-up re-orders every run of equal keys by that chain, re-evaluating
//                            the tied splats' keys under the rows recorded since B (gs_renderer::visHist, most recent first; a matrix that
//                            occurs twice only counts where it occurs first, so a static camera keeps ONE) and ending in B: the index while B
//                            is CSSetIndices' identity, else rank[] = the inverse of gs_renderer::order (invert_order_kernel).  Runs of 2..4 are
//                            ordered by the thread that finds them, 5..64 by a wave, longer ones by the workgroup (a bitonic network in place).
// Nothing is ever dropped from the chain: when the history is full (kVisHistory rows) gs_api.hip's vis_consolidate carries the recorded sorts
// out on all N -- one reference-shaped sort of B by the most recent row + tie_fix_kernel<TB_POSITION> over N -- and that buffer is the new B.
// The draw (gs_raster.hip: vis_count / vis_offsets / vis_emit) then bins the V sorted entries instead of walking N positions of the full order.
// Host-side bookkeeping (the matrix history) lives at the end of this file; the API around it in gs_api.hip.  DESIGN.md section 4.4.
#include "gs_common.h"
#include <cstdlib>

namespace gs {

namespace {

constexpr int VTHREADS = 512;                 // <= 80 VGPRs: three blocks = 24 waves per CU (the kernel is a chain of dependent round trips; waves hide them)
constexpr int VWAVES = VTHREADS / 64;
constexpr uint32_t VIS_SPIN_LIMIT = 1u << 24;

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// add 1 to an LDS histogram bin, wave-aggregated when every active lane hits the same bin (the high digits of depth keys)
__device__ __forceinline__ void lds_hist_add(uint32_t* h, uint32_t d) {
    const uint32_t first = __builtin_amdgcn_readfirstlane(d);
    const unsigned long long act = __ballot(1);
    if (__all(d == first)) {
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&h[first], (uint32_t)__popcll(act));
    } else {
        atomicAdd(&h[d], 1u);
    }
}

// Block b owns the visibility words [b * blockWords, (b + 1) * blockWords) -- 64 splats per word, blockWords a multiple of 4 so that a
// word never straddles a 256-splat chunk.  Its first output slot is the number of visible splats before it: every block publishes its
// own count as soon as it has summed its words (status[b] = count + 1) and sums the counts of the blocks before it -- at most
// kVisMaxBlocks words, one or two per thread.  Block b only waits on blocks with a SMALLER index, which the dispatcher has started before it,
// so the wait does not depend on how many workgroups are resident.  HIST = false: no sort follows (nothing was ever sorted: the order
// is the index order), only the compaction.
// RANKKEY: no sort has been made on this base yet (gs_renderer_upload_order, or sorts in GS_SORT_FULL, then a frame without SortPoints): the
// reference draws in the base order itself, so the key of a visible splat is its rank in it (rank = the inverse of order[]).
template <int POSFMT, bool HIST, bool RANKKEY>
__global__ __launch_bounds__(VTHREADS, 6) void visible_keys_kernel(gsm::AssetView a, float m20, float m21, float m22, float m23, const uint32_t* __restrict__ rank,
                                                                const unsigned long long* __restrict__ visMask, uint32_t words, uint32_t blockWords,
                                                                uint32_t* __restrict__ outKeys, uint32_t* __restrict__ outIdx, uint32_t* __restrict__ hist,
                                                                VisControl* vc, uint32_t* __restrict__ nextVc,
                                                                unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords, uint32_t* __restrict__ nextControl, uint32_t copies) {
    GS_CHAIN_PRIORITY();
    __shared__ uint32_t s_h[4 * 256];
    __shared__ unsigned long long s_m[VTHREADS];
    __shared__ uint32_t s_off[VTHREADS];
    __shared__ uint32_t s_live[VTHREADS];
    __shared__ uint32_t s_w[VWAVES], s_w2[VWAVES];
    __shared__ uint32_t s_bcast[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_bcast[1] = 0;                                 // (read after the barriers of the first scan)
    // housekeeping for the sort passes that follow and for the NEXT sort (the control blocks alternate: no memset launch per frame)
    if (HIST) {
        for (int j = tid; j < 4 * 256; j += VTHREADS) s_h[j] = 0;
        for (uint32_t j = blockIdx.x * (uint32_t)VTHREADS + tid; j < groupAggWords; j += gridDim.x * (uint32_t)VTHREADS) groupAgg[j] = 0ull;
    }
    for (uint32_t j = blockIdx.x * (uint32_t)VTHREADS + tid; j < (uint32_t)(sizeof(SortControl) / 4); j += gridDim.x * (uint32_t)VTHREADS) nextControl[j] = 0u;
    for (uint32_t j = blockIdx.x * (uint32_t)VTHREADS + tid; j < (uint32_t)(sizeof(VisControl) / 4); j += gridDim.x * (uint32_t)VTHREADS) nextVc[j] = 0u;

    const uint32_t w0 = blockIdx.x * blockWords, w1 = min(w0 + blockWords, words);
    const bool chunked = a.chunkCount != 0u;
    const uint8_t* cbase = chunked ? a.chunk : (const uint8_t*)visMask;      // (no chunks: any readable 64 bytes, never used)
    const uint32_t lastChunk = chunked ? a.chunkCount - 1u : 0u;
    typedef gsm::RawVec<POSFMT> Raw;
    constexpr uint32_t ILP = 4;                                   // live words in flight per wave
    struct Batch { Raw raw[ILP]; uint4 bx[ILP]; uint2 bz[ILP]; uint32_t sidx[ILP], off[ILP]; unsigned long long mm[ILP]; };
    // the words of a sub-tile that hold a visible splat are compacted into s_live, so that the waves share them evenly: entry k0 .. k0 + ILP - 1
    auto load_batch = [&](Batch& B, uint32_t s0, uint32_t k0, uint32_t nLive) {
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            const uint32_t t = s_live[min(k0 + k, nLive - 1u)];               // (a clamped duplicate is masked out below)
            B.mm[k] = (k0 + k < nLive) ? s_m[t] : 0ull;
            B.off[k] = s_off[t];
            const uint32_t word = s0 + t;
            B.sidx[k] = word * 64u + (uint32_t)lane;
            // unconditional loads (index clamped): a load inside a divergent branch is waited for at the end of the branch
            const uint32_t li = min(B.sidx[k], a.n - 1u);
            if (RANKKEY) { B.bx[k] = make_uint4(rank[li], 0u, 0u, 0u); continue; }
            B.raw[k] = gsm::LoadRawT<POSFMT>(a.pos, (uint64_t)li * gsm::vecStrideT<POSFMT>());
            const uint8_t* cp = cbase + (size_t)min(word >> 2, lastChunk) * 64u;               // ChunkInfo.posX/Y/Z bounds: wave-uniform address
            B.bx[k] = *(const uint4*)(cp + 16);
            B.bz[k] = *(const uint2*)(cp + 32);
        }
    };
    auto process_batch = [&](const Batch& B, uint32_t firstSlot) {
#pragma unroll
        for (uint32_t k = 0; k < ILP; ++k) {
            if (!((B.mm[k] >> lane) & 1ull)) continue;
            uint32_t key;
            if (RANKKEY) key = B.bx[k].x;
            else {
                gsm::V3 pos = gsm::DecodeRawT<POSFMT>(B.raw[k], (uint64_t)B.sidx[k] * gsm::vecStrideT<POSFMT>());
                if (chunked && (B.sidx[k] >> 8) <= lastChunk) {              // LoadSplatPos' chunk de-normalisation (ChunkLerpPos), same expressions
                    pos.x = gsm::lerpf(gsm::u2f(B.bx[k].x), gsm::u2f(B.bx[k].y), pos.x);
                    pos.y = gsm::lerpf(gsm::u2f(B.bx[k].z), gsm::u2f(B.bx[k].w), pos.y);
                    pos.z = gsm::lerpf(gsm::u2f(B.bz[k].x), gsm::u2f(B.bz[k].y), pos.z);
                }
                key = gsm::SortKeyOf(pos, m20, m21, m22, m23);
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(B.mm[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B.mm[k], 0u));
            const uint32_t p = firstSlot + B.off[k] + below;
            outKeys[p] = key;
            outIdx[p] = B.sidx[k];
            if (HIST) {
                lds_hist_add(s_h, key & 255u);
                lds_hist_add(s_h + 256, (key >> 8) & 255u);
                lds_hist_add(s_h + 512, (key >> 16) & 255u);
                lds_hist_add(s_h + 768, key >> 24);
            }
        }
    };
    // one sub-tile of VTHREADS words (one per thread): the words into s_m, their sub-tile-local first slots into s_off (block scan), the live ones
    // into s_live; returns {visible splats : 20 | live words << 20} of the sub-tile
    auto scan_subtile = [&](unsigned long long m) -> uint32_t {
        const uint32_t c = (uint32_t)__popcll(m);
        const uint32_t packed = c | ((m != 0ull ? 1u : 0u) << 20);           // visible splats (< 2^17 per sub-tile) and live words in one scan
        const uint32_t incl = wave_incl_scan32(packed, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, subTotal = 0;
#pragma unroll
        for (int k = 0; k < VWAVES; ++k) { const uint32_t t = s_w[k]; wbase += (k < w) ? t : 0u; subTotal += t; }
        const uint32_t excl = wbase + incl - packed;
        s_m[tid] = m;
        s_off[tid] = excl & 0xfffffu;
        if (m != 0ull) s_live[excl >> 20] = (uint32_t)tid;
        __syncthreads();
        return subTotal;
    };

    // ---- the first sub-tile (the whole block unless the asset is huge): its words are read ONCE, its scan gives the block's count, and the
    //      positions of every wave's first batch are requested BEFORE the block waits for the counts of the blocks before it
    const unsigned long long m0 = w0 + (uint32_t)tid < w1 ? visMask[w0 + tid] : 0ull;          // (bits of splats >= n are never set)
    uint32_t extra = 0;
    for (uint32_t wi = w0 + VTHREADS + tid; wi < w1; wi += VTHREADS) extra += (uint32_t)__popcll(visMask[wi]);       // (further sub-tiles: count only, for now)
    const uint32_t sub0 = scan_subtile(m0);
    const uint32_t nLive0 = sub0 >> 20;
    if (w1 - w0 > (uint32_t)VTHREADS) {                           // (uniform) block total = first sub-tile + the rest
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) extra += __shfl_xor(extra, o, 64);
        if (lane == 0) atomicAdd(&s_bcast[1], extra);            // (s_bcast[1] zeroed below, before the barrier inside scan_subtile ... see init)
        __syncthreads();
        extra = s_bcast[1];
    } else extra = 0;
    const uint32_t total = (sub0 & 0xfffffu) + extra;
    if (tid == 0) __hip_atomic_store(&vc->status[blockIdx.x * kVisStatusStride], total + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Batch first;
    const uint32_t kFirst = (uint32_t)w * ILP;
    const bool haveFirst = kFirst < nLive0;                       // (wave-uniform)
    if (haveFirst) load_batch(first, w0, kFirst, nLive0);
    // ---- the counts of the blocks before this one
    uint32_t before = 0;
    for (uint32_t j = (uint32_t)tid; j < blockIdx.x; j += VTHREADS) {
        uint32_t v = 0, spins = 0;
        for (;;) {
            v = __hip_atomic_load(&vc->status[j * kVisStatusStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v) break;
            if (++spins > VIS_SPIN_LIMIT) { atomicOr(&vc->error, 1u); v = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        before += v - 1u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) s_w2[w] = before;
    __syncthreads();
    uint32_t running = 0;                                         // first output slot of the block / of the next sub-tile (uniform)
#pragma unroll
    for (int k = 0; k < VWAVES; ++k) running += s_w2[k];
    if (blockIdx.x == gridDim.x - 1u && tid == 0) vc->count = running + total;
    if (total != 0u) {                                            // (uniform)
        if (haveFirst) process_batch(first, running);
        for (uint32_t k0 = kFirst + VWAVES * ILP; k0 < nLive0; k0 += VWAVES * ILP) {        // wave-uniform
            Batch B;
            load_batch(B, w0, k0, nLive0);
            process_batch(B, running);
        }
        running += sub0 & 0xfffffu;
        // ---- further sub-tiles (assets beyond ~32 M splats)
        for (uint32_t s0 = w0 + VTHREADS; s0 < w1; s0 += VTHREADS) {
            __syncthreads();                                      // the previous sub-tile's s_m / s_off / s_live / s_w are no longer read
            const uint32_t wi = s0 + tid;
            const uint32_t sub = scan_subtile(wi < w1 ? visMask[wi] : 0ull);
            const uint32_t nLive = sub >> 20;
            for (uint32_t k0 = (uint32_t)w * ILP; k0 < nLive; k0 += VWAVES * ILP) {
                Batch B;
                load_batch(B, s0, k0, nLive);
                process_batch(B, running);
            }
            running += sub & 0xfffffu;
        }
    }
    if (HIST && total != 0u) {
        // two neighbouring bins per 64-bit atomic (a bin never reaches 2^32: no carry into its neighbour), into one of the copies
        __syncthreads();
        unsigned long long* myHist = (unsigned long long*)(hist + (blockIdx.x % copies) * (uint32_t)kHistStride);      // SortControl::hist: one of the copies
        for (int j = tid; j < 2 * 256; j += VTHREADS) {
            const unsigned long long c = (unsigned long long)s_h[2 * j] | ((unsigned long long)s_h[2 * j + 1] << 32);
            if (c) atomicAdd(myHist + j, c);
        }
    }
}

// ---- the reference's order among equal keys ---------------------------------------------------------------------------------------
struct TiePos { float x, y, z; };
__device__ __forceinline__ bool tie_same_pos(const TiePos& a, const TiePos& b) {
    return gsm::f2u(a.x) == gsm::f2u(b.x) && gsm::f2u(a.y) == gsm::f2u(b.y) && gsm::f2u(a.z) == gsm::f2u(b.z);
}
// rows [first, depth) of the history (LDS, row h = 4 floats, most recent first): -1 = A precedes B, +1 = B precedes A, 0 = tied under all of them
__device__ __forceinline__ int tie_chain(const float* rows, uint32_t first, uint32_t depth, const TiePos& pa, const TiePos& pb) {
    if (tie_same_pos(pa, pb)) return 0;                          // equal under every matrix
    for (uint32_t h = first; h < depth; ++h) {
        const float* r = rows + 4 * h;
        const uint32_t ka = gsm::SortKeyOf(gsm::V3{ pa.x, pa.y, pa.z }, r[0], r[1], r[2], r[3]);
        const uint32_t kb = gsm::SortKeyOf(gsm::V3{ pb.x, pb.y, pb.z }, r[0], r[1], r[2], r[3]);
        if (ka != kb) return ka < kb ? -1 : 1;
    }
    return 0;
}
// does splat A precede splat B in the reference's order buffer, given that their keys under row 0 are equal?  The chain ends in the base
// order: the splat index (TB_INDEX), rank[splat] (TB_RANK) or the position the stable sort of the base left the splat at (TB_POSITION) --
// ta / tb carry the index or the position.
template <int TB>
__device__ __forceinline__ bool tie_precedes(const float* rows, uint32_t depth, const TiePos& pa, uint32_t ea, uint32_t ta, const TiePos& pb, uint32_t eb, uint32_t tb,
                                             const uint32_t* __restrict__ rank) {
    const int c = tie_chain(rows, 1u, depth, pa, pb);
    if (c) return c < 0;
    if (TB == TB_RANK) return rank[ea] < rank[eb];
    return ta < tb;
}

constexpr int TIE_THREADS = 256, TIE_ITEMS = 8, TIE_SEG = TIE_THREADS * TIE_ITEMS;
constexpr int TIE_XL_MAX = TIE_SEG / 65 + 2;                      // runs of more than 64 positions that can START in one segment
constexpr int TIE_XL_ILP = 4;                                     // compare-exchanges a thread of the sorting network keeps in flight

// A run of MORE than 64 equal keys (coplanar / lattice geometry seen along an axis), ordered by the whole workgroup: a bitonic sorting
// network (the all-ascending form: a flip step then half-cleaners, so that positions >= L simply do not exist) over idx[i .. i + L) in
// place, through global memory -- any length, no scratch of its own.  The comparator is a strict total order (the chain ends in the base
// order, which is a permutation), so the network's result is THE order whatever the network does in between.  Almost every comparison is
// decided by the key under row 1, which is cached per splat (k1) before the network starts; t (TB_POSITION only) is the position the
// stable sort left the splat at.  Rare by construction (a real scene's depth keys do not tie 65 deep): built to be exact, not fast.
template <int TB>
__device__ void tie_sort_long_run(const gsm::AssetView& a, const float* rows, uint32_t depth, const uint32_t* __restrict__ keys, uint32_t* idx, uint32_t n, uint32_t i,
                                  const uint32_t* __restrict__ rank, uint32_t* k1, uint32_t* tpos, VisControl* vc, uint32_t* s_len) {
    const int tid = threadIdx.x;
    const uint32_t kc = keys[i];
    constexpr uint32_t first = 1u;                                // the chain starts behind the row the keys were made with
    if (tid == 0) *s_len = 0xffffffffu;
    __syncthreads();
    for (uint32_t b = i + 64u;; b += (uint32_t)TIE_THREADS * 4u) {          // (positions i .. i + 64 are known to be equal)
        uint32_t firstStop = 0xffffffffu;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const uint32_t p = b + (uint32_t)k * TIE_THREADS + (uint32_t)tid;
            if (p >= n || keys[min(p, n - 1u)] != kc) firstStop = p - i;
        }
        if (__syncthreads_or(firstStop != 0xffffffffu)) {
            if (firstStop != 0xffffffffu) atomicMin(s_len, firstStop);
            __syncthreads();
            break;
        }
    }
    const uint32_t L = *s_len;
    if (tid == 0 && vc) { atomicAdd(&vc->tieLongRuns, 1u); atomicMax(&vc->tieLongest, L); }
    const bool deep = depth > first;                              // rows that can decide exist
    if (!deep && TB != TB_RANK) { __syncthreads(); return; }      // nothing but the base order: the stable sort left the run in it
    const float* r1 = rows + 4u * first;
    for (uint32_t j = (uint32_t)tid; j < L; j += TIE_THREADS) {
        const uint32_t e = idx[i + j];
        if (deep) k1[e] = gsm::SortKeyOf(gsm::LoadSplatPos(a, e), r1[0], r1[1], r1[2], r1[3]);
        if (TB == TB_POSITION) tpos[e] = i + j;
    }
    __threadfence_block();
    __syncthreads();
    volatile uint32_t* v = idx + i;
    // b precedes a (keys under the rows up to and including `first` equal)?
    auto deep_precedes = [&](uint32_t eb, uint32_t ea) -> bool {
        if (depth > first + 1u) {
            const gsm::V3 qa = gsm::LoadSplatPos(a, ea), qb = gsm::LoadSplatPos(a, eb);
            const int c = tie_chain(rows, first + 1u, depth, TiePos{ qb.x, qb.y, qb.z }, TiePos{ qa.x, qa.y, qa.z });
            if (c) return c < 0;
        }
        if (TB == TB_RANK) return rank[eb] < rank[ea];
        if (TB == TB_POSITION) return ((volatile uint32_t*)tpos)[eb] < ((volatile uint32_t*)tpos)[ea];
        return eb < ea;
    };
    auto exchange = [&](const uint32_t (&x)[TIE_XL_ILP], const uint32_t (&y)[TIE_XL_ILP]) {
        uint32_t ex[TIE_XL_ILP], ey[TIE_XL_ILP], kx[TIE_XL_ILP], ky[TIE_XL_ILP];
#pragma unroll
        for (int u = 0; u < TIE_XL_ILP; ++u) { ex[u] = v[min(x[u], L - 1u)]; ey[u] = v[min(y[u], L - 1u)]; }
#pragma unroll
        for (int u = 0; u < TIE_XL_ILP; ++u) { kx[u] = deep ? ((volatile uint32_t*)k1)[ex[u]] : 0u; ky[u] = deep ? ((volatile uint32_t*)k1)[ey[u]] : 0u; }
#pragma unroll
        for (int u = 0; u < TIE_XL_ILP; ++u) {
            if (y[u] >= L) continue;                              // (x < y: the partner does not exist -- "+inf", stays where it is)
            const bool sw = ky[u] < kx[u] || (ky[u] == kx[u] && deep_precedes(ey[u], ex[u]));
            if (sw) { v[x[u]] = ey[u]; v[y[u]] = ex[u]; }
        }
    };
    uint32_t lp = 0;
    while ((1u << lp) < L) ++lp;                                  // the network's size 2^lp >= L (L <= 2^30)
    const uint32_t half = lp ? 1u << (lp - 1u) : 0u;               // compare-exchanges per step
    for (uint32_t lk = 1; lk <= lp; ++lk) {
        // flip step: position o of the lower half of every block of 2^lk meets its mirror image in the upper half
        for (uint32_t p0 = (uint32_t)tid; p0 < half; p0 += (uint32_t)TIE_THREADS * TIE_XL_ILP) {
            uint32_t x[TIE_XL_ILP], y[TIE_XL_ILP];
#pragma unroll
            for (int u = 0; u < TIE_XL_ILP; ++u) {
                const uint32_t p = p0 + (uint32_t)u * TIE_THREADS;
                const uint32_t blk = p >> (lk - 1u), o = p & ((1u << (lk - 1u)) - 1u);
                x[u] = (blk << lk) + o;
                y[u] = p < half ? (blk << lk) + (1u << lk) - 1u - o : 0xffffffffu;
            }
            exchange(x, y);
        }
        __syncthreads();
        for (uint32_t lj = lk - 1u; lj-- > 0u;) {                  // half-cleaners: strides 2^(lk-2) .. 1
            for (uint32_t p0 = (uint32_t)tid; p0 < half; p0 += (uint32_t)TIE_THREADS * TIE_XL_ILP) {
                uint32_t x[TIE_XL_ILP], y[TIE_XL_ILP];
#pragma unroll
                for (int u = 0; u < TIE_XL_ILP; ++u) {
                    const uint32_t p = p0 + (uint32_t)u * TIE_THREADS;
                    x[u] = ((p >> lj) << (lj + 1u)) | (p & ((1u << lj) - 1u));
                    y[u] = p < half ? x[u] + (1u << lj) : 0xffffffffu;
                }
                exchange(x, y);
            }
            __syncthreads();
        }
    }
}

// One workgroup per segment of TIE_SEG sorted positions: every thread looks at TIE_ITEMS of them for the START of a run of equal keys
// (about one position in twenty at C2: 2.2 M keys on the ~2^25 floats of the depth range), the starts are compacted into LDS and dealt
// to the threads, so that the re-ordering -- two or three dependent random gathers (index -> position, ChunkInfo) and a few dozen VALU
// instructions per run -- runs on full waves.  A run belongs to the segment its first position lies in, whatever it extends into.
// Runs of 2..4 are ranked by the thread that finds them, 5..64 by a wave (a lane per member), longer ones by the workgroup (above).
template <int TB>
__global__ __launch_bounds__(TIE_THREADS) void tie_fix_kernel(gsm::AssetView a, TieHistory H, const uint32_t* __restrict__ keys, uint32_t* idx,
                                                              const uint32_t* __restrict__ nPtr, uint32_t nImm, VisControl* vc,
                                                              const uint32_t* __restrict__ rank, uint32_t* k1BySplat, uint32_t* tBySplat) {
    GS_CHAIN_PRIORITY();
    __shared__ float s_rows[kVisHistory * 4];
    __shared__ uint32_t s_start[TIE_SEG / 2];                     // a run has >= 2 positions
    __shared__ uint32_t s_long[TIE_SEG / 5 + 8];
    __shared__ uint32_t s_xl[TIE_XL_MAX];
    __shared__ uint32_t s_nStart, s_nLong, s_nXl, s_len;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = nPtr ? min(*nPtr, nImm) : nImm;
    const uint32_t depth = H.depth;
    for (uint32_t j = tid; j < depth * 4u; j += TIE_THREADS) s_rows[j] = H.row[j >> 2][j & 3u];
    for (uint32_t seg = blockIdx.x; (unsigned long long)seg * TIE_SEG < n; seg += gridDim.x) {
        __syncthreads();
        if (tid == 0) { s_nStart = 0; s_nLong = 0; s_nXl = 0; }
        __syncthreads();
        const uint32_t base = seg * (uint32_t)TIE_SEG;
#pragma unroll
        for (int k = 0; k < TIE_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)k * TIE_THREADS + (uint32_t)tid;
            // (three loads per position, two of them cache hits; indices clamped so that the loads are unconditional)
            const uint32_t kc = keys[min(i, n - 1u)], kp = keys[min(i, n - 1u) - (i > 0u && i < n ? 1u : 0u)], kn = keys[min(i + 1u, n - 1u)];
            const bool start = i + 1u < n && kn == kc && (i == 0u || kp != kc);
            if (start) s_start[atomicAdd(&s_nStart, 1u)] = i;
        }
        __syncthreads();
        const uint32_t nStart = s_nStart;
        for (uint32_t s = tid; s < nStart; s += TIE_THREADS) {
            const uint32_t i = s_start[s];
            // (every load below is independent of the others: the keys behind the start and the run's first four indices in one round trip)
            const uint32_t kc = keys[i], k2 = keys[min(i + 2u, n - 1u)], k3 = keys[min(i + 3u, n - 1u)], k4 = keys[min(i + 4u, n - 1u)];
            uint32_t e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = idx[min(i + (uint32_t)j, n - 1u)];
            const bool m2 = i + 2u < n && k2 == kc, m3 = m2 && i + 3u < n && k3 == kc, m4 = m3 && i + 4u < n && k4 == kc;
            const uint32_t L = m4 ? 5u : (m3 ? 4u : (m2 ? 3u : 2u));
            if (L > 4u) { s_long[atomicAdd(&s_nLong, 1u)] = i; continue; }
            // ---- runs of 2..4: rank by counting with the chain comparison, in registers
            TiePos p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const gsm::V3 q = gsm::LoadSplatPos(a, e[j]);
                p[j] = TiePos{ q.x, q.y, q.z };
            }
            uint32_t rk[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = x + 1; y < 4; ++y)
                    if ((uint32_t)y < L) {
                        const bool xy = tie_precedes<TB>(s_rows, depth, p[x], e[x], TB == TB_POSITION ? (uint32_t)x : e[x], p[y], e[y], TB == TB_POSITION ? (uint32_t)y : e[y], rank);
                        rk[xy ? y : x] += 1u;
                    }
            // (the run arrives in the base order -- by index from the visible compaction, by position from a sort of the base -- so most
            // runs, whose earlier keys ascend with it as often as not, are already in place)
            bool moved = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) moved = moved || ((uint32_t)j < L && rk[j] != (uint32_t)j);
            if (moved) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < L) idx[i + rk[j]] = e[j];
            }
        }
        __syncthreads();
        // ---- runs of 5..64: one wave each, a lane per member, rank by counting against every other member (v_readlane broadcasts)
        const uint32_t nLong = s_nLong;
        for (uint32_t q = (uint32_t)w; q < nLong; q += TIE_THREADS / 64) {
            const uint32_t i = s_long[q];
            const uint32_t kc = keys[i];
            const bool same = i + (uint32_t)lane < n && keys[min(i + (uint32_t)lane, n - 1u)] == kc;
            const unsigned long long bal = __ballot(same);
            const uint32_t L = bal == ~0ull ? 64u : (uint32_t)__ffsll((long long)~bal) - 1u;
            if (L == 64u && i + 64u < n && keys[i + 64u] == kc) {       // longer than a wave: the workgroup's sorting network, below
                if (lane == 0) s_xl[atomicAdd(&s_nXl, 1u)] = i;
                continue;
            }
            const bool mine = (uint32_t)lane < L;
            const uint32_t e = idx[min(i + (uint32_t)lane, n - 1u)];
            const gsm::V3 pq = gsm::LoadSplatPos(a, mine ? e : idx[i]);
            const TiePos p = { pq.x, pq.y, pq.z };
            const uint32_t t = TB == TB_POSITION ? (uint32_t)lane : e;
            uint32_t rk = 0;
            for (uint32_t j = 0; j < L; ++j) {
                const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)j);
                const TiePos pj = { __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), (int)j)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), (int)j)),
                                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), (int)j)) };
                if (mine && j != (uint32_t)lane) {
                    const bool jFirst = tie_precedes<TB>(s_rows, depth, pj, ej, TB == TB_POSITION ? j : ej, p, e, t, rank);
                    rk += jFirst ? 1u : 0u;
                }
            }
            if (mine) idx[i + rk] = e;
        }
        __syncthreads();
        const uint32_t nXl = s_nXl;
        for (uint32_t q = 0; q < nXl; ++q) tie_sort_long_run<TB>(a, s_rows, depth, keys, idx, n, s_xl[q], rank, k1BySplat, tBySplat, vc, &s_len);
    }
}

// rank[order[i]] = i: the inverse of the base order (TB_RANK's end of the chain; the sort key of a frame drawn before any sort was made on this base)
__global__ __launch_bounds__(256) void invert_order_kernel(const uint32_t* __restrict__ order, uint32_t* __restrict__ rank, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) rank[order[i]] = i;
}

} // namespace

int32_t vis_alloc(gs_renderer* r) {
    if (r->visKeys) return GS_OK;
    uint32_t *k = nullptr, *v = nullptr, *x = nullptr, *y = nullptr, *po = nullptr; VisControl* c = nullptr;
    hipError_t e = hipMalloc((void**)&k, ((size_t)r->n + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v, ((size_t)r->n + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&x, ((size_t)r->n + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&y, ((size_t)r->n + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&po, ((size_t)r->n + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&c, 2 * sizeof(VisControl));
    if (e == hipSuccess) e = hipMemsetAsync(c, 0, 2 * sizeof(VisControl), r->ctx->stream);
    if (e != hipSuccess) {
        if (k) (void)hipFree(k);
        if (v) (void)hipFree(v);
        if (x) (void)hipFree(x);
        if (y) (void)hipFree(y);
        if (po) (void)hipFree(po);
        if (c) (void)hipFree(c);
        return fail_hip(e, "allocate the visible-sort buffers", __FILE__, __LINE__);
    }
    r->visKeys = k; r->visIdx = v; r->visRectX = x; r->visRectY = y; r->visPairOffset = po; r->visControl = c; r->visControlIdx = 0;
    static const int envLimit = [] { const char* s = getenv("GSPLAT_VIS_HISTORY"); const int v = s ? atoi(s) : 0; return (v >= 2 && v <= kVisHistory) ? v : 0; }();
    if (envLimit && r->visHistLimit == kVisHistory) r->visHistLimit = envLimit;
    return GS_OK;
}

void vis_free(gs_renderer* r) {
    if (r->visKeys) (void)hipFree(r->visKeys);
    if (r->visIdx) (void)hipFree(r->visIdx);
    if (r->visRectX) (void)hipFree(r->visRectX);
    if (r->visRectY) (void)hipFree(r->visRectY);
    if (r->visPairOffset) (void)hipFree(r->visPairOffset);
    if (r->visChunkStart) (void)hipFree(r->visChunkStart);
    if (r->visControl) (void)hipFree(r->visControl);
    if (r->visBaseRank) (void)hipFree(r->visBaseRank);
    r->visKeys = r->visIdx = r->visRectX = r->visRectY = r->visPairOffset = r->visChunkStart = r->visBaseRank = nullptr; r->visControl = nullptr; r->visChunkCap = 0;
    r->visRankValid = false;
}

// The sorts made since the base, as a list of DISTINCT rows, most recent first.  Sorting by a matrix that is already the head changes
// nothing (a stable sort of a sorted sequence); one that occurs deeper moves to the front (where it occurs again further down the
// lexicographic chain it can no longer decide anything: everything still tied there is tied under it).  A NEW row that does not fit
// consolidates the base first (vis_consolidate: order[] := the reference's buffer now, the history shrinks to its head) -- nothing is
// ever dropped, so the chain is the reference's at any number of sorts.
int32_t vis_push_matrix(gs_renderer* r, const float* m) {
    const float* row = m + 8;
    int found = -1;
    for (int j = 0; j < r->visHistDepth && found < 0; ++j)
        if (memcmp(r->visHist[j], row, 16) == 0) found = j;
    if (found == 0) return GS_OK;
    const int limit = r->visHistLimit < 2 ? 2 : (r->visHistLimit > kVisHistory ? kVisHistory : r->visHistLimit);
    if (found < 0 && r->visHistDepth >= limit) GS_TRY(vis_consolidate(r));      // (leaves one row)
    const int last = found > 0 ? found : r->visHistDepth++;      // found > 0: rows 0 .. found-1 move down one, over the duplicate
    for (int j = last; j > 0; --j) memcpy(r->visHist[j], r->visHist[j - 1], 16);
    memcpy(r->visHist[0], row, 16);
    r->visOrderValid = false;
    return GS_OK;
}

static void tie_history(const gs_renderer* r, TieHistory& H) {
    memset(&H, 0, sizeof(H));
    memcpy(H.row, r->visHist, sizeof(float) * 4 * (size_t)r->visHistDepth);
    H.depth = (uint32_t)r->visHistDepth;
}

// (keys, idx) = a STABLE sort of the base order[] by row 0 of the history, all N of them: re-order every run of equal keys by rows 1.. --
// what the sorts before the last one made of it -- and, where they tie too, keep the position (= the base order).  Consolidation's second half.
int32_t enqueue_tie_fix_full(gs_renderer* r, const uint32_t* keys, uint32_t* idx) {
    if (r->visHistDepth < 2) return GS_OK;
    TieHistory H;
    tie_history(r, H);
    const uint32_t n = r->n;
    const uint32_t tgrid = max(1u, min(div_up(n, (uint32_t)TIE_SEG), (uint32_t)r->ctx->cuCount * 8u));
    hipLaunchKernelGGL(tie_fix_kernel<TB_POSITION>, dim3(tgrid), dim3(TIE_THREADS), 0, r->ctx->stream, r->asset->view, H, keys, idx, (const uint32_t*)nullptr, n,
                       (VisControl*)nullptr, (const uint32_t*)nullptr, r->depthSort.altKeys, r->depthSort.altVals);      // (the sort's ping-pong buffers are free again: per-splat scratch)
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int32_t enqueue_visible_sort(gs_renderer* r) {
    gs_context* ctx = r->ctx;
    hipStream_t st = ctx->stream;
    GS_TRY(vis_alloc(r));
    const gsm::AssetView& a = r->asset->view;
    const uint32_t n = r->n;
    const uint32_t words = div_up(n, 64u);
    if (!r->visBaseIdentity && !r->visRankValid) {               // the base is a real order buffer: its inverse ends the tie chain
        if (!r->visBaseRank) GS_HIP(hipMalloc((void**)&r->visBaseRank, ((size_t)n + 16) * 4));
        GS_TRY(join_sort(r));
        hipLaunchKernelGGL(invert_order_kernel, dim3(max(1u, min(div_up(n, 256u), (uint32_t)ctx->cuCount * 8u))), dim3(256), 0, st, (const uint32_t*)r->order, r->visBaseRank, n);
        GS_HIP(hipGetLastError());
        r->visRankValid = true;
    }
    // blocks of >= 64 visibility words (4,096 splats), at most ~1000 of them (C2: 998 blocks of 96 words; measured on MI355X, r05 call 11: 22.7 us
    // against 25.0 with 128-word and 30.0 with 256-word blocks -- the kernel is a chain of dependent round trips, more blocks in flight hide them)
    uint32_t blockWords = max(64u, div_up(words, 1000u));
    blockWords = (blockWords + 3u) & ~3u;
    const uint32_t grid = div_up(words, blockWords);
    if (grid > kVisMaxBlocks) return fail(GS_ERR_INVALID_ARGUMENT, "visible sort: too many blocks");
    r->visControlIdx ^= 1;
    VisControl* vc = r->visControl + r->visControlIdx;             // zeroed by the previous visible sort (or at allocation)
    VisControl* nextVc = r->visControl + (r->visControlIdx ^ 1);
    r->depthControlIdx ^= 1;
    SortControl* control = r->depthControl + r->depthControlIdx;
    SortControl* nextControl = r->depthControl + (r->depthControlIdx ^ 1);
    const bool byMatrix = r->visHistDepth > 0;                     // a sort has been made on this base: keys under its matrix
    const bool byRank = !byMatrix && !r->visBaseIdentity;         // none yet, but the base is an order buffer: its ranks are the keys
    const bool sorted = byMatrix || byRank;                        // neither: CSSetIndices' order = the index order of the compaction
    static const float zeroRow[4] = { 0.f, 0.f, 0.f, 0.f };
    const float* row = byMatrix ? r->visHist[0] : zeroRow;
    r->depthSort.histCopies = hist_copies((int)grid);
    r->lastDepthPasses = sorted ? 4u : 0u;
    prof_record(r, 0, st);
#define GS_LAUNCH_VK(F, HI, RK) hipLaunchKernelGGL((visible_keys_kernel<F, HI, RK>), dim3(grid), dim3(VTHREADS), 0, st, a, row[0], row[1], row[2], row[3], (const uint32_t*)r->visBaseRank, \
                                               (const unsigned long long*)r->visMask, words, blockWords, r->visKeys, r->visIdx, control->hist, vc, (uint32_t*)nextVc, \
                                               r->depthSort.groupAgg, sort_group_words(r->depthSort, n, 4), (uint32_t*)nextControl, r->depthSort.histCopies)
#define GS_LAUNCH_VKF(F) do { if (byRank) GS_LAUNCH_VK(F, true, true); else if (sorted) GS_LAUNCH_VK(F, true, false); else GS_LAUNCH_VK(F, false, false); } while (0)
    switch (a.posFmt) { case 0: GS_LAUNCH_VKF(0); break; case 1: GS_LAUNCH_VKF(1); break; case 2: GS_LAUNCH_VKF(2); break; default: GS_LAUNCH_VKF(3); break; }
#undef GS_LAUNCH_VKF
#undef GS_LAUNCH_VK
    GS_HIP(hipGetLastError());
    prof_record(r, 1, st);
    if (sorted) {
        // the pass shape follows the visible count of the last draw that reported (the host only knows the bound N)
        const uint32_t lastVisible = (r->hostReport && r->frameInFlight) ? *(volatile uint32_t*)&r->hostReport->visible : 0u;
        // the fix-up reads the sorted keys.  Not needed while ties are already in the base order: one matrix on the identity (a stable sort of
        // the index-ordered compaction), or ranks as keys (no ties at all)
        const bool needFix = byMatrix && (r->visHistDepth > 1 || !r->visBaseIdentity);
        GS_TRY(enqueue_sort_passes(ctx, st, r->depthSort, control, r->visKeys, r->visIdx, n, &vc->count, 4, 255u, r, 10, 8, nullptr, !needFix,
                                   lastVisible ? lastVisible : max(n / 3u, 1u)));
        if (needFix) {
            TieHistory H;
            tie_history(r, H);
            const uint32_t tgrid = max(1u, min(div_up(n, (uint32_t)TIE_SEG), (uint32_t)ctx->cuCount * 8u));
            if (r->visBaseIdentity)
                hipLaunchKernelGGL(tie_fix_kernel<TB_INDEX>, dim3(tgrid), dim3(TIE_THREADS), 0, st, a, H, (const uint32_t*)r->visKeys, r->visIdx, (const uint32_t*)&vc->count, n, vc,
                                   (const uint32_t*)nullptr, r->depthSort.altKeys, (uint32_t*)nullptr);
            else
                hipLaunchKernelGGL(tie_fix_kernel<TB_RANK>, dim3(tgrid), dim3(TIE_THREADS), 0, st, a, H, (const uint32_t*)r->visKeys, r->visIdx, (const uint32_t*)&vc->count, n, vc,
                                   (const uint32_t*)r->visBaseRank, r->depthSort.altKeys, (uint32_t*)nullptr);
            GS_HIP(hipGetLastError());
        }
    }
    prof_record(r, 2, st);
    r->visOrderValid = true;
    return GS_OK;
}

} // namespace gs
