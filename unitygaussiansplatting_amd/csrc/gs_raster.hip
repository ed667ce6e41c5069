// gs_raster.hip -- the splat "draw call" and the composite, re-expressed as compute for gfx950.
//
// Replaces RenderGaussianSplats.shader (instanced quad VS :35-77, gaussian FS :79-108, fixed-function
// "Blend OneMinusDstAlpha One" :10-12 into an RGBA16F target, GaussianSplatRenderer.cs:156-166,194-196) and
// GaussianComposite.shader:25-39.  MI355X has no rasteriser/ROP, so the draw is:
//   0. (gs_view.hip) calc_view culls each splat like the rasteriser would and writes, in splat order, a 32-byte
//                  record rec[s] (centre, axes, rgba16f), its inclusive PIXEL rectangle rect[s] and 1 visibility bit.
//   1. bin_emit:   for sorted position i (front to back): visibility byte, then gather rect[order[i]] (8 B), turn it into a
//                  rectangle of tiles of the draw's tile shape (16x16, 32x16 or 32x32 pixels: pick_tile_shape) and emit one
//                  (tile, splat) pair per overlapped tile.  Pair offsets come from a single-pass chained scan (decoupled
//                  look-back) so pairs are emitted in i order.  Persistent grid, ticketed partitions.
//   2. pair sort:  STABLE Onesweep sort of the pairs by tile id only (2 passes for <= 65536 tiles): every
//                  tile's list is then already depth ordered -- no per-tile depth sort.
//   3. ranges:     tile -> [start, end) in the sorted pair array.
//   4. blend:      one workgroup per tile (4, 8 or 16 waves), one pixel per lane, each wave owns an 8x8 quadrant; tiles are
//                  scheduled by descending cost (the schedule is made by an extra workgroup of bin_emit meanwhile).
//                  The tile's list is streamed in batches of one record per thread staged in LDS (software-pipelined loads); each
//                  wave culls the batch against its quadrant (bounding box + separating-axis test) with a ballot and walks
//                  only the survivors, reading the record with wave-uniform LDS loads, blending front-to-back in registers.
//                  It also performs a pending gs_target_clear (it writes every pixel of the target).
// Semantics (DESIGN.md "compositor semantics") are those of the reference rasteriser: oriented quad |q|<=2,
// alpha = saturate(exp(-|q|^2) * a), discard < 1/255, dst = src*(1-dst.a) + dst, fp16 rounding per blend.
#include "gs_common.h"
#include <cstdlib>

namespace gs {

namespace {

struct RasterConsts {
    float W, H, nearClip, farClip;
    uint32_t tilesX, tilesY, width, height;
};

constexpr unsigned long long BFLAG_AGG = 1ull << 62, BFLAG_INCL = 2ull << 62, BVAL_MASK = (1ull << 62) - 1ull;
constexpr uint32_t BIN_SPIN_LIMIT = 1u << 24;

// broadcast lane `b` (wave-uniform) of x to every lane through v_readlane_b32: the result is a scalar operand
__device__ __forceinline__ float rl(float x, int b) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), b)); }

// inclusive sum-scan over the 64 lanes in 6 DPP steps (v_add_u32 with a DPP source; lanes without a source add 0)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#define GS_DPP(x, ctrl, rowmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rowmask, 0xf, false)
    v += GS_DPP(v, 0x111, 0xf);             // row_shr:1
    v += GS_DPP(v, 0x112, 0xf);             // row_shr:2
    v += GS_DPP(v, 0x114, 0xf);             // row_shr:4
    v += GS_DPP(v, 0x118, 0xf);             // row_shr:8
    v += GS_DPP(v, 0x142, 0xa);             // row_bcast:15 -> rows 1 and 3
    v += GS_DPP(v, 0x143, 0xc);             // row_bcast:31 -> rows 2 and 3
#undef GS_DPP
    return v;
}

// o / w and o % w for the slot o of a tile rectangle w tiles wide (o < w * h <= 2^24 tiles, w < 2^16).  A u32 division
// compiles to ~22 VALU instructions, six of them quarter-rate 32-bit multiplies, and this runs once per emitted pair:
// instead, the fp32 reciprocal estimate (v_rcp_f32, 1 ulp) gives floor(o / w) +- 1 while o < 2^20 (relative error of
// the product <= 2^-21.9, so the estimate is within 0.27 of the true quotient), and one correction step in either
// direction with 24-bit multiplies makes it exact.  Larger rectangles (a splat covering more than a million tiles) take
// the integer division.
__device__ __forceinline__ void slot_to_xy(uint32_t o, uint32_t w, uint32_t& ty, uint32_t& tx) {
    if (__builtin_expect(__ballot(o >= (1u << 20)) != 0ull, 0)) { ty = o / w; tx = o - ty * w; return; }
    const uint32_t q = (uint32_t)((float)o * __builtin_amdgcn_rcpf((float)w));
    const int r = (int)o - (int)__umul24(q, w);
    const int lo = r < 0 ? 1 : 0, hi = r >= (int)w ? 1 : 0;      // at most one of them (selects, no branches)
    ty = q - (uint32_t)lo + (uint32_t)hi;
    tx = (uint32_t)(r + (lo ? (int)w : 0) - (hi ? (int)w : 0));
}

// LDS histogram add for one digit of the pair key.  Lanes of a wave that hit the same bin would serialise inside the
// LDS atomic unit (64 deep when the digit is constant, as the high digits of a tile id are), so the add is
// aggregated per distinct value with a ballot loop when few values are present.
__device__ __forceinline__ void hist_add_aggregated(uint32_t* h, uint32_t d, bool active) {
    unsigned long long todo = __ballot(active);
    int guard = 0;
    while (todo && guard < 4) {                            // up to 4 distinct values are handled by one lane each
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t dv = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
        const unsigned long long same = __ballot(active && d == dv);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[dv], (uint32_t)__popcll(same));
        todo &= ~same;
        active = active && d != dv;
        ++guard;
    }
    if (active) atomicAdd(&h[d], 1u);                      // many distinct values left: plain per-lane adds
}

// inclusive max-scan over the 64 lanes in 6 DPP steps (v_max_u32 with a DPP source: no LDS / bpermute round trips)
__device__ __forceinline__ uint32_t wave_incl_max_scan_dpp(uint32_t v) {
#define GS_DPP(x, ctrl, rowmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rowmask, 0xf, false)
    v = max(v, GS_DPP(v, 0x111, 0xf));      // row_shr:1
    v = max(v, GS_DPP(v, 0x112, 0xf));      // row_shr:2
    v = max(v, GS_DPP(v, 0x114, 0xf));      // row_shr:4
    v = max(v, GS_DPP(v, 0x118, 0xf));      // row_shr:8
    v = max(v, GS_DPP(v, 0x142, 0xa));      // row_bcast:15 -> rows 1 and 3
    v = max(v, GS_DPP(v, 0x143, 0xc));      // row_bcast:31 -> rows 2 and 3
#undef GS_DPP
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

// The draw's report, straight into the host's mapped memory (visible to it once the stream is idle) -- written before the
// blend's tiles run, so that an overflowing scene is noticed within the pipeline depth.
__device__ __forceinline__ void write_report(const BinControl* binCtl, const uint32_t* pairSortError, FrameReport* report, uint32_t tileShape) {
    report->pairCount = binCtl->pairCount; report->binError = binCtl->error; report->visible = binCtl->visible;
    report->pairSortError = *pairSortError;
    report->tileShape = tileShape;                  // log2 tile width | log2 tile height << 8: what pairCount counts
    report->tieLongRuns = binCtl->tieLongRuns; report->tieLongest = binCtl->tieLongest;      // (GS_SORT_VISIBLE draws; else 0)
}

// Scheduling order of the blend's workgroups: tiles by descending expected cost (the hardware hands out workgroups in
// blockIdx order), a counting sort over 256 cost buckets by one workgroup.  The cost is a hint, never a result: the
// batches the tile walked in an earlier draw (tileCost), else a guess from the length of its list; empty tiles last.
__device__ __forceinline__ uint32_t tile_bucket(uint32_t len, uint32_t lastCost) {   // 0 = most expensive ... 255 = empty
    if (len == 0) return 255u;
    // the cost the tile reported in an earlier frame (blend_kernel: critical chain + batches, units of 4); no history: a third of a long list at most
    const uint32_t pred = lastCost ? lastCost : min((len + 255u) >> 8, 12u) * 16u;
    return 254u - min(pred, 254u);
}
// The cost hint of tile t is one or two draws old and the image moves (5 px per frame on the C2 orbit): the prediction is the
// maximum over the tile and its 8 neighbours, which keeps a heavy tile early when its content has moved next door
// (scheduling a light tile too early costs nothing; a heavy one too late costs the tail of the launch).
__device__ __forceinline__ uint32_t tile_cost_dilated(const uint32_t* tileCost, uint32_t t, uint32_t tilesX, uint32_t numTiles) {
    // nine UNCONDITIONAL loads (neighbour coordinates clamped to the grid: a duplicate changes no maximum) -- a load inside a
    // branch makes the compiler wait for it there, i.e. nine dependent round trips per tile instead of one
    const uint32_t tilesY = numTiles / tilesX;                  // the grid is exactly tilesX x tilesY
    const uint32_t y = t / tilesX, x = t - y * tilesX;
    const uint32_t x0 = x > 0u ? x - 1u : 0u, x1 = min(x + 1u, tilesX - 1u);
    const uint32_t y0 = y > 0u ? y - 1u : 0u, y1 = min(y + 1u, tilesY - 1u);
    const uint32_t r0 = y0 * tilesX, r1 = y * tilesX, r2 = y1 * tilesX;
    const uint32_t c[9] = { tileCost[r0 + x0], tileCost[r0 + x], tileCost[r0 + x1], tileCost[r1 + x0], tileCost[r1 + x], tileCost[r1 + x1],
                            tileCost[r2 + x0], tileCost[r2 + x], tileCost[r2 + x1] };
    uint32_t m = c[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) m = max(m, c[i]);
    return m;
}
// s_cnt, s_off: 256 words each, s_w: 4 words of LDS; called by every thread of a workgroup of `nthreads` >= 256 threads.
// tileStart == null: the list lengths of the draw are not known yet (the schedule is made while the draw's pairs are still
// being emitted): cost hint only, a tile without one counts as cheap.
__device__ __forceinline__ void tile_order_body(const uint32_t* __restrict__ tileStart, const uint32_t* __restrict__ tileEnd,
                                                const uint32_t* tileCost, uint32_t numTiles, uint32_t tilesX, uint32_t* __restrict__ tileOrder,
                                                uint32_t* s_cnt, uint32_t* s_off, uint32_t* s_w, uint32_t nthreads) {
    auto bucket = [&](uint32_t t) -> uint32_t {
        const uint32_t c = tile_cost_dilated(tileCost, t, tilesX, numTiles);
        if (tileStart) return tile_bucket(tileEnd[t] - tileStart[t], c);
        return 254u - min(c, 254u);
    };
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
    // four tiles per thread and step: their 36 cost loads are in flight together (the workgroup shares a kernel with the
    // binning; one tile at a time it took longer than the binning itself at 1080p: 8,160 tiles, 64 dependent round trips)
    for (uint32_t t0 = tid; t0 < numTiles; t0 += 4u * nthreads) {
        uint32_t bk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t t = t0 + (uint32_t)i * nthreads; bk[i] = t < numTiles ? bucket(t) : 0xffffffffu; }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (bk[i] != 0xffffffffu) atomicAdd(&s_cnt[bk[i]], 1u);
    }
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (tid < 256) {
        v = s_cnt[tid];
        incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane == 63) s_w[w] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t base = 0;
        for (int k = 0; k < w; ++k) base += s_w[k];
        s_off[tid] = base + incl - v;
    }
    __syncthreads();
    // (both sweeps must see the same cost of a tile, or a bucket would overflow: tileCost is the buffer of an EARLIER draw,
    // which nobody writes while this runs -- the draw in flight writes the other one)
    for (uint32_t t0 = tid; t0 < numTiles; t0 += 4u * nthreads) {
        uint32_t bk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t t = t0 + (uint32_t)i * nthreads; bk[i] = t < numTiles ? bucket(t) : 0xffffffffu; }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (bk[i] != 0xffffffffu) tileOrder[atomicAdd(&s_off[bk[i]], 1u)] = t0 + (uint32_t)i * nthreads;
    }
}
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ tileStart, const uint32_t* __restrict__ tileEnd,
                                                          const uint32_t* __restrict__ tileCost, uint32_t numTiles, uint32_t tilesX, uint32_t* __restrict__ tileOrder,
                                                          const BinControl* __restrict__ binCtl, const uint32_t* __restrict__ pairSortError,
                                                          FrameReport* __restrict__ report, uint32_t tileShape) {
    __shared__ uint32_t s_cnt[256], s_off[256], s_w[4];
    if (threadIdx.x == 0) write_report(binCtl, pairSortError, report, tileShape);
    tile_order_body(tileStart, tileEnd, tileCost, numTiles, tilesX, tileOrder, s_cnt, s_off, s_w, 1024u);
}

// Binning: one workgroup per partition of kBinPart consecutive SORTED positions (front to back), handed out by ticket.
// Wave w owns positions [w*1024, (w+1)*1024) of the partition, item (k, lane) = position k*64 + lane, so every load of
// order[] is a coalesced 256-B row.  Pair offsets = exclusive scan of the per-position tile counts in position order:
// inside the wave by DPP scans, across waves through LDS, across partitions by a two-level scan (group aggregates + the
// status words of the own group, see below) done by wave 0.  Emission is output-centric: the wave's pairs of 256 positions
// are produced 64 consecutive output slots at a time, each lane finding its source position from marks dropped at the
// first slot of every position and a DPP max-scan -- global stores of pairs are therefore fully coalesced whatever the
// footprints are (a splat covering the whole screen is just a long run), and no lane idles behind a neighbour's big splat.
// TILECNT (targets of <= kBinTileCounters tiles): the pair sort's digit histograms are not accumulated per emitted pair (one LDS atomic for
// the low digit + a ballot-aggregated add per higher digit: ~35 of the emission's ~140 wave instructions per 64 pairs) but as ONE LDS
// atomic per pair into a per-TILE counter -- neighbouring output slots are neighbouring tiles: distinct addresses -- folded into the
// digit histograms once, when the persistent workgroup is done.
constexpr uint32_t kBinTileCounters = 2048;
template <int PASSES, bool TILECNT>
__global__ __launch_bounds__(kBinThreads) void bin_emit_kernel(const uint2* __restrict__ rects, const uint8_t* __restrict__ waveFlags,
                                                                const uint32_t* __restrict__ order, uint32_t n, uint32_t tilesX, uint32_t tileShift,
                                                                uint32_t* __restrict__ pairKeys, uint32_t* __restrict__ pairVals,
                                                                uint32_t capacity, BinControl* ctl, unsigned long long* binStatus, unsigned long long* binGroupAgg, unsigned long long* binGroupBase,
                                                                uint32_t* pairHist, unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords,
                                                                uint32_t* __restrict__ nextArena, uint32_t nextArenaWords, uint32_t digitBitsAndFlags,
                                                                const uint32_t* __restrict__ schedCost, uint32_t schedTiles, uint32_t* __restrict__ schedOut, uint32_t histCopies) {
    GS_CHAIN_PRIORITY();
    constexpr int SUB = 4;                                   // k's per emission batch: 256 positions per wave
    __shared__ uint32_t s_hist[3 * 256];
    __shared__ uint32_t s_tile[TILECNT ? kBinTileCounters : 1];
    __shared__ uint32_t s_wtot[4], s_wvis[4];
    __shared__ uint32_t s_part;
    __shared__ unsigned long long s_base;
    __shared__ uint4 s_rec[4][SUB * 64];                     // per staged position: first output slot, splat index, tile rectangle (one 16-byte LDS access each way)
    __shared__ uint32_t s_mark[4][64];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // One extra workgroup (the last) is not a binning workgroup: it makes the blend's tile schedule of THIS draw from the costs
    // the previous draw left (tile_order_body, cost hint only) while the others bin -- this kernel is latency-bound with idle
    // issue slots, and the one-workgroup schedule kernel (9 us of pure latency between the pair sort and the blend) then only
    // runs when there is no cost history for the tile count.
    // It is workgroup 0 -- dispatched first, beside the others; as the last one it would only get a slot of the persistent grid
    // when the binning is over (measured: +11 us).
    const uint32_t binBlocks = gridDim.x - (schedOut ? 1u : 0u);
    if (schedOut && blockIdx.x == 0u) {
        tile_order_body(nullptr, nullptr, schedCost, schedTiles, tilesX, schedOut, s_hist, s_hist + 256, s_hist + 512, (uint32_t)kBinThreads);
        return;
    }
    const uint32_t bid = blockIdx.x - (schedOut ? 1u : 0u);      // index among the binning workgroups
    for (int j = tid; j < 3 * 256; j += kBinThreads) s_hist[j] = 0;
    if (TILECNT) for (uint32_t j = tid; j < kBinTileCounters; j += kBinThreads) s_tile[j] = 0;
    for (uint32_t j = bid * (uint32_t)kBinThreads + (uint32_t)tid; j < groupAggWords; j += binBlocks * (uint32_t)kBinThreads) groupAgg[j] = 0ull;   // for the pair sort's look-back
    // the zero-initialised per-draw arena of the NEXT draw (the two copies alternate: no memset launch per draw)
    for (uint32_t j = bid * (uint32_t)kBinThreads + (uint32_t)tid; j < nextArenaWords; j += binBlocks * (uint32_t)kBinThreads) nextArena[j] = 0u;
    const uint32_t numParts = (n + kBinPart - 1) / kBinPart;
    const uint32_t digitBits = digitBitsAndFlags & 0xffu;
    const bool staticFirst = (digitBitsAndFlags & 0x100u) != 0u;   // the context shares the GPU with nobody (gs_shared_gpu() false): see below
    const uint32_t digitMask = (1u << digitBits) - 1u;          // the pair sort's digit width (6..8 bits by tile count)
    uint32_t visAcc = 0;                                         // thread 0: visible splats of this workgroup's partitions
    // Persistent grid.  A partition's scan waits on the totals of every partition before it, so -- as in gs_sort.hip -- partitions are taken in dependency
    // order: when the grid covers every partition binning workgroup b takes partition b (no atomic) and exits; otherwise EVERY partition comes from ONE
    // counter, so that the lowest unclaimed partition is always taken by a workgroup that is running (a static first round would leave it with a workgroup
    // that may not have been dispatched while another stream's kernels hold the wave slots: gs_sort.hip).  A context that shares the GPU with no other
    // spinning kernel (staticFirst) keeps the static first round: the ~1280 simultaneous requests at the head of the kernel cost 16 us at C2.
    const bool oneRound = binBlocks >= numParts;
    for (uint32_t round = 0;; ++round) {
    __syncthreads();                                             // s_part / s_wtot / s_base of the previous partition are no longer read
    if (tid == 0)
        s_part = oneRound ? (round == 0u ? bid : numParts)
               : (staticFirst && round == 0u) ? bid
               : (staticFirst ? binBlocks : 0u) + __hip_atomic_fetch_add(&ctl->tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t part = s_part;
    if (part >= numParts) break;
    const uint32_t waveBase = part * (uint32_t)kBinPart + (uint32_t)w * (64u * kBinItems);

    // ---- per sorted position: gather the splat's tile rectangle (8 B, written by calc_view) ----------------------
    uint32_t sid[kBinItems];
    uint2 rc[kBinItems];
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) {
        const uint32_t i = waveBase + (uint32_t)k * 64u + (uint32_t)lane;
        sid[k] = (i < n) ? order[i] : 0xffffffffu;
    }
    // 1 byte per WAVE of calc_view first (96 KB for 6.1 M splats, cache resident), so that only positions
    // whose 64 index neighbours hold a visible splat pay for the random 8-byte gather of their rectangle from the N x 8 B array
    // (a culled splat beside visible ones has a zero rectangle there).  Visibility is coherent in index = Morton order -- 59 % of
    // C2's waves hold no visible splat, the others are 89 % full -- so this is 2.5 M rectangle requests per frame instead of
    // 6.1 M gathers of a per-splat visibility word + 2.3 M rectangles; this kernel is bound by the L2 request rate.
    uint32_t visw[kBinItems];
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) visw[k] = (sid[k] != 0xffffffffu) ? (uint32_t)waveFlags[sid[k] >> 6] : 0u;
    uint32_t mySum = 0, myVis = 0;
    const uint32_t shx = tileShift & 0xffu, shy = tileShift >> 8;       // log2 of the draw's tile width / height (wave-uniform)
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) {
        rc[k] = make_uint2(0u, 0u);
        if (visw[k]) rc[k] = rects[sid[k]];
        // pixel rectangle {x0 | y0 << 16, (x1 + 1) | (y1 + 1) << 16} (0 = not drawn) -> tile rectangle {tx0 | ty0 << 16, wide | high << 16}
        if (rc[k].y != 0u) {
            const uint32_t tx0 = (rc[k].x & 0xffffu) >> shx, ty0 = (rc[k].x >> 16) >> shy;
            const uint32_t tx1 = ((rc[k].y & 0xffffu) - 1u) >> shx, ty1 = ((rc[k].y >> 16) - 1u) >> shy;
            rc[k] = make_uint2(tx0 | (ty0 << 16), (tx1 - tx0 + 1u) | ((ty1 - ty0 + 1u) << 16));
        }
        const uint32_t c = (rc[k].y & 0xffffu) * (rc[k].y >> 16);
        mySum += c;
        myVis += c ? 1u : 0u;
    }
    const uint32_t waveTotal = wave_sum_u32(mySum);
    const uint32_t waveVis = wave_sum_u32(myVis);
    if (lane == 0) { s_wtot[w] = waveTotal; s_wvis[w] = waveVis; }
    __syncthreads();
    uint32_t wbase = 0, blockTotal = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t t = s_wtot[j]; wbase += (j < w) ? t : 0u; blockTotal += t; }

    // ---- scan across partitions (wave 0).  All partitions of the first round of tickets are resident at once and publish
    //      their totals at about the same time, so a plain decoupled look-back (64 predecessors per step until an inclusive
    //      prefix turns up) is a serial chain: the inclusive front advances 64 partitions per step and the 1280th workgroup
    //      waits ~20 steps.  Two levels instead, with groups of 64 partitions:
    //        * every partition publishes its total (status word) and adds it to its group's word {members:8 | sum:56};
    //        * the FIRST partition of group g sums the words of groups 0..g-1 (each complete once its 64 members have added --
    //          that depends on nobody's look-back) and publishes the group's base;
    //        * every other partition sums the status words of the earlier partitions of its own group (< 64, one step) and
    //          adds the group's base.
    //      (Everybody summing all group words themselves was measured: 1280 waves polling the same ~50 words with agent-scope
    //      loads serialise on a few L2 lines, 0.10 -> 0.18 ms.)
    if (w == 0) {
        const uint32_t grp = part >> 6, grpStart = grp << 6;
        if (lane == 0) {
            __hip_atomic_store(binStatus + part, BFLAG_AGG | (unsigned long long)blockTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(binGroupAgg + grp, (1ull << 56) | (unsigned long long)blockTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            visAcc += s_wvis[0] + s_wvis[1] + s_wvis[2] + s_wvis[3];
        }
        unsigned long long excl = 0;
        uint32_t spins = 0;
        bool failed = false;
        if (part == grpStart) {                                          // ---- first of its group: the base of the group
            int g = (int)grp - 1;
            while (g >= 0 && !failed) {
                const int gi = g - lane;
                unsigned long long agg = 0ull;
                if (gi >= 0) agg = __hip_atomic_load(binGroupAgg + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot(gi >= 0 && (agg >> 56) != 64ull)) {         // a member of one of these groups has not published yet
                    if (++spins > BIN_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl->error, 2u); failed = true; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                unsigned long long v = gi >= 0 ? (agg & ((1ull << 56) - 1ull)) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                excl += v;
                g -= 64;
            }
            if (lane == 0) __hip_atomic_store(binGroupBase + grp, BFLAG_INCL | excl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {                                                         // ---- the earlier partitions of the own group + the group's base
            // First wait for the group's base with ONE request per poll.  A 64-lane poll of the status words is 64 agent-scope (L1-bypassing)
            // L2 requests, and in the steady state ~40 % of the resident workgroups are waiting here at any time: their polling alone
            // loads the L2 request path that the other workgroups' rectangle gathers and pair stores need (per-partition timeline:
            // emission 4-7 us in the first round, when nobody polls, 15-21 us later).  The base is the last thing to arrive -- it needs
            // every earlier group complete -- so once it is there the status words of the own group almost always are, too.
            while (!failed) {
                unsigned long long b0 = 0ull;
                if (lane == 0) b0 = __hip_atomic_load(binGroupBase + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) & (uint32_t)(BFLAG_INCL >> 32)) break;
                if (++spins > BIN_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl->error, 2u); failed = true; }
                __builtin_amdgcn_s_sleep(4);
            }
            for (;;) {
                if (failed) break;
                const int idx = (int)part - 1 - lane;
                const bool mine = idx >= (int)grpStart;
                unsigned long long s = 0ull;
                if (mine) s = __hip_atomic_load(binStatus + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long base = 0ull;
                if (lane == 63) base = __hip_atomic_load(binGroupBase + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // lane 63 never has a predecessor (<= 63 of them)
                if (__ballot((mine && (s & BFLAG_AGG) == 0) || (lane == 63 && (base & BFLAG_INCL) == 0))) {
                    if (++spins > BIN_SPIN_LIMIT) { if (lane == 0) atomicOr(&ctl->error, 2u); failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                unsigned long long v = mine ? (s & BVAL_MASK) : (lane == 63 ? (base & BVAL_MASK) : 0ull);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                excl = v;
                break;
            }
        }
        if (lane == 0) {
            s_base = excl;
            if (part == numParts - 1) {
                const unsigned long long total = excl + blockTotal;
                ctl->pairCount = total;
                ctl->pairCountClamped = (uint32_t)(total < (unsigned long long)capacity ? total : (unsigned long long)capacity);
                if (total > (unsigned long long)capacity) atomicOr(&ctl->error, 1u);
            }
        }
    }
    __syncthreads();
    // global offset of this wave's first pair: wave-uniform, so the pair arrays are addressed as scalar base + 32-bit slot
    // and the capacity test is a 32-bit compare against the number of this wave's slots that fit
    const unsigned long long gbaseV = s_base + wbase;
    const unsigned long long gbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gbaseV >> 32)) << 32) |
                                     (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gbaseV);
    const unsigned long long roomAll = gbase < (unsigned long long)capacity ? (unsigned long long)capacity - gbase : 0ull;
    uint32_t* __restrict__ pk = pairKeys + (roomAll ? gbase : 0ull);
    uint32_t* __restrict__ pv = pairVals + (roomAll ? gbase : 0ull);

    // ---- emit (tile, splat) pairs, 256 positions of this wave at a time (no workgroup barriers below) -------------
    uint4* recsL = s_rec[w];
    uint32_t* mark = s_mark[w];
    mark[lane] = 0u;
    uint32_t run = 0;                                            // wave-local exclusive offset, wave-uniform
#pragma unroll
    for (int sb = 0; sb < kBinItems / SUB; ++sb) {
        const uint32_t subStart = run;
        uint32_t offsR[SUB], cntR[SUB];
#pragma unroll
        for (int kk = 0; kk < SUB; ++kk) {
            const int k = sb * SUB + kk;
            const uint32_t c = (rc[k].y & 0xffffu) * (rc[k].y >> 16);
            const uint32_t incl = wave_incl_scan_u32(c);
            offsR[kk] = run + incl - c; cntR[kk] = c;
            recsL[kk * 64 + lane] = make_uint4(run + incl - c, sid[k], rc[k].x, rc[k].y);
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t carry = 0;                                      // owner (+1) of the slot before j0 (wave-uniform)
        for (uint32_t j0 = subStart; j0 < run; j0 += 64u) {       // wave-uniform trip count
            const uint32_t j = j0 + (uint32_t)lane;
            const uint32_t room = (uint32_t)(roomAll < (unsigned long long)run ? roomAll : (unsigned long long)run);
            // Which position owns output slot j?  Every non-empty position whose first slot falls into this batch of 64
            // drops its index (+1) at that slot; an inclusive max-scan over the lanes (positions ascend with the slots)
            // carries it forward, `carry` across batches.  One LDS write/read pair and 6 DPP steps instead of an 8-step
            // binary search through LDS (8 dependent round trips).
#pragma unroll
            for (int kk = 0; kk < SUB; ++kk) {
                const uint32_t d = offsR[kk] - j0;
                if (cntR[kk] != 0u && d < 64u) mark[d] = (uint32_t)(kk * 64 + lane) + 1u;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = mark[lane];
            __builtin_amdgcn_wave_barrier();
            mark[lane] = 0u;
            const uint32_t e1 = max(wave_incl_max_scan_dpp(m), carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)e1, 63);
            const uint32_t e = max(e1, 1u) - 1u;
            const uint4 rec4 = recsL[e];
            const uint32_t o = j - rec4.x;
            const uint2 r = make_uint2(rec4.z, rec4.w);
            const uint32_t s = rec4.y;
            const uint32_t tw = max(r.y & 0xffffu, 1u);
            uint32_t ty, tx;
            slot_to_xy(o, tw, ty, tx);
            const uint32_t tile = __umul24((r.x >> 16) + ty, tilesX) + (r.x & 0xffffu) + tx;      // rows, columns < 2^16
            const bool wr = j < room;                            // room <= run - of the wave's slots, those below the capacity
            if (wr) {
                const uint32_t boff = j << 2;                    // j < capacity <= 2^30: scalar base + 32-bit byte offset
                *(uint32_t*)((uint8_t*)pk + boff) = tile;
                *(uint32_t*)((uint8_t*)pv + boff) = s;           // payload = splat index: the blend kernel reads rec[splat]
                if (TILECNT) atomicAdd(&s_tile[tile], 1u);       // neighbouring slots are neighbouring tiles: distinct counters
                else atomicAdd(&s_hist[tile & digitMask], 1u);   // ... distinct bins
            }
            if (!TILECNT && PASSES >= 2) hist_add_aggregated(s_hist + 256, (tile >> digitBits) & digitMask, wr);
            if (!TILECNT && PASSES >= 3) hist_add_aggregated(s_hist + 512, (tile >> (2u * digitBits)) & digitMask, wr);
        }
        __builtin_amdgcn_wave_barrier();
    }

    }   // next partition

    // ---- flush the pair-sort digit histograms and the visible count ---------------------------------------------
    if (TILECNT) {                                               // per-tile counts -> digit histograms (the tile grid has <= kBinTileCounters tiles)
        __syncthreads();
        for (uint32_t t = tid; t < kBinTileCounters; t += kBinThreads) {
            const uint32_t c = s_tile[t];
            if (c) {
                atomicAdd(&s_hist[t & digitMask], c);
                if (PASSES >= 2) atomicAdd(&s_hist[256 + ((t >> digitBits) & digitMask)], c);
            }
        }
        __syncthreads();
    }
    if (tid == 0 && visAcc) atomicAdd(&ctl->visible, visAcc);
    uint32_t* myHist = pairHist + (bid % histCopies) * (uint32_t)kHistStride;      // SortControl::hist: one of the copies
    for (int j = tid; j < PASSES * 256; j += kBinThreads) {
        const uint32_t c = s_hist[j];
        if (c) atomicAdd(&myHist[j], c);
    }
}


// ---- GS_SORT_VISIBLE draws: the binning of V depth-sorted, all-visible positions as two streaming kernels -----------------------------
// bin_emit (above) gives every wave its own 256 positions and lets it emit their pairs alone.  Over the FULL order that balances itself:
// near the camera, where the splats are large on screen, most splats are culled, so a partition holds few of them.  The visible order has no
// such dilution -- its first partitions are 1024 of the largest splats each, emitted by four waves while a thousand workgroups wait (measured,
// C3: 308 us against 106 us for the full order; C2d 260 against 99).  Here the pair array itself is what is partitioned:
//   vis_count_kernel     the rectangle gather rects[order[i]] (left by sorted position for the emission), tile counts, block-local offsets,
//                        block and group totals -- thin workgroups, nobody waits;
//   vis_offsets_kernel   global offsets (groups before + earlier blocks of the own group: one load per thread) and, for every chunk of kEmitChunk
//                        output slots, the position its first slot belongs to (the position that straddles the boundary writes it: no search
//                        anywhere); the draw's totals;
//   vis_emit_kernel      one chunk of kEmitChunk output slots per workgroup and step, whatever the footprints are: load the positions that
//                        own them (coalesced: offset, index, rectangle by sorted position -- vis_offsets_kernel made the gather and left the
//                        rectangles there), find every slot's owner with marks + a max-scan, write four consecutive pairs per thread.
// No chunk waits for another one.  Side duties of bin_emit (pair-sort histograms, zeroing the next draw's arena and the pair sort's
// aggregates, visible count, the draw's pair count, the tile schedule) are split between the two.
constexpr uint32_t kEmitChunk = 1024;
constexpr int kEmitThreads = 256;
constexpr int kEmitBatches = 5;                                  // x 256 positions >= kEmitChunk + 1 owners of one chunk

__device__ __forceinline__ uint32_t rect_tiles(uint32_t rx, uint32_t ry, uint32_t shx, uint32_t shy, uint32_t& tx, uint32_t& twh) {
    // pixel rectangle {x0 | y0 << 16, (x1 + 1) | (y1 + 1) << 16} (0 = not drawn) -> tile rectangle {tx0 | ty0 << 16, wide | high << 16}; returns wide * high
    if (ry == 0u) { tx = 0u; twh = 0u; return 0u; }
    const uint32_t tx0 = (rx & 0xffffu) >> shx, ty0 = (rx >> 16) >> shy;
    const uint32_t tx1 = ((ry & 0xffffu) - 1u) >> shx, ty1 = ((ry >> 16) - 1u) >> shy;
    tx = tx0 | (ty0 << 16); twh = (tx1 - tx0 + 1u) | ((ty1 - ty0 + 1u) << 16);
    return (tx1 - tx0 + 1u) * (ty1 - ty0 + 1u);
}

__device__ __forceinline__ uint32_t sat32(unsigned long long v) { return v < 0xffffffffull ? (uint32_t)v : 0xffffffffu; }

// vis_count_kernel: 256 sorted positions per WAVE, four CONSECUTIVE positions per lane (16-byte loads and stores), no barrier anywhere, as many
// waves resident as fit -- the rectangle gather rects[order[i]] is the one random access of the binning (2.2 M sectors at C2) and its rate
// is the misses the CUs keep in flight, i.e. resident waves x independent gathers per lane.  Leaves, by sorted position, the rectangle
// (rectX / rectY) and the wave-LOCAL first pair slot; per wave block its pair total (+ drawn positions), added to the word of its group of 64
// wave blocks.  Nobody waits for anybody.  The grid follows the visible count the last draw reported; it strides, so any count is covered.
constexpr int kVcThreads = 256;
constexpr uint32_t kVcBlock = 256u;                              // positions per wave block
constexpr uint32_t kVcGroup = 64;                                // wave blocks per group word
__global__ __launch_bounds__(kVcThreads) void vis_count_kernel(const uint2* __restrict__ rects, const uint32_t* __restrict__ order, const VisControl* __restrict__ vis,
                                                               uint32_t nImm, uint32_t tileShift, BinControl* ctl, unsigned long long* __restrict__ blockSum,
                                                               unsigned long long* groupSum, uint32_t* __restrict__ pairOffset,
                                                               uint32_t* __restrict__ rectX, uint32_t* __restrict__ rectY,
                                                               unsigned long long* __restrict__ groupAgg, uint32_t groupAggWords, uint32_t* __restrict__ nextArena, uint32_t nextArenaWords) {
    GS_CHAIN_PRIORITY();
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t j = blockIdx.x * (uint32_t)kVcThreads + tid; j < groupAggWords; j += gridDim.x * (uint32_t)kVcThreads) groupAgg[j] = 0ull;      // for the pair sort's look-back
    for (uint32_t j = blockIdx.x * (uint32_t)kVcThreads + tid; j < nextArenaWords; j += gridDim.x * (uint32_t)kVcThreads) nextArena[j] = 0u;       // the NEXT draw's zeroed arena
    const uint32_t V = min(vis->count, nImm);
    if (blockIdx.x == 0u && tid == 0) { ctl->tieLongRuns = vis->tieLongRuns; ctl->tieLongest = vis->tieLongest; }      // the visible sort's fix-up statistics, for the draw's report
    const uint32_t shx = tileShift & 0xffu, shy = tileShift >> 8;
    const uint32_t waves = gridDim.x * (uint32_t)(kVcThreads / 64);
    for (uint32_t wb = blockIdx.x * (uint32_t)(kVcThreads / 64) + (uint32_t)(tid >> 6); (unsigned long long)wb * kVcBlock < V; wb += waves) {      // (wave-uniform)
        const uint32_t b0 = wb * kVcBlock, b1 = min(b0 + kVcBlock, V);
        const uint32_t i0 = b0 + (uint32_t)lane * 4u;
        const bool all4 = i0 + 3u < b1;
        uint32_t o[4];
        if (all4) { const uint4 ov = *(const uint4*)(order + i0); o[0] = ov.x; o[1] = ov.y; o[2] = ov.z; o[3] = ov.w; }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = order[min(i0 + (uint32_t)k, V - 1u)];
        }
        uint2 rc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rc[k] = rects[o[k]];
        uint32_t c[4], drawn = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint32_t a, b; c[k] = (i0 + (uint32_t)k < b1) ? rect_tiles(rc[k].x, rc[k].y, shx, shy, a, b) : 0u; drawn += c[k] ? 1u : 0u; }
        const unsigned long long mine = (unsigned long long)c[0] + c[1] + c[2] + c[3];
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) drawn += __shfl_xor(drawn, d, 64);
        unsigned long long l = incl - mine;
        uint32_t lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = sat32(l); l += c[k]; }
        if (all4) {
            *(uint4*)(rectX + i0) = make_uint4(rc[0].x, rc[1].x, rc[2].x, rc[3].x);
            *(uint4*)(rectY + i0) = make_uint4(rc[0].y, rc[1].y, rc[2].y, rc[3].y);
            *(uint4*)(pairOffset + i0) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i0 + (uint32_t)k < b1) { rectX[i0 + k] = rc[k].x; rectY[i0 + k] = rc[k].y; pairOffset[i0 + k] = lo[k]; }
        }
        if (lane == 63) {
            // {drawn positions:20 | pairs:44}: a wave block holds 256 positions of <= 2^24 tiles each, a group 64 of them: neither field overflows
            const unsigned long long word = ((unsigned long long)drawn << 44) | incl;
            blockSum[wb] = word;
            __hip_atomic_fetch_add(groupSum + wb / kVcGroup, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// vis_offsets_kernel: the same wave blocks again (a launch boundary = every total is there).  A wave block's first slot = the words of the
// groups before its own + the words of the earlier blocks of its group (a few loads per lane, no waiting); the wave-local offsets become global
// ones and every position whose slots straddle a multiple of kEmitChunk claims that chunk for the emission.  Workgroup 0 also leaves the draw's
// totals.
__global__ __launch_bounds__(kVcThreads) void vis_offsets_kernel(const VisControl* __restrict__ vis, uint32_t nImm, uint32_t capacity, BinControl* ctl,
                                                                 const unsigned long long* __restrict__ blockSum, const unsigned long long* __restrict__ groupSum,
                                                                 uint32_t* pairOffset, uint32_t* __restrict__ chunkStart, uint32_t capChunks) {
    GS_CHAIN_PRIORITY();
    constexpr int NW = kVcThreads / 64;
    constexpr unsigned long long PAIRS = (1ull << 44) - 1ull;
    __shared__ unsigned long long s_w64[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t V = min(vis->count, nImm);
    const uint32_t numBlocks = (V + kVcBlock - 1u) / kVcBlock, numGroups = (numBlocks + kVcGroup - 1u) / kVcGroup;
    if (blockIdx.x == 0u) {                                      // the draw's totals (also when nothing is visible)
        unsigned long long pairs = 0ull, drawn = 0ull;          // (the two fields separately: over many groups either would carry)
        for (uint32_t j = (uint32_t)tid; j < numGroups; j += kVcThreads) { const unsigned long long g = groupSum[j]; pairs += g & PAIRS; drawn += g >> 44; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { pairs += __shfl_xor(pairs, d, 64); drawn += __shfl_xor(drawn, d, 64); }
        if (lane == 0) { s_w64[w] = pairs; s_w64[NW + w] = drawn; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long P = 0ull, D = 0ull;
            for (int k = 0; k < NW; ++k) { P += s_w64[k]; D += s_w64[NW + k]; }
            ctl->pairCount = P;
            ctl->pairCountClamped = (uint32_t)(P < (unsigned long long)capacity ? P : (unsigned long long)capacity);
            if (P > (unsigned long long)capacity) atomicOr(&ctl->error, 1u);
            ctl->visible = (uint32_t)D;
        }
    }
    const uint32_t waves = gridDim.x * (uint32_t)NW;
    for (uint32_t wb = blockIdx.x * (uint32_t)NW + (uint32_t)w; (unsigned long long)wb * kVcBlock < V; wb += waves) {      // (wave-uniform)
        const uint32_t b0 = wb * kVcBlock, b1 = min(b0 + kVcBlock, V);
        const uint32_t grp = wb / kVcGroup, grpStart = grp * kVcGroup;
        unsigned long long acc = 0ull;
        for (uint32_t j = (uint32_t)lane; j < grp; j += 64u) acc += groupSum[j] & PAIRS;             // (2^14 positions x 2^24 tiles per group fit the field)
        if (grpStart + (uint32_t)lane < wb) acc += blockSum[grpStart + lane] & PAIRS;                // < 64 earlier wave blocks of the own group
        const unsigned long long total = blockSum[wb] & PAIRS;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        const unsigned long long base = acc;
        const uint32_t i0 = b0 + (uint32_t)lane * 4u;
        uint32_t l[5];
        const bool all4 = i0 + 3u < b1;
        if (all4) { const uint4 lv = *(const uint4*)(pairOffset + i0); l[0] = lv.x; l[1] = lv.y; l[2] = lv.z; l[3] = lv.w; }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) l[k] = (i0 + (uint32_t)k < b1) ? pairOffset[i0 + k] : sat32(total);
        }
        l[4] = (uint32_t)__shfl_down((int)l[0], 1, 64);          // the next lane's first (wave-local offsets: read before anyone writes)
        if (lane == 63 || i0 + 4u >= b1) l[4] = sat32(total);
        uint32_t g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = sat32(l[k] == 0xffffffffu ? 0xffffffffull : base + l[k]);      // (slots beyond the capacity, <= 2^30, are never emitted)
        if (all4) *(uint4*)(pairOffset + i0) = make_uint4(g[0], g[1], g[2], g[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + (uint32_t)k >= b1) continue;
            const unsigned long long off = l[k] == 0xffffffffu ? 0xffffffffull : base + l[k];
            if (!all4) pairOffset[i0 + k] = g[k];
            const uint32_t c = l[k + 1] - l[k];
            if (c && off < 0xffffffffull) {
                const unsigned long long c0 = (off + kEmitChunk - 1ull) / kEmitChunk, c1 = (off + c - 1ull) / kEmitChunk;
                for (unsigned long long cc = c0; cc <= c1 && cc < (unsigned long long)capChunks; ++cc) chunkStart[cc] = i0 + (uint32_t)k;
            }
        }
    }
}

template <int PASSES, bool TILECNT>
__global__ __launch_bounds__(kEmitThreads) void vis_emit_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ rectX, const uint32_t* __restrict__ rectY,
                                                                const uint32_t* __restrict__ pairOffset, const uint32_t* __restrict__ chunkStart,
                                                                const VisControl* __restrict__ vis, uint32_t nImm, const BinControl* __restrict__ ctl,
                                                                uint32_t tilesX, uint32_t tileShift, uint32_t* __restrict__ pairKeys, uint32_t* __restrict__ pairVals,
                                                                uint32_t* pairHist, uint32_t digitBits,
                                                                const uint32_t* __restrict__ schedCost, uint32_t schedTiles, uint32_t* __restrict__ schedOut, uint32_t histCopies) {
    GS_CHAIN_PRIORITY();
    constexpr int NPOS = kEmitBatches * kEmitThreads;
    __shared__ uint32_t s_hist[3 * 256];
    __shared__ uint32_t s_tile[TILECNT ? kBinTileCounters : 1];
    __shared__ uint32_t s_off[NPOS], s_sid[NPOS], s_tx[NPOS], s_twh[NPOS];
    __shared__ uint32_t s_mark[kEmitChunk];
    __shared__ uint32_t s_wmax[kEmitThreads / 64];
    __shared__ uint32_t s_npos;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t binBlocks = gridDim.x - (schedOut ? 1u : 0u);
    if (schedOut && blockIdx.x == 0u) {       // the blend's tile schedule of THIS draw, from the costs the previous draw left (as in bin_emit)
        tile_order_body(nullptr, nullptr, schedCost, schedTiles, tilesX, schedOut, s_hist, s_hist + 256, s_hist + 512, (uint32_t)kEmitThreads);
        return;
    }
    const uint32_t bid = blockIdx.x - (schedOut ? 1u : 0u);
    for (int j = tid; j < 3 * 256; j += kEmitThreads) s_hist[j] = 0;
    if (TILECNT) for (uint32_t j = tid; j < kBinTileCounters; j += kEmitThreads) s_tile[j] = 0;
    const uint32_t V = min(vis->count, nImm);
    const uint32_t Pc = ctl->pairCountClamped;
    const uint32_t shx = tileShift & 0xffu, shy = tileShift >> 8;
    const uint32_t digitMask = (1u << digitBits) - 1u;
    uint32_t p0 = bid * kEmitChunk < Pc ? chunkStart[bid] : 0u;
    for (uint32_t s0 = bid * kEmitChunk; s0 < Pc; s0 += binBlocks * kEmitChunk) {
        const uint32_t s1 = min(s0 + kEmitChunk, Pc);
        const uint32_t sNext = s0 + binBlocks * kEmitChunk;
        const uint32_t p0Next = sNext < Pc ? chunkStart[sNext / kEmitChunk] : 0u;      // (in flight while this chunk is emitted)
        __syncthreads();                                         // the previous chunk's LDS is no longer read
        if (tid == 0) s_npos = 0;
#pragma unroll
        for (int k = 0; k < (int)(kEmitChunk / kEmitThreads); ++k) s_mark[k * kEmitThreads + tid] = 0u;
        __syncthreads();
        // ---- the positions that own slots of [s0, s1): p0, p0 + 1, ... while their first slot is below s1 (offsets ascend).  Two batches of
        //      256 positions are requested at once (a chunk of 1024 slots rarely has more owners); the others only if those were all taken.
        auto place = [&](int k, uint32_t off, uint32_t sid, uint32_t rx, uint32_t ry) {
            const uint32_t j = (uint32_t)k * kEmitThreads + (uint32_t)tid;
            const bool take = p0 + j < V && off < s1;
            if (take) {
                uint32_t tx, twh;
                const uint32_t c = rect_tiles(rx, ry, shx, shy, tx, twh);
                s_off[j] = off; s_sid[j] = sid; s_tx[j] = tx; s_twh[j] = twh;
                // the position's first slot inside the chunk gets its index (+1); one that began in an earlier chunk owns slot 0 onwards
                if (c) s_mark[off > s0 ? off - s0 : 0u] = j + 1u;
            }
            const unsigned long long tk = __ballot(take);
            if (lane == 0 && tk) atomicAdd(&s_npos, (uint32_t)__popcll(tk));
        };
        {
            const uint32_t qa = min(p0 + (uint32_t)tid, V - 1u), qb = min(p0 + kEmitThreads + (uint32_t)tid, V - 1u);      // (V >= 1 here: there are pairs)
            const uint32_t offA = pairOffset[qa], sidA = order[qa], rxA = rectX[qa], ryA = rectY[qa];                           // unconditional loads
            const uint32_t offB = pairOffset[qb], sidB = order[qb], rxB = rectX[qb], ryB = rectY[qb];
            place(0, offA, sidA, rxA, ryA);
            place(1, offB, sidB, rxB, ryB);
        }
        __syncthreads();
        for (int k = 2; k < kEmitBatches && s_npos >= (uint32_t)k * kEmitThreads; ++k) {      // (uniform)
            const uint32_t q = min(p0 + (uint32_t)k * kEmitThreads + (uint32_t)tid, V - 1u);
            place(k, pairOffset[q], order[q], rectX[q], rectY[q]);
            __syncthreads();
        }
        p0 = p0Next;
        // ---- owner of every slot: inclusive max-scan of the marks (positions ascend with the slots); four consecutive slots per thread
        constexpr int SPT = kEmitChunk / kEmitThreads;
        uint32_t m[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) m[k] = s_mark[tid * SPT + k];
#pragma unroll
        for (int k = 1; k < SPT; ++k) m[k] = max(m[k], m[k - 1]);
        const uint32_t wincl = wave_incl_max_scan_dpp(m[SPT - 1]);
        if (lane == 63) s_wmax[w] = wincl;
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wincl, 0x138, 0xf, 0xf, false);      // wave_shr:1: the lanes below (lane 0 gets 0)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kEmitThreads / 64; ++k) prev = max(prev, (k < w) ? s_wmax[k] : 0u);
        uint32_t tiles[SPT], sids[SPT];
        bool live[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const uint32_t slot = s0 + (uint32_t)(tid * SPT + k);
            const uint32_t e = max(max(m[k], prev), 1u) - 1u;
            const uint32_t o = slot - s_off[e];
            const uint32_t twh = s_twh[e], tx0 = s_tx[e];
            uint32_t ty, tx;
            slot_to_xy(o, max(twh & 0xffffu, 1u), ty, tx);
            tiles[k] = __umul24((tx0 >> 16) + ty, tilesX) + (tx0 & 0xffffu) + tx;
            sids[k] = s_sid[e];
            live[k] = slot < s1;
        }
        if (live[SPT - 1]) {                                     // all four: one 16-byte store per array
            static_assert(SPT == 4, "four slots per thread");
            *(uint4*)(pairKeys + s0 + tid * SPT) = make_uint4(tiles[0], tiles[1], tiles[2], tiles[3]);
            *(uint4*)(pairVals + s0 + tid * SPT) = make_uint4(sids[0], sids[1], sids[2], sids[3]);
        } else {
#pragma unroll
            for (int k = 0; k < SPT; ++k) if (live[k]) { pairKeys[s0 + tid * SPT + k] = tiles[k]; pairVals[s0 + tid * SPT + k] = sids[k]; }
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            if (live[k]) {
                if (TILECNT) atomicAdd(&s_tile[tiles[k]], 1u);
                else atomicAdd(&s_hist[tiles[k] & digitMask], 1u);
            }
            if (!TILECNT && PASSES >= 2) hist_add_aggregated(s_hist + 256, (tiles[k] >> digitBits) & digitMask, live[k]);
            if (!TILECNT && PASSES >= 3) hist_add_aggregated(s_hist + 512, (tiles[k] >> (2u * digitBits)) & digitMask, live[k]);
        }
    }
    // ---- flush the pair-sort digit histograms
    __syncthreads();
    if (TILECNT) {
        for (uint32_t t = tid; t < kBinTileCounters; t += kEmitThreads) {
            const uint32_t c = s_tile[t];
            if (c) {
                atomicAdd(&s_hist[t & digitMask], c);
                if (PASSES >= 2) atomicAdd(&s_hist[256 + ((t >> digitBits) & digitMask)], c);
            }
        }
        __syncthreads();
    }
    uint32_t* myHist = pairHist + (bid % histCopies) * (uint32_t)kHistStride;      // SortControl::hist: one of the copies
    for (int j = tid; j < PASSES * 256; j += kEmitThreads) {
        const uint32_t c = s_hist[j];
        if (c) atomicAdd(&myHist[j], c);
    }
}

// tile -> [start, end) in the tile-sorted pair array (both zero in the fresh arena: a tile nothing lands on stays empty).
// Eight keys per thread from two 16-byte loads (the key buffers are 16-byte aligned and padded by 16 entries), plus the one
// key before and the one after them.
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ pairKeys, const uint32_t* nPtr,
                                                          uint32_t* __restrict__ tileStart, uint32_t* __restrict__ tileEnd, uint32_t numTiles) {
    GS_CHAIN_PRIORITY();
    const uint32_t n = *nPtr;
    const uint32_t octs = (n + 7u) >> 3;
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < octs; q += gridDim.x * 256u) {
        const uint32_t j0 = q << 3;
        const uint4 ka = ((const uint4*)pairKeys)[2u * q], kb = ((const uint4*)pairKeys)[2u * q + 1u];
        const uint32_t k[10] = { j0 ? pairKeys[j0 - 1u] : 0xffffffffu, ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w,
                                 (j0 + 8u < n) ? pairKeys[j0 + 8u] : 0xffffffffu };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t j = j0 + (uint32_t)c;
            if (j >= n) break;
            const uint32_t t = k[c + 1];
            if (t >= numTiles) continue;
            if (j == 0 || k[c] != t) tileStart[t] = j;
            if (j == n - 1 || k[c + 2] != t) tileEnd[t] = j + 1;
        }
    }
}

// gfx950 mixed-precision FMA (v_fma_mix*): fp32 fma whose sources may be fp16 halves of a register and whose result
// is either fp32 or RTNE-rounded into one fp16 half of the destination (the other half is preserved).  LLVM selects the
// same instructions for  (half)fmaf(a, b, (float)h)  (so their semantics are fp32-fma-then-round), but only when its
// SLP vectoriser has not packed the maths first; the helpers below pin them.
__device__ __forceinline__ float mix_mul_lo(uint32_t h, float x) {      // (float)lo16(h) * x
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
__device__ __forceinline__ float mix_mul_hi(uint32_t h, float x) {      // (float)hi16(h) * x
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
// x is the result of v_exp_f32: a transcendental's result needs one wait state before a non-transcendental VALU op
// reads it (gfx940+ "trans forwarding" hazard).  hipcc pads its own instructions but not the inside of an asm string.
__device__ __forceinline__ float mix_mul_lo_sat_after_trans(uint32_t h, float x) {  // saturate((float)lo16(h) * x)
    float d;
    asm("s_nop 0\n\tv_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
__device__ __forceinline__ float mix_one_minus_hi(uint32_t h) {         // 1 - (float)hi16(h)
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, 1.0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h));
    return d;
}
// rg = { lo: f16(fma(pr, t, lo16(rg))), hi: f16(fma(pg, t, hi16(rg))) },  ba likewise with (pb, pa).  In place: each half
// reads only itself.  A VALU op that writes HALF a register (op_sel destination) needs one wait state before a VALU op
// reads that register (gfx940+ "dst_sel forwarding" hazard): the two registers are interleaved so that each mixhi is
// one instruction away from the mixlo of its own register, no s_nop needed.
__device__ __forceinline__ void mix_blend4(uint32_t& rg, uint32_t& ba, float pr, float pg, float pb, float pa, float t) {
    asm("v_fma_mixlo_f16 %0, %2, %6, %0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, %6, %1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %6, %0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, %6, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "+v"(rg), "+v"(ba) : "v"(pr), "v"(pg), "v"(pb), "v"(pa), "v"(t));
}

// Accumulator of one pixel.  EXACT mode keeps the four channels as the two packed fp16 dwords the RGBA16F render
// target holds between two blends of the reference's ROP (r | g << 16, b | a << 16).  Each blend is
// f16(fma_f32(src, 1 - A, dst)): the product rgb*alpha is rounded to fp32 first (the fragment shader's output), the fma
// and the RTNE to fp16 are one v_fma_mixlo/hi_f16.  FAST mode accumulates in fp32 and rounds once at the end.
// Splat colour comes packed as in SplatViewData: c0 = f16 r << 16 | f16 g, c1 = f16 b << 16 | f16 a.
__device__ __forceinline__ float half_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
template <int MODE> struct PixelAcc;
template <> struct PixelAcc<0> {
    uint32_t rg, ba;
    __device__ __forceinline__ void load(uint2 d) { rg = d.x; ba = d.y; }
    __device__ __forceinline__ void blend(uint32_t c0, uint32_t c1, float alpha) {
        const float t = mix_one_minus_hi(ba);
        mix_blend4(rg, ba, mix_mul_hi(c0, alpha), mix_mul_lo(c0, alpha), mix_mul_hi(c1, alpha), alpha, t);
    }
    // src = (pr, pg, pb, pa) already premultiplied, fp32 (the box fragment shader's output)
    __device__ __forceinline__ void blend_src(float pr, float pg, float pb, float pa) { const float t = mix_one_minus_hi(ba); mix_blend4(rg, ba, pr, pg, pb, pa, t); }
    __device__ __forceinline__ bool finished() const { return (ba >> 16) == 0x3c00u; }      // A == 1.0: further blends add exactly 0
    __device__ __forceinline__ bool saturated() const { return false; }
    __device__ __forceinline__ uint2 pack() const { return make_uint2(rg, ba); }
};
template <> struct PixelAcc<1> {
    float r, g, b, a;
    __device__ __forceinline__ void load(uint2 d) {
        r = gsm::f16tof32(d.x); g = gsm::f16tof32(d.x >> 16); b = gsm::f16tof32(d.y); a = gsm::f16tof32(d.y >> 16);
    }
    __device__ __forceinline__ void blend(uint32_t c0, uint32_t c1, float alpha) {
        const float t = 1.0f - a;
        r = fmaf(mix_mul_hi(c0, alpha), t, r); g = fmaf(mix_mul_lo(c0, alpha), t, g); b = fmaf(mix_mul_hi(c1, alpha), t, b);
        a = fmaf(alpha, t, a);
    }
    __device__ __forceinline__ void blend_src(float pr, float pg, float pb, float pa) { const float t = 1.0f - a; r = fmaf(pr, t, r); g = fmaf(pg, t, g); b = fmaf(pb, t, b); a = fmaf(pa, t, a); }
    __device__ __forceinline__ bool finished() const { return (1.0f - a) < (1.0f / 4096.0f); }
    __device__ __forceinline__ bool saturated() const { return (1.0f - a) < (1.0f / 4096.0f); }
    __device__ __forceinline__ uint2 pack() const {
        uint2 o;
        o.x = gsm::f32tof16(r) | (gsm::f32tof16(g) << 16);
        o.y = gsm::f32tof16(b) | (gsm::f32tof16(a) << 16);
        return o;
    }
};

typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// One workgroup per tile of 2^TWL x 2^THL pixels (16x16, 32x16 or 32x32: 4, 8 or 16 waves), wave w owns the 8x8 quadrant
// (w % (TW/8), w / (TW/8)), one pixel per lane.  The tile's depth-ordered list is streamed NT = one record per thread at a
// time: thread t gathers rec[pairVals[bs + t]] and stages, in LDS, everything that does not depend on the pixel
// (inverse-scaled axes u_k = axis_k/|axis_k|^2, bounding half extents).  Each wave then
//   (1) tests 64 staged records at once against its quadrant (lane j <-> record j, one ballot), and
//   (2) walks the survivors in order, reading the record with WAVE-UNIFORM LDS loads (two ds_read_b128 per record:
//       they issue on the LDS pipe, not the VALU), so the per-(quadrant, splat) VALU cost is the fragment maths alone.
// The tile shape is the draw's (pick_tile_shape): the (tile, splat) pairs -- what bin_emit emits, the pair sort moves and this
// kernel stages -- shrink with the tile area (a 14-pixel splat touches 3.5 tiles of 16x16 but 2.1 of 32x32).  The survivors of a
// quadrant are the same whatever tile it belongs to, but every wave tests the whole tile's list against its quadrant (64 records per
// ballot, ~40 wave instructions): 0.57 x the pairs at 4 x the waves per record is 2.3 x the tests, +18 % blend time at C2 for 32x32
// (+7 % for 32x16) -- which is why the shape is picked per scene (pick_tile_shape).  Compacting every batch into per-16x16-sub-block
// index lists first (so that a wave tests only its sub-block's records) was built and measured: the extra barrier and the index
// indirection in the survivor walk cost more than the tests saved (+9 % blend, profiles/r04_variants.txt call 3).  So was letting a
// wave LEAVE as soon as its quadrant is finished (s_barrier only counts the waves alive; the others re-deal the list): 22 % of a tile's
// wave x batch slots belong to finished waves, but freeing them changes nothing (+-1 %, call 6) -- the launch is not short of wave slots,
// its SIMDs are VALU-busy 126 of 187 us on average and unevenly loaded.  Blend order per
// pixel is unchanged, so the frame is bit-identical across tile shapes (tests/test_gpu_draw.py::test_tile_shapes_give_the_same_frame).
// DEPTH: the target has a depth attachment (gs_target_set_scene_depth): the reference draws the splats with the default
// ZTest LEqual, ZWrite Off against the camera's depth buffer (RenderGaussianSplats.shader:10; the RT is bound with the
// current depth, GaussianSplatRenderer.cs:195), and all four vertices of a quad carry the centre's depth (:56-60), so a
// fragment survives iff the splat's view depth clip.w <= the opaque scene's view depth at that pixel.
#ifdef GS_BLEND_STATS
// instrumented build only (scripts/blend_stats.py): where the blend's wave-instructions go.  [0] wave x batch stagings, [1] wave x 64-record chunks tested,
// [2] bounding-box hits, [3] survivors walked (wave x record), [4] live fragments blended (lane x record), [5] workgroups that had a list
__device__ unsigned long long g_blendStats[8];
#define GS_STAT(i, v) do { if (lane == 0) atomicAdd(&g_blendStats[i], (unsigned long long)(v)); } while (0)
#else
#define GS_STAT(i, v) do { } while (0)
#endif
#ifdef GS_BLEND_TL
// instrumented build only: per workgroup (in dispatch order): start, end (100 MHz wall clock), batches walked, the most survivors one wave walked, list length,
// tile, HW_ID | XCC_ID << 32, 0.  Two clock reads and one 64-byte store per workgroup: the launch is not slowed measurably (unlike GS_BLEND_STATS' atomics).
__device__ unsigned long long g_blendTl[8192 * 8];
#endif
template <int MODE, bool DEPTH, int TWL, int THL>
__global__ __launch_bounds__(64 << (TWL + THL - 6)) void blend_kernel(const uint32_t* __restrict__ pairVals, const uint32_t* __restrict__ tileStart,
                                                    const uint32_t* __restrict__ tileEnd, const uint32_t* __restrict__ tileOrder,
                                                    uint32_t* __restrict__ tileCost, const SplatRec* __restrict__ recs,
                                                    uint16_t* __restrict__ rt, RasterConsts rc, int dstIsZero,
                                                    const float* __restrict__ recW, const float* __restrict__ sceneDepth,
                                                    const BinControl* __restrict__ binCtl, const uint32_t* __restrict__ pairSortError,
                                                    FrameReport* __restrict__ report) {
    constexpr int TW = 1 << TWL, TH = 1 << THL;                   // tile, pixels
    constexpr int NWX = TW / 8, NW = NWX * (TH / 8);              // waves: one per 8x8 quadrant
    constexpr int NT = 64 * NW;                                   // threads = pixels of the tile = records per batch
    constexpr uint32_t SLOTS = 2048u * 256u / (uint32_t)NT;       // workgroups resident at once on 256 CUs (8 waves per SIMD)
    __shared__ float4 s_a[NT];       // cx, cy, u1x, u2x      (u_k = axis_k / |axis_k|^2; the x's and the y's of the two axes side by side:
    __shared__ uint4 s_b[NT];        // u1y, u2y (float bits), f16 r << 16 | f16 g, f16 b << 16 | f16 a      operands of packed fp32 instructions)
    __shared__ float4 s_e[NT];       // half extents of the footprint's bounding box, pixels; r^2 = ln(255 a) with slack
    __shared__ int s_done;
    __shared__ uint32_t s_cost;
    uint32_t survWalked = 0;                                      // survivors this wave walked (wave-uniform)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef GS_BLEND_TL
    const unsigned long long tl0 = wall_clock64();
#endif
    if (blockIdx.x == 0 && tid == 0) write_report(binCtl, pairSortError, report, (uint32_t)TWL | ((uint32_t)THL << 8));      // the first workgroup to run: before any tile is blended
    const uint32_t tile = tileOrder[blockIdx.x];
    // tiles are dispatched heaviest first (tileOrder); the heaviest also get the higher issue priority on their SIMD, so that the longest
    // survivor chains -- the launch lasts as long as they do -- are not slowed by the light tiles beside them (measured: -1 %)
#ifndef GS_BLEND_NOPRIO
    if (blockIdx.x < SLOTS / 8u) __builtin_amdgcn_s_setprio(3);
    else if (blockIdx.x < 3u * SLOTS / 8u) __builtin_amdgcn_s_setprio(2);
    else if (blockIdx.x < 6u * SLOTS / 8u) __builtin_amdgcn_s_setprio(1);
#endif
    const uint32_t tx = tile % rc.tilesX, ty = tile / rc.tilesX;
    const uint32_t start = tileStart[tile], end = tileEnd[tile];
    const int qx0 = (int)tx * TW + (w % NWX) * 8, qy0 = (int)ty * TH + (w / NWX) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < (int)rc.width && py < (int)rc.height;
    uint2* dst = (uint2*)(rt + ((size_t)py * rc.width + (size_t)px) * 4);
    if (start >= end) {                                        // nothing lands on this tile: target unchanged ...
        if (threadIdx.x == 0) tileCost[tile] = 0;
#ifdef GS_BLEND_TL
        if (threadIdx.x == 0 && blockIdx.x < 8192u) { unsigned long long* q = g_blendTl + blockIdx.x * 8u; q[0] = tl0; q[1] = wall_clock64(); q[2] = 0; q[3] = 0; q[4] = 0; q[5] = tile; q[6] = 0; }
#endif
        if (dstIsZero && inside) *dst = make_uint2(0u, 0u);    // ... or cleared here, when this draw also performs the pending clear
        return;
    }
    uint32_t batchesWalked = 0;

    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const gsm::F2 fxy = { fx, fy };
    const float qminx = (float)qx0 + 0.5f, qmaxx = (float)qx0 + 7.5f, qminy = (float)qy0 + 0.5f, qmaxy = (float)qy0 + 7.5f;

    PixelAcc<MODE> acc;
    float sceneZ = 0.0f;
    if (DEPTH) sceneZ = inside ? sceneDepth[(size_t)py * rc.width + (size_t)px] : 0.0f;
    acc.load((inside && !dstIsZero) ? *dst : make_uint2(0u, 0u));
    if (tid == 0) s_done = 0;
    if (tid == 0) s_cost = 0u;
    bool waveDone = false;

    // Two-deep software pipeline of the staging loads (pair -> splat index -> 32-byte record: two dependent global loads).
    // While batch k is evaluated, the records of batch k+1 and the splat indices of batch k+2 are in flight.  Every load is
    // unconditional (index clamped to the tile's last pair): a load inside a divergent branch makes the compiler wait for
    // it at the end of the branch, which is exactly the latency this is meant to hide.
    const uint32_t lastPair = end - 1u;
    float4 r0, r1;
    float rw = 0.0f;                                              // DEPTH: the record's view depth
    {
        const uint32_t s0 = pairVals[min(start + (uint32_t)tid, lastPair)];
        const float4* rp = (const float4*)(recs + s0);
        r0 = rp[0]; r1 = rp[1];                                   // cx cy a1x a1y | a2x a2y c0 c1
        if (DEPTH) rw = recW[s0];
    }
    uint32_t sidxNext = pairVals[min(start + (uint32_t)NT + (uint32_t)tid, lastPair)];
    for (uint32_t bs = start; bs < end; bs += (uint32_t)NT) {
        __syncthreads();
        if (s_done == NW) break;
        ++batchesWalked;
        GS_STAT(0, 1);
        const uint32_t cnt = min((uint32_t)NT, end - bs);
        if ((uint32_t)tid < cnt) {
            const float inv1 = 1.0f / gsm::dot2f(r0.z, r0.w, r0.z, r0.w);
            const float inv2 = 1.0f / gsm::dot2f(r1.x, r1.y, r1.x, r1.y);
            const float ca = gsm::f16tof32(gsm::f2u(r1.w));
            // bounding box of  quad |q|<=2  INTERSECT  {exp(-|q|^2) a >= 1/255}  (same formula as PrepareSplat)
            const float exr = 2.0f * (fabsf(r0.z) + fabsf(r1.x)), eyr = 2.0f * (fabsf(r0.w) + fabsf(r1.y));
            const float r2 = fmaf(gsm::LogDet(255.0f * ca), 1.0001f, 1.0e-3f);
            const float rr = sqrtf(fmaxf(r2, 0.0f));
            const float exe = rr * sqrtf(gsm::dot2f(r0.z, r1.x, r0.z, r1.x)), eye = rr * sqrtf(gsm::dot2f(r0.w, r1.y, r0.w, r1.y));
            s_a[tid] = make_float4(r0.x, r0.y, r0.z * inv1, r1.x * inv2);
            s_b[tid] = make_uint4(gsm::f2u(r0.w * inv1), gsm::f2u(r1.y * inv2), gsm::f2u(r1.z), gsm::f2u(r1.w));
            s_e[tid] = make_float4(fminf(exr, exe) + 0.02f, fminf(eyr, eye) + 0.02f, r2, rw);
        }
        {
            const float4* rp = (const float4*)(recs + sidxNext);
            r0 = rp[0]; r1 = rp[1];
            if (DEPTH) rw = recW[sidxNext];
            sidxNext = pairVals[min(bs + 2u * (uint32_t)NT + (uint32_t)tid, lastPair)];
        }
        __syncthreads();
        if (!waveDone) {
            for (uint32_t c = 0; c < cnt; c += 64u) {
                const uint32_t j = c + lane;
                bool hit = false;
                GS_STAT(1, 1);
                if (j < cnt) {
                    const float4 ra = s_a[j];
                    const float4 re = s_e[j];
                    hit = (ra.x + re.x >= qminx) && (ra.x - re.x <= qmaxx) && (ra.y + re.y >= qminy) && (ra.y - re.y <= qmaxy);
#ifdef GS_BLEND_STATS
                    { const unsigned long long bb = __ballot(hit); if (__ffsll((long long)__ballot(true)) - 1 == lane) atomicAdd(&g_blendStats[2], (unsigned long long)__popcll(bb)); }
#endif
                    // oriented test: 30 % of the bounding-box survivors cannot put a live fragment on this 8x8 quadrant
                    const uint4 rb = s_b[j];
                    hit = hit && gsm::BlockMayTouch((float)qx0 + 4.0f, (float)qy0 + 4.0f, 3.5f, ra.x, ra.y, ra.z, gsm::u2f(rb.x), ra.w, gsm::u2f(rb.y), re.z);
                }
                unsigned long long mask = __ballot(hit);
                survWalked += (uint32_t)__popcll(mask);
                // one survivor: the fragment of record `rec` (its staged dwords A4, B4) at this lane's pixel
                auto fragment = [&](const float4 A4, const u4v B4, const uint32_t rec) {
                    // q_k = dx u_kx + dy u_ky for both axes at once: three packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma)
                    // instead of six scalar ones, element by element the same operations
                    const gsm::F2 d = fxy - gsm::F2{ A4.x, A4.y };
                    const gsm::F2 q = gsm::fma2(gsm::F2{ d.y, d.y }, gsm::F2{ gsm::u2f(B4.x), gsm::u2f(B4.y) }, gsm::F2{ d.x, d.x } * gsm::F2{ A4.z, A4.w });
                    const float q1 = q.x, q2 = q.y;
                    const float power = -fmaf(q2, q2, q1 * q1);
                    const float y2 = power * 1.44269504088896340736f;                // exp(x) = exp2(x * log2 e), DESIGN.md section 5 #6
                    float alpha = mix_mul_lo_sat_after_trans(B4.w, __builtin_amdgcn_exp2f(y2));
                    const int inQuad = (int)(fmaxf(fabsf(q1), fabsf(q2)) <= 2.0f);     // one v_max with |.| modifiers + one compare (NaN q: not inside)
                    bool live;
                    if (MODE == 0) {
                        // discard decision identical to the oracle's (gsm::DecideAlpha): outside a 16-ulp window around 1/255 the native
                        // alpha decides; inside it -- ~1e-6 of the fragments, a wave-rare branch -- the alpha is recomputed from the
                        // deterministic exp2.  Two float compares (one more than the plain test), the window itself is scalar mask logic.
                        const bool geHi = alpha >= gsm::u2f(gsm::kAlphaWindowLo + gsm::kAlphaWindow);
                        const bool geLo = alpha >= gsm::u2f(gsm::kAlphaWindowLo);
                        live = (inQuad & (int)geHi) != 0;
                        const bool nearT = (inQuad & (int)geLo & (int)!geHi) != 0;
                        if (__builtin_expect(__any(nearT), 0)) {
                            if (nearT) alpha = gsm::DecideAlpha(alpha, y2, half_lo(B4.w), live);
                        }
                    } else {
                        live = (inQuad & (int)(alpha >= 1.0f / 255.0f)) != 0;
                    }
                    if (DEPTH) live = live && (s_e[rec].w <= sceneZ);                  // ZTest LEqual on the quad's (single) depth
                    if (MODE == 1) live = live && !acc.saturated();
#ifdef GS_BLEND_STATS
                    { const unsigned long long lv = __ballot(live); GS_STAT(4, __popcll(lv)); }
#endif
                    if (live) acc.blend(B4.z, B4.w, alpha);
                };
#ifdef GS_BLEND_PREFETCH
                // the survivors' records one ahead: the ds_reads of the next survivor are issued before this one's arithmetic, so a wave that has its SIMD
                // almost to itself (the heavy tiles at the end of the launch: the launch lasts as long as their chains) does not sit out an LDS round trip per
                // survivor.  Two copies of the body alternate between two register sets (no moves).
                if (mask) {
                    int b0 = __ffsll((long long)mask) - 1;
                    mask &= ~(1ull << b0);
                    float4 A0 = s_a[c + b0];
                    u4v B0 = *(const u4v*)&s_b[c + b0];
                    for (;;) {
                        // (the next record is loaded unconditionally -- the last survivor's own record again when there is no next one -- so that the number of
                        // LDS loads in flight at the first use of this one's is known at compile time: s_waitcnt lgkmcnt(2), not (0))
                        const bool more1 = mask != 0;
                        const int b1 = more1 ? __ffsll((long long)mask) - 1 : b0;
                        mask &= ~(1ull << b1);
                        const float4 A1 = s_a[c + b1];
                        u4v B1 = *(const u4v*)&s_b[c + b1];
                        asm volatile("" : "+v"(B0));
                        fragment(A0, B0, c + (uint32_t)b0);
                        if (!more1) break;
                        const bool more0 = mask != 0;
                        b0 = more0 ? __ffsll((long long)mask) - 1 : b1;
                        mask &= ~(1ull << b0);
                        A0 = s_a[c + b0];
                        B0 = *(const u4v*)&s_b[c + b0];
                        asm volatile("" : "+v"(B1));
                        fragment(A1, B1, c + (uint32_t)b1);
                        if (!more0) break;
                    }
                }
#else
                while (mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= ~(1ull << b);                              // one s_bitset0_b64 instead of a 64-bit subtract + and
                    const float4 A4 = s_a[c + b];                      // wave-uniform address: LDS broadcast
                    u4v B4 = *(const u4v*)&s_b[c + b];
                    asm volatile("" : "+v"(B4));                       // keep it ONE ds_read_b128 (no piece sunk into the branch)
                    fragment(A4, B4, c + (uint32_t)b);
                }
#endif
                if (__all(!inside || acc.finished())) {
                    waveDone = true;
                    if (lane == 0) atomicAdd(&s_done, 1);
                    break;
                }
            }
        }
    }
    if (inside) *dst = acc.pack();
    // next frame's scheduling hint (tile_order_body), 1 .. 254: the tile's critical chain -- the most survivors one of its waves
    // walked -- plus NT / 8 per batch (its ballots: 32 for the 256-record batches of a 16x16 tile), in units of 4.  (Batches alone: +2 % at
    // C2 / C3 -- survivors per batch vary tenfold between tiles; the measured duration of the tile schedules worse than either, 0.189 vs
    // 0.185 ms: it depends on who the tile shared its SIMDs with.)
    if (lane == 0) atomicMax(&s_cost, survWalked);
    GS_STAT(3, survWalked);
    if (w == 0) GS_STAT(5, 1);
    __syncthreads();
    if (threadIdx.x == 0) tileCost[tile] = min(254u, (s_cost + batchesWalked * (uint32_t)(NT / 8)) / 4u);
#ifdef GS_BLEND_TL
    if (threadIdx.x == 0 && blockIdx.x < 8192u) {
        unsigned long long* q = g_blendTl + blockIdx.x * 8u;
        uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        q[0] = tl0; q[1] = wall_clock64(); q[2] = batchesWalked; q[3] = s_cost; q[4] = end - start; q[5] = tile; q[6] = hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

// View depth (centerClipPos.w, SplatUtilities.compute:199-200) of every visible splat, for the scene-depth test of the blend:
// only launched by a draw whose target has a depth attachment, so the default path neither computes nor stores it.  Same
// operations as CalcViewGeom, hence the same bits as the w the splat was culled and drawn with.
__global__ __launch_bounds__(256) void splat_depth_kernel(gsm::AssetView a, gsm::FrameConsts P, const uint32_t* __restrict__ visMask32, float* __restrict__ recW) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.n || !((visMask32[idx >> 5] >> (idx & 31u)) & 1u)) return;
    const gsm::V3 pos = gsm::LoadSplatPosChunk(a, idx, blockIdx.x);
    const float wx = gsm::mrow(P.o2w, 0, pos.x, pos.y, pos.z), wy = gsm::mrow(P.o2w, 1, pos.x, pos.y, pos.z), wz = gsm::mrow(P.o2w, 2, pos.x, pos.y, pos.z);
    recW[idx] = gsm::mrow(P.vp, 3, wx, wy, wz);
}

// RenderMode.DebugBoxes / DebugChunkBounds (GaussianDebugRenderBoxes.shader; GaussianSplatRenderer.cs:126-131,156-166): one
// box per splat -- centre = the splat, half axes = 2 * rotation * scale * _SplatScale, colour saturate(col) with alpha
// saturate(opacity * _SplatOpacityScale), drawn through _OrderBuffer -- or one per chunk (the chunk's position bounds, a
// palette colour, alpha 0.1, chunk order), blended "OneMinusDstAlpha One" like the splats.  Same pipeline as the splat draw:
// box_setup writes a 64-byte record + tile rectangle + visibility bit per box, bin_emit / pair sort / tile ranges are shared,
// blend_box evaluates the ray / box test of gs_device_math.h per pixel.
template <bool CHUNKS>
__global__ __launch_bounds__(256) void box_setup_kernel(gsm::AssetView a, gsm::FrameConsts P, gsm::RayConsts ray, uint32_t count, gsm::BoxRec* __restrict__ recs,
                                                        uint2* __restrict__ rects, unsigned long long* __restrict__ visMask, uint8_t* __restrict__ waveFlags) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    bool visible = false;
    uint2 rect = make_uint2(0u, 0u);
    if (idx < count) {
        float c[3], B[9], r, g, b, al;
        if (!CHUNKS) {
            const gsm::V3 pos = gsm::LoadSplatPos(a, idx);
            gsm::V4 q; gsm::V3 sc; float opacity;
            gsm::LoadSplatRotScaleOpacity(a, idx, q, sc, opacity);
            const float sx = sc.x * P.splatScale, sy = sc.y * P.splatScale, sz = sc.z * P.splatScale;
            for (int k = 0; k < 3; ++k) c[k] = gsm::mrow(P.o2w, k, pos.x, pos.y, pos.z);
            const float x = q.x, y = q.y, z = q.z, w = q.w;                           // CalcMatrixFromRotationScale (GaussianSplatting.hlsl:29-46)
            const float m1[9] = { fmaf(-2.0f, fmaf(z, z, y * y), 1.0f) * sx, (2.0f * fmaf(-w, z, x * y)) * sy, (2.0f * fmaf(w, y, x * z)) * sz,
                                  (2.0f * fmaf(w, z, x * y)) * sx, fmaf(-2.0f, fmaf(z, z, x * x), 1.0f) * sy, (2.0f * fmaf(-w, x, y * z)) * sz,
                                  (2.0f * fmaf(-w, y, x * z)) * sx, (2.0f * fmaf(w, x, y * z)) * sy, fmaf(-2.0f, fmaf(y, y, x * x), 1.0f) * sz };
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) B[i * 3 + j] = gsm::dot3f(P.o2w[i * 4], P.o2w[i * 4 + 1], P.o2w[i * 4 + 2], m1[j], m1[3 + j], m1[6 + j]) * 2.0f;
            const gsm::V3 col = gsm::LoadSplatBaseColor(a, idx);
            r = gsm::sat(col.x); g = gsm::sat(col.y); b = gsm::sat(col.z);
            al = gsm::sat(opacity * P.opacityScale);
        } else {
            const uint8_t* ck = a.chunk + (uint64_t)idx * 64;
            float mid[3], half[3];
            for (int k = 0; k < 3; ++k) { const float mn = gsm::u2f(gsm::ld32a(ck, 16 + 8 * k)), mx = gsm::u2f(gsm::ld32a(ck, 20 + 8 * k)); mid[k] = (mn + mx) * 0.5f; half[k] = (mx - mn) * 0.5f; }
            for (int k = 0; k < 3; ++k) c[k] = gsm::mrow(P.o2w, k, mid[0], mid[1], mid[2]);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) B[i * 3 + j] = P.o2w[i * 4 + j] * half[j];
            const float t = (float)idx / (float)count;                                  // palette(t, 0.5, 0.5, 1, (0, 0.33, 0.67))
            r = fmaf(0.5f, cosf(6.28318f * (t + 0.0f)), 0.5f); g = fmaf(0.5f, cosf(6.28318f * (t + 0.33f)), 0.5f); b = fmaf(0.5f, cosf(6.28318f * (t + 0.67f)), 0.5f);
            al = 0.1f;
        }
        gsm::BoxRec rec;
        int x0, x1, y0, y1;
        if (gsm::BuildBox(c, B, ray, P.vp, r, g, b, al, rec, x0, x1, y0, y1)) {
            visible = true;
            rect.x = (uint32_t)x0 | ((uint32_t)y0 << 16);                     // pixel rectangle, as gsm::PackPixelRect
            rect.y = (uint32_t)(x1 + 1) | ((uint32_t)(y1 + 1) << 16);
            uint4* rp = (uint4*)(recs + idx);
            const uint32_t* w32 = (const uint32_t*)&rec;
#pragma unroll
            for (int k = 0; k < 4; ++k) rp[k] = make_uint4(w32[4 * k], w32[4 * k + 1], w32[4 * k + 2], w32[4 * k + 3]);
        }
        rects[idx] = rect;
    }
    const unsigned long long vb = __ballot(visible);
    if ((threadIdx.x & 63u) == 0u && idx < count) visMask[idx >> 6] = vb;
    if ((threadIdx.x & 63u) == 0u && idx < count) waveFlags[idx >> 6] = vb != 0ull ? 1u : 0u;
}

template <int MODE, bool DEPTH>
__global__ __launch_bounds__(256) void blend_box_kernel(const uint32_t* __restrict__ pairVals, const uint32_t* __restrict__ tileStart,
                                                        const uint32_t* __restrict__ tileEnd, const uint32_t* __restrict__ tileOrder,
                                                        uint32_t* __restrict__ tileCost, const gsm::BoxRec* __restrict__ recs, uint16_t* __restrict__ rt,
                                                        RasterConsts rc, gsm::RayConsts ray, int dstIsZero, const float* __restrict__ sceneDepth) {
    __shared__ uint4 s_rec[64 * 4];                               // 64 records of 64 B
    const int tid = threadIdx.x;
    const uint32_t tile = tileOrder[blockIdx.x];
    const uint32_t tx = tile % rc.tilesX, ty = tile / rc.tilesX;
    const uint32_t start = tileStart[tile], end = tileEnd[tile];
    const int px = (int)tx * 16 + (tid & 15), py = (int)ty * 16 + (tid >> 4);
    const bool inside = px < (int)rc.width && py < (int)rc.height;
    uint2* dst = (uint2*)(rt + ((size_t)py * rc.width + (size_t)px) * 4);
    if (start >= end) {
        if (tid == 0) tileCost[tile] = 0;
        if (dstIsZero && inside) *dst = make_uint2(0u, 0u);
        return;
    }
    PixelAcc<MODE> acc;
    acc.load((inside && !dstIsZero) ? *dst : make_uint2(0u, 0u));
    float sceneZ = 0.0f;
    if (DEPTH) sceneZ = inside ? sceneDepth[(size_t)py * rc.width + (size_t)px] : 0.0f;
    float d[3];
    gsm::PixelRay(ray, px, py, d);
    uint32_t walked = 0;
    for (uint32_t bs = start; bs < end; bs += 64u) {
        __syncthreads();
        const uint32_t cnt = min(64u, end - bs);
        {   // 256 threads stage 64 records: thread t copies 16 bytes (quarter t & 3 of record t >> 2)
            const uint32_t rix = (uint32_t)tid >> 2;
            if (rix < cnt) s_rec[tid] = ((const uint4*)(recs + pairVals[bs + rix]))[tid & 3];
        }
        __syncthreads();
        ++walked;
        for (uint32_t j = 0; j < cnt; ++j) {
            const float* R = (const float*)&s_rec[j * 4];        // wave-uniform: LDS broadcast
            float ld[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ld[k] = fmaf(R[k * 3 + 2], d[2], fmaf(R[k * 3 + 1], d[1], R[k * 3] * d[0]));
            const float al = R[15];
            const float t = gsm::BoxFaceDepth(R + 9, ld, al < 0.0f);
            bool live = inside && t > 0.0f && t >= rc.nearClip && t <= rc.farClip;
            if (DEPTH) live = live && (t <= sceneZ);
            if (MODE == 1) live = live && !acc.saturated();
            const float a = fabsf(al);
            if (live) acc.blend_src(R[12] * a, R[13] * a, R[14] * a, a);
        }
    }
    if (inside) *dst = acc.pack();
    if (tid == 0) tileCost[tile] = walked * 8u;
}

// RenderMode.DebugPoints / DebugPointIndices (GaussianDebugRenderPoints.shader; GaussianSplatRenderer.cs:126-131,148-161):
// every splat, in index order (no order buffer), is an opaque screen-space square of _SplatSize pixels around its projected
// centre, colour = saturate(DC colour) or an index code, drawn with ZWrite On + the default ZTest LEqual.  As compute: the
// square's pixels race with a 64-bit atomicMin on {view depth, ~index} (nearest wins; at equal depth the LATER instance, as
// LEqual lets it overwrite), then a second kernel turns the winners into colours and resets the depth words.  Pixel centres
// are sampled with the top-left rule; a square whose centre depth is outside [near, far] is clipped as a whole.
__global__ __launch_bounds__(256) void debug_points_kernel(gsm::AssetView a, gsm::FrameConsts P, float halfSize, uint32_t width, uint32_t height,
                                                           const float* __restrict__ sceneDepth, unsigned long long* __restrict__ zbuf) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= a.n) return;
    const gsm::V3 pos = gsm::LoadSplatPos(a, idx);
    const float wx = gsm::mrow(P.o2w, 0, pos.x, pos.y, pos.z), wy = gsm::mrow(P.o2w, 1, pos.x, pos.y, pos.z), wz = gsm::mrow(P.o2w, 2, pos.x, pos.y, pos.z);
    const float cxc = gsm::mrow(P.vp, 0, wx, wy, wz), cyc = gsm::mrow(P.vp, 1, wx, wy, wz), w = gsm::mrow(P.vp, 3, wx, wy, wz);
    if (!(w >= P.nearClip && w <= P.farClip)) return;
    const float invw = 1.0f / w;
    const float cx = fmaf(0.5f * (cxc * invw), P.screenW, 0.5f * P.screenW);
    const float cy = fmaf(-0.5f * (cyc * invw), P.screenH, 0.5f * P.screenH);
    if (!(gsm::finite32(cx) && gsm::finite32(cy))) return;
    // pixel (i, j) is covered iff cx - h <= i + 0.5 < cx + h (left / top edges belong to the square)
    const float x0f = fmaxf(ceilf((cx - halfSize) - 0.5f), 0.0f), x1f = fminf(ceilf((cx + halfSize) - 0.5f) - 1.0f, (float)width - 1.0f);
    const float y0f = fmaxf(ceilf((cy - halfSize) - 0.5f), 0.0f), y1f = fminf(ceilf((cy + halfSize) - 0.5f) - 1.0f, (float)height - 1.0f);
    if (!(x0f <= x1f && y0f <= y1f)) return;
    const unsigned long long key = ((unsigned long long)gsm::f2u(w) << 32) | (unsigned long long)(0xffffffffu - idx);
    for (int y = (int)y0f; y <= (int)y1f; ++y)
        for (int x = (int)x0f; x <= (int)x1f; ++x) {
            const size_t pi = (size_t)y * width + (size_t)x;
            if (sceneDepth && !(w <= sceneDepth[pi])) continue;
            atomicMin(&zbuf[pi], key);
        }
}

__global__ __launch_bounds__(256) void debug_points_resolve_kernel(gsm::AssetView a, uint32_t numPix, int displayIndex, unsigned long long* __restrict__ zbuf,
                                                                   uint16_t* __restrict__ rt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numPix) return;
    const unsigned long long key = zbuf[i];
    if (key == ~0ull) return;                                   // nothing drawn here: the target keeps what it had
    zbuf[i] = ~0ull;
    const uint32_t idx = 0xffffffffu - (uint32_t)key;
    gsm::V3 c = displayIndex ? gsm::DebugIndexColor(idx, a.n) : gsm::LoadSplatBaseColor(a, idx);
    if (!displayIndex) { c.x = gsm::sat(c.x); c.y = gsm::sat(c.y); c.z = gsm::sat(c.z); }
    uint2 o;
    o.x = gsm::f32tof16(c.x) | (gsm::f32tof16(c.y) << 16);
    o.y = gsm::f32tof16(c.z) | (0x3c00u << 16);                  // frag: half4(color, 1), no blending
    ((uint2*)rt)[i] = o;
}

// GaussianComposite.shader:25-39 + "Blend SrcAlpha OneMinusSrcAlpha" over a constant background
__global__ __launch_bounds__(256) void resolve_kernel(const uint16_t* __restrict__ rt, uint32_t numPix, float bgr, float bgg, float bgb,
                                                      float bga, float* __restrict__ out32, uint8_t* __restrict__ out8) {
    GS_CHAIN_PRIORITY();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= numPix) return;
    const uint2 d = ((const uint2*)rt)[i];
    const float C[3] = { gsm::f16tof32(d.x), gsm::f16tof32(d.x >> 16), gsm::f16tof32(d.y) };
    const float A = gsm::f16tof32(d.y >> 16);
    const float bg[4] = { bgr, bgg, bgb, bga };
    float o[4];
    if (!(A > 0.0f)) { o[0] = bgr; o[1] = bgg; o[2] = bgb; o[3] = bga; }
    else {
        const float invA = 1.0f / A;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = C[c] * invA;
            const float lin = s * fmaf(s, fmaf(s, 0.305306011f, 0.682171111f), 0.012522878f);   // UnityCG GammaToLinearSpace
            o[c] = fmaf(A, lin - bg[c], bg[c]);
        }
        o[3] = fmaf(A, A - bga, bga);        // no separate alpha blend factors in the reference: dst.a = A*A + bg.a*(1-A)
    }
    ((float4*)out32)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (out8) {
        uint32_t pk = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float l = gsm::sat(o[c]);
            const float s = (l <= 0.0031308f) ? 12.92f * l : fmaf(1.055f, powf(l, 1.0f / 2.4f), -0.055f);
            pk |= (uint32_t)floorf(fmaf(gsm::sat(s), 255.0f, 0.5f)) << (8 * c);
        }
        pk |= (uint32_t)floorf(fmaf(gsm::sat(o[3]), 255.0f, 0.5f)) << 24;
        ((uint32_t*)out8)[i] = pk;
    }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

int32_t ensure_arena(gs_renderer* r, uint32_t numTiles) {
    if (r->frameArena && numTiles <= r->arenaTiles) return GS_OK;
    if (r->frameArena) { GS_HIP(hipStreamSynchronize(r->ctx->stream)); (void)hipFree(r->frameArena); r->frameArena = nullptr; }
    size_t off = 0;
    off = align_up(sizeof(BinControl), 256);
    r->offPairControl = off; off += align_up(sizeof(SortControl), 256);
    // (GS_SORT_VISIBLE draws reuse the two arrays: one word per 256 positions = 8 binParts, one per 64 of those)
    r->offBinStatus = off;   off += align_up((size_t)(8u * r->binParts + 1u) * 8, 256);
    r->offBinGroupAgg = off; off += align_up((size_t)((8u * r->binParts + 64u) / 64u + 1u) * 8, 256);
    r->offBinGroupBase = off; off += align_up((size_t)((r->binParts + 63) / 64) * 8, 256);
    r->offTileStart = off;   off += align_up((size_t)numTiles * 4, 256);
    r->offTileEnd = off;     off += align_up((size_t)numTiles * 4, 256);
    off = align_up(off, 256);
    r->frameArenaBytes = off;
    r->arenaTiles = numTiles;
    GS_HIP(hipMalloc((void**)&r->frameArena, 2 * off));
    GS_HIP(hipMemsetAsync(r->frameArena, 0, 2 * off, r->ctx->stream));
    r->arenaIdx = 0;
    // persist across frames (not part of the zeroed arena): the scheduling hints and the schedule itself, two copies each
    if (r->tileCost) (void)hipFree(r->tileCost);
    if (r->tileOrderBuf) (void)hipFree(r->tileOrderBuf);
    r->tileCost = nullptr; r->tileOrderBuf = nullptr;
    GS_HIP(hipMalloc((void**)&r->tileCost, (size_t)2 * numTiles * 4));
    GS_HIP(hipMemsetAsync(r->tileCost, 0, (size_t)2 * numTiles * 4, r->ctx->stream));
    GS_HIP(hipMalloc((void**)&r->tileOrderBuf, (size_t)numTiles * 4));
    r->costIdx = 0; r->costTiles[0] = r->costTiles[1] = 0;
    return GS_OK;
}

} // namespace

ViewOutputs view_outputs(gs_renderer* r) { return ViewOutputs{ r->view, r->recs, r->rects, r->visMask }; }

int32_t renderer_alloc_raster(gs_renderer* r) {
    gs_context* ctx = r->ctx;
    r->binParts = div_up(r->n, kBinPart);
    GS_HIP(hipMalloc((void**)&r->recs, (size_t)r->n * sizeof(SplatRec) + 64));
    GS_HIP(hipMalloc((void**)&r->recW, (size_t)r->n * sizeof(float) + 64));
    GS_HIP(hipMalloc((void**)&r->rects, (size_t)r->n * sizeof(uint2) + 64));
    GS_HIP(hipMemsetAsync(r->rects, 0, (size_t)r->n * sizeof(uint2), ctx->stream));
    GS_HIP(hipMalloc((void**)&r->visMask, vis_alloc_bytes(r->n)));
    GS_HIP(hipMemsetAsync(r->visMask, 0, vis_alloc_bytes(r->n), ctx->stream));
    if (r->pairCapacity == 0) {
        unsigned long long cap = (unsigned long long)r->n * 8ull;
        if (cap < (1ull << 22)) cap = 1ull << 22;
        if (cap > kSortMaxCount) cap = kSortMaxCount;           // 32-bit byte offsets inside the sort kernels
        r->pairCapacity = cap;
    }
    GS_HIP(hipMalloc((void**)&r->pairKeys, ((size_t)r->pairCapacity + 16) * 4));
    GS_HIP(hipMalloc((void**)&r->pairVals, ((size_t)r->pairCapacity + 16) * 4));
    GS_TRY(sort_state_create(ctx, r->pairSort, (uint32_t)r->pairCapacity));
    GS_HIP(hipHostMalloc((void**)&r->hostReport, sizeof(FrameReport), hipHostMallocMapped));
    memset(r->hostReport, 0, sizeof(FrameReport));
    GS_HIP(hipHostGetDevicePointer((void**)&r->hostReportDev, r->hostReport, 0));
    return GS_OK;
}

void renderer_free_raster(gs_renderer* r) {
    if (r->recs) (void)hipFree(r->recs);
    if (r->rects) (void)hipFree(r->rects);
    if (r->recW) (void)hipFree(r->recW);
    if (r->boxRecs) (void)hipFree(r->boxRecs);
    if (r->chunkOrder) (void)hipFree(r->chunkOrder);
    r->recW = nullptr; r->boxRecs = nullptr; r->chunkOrder = nullptr;
    if (r->visMask) (void)hipFree(r->visMask);
    if (r->pairKeys) (void)hipFree(r->pairKeys);
    if (r->pairVals) (void)hipFree(r->pairVals);
    sort_state_destroy(r->pairSort);
    if (r->frameArena) (void)hipFree(r->frameArena);
    if (r->tileCost) (void)hipFree(r->tileCost);
    if (r->tileOrderBuf) (void)hipFree(r->tileOrderBuf);
    r->tileCost = nullptr; r->tileOrderBuf = nullptr;
    if (r->hostReport) (void)hipHostFree(r->hostReport);
    r->recs = nullptr; r->rects = nullptr; r->visMask = nullptr; r->pairKeys = r->pairVals = nullptr; r->frameArena = nullptr; r->hostReport = nullptr; r->hostReportDev = nullptr;
}

namespace {
struct DrawSetup { RasterConsts rc; uint32_t numTiles; uint32_t *tileStart, *tileEnd, *tileOrder, *costWrite; const uint32_t* costRead;
                   const BinControl* binCtl; const uint32_t* pairSortError; int dstIsZero; };

// The part of a draw that does not depend on what a "fragment" is: (tile, item) pairs of the visible items in `order`
// (bin_emit), the stable pair sort by tile, tile ranges, tile schedule + the draw's report.
int32_t bin_and_sort(gs_renderer* r, const gs_frame_params* p, gs_target* rt, const uint32_t* order, uint32_t count, DrawSetup& o, bool forceOrderKernel,
                     uint32_t tileWL, uint32_t tileHL, const VisControl* vis = nullptr) {
    gs_context* ctx = r->ctx;
    hipStream_t st = ctx->stream;
    RasterConsts& rc = o.rc;
    rc.W = (float)rt->width; rc.H = (float)rt->height; rc.nearClip = p->near_clip; rc.farClip = p->far_clip;
    rc.width = rt->width; rc.height = rt->height;
    rc.tilesX = div_up(rt->width, 1u << tileWL); rc.tilesY = div_up(rt->height, 1u << tileHL);
    const uint32_t numTiles = o.numTiles = rc.tilesX * rc.tilesY;
    if (numTiles > (1u << 24)) return fail(GS_ERR_INVALID_ARGUMENT, "target too large (more than 2^24 tiles)");
    GS_TRY(ensure_arena(r, numTiles));
    r->lastTilesX = rc.tilesX; r->lastTilesY = rc.tilesY; r->lastTileWL = tileWL; r->lastTileHL = tileHL;

    r->arenaIdx ^= 1;
    uint8_t* arena = r->frameArena + (size_t)r->arenaIdx * r->frameArenaBytes;          // zeroed by the previous draw's bin_emit (or at allocation)
    uint8_t* nextArena = r->frameArena + (size_t)(r->arenaIdx ^ 1) * r->frameArenaBytes;
    BinControl* binCtl = (BinControl*)arena;
    SortControl* pairCtl = (SortControl*)(arena + r->offPairControl);
    unsigned long long* binStatus = (unsigned long long*)(arena + r->offBinStatus);
    unsigned long long* binGroupAgg = (unsigned long long*)(arena + r->offBinGroupAgg);
    unsigned long long* binGroupBase = (unsigned long long*)(arena + r->offBinGroupBase);
    o.tileStart = (uint32_t*)(arena + r->offTileStart);
    o.tileEnd = (uint32_t*)(arena + r->offTileEnd);
    o.tileOrder = r->tileOrderBuf;
    o.costWrite = r->tileCost + (size_t)r->costIdx * r->arenaTiles;          // this draw's blend writes it ...
    o.costRead = r->tileCost + (size_t)(r->costIdx ^ 1) * r->arenaTiles;      // ... and is scheduled by what the previous draw wrote
    const uint32_t shapeKey = tileWL | (tileHL << 8);
    const bool haveCosts = r->costTiles[r->costIdx ^ 1] == numTiles && r->costShape[r->costIdx ^ 1] == shapeKey;   // (a hint from another tile grid is no hint)
    r->costTiles[r->costIdx] = numTiles; r->costShape[r->costIdx] = shapeKey;
    r->costIdx ^= 1;
    const uint32_t cap = (uint32_t)r->pairCapacity;

    prof_record(r, 3);
    // digit width of the pair sort by tile count: 12-bit tile ids sort in two 6-bit passes (6 ballots per key instead of 8,
    // 64 status words per partition instead of 256), 13..14 bits in two 7-bit passes
    int passes, bits;
    if (numTiles <= 256) { passes = 1; bits = numTiles <= 64 ? 6 : (numTiles <= 128 ? 7 : 8); }
    else if (numTiles <= 65536) { passes = 2; bits = numTiles <= 4096 ? 6 : (numTiles <= 16384 ? 7 : 8); }
    else { passes = 3; bits = 8; }
    const bool tileCnt = numTiles <= kBinTileCounters;       // (<= 2048 tiles means <= 2 passes)
    auto binKernel = passes == 1 ? (tileCnt ? bin_emit_kernel<1, true> : bin_emit_kernel<1, false>)
                   : (passes == 2 ? (tileCnt ? bin_emit_kernel<2, true> : bin_emit_kernel<2, false>) : bin_emit_kernel<3, false>);
    const uint32_t binThreads = kBinThreads;
    // persistent: as many workgroups as are resident at once, a multiple of the ticket classes
    constexpr uint32_t kBinBlocksPerCu = 5;      // 90 VGPRs at 8 positions per thread: five 256-thread workgroups per CU
    const uint32_t binCap = max((uint32_t)ctx->cuCount * kBinBlocksPerCu / kBinTicketClasses * kBinTicketClasses, kBinTicketClasses);
    const uint32_t binGrid = min(div_up(div_up(count, kBinPart), kBinTicketClasses) * kBinTicketClasses, binCap);
    const bool schedInBin = haveCosts && !forceOrderKernel;     // one extra workgroup makes the blend's tile schedule meanwhile
    r->pairSort.histCopies = hist_copies((int)binGrid);
    if (vis) {
        // GS_SORT_VISIBLE: offsets + output-partitioned emission (see vis_offsets_kernel)
        const uint32_t capChunks = div_up(cap, kEmitChunk) + 1u;
        if (r->visChunkCap < capChunks) {
            if (r->visChunkStart) { GS_HIP(hipStreamSynchronize(st)); (void)hipFree(r->visChunkStart); r->visChunkStart = nullptr; r->visChunkCap = 0; }
            GS_HIP(hipMalloc((void**)&r->visChunkStart, (size_t)capChunks * 4));
            r->visChunkCap = capChunks;
        }
        // (blockSum lives in the arena's bin-status words, groupSum -- zeroed, accumulated with atomics -- in its bin-group words: ensure_arena sizes both)
        // (the host only knows the bound N: the grids follow the visible count of the last draw that reported, + 1/8, and stride)
        const uint32_t lastVis = (r->hostReport && r->frameInFlight) ? *(volatile uint32_t*)&r->hostReport->visible : 0u;
        const uint32_t gridFor = lastVis ? min(count, lastVis + lastVis / 8u + 4096u) : count;
        // (8 workgroups per CU = every wave slot; 2 .. 64 per CU measured within 1 us of each other at C2, r05 call 11: the gather is at the
        // memory system's random-sector rate whatever is in flight)
        const uint32_t vcGrid = max(1u, min(div_up(gridFor, kVcBlock * (uint32_t)(kVcThreads / 64)), (uint32_t)ctx->cuCount * 8u));
        hipLaunchKernelGGL(vis_count_kernel, dim3(vcGrid), dim3(kVcThreads), 0, st, (const uint2*)r->rects, order, vis, count, shapeKey, binCtl, binStatus, binGroupAgg,
                           r->visPairOffset, r->visRectX, r->visRectY,
                           r->pairSort.groupAgg, sort_group_words(r->pairSort, cap, passes), (uint32_t*)nextArena, (uint32_t)(r->frameArenaBytes / 4));
        hipLaunchKernelGGL(vis_offsets_kernel, dim3(vcGrid), dim3(kVcThreads), 0, st, vis, count, cap, binCtl, (const unsigned long long*)binStatus, (const unsigned long long*)binGroupAgg,
                           r->visPairOffset, r->visChunkStart, capChunks);
        const uint32_t emitGrid = (uint32_t)ctx->cuCount * (tileCnt ? 4u : 5u);      // resident at once (36 / 28 KB of LDS)
        r->pairSort.histCopies = hist_copies((int)emitGrid);
        auto emitKernel = passes == 1 ? (tileCnt ? vis_emit_kernel<1, true> : vis_emit_kernel<1, false>)
                        : (passes == 2 ? (tileCnt ? vis_emit_kernel<2, true> : vis_emit_kernel<2, false>) : vis_emit_kernel<3, false>);
        hipLaunchKernelGGL(emitKernel, dim3(emitGrid + (schedInBin ? 1u : 0u)), dim3(kEmitThreads), 0, st, order, (const uint32_t*)r->visRectX, (const uint32_t*)r->visRectY,
                           (const uint32_t*)r->visPairOffset, (const uint32_t*)r->visChunkStart, vis, count, (const BinControl*)binCtl, rc.tilesX, shapeKey, r->pairKeys, r->pairVals,
                           pairCtl->hist, (uint32_t)bits, o.costRead, numTiles, schedInBin ? o.tileOrder : (uint32_t*)nullptr, r->pairSort.histCopies);
    } else
    hipLaunchKernelGGL(binKernel, dim3(binGrid + (schedInBin ? 1u : 0u)), dim3(binThreads), 0, st, r->rects, wave_flags_of(r->visMask, r->n), order, count, rc.tilesX, shapeKey, r->pairKeys,
                       r->pairVals, cap, binCtl, binStatus, binGroupAgg, binGroupBase, pairCtl->hist, r->pairSort.groupAgg, sort_group_words(r->pairSort, cap, passes), (uint32_t*)nextArena, (uint32_t)(r->frameArenaBytes / 4),
                       (uint32_t)bits | (gs_shared_gpu(ctx) ? 0u : 0x100u), o.costRead, numTiles, schedInBin ? o.tileOrder : (uint32_t*)nullptr, r->pairSort.histCopies);
    GS_TRY(mark_order_use(r));                                  // the next frame's depth sort may overwrite order[] from here on
    prof_record(r, 4);
    // the host only knows the capacity; the pair count of the last finished frame (pinned report) picks the sort's pass shape
    const unsigned long long lastPairs = r->hostReport ? *(volatile unsigned long long*)&r->hostReport->pairCount : 0ull;
    const uint32_t expectPairs = (uint32_t)(lastPairs < (unsigned long long)cap ? lastPairs : (unsigned long long)cap);
    GS_TRY(enqueue_sort_passes(ctx, st, r->pairSort, pairCtl, r->pairKeys, r->pairVals, cap, &binCtl->pairCountClamped, passes, 255u, r, 12, bits, nullptr, false,
                               expectPairs ? expectPairs : 1u));
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(max(1u, min(div_up(cap, 2048), (uint32_t)ctx->cuCount * 8u))), dim3(256), 0, st, r->pairKeys,
                       &binCtl->pairCountClamped, o.tileStart, o.tileEnd, numTiles);
    r->lastPairPasses = (uint32_t)passes;
    o.binCtl = binCtl; o.pairSortError = &pairCtl->error;
    // the tile schedule: normally made by the extra workgroup of this draw's bin_emit; a kernel of its own (which also knows the
    // list lengths) only when there is no cost history for this tile grid (first draw, another target size) or the caller asks
    if (!schedInBin)
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, o.tileStart, o.tileEnd, o.costRead, numTiles, rc.tilesX, o.tileOrder, binCtl, &pairCtl->error,
                           r->hostReportDev, shapeKey);
    prof_record(r, 5);
    o.dstIsZero = rt->clearPending ? 1 : 0;                      // this draw writes every pixel of the target: the clear is folded in
    rt->clearPending = false;
    return GS_OK;
}
} // namespace

// The compositor's tile shape for a target: 16x16, 32x16 or 32x32 pixels (log2 width, log2 height).  A performance parameter only
// -- frames are bit-identical across shapes (tests/test_gpu_draw.py::test_tile_shapes_give_the_same_frame).  Larger tiles mean fewer
// (tile, splat) pairs to emit, sort and stage, but every wave of the blend tests the whole tile's list against its 8x8 quadrant, so the
// blend pays for them (profiles/r04_variants.txt: C2 3.8 tiles of 16x16 per visible splat -> 32x16 wins by 1.4 %, 32x32 loses 0.6 %;
// C2d 17.5 tiles per splat -> 32x32 wins by 33 %).  So: small targets (< 2^18 pixels: too few large tiles to fill 256 CUs) 16x16;
// otherwise 32x16, and 32x32 while the scene's splats are large -- judged by the tiles per visible splat of the most recent draw that
// reported (pinned host memory, no stream query), with hysteresis.  gs_renderer_set_tile_shape overrides per renderer,
// GSPLAT_TILE=16x16|32x16|32x32 process-wide.
static int forced_tile_shape() {
    static const int forced = [] {
        const char* e = getenv("GSPLAT_TILE");
        if (!e) return 0;
        if (!strcmp(e, "16x16")) return 1;
        if (!strcmp(e, "32x16")) return 2;
        if (!strcmp(e, "32x32")) return 3;
        return 0;
    }();
    return forced;
}
void auto_tile_shape(uint32_t width, uint32_t height, uint32_t& wl, uint32_t& hl) {
    int shape = forced_tile_shape();
    if (!shape) shape = (unsigned long long)width * height >= (1ull << 18) ? 2 : 1;
    wl = shape == 1 ? 4u : 5u; hl = shape == 3 ? 5u : 4u;
}
void pick_tile_shape(const gs_renderer* r, uint32_t width, uint32_t height, uint32_t& wl, uint32_t& hl) {
    if (r && r->tileOverrideWL) { wl = r->tileOverrideWL; hl = r->tileOverrideHL; return; }
    auto_tile_shape(width, height, wl, hl);
    if (r && !forced_tile_shape() && wl == 5u && r->adaptTall) hl = 5u;      // large splats: 32x32 (adapt_tile_shape)
}
// Called once per splat draw, before the shape is picked: 32x16 <-> 32x32 by the tiles per visible splat the last reporting draw saw.
static void adapt_tile_shape(gs_renderer* r) {
    if (!r->hostReport) return;
    const volatile FrameReport* rep = r->hostReport;                         // (a torn read only mis-steers a hint)
    const unsigned long long pairs = rep->pairCount;
    const uint32_t visible = rep->visible, shape = rep->tileShape;
    if (visible < 1024u) return;
    const double ratio = (double)pairs / (double)visible;
    if (shape == (5u | (4u << 8)) && ratio > 6.0) r->adaptTall = true;
    else if (shape == (5u | (5u << 8)) && ratio < 3.5) r->adaptTall = false;
}

// A draw with a SMALLER tile than the last one (the adaptive 32x32 -> 32x16 flip, or gs_renderer_set_tile_shape) lists the same splats
// on up to area-ratio times as many tiles, while the pair buffers are only kept at 1.25 x what the last draw needed: grow them first, so
// that the knob stays what the header says it is -- performance only, the frame bit-identical -- instead of truncating the first frame.
static int32_t reserve_for_smaller_tile(gs_renderer* r, uint32_t twl, uint32_t thl) {
    if (!r->frameInFlight || !r->hostReport || !r->lastTileWL) return GS_OK;
    const uint32_t lastArea = r->lastTileWL + r->lastTileHL, area = twl + thl;      // log2 areas
    if (area >= lastArea) return GS_OK;
    const unsigned long long seen = *(volatile unsigned long long*)&r->hostReport->pairCount;
    unsigned long long want = (seen << (lastArea - area));
    want += want / 4u;
    if (want > kSortMaxCount) want = kSortMaxCount;
    if (want > r->pairCapacity) return gs_renderer_reserve_pairs(r, want);
    return GS_OK;
}

int32_t enqueue_draw(gs_renderer* r, const gs_frame_params* p, gs_target* rt) {
    hipStream_t st = r->ctx->stream;
    // the per-splat footprints were computed by calc_view: it must have run with the same screen size and clip planes
    if (!r->viewValid || r->viewW != (float)rt->width || r->viewH != (float)rt->height || r->viewNear != p->near_clip || r->viewFar != p->far_clip)
        return fail(GS_ERR_INVALID_ARGUMENT, "gs_renderer_draw: call gs_renderer_calc_view with the same screen size / clip planes first");
    GS_TRY(join_sort(r));                                       // bin_emit reads order[]
    DrawSetup ds;
    uint32_t twl, thl;
    adapt_tile_shape(r);
    pick_tile_shape(r, rt->width, rt->height, twl, thl);
    GS_TRY(reserve_for_smaller_tile(r, twl, thl));
    if (vis_active(r)) {
        // GS_SORT_VISIBLE: the depth sort runs HERE, over the splats calc_view found visible (gs_vissort.hip), with the matrix of the last
        // gs_renderer_sort -- also on a frame that did not call it (m_SortNthFrame > 1: the reference's stale order, restricted to this
        // frame's visible set, is this frame's visible set sorted by the stale matrix)
        if (!r->visOrderValid) GS_TRY(enqueue_visible_sort(r));
        GS_TRY(bin_and_sort(r, p, rt, r->visIdx, r->n, ds, false, twl, thl, vis_control(r)));
        r->visDrawn = true;
    } else {
        GS_TRY(bin_and_sort(r, p, rt, r->order, r->n, ds, false, twl, thl));
        r->visDrawn = false;
    }
    const RasterConsts& rc = ds.rc;
    const uint32_t numTiles = ds.numTiles;
    uint32_t *tileStart = ds.tileStart, *tileEnd = ds.tileEnd, *tileOrder = ds.tileOrder;
    const int dstIsZero = ds.dstIsZero;
    // A lane (gs_renderer_set_frames_in_flight) draws into a target of its owner's context: everything up to here ran beside whatever that context's stream
    // holds (the previous frame's blend and resolve, the host's own work on the target); the blend -- the first kernel to touch the target or its depth
    // attachment -- waits for it, and the stream waits for the blend.
    const bool foreignTarget = rt->ctx != r->ctx;
    if (foreignTarget) {
        // the target's last use (target_touched); everything the context's stream holds only if the host may have put work of its own on the memory there
        // (it asked for the device pointers, or lent a depth buffer it fills itself)
        const bool borrowedDepth = rt->sceneDepth && rt->sceneDepth != rt->sceneDepthOwned;
        if (rt->exposed || borrowedDepth) {
            GS_HIP(hipEventRecord(r->evTargetFree, rt->ctx->stream));
            GS_HIP(hipStreamWaitEvent(st, r->evTargetFree, 0));
        } else if (rt->lastUseValid) GS_HIP(hipStreamWaitEvent(st, rt->evLastUse, 0));
    }
    if (rt->sceneDepth) {
        gsm::FrameConsts fc;
        flatten_params(p, fc);
        hipLaunchKernelGGL(splat_depth_kernel, dim3(div_up(r->n, 256)), dim3(256), 0, st, r->asset->view, fc, (const uint32_t*)r->visMask, r->recW);
    }
#define GS_LAUNCH_BLEND_S(M, D, WL, HL) hipLaunchKernelGGL((blend_kernel<M, D, WL, HL>), dim3(numTiles), dim3(64u << (WL + HL - 6)), 0, st, r->pairVals, tileStart, tileEnd, \
                                             tileOrder, ds.costWrite, r->recs, rt->rgba16f, rc, dstIsZero, r->recW, rt->sceneDepth, \
                                             ds.binCtl, ds.pairSortError, r->hostReportDev)
#define GS_LAUNCH_BLEND(M, D) do { if (twl == 4u) GS_LAUNCH_BLEND_S(M, D, 4, 4); else if (thl == 4u) GS_LAUNCH_BLEND_S(M, D, 5, 4); else GS_LAUNCH_BLEND_S(M, D, 5, 5); } while (0)
    if (rt->sceneDepth) { if (r->blendMode == 0) GS_LAUNCH_BLEND(0, true); else GS_LAUNCH_BLEND(1, true); }
    else { if (r->blendMode == 0) GS_LAUNCH_BLEND(0, false); else GS_LAUNCH_BLEND(1, false); }
#undef GS_LAUNCH_BLEND
#undef GS_LAUNCH_BLEND_S
    prof_record(r, 6);
    GS_HIP(hipGetLastError());
    if (foreignTarget) {
        GS_HIP(hipEventRecord(r->evBlendDone, st));
        GS_HIP(hipStreamWaitEvent(rt->ctx->stream, r->evBlendDone, 0));
    }
    GS_TRY(target_touched(rt, st));
    r->frameInFlight = true;
    prof_end_frame(r);
    return GS_OK;
}

// RenderMode.DebugBoxes (chunks = false: one box per splat, through order[]) / DebugChunkBounds (chunks = true: one per chunk)
int32_t enqueue_debug_boxes(gs_renderer* r, const gs_frame_params* p, gs_target* rt, bool chunks) {
    gs_context* ctx = r->ctx;
    hipStream_t st = ctx->stream;
    const gsm::AssetView& a = r->asset->view;
    const uint32_t count = chunks ? a.chunkCount : r->n;
    if (chunks && count == 0) return GS_OK;                      // m_GpuChunksValid == false: instanceCount = 0 (GaussianSplatRenderer.cs:161-162)
    if (chunks && (uint64_t)count * 256u < (uint64_t)r->n) return fail(GS_ERR_INVALID_ASSET, "chunk blob smaller than the splat count");
    if (!r->boxRecs) GS_HIP(hipMalloc((void**)&r->boxRecs, (size_t)r->n * sizeof(gsm::BoxRec) + 64));
    if (chunks && !r->chunkOrder) {
        GS_HIP(hipMalloc((void**)&r->chunkOrder, ((size_t)count + 16) * 4));
        GS_TRY(enqueue_set_indices(ctx, r->chunkOrder, count));
    }
    if (!chunks) GS_TRY(join_sort(r));
    gsm::FrameConsts fc;
    flatten_params(p, fc);
    gsm::RayConsts ray;
    gsm::RayConstsFromFrame(ray, p->matrix_vp, p->proj_m00, p->proj_m11, p->cam_pos_world[0], p->cam_pos_world[1], p->cam_pos_world[2], (float)rt->width, (float)rt->height);
    prof_record(r, 7);
    if (chunks) hipLaunchKernelGGL(box_setup_kernel<true>, dim3(div_up(count, 256)), dim3(256), 0, st, a, fc, ray, count, r->boxRecs, r->rects, r->visMask, wave_flags_of(r->visMask, r->n));
    else hipLaunchKernelGGL(box_setup_kernel<false>, dim3(div_up(count, 256)), dim3(256), 0, st, a, fc, ray, count, r->boxRecs, r->rects, r->visMask, wave_flags_of(r->visMask, r->n));
    prof_record(r, 8);
    r->viewValid = false;                                        // rects / visibility bits now describe the boxes: a splat draw needs calc_view again
    DrawSetup ds;
    if (!chunks && vis_active(r)) {
        // GS_SORT_VISIBLE: the boxes are drawn through the order buffer too -- the same visible-only sort, over the boxes' visibility bits
        GS_TRY(enqueue_visible_sort(r));
        r->visOrderValid = false;                                // (of the boxes, not of a calc_view)
        GS_TRY(bin_and_sort(r, p, rt, r->visIdx, count, ds, true, 4u, 4u, vis_control(r)));
        r->visDrawn = true;
    } else {
        GS_TRY(bin_and_sort(r, p, rt, chunks ? r->chunkOrder : r->order, count, ds, true, 4u, 4u));   // the box blend writes no report: tile_order_kernel does; 16x16 tiles
        r->visDrawn = false;
    }
#define GS_LAUNCH_BOX(M, D) hipLaunchKernelGGL((blend_box_kernel<M, D>), dim3(ds.numTiles), dim3(256), 0, st, r->pairVals, ds.tileStart, ds.tileEnd, ds.tileOrder, \
                                           ds.costWrite, r->boxRecs, rt->rgba16f, ds.rc, ray, ds.dstIsZero, rt->sceneDepth)
    if (rt->sceneDepth) { if (r->blendMode == 0) GS_LAUNCH_BOX(0, true); else GS_LAUNCH_BOX(1, true); }
    else { if (r->blendMode == 0) GS_LAUNCH_BOX(0, false); else GS_LAUNCH_BOX(1, false); }
#undef GS_LAUNCH_BOX
    prof_record(r, 6);
    GS_HIP(hipGetLastError());
    GS_TRY(target_touched(rt, st));
    r->frameInFlight = true;
    prof_end_frame(r);
    return GS_OK;
}

int32_t enqueue_debug_points(gs_renderer* r, const gs_frame_params* p, gs_target* rt) {
    gs_context* ctx = r->ctx;
    hipStream_t st = ctx->stream;
    GS_TRY(flush_clear(rt));                                    // the squares only touch the pixels they cover
    const uint32_t numPix = rt->width * rt->height;
    if (!rt->zbuf) {
        GS_HIP(hipMalloc((void**)&rt->zbuf, (size_t)numPix * 8));
        GS_HIP(hipMemsetAsync(rt->zbuf, 0xff, (size_t)numPix * 8, st));      // afterwards the resolve kernel resets what it consumes
    }
    gsm::FrameConsts c;
    flatten_params(p, c);
    prof_record(r, 3);
    hipLaunchKernelGGL(debug_points_kernel, dim3(div_up(r->n, 256)), dim3(256), 0, st, r->asset->view, c, 0.5f * r->pointDisplaySize, rt->width, rt->height,
                       rt->sceneDepth, rt->zbuf);
    prof_record(r, 5);
    hipLaunchKernelGGL(debug_points_resolve_kernel, dim3(div_up(numPix, 256)), dim3(256), 0, st, r->asset->view, numPix, r->renderMode == GS_RENDER_DEBUG_POINT_INDICES ? 1 : 0,
                       rt->zbuf, rt->rgba16f);
    prof_record(r, 6);
    GS_HIP(hipGetLastError());
    GS_TRY(target_touched(rt, st));
    prof_end_frame(r);
    return GS_OK;
}

int32_t flush_clear(gs_target* t) {
    if (!t->clearPending) return GS_OK;
    GS_HIP(hipSetDevice(t->ctx->device));
    GS_HIP(hipMemsetAsync(t->rgba16f, 0, (size_t)t->width * t->height * 8, t->ctx->stream));
    t->clearPending = false;
    return target_touched(t, t->ctx->stream);
}

int32_t target_touched(gs_target* t, hipStream_t st) {
    if (t->ctx->children.empty()) { t->lastUseValid = false; return GS_OK; }      // no lanes: the context's stream orders everything by itself
    if (!t->evLastUse) GS_HIP(hipEventCreateWithFlags(&t->evLastUse, hipEventDisableTiming));
    GS_HIP(hipEventRecord(t->evLastUse, st));
    t->lastUseValid = true;
    return GS_OK;
}

int32_t enqueue_resolve(gs_target* t, const float bg[4], bool want8) {
    const uint32_t numPix = t->width * t->height;
    if (!t->resolved) {
        GS_HIP(hipMalloc((void**)&t->resolved, (size_t)numPix * 16));
        GS_HIP(hipMalloc((void**)&t->resolved8, (size_t)numPix * 4));
    }
    hipLaunchKernelGGL(resolve_kernel, dim3(div_up(numPix, 256)), dim3(256), 0, t->ctx->stream, t->rgba16f, numPix, bg[0], bg[1], bg[2], bg[3],
                       t->resolved, want8 ? t->resolved8 : (uint8_t*)nullptr);      // the sRGB 8-bit image (3 powf per pixel) only when asked for
    GS_HIP(hipGetLastError());
    return target_touched(t, t->ctx->stream);
}

} // namespace gs

#ifdef GS_BLEND_TL
extern "C" int32_t gs_debug_blend_timeline(void* out, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_blendTl), bytes) == hipSuccess ? 0 : -1;
}
#endif
#ifdef GS_BLEND_STATS
extern "C" int32_t gs_debug_blend_stats(uint64_t* out8, int32_t reset) {
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gs::g_blendStats), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out8[i] = h[i];
    if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(gs::g_blendStats), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
#endif
