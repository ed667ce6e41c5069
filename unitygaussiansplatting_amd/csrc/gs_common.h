// gs_common.h -- host-side object layouts and helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/gsplat_c.h"
#include "gs_device_math.h"

// Issue priority of the frame's latency-bound kernels (the sort passes, the key / binning / fix-up kernels, resolve): with frames in flight they share their SIMDs
// with another frame's blend, whose heaviest tiles raise their own priority (gs_raster.hip); a chain kernel's few instructions between two waits then queue behind
// the blend's.  -DGS_CHAIN_PRIO=n / -DGS_VIEW_PRIO=n: experiment builds (scripts/build_variants.py).
#ifdef GS_CHAIN_PRIO
#define GS_CHAIN_PRIORITY() __builtin_amdgcn_s_setprio(GS_CHAIN_PRIO)
#else
#define GS_CHAIN_PRIORITY() do { } while (0)
#endif
#ifdef GS_VIEW_PRIO
#define GS_VIEW_PRIORITY() __builtin_amdgcn_s_setprio(GS_VIEW_PRIO)
#else
#define GS_VIEW_PRIORITY() do { } while (0)
#endif

namespace gs {

// thread-local error detail -------------------------------------------------------------------------
void set_error_detail(const char* fmt, ...);
int32_t fail(int32_t code, const char* what);
int32_t fail_hip(hipError_t e, const char* what, const char* file, int line);

#define GS_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return gs::fail_hip(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define GS_TRY(call)                    \
    do {                                \
        int32_t _r = (call);            \
        if (_r != GS_OK) return _r;     \
    } while (0)

// (the compositor's tile is 16x16, 32x16 or 32x32 pixels, picked per draw: pick_tile_shape in gs_raster.hip)
constexpr int kBinThreads = 256;
constexpr int kBinItems = 8;              // sorted positions per thread: 2048 per partition (~3 rounds of partitions over the persistent grid balance better
                                          // than 1.5; 12 / 16 per thread: -5 % at C2 / C4, +6 % at C3 -- r04 calls 7, 8)
constexpr int kBinPart = kBinThreads * kBinItems;
constexpr uint32_t kBinTicketClasses = 16;
constexpr int kEvPerFrame = 28;           // hipEvents per profiled frame: 0..13 stage brackets; 14..21 / 22..27 start + stop of each depth / pair Onesweep launch

// 32-byte per-splat record consumed by the blend kernel (written by calc_view, splat-index order)
struct alignas(16) SplatRec {
    float cx, cy;           // centre, pixels (y down)
    float a1x, a1y;         // axis1, pixels
    float a2x, a2y;         // axis2, pixels
    uint32_t color0;        // f16 r << 16 | f16 g
    uint32_t color1;        // f16 b << 16 | f16 a
};
static_assert(sizeof(SplatRec) == 32, "SplatRec is 32 B");

// everything calc_view writes (all in splat-index order)
struct ViewOutputs {
    gsm::ViewData* view;            // N x 40 B SplatViewData (FULL launches only)
    SplatRec* recs;                 // N x 32 B blend records (visible splats only)
    uint2* rects;                   // N x 8 B tile rectangles
    unsigned long long* visMask;    // 1 bit per splat
};
// 1 byte per 64 splats (a wave of calc_view) behind the visibility bits: nonzero = the wave holds a visible splat -- what bin_emit tests
// first.  It lives in visMask's allocation at an offset that follows from N, so that calc_view_kernel -- whose SGPRs are full of frame
// constants: one more pointer argument spills a dozen of them to VGPR lanes, +4 % -- needs no argument for it.
__host__ __device__ inline size_t vis_mask_words(uint32_t n) { return (((size_t)n + 63) / 64 + 7) & ~(size_t)7; }
__host__ __device__ inline uint8_t* wave_flags_of(unsigned long long* visMask, uint32_t n) { return (uint8_t*)(visMask + vis_mask_words(n)); }
__host__ __device__ inline size_t vis_alloc_bytes(uint32_t n) { return vis_mask_words(n) * 8 + ((size_t)n + 63) / 64 + 64; }

// Onesweep look-back state for one sort (shared by all passes: every pass uses a fresh epoch)
struct SortState {
    uint32_t* altKeys = nullptr;
    uint32_t* altVals = nullptr;
    uint32_t* status = nullptr;             // maxParts x 256 words {epoch:18 | count:14}: a partition's digit counts
    unsigned long long* groupAgg = nullptr; // 4 passes x maxGroups x 256 words {members:24 | sum:40}, zeroed per sort
    unsigned long long* groupIncl = nullptr;// maxGroups x 256 words {epoch:32 | inclusive prefix:32} through the end of a group
    uint32_t maxGroups = 0;
    uint32_t maxCount = 0;
    uint32_t maxParts = 0;
    uint32_t partMin = 8192;                // keys in the smallest partition a pass may use (gs_sort.hip: shape A, or shape C for a depth sort)
    uint32_t histCopies = 1;                // copies of SortControl::hist the producer of the current histograms spread its flushes over (hist_copies()); the passes sum them
    uint32_t epoch = 0;                     // last epoch used on `status` / `groupIncl` (18 bits, never 0)
};

// The digit histograms of a sort are accumulated by hundreds of workgroups flushing their LDS counts with global atomics; atomics on
// the same 128-byte line queue behind each other at the memory side, and 1024 bins are only 32 lines (visible_keys_kernel: 749 workgroups x
// 512 atomics = 12,000 per line, most of its 26 us).  So there are kHistReplicas copies, 4 KB apart; workgroup b adds to copy b % kHistReplicas
// and the one reader (every Onesweep workgroup, once per pass) sums the copies.
constexpr int kHistReplicas = 8;
constexpr int kHistStride = 4 * 256;      // words between two copies
// how many of the copies a sort's producer spreads its flushes over (and its Onesweep passes sum): a property of the producer's grid
uint32_t hist_copies(int producerBlocks);
// small per-sort control block (zeroed by one memset before each sort)
struct SortControl {
    uint32_t hist[kHistReplicas * kHistStride];   // digit histograms [copy][pass][digit]: raw counts
    uint32_t tickets[4][16 * 32]; // partition tickets: per pass, 16 counters (ticket classes) in separate 128-B lines
    uint32_t error;               // != 0: bounded spin expired
    uint32_t pad[3];
};

// What the host wants to know about a finished draw: stored by the blend's first workgroup (or tile_order_kernel) straight into mapped pinned host memory
struct FrameReport {
    unsigned long long pairCount;
    uint32_t binError, pairSortError, visible, tileShape;   // tileShape: log2 tile width | log2 tile height << 8 of the draw that reported
    uint32_t tieLongRuns, tieLongest;                       // GS_SORT_VISIBLE draws: VisControl::tieLongRuns / tieLongest of the draw's visible sort (else 0)
};

// ---- GS_SORT_VISIBLE (gs_vissort.hip): the depth sort of the VISIBLE splats only --------------------------------------------------
// The reference's order buffer after sorts M_1 .. M_k of a base order B is sorted lexicographically by (key under M_k, ..., key under M_1,
// rank in B).  The mode keeps B in gs_renderer::order (CSSetIndices' identity at first) and the rows of the sorts made SINCE B in
// gs_renderer::visHist, most recent first.  When the history is full the base is CONSOLIDATED: one full sort of B by the most recent row +
// the chain fix-up over all N give the reference's whole buffer, which becomes the new B (history: that one row) -- so the chain is exact at
// any length and its walk is bounded.
constexpr int kVisHistory = 128;           // sort-matrix rows (row 2) kept since the base order, most recent first; kernel argument: 2 KB
constexpr uint32_t kVisMaxBlocks = 1024;   // workgroups of visible_keys_kernel (one status word each)
constexpr uint32_t kVisTieWaveMax = 64;    // longest run of equal keys a wave ranks by counting; longer runs: the workgroup's sorting network
// small per-sort control block, two copies used alternately: each visible_keys launch zeroes the other one for the next sort
struct VisControl {
    uint32_t count;                // V: visible splats compacted this frame (the depth sort's device-side key count)
    uint32_t tieLongRuns;          // statistic: runs of more than kVisTieWaveMax equal keys the fix-up ordered (workgroup path)
    uint32_t tieLongest;           // statistic: the longest of them
    uint32_t error;                // bounded spin expired
    uint32_t pad[28];
    // visible count of block b, + 1 (0 = not published yet), one word per 64 bytes: block b polls the words of ALL blocks before it with
    // agent-scope loads, and loads of one 128-byte line queue behind each other at the memory side (749 blocks: 280,000 loads on 24 lines
    // when the words were packed)
    uint32_t status[kVisMaxBlocks * 16];
};
constexpr uint32_t kVisStatusStride = 16;
struct TieHistory { float row[kVisHistory][4]; uint32_t depth; };   // row[0] = the matrix the keys were made with; depth >= 1
// what ends the chain of a tied pair that no kept row separates: the base order
enum TieBase { TB_INDEX = 0,      // base = identity: the splat index
               TB_RANK = 1,       // base = gs_renderer::order: rank[splat] (its inverse, gs_renderer::visBaseRank)
               TB_POSITION = 2 }; // the input is a stable sort OF the base: the position inside the run (consolidation over all N)

// The two words every workgroup hits with an atomic (ticket, visible) sit in their own 128-B lines: same-address
// atomics serialise in one L2 channel (~11 ns each), so they must not also queue behind each other.
struct BinControl {
    unsigned long long pairCount; // P: total pairs this frame (may exceed capacity => overflow)
    uint32_t pairCountClamped;    // min(P, capacity), what the pair sort / ranges / blend see
    uint32_t error;
    uint32_t tieLongRuns, tieLongest;  // copied from the visible sort's VisControl by vis_count (GS_SORT_VISIBLE draws), for the report
    uint32_t pad0[26];
    uint32_t visible;
    uint32_t pad1[31];
    uint32_t tickets[16 * 32];    // kBinTicketClasses partition-ticket counters, one per 128-B line
};

} // namespace gs

struct gs_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // second in-order queue: the depth sort of a frame (SortPoints) has no data dependence on CalcViewData, so it runs
    // here, forked from / joined to `stream` with events, and the two latency-bound kernels share the GPU
    hipStream_t aux = nullptr;
    bool overlap = false;
    // other kernels that spin on their own workgroups may share this GPU (another process running this library, another context of this process sorting
    // at the same time): the depth sort's gather pass then hands its partitions out in dependency order instead of in XCD blocks (gs_sort.hip)
    // -1 automatic (shared iff the process holds more than one context on the device), 0 / 1 pinned by the host
    int sharedGpu = -1;
    bool counted = false;                 // this context is in the live count of its device
    int cuCount = 0;
    hipDeviceProp_t props;
    // contexts the library made itself for renderers of THIS context (gs_renderer_set_frames_in_flight's lanes): gs_context_synchronize waits for them too
    std::vector<gs_context*> children;
    bool internalLane = false;              // one of those: not in the live count of the device; its own (rare) full sorts -- consolidations -- always take the shared-GPU form
    bool sortBesideSiblings = false;        // set around a consolidation of a renderer that has lanes: they consolidate at the same frame, on their own streams
};
bool gs_shared_gpu(const gs_context* ctx);      // may another kernel that waits on its own workgroups run beside this context's? (gs_api.hip)

struct gs_asset {
    gs_context* ctx = nullptr;
    gsm::AssetView view{};
    void* blobs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t sizes[5] = {0, 0, 0, 0, 0};
    bool owned = true;
};

struct gs_sorter {
    gs_context* ctx = nullptr;
    gs::SortState st;
    gs::SortControl* control = nullptr;     // device
    uint32_t* tmpKeys = nullptr;            // for sort_host
    uint32_t* tmpVals = nullptr;
};

struct gs_target {
    gs_context* ctx = nullptr;
    uint32_t width = 0, height = 0;
    uint16_t* rgba16f = nullptr;            // W*H*4 halfs
    bool clearPending = false;              // gs_target_clear was called and nothing has touched the target since: the next
                                            // draw writes every pixel itself (no 8 B/px memset); any other reader clears first
    // optional depth attachment (the camera's depth buffer the reference draws the splats against, GaussianSplatRenderer.cs:195):
    // W*H view depths of the opaque scene; a fragment survives iff the splat's clip.w <= depth
    const float* sceneDepth = nullptr;      // device
    float* sceneDepthOwned = nullptr;       // the copy made of a host buffer (sceneDepth points at it), or null
    unsigned long long* zbuf = nullptr;     // W*H x u64 {view depth bits, ~splat index}: the depth buffer of the debug point modes (all ones = empty)
    float* resolved = nullptr;              // W*H*4 floats, lazily allocated
    uint8_t* resolved8 = nullptr;
    // optional timing of gs_target_resolve: a ring of event pairs on the context's stream
    static constexpr int kResolveRing = 64;
    hipEvent_t* rev = nullptr;              // 2 x kResolveRing events, or null (profiling off)
    bool profiling = false;
    int revCount = 0;                       // resolves recorded since the last read (may exceed the ring: the oldest are overwritten)
    // Lanes (gs_renderer_set_frames_in_flight) draw into the target from streams of their own: a lane's blend waits for the target's LAST USE -- the resolve
    // of the frame drawn into it before, a clear, another draw -- not for everything the context's stream holds (the other frames' blends into OTHER targets:
    // a host that alternates two targets lets consecutive blends overlap).  Recorded only while the context has lanes (gs::target_touched).
    // A host that draws every frame into the SAME target would still queue every blend behind the previous frame's composite (which reads the pixels the blend is
    // about to overwrite).  So while the context has lanes the target holds TWO pixel buffers and gs_target_clear -- after which nothing of the old content can be
    // seen -- moves on to the other one: frame k + 1 is blended into one buffer while frame k's is being resolved from the other.  rgba16f is the current one.
    hipEvent_t evLastUse = nullptr;         // of the current buffer (swapped with evLastUseAlt by the flip)
    bool lastUseValid = false;
    uint16_t* rgba16fAlt = nullptr;         // the other buffer, allocated at the first flip
    hipEvent_t evLastUseAlt = nullptr;
    bool lastUseValidAlt = false;
    bool exposed = false;                   // gs_target_device_ptr handed the memory out: the host's own work on the context's stream may touch it (and the pointer stays put)
};

struct gs_renderer {
    gs_context* ctx = nullptr;
    gs_asset* asset = nullptr;
    uint32_t n = 0;
    // reference buffers (GaussianSplatRenderer.cs:407,431-432)
    gsm::ViewData* view = nullptr;          // m_GpuView
    uint32_t* keyBySplat = nullptr;         // N x u32: the frame's sort key of every splat, in splat-index order
    uint32_t* distances = nullptr;          // m_GpuSortDistances
    uint32_t* order = nullptr;              // m_GpuSortKeys (_OrderBuffer)
    gs::SortState depthSort;
    gs::SortControl* depthControl = nullptr;   // two blocks, used alternately: each sort zeroes the other one for the next
    int depthControlIdx = 0;                   // the block the last / current sort uses
    // compositor buffers
    gs::SplatRec* recs = nullptr;           // N x 32 B, indexed by splat (written by calc_view)
    gsm::BoxRec* boxRecs = nullptr;         // N x 64 B, debug box modes only (allocated on first use)
    uint32_t* chunkOrder = nullptr;         // identity order of the chunks (DebugChunkBounds draws them in index order)
    float* recW = nullptr;                  // N x 4 B: clip.w of the visible splats, filled by the draw only when the target has a depth attachment
    uint2* rects = nullptr;                 // N x 8 B: x = x0 | y0 << 16, y = (x1 + 1) | (y1 + 1) << 16, pixels (0 = culled): gsm::PackPixelRect
    unsigned long long* visMask = nullptr;  // ceil(N/64) x 8 B: bit s = splat s reaches at least one tile (written by calc_view); + ceil(N/64) B per-wave flags (wave_flags_of)
    // edit state read by calc_view (m_GpuEditDeleted / m_GpuEditCutouts, GaussianSplatRenderer.cs:266,269)
    uint32_t* deletedBits = nullptr;        // ceil(N/32) words, or null (_SplatBitsValid = 0)
    uint32_t* cutouts = nullptr;            // GS_MAX_CUTOUTS x 17 dwords
    uint32_t cutoutCount = 0;
    uint8_t* cutoutsHost = nullptr;         // pinned shadow of the last uploaded set
    uint32_t cutoutsHostCount = 0;
    hipEvent_t cutoutsCopied = nullptr;
    bool cutoutsCopyPending = false;
    float viewW = 0.f, viewH = 0.f, viewNear = 0.f, viewFar = 0.f;   // what the last calc_view was run with
    bool viewValid = false;
    bool viewMaterialised = false;          // the N x 40 B view buffer holds the last calc_view's records (written on demand)
    bool alwaysWriteView = false;           // gs_renderer_set_view_buffer_mode(1): write it every frame like the reference
    gs_frame_params lastParams;             // of the last gs_renderer_calc_view (for the on-demand FULL launch)
    uint32_t* pairKeys = nullptr;           // tile ids
    uint32_t* pairVals = nullptr;           // sorted positions
    gs::SortState pairSort;
    uint64_t pairCapacity = 0;
    // per-frame zero arena: [BinControl | SortControl(pair) | binStatus | tileStart | tileEnd]
    uint8_t* frameArena = nullptr;          // two copies, used alternately: each draw zeroes the other one for the next
    int arenaIdx = 0;
    size_t frameArenaBytes = 0;             // of one copy
    size_t offBinStatus = 0, offBinGroupAgg = 0, offBinGroupBase = 0, offTileStart = 0, offTileEnd = 0, offPairControl = 0;
    uint32_t arenaTiles = 0;                // tiles the arena was sized for
    uint32_t* tileCost = nullptr;           // 2 x arenaTiles x u32: batches each tile walked -- a draw writes copy costIdx and reads (for scheduling) the other
    int costIdx = 0;
    uint32_t costTiles[2] = {0, 0};         // tile count of the draw that wrote each copy (0 = none): a schedule can be made from it for the same count only
    uint32_t costShape[2] = {0, 0};         // ... and the same tile shape (log2 w | log2 h << 8)
    uint32_t tileOverrideWL = 0, tileOverrideHL = 0;   // gs_renderer_set_tile_shape: log2 tile width / height, 0 = automatic
    uint32_t lastTileWL = 0, lastTileHL = 0;           // of the last draw (0 x 0: nothing drawn yet)
    bool adaptTall = false;                            // automatic shape: 32x32 instead of 32x16 (large splats; adapt_tile_shape)
    uint32_t* tileOrderBuf = nullptr;       // arenaTiles x u32: the blend's tile schedule of the draw in flight
    uint32_t binParts = 0;
    int blendMode = 0;
    int renderMode = 0;                     // gs_render_mode (GaussianSplatRenderer.RenderMode, :126-131)
    float pointDisplaySize = 3.0f;          // m_PointDisplaySize
    // profiling: a ring of per-frame hipEvent sets (slot advances at the end of gs_renderer_draw)
    bool profiling = false;
    bool kernelTiming = false;              // gs_renderer_set_kernel_timing: Onesweep launches carry their own start / stop events
    hipEvent_t* ev = nullptr;               // profCapacity x kEvPerFrame
    uint8_t* evValid = nullptr;
    int profCapacity = 0, profCur = 0, profCompleted = 0;
    // host copy of last frame's control (pinned), read lazily
    gs::FrameReport* hostReport = nullptr;  // pinned + mapped: written by the last small kernel of a draw (no copy launch)
    gs::FrameReport* hostReportDev = nullptr;   // its device-side address
    hipEvent_t evSortDone = nullptr;        // aux -> main join (timing disabled)
    hipEvent_t evOrderFree = nullptr;       // main -> aux fork: the last operation of the main queue that reads or writes order[]
    bool distancesStale = false;      // the last depth pass skipped the sorted-key write: gs_renderer_download_distances gathers them
    bool sortPending = false;               // a sort on ctx->aux has not been joined into ctx->stream yet
    uint32_t lastTilesX = 0, lastTilesY = 0, lastPairPasses = 0, lastDepthPasses = 4;
    bool frameInFlight = false;
    float resolveMs = 0.f;
    // a truncated draw, latched when the next draw notices it (maybe_grow_pairs) and handed out once by gs_renderer_poll_pairs: the
    // report itself is overwritten and the capacity grown before a pipelined host gets to poll
    unsigned long long truncPairs = 0, truncCapacity = 0;
    // ---- GS_SORT_VISIBLE (gs_renderer_set_sort_mode; gs_vissort.hip) ----
    int sortMode = 0;                       // gs_sort_mode
    // order[] is the BASE of the visible-only sort: the reference's order buffer as of the last full sort / consolidation / upload / reset;
    // the reference's buffer NOW = order[] stably sorted by visHist, oldest first
    bool visBaseIdentity = true;            // order[] is CSSetIndices' identity (rank[s] = s: no rank array needed)
    bool visRankValid = false;              // visBaseRank holds the inverse of order[]
    uint32_t* visBaseRank = nullptr;        // N x u32, allocated when first needed
    bool visOrderValid = false;             // visIdx holds the sorted visible order of the last calc_view under the current history
    bool visDrawn = false;                  // the draw in flight was binned from visIdx
    float visHist[gs::kVisHistory][4];      // sort-matrix rows (m[8..11]) since the base, most recent first, no row twice
    int visHistDepth = 0;
    int visHistLimit = gs::kVisHistory;     // rows kept before the base is consolidated (gs_renderer_set_sort_history_limit; GSPLAT_VIS_HISTORY)
    unsigned long long visConsolidations = 0;
    uint32_t* visKeys = nullptr;            // N x u32 each, allocated on first use: compacted (key, splat index) of the visible splats, sorted in place
    uint32_t* visIdx = nullptr;
    uint32_t* visRectX = nullptr;           // N x u32 each: the pixel rectangle (rects[visIdx[i]].x / .y) by SORTED position (vis_offsets_kernel's gather)
    uint32_t* visRectY = nullptr;
    uint32_t* visPairOffset = nullptr;      // N x u32: first (tile, splat) pair slot of every sorted position (vis_offsets_kernel)
    uint32_t* visChunkStart = nullptr;      // per chunk of 1024 pair slots: the sorted position its first slot belongs to; sized by the pair capacity
    uint32_t visChunkCap = 0;
    gs::VisControl* visControl = nullptr;   // two blocks, used alternately
    int visControlIdx = 0;
    // ---- frames in flight inside the library (gs_renderer_set_frames_in_flight; gs_api.hip) ----
    // lanes: renderers on contexts (= streams) of their own over this renderer's asset, owned by it; while GS_SORT_VISIBLE draws splats every gs_renderer_calc_view
    // moves on to the next lane and the frame's kernels run there -- one frame's latency-bound chain under another's blend.  Empty: one frame at a time, on ctx.
    std::vector<gs_renderer*> lanes;
    int laneCur = -1;                       // the lane of the frame in progress (-1: none yet)
    gs_renderer* laneOf = nullptr;          // a lane's owner
    hipEvent_t evTargetFree = nullptr;      // a lane drawing into its owner's target: target's stream -> lane (before the blend) ...
    hipEvent_t evBlendDone = nullptr;       // ... and lane -> target's stream (after it)
};

namespace gs {
void prof_record(gs_renderer* r, int k, hipStream_t st = nullptr);   // gs_api.hip: record event k of the current profiling slot (on st, default ctx->stream)
int32_t target_touched(gs_target* t, hipStream_t st);   // gs_raster.hip: the last operation on the target's memory has just been enqueued on st
int32_t join_sort(gs_renderer* r);          // make ctx->stream wait for a sort still running on ctx->aux
int32_t mark_order_use(gs_renderer* r);     // the main queue has just been given work that reads / writes order[]: the next sort waits for it
void prof_end_frame(gs_renderer* r);
// sort entry points (gs_sort.hip)
int32_t sort_state_create(gs_context* ctx, SortState& st, uint32_t maxCount, bool smallPartitions = false);
void sort_state_destroy(SortState& st);
// keys of all splats in index order (CSCalcDistances' arithmetic) + the four digit histograms; the gather through the
// previous order is done by the first sort pass (enqueue_sort_passes with gatherKeys)
int32_t enqueue_sort_keys(gs_context* ctx, hipStream_t st, const gsm::AssetView& a, const float* matSort, uint32_t* keyBySplat,
                          SortControl* control, SortControl* nextControl, uint32_t n, SortState& sort);
uint32_t sort_group_words(const SortState& st, uint32_t nUpper, int passes);   // 8-byte words of SortState::groupAgg a sort of nUpper keys uses (to be zeroed before the passes)
int32_t enqueue_histogram(gs_context* ctx, hipStream_t st, const uint32_t* keys, uint32_t n, const uint32_t* nPtr, int passes, uint32_t lastMask,
                          SortControl* control, SortState& sort);
// `passes` Onesweep passes.  The raw digit histograms must already be in control->hist and sort.groupAgg zeroed.  Result ends in (keys, vals) when
// passes is even, otherwise it is copied back.
// profR/evFirst: optional hipEvent slots (evFirst = just before the first Onesweep launch, evFirst + 1 = after the last)
// bitsPerPass: digit width (6..8; the histograms in control->hist must have been taken with the same width).
// gatherKeys != null (8-bit passes, an even number of them): the first pass reads its keys as gatherKeys[vals[i]].
int32_t enqueue_sort_passes(gs_context* ctx, hipStream_t stream, SortState& st, SortControl* control, uint32_t* keys, uint32_t* vals,
                            uint32_t nUpper, const uint32_t* nPtr, int passes, uint32_t lastMask = 255u,
                            gs_renderer* profR = nullptr, int evFirst = -1, int bitsPerPass = 8, const uint32_t* gatherKeys = nullptr, bool skipLastKeys = false,
                            uint32_t expected = 0);   // expected: the key count to tune the pass shape for when nUpper is only a bound (0 = nUpper)
int32_t enqueue_gather_keys(gs_context* ctx, const uint32_t* keyBySplat, const uint32_t* order, uint32_t* out, uint32_t n);
constexpr uint32_t kSortMaxCount = 1u << 30;   // 32-bit byte offsets inside the sort kernels
int32_t enqueue_set_indices(gs_context* ctx, uint32_t* order, uint32_t n);
// visible-only depth sort (gs_vissort.hip)
inline bool vis_active(const gs_renderer* r) { return r->sortMode == GS_SORT_VISIBLE; }
int32_t vis_alloc(gs_renderer* r);                                   // visKeys / visIdx / visControl, on first use
void vis_free(gs_renderer* r);
int32_t vis_push_matrix(gs_renderer* r, const float* matrixSort);    // gs_renderer_sort in visible mode: history bookkeeping (+ a consolidation when the history is full)
// order[] := the reference's whole order buffer now (one full sort of the base by the most recent row + the chain fix-up over N); the history shrinks
// to that row.  Also what hands the buffer to GS_SORT_FULL / gs_renderer_download_order.  (gs_api.hip)
int32_t vis_consolidate(gs_renderer* r);
// the chain fix-up over a stable sort of the base (keys = sorted keys, idx = the order), rows 1.. of the history
int32_t enqueue_tie_fix_full(gs_renderer* r, const uint32_t* keys, uint32_t* idx);
// visMask (calc_view's or box_setup's visibility bits) -> r->visIdx = the visible items in the reference's depth order, count in visControl
int32_t enqueue_visible_sort(gs_renderer* r);
inline const VisControl* vis_control(const gs_renderer* r) { return r->visControl + r->visControlIdx; }
// view (gs_view.hip)
int32_t enqueue_calc_view(gs_context* ctx, const gsm::AssetView& a, const gs_frame_params* p, const gsm::EditView& e, const ViewOutputs& out, bool full);
ViewOutputs view_outputs(gs_renderer* r);
void flatten_params(const gs_frame_params* p, gsm::FrameConsts& c);
// raster (gs_raster.hip)
int32_t renderer_alloc_raster(gs_renderer* r);
void renderer_free_raster(gs_renderer* r);
int32_t enqueue_draw(gs_renderer* r, const gs_frame_params* p, gs_target* rt);
void auto_tile_shape(uint32_t width, uint32_t height, uint32_t& wl, uint32_t& hl);                       // the automatic tile shape for a target size
void pick_tile_shape(const gs_renderer* r, uint32_t width, uint32_t height, uint32_t& wl, uint32_t& hl);  // ... or the renderer's override
int32_t enqueue_debug_points(gs_renderer* r, const gs_frame_params* p, gs_target* rt);   // RenderMode.DebugPoints / DebugPointIndices
int32_t enqueue_debug_boxes(gs_renderer* r, const gs_frame_params* p, gs_target* rt, bool chunks);   // RenderMode.DebugBoxes / DebugChunkBounds
int32_t enqueue_resolve(gs_target* t, const float bg[4], bool want8);
int32_t flush_clear(gs_target* t);          // perform a pending gs_target_clear now
} // namespace gs
