// gs_api.hip -- the C-ABI of include/gsplat_c.h: object lifetime, argument validation, stream sequencing.
// Mirrors what GaussianSplatRenderer.cs / GpuSorting.cs do on Unity's main thread (buffer creation :373-445,
// dispatch order :108-211,579-639, disposal :527-577); every kernel lives in gs_sort/gs_view/gs_raster.hip.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <new>
#include <utility>

#include "gs_common.h"

namespace gs {

static thread_local char g_err[512] = "";

void set_error_detail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int32_t fail(int32_t code, const char* what) {
    set_error_detail("%s", what);
    return code;
}
int32_t fail_hip(hipError_t e, const char* what, const char* file, int line) {
    set_error_detail("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return e == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP;
}

void prof_record(gs_renderer* r, int k, hipStream_t st) {
    if (!r->profiling || !r->ev) return;
    const int idx = r->profCur * kEvPerFrame + k;
    if (hipEventRecord(r->ev[idx], st ? st : r->ctx->stream) == hipSuccess) r->evValid[idx] = 1;
}
int32_t join_sort(gs_renderer* r) {
    if (!r->sortPending) return GS_OK;
    GS_HIP(hipStreamWaitEvent(r->ctx->stream, r->evSortDone, 0));
    r->sortPending = false;
    return GS_OK;
}
int32_t mark_order_use(gs_renderer* r) {
    if (r->ctx->overlap) GS_HIP(hipEventRecord(r->evOrderFree, r->ctx->stream));
    return GS_OK;
}
void prof_end_frame(gs_renderer* r) {
    if (!r->profiling || !r->ev) return;
    r->profCompleted++;
    r->profCur = (r->profCur + 1) % r->profCapacity;
    memset(r->evValid + (size_t)r->profCur * kEvPerFrame, 0, kEvPerFrame);
}

} // namespace gs

using namespace gs;

static int32_t bind_device(gs_context* ctx) {
    GS_HIP(hipSetDevice(ctx->device));
    return GS_OK;
}

// live contexts of this process per device: two of them may sort at the same time (gs_shared_gpu)
static constexpr int kMaxTrackedDevices = 64;
static std::atomic<int> g_liveContexts[kMaxTrackedDevices];

bool gs_shared_gpu(const gs_context* ctx) {
    if (ctx->internalLane || ctx->sortBesideSiblings) return true;
    if (ctx->sharedGpu >= 0) return ctx->sharedGpu != 0;
    return g_liveContexts[ctx->device % kMaxTrackedDevices].load(std::memory_order_relaxed) > 1;
}

extern "C" {

int32_t gs_abi_version(void) { return GS_ABI_VERSION; }

const char* gs_error_string(int32_t err) {
    switch (err) {
        case GS_OK: return "ok";
        case GS_ERR_INVALID_ARGUMENT: return "invalid argument";
        case GS_ERR_HIP: return "HIP runtime error";
        case GS_ERR_UNSUPPORTED_FORMAT: return "unsupported format";
        case GS_ERR_OUT_OF_MEMORY: return "out of device memory";
        case GS_ERR_INVALID_ASSET: return "invalid asset (blob sizes do not match splat count / formats)";
        case GS_ERR_PAIR_OVERFLOW: return "tile-pair buffer overflow (buffer was grown; render the frame again)";
        case GS_ERR_SORT_TIMEOUT: return "sort look-back timed out";
        case GS_ERR_NO_DEVICE: return "no HIP device";
        case GS_ERR_COMM: return "RCCL communication error";
        default: return "unknown error";
    }
}
const char* gs_last_error_string(void) { return g_err; }

// ---- context ---------------------------------------------------------------------------------------------
int32_t gs_context_create(int32_t device, void* hip_stream, gs_context** out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(GS_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(GS_ERR_INVALID_ARGUMENT, "device index out of range");
    gs_context* ctx = new (std::nothrow) gs_context();
    if (!ctx) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return fail(GS_ERR_HIP, "hipSetDevice"); }
    if (hipGetDeviceProperties(&ctx->props, device) != hipSuccess) { delete ctx; return fail(GS_ERR_HIP, "hipGetDeviceProperties"); }
    ctx->cuCount = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->ownStream = false; }
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return fail(GS_ERR_HIP, "hipStreamCreate"); }
        ctx->ownStream = true;
    }
    if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess) { gs_context_destroy(ctx); return fail(GS_ERR_HIP, "hipStreamCreate (aux)"); }
    // Measured on MI355X: running the depth sort on the second queue does not shorten the frame -- neither beside
    // calc_view only (round 1: 0.995 ms overlapped vs 0.962 serial at C2) nor in the pipelined form, beside the previous
    // frame's pair sort / blend / resolve and this frame's calc_view (round 2: 0.679 vs 0.663 ms at C2, 0.981 vs 0.956 at C3;
    // smaller persistent sort grids and a lower / higher queue priority move it by < 1 %, profiles/r02_variants.txt).  Every
    // kernel of the frame already fills the wave slots of the chip, so two queues share them instead of adding to them.
    // The default is therefore off; GSPLAT_OVERLAP=1 or gs_context_set_overlap(ctx, 1) turns it on.
    const char* ov = getenv("GSPLAT_OVERLAP");
    ctx->overlap = ov && ov[0] == '1';
    // -1 = automatic: shared as soon as this process holds a second context on the device (gs_shared_gpu); GSPLAT_SHARED_GPU=1 / 0 or
    // gs_context_set_shared_gpu pin it (another PROCESS on the GPU is something only the host knows)
    const char* sh = getenv("GSPLAT_SHARED_GPU");
    ctx->sharedGpu = (sh && (sh[0] == '0' || sh[0] == '1')) ? (sh[0] - '0') : -1;
    g_liveContexts[device % kMaxTrackedDevices].fetch_add(1, std::memory_order_relaxed);
    ctx->counted = true;
    *out = ctx;
    return GS_OK;
}

int32_t gs_context_destroy(gs_context* ctx) {
    if (!ctx) return GS_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->counted) g_liveContexts[ctx->device % kMaxTrackedDevices].fetch_sub(1, std::memory_order_relaxed);
    delete ctx;
    return GS_OK;
}

int32_t gs_context_synchronize(gs_context* ctx) {
    if (!ctx) return fail(GS_ERR_INVALID_ARGUMENT, "ctx is null");
    for (gs_context* c : ctx->children) GS_TRY(gs_context_synchronize(c));      // lanes of this context's renderers (their frames may not have been joined by a draw)
    GS_TRY(bind_device(ctx));
    GS_HIP(hipStreamSynchronize(ctx->aux));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    return GS_OK;
}

int32_t gs_context_set_overlap(gs_context* ctx, int32_t enabled) {
    if (!ctx) return fail(GS_ERR_INVALID_ARGUMENT, "ctx is null");
    GS_TRY(bind_device(ctx));
    GS_HIP(hipStreamSynchronize(ctx->aux));
    GS_HIP(hipStreamSynchronize(ctx->stream));                  // no order[] use of the main queue is outstanding when the mode changes
    ctx->overlap = enabled != 0;
    return GS_OK;
}

int32_t gs_context_set_shared_gpu(gs_context* ctx, int32_t shared) {
    if (!ctx) return fail(GS_ERR_INVALID_ARGUMENT, "ctx is null");
    ctx->sharedGpu = shared < 0 ? -1 : (shared != 0);
    return GS_OK;
}

int32_t gs_context_device_info(gs_context* ctx, char* name_out, size_t name_cap, int32_t* cu_count, uint64_t* hbm_bytes) {
    if (!ctx) return fail(GS_ERR_INVALID_ARGUMENT, "ctx is null");
    if (name_out && name_cap) { snprintf(name_out, name_cap, "%s (%s)", ctx->props.name, ctx->props.gcnArchName); }
    if (cu_count) *cu_count = ctx->cuCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->props.totalGlobalMem;
    return GS_OK;
}

// ---- asset -----------------------------------------------------------------------------------------------
static uint64_t sh_item_size(uint32_t f) { return f == 0 ? 192 : (f == 2 ? 60 : (f == 3 ? 32 : 96)); }
static uint64_t sh_count(uint32_t f, uint64_t n) {
    switch (f) { case 4: return 65536; case 5: return 32768; case 6: return 16384; case 7: return 8192; case 8: return 4096; default: return n; }
}

int32_t gs_asset_create(gs_context* ctx, const gs_asset_desc* d, gs_asset** out) {
    if (!ctx || !d || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (d->splat_count == 0) return fail(GS_ERR_INVALID_ASSET, "splat_count is 0");
    if (d->pos_format > 3 || d->scale_format > 3 || d->color_format > 3 || d->sh_format > 8) return fail(GS_ERR_INVALID_ARGUMENT, "format enum out of range");
    if (!d->pos_data || !d->other_data || !d->color_data || !d->sh_data) return fail(GS_ERR_INVALID_ASSET, "a required blob is null");
    const uint64_t n = d->splat_count;
    const uint64_t vs[4] = {12, 6, 4, 2};
    const uint64_t cs[4] = {16, 8, 4, 1};               // bytes per texel; BC7: 16-byte blocks of 4x4 texels
    const uint64_t otherStride = 4 + vs[d->scale_format] + (d->sh_format > 3 ? 2 : 0);
    const uint64_t texH = ((n + 2047) / 2048 + 15) / 16 * 16;
    const uint64_t need[5] = { n * vs[d->pos_format], n * otherStride, 2048 * texH * cs[d->color_format],
                               sh_count(d->sh_format, n) * sh_item_size(d->sh_format), 0 };
    const uint64_t have[5] = { d->pos_size, d->other_size, d->color_size, d->sh_size, d->chunk_size };
    const void* src[5] = { d->pos_data, d->other_data, d->color_data, d->sh_data, d->chunk_data };
    for (int k = 0; k < 4; ++k)
        if (have[k] < need[k]) { set_error_detail("blob %d too small: %llu < %llu", k, (unsigned long long)have[k], (unsigned long long)need[k]); return GS_ERR_INVALID_ASSET; }
    if (d->memory_kind == 1) {
        // borrowed device blobs: the kernels use 16-byte vector loads on the blobs (SH staging, chunk bounds) and aligned dword
        // loads everywhere else, and the 2-byte-aligned dword stitching of LoadUInt / LoadUShort may touch the dword after
        // the last record (owned uploads are padded by 16 bytes)
        for (int k = 0; k < 5; ++k)
            if (src[k] && (((uintptr_t)src[k]) & 15u)) return fail(GS_ERR_INVALID_ARGUMENT, "borrowed blobs must be 16-byte aligned");
        for (int k : {0, 1, 3})
            if (have[k] < need[k] + 4) { set_error_detail("borrowed blob %d needs 4 readable bytes after its last record (declare size >= %llu)", k, (unsigned long long)(need[k] + 4)); return GS_ERR_INVALID_ASSET; }
    }
    uint32_t chunkCount = 0;
    if (d->chunk_data && d->chunk_size) {
        chunkCount = (uint32_t)(d->chunk_size / 64);
        if ((uint64_t)chunkCount < (n + 255) / 256) return fail(GS_ERR_INVALID_ASSET, "chunk blob too small");
    }
    GS_TRY(bind_device(ctx));
    gs_asset* a = new (std::nothrow) gs_asset();
    if (!a) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    a->ctx = ctx;
    a->owned = d->memory_kind == 0;
    for (int k = 0; k < 5; ++k) {
        a->sizes[k] = have[k];
        if (!src[k] || have[k] == 0) { a->blobs[k] = nullptr; a->sizes[k] = 0; continue; }
        if (a->owned) {
            // +16 B: the 2-byte-aligned dword stitching of LoadUInt may touch the dword after the last record
            hipError_t e = hipMalloc(&a->blobs[k], have[k] + 16);
            if (e == hipSuccess) e = hipMemsetAsync((uint8_t*)a->blobs[k] + have[k], 0, 16, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(a->blobs[k], src[k], have[k], hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { gs_asset_destroy(a); return fail_hip(e, "asset upload", __FILE__, __LINE__); }
        } else {
            a->blobs[k] = const_cast<void*>(src[k]);
        }
    }
    if (a->owned) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) { gs_asset_destroy(a); return fail_hip(e, "asset upload sync", __FILE__, __LINE__); } }
    a->view.pos = (const uint8_t*)a->blobs[0]; a->view.other = (const uint8_t*)a->blobs[1]; a->view.color = (const uint8_t*)a->blobs[2];
    a->view.sh = (const uint8_t*)a->blobs[3]; a->view.chunk = (const uint8_t*)a->blobs[4];
    a->view.n = d->splat_count; a->view.posFmt = d->pos_format; a->view.scaleFmt = d->scale_format;
    a->view.colorFmt = d->color_format; a->view.shFmt = d->sh_format; a->view.chunkCount = chunkCount;
    *out = a;
    return GS_OK;
}

int32_t gs_asset_destroy(gs_asset* a) {
    if (!a) return GS_OK;
    (void)hipSetDevice(a->ctx->device);
    if (a->owned) for (int k = 0; k < 5; ++k) if (a->blobs[k]) (void)hipFree(a->blobs[k]);
    delete a;
    return GS_OK;
}

int32_t gs_asset_splat_count(const gs_asset* a, uint32_t* out) {
    if (!a || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = a->view.n;
    return GS_OK;
}

int32_t gs_asset_info(const gs_asset* a, uint32_t out[6]) {
    if (!a || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    out[0] = a->view.n; out[1] = a->view.posFmt; out[2] = a->view.scaleFmt; out[3] = a->view.colorFmt; out[4] = a->view.shFmt; out[5] = a->view.chunkCount;
    return GS_OK;
}

int32_t gs_asset_device_blobs(const gs_asset* a, void* ptrs[5], uint64_t sizes[5]) {
    if (!a || !ptrs || !sizes) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    for (int k = 0; k < 5; ++k) { ptrs[k] = a->blobs[k]; sizes[k] = a->sizes[k]; }
    return GS_OK;
}

// ---- renderer --------------------------------------------------------------------------------------------
// ---- frames in flight inside the library: lanes ---------------------------------------------------------------
// (gs_renderer_set_frames_in_flight, gsplat_c.h.)  A lane is an ordinary renderer on a context of its own; the owner tells every lane every sort matrix
// (bookkeeping in GS_SORT_VISIBLE) and every setting, deals the frames round-robin at gs_renderer_calc_view and answers the whole-buffer questions
// (gs_renderer_download_order, _distances, _sort_history) from its own copy of the bookkeeping.
static inline bool lanes_on(const gs_renderer* r) { return !r->lanes.empty() && r->sortMode == GS_SORT_VISIBLE && r->renderMode == GS_RENDER_SPLATS; }
static inline gs_renderer* lane_cur(gs_renderer* r) { return lanes_on(r) && r->laneCur >= 0 ? r->lanes[(size_t)r->laneCur] : r; }

static void lanes_destroy(gs_renderer* r) {
    for (gs_renderer* L : r->lanes) {
        gs_context* c = L->ctx;
        (void)gs_renderer_destroy(L);
        for (size_t k = 0; k < r->ctx->children.size(); ++k)
            if (r->ctx->children[k] == c) { r->ctx->children.erase(r->ctx->children.begin() + (long)k); break; }
        (void)gs_context_destroy(c);
    }
    r->lanes.clear();
    r->laneCur = -1;
}

// The lanes take over the owner's order: its buffer as the reference holds it now (the recorded sorts carried out) becomes every lane's base, the
// owner's remaining history (its head row at most) theirs.  Rare: when lanes are made, when the mode is switched to GS_SORT_VISIBLE, after reset / upload.
static int32_t lanes_resync(gs_renderer* r) {
    if (r->lanes.empty()) return GS_OK;
    GS_TRY(bind_device(r->ctx));
    if (r->sortMode != GS_SORT_VISIBLE) return GS_OK;            // GS_SORT_FULL runs on the owner alone: the lanes idle until the mode comes back (and are resynchronised then)
    GS_TRY(vis_consolidate(r));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    for (gs_renderer* L : r->lanes) {
        GS_TRY(gs_renderer_set_sort_mode(L, GS_SORT_VISIBLE));
        GS_HIP(hipStreamSynchronize(L->ctx->stream));
        if (r->visBaseIdentity) GS_TRY(enqueue_set_indices(L->ctx, L->order, L->n));
        else GS_HIP(hipMemcpyAsync(L->order, r->order, (size_t)r->n * 4, hipMemcpyDeviceToDevice, L->ctx->stream));
        GS_HIP(hipStreamSynchronize(L->ctx->stream));
        L->visBaseIdentity = r->visBaseIdentity; L->visRankValid = false; L->visOrderValid = false;
        L->visHistDepth = r->visHistDepth;
        memcpy(L->visHist, r->visHist, sizeof(r->visHist));
        L->visHistLimit = r->visHistLimit;
        L->distancesStale = false;
    }
    return GS_OK;
}

int32_t gs_renderer_set_frames_in_flight(gs_renderer* r, int32_t frames) {
    if (!r || frames < 1 || frames > GS_MAX_FRAMES_IN_FLIGHT) return fail(GS_ERR_INVALID_ARGUMENT, "frames in flight must be in [1, GS_MAX_FRAMES_IN_FLIGHT]");
    if (r->laneOf) return fail(GS_ERR_INVALID_ARGUMENT, "a lane has no lanes of its own");
    const size_t want = frames > 1 ? (size_t)frames : 0u;       // one frame at a time: the renderer's own context, no lanes
    if (r->lanes.size() == want) return GS_OK;
    GS_TRY(gs_context_synchronize(r->ctx));
    lanes_destroy(r);
    for (size_t k = 0; k < want; ++k) {
        gs_context* c = nullptr;
        gs_renderer* L = nullptr;
        int32_t rc = gs_context_create(r->ctx->device, nullptr, &c);
        if (rc == GS_OK) {                                       // not a context of the host's: see gs_shared_gpu
            c->internalLane = true;
            if (c->counted) { g_liveContexts[c->device % kMaxTrackedDevices].fetch_sub(1, std::memory_order_relaxed); c->counted = false; }
        }
        if (rc == GS_OK) rc = gs_renderer_create(c, r->asset, &L);
        if (rc == GS_OK) {
            L->laneOf = r;
            hipError_t e = hipEventCreateWithFlags(&L->evTargetFree, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&L->evBlendDone, hipEventDisableTiming);
            if (e != hipSuccess) rc = fail_hip(e, "create lane events", __FILE__, __LINE__);
        }
        // the owner's settings as they are now; later changes are forwarded by the setters themselves
        if (rc == GS_OK) {
            L->blendMode = r->blendMode; L->alwaysWriteView = r->alwaysWriteView; L->kernelTiming = r->kernelTiming;
            L->tileOverrideWL = r->tileOverrideWL; L->tileOverrideHL = r->tileOverrideHL; L->adaptTall = r->adaptTall;
            L->visHistLimit = r->visHistLimit;
            if (r->pairCapacity > L->pairCapacity) rc = gs_renderer_reserve_pairs(L, r->pairCapacity);
        }
        if (rc == GS_OK && r->cutoutCount) rc = gs_renderer_set_cutouts(L, (const gs_cutout*)r->cutoutsHost, r->cutoutCount);
        if (rc == GS_OK && r->deletedBits) {
            const size_t words = ((size_t)r->n + 31) / 32;
            uint32_t* h = new (std::nothrow) uint32_t[words];
            if (!h) rc = fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
            else {
                rc = bind_device(r->ctx);
                if (rc == GS_OK && hipMemcpy(h, r->deletedBits, words * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GS_ERR_HIP, "copy deleted bits");
                if (rc == GS_OK) rc = gs_renderer_set_deleted_bits(L, h, words);
                delete[] h;
            }
        }
        if (rc != GS_OK) {
            if (L) (void)gs_renderer_destroy(L);
            if (c) (void)gs_context_destroy(c);
            lanes_destroy(r);
            return rc;
        }
        r->lanes.push_back(L);
        r->ctx->children.push_back(c);
    }
    return lanes_resync(r);
}

int32_t gs_renderer_frames_in_flight(const gs_renderer* r, int32_t* frames, int32_t* active) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    if (frames) *frames = r->lanes.empty() ? 1 : (int32_t)r->lanes.size();
    if (active) *active = lanes_on(r) ? 1 : 0;
    return GS_OK;
}

int32_t gs_renderer_create(gs_context* ctx, gs_asset* asset, gs_renderer** out) {
    if (!ctx || !asset || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    // the asset's blobs are immutable: renderers of OTHER contexts on the same GPU may read them too (several frames / views in flight on
    // several streams, one copy of the asset); the caller keeps the asset alive, as for any renderer
    if (asset->ctx != ctx && asset->ctx->device != ctx->device) return fail(GS_ERR_INVALID_ARGUMENT, "asset lives on another GPU (gs_asset_replicate / gs_asset_broadcast)");
    GS_TRY(bind_device(ctx));
    gs_renderer* r = new (std::nothrow) gs_renderer();
    if (!r) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    r->ctx = ctx; r->asset = asset; r->n = asset->view.n;
    int32_t rc = GS_OK;
    auto chk = [&](hipError_t e, const char* what) { if (rc == GS_OK && e != hipSuccess) rc = fail_hip(e, what, __FILE__, __LINE__); };
    chk(hipMalloc((void**)&r->view, (size_t)r->n * sizeof(gsm::ViewData) + 64), "alloc view");
    chk(hipMalloc((void**)&r->keyBySplat, ((size_t)r->n + 16) * 4), "alloc sort keys");
    chk(hipMalloc((void**)&r->distances, ((size_t)r->n + 16) * 4), "alloc distances");
    chk(hipMalloc((void**)&r->order, ((size_t)r->n + 16) * 4), "alloc order");
    chk(hipMalloc((void**)&r->depthControl, 2 * sizeof(SortControl)), "alloc sort control");
    if (rc == GS_OK) chk(hipMemsetAsync(r->depthControl, 0, 2 * sizeof(SortControl), ctx->stream), "clear sort control");
    chk(hipEventCreateWithFlags(&r->evOrderFree, hipEventDisableTiming), "create event");
    chk(hipEventCreateWithFlags(&r->evSortDone, hipEventDisableTiming), "create event");
    if (rc == GS_OK) rc = sort_state_create(ctx, r->depthSort, r->n, true);        // (small partitions: the visible-only sort)
    if (rc == GS_OK) rc = renderer_alloc_raster(r);
    if (rc == GS_OK) rc = enqueue_set_indices(ctx, r->order, r->n);
    if (rc == GS_OK) chk(hipMemsetAsync(r->view, 0, (size_t)r->n * sizeof(gsm::ViewData), ctx->stream), "clear view");
    if (rc == GS_OK) rc = mark_order_use(r);                    // the first sort (second queue) waits for these initialisations
    if (rc != GS_OK) { gs_renderer_destroy(r); return rc; }
    *out = r;
    return GS_OK;
}

int32_t gs_renderer_destroy(gs_renderer* r) {
    if (!r) return GS_OK;
    lanes_destroy(r);
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->aux);
    (void)hipStreamSynchronize(r->ctx->stream);
    if (r->evTargetFree) (void)hipEventDestroy(r->evTargetFree);
    if (r->evBlendDone) (void)hipEventDestroy(r->evBlendDone);
    if (r->evOrderFree) (void)hipEventDestroy(r->evOrderFree);
    if (r->evSortDone) (void)hipEventDestroy(r->evSortDone);
    if (r->view) (void)hipFree(r->view);
    if (r->keyBySplat) (void)hipFree(r->keyBySplat);
    if (r->distances) (void)hipFree(r->distances);
    if (r->order) (void)hipFree(r->order);
    if (r->depthControl) (void)hipFree(r->depthControl);
    if (r->deletedBits) (void)hipFree(r->deletedBits);
    if (r->cutouts) (void)hipFree(r->cutouts);
    if (r->cutoutsHost) (void)hipHostFree(r->cutoutsHost);
    if (r->cutoutsCopied) (void)hipEventDestroy(r->cutoutsCopied);
    sort_state_destroy(r->depthSort);
    renderer_free_raster(r);
    vis_free(r);
    if (r->ev) { for (int k = 0; k < r->profCapacity * kEvPerFrame; ++k) (void)hipEventDestroy(r->ev[k]); delete[] r->ev; delete[] r->evValid; }
    delete r;
    return GS_OK;
}

// m_GpuSortDistances holds the sorted keys after a sort (GpuSorting.cs:142-198).  The last depth pass skips that write (nothing on the
// frame's path reads them) and gs_renderer_download_distances rebuilds them as keyBySplat[order[i]] -- which is only right while order[]
// is the sort's own output: anything that is about to overwrite order[] materialises them first.
static int32_t materialise_distances(gs_renderer* r) {
    if (!r->distancesStale) return GS_OK;
    GS_TRY(enqueue_gather_keys(r->ctx, r->keyBySplat, r->order, r->distances, r->n));
    r->distancesStale = false;
    return GS_OK;
}

static int32_t enqueue_full_sort(gs_renderer* r, const float m[16], bool consolidating);

// order[] is about to stop being what the visible-only mode calls its base (reset, upload, a sort in GS_SORT_FULL)
static void vis_base_changed(gs_renderer* r, bool identity) {
    r->visBaseIdentity = identity; r->visRankValid = false; r->visHistDepth = 0; r->visOrderValid = false;
}

int32_t gs_renderer_reset_order(gs_renderer* r) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    GS_TRY(bind_device(r->ctx));
    GS_TRY(join_sort(r));
    // m_GpuSortDistances keeps the sorted keys of the last SortPoints across CSSetIndices: in the visible-only mode they only exist once
    // the sorts recorded since the base have been carried out on all N
    if (vis_active(r)) GS_TRY(vis_consolidate(r));
    GS_TRY(materialise_distances(r));
    GS_TRY(enqueue_set_indices(r->ctx, r->order, r->n));
    vis_base_changed(r, true);                                   // CSSetIndices: the order buffer is the identity again and the stable-sort history starts over
    GS_TRY(mark_order_use(r));
    return lanes_resync(r);
}

static void rec_ev(gs_renderer* r, int k) { gs::prof_record(r, k); }

// GS_SORT_VISIBLE keeps the sorts made since its base order[] as a list of matrix rows (gs_vissort.hip).  This carries them out on ALL N
// splats: the reference's order buffer now = the base stably sorted by every row, oldest first = ONE stable sort of the base by the most
// recent row (ties are left in base order) + the chain fix-up over N (every run of equal keys re-ordered by the older rows; where they tie
// too the base order stands).  Exact at any history length, the cost of one reference-shaped sort.  Afterwards order[] is that buffer (the
// new base), distances[] its sorted keys, and the history is its head alone (sorting the new base by it again changes nothing).
// Called when the history is full, when the buffer itself is asked for (gs_renderer_download_order / _distances, reset) and when the
// renderer goes back to GS_SORT_FULL.
extern "C++" int32_t gs::vis_consolidate(gs_renderer* r) {
    if (r->visHistDepth == 0) return GS_OK;                      // no sort since the base: order[] is the reference's buffer already
    float m[16] = { 0.f };
    memcpy(m + 8, r->visHist[0], 16);
    GS_TRY(join_sort(r));
    r->ctx->sortBesideSiblings = !r->lanes.empty();              // (the lanes reach a full history at the same frame: their sorts run beside this one)
    const int32_t rcSort = enqueue_full_sort(r, m, true);
    r->ctx->sortBesideSiblings = false;
    GS_TRY(rcSort);
    GS_TRY(enqueue_tie_fix_full(r, r->distances, r->order));
    GS_TRY(mark_order_use(r));
    r->visHistDepth = 1; r->visBaseIdentity = false; r->visRankValid = false;
    r->visConsolidations++;
    return GS_OK;
}

int32_t gs_renderer_sort(gs_renderer* r, const float m[16]) {
    if (!r || !m) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    GS_TRY(bind_device(r->ctx));
    // GS_SORT_VISIBLE: SortPoints only says which matrix the order is sorted by from now on; the sort itself runs in gs_renderer_draw,
    // over the splats that are drawn (gs_vissort.hip)
    if (vis_active(r)) {
        GS_TRY(vis_push_matrix(r, m));
        for (gs_renderer* L : r->lanes) { GS_TRY(bind_device(L->ctx)); GS_TRY(vis_push_matrix(L, m)); }      // every lane is told every matrix
        return GS_OK;
    }
    vis_base_changed(r, false);                                  // the order buffer now holds this sort
    return enqueue_full_sort(r, m, false);
}

static int32_t enqueue_full_sort(gs_renderer* r, const float m[16], bool consolidating) {
    gs_context* ctx = r->ctx;
    // SortPoints depends on nothing else the frame computes (the C# merely records it before CalcViewData, :120-126), and
    // nothing but the draw's bin_emit reads its result.  With overlap on, the whole sort (keys + the four Onesweep passes)
    // runs on the context's second queue: it starts as soon as the main queue is done with order[] (evOrderFree: the
    // previous draw's bin_emit), i.e. beside the previous frame's pair sort / blend / resolve and this frame's calc_view
    // -- latency-bound kernels with little VALU work next to the VALU-bound ones -- and is joined by the first consumer of
    // order[] (gs_renderer_draw, or any readback).  The second queue is in-order, so consecutive sorts serialise there.
    // (A consolidation of the visible-only mode is not a frame's SortPoints: main queue, sorted keys kept for the fix-up, no stage events.)
    hipStream_t st = ctx->stream;
    const bool aux = ctx->overlap && !consolidating;
    if (aux) {
        st = ctx->aux;
        GS_HIP(hipStreamWaitEvent(st, r->evOrderFree, 0));
    }
    gs_renderer* profR = consolidating ? nullptr : r;
    if (profR) gs::prof_record(r, 0, st);
    r->depthControlIdx ^= 1;
    SortControl* control = r->depthControl + r->depthControlIdx;
    // CSCalcDistances: keys of all splats in index order (+ the digit histograms); the gather through the previous order
    // (_SplatSortKeys, SplatUtilities.compute:76) is the first Onesweep pass's key load
    GS_TRY(enqueue_sort_keys(ctx, st, r->asset->view, m, r->keyBySplat, control, r->depthControl + (r->depthControlIdx ^ 1), r->n, r->depthSort));
    if (profR) gs::prof_record(r, 1, st);
    // (skipLastKeys: the last depth pass writes only the order -- nothing on the frame's path reads the sorted keys; materialise_distances)
    GS_TRY(enqueue_sort_passes(ctx, st, r->depthSort, control, r->distances, r->order, r->n, nullptr, 4, 255u, profR, 10, 8, r->keyBySplat, !consolidating));
    r->distancesStale = !consolidating;
    if (profR) { gs::prof_record(r, 2, st); r->lastDepthPasses = 4; }
    if (aux) {
        GS_HIP(hipEventRecord(r->evSortDone, st));
        r->sortPending = true;
    }
    return GS_OK;
}

int32_t gs_renderer_calc_view(gs_renderer* r, const gs_frame_params* p) {
    if (!r || !p) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (!(p->screen_w >= 1.0f && p->screen_w <= 65535.0f && p->screen_h >= 1.0f && p->screen_h <= 65535.0f))      // (pixel rectangles are packed in 16-bit fields)
        return fail(GS_ERR_INVALID_ARGUMENT, "screen_w / screen_h must be in [1, 65535]");
    if (lanes_on(r)) {                                           // a new frame (or view): the next lane's
        r->laneCur = (r->laneCur + 1) % (int)r->lanes.size();
        return gs_renderer_calc_view(r->lanes[(size_t)r->laneCur], p);
    }
    GS_TRY(bind_device(r->ctx));
    rec_ev(r, 7);
    gsm::EditView e;
    e.deletedBits = r->deletedBits; e.cutouts = r->cutouts; e.cutoutCount = r->cutoutCount;
    GS_TRY(enqueue_calc_view(r->ctx, r->asset->view, p, e, view_outputs(r), r->alwaysWriteView));
    r->viewMaterialised = r->alwaysWriteView;
    r->visOrderValid = false;                                    // the visible set may have changed
    r->lastParams = *p;
    r->viewW = p->screen_w; r->viewH = p->screen_h; r->viewNear = p->near_clip; r->viewFar = p->far_clip; r->viewValid = true;
    rec_ev(r, 8);
    return GS_OK;
}

// Grow the pair buffers before they overflow again.  The report of a draw is stored by the first workgroup of its blend straight
// into mapped pinned host memory (before any tile is blended), so whatever it holds is the most recent draw that got
// that far -- also in a pipelined loop in which the host runs ahead and the stream is never idle.  An overflowing scene is
// therefore noticed within the pipeline depth, not only when the caller polls gs_renderer_frame_stats.
static int32_t maybe_grow_pairs(gs_renderer* r) {
    if (!r->frameInFlight) return GS_OK;
    const unsigned long long seen = *(volatile unsigned long long*)&r->hostReport->pairCount;
    // at the 2^30 ceiling there is nothing to grow: the draw goes ahead truncated (gs_renderer_frame_stats reports the frame), so
    // that a later frame that fits renders normally and rewrites the report
    if (seen > r->pairCapacity) {
        // latch the truncated draw for gs_renderer_poll_pairs: by the time a pipelined host polls, the capacity has grown (below) and the
        // report belongs to a later draw
        if (seen > r->truncPairs) { r->truncPairs = seen; r->truncCapacity = r->pairCapacity; }
        if (r->pairCapacity < kSortMaxCount) return gs_renderer_reserve_pairs(r, seen + seen / 4);
    }
    return GS_OK;
}

int32_t gs_renderer_draw(gs_renderer* r, const gs_frame_params* p, gs_target* rt) {
    if (!r || !p || !rt) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (lanes_on(r)) {
        if (rt->ctx != r->ctx) return fail(GS_ERR_INVALID_ARGUMENT, "target belongs to another context");
        if (r->laneCur < 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_renderer_draw: call gs_renderer_calc_view with the same screen size / clip planes first");
        return gs_renderer_draw(r->lanes[(size_t)r->laneCur], p, rt);
    }
    if (rt->ctx != r->ctx && !(r->laneOf && rt->ctx == r->laneOf->ctx)) return fail(GS_ERR_INVALID_ARGUMENT, "target belongs to another context");
    if ((uint32_t)p->screen_w != rt->width || (uint32_t)p->screen_h != rt->height) return fail(GS_ERR_INVALID_ARGUMENT, "screen_w/h do not match the target");
    GS_TRY(bind_device(r->ctx));
    if (r->renderMode == GS_RENDER_DEBUG_POINTS || r->renderMode == GS_RENDER_DEBUG_POINT_INDICES) return enqueue_debug_points(r, p, rt);
    GS_TRY(maybe_grow_pairs(r));
    if (r->renderMode == GS_RENDER_DEBUG_BOXES) return enqueue_debug_boxes(r, p, rt, false);
    if (r->renderMode == GS_RENDER_DEBUG_CHUNK_BOUNDS) return enqueue_debug_boxes(r, p, rt, true);
    return enqueue_draw(r, p, rt);
}

int32_t gs_renderer_render(gs_renderer* r, const float m[16], const gs_frame_params* p, gs_target* rt, int32_t do_sort) {
    if (!r || !p || !rt) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (do_sort) { if (!m) return fail(GS_ERR_INVALID_ARGUMENT, "matrix_sort is null"); GS_TRY(gs_renderer_sort(r, m)); }
    GS_TRY(gs_renderer_calc_view(r, p));
    GS_TRY(gs_target_clear(rt));
    return gs_renderer_draw(r, p, rt);
}

int32_t gs_renderer_set_cutouts(gs_renderer* r, const gs_cutout* cutouts, uint32_t count) {
    if (!r || (count && !cutouts)) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (count > GS_MAX_CUTOUTS) return fail(GS_ERR_INVALID_ARGUMENT, "more than GS_MAX_CUTOUTS cutouts");
    static_assert(sizeof(gs_cutout) == 17 * 4, "gs_cutout is 17 dwords");
    GS_TRY(bind_device(r->ctx));
    if (count) {
        if (!r->cutouts) {                                   // all three resources or none
            uint32_t* dev = nullptr; uint8_t* host = nullptr; hipEvent_t ev = nullptr;
            hipError_t e = hipMalloc((void**)&dev, (size_t)GS_MAX_CUTOUTS * sizeof(gs_cutout));
            if (e == hipSuccess) e = hipHostMalloc((void**)&host, (size_t)GS_MAX_CUTOUTS * sizeof(gs_cutout), hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) {
                if (dev) (void)hipFree(dev);
                if (host) (void)hipHostFree(host);
                return fail_hip(e, "allocate cutout buffers", __FILE__, __LINE__);
            }
            r->cutouts = dev; r->cutoutsHost = host; r->cutoutsCopied = ev;
            r->cutoutsHostCount = 0;
        }
        // the C# re-uploads the buffer every CalcViewData (UpdateCutoutsBuffer); here an unchanged set costs nothing.
        // A changed set goes through a pinned shadow copy, so the caller's memory is only read during this call and the
        // copy is stream-ordered after the calc_view launches that still read the previous set.
        const size_t bytes = (size_t)count * sizeof(gs_cutout);
        if (count != r->cutoutsHostCount || memcmp(r->cutoutsHost, cutouts, bytes) != 0) {
            if (r->cutoutsCopyPending) GS_HIP(hipEventSynchronize(r->cutoutsCopied));      // the shadow is still being read
            memcpy(r->cutoutsHost, cutouts, bytes);
            r->cutoutsHostCount = count;
            GS_HIP(hipMemcpyAsync(r->cutouts, r->cutoutsHost, bytes, hipMemcpyHostToDevice, r->ctx->stream));
            GS_HIP(hipEventRecord(r->cutoutsCopied, r->ctx->stream));
            r->cutoutsCopyPending = true;
        }
    }
    r->cutoutCount = count;
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_cutouts(L, cutouts, count));
    return GS_OK;
}

int32_t gs_renderer_set_deleted_bits(gs_renderer* r, const uint32_t* words, size_t word_count) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    GS_TRY(bind_device(r->ctx));
    const size_t need = ((size_t)r->n + 31) / 32;
    if (!words) {                                           // _SplatBitsValid = 0
        if (r->deletedBits) { GS_HIP(hipStreamSynchronize(r->ctx->stream)); (void)hipFree(r->deletedBits); r->deletedBits = nullptr; }
        for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_deleted_bits(L, nullptr, 0));
        return GS_OK;
    }
    if (word_count != need) return fail(GS_ERR_INVALID_ARGUMENT, "deleted bits: word_count must be ceil(splat_count / 32)");
    if (!r->deletedBits) GS_HIP(hipMalloc((void**)&r->deletedBits, need * 4));
    // an edit-time operation (EditDeleteSelected): stream-ordered copy, then block so `words` is only read during the call
    GS_HIP(hipMemcpyAsync(r->deletedBits, words, need * 4, hipMemcpyHostToDevice, r->ctx->stream));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_deleted_bits(L, words, word_count));
    return GS_OK;
}

int32_t gs_renderer_set_view_buffer_mode(gs_renderer* r, int32_t every_frame) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    r->alwaysWriteView = every_frame != 0;
    for (gs_renderer* L : r->lanes) L->alwaysWriteView = r->alwaysWriteView;
    return GS_OK;
}

int32_t gs_renderer_set_render_mode(gs_renderer* r, int32_t mode, float point_display_size) {
    if (!r || mode < GS_RENDER_SPLATS || mode > GS_RENDER_DEBUG_CHUNK_BOUNDS) return fail(GS_ERR_INVALID_ARGUMENT, "render mode out of range");
    if (!(point_display_size >= 0.0f) || point_display_size > 4096.0f) return fail(GS_ERR_INVALID_ARGUMENT, "point_display_size out of range");
    r->renderMode = mode;
    r->pointDisplaySize = point_display_size;
    return GS_OK;
}

int32_t gs_renderer_set_blend_mode(gs_renderer* r, int32_t mode) {
    if (!r || (mode != 0 && mode != 1)) return fail(GS_ERR_INVALID_ARGUMENT, "blend mode must be 0 or 1");
    r->blendMode = mode;
    for (gs_renderer* L : r->lanes) L->blendMode = mode;
    return GS_OK;
}

int32_t gs_renderer_set_tile_shape(gs_renderer* r, uint32_t tile_w, uint32_t tile_h) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_tile_shape(L, tile_w, tile_h));
    if (tile_w == 0 && tile_h == 0) { r->tileOverrideWL = r->tileOverrideHL = 0; return GS_OK; }
    if (!((tile_w == 16 && tile_h == 16) || (tile_w == 32 && tile_h == 16) || (tile_w == 32 && tile_h == 32)))
        return fail(GS_ERR_INVALID_ARGUMENT, "tile shape must be 16x16, 32x16, 32x32 or 0x0 (automatic)");
    r->tileOverrideWL = tile_w == 16 ? 4u : 5u; r->tileOverrideHL = tile_h == 16 ? 4u : 5u;
    return GS_OK;
}

int32_t gs_renderer_tile_shape(const gs_renderer* r, uint32_t width, uint32_t height, uint32_t* tile_w, uint32_t* tile_h) {
    if (!tile_w || !tile_h) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    uint32_t wl, hl;
    if (r && lanes_on(r) && r->laneCur >= 0) r = r->lanes[(size_t)r->laneCur];      // (the adaptive choice follows the lane's own draws)
    pick_tile_shape(r, width, height, wl, hl);           // r may be null: the automatic choice
    *tile_w = 1u << wl; *tile_h = 1u << hl;
    return GS_OK;
}

int32_t gs_renderer_set_profiling(gs_renderer* r, int32_t frames) {
    if (!r || frames < 0 || frames > 4096) return fail(GS_ERR_INVALID_ARGUMENT, "frames must be in [0, 4096]");
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_profiling(L, frames));
    GS_TRY(bind_device(r->ctx));
    if (frames > r->profCapacity) {
        GS_HIP(hipStreamSynchronize(r->ctx->stream));
        const size_t cnt = (size_t)frames * kEvPerFrame;
        hipEvent_t* ev = new (std::nothrow) hipEvent_t[cnt]();            // value-initialised: null handles
        uint8_t* valid = new (std::nothrow) uint8_t[cnt]();
        hipError_t e = (ev && valid) ? hipSuccess : hipErrorOutOfMemory;
        size_t made = 0;
        for (; e == hipSuccess && made < cnt; ++made) e = hipEventCreate(&ev[made]);
        if (e != hipSuccess) {                                            // the renderer keeps its previous ring
            for (size_t k = 0; ev && k < made; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
            delete[] ev; delete[] valid;
            return fail_hip(e, "create profiling events", __FILE__, __LINE__);
        }
        if (r->ev) { for (int k = 0; k < r->profCapacity * kEvPerFrame; ++k) (void)hipEventDestroy(r->ev[k]); delete[] r->ev; delete[] r->evValid; }
        r->ev = ev; r->evValid = valid; r->profCapacity = frames;
    }
    r->profiling = frames > 0;
    r->profCur = 0; r->profCompleted = 0;
    if (r->evValid) memset(r->evValid, 0, (size_t)r->profCapacity * kEvPerFrame);
    return GS_OK;
}

int32_t gs_renderer_set_kernel_timing(gs_renderer* r, int32_t enabled) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    r->kernelTiming = enabled != 0;
    for (gs_renderer* L : r->lanes) L->kernelTiming = r->kernelTiming;
    return GS_OK;
}

int32_t gs_renderer_reserve_pairs(gs_renderer* r, uint64_t cap) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_reserve_pairs(L, cap));
    if (cap <= r->pairCapacity) return GS_OK;
    if (cap > kSortMaxCount) cap = kSortMaxCount;          // 32-bit byte offsets inside the sort kernels
    if (cap <= r->pairCapacity) return fail(GS_ERR_PAIR_OVERFLOW, "the frame needs more than 2^30 (tile, splat) pairs");
    GS_TRY(bind_device(r->ctx));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    // new buffers first; the renderer only changes once every allocation has succeeded
    uint32_t *nk = nullptr, *nv = nullptr;
    SortState ns;
    hipError_t e = hipMalloc((void**)&nk, ((size_t)cap + 16) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&nv, ((size_t)cap + 16) * 4);
    int32_t rc = e == hipSuccess ? sort_state_create(r->ctx, ns, (uint32_t)cap) : fail_hip(e, "grow pair buffers", __FILE__, __LINE__);
    if (rc != GS_OK) { if (nk) (void)hipFree(nk); if (nv) (void)hipFree(nv); sort_state_destroy(ns); return rc; }
    if (r->pairKeys) (void)hipFree(r->pairKeys);
    if (r->pairVals) (void)hipFree(r->pairVals);
    sort_state_destroy(r->pairSort);
    r->pairKeys = nk; r->pairVals = nv; r->pairSort = ns; r->pairCapacity = cap;
    // a lane that had to grow: its siblings draw the same scene (a frame repeated after GS_ERR_PAIR_OVERFLOW goes to the next lane)
    if (r->laneOf)
        for (gs_renderer* S : r->laneOf->lanes)
            if (S != r && S->pairCapacity < cap) GS_TRY(gs_renderer_reserve_pairs(S, cap));
    return GS_OK;
}

int32_t gs_renderer_poll_pairs(gs_renderer* r, uint64_t* tile_pairs, uint64_t* pair_capacity) {
    if (!r || !tile_pairs || !pair_capacity) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    r = lane_cur(r);
    if (r->truncPairs) {                                         // a truncated draw the library has already reacted to: handed out once
        *tile_pairs = r->truncPairs; *pair_capacity = r->truncCapacity;
        r->truncPairs = r->truncCapacity = 0;
        return GS_OK;
    }
    *tile_pairs = r->hostReport ? (uint64_t)*(volatile unsigned long long*)&r->hostReport->pairCount : 0u;
    *pair_capacity = r->pairCapacity;
    return GS_OK;
}

static int32_t download(gs_context* ctx, void* dst, const void* src, size_t bytes) {
    GS_TRY(bind_device(ctx));
    GS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    return GS_OK;
}

int32_t gs_renderer_download_order(gs_renderer* r, uint32_t* out, size_t count) {
    if (!r || !out || count > r->n) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    GS_TRY(bind_device(r->ctx));
    GS_TRY(join_sort(r));
    if (vis_active(r)) GS_TRY(vis_consolidate(r));               // the reference's whole buffer: the recorded sorts carried out on all N (one sort + the chain fix-up)
    return download(r->ctx, out, r->order, count * 4);
}
int32_t gs_renderer_download_visible_order(gs_renderer* r, uint32_t* out, size_t capacity, uint32_t* count) {
    if (!r || !count || (capacity && !out)) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    r = lane_cur(r);
    *count = 0;
    if (!vis_active(r)) return fail(GS_ERR_INVALID_ARGUMENT, "the visible-only sort mode is not set (gs_renderer_set_sort_mode)");
    if (!r->viewValid) return fail(GS_ERR_INVALID_ARGUMENT, "gs_renderer_calc_view has not run");
    GS_TRY(bind_device(r->ctx));
    if (!r->visOrderValid) {                                     // (not part of a draw: no stage events)
        const bool prof = r->profiling;
        r->profiling = false;
        const int32_t rc = enqueue_visible_sort(r);
        r->profiling = prof;
        GS_TRY(rc);
    }
    uint32_t v = 0;
    GS_TRY(download(r->ctx, &v, &vis_control(r)->count, 4));
    *count = v;
    const size_t take = v < capacity ? v : capacity;
    if (take) GS_TRY(download(r->ctx, out, r->visIdx, take * 4));
    return GS_OK;
}
int32_t gs_renderer_set_sort_mode(gs_renderer* r, int32_t mode) {
    if (!r || (mode != GS_SORT_FULL && mode != GS_SORT_VISIBLE)) return fail(GS_ERR_INVALID_ARGUMENT, "sort mode must be GS_SORT_FULL or GS_SORT_VISIBLE");
    if (mode == r->sortMode) return GS_OK;
    GS_TRY(bind_device(r->ctx));
    if (mode == GS_SORT_VISIBLE) {
        // whatever order[] holds -- CSSetIndices' identity, full sorts, an uploaded order -- is the base the mode's sorts start from
        GS_TRY(vis_alloc(r));
        GS_TRY(join_sort(r));
        r->sortMode = mode;
        r->visHistDepth = 0; r->visRankValid = false; r->visOrderValid = false;
        return lanes_resync(r);                                  // (lanes: the same base)
    }
    // back to the reference's SortPoints: it continues from the order buffer the reference would hold now
    GS_TRY(vis_consolidate(r));
    r->visHistDepth = 0;
    r->sortMode = mode;
    return lanes_resync(r);
}
int32_t gs_renderer_sort_mode(const gs_renderer* r, int32_t* mode, int32_t* active) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    if (mode) *mode = r->sortMode;
    if (active) *active = vis_active(r) ? 1 : 0;
    return GS_OK;
}
int32_t gs_renderer_set_sort_history_limit(gs_renderer* r, uint32_t rows) {
    if (!r || rows < 2 || rows > (uint32_t)kVisHistory) return fail(GS_ERR_INVALID_ARGUMENT, "sort history limit must be in [2, 128]");
    GS_TRY(bind_device(r->ctx));
    for (gs_renderer* L : r->lanes) GS_TRY(gs_renderer_set_sort_history_limit(L, rows));
    GS_TRY(bind_device(r->ctx));
    if ((uint32_t)r->visHistDepth > rows) GS_TRY(vis_consolidate(r));
    r->visHistLimit = (int)rows;
    return GS_OK;
}
int32_t gs_renderer_sort_history(const gs_renderer* r, uint32_t* rows, uint32_t* limit, uint64_t* consolidations) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    if (rows) *rows = (uint32_t)r->visHistDepth;
    if (limit) *limit = (uint32_t)r->visHistLimit;
    if (consolidations) *consolidations = r->visConsolidations;
    return GS_OK;
}
int32_t gs_renderer_download_distances(gs_renderer* r, uint32_t* out, size_t count) {
    if (!r || !out || count > r->n) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    GS_TRY(bind_device(r->ctx));
    GS_TRY(join_sort(r));
    if (vis_active(r)) GS_TRY(vis_consolidate(r));               // GS_SORT_VISIBLE: the sorted keys of all N exist once the recorded sorts have been carried out
    GS_TRY(materialise_distances(r));      // the sorted keys, materialised on demand
    return download(r->ctx, out, r->distances, count * 4);
}
int32_t gs_renderer_upload_order(gs_renderer* r, const uint32_t* in, size_t count) {
    if (!r || !in || count != r->n) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    GS_TRY(bind_device(r->ctx));
    GS_TRY(join_sort(r));
    if (vis_active(r)) GS_TRY(vis_consolidate(r));               // (distances[]: the last SortPoints' sorted keys, as in reset_order)
    GS_TRY(materialise_distances(r));      // before order[] stops being the sort's output
    GS_HIP(hipMemcpyAsync(r->order, in, count * 4, hipMemcpyHostToDevice, r->ctx->stream));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    vis_base_changed(r, false);                                  // a custom order: the new base of the visible-only mode
    return lanes_resync(r);
}
int32_t gs_renderer_download_view(gs_renderer* r, void* out, size_t bytes) {
    if (!r || !out || bytes > (size_t)r->n * sizeof(gsm::ViewData)) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    r = lane_cur(r);
    if (r->viewValid && !r->viewMaterialised) {
        // m_GpuView is materialised on demand: the per-frame launch skips it (nothing in this renderer reads it); re-run the
        // frame's launch as the reference's full kernel.  rec/rect/visibility are rewritten with identical values.
        GS_TRY(bind_device(r->ctx));
        gsm::EditView e;
        e.deletedBits = r->deletedBits; e.cutouts = r->cutouts; e.cutoutCount = r->cutoutCount;
        GS_TRY(enqueue_calc_view(r->ctx, r->asset->view, &r->lastParams, e, view_outputs(r), true));
        r->viewMaterialised = true;
    }
    return download(r->ctx, out, r->view, bytes);
}

int32_t gs_renderer_download_raster_records(gs_renderer* r, void* recs, uint32_t* rects, uint64_t* vis_mask) {
    if (!r) return fail(GS_ERR_INVALID_ARGUMENT, "renderer is null");
    r = lane_cur(r);
    if (!r->viewValid) return fail(GS_ERR_INVALID_ARGUMENT, "gs_renderer_calc_view has not run");
    GS_TRY(bind_device(r->ctx));
    hipStream_t st = r->ctx->stream;
    if (recs) GS_HIP(hipMemcpyAsync(recs, r->recs, (size_t)r->n * sizeof(SplatRec), hipMemcpyDeviceToHost, st));
    if (rects) GS_HIP(hipMemcpyAsync(rects, r->rects, (size_t)r->n * sizeof(uint2), hipMemcpyDeviceToHost, st));
    if (vis_mask) GS_HIP(hipMemcpyAsync(vis_mask, r->visMask, ((size_t)r->n + 63) / 64 * 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

int32_t gs_renderer_frame_stats(gs_renderer* r, gs_frame_stats* out) {
    if (!r || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    r = lane_cur(r);                                             // the frame in progress (the lane that drew last)
    GS_TRY(bind_device(r->ctx));
    GS_TRY(join_sort(r));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    uint32_t depthErr = 0;
    GS_HIP(hipMemcpy(&depthErr, &r->depthControl[r->depthControlIdx].error, 4, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->pair_capacity = r->pairCapacity;
    out->tiles_x = r->lastTilesX; out->tiles_y = r->lastTilesY;
    out->tile_w = r->lastTileWL ? 1u << r->lastTileWL : 0u; out->tile_h = r->lastTileHL ? 1u << r->lastTileHL : 0u;      // 0 x 0: nothing drawn yet
    if (r->frameInFlight) {
        out->tile_pairs = r->hostReport->pairCount;
        out->visible_splats = r->hostReport->visible;
        out->sort_error = depthErr | r->hostReport->pairSortError | (r->hostReport->binError & 2u);
        if (r->visDrawn) {
            uint32_t visErr = 0;
            GS_HIP(hipMemcpy(&visErr, &vis_control(r)->error, 4, hipMemcpyDeviceToHost));
            out->sort_error |= visErr;
            out->sort_mode = GS_SORT_VISIBLE;
            out->tie_long_runs = r->hostReport->tieLongRuns;
            out->tie_longest_run = r->hostReport->tieLongest;
        }
    } else out->sort_error = depthErr;
    if (out->sort_error) return fail(GS_ERR_SORT_TIMEOUT, "a bounded look-back spin expired");
    if (r->frameInFlight && out->tile_pairs > r->pairCapacity) {
        const unsigned long long want = out->tile_pairs + out->tile_pairs / 4;
        r->frameInFlight = false;                                    // reported once, whatever the growth below does
        GS_TRY(gs_renderer_reserve_pairs(r, want));                  // (at the 2^30 ceiling this is the GS_ERR_PAIR_OVERFLOW itself)
        out->pair_capacity = r->pairCapacity;
        return fail(GS_ERR_PAIR_OVERFLOW, "tile-pair buffer overflowed this frame; it has been grown, render again");
    }
    return GS_OK;
}

int32_t gs_renderer_frame_times(gs_renderer* r, float* out_ms, int32_t capacity, int32_t* count) {
    if (!r || !out_ms || !count || capacity < 0) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    *count = 0;
    r = lane_cur(r);                                             // (with lanes: the ring of the lane that drew last -- every lane times its own frames)
    if (!r->ev) return fail(GS_ERR_INVALID_ARGUMENT, "profiling was never enabled");
    GS_TRY(bind_device(r->ctx));
    GS_HIP(hipStreamSynchronize(r->ctx->aux));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    const int slots = r->profCompleted < r->profCapacity ? r->profCompleted : r->profCapacity;
    for (int sidx = 0; sidx < slots && *count < capacity; ++sidx) {
        const int base = sidx * kEvPerFrame;
        // first event of the frame: key generation if the frame sorted before its calc_view (GS_SORT_FULL), else calc_view (a GS_SORT_VISIBLE
        // frame's sort runs inside the draw) -- decided per slot by the events themselves, so a ring that mixes the modes is timed correctly
        int first = 7;
        float lead = -1.f;
        if (r->evValid[base + 0] && (!r->evValid[base + 7] || (hipEventElapsedTime(&lead, r->ev[base + 0], r->ev[base + 7]) == hipSuccess && lead >= 0.f))) first = 0;
        float ms = 0.f;
        if (r->evValid[base + first] && r->evValid[base + 6] && hipEventElapsedTime(&ms, r->ev[base + first], r->ev[base + 6]) == hipSuccess)
            out_ms[(*count)++] = ms;
    }
    return GS_OK;
}

int32_t gs_renderer_stage_times(gs_renderer* r, gs_stage_times* out) {
    if (!r || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    memset(out, 0, sizeof(*out));
    r = lane_cur(r);
    if (!r->ev) return fail(GS_ERR_INVALID_ARGUMENT, "profiling was never enabled");
    GS_TRY(bind_device(r->ctx));
    GS_HIP(hipStreamSynchronize(r->ctx->aux));
    GS_HIP(hipStreamSynchronize(r->ctx->stream));
    // average every stage over the completed slots of the ring (the slot in progress is included if it holds events)
    const int slots = r->profCompleted < r->profCapacity ? r->profCompleted + 1 : r->profCapacity;
    auto avg = [&](int a, int b) -> float {
        double sum = 0.0; int cnt = 0;
        for (int sidx = 0; sidx < slots; ++sidx) {
            const int base = sidx * kEvPerFrame;
            float ms = 0.f;
            if (r->evValid[base + a] && r->evValid[base + b] && hipEventElapsedTime(&ms, r->ev[base + a], r->ev[base + b]) == hipSuccess) { sum += ms; cnt++; }
        }
        return cnt ? (float)(sum / cnt) : 0.f;
    };
    out->calc_distances_ms = avg(0, 1);
    out->sort_ms = avg(1, 2);
    out->calc_view_ms = avg(7, 8);
    out->bin_ms = avg(3, 4);
    out->pair_sort_ms = avg(4, 5);
    out->blend_ms = avg(5, 6);
    out->resolve_ms = 0.f;
    out->onesweep_depth_ms = avg(10, 11);
    out->onesweep_pairs_ms = avg(12, 13);
    out->onesweep_pair_launches = r->lastPairPasses;
    out->onesweep_depth_launches = r->lastDepthPasses;
    for (int ps = 0; ps < 4; ++ps) out->onesweep_depth_kernel_ms += avg(14 + 2 * ps, 15 + 2 * ps);
    for (int ps = 0; ps < (int)r->lastPairPasses && ps < 3; ++ps) out->onesweep_pairs_kernel_ms += avg(22 + 2 * ps, 23 + 2 * ps);
    out->total_ms = out->calc_distances_ms + out->sort_ms + out->calc_view_ms + out->bin_ms + out->pair_sort_ms + out->blend_ms;
    out->frames = (uint32_t)(r->profCompleted < r->profCapacity ? r->profCompleted : r->profCapacity);
    r->profCur = 0; r->profCompleted = 0;
    memset(r->evValid, 0, (size_t)r->profCapacity * kEvPerFrame);
    return GS_OK;
}

// ---- target ----------------------------------------------------------------------------------------------
int32_t gs_target_create(gs_context* ctx, uint32_t w, uint32_t h, gs_target** out) {
    // (<= 65535: the per-splat pixel rectangle stores x1 + 1 and y1 + 1 in 16-bit fields, gsm::PackPixelRect)
    if (!ctx || !out || w == 0 || h == 0 || w > 65535 || h > 65535) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    GS_TRY(bind_device(ctx));
    gs_target* t = new (std::nothrow) gs_target();
    if (!t) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    t->ctx = ctx; t->width = w; t->height = h;
    hipError_t e = hipMalloc((void**)&t->rgba16f, (size_t)w * h * 8);
    if (e == hipSuccess) e = hipMemsetAsync(t->rgba16f, 0, (size_t)w * h * 8, ctx->stream);
    if (e != hipSuccess) { delete t; return fail_hip(e, "alloc target", __FILE__, __LINE__); }
    *out = t;
    return GS_OK;
}

int32_t gs_target_destroy(gs_target* t) {
    if (!t) return GS_OK;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);        // (a lane's blend into the target is joined into this stream by its draw)
    if (t->rgba16f) (void)hipFree(t->rgba16f);
    if (t->rgba16fAlt) (void)hipFree(t->rgba16fAlt);
    if (t->evLastUseAlt) (void)hipEventDestroy(t->evLastUseAlt);
    if (t->resolved) (void)hipFree(t->resolved);
    if (t->resolved8) (void)hipFree(t->resolved8);
    if (t->sceneDepthOwned) (void)hipFree(t->sceneDepthOwned);
    if (t->zbuf) (void)hipFree(t->zbuf);
    if (t->rev) { for (int k = 0; k < 2 * gs_target::kResolveRing; ++k) if (t->rev[k]) (void)hipEventDestroy(t->rev[k]); delete[] t->rev; }
    if (t->evLastUse) (void)hipEventDestroy(t->evLastUse);
    delete t;
    return GS_OK;
}

int32_t gs_target_clear(gs_target* t) {
    if (!t) return fail(GS_ERR_INVALID_ARGUMENT, "target is null");
    t->clearPending = true;                 // the next draw writes every pixel (blend kernel); anything else clears first
    // lanes on this context: the cleared frame goes into the target's other pixel buffer (gs_common.h), so that its blend does not wait for the composite of the
    // frame before it.  Not once the host holds the device pointer.
    if (!t->ctx->children.empty() && !t->exposed) {
        if (!t->rgba16fAlt) {
            GS_TRY(bind_device(t->ctx));
            GS_HIP(hipMalloc((void**)&t->rgba16fAlt, (size_t)t->width * t->height * 8));
        }
        std::swap(t->rgba16f, t->rgba16fAlt);
        std::swap(t->evLastUse, t->evLastUseAlt);
        std::swap(t->lastUseValid, t->lastUseValidAlt);
    }
    return GS_OK;
}

int32_t gs_target_set_scene_depth(gs_target* t, const float* depth, int32_t memory_kind) {
    if (!t || (memory_kind != 0 && memory_kind != 1)) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    GS_TRY(bind_device(t->ctx));
    if (!depth) { t->sceneDepth = nullptr; return GS_OK; }      // a buffer we own stays allocated for the next host upload
    if (memory_kind == 1) { t->sceneDepth = depth; return GS_OK; }
    const size_t bytes = (size_t)t->width * t->height * sizeof(float);
    if (!t->sceneDepthOwned) GS_HIP(hipMalloc((void**)&t->sceneDepthOwned, bytes));
    GS_HIP(hipMemcpyAsync(t->sceneDepthOwned, depth, bytes, hipMemcpyHostToDevice, t->ctx->stream));
    GS_HIP(hipStreamSynchronize(t->ctx->stream));               // `depth` is only read during the call
    t->sceneDepth = t->sceneDepthOwned;
    return GS_OK;
}

int32_t gs_target_download(gs_target* t, void* out, size_t bytes) {
    if (!t || !out || bytes > (size_t)t->width * t->height * 8) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    GS_TRY(flush_clear(t));
    return download(t->ctx, out, t->rgba16f, bytes);
}

int32_t gs_target_resolve(gs_target* t, const float bg[4], float* out32, uint8_t* out8) {
    if (!t || !bg) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    GS_TRY(bind_device(t->ctx));
    GS_TRY(flush_clear(t));
    const int slot = t->revCount % gs_target::kResolveRing;
    if (t->profiling && t->rev) GS_HIP(hipEventRecord(t->rev[2 * slot], t->ctx->stream));
    GS_TRY(enqueue_resolve(t, bg, out8 != nullptr));
    if (t->profiling && t->rev) { GS_HIP(hipEventRecord(t->rev[2 * slot + 1], t->ctx->stream)); t->revCount++; }
    const size_t px = (size_t)t->width * t->height;
    if (out32) GS_HIP(hipMemcpyAsync(out32, t->resolved, px * 16, hipMemcpyDeviceToHost, t->ctx->stream));
    if (out8) GS_HIP(hipMemcpyAsync(out8, t->resolved8, px * 4, hipMemcpyDeviceToHost, t->ctx->stream));
    if (out32 || out8) GS_HIP(hipStreamSynchronize(t->ctx->stream));
    return GS_OK;
}

int32_t gs_target_set_profiling(gs_target* t, int32_t enabled) {
    if (!t) return fail(GS_ERR_INVALID_ARGUMENT, "target is null");
    GS_TRY(bind_device(t->ctx));
    if (enabled && !t->rev) {
        const int cnt = 2 * gs_target::kResolveRing;
        hipEvent_t* ev = new (std::nothrow) hipEvent_t[cnt]();
        if (!ev) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
        for (int k = 0; k < cnt; ++k)
            if (hipEventCreate(&ev[k]) != hipSuccess) { for (int j = 0; j < k; ++j) (void)hipEventDestroy(ev[j]); delete[] ev; return fail(GS_ERR_HIP, "hipEventCreate"); }
        t->rev = ev;
    }
    t->profiling = enabled != 0;
    t->revCount = 0;
    return GS_OK;
}

int32_t gs_target_resolve_time(gs_target* t, float* mean_ms, int32_t* count) {
    if (!t || !mean_ms) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *mean_ms = 0.f;
    if (count) *count = 0;
    if (!t->rev) return fail(GS_ERR_INVALID_ARGUMENT, "profiling was never enabled on this target");
    GS_TRY(bind_device(t->ctx));
    GS_HIP(hipStreamSynchronize(t->ctx->stream));
    const int n = t->revCount < gs_target::kResolveRing ? t->revCount : gs_target::kResolveRing;
    double sum = 0.0; int got = 0;
    for (int k = 0; k < n; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, t->rev[2 * k], t->rev[2 * k + 1]) == hipSuccess) { sum += ms; got++; } }
    if (got) *mean_ms = (float)(sum / got);
    if (count) *count = got;
    t->revCount = 0;
    return GS_OK;
}

int32_t gs_target_device_ptr(gs_target* t, void** rgba16f_dev, void** resolved_dev) {
    if (!t) return fail(GS_ERR_INVALID_ARGUMENT, "target is null");
    GS_TRY(flush_clear(t));                 // the caller is about to read the memory directly
    t->exposed = true;                      // (lanes: from now on a draw into this target waits for everything the context's stream holds)
    if (rgba16f_dev) *rgba16f_dev = t->rgba16f;
    if (resolved_dev) *resolved_dev = t->resolved;
    return GS_OK;
}

// ---- stand-alone sorter (GpuSorting) ---------------------------------------------------------------------
int32_t gs_sorter_create(gs_context* ctx, uint32_t max_count, gs_sorter** out) {
    if (!ctx || !out || max_count == 0) return fail(GS_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    GS_TRY(bind_device(ctx));
    gs_sorter* s = new (std::nothrow) gs_sorter();
    if (!s) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    s->ctx = ctx;
    int32_t rc = sort_state_create(ctx, s->st, max_count, true);     // (every pass shape available: small counts sort in 4,096-key partitions)
    if (rc == GS_OK && hipMalloc((void**)&s->control, sizeof(SortControl)) != hipSuccess) rc = fail(GS_ERR_OUT_OF_MEMORY, "alloc sort control");
    if (rc != GS_OK) { gs_sorter_destroy(s); return rc; }
    *out = s;
    return GS_OK;
}

int32_t gs_sorter_destroy(gs_sorter* s) {
    if (!s) return GS_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    sort_state_destroy(s->st);
    if (s->control) (void)hipFree(s->control);
    if (s->tmpKeys) (void)hipFree(s->tmpKeys);
    if (s->tmpVals) (void)hipFree(s->tmpVals);
    delete s;
    return GS_OK;
}

int32_t gs_sorter_dispatch(gs_sorter* s, void* keys_dev, void* values_dev, uint32_t count, uint32_t key_bits) {
    if (!s || !keys_dev || !values_dev) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (key_bits < 1 || key_bits > 32) return fail(GS_ERR_INVALID_ARGUMENT, "key_bits must be in [1,32]");
    if (count > s->st.maxCount) return fail(GS_ERR_INVALID_ARGUMENT, "count exceeds the sorter's capacity");
    if (count == 0) return GS_OK;
    GS_TRY(bind_device(s->ctx));
    const int passes = (int)((key_bits + 7) / 8);
    const uint32_t lastBits = key_bits - 8u * (uint32_t)(passes - 1);
    const uint32_t lastMask = (1u << lastBits) - 1u;
    GS_TRY(enqueue_histogram(s->ctx, s->ctx->stream, (const uint32_t*)keys_dev, count, nullptr, passes, lastMask, s->control, s->st));
    return enqueue_sort_passes(s->ctx, s->ctx->stream, s->st, s->control, (uint32_t*)keys_dev, (uint32_t*)values_dev, count, nullptr, passes, lastMask);
}

int32_t gs_sorter_sort_host(gs_sorter* s, uint32_t* keys, uint32_t* values, uint32_t count, uint32_t key_bits) {
    if (!s || !keys || !values) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    if (count > s->st.maxCount) return fail(GS_ERR_INVALID_ARGUMENT, "count exceeds the sorter's capacity");
    if (count == 0) return GS_OK;
    GS_TRY(bind_device(s->ctx));
    if (!s->tmpKeys) {
        GS_HIP(hipMalloc((void**)&s->tmpKeys, (size_t)(s->st.maxCount + 16) * 4));
        GS_HIP(hipMalloc((void**)&s->tmpVals, (size_t)(s->st.maxCount + 16) * 4));
    }
    hipStream_t st = s->ctx->stream;
    GS_HIP(hipMemcpyAsync(s->tmpKeys, keys, (size_t)count * 4, hipMemcpyHostToDevice, st));
    GS_HIP(hipMemcpyAsync(s->tmpVals, values, (size_t)count * 4, hipMemcpyHostToDevice, st));
    GS_TRY(gs_sorter_dispatch(s, s->tmpKeys, s->tmpVals, count, key_bits));
    GS_HIP(hipMemcpyAsync(keys, s->tmpKeys, (size_t)count * 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipMemcpyAsync(values, s->tmpVals, (size_t)count * 4, hipMemcpyDeviceToHost, st));
    uint32_t err = 0;
    GS_HIP(hipMemcpyAsync(&err, &s->control->error, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if (err) return fail(GS_ERR_SORT_TIMEOUT, "a bounded look-back spin expired");
    return GS_OK;
}

} // extern "C"
