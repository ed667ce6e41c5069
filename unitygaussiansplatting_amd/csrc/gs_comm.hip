// gs_comm.hip -- multi-GPU: the one exchange the path has, the load-time broadcast of the asset blobs.
//
// The reference is single-GPU; this layer is new.  The render path shards by VIEW: every GPU holds a replica of the
// immutable GaussianSplatAsset blobs (GaussianSplatAsset.cs:205-229) and runs sort -> view data -> composite on its own
// camera, so there is no per-frame collective.  At load, the rank that has the asset broadcasts the five blobs to the
// others: one ncclBroadcast per blob (296 MB for the bicycle-sized Medium asset -- five large messages over xGMI, not
// thousands of small ones), on the context's stream, through RCCL called DIRECTLY from this library (no torch, no MPI:
// a .NET host binds these entry points with P/Invoke like the rest of the ABI and moves the 128-byte unique id over any
// channel it has).  librccl is loaded on first use (dlopen), so a single-GPU host never pays for it and the library has
// no link-time dependency on it; a copy already loaded by the process (e.g. the one inside PyTorch) is reused.
#include <dlfcn.h>
#include <stdlib.h>

#include <new>

#include <rccl/rccl.h>

#include "gs_common.h"

static_assert(NCCL_UNIQUE_ID_BYTES == GS_COMM_ID_BYTES, "gs_comm id size = ncclUniqueId");

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = { getenv("GSPLAT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (int pass = 0; pass < 2 && !x.handle; ++pass)               // pass 0: a copy the process has loaded already
            for (const char* nm : names) {
                if (!nm || !nm[0]) continue;
                x.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (x.handle) break;
            }
        if (!x.handle) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.Broadcast = (decltype(x.Broadcast))dlsym(x.handle, "ncclBroadcast");
        x.AllReduce = (decltype(x.AllReduce))dlsym(x.handle, "ncclAllReduce");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.Broadcast && x.AllReduce && x.GetErrorString;
        return x;
    }();
    return r;
}

int32_t need_rccl() {
    if (!rccl().ok) return gs::fail(GS_ERR_COMM, "librccl could not be loaded (set GSPLAT_RCCL_LIB to its path)");
    return GS_OK;
}

int32_t fail_nccl(ncclResult_t e, const char* what) {
    gs::set_error_detail("%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    return GS_ERR_COMM;
}

#define GS_NCCL(call)                                         \
    do {                                                      \
        ncclResult_t _e = (call);                             \
        if (_e != ncclSuccess) return fail_nccl(_e, #call);   \
    } while (0)

constexpr uint64_t kHeaderMagic = 0x3154414c50534753ull;        // "GSSPLAT1"
constexpr int kHeaderWords = 16;

// ---- the two halves of a replication, shared by gs_asset_broadcast (blobs moved by ncclBroadcast) and gs_asset_replicate
// (blobs moved by a device copy): what the SENDER says about its asset, and how a RECEIVER turns that into an asset of its own.
void pack_header(const gs_asset* a, uint64_t h[kHeaderWords]) {
    memset(h, 0, kHeaderWords * sizeof(uint64_t));
    const gsm::AssetView& v = a->view;
    h[0] = kHeaderMagic; h[1] = v.n; h[2] = v.posFmt; h[3] = v.scaleFmt; h[4] = v.colorFmt; h[5] = v.shFmt; h[6] = v.chunkCount;
    for (int k = 0; k < 5; ++k) h[7 + k] = a->blobs[k] ? a->sizes[k] : 0;
}

bool header_ok(const uint64_t h[kHeaderWords]) { return h[0] == kHeaderMagic && h[1] != 0; }

// allocates the receiver's padded blobs (nothing is committed to *out unless every allocation succeeded)
int32_t receive_alloc(gs_context* ctx, const uint64_t h[kHeaderWords], gs_asset** out) {
    *out = nullptr;
    gs_asset* a = new (std::nothrow) gs_asset();
    if (!a) return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    a->ctx = ctx; a->owned = true;
    for (int k = 0; k < 5; ++k) {
        a->sizes[k] = h[7 + k];
        if (!a->sizes[k]) continue;
        hipError_t e = hipMalloc(&a->blobs[k], a->sizes[k] + 16);           // + the decoders' tail pad, as gs_asset_create
        if (e == hipSuccess) e = hipMemsetAsync((uint8_t*)a->blobs[k] + a->sizes[k], 0, 16, ctx->stream);
        if (e != hipSuccess) { gs_asset_destroy(a); return gs::fail_hip(e, "asset replica: allocate blob", __FILE__, __LINE__); }
    }
    *out = a;
    return GS_OK;
}

// the blobs have arrived: describe them
void receive_finish(gs_asset* a, const uint64_t h[kHeaderWords]) {
    a->view.pos = (const uint8_t*)a->blobs[0]; a->view.other = (const uint8_t*)a->blobs[1]; a->view.color = (const uint8_t*)a->blobs[2];
    a->view.sh = (const uint8_t*)a->blobs[3]; a->view.chunk = (const uint8_t*)a->blobs[4];
    a->view.n = (uint32_t)h[1]; a->view.posFmt = (uint32_t)h[2]; a->view.scaleFmt = (uint32_t)h[3];
    a->view.colorFmt = (uint32_t)h[4]; a->view.shFmt = (uint32_t)h[5]; a->view.chunkCount = (uint32_t)h[6];
}

} // namespace

struct gs_comm {
    gs_context* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    uint64_t* headerDev = nullptr;          // 16 x u64 scratch for the asset header
};

using namespace gs;

extern "C" {

int32_t gs_comm_unique_id(uint8_t id_out[GS_COMM_ID_BYTES]) {
    if (!id_out) return fail(GS_ERR_INVALID_ARGUMENT, "id_out is null");
    GS_TRY(need_rccl());
    ncclUniqueId id;
    GS_NCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, GS_COMM_ID_BYTES);
    return GS_OK;
}

int32_t gs_comm_create(gs_context* ctx, int32_t nranks, int32_t rank, const uint8_t id[GS_COMM_ID_BYTES], gs_comm** out) {
    if (!ctx || !id || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(GS_ERR_INVALID_ARGUMENT, "rank / nranks out of range");
    GS_TRY(need_rccl());
    GS_HIP(hipSetDevice(ctx->device));
    gs_comm* c = new (std::nothrow) gs_comm();
    if (!c) return fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    ncclUniqueId uid;
    memcpy(uid.internal, id, GS_COMM_ID_BYTES);
    ncclResult_t e = rccl().CommInitRank(&c->comm, nranks, uid, rank);          // collective: returns once every rank has joined
    if (e != ncclSuccess) { delete c; return fail_nccl(e, "ncclCommInitRank"); }
    if (hipMalloc((void**)&c->headerDev, 16 * sizeof(uint64_t)) != hipSuccess) { (void)rccl().CommDestroy(c->comm); delete c; return fail(GS_ERR_OUT_OF_MEMORY, "comm scratch"); }
    *out = c;
    return GS_OK;
}

int32_t gs_comm_destroy(gs_comm* c) {
    if (!c) return GS_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->headerDev) (void)hipFree(c->headerDev);
    delete c;
    return GS_OK;
}

int32_t gs_comm_info(const gs_comm* c, int32_t* nranks, int32_t* rank) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "comm is null");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return GS_OK;
}

// Collective over `comm`.  On `root`, `asset_on_root` is the asset to replicate (created on the comm's context) and *out
// receives the same handle; on every other rank `asset_on_root` is ignored and *out receives a new asset that owns
// device copies of the five blobs.  Blocks until the blobs have arrived (a load-time operation).
// Every rank leaves through the same door: after the header every rank contributes a status word (its argument check on the
// root, its allocations elsewhere, any local HIP failure up to that point) to one ncclAllReduce(min), so that a rank that cannot
// take part makes ALL ranks return GS_ERR_COMM together instead of leaving the others parked in a broadcast.  Errors AFTER the
// status exchange (a blob broadcast or the final synchronise failing) and errors returned by a collective call itself are fatal
// for the communicator: destroy it (the other ranks' collectives fail in turn).
int32_t gs_asset_broadcast(gs_comm* c, gs_asset* asset_on_root, int32_t root, gs_asset** out) {
    if (!c || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");      // (no communicator: there is no collective to leave others parked in)
    *out = nullptr;
    // Local failures on the way to the status exchange are FOLDED into `mine` instead of returning: a rank that left here would
    // leave the others parked in the next collective -- that includes a bad `root` (the rank still takes part, as a receiver of root 0's
    // header, and reports its own error after the exchange) and a hipSetDevice that fails.  What cannot be folded is a failure of a
    // collective call itself (ncclBroadcast / ncclAllReduce returning an error): the communicator is then unusable on this rank -- the
    // caller destroys it (gs_comm_destroy aborts what is in flight) -- and the other ranks see their own collective fail.
    int32_t mine = GS_OK;
    auto note_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && mine == GS_OK) mine = fail_hip(e, what, __FILE__, __LINE__); };
    if (root < 0 || root >= c->nranks) { mine = fail(GS_ERR_INVALID_ARGUMENT, "root out of range"); root = 0; }
    const bool isRoot = c->rank == root;
    gs_context* ctx = c->ctx;
    note_hip(hipSetDevice(ctx->device), "asset broadcast: hipSetDevice");
    hipStream_t st = ctx->stream;
    // a root without a usable asset (or with an error of its own so far) still enters the collectives, with a header nobody accepts, so that nobody hangs
    const bool rootArgOk = !isRoot || (mine == GS_OK && asset_on_root && asset_on_root->ctx == c->ctx);

    // ---- header: formats, count, blob sizes
    uint64_t h[kHeaderWords] = {0};
    if (isRoot) {
        if (rootArgOk) pack_header(asset_on_root, h);
        const hipError_t e = hipMemcpyAsync(c->headerDev, h, sizeof(h), hipMemcpyHostToDevice, st);
        note_hip(e, "asset broadcast: header upload");
        if (e != hipSuccess) (void)hipMemsetAsync(c->headerDev, 0, sizeof(h), st);      // a header nobody accepts
    }
    GS_NCCL(rccl().Broadcast(c->headerDev, c->headerDev, sizeof(h), ncclUint8, root, c->comm, st));
    {
        hipError_t e = hipMemcpyAsync(h, c->headerDev, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        note_hip(e, "asset broadcast: header download");
        if (e != hipSuccess) memset(h, 0, sizeof(h));
    }

    // ---- this rank's half, then the status exchange
    gs_asset* a = asset_on_root;
    bool allocated = false;
    if (mine == GS_OK) {
        if (!header_ok(h)) mine = isRoot ? fail(GS_ERR_INVALID_ARGUMENT, "the root must pass an asset of the comm's context") : fail(GS_ERR_COMM, "asset broadcast: bad header received");
        else if (!isRoot) { mine = receive_alloc(ctx, h, &a); allocated = mine == GS_OK; }
    }
    int32_t* statusDev = (int32_t*)(c->headerDev + kHeaderWords - 1);             // last header word: scratch for the status
    {
        const hipError_t e = hipMemcpyAsync(statusDev, &mine, sizeof(int32_t), hipMemcpyHostToDevice, st);
        note_hip(e, "asset broadcast: status upload");
        if (e != hipSuccess) (void)hipMemsetAsync(statusDev, 0xff, sizeof(int32_t), st);   // -1: "this rank failed"
    }
    GS_NCCL(rccl().AllReduce(statusDev, statusDev, 1, ncclInt32, ncclMin, c->comm, st));
    int32_t all = GS_OK;
    {
        hipError_t e = hipMemcpyAsync(&all, statusDev, sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { note_hip(e, "asset broadcast: status download"); all = mine; }   // (this rank cannot learn the verdict: it does not take part in the blobs)
    }
    if (all != GS_OK || mine != GS_OK) {
        if (allocated) gs_asset_destroy(a);
        if (mine != GS_OK) return mine;                                             // the detail of this rank's own failure stands
        gs::set_error_detail("asset broadcast: another rank could not take part (its error %d)", all);
        return GS_ERR_COMM;
    }

    // ---- the blobs, one broadcast each (in place on the root)
    for (int k = 0; k < 5; ++k) {
        if (!h[7 + k]) continue;
        ncclResult_t e = rccl().Broadcast(a->blobs[k], a->blobs[k], (size_t)h[7 + k], ncclUint8, root, c->comm, st);
        if (e != ncclSuccess) { if (!isRoot) gs_asset_destroy(a); return fail_nccl(e, "ncclBroadcast(blob)"); }
    }
    { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) { if (!isRoot) gs_asset_destroy(a); return fail_hip(e, "asset broadcast sync", __FILE__, __LINE__); } }
    if (!isRoot) receive_finish(a, h);
    *out = a;
    return GS_OK;
}

// A replica of `src` on `dst_ctx` (the same or another GPU of this process): the receive half of gs_asset_broadcast --
// header, padded allocations, blob copies, view -- with a device copy in place of ncclBroadcast.  A single-process host that
// drives several GPUs (one gs_context each, as a Unity player would) replicates its asset with this instead of a communicator.
int32_t gs_asset_replicate(gs_context* dst_ctx, const gs_asset* src, gs_asset** out) {
    if (!dst_ctx || !src || !out) return fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    uint64_t h[kHeaderWords];
    pack_header(src, h);
    if (!header_ok(h)) return fail(GS_ERR_INVALID_ASSET, "source asset is empty");
    GS_HIP(hipSetDevice(src->ctx->device));
    GS_HIP(hipStreamSynchronize(src->ctx->stream));                                // uploads of the source have landed
    GS_HIP(hipSetDevice(dst_ctx->device));
    gs_asset* a = nullptr;
    GS_TRY(receive_alloc(dst_ctx, h, &a));
    for (int k = 0; k < 5; ++k) {
        if (!h[7 + k]) continue;
        const hipError_t e = hipMemcpyPeerAsync(a->blobs[k], dst_ctx->device, src->blobs[k], src->ctx->device, (size_t)h[7 + k], dst_ctx->stream);
        if (e != hipSuccess) { gs_asset_destroy(a); return fail_hip(e, "asset replica: copy blob", __FILE__, __LINE__); }
    }
    { const hipError_t e = hipStreamSynchronize(dst_ctx->stream); if (e != hipSuccess) { gs_asset_destroy(a); return fail_hip(e, "asset replica sync", __FILE__, __LINE__); } }
    receive_finish(a, h);
    *out = a;
    return GS_OK;
}

} // extern "C"
