// gs_view.hip -- CSCalcViewData (SplatUtilities.compute:189-252) for gfx950, fused with the set-up the
// rasteriser's vertex stage would do per splat (RenderGaussianSplats.shader:35-77).
//
// One thread per splat, 256-thread workgroups aligned to the 256-splat chunks of the asset, so the 64-byte
// ChunkInfo is wave-uniform (scalar loads) and the 16x16 Morton colour tile of a workgroup is exactly one
// texture tile.  Per splat the kernel writes, all in splat-index order and all staged through LDS so that the
// global stores are whole, coalesced uint4s:
//   view[s]  40 B  the reference's SplatViewData (m_GpuView; parity surface)
//   rec[s]   32 B  centre in pixels + the two axes + rgba16f: what the blend kernel reads per (tile, splat) pair
//   rect[s]   8 B  inclusive pixel rectangle of the splat's footprint (gsm::PackPixelRect), or 0 if it is culled; the draw turns it
//                  into a tile rectangle of whatever tile shape it composites with
// Writing rec/rect here (where everything is in registers) means the binning kernel only gathers 8 B per sorted
// position instead of the 40-B view record, and never writes records itself.
#include "gs_common.h"

namespace gs {

namespace {

// Reader of one splat's SH record staged in LDS as dwords (record stride padded to an odd dword count: conflict-free).
// k is a compile-time constant wherever it is called from, so every index below folds to an immediate LDS offset.
template <int FMT> struct SHFromLds {
    const uint32_t* rec;
    __device__ __forceinline__ void begin(const uint8_t*, uint32_t) {}
    __device__ __forceinline__ uint32_t half_at(int h) const { return (rec[h >> 1] >> ((h & 1) * 16)) & 0xffffu; }
    __device__ __forceinline__ gsm::V3 load(int k) const {
        if (FMT == 0) return { gsm::u2f(rec[(k - 1) * 3]), gsm::u2f(rec[(k - 1) * 3 + 1]), gsm::u2f(rec[(k - 1) * 3 + 2]) };
        if (FMT == 2) return gsm::Dec_11_10_11(rec[k - 1]);
        if (FMT == 3) return gsm::Dec_5_6_5(half_at(k - 1));
        return { gsm::f16tof32(half_at((k - 1) * 3)), gsm::f16tof32(half_at((k - 1) * 3 + 1)), gsm::f16tof32(half_at((k - 1) * 3 + 2)) };
    }
    __device__ __forceinline__ gsm::RGB load_rgb(int k) const {
        if (FMT == 3) {                                          // 5.6.5: the two decode multiplies of (r, g) as one packed instruction
            const uint32_t e = half_at(k - 1);
            return { gsm::F2{ (float)(e & 31), (float)((e >> 5) & 63) } * gsm::F2{ GS_R31, GS_R63 }, (float)((e >> 11) & 31) * GS_R31 };
        }
        const gsm::V3 v = load(k);
        return { gsm::F2{ v.x, v.y }, v.z };
    }
};

constexpr int sh_rec_dwords(int fmt) { return fmt == 0 ? 48 : (fmt == 1 ? 24 : (fmt == 2 ? 15 : 8)); }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// SHMODE 0..3: per-splat SH records of SHFormat Float32 / Float16 / Norm11 / Norm6, staged through LDS;
// SHMODE 4: SH read straight from the blob (Cluster* tables are gathered by a per-splat index, nothing to stage).
//
// SH is two thirds of the bytes this kernel reads (32 of 48.25 B per splat at Medium, 192 of 236 at VeryHigh), laid out
// as one record per splat.  Read record-per-lane it is 15 narrow loads at a 32..192-byte lane stride, each touching 16-64
// cache lines for 4 bytes apiece and relying on L1 to keep ~2-12 KB per wave alive in between.  Instead the workgroup
// copies its 256 contiguous records (8-48 KB) with lane-contiguous 16-byte loads, parks them in LDS and decodes from there.
//
// FULL = true is the reference's kernel: every splat in front of the camera gets its colour and the N x 40 B view buffer
// (m_GpuView) is written.  Nothing in this renderer reads that buffer -- the compositor consumes rec/rect -- so the
// per-frame launch is FULL = false: geometry for every splat, then SH + colour ONLY for the splats whose footprint
// reaches a tile (a third of the scene for C2; culling is spatially coherent and the asset is in Morton order, so whole
// waves and whole workgroups drop out), and no view write.  gs_renderer_download_view re-runs the frame's launch with
// FULL = true on demand, so the parity surface is unchanged.  VALU-bound either way (~1200 instructions per splat FULL).
template <int SHMODE, bool FULL>
__global__ __launch_bounds__(256) void calc_view_kernel(gsm::AssetView a, gsm::FrameConsts P, gsm::EditView E, ViewOutputs O) {
    GS_VIEW_PRIORITY();
    gsm::ViewData* __restrict__ out = O.view;
    SplatRec* __restrict__ recs = O.recs;
    uint2* __restrict__ rects = O.rects;
    unsigned long long* __restrict__ visMask = O.visMask;
    constexpr int REC = sh_rec_dwords(SHMODE < 4 ? SHMODE : 3);       // dwords per record
    constexpr int STRIDE = REC | 1;                                    // odd LDS stride
    constexpr int SH_DW = SHMODE < 4 ? 256 * STRIDE : 0;
    constexpr int NVEC = (REC * 256 / 4 + 255) / 256;                  // 16-byte vectors per thread
    __shared__ uint4 s_stage[cmax(SH_DW, FULL ? 256 * 10 : 4) / 4 + 1];
    __shared__ int s_any;
    uint32_t* s_dw = (uint32_t*)s_stage;
    const uint32_t base = blockIdx.x * 256u;
    const uint32_t idx = base + threadIdx.x;
    const uint32_t cnt = min(256u, a.n - base);
    const uint32_t totalDw = cnt * (uint32_t)REC, totalVec = totalDw >> 2;
    const uint32_t* shSrc = (const uint32_t*)(a.sh + (size_t)base * (REC * 4));

    // copy cnt*REC dwords from a 16-byte aligned address (256*REC*4 bytes per workgroup) into LDS: record r, dword o -> r*STRIDE + o
    uint4 shv[SHMODE < 4 ? NVEC : 1];
    uint32_t shTail = 0;
    auto sh_issue = [&]() {
#pragma unroll
        for (int it = 0; it < NVEC; ++it) {
            const uint32_t j = (uint32_t)it * 256u + threadIdx.x;
            shv[it] = make_uint4(0u, 0u, 0u, 0u);
            if (j < totalVec) shv[it] = ((const uint4*)shSrc)[j];
        }
        if ((totalDw & 3u) && threadIdx.x < (totalDw & 3u)) shTail = shSrc[(totalVec << 2) + threadIdx.x];
    };
    auto sh_park = [&]() {
#pragma unroll
        for (int it = 0; it < NVEC; ++it) {
            const uint32_t j = (uint32_t)it * 256u + threadIdx.x;
            if (j < totalVec) {
                const uint32_t d = j << 2;
                const uint32_t v[4] = { shv[it].x, shv[it].y, shv[it].z, shv[it].w };
                if (REC % 4 == 0) {
                    const uint32_t r = d / (uint32_t)REC, o = d - r * (uint32_t)REC;
#pragma unroll
                    for (int c = 0; c < 4; ++c) s_dw[r * STRIDE + o + c] = v[c];
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const uint32_t dd = d + c, r = dd / (uint32_t)REC; s_dw[r * STRIDE + (dd - r * (uint32_t)REC)] = v[c]; }
                }
            }
        }
        if ((totalDw & 3u) && threadIdx.x < (totalDw & 3u)) { const uint32_t dd = (totalVec << 2) + threadIdx.x, r = dd / (uint32_t)REC; s_dw[r * STRIDE + (dd - r * (uint32_t)REC)] = shTail; }
    };
    auto shade = [&](gsm::ViewPartial& vp) {
        if (SHMODE < 4) {
            SHFromLds<SHMODE < 4 ? SHMODE : 3> src;
            src.rec = s_dw + threadIdx.x * STRIDE;
            gsm::CalcViewColor(a, P, idx, vp, src);
        } else {
            gsm::SHFromBlob src;
            gsm::CalcViewColor(a, P, idx, vp, src);
        }
    };

    gsm::ViewPartial vp;
    gsm::SplatFootprint fp;
    uint2 rect = make_uint2(0u, 0u);
    bool visible = false;
    const bool shStaged = SHMODE < 4;

    if (FULL) {
        // the loads are in flight while the positions are projected
        if (shStaged) { sh_issue(); sh_park(); __syncthreads(); }
        if (idx < a.n) {
            gsm::CalcViewGeom(a, P, E, idx, vp);
            if (vp.front) shade(vp);
            const bool ok = gsm::PrepareSplat(vp.view, P.screenW, P.screenH, P.nearClip, P.farClip, fp);
            visible = ok && fp.x0 <= fp.x1;
        }
    } else {
        // whole-chunk frustum cull: lane c & 7 tests corner c of the chunk's position box against the 6 (pushed-out) planes
        if (P.cullOn && blockIdx.x < a.chunkCount) {
            const uint32_t m = gsm::ChunkCornerOutside(a, P, blockIdx.x, threadIdx.x & 7u);
            bool outside = false;
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) outside = outside || __all((m >> pl) & 1u);      // all 8 corners beyond plane pl
            if (outside) {                                                                    // workgroup-uniform
                if (idx < a.n) rects[idx] = make_uint2(0u, 0u);
                if ((threadIdx.x & 63u) == 0u && idx < a.n) visMask[idx >> 6] = 0ull;
                if ((threadIdx.x & 63u) == 0u && idx < a.n) wave_flags_of(visMask, a.n)[idx >> 6] = 0u;
                return;
            }
        }
        if (threadIdx.x == 0) s_any = 0;
        if (idx < a.n) {
            gsm::CalcViewGeom(a, P, E, idx, vp, true, true);       // early out for splats that cannot reach the screen; vp only read if drawn
            const bool ok = vp.front && !vp.culled && gsm::PrepareSplat(vp.view, P.screenW, P.screenH, P.nearClip, P.farClip, fp);
            visible = ok && fp.x0 <= fp.x1;
        }
        __syncthreads();
        if (__any(visible) && (threadIdx.x & 63u) == 0u) s_any = 1;          // benign race: every writer stores 1
        __syncthreads();
        if (s_any) {                                                          // workgroup-uniform
            if (shStaged) { sh_issue(); sh_park(); __syncthreads(); }
            if (visible) shade(vp);
        }
    }

    if (idx < a.n) {
        if (visible) {
            gsm::PackPixelRect(fp, rect.x, rect.y);
            // the blend kernel only ever reads records of splats that reach a tile
            uint4* rp = (uint4*)(recs + idx);
            rp[0] = make_uint4(gsm::f2u(fp.cx), gsm::f2u(fp.cy), gsm::f2u(vp.view.axis1[0]), gsm::f2u(vp.view.axis1[1]));
            rp[1] = make_uint4(gsm::f2u(vp.view.axis2[0]), gsm::f2u(vp.view.axis2[1]), vp.view.color[0], vp.view.color[1]);
        }
        rects[idx] = rect;
    }
    const unsigned long long vb = __ballot(visible);
    if ((threadIdx.x & 63u) == 0u && (idx < a.n)) visMask[idx >> 6] = vb;          // 1 bit per splat
    // 1 byte per wave: what the binning pass tests first (N / 64 bytes, cache resident: a position of the depth order whose 64 index
    // neighbours are all culled costs the binning no further memory request)
    if ((threadIdx.x & 63u) == 0u && (idx < a.n)) wave_flags_of(visMask, a.n)[idx >> 6] = vb != 0ull ? 1u : 0u;
    if (!FULL) return;

    // ---- FULL: the 40-byte records, staged through LDS so that the global stores are whole uint4s
    if (shStaged) __syncthreads();                                                   // every thread is done with the SH records
    if (idx < a.n) {
        const gsm::ViewData& v = vp.view;
        uint32_t* o = s_dw + threadIdx.x * 10;
        o[0] = gsm::f2u(v.pos[0]); o[1] = gsm::f2u(v.pos[1]); o[2] = gsm::f2u(v.pos[2]); o[3] = gsm::f2u(v.pos[3]);
        o[4] = gsm::f2u(v.axis1[0]); o[5] = gsm::f2u(v.axis1[1]); o[6] = gsm::f2u(v.axis2[0]); o[7] = gsm::f2u(v.axis2[1]);
        o[8] = v.color[0]; o[9] = v.color[1];
    }
    __syncthreads();
    {
        const uint32_t dwords = cnt * 10u;
        uint32_t* g = (uint32_t*)(out + base);             // base*40 B is 16-B aligned (256*40 = 10240)
        const uint32_t vec = dwords / 4u;
        for (uint32_t j = threadIdx.x; j < vec; j += 256u) ((uint4*)g)[j] = s_stage[j];
        for (uint32_t j = vec * 4u + threadIdx.x; j < dwords; j += 256u) g[j] = s_dw[j];
    }
}

} // namespace

void flatten_params(const gs_frame_params* p, gsm::FrameConsts& c) {
    memcpy(c.mv, p->matrix_mv, 12 * sizeof(float));
    memcpy(c.o2w, p->matrix_object_to_world, 12 * sizeof(float));
    memcpy(c.w2o, p->matrix_world_to_object, 12 * sizeof(float));
    memcpy(c.vp, p->matrix_vp, 16 * sizeof(float));
    gsm::FrameConstsFromProjection(c, p->proj_m00, p->proj_m11, p->screen_w); c.screenW = p->screen_w; c.screenH = p->screen_h;
    c.camx = p->cam_pos_world[0]; c.camy = p->cam_pos_world[1]; c.camz = p->cam_pos_world[2];
    c.splatScale = p->splat_scale; c.opacityScale = p->opacity_scale;
    c.shOrder = p->sh_order; c.shOnly = p->sh_only;
    c.nearClip = p->near_clip; c.farClip = p->far_clip;
    gsm::FrameConstsChunkCull(c);
}

template <bool FULL>
static void launch_calc_view(int mode, uint32_t grid, hipStream_t st, const gsm::AssetView& a, const gsm::FrameConsts& c, const gsm::EditView& e,
                             const ViewOutputs& o) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((calc_view_kernel<0, FULL>), dim3(grid), dim3(256), 0, st, a, c, e, o); break;
        case 1: hipLaunchKernelGGL((calc_view_kernel<1, FULL>), dim3(grid), dim3(256), 0, st, a, c, e, o); break;
        case 2: hipLaunchKernelGGL((calc_view_kernel<2, FULL>), dim3(grid), dim3(256), 0, st, a, c, e, o); break;
        case 3: hipLaunchKernelGGL((calc_view_kernel<3, FULL>), dim3(grid), dim3(256), 0, st, a, c, e, o); break;
        default: hipLaunchKernelGGL((calc_view_kernel<4, FULL>), dim3(grid), dim3(256), 0, st, a, c, e, o); break;
    }
}

// full = true: also evaluate the colour of every splat in front of the camera and write the N x 40 B view buffer
int32_t enqueue_calc_view(gs_context* ctx, const gsm::AssetView& a, const gs_frame_params* p, const gsm::EditView& e, const ViewOutputs& out, bool full) {
    gsm::FrameConsts c;
    flatten_params(p, c);
    const uint32_t grid = (a.n + 255u) / 256u;
    // per-splat SH records are staged through LDS (needs the blob 16-byte aligned, which hipMalloc and torch guarantee);
    // Cluster* tables, an unaligned borrowed blob, or SH switched off read straight from the blob
    const int mode = (a.shFmt <= 3 && (((uintptr_t)a.sh) & 15u) == 0 && p->sh_order >= 1) ? (int)a.shFmt : 4;
    if (full) launch_calc_view<true>(mode, grid, ctx->stream, a, c, e, out);
    else launch_calc_view<false>(mode, grid, ctx->stream, a, c, e, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

} // namespace gs
