// gs_view.hip -- CSCalcViewData (SplatUtilities.compute:189-252) for gfx950, fused with the set-up the
// rasteriser's vertex stage would do per splat (RenderGaussianSplats.shader:35-77).
//
// One thread per splat, 256-thread workgroups aligned to the 256-splat chunks of the asset, so the 64-byte
// ChunkInfo is wave-uniform (scalar loads) and the 16x16 Morton colour tile of a workgroup is exactly one
// texture tile.  Per splat the kernel writes, all in splat-index order and all staged through LDS so that the
// global stores are whole, coalesced uint4s:
//   view[s]  40 B  the reference's SplatViewData (m_GpuView; parity surface)
//   rec[s]   32 B  centre in pixels + the two axes + rgba16f: what the blend kernel reads per (tile, splat) pair
//   rect[s]   8 B  inclusive 16x16-tile rectangle of the splat's footprint, or 0 if it is culled
// Writing rec/rect here (where everything is in registers) means the binning kernel only gathers 8 B per sorted
// position instead of the 40-B view record, and never writes records itself.
#include "gs_common.h"

namespace gs {

namespace {

__global__ __launch_bounds__(256) void calc_view_kernel(gsm::AssetView a, gsm::FrameConsts P, gsm::ViewData* __restrict__ out,
                                                        SplatRec* __restrict__ recs, uint2* __restrict__ rects) {
    __shared__ uint4 s_stage[256 * 10 / 4];
    uint32_t* s_out = (uint32_t*)s_stage;
    const uint32_t base = blockIdx.x * 256u;
    const uint32_t idx = base + threadIdx.x;
    const uint32_t cnt = min(256u, a.n - base);
    gsm::ViewData v;
    gsm::SplatFootprint fp;
    uint2 rect = make_uint2(0u, 0u);
    if (idx < a.n) {
        v = gsm::CalcViewData(a, P, idx);
        const bool ok = gsm::PrepareSplat(v, P.screenW, P.screenH, P.nearClip, P.farClip, fp);
        if (ok && fp.tx0 <= fp.tx1) {
            rect.x = (uint32_t)fp.tx0 | ((uint32_t)fp.ty0 << 16);
            rect.y = (uint32_t)(fp.tx1 - fp.tx0 + 1) | ((uint32_t)(fp.ty1 - fp.ty0 + 1) << 16);
        }
        rects[idx] = rect;
        uint32_t* o = s_out + threadIdx.x * 10;
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
        for (uint32_t j = vec * 4u + threadIdx.x; j < dwords; j += 256u) g[j] = s_out[j];
    }
    __syncthreads();
    if (idx < a.n) {
        uint32_t* o = s_out + threadIdx.x * 8;
        o[0] = gsm::f2u(fp.cx); o[1] = gsm::f2u(fp.cy);
        o[2] = gsm::f2u(v.axis1[0]); o[3] = gsm::f2u(v.axis1[1]); o[4] = gsm::f2u(v.axis2[0]); o[5] = gsm::f2u(v.axis2[1]);
        o[6] = v.color[0]; o[7] = v.color[1];
    }
    __syncthreads();
    {
        uint4* g = (uint4*)(recs + base);
        for (uint32_t j = threadIdx.x; j < cnt * 2u; j += 256u) g[j] = s_stage[j];
    }
}

} // namespace

void flatten_params(const gs_frame_params* p, gsm::FrameConsts& c) {
    memcpy(c.mv, p->matrix_mv, 12 * sizeof(float));
    memcpy(c.o2w, p->matrix_object_to_world, 12 * sizeof(float));
    memcpy(c.w2o, p->matrix_world_to_object, 12 * sizeof(float));
    memcpy(c.vp, p->matrix_vp, 16 * sizeof(float));
    c.p00 = p->proj_m00; c.p11 = p->proj_m11; c.screenW = p->screen_w; c.screenH = p->screen_h;
    c.camx = p->cam_pos_world[0]; c.camy = p->cam_pos_world[1]; c.camz = p->cam_pos_world[2];
    c.splatScale = p->splat_scale; c.opacityScale = p->opacity_scale;
    c.shOrder = p->sh_order; c.shOnly = p->sh_only;
    c.nearClip = p->near_clip; c.farClip = p->far_clip;
}

int32_t enqueue_calc_view(gs_context* ctx, const gsm::AssetView& a, const gs_frame_params* p, gsm::ViewData* out, SplatRec* recs,
                          uint2* rects) {
    gsm::FrameConsts c;
    flatten_params(p, c);
    const uint32_t grid = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(calc_view_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, c, out, recs, rects);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

} // namespace gs
