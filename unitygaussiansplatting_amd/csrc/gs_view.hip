// gs_view.hip -- CSCalcViewData (SplatUtilities.compute:189-252) for gfx950.
//
// One thread per splat, 256-thread workgroups aligned to the 256-splat chunks of the asset, so the 64-byte
// ChunkInfo is wave-uniform (scalar loads) and the 16x16 Morton colour tile of a workgroup is exactly one
// texture tile.  The per-splat arithmetic is gsm::CalcViewData (gs_device_math.h); the 40-byte records are
// staged through LDS so that the global stores are fully coalesced 16-byte stores.
#include "gs_common.h"

namespace gs {

namespace {

__global__ __launch_bounds__(256) void calc_view_kernel(gsm::AssetView a, gsm::FrameConsts P, gsm::ViewData* __restrict__ out) {
    __shared__ uint32_t s_out[256 * 10];
    const uint32_t base = blockIdx.x * 256u;
    const uint32_t idx = base + threadIdx.x;
    if (idx < a.n) {
        const gsm::ViewData v = gsm::CalcViewData(a, P, idx);
        uint32_t* o = s_out + threadIdx.x * 10;
        o[0] = gsm::f2u(v.pos[0]); o[1] = gsm::f2u(v.pos[1]); o[2] = gsm::f2u(v.pos[2]); o[3] = gsm::f2u(v.pos[3]);
        o[4] = gsm::f2u(v.axis1[0]); o[5] = gsm::f2u(v.axis1[1]); o[6] = gsm::f2u(v.axis2[0]); o[7] = gsm::f2u(v.axis2[1]);
        o[8] = v.color[0]; o[9] = v.color[1];
    }
    __syncthreads();
    const uint32_t cnt = min(256u, a.n - base);
    const uint32_t dwords = cnt * 10u;
    uint32_t* g = (uint32_t*)(out + base);             // base*40 B is 16-B aligned (256*40 = 10240)
    const uint32_t vec = dwords / 4u;                   // cnt*10/4: whole uint4s
    for (uint32_t j = threadIdx.x; j < vec; j += 256u) ((uint4*)g)[j] = ((const uint4*)s_out)[j];
    for (uint32_t j = vec * 4u + threadIdx.x; j < dwords; j += 256u) g[j] = s_out[j];
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

int32_t enqueue_calc_view(gs_context* ctx, const gsm::AssetView& a, const gs_frame_params* p, gsm::ViewData* out) {
    gsm::FrameConsts c;
    flatten_params(p, c);
    const uint32_t grid = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(calc_view_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, c, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

} // namespace gs
