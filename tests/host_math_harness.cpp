// Test-only: compiles the kernels' arithmetic header (csrc/gs_device_math.h) for the HOST so that it can be
// compared with the oracle bit for bit on a box without a GPU.  Never part of the shipped library.
#include "../include/gsplat_c.h"
#include "../unitygaussiansplatting_amd/csrc/gs_device_math.h"

static gsm::AssetView mk(const gs_asset_desc* d) {
    gsm::AssetView a;
    a.pos = (const uint8_t*)d->pos_data; a.other = (const uint8_t*)d->other_data; a.color = (const uint8_t*)d->color_data;
    a.sh = (const uint8_t*)d->sh_data; a.chunk = (const uint8_t*)d->chunk_data;
    a.n = d->splat_count; a.posFmt = d->pos_format; a.scaleFmt = d->scale_format; a.colorFmt = d->color_format; a.shFmt = d->sh_format;
    a.chunkCount = (d->chunk_data && d->chunk_size) ? (uint32_t)(d->chunk_size / 64) : 0;
    return a;
}
static gsm::FrameConsts fl(const gs_frame_params* p) {
    gsm::FrameConsts c;
    memcpy(c.mv, p->matrix_mv, 48); memcpy(c.o2w, p->matrix_object_to_world, 48); memcpy(c.w2o, p->matrix_world_to_object, 48);
    memcpy(c.vp, p->matrix_vp, 64);
    gsm::FrameConstsFromProjection(c, p->proj_m00, p->proj_m11, p->screen_w); c.screenW = p->screen_w; c.screenH = p->screen_h;
    c.camx = p->cam_pos_world[0]; c.camy = p->cam_pos_world[1]; c.camz = p->cam_pos_world[2];
    c.splatScale = p->splat_scale; c.opacityScale = p->opacity_scale; c.shOrder = p->sh_order; c.shOnly = p->sh_only;
    c.nearClip = p->near_clip; c.farClip = p->far_clip;
    gsm::FrameConstsChunkCull(c);
    return c;
}
extern "C" {
void hm_calc_view_ex(const gs_asset_desc* d, const gs_frame_params* p, const gs_cutout* cutouts, uint32_t cutoutCount,
                     const uint32_t* deletedBits, void* out) {
    const gsm::AssetView a = mk(d); const gsm::FrameConsts c = fl(p);
    gsm::EditView e; e.deletedBits = deletedBits; e.cutouts = (const uint32_t*)cutouts; e.cutoutCount = cutoutCount;
    gsm::ViewData* o = (gsm::ViewData*)out;
    for (uint32_t i = 0; i < a.n; ++i) o[i] = gsm::CalcViewData(a, c, e, i);
}
void hm_calc_view(const gs_asset_desc* d, const gs_frame_params* p, void* out) { hm_calc_view_ex(d, p, nullptr, 0, nullptr, out); }
void hm_calc_distances(const gs_asset_desc* d, const uint32_t* order, const float* m, uint32_t* keys) {
    const gsm::AssetView a = mk(d);
    for (uint32_t i = 0; i < a.n; ++i) keys[i] = gsm::SortKey(a, order[i], m[8], m[9], m[10], m[11]);
}
// footprints: out[i] = {tx0,tx1,ty0,ty1, ok}; centres: cxy[i] = {cx,cy}; tiles[i] (optional) = tiles the binning emits
void hm_prepare_ex(const void* view, uint32_t n, const gs_frame_params* p, int32_t* out, float* cxy, uint32_t* tiles);
void hm_prepare(const void* view, uint32_t n, const gs_frame_params* p, int32_t* out, float* cxy) { hm_prepare_ex(view, n, p, out, cxy, nullptr); }
void hm_prepare_ex(const void* view, uint32_t n, const gs_frame_params* p, int32_t* out, float* cxy, uint32_t* tiles) {
    const gsm::ViewData* v = (const gsm::ViewData*)view;
    for (uint32_t i = 0; i < n; ++i) {
        gsm::SplatFootprint fp;
        const bool ok = gsm::PrepareSplat(v[i], p->screen_w, p->screen_h, p->near_clip, p->far_clip, fp);
        // the header's footprint is a pixel rectangle; reported here as 16x16 tiles (the oracle's default tile shape)
        const bool any = fp.x0 <= fp.x1;
        out[i * 5 + 0] = any ? fp.x0 >> 4 : 1; out[i * 5 + 1] = any ? fp.x1 >> 4 : 0; out[i * 5 + 2] = any ? fp.y0 >> 4 : 1; out[i * 5 + 3] = any ? fp.y1 >> 4 : 0; out[i * 5 + 4] = ok;
        cxy[i * 2] = fp.cx; cxy[i * 2 + 1] = fp.cy;
        if (tiles) tiles[i] = (ok && any) ? (uint32_t)(((fp.x1 >> 4) - (fp.x0 >> 4) + 1) * ((fp.y1 >> 4) - (fp.y0 >> 4) + 1)) : 0u;
    }
}
// the culling predicate shared by the binning (16x16 tiles, half = 7.5) and the blend kernel (8x8 quadrants, half = 3.5)
int32_t hm_block_may_touch(float bcx, float bcy, float half, float cx, float cy, float u1x, float u1y, float u2x, float u2y, float r2) {
    return gsm::BlockMayTouch(bcx, bcy, half, cx, cy, u1x, u1y, u2x, u2y, r2) ? 1 : 0;
}
float hm_log_det(float x) { return gsm::LogDet(x); }
float hm_exp2_det(float y) { return gsm::Exp2Det(y); }
// the header's discard decision given a native alpha; returns the alpha, *live_out = kept
float hm_decide_alpha(float alphaNative, float y, float a, int32_t* live_out) { bool live; const float r = gsm::DecideAlpha(alphaNative, y, a, live); *live_out = live ? 1 : 0; return r; }
// early-cull property: out[i] = {culled by CalcViewGeom(allowCull), tiles of the full path's footprint}
void hm_cull_check(const gs_asset_desc* d, const gs_frame_params* p, uint32_t* out) {
    const gsm::AssetView a = mk(d); const gsm::FrameConsts c = fl(p);
    gsm::EditView e; e.deletedBits = nullptr; e.cutouts = nullptr; e.cutoutCount = 0;
    for (uint32_t i = 0; i < a.n; ++i) {
        gsm::ViewPartial full, fast;
        gsm::CalcViewGeom(a, c, e, i, full, false);
        gsm::CalcViewGeom(a, c, e, i, fast, true);
        gsm::SplatFootprint fp;
        const bool ok = full.front && gsm::PrepareSplat(full.view, c.screenW, c.screenH, c.nearClip, c.farClip, fp);
        out[i * 2] = fast.culled ? 1u : 0u;
        out[i * 2 + 1] = (ok && fp.x0 <= fp.x1) ? (uint32_t)(((fp.x1 >> 4) - (fp.x0 >> 4) + 1) * ((fp.y1 >> 4) - (fp.y0 >> 4) + 1)) : 0u;
        if (!fast.culled && fast.front) {       // not culled: the geometry must be the full path's, bit for bit
            if (memcmp(&fast.view, &full.view, sizeof(gsm::ViewData)) != 0) out[i * 2] |= 2u;
        }
    }
}
// whole-chunk cull: out[ci] = 1 if the chunk is culled; returns cullOn
uint32_t hm_chunk_cull(const gs_asset_desc* d, const gs_frame_params* p, uint8_t* out) {
    const gsm::AssetView a = mk(d); const gsm::FrameConsts c = fl(p);
    for (uint32_t ci = 0; ci < a.chunkCount; ++ci) out[ci] = gsm::ChunkOutside(a, c, ci) ? 1 : 0;
    return c.cullOn;
}
uint32_t hm_f32tof16(float f) { return gsm::f32tof16(f); }
float hm_f16tof32(uint32_t h) { return gsm::f16tof32(h); }
}
