/* Plain-C consumer of include/gsplat_c.h: links libgsplat_hip.so like a non-C++ host would and walks the parts of the ABI
 * that need no GPU (version, error strings, argument validation, the native importer) and, where a GPU is present, the frame itself:
 * asset from the imported blobs, GS_SORT_VISIBLE on three renderers -- one drawing every frame, two on contexts of their own SHARING the
 * asset with the frames dealt alternately and a sort history of two rows -- and on a fourth with the frames in flight INSIDE the library
 * (gs_renderer_set_frames_in_flight: one renderer, two targets in rotation): every frame the same bits, the order buffers equal at the end.
 * Built and run by tests/test_abi.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gsplat_c.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "abi_smoke: %s failed (line %d): %s\n", #c, __LINE__, gs_last_error_string()); return 1; } } while (0)

int main(void) {
    CHECK(gs_abi_version() == GS_ABI_VERSION);
    CHECK(strcmp(gs_error_string(GS_OK), "ok") == 0);
    CHECK(gs_renderer_sort(NULL, NULL) == GS_ERR_INVALID_ARGUMENT);
    CHECK(gs_target_create(NULL, 4, 4, NULL) == GS_ERR_INVALID_ARGUMENT);

    /* importer: 600 splats on a line, Medium formats */
    enum { N = 600 };
    float *pos = calloc(N * 3, 4), *dc0 = calloc(N * 3, 4), *sh = calloc(N * 45, 4), *op = calloc(N, 4), *sc = calloc(N * 3, 4), *rot = calloc(N * 4, 4);
    for (int i = 0; i < N; ++i) {
        pos[i * 3] = (float)i * 0.01f; pos[i * 3 + 1] = (float)(i % 7) * 0.1f; pos[i * 3 + 2] = (float)(i % 13) * -0.05f;
        op[i] = 2.0f; rot[i * 4] = 1.0f;
        for (int c = 0; c < 3; ++c) { dc0[i * 3 + c] = 0.5f; sc[i * 3 + c] = -4.0f + 0.001f * (float)i; }
    }
    gs_import_input in = { N, pos, dc0, sh, op, sc, rot };
    gs_import_formats fmt = { GS_VECTOR_NORM11, GS_VECTOR_NORM11, GS_COLOR_NORM8X4, GS_SH_NORM6, 1, 1 };
    uint64_t sizes[5];
    CHECK(gs_import_blob_sizes(N, &fmt, sizes) == GS_OK);
    CHECK(sizes[0] == 2400 && sizes[1] == 4800 && sizes[3] == (uint64_t)N * 32 && sizes[4] == 3 * 64);
    void* blobs[5];
    for (int k = 0; k < 5; ++k) blobs[k] = calloc(sizes[k] ? sizes[k] : 1, 1);
    float bmin[3], bmax[3];
    CHECK(gs_import_encode(&in, &fmt, blobs, sizes, bmin, bmax) == GS_OK);
    CHECK(bmin[0] == 0.0f && bmax[0] == 5.99f);
    sizes[3] -= 1;                                              /* a blob that is too small is refused */
    CHECK(gs_import_encode(&in, &fmt, blobs, sizes, bmin, bmax) == GS_ERR_INVALID_ARGUMENT);
    fmt.color_format = GS_COLOR_BC7;                            /* 1 byte per texel: 2048 x 16 texels for 600 splats */
    CHECK(gs_import_blob_sizes(N, &fmt, sizes) == GS_OK && sizes[2] == 2048u * 16u);
    fmt.sh_format = GS_SH_CLUSTER4K;                            /* a 4096-entry SH table needs more than 4096 splats */
    CHECK(gs_import_blob_sizes(N, &fmt, sizes) == GS_ERR_INVALID_ARGUMENT);

    /* without a GPU the context must fail loudly */
    gs_context* ctx = NULL;
    const int32_t rc = gs_context_create(0, NULL, &ctx);
    CHECK(rc == GS_OK || rc == GS_ERR_NO_DEVICE);
    if (rc != GS_OK) { printf("abi_smoke ok (context: no device, as reported)\n"); return 0; }

    /* ---- with a GPU: the frame, through the C-ABI only */
    fmt.color_format = GS_COLOR_NORM8X4; fmt.sh_format = GS_SH_NORM6;
    CHECK(gs_import_blob_sizes(N, &fmt, sizes) == GS_OK);
    CHECK(gs_import_encode(&in, &fmt, blobs, sizes, bmin, bmax) == GS_OK);
    gs_asset_desc d;
    memset(&d, 0, sizeof d);
    d.splat_count = N; d.pos_format = fmt.pos_format; d.scale_format = fmt.scale_format; d.color_format = fmt.color_format; d.sh_format = fmt.sh_format;
    d.pos_data = blobs[0]; d.pos_size = sizes[0]; d.other_data = blobs[1]; d.other_size = sizes[1]; d.color_data = blobs[2]; d.color_size = sizes[2];
    d.sh_data = blobs[3]; d.sh_size = sizes[3]; d.chunk_data = blobs[4]; d.chunk_size = sizes[4];
    gs_asset* asset = NULL;
    CHECK(gs_asset_create(ctx, &d, &asset) == GS_OK);
    gs_context *ctx1 = NULL, *ctx2 = NULL;
    CHECK(gs_context_create(0, NULL, &ctx1) == GS_OK && gs_context_create(0, NULL, &ctx2) == GS_OK);
    enum { W = 160, H = 96, FRAMES = 7 };
    gs_context* cx[3] = { ctx, ctx1, ctx2 };
    gs_renderer* r[3]; gs_target* rt[3];
    for (int k = 0; k < 3; ++k) {                                /* r[0]: every frame; r[1], r[2]: other contexts, the SAME asset */
        CHECK(gs_renderer_create(cx[k], asset, &r[k]) == GS_OK);
        CHECK(gs_renderer_set_sort_mode(r[k], GS_SORT_VISIBLE) == GS_OK);
        CHECK(gs_target_create(cx[k], W, H, &rt[k]) == GS_OK);
    }
    CHECK(gs_renderer_set_sort_history_limit(r[2], 2) == GS_OK && gs_renderer_set_sort_history_limit(r[2], 1) == GS_ERR_INVALID_ARGUMENT);
    gs_renderer* rl = NULL; gs_target* rtl[2];                  /* the same, behind ONE renderer: lanes inside the library */
    CHECK(gs_renderer_create(ctx, asset, &rl) == GS_OK && gs_renderer_set_sort_mode(rl, GS_SORT_VISIBLE) == GS_OK);
    CHECK(gs_renderer_set_frames_in_flight(rl, 0) == GS_ERR_INVALID_ARGUMENT && gs_renderer_set_frames_in_flight(rl, GS_MAX_FRAMES_IN_FLIGHT + 1) == GS_ERR_INVALID_ARGUMENT);
    CHECK(gs_renderer_set_frames_in_flight(rl, 2) == GS_OK);
    int32_t lanes_set = 0, lanes_active = 0;
    CHECK(gs_renderer_frames_in_flight(rl, &lanes_set, &lanes_active) == GS_OK && lanes_set == 2 && lanes_active == 1);
    CHECK(gs_target_create(ctx, W, H, &rtl[0]) == GS_OK && gs_target_create(ctx, W, H, &rtl[1]) == GS_OK);
    static uint16_t img0[W * H * 4], img1[W * H * 4];
    static uint32_t ord0[N], ord1[N], vis0[N], vis1[N];
    for (int f = 0; f < FRAMES; ++f) {
        /* camera at `eye`, axes aligned with the world's (GL view matrix: looks down -Z), moving along x and z */
        const float eye[3] = { 3.0f + 0.2f * (float)f, 0.3f, 5.0f - 0.15f * (float)f };
        const float fov = 60.0f * 3.14159265f / 180.0f, tn = tanf(0.5f * fov), zn = 0.3f, zf = 1000.0f, asp = (float)W / (float)H;
        gs_frame_params p;
        memset(&p, 0, sizeof p);
        float* V = p.matrix_mv;                                  /* model = identity */
        V[0] = V[5] = V[10] = V[15] = 1.0f; V[3] = -eye[0]; V[7] = -eye[1]; V[11] = -eye[2];
        p.matrix_object_to_world[0] = p.matrix_object_to_world[5] = p.matrix_object_to_world[10] = p.matrix_object_to_world[15] = 1.0f;
        memcpy(p.matrix_world_to_object, p.matrix_object_to_world, sizeof p.matrix_world_to_object);
        float P[16] = { 0 };
        P[0] = 1.0f / (asp * tn); P[5] = 1.0f / tn; P[10] = -(zf + zn) / (zf - zn); P[11] = -2.0f * zf * zn / (zf - zn); P[14] = -1.0f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float a = 0.0f; for (int k = 0; k < 4; ++k) a += P[i * 4 + k] * V[k * 4 + j]; p.matrix_vp[i * 4 + j] = a; }
        p.proj_m00 = P[0]; p.proj_m11 = P[5]; p.screen_w = (float)W; p.screen_h = (float)H;
        memcpy(p.cam_pos_world, eye, sizeof eye);
        p.splat_scale = 1.0f; p.opacity_scale = 1.0f; p.sh_order = 3; p.near_clip = zn; p.far_clip = zf;
        float S[16];                                             /* SortPoints' matrix: the view matrix with m20, m21, m22 negated (GaussianSplatRenderer.cs:617-629) */
        memcpy(S, V, sizeof S); S[8] = -S[8]; S[9] = -S[9]; S[10] = -S[10];
        for (int k = 0; k < 3; ++k) CHECK(gs_renderer_sort(r[k], S) == GS_OK);      /* every renderer learns every matrix */
        CHECK(gs_renderer_sort(rl, S) == GS_OK);                 /* (the library tells its lanes) */
        CHECK(gs_renderer_calc_view(rl, &p) == GS_OK && gs_target_clear(rtl[f & 1]) == GS_OK && gs_renderer_draw(rl, &p, rtl[f & 1]) == GS_OK);
        const int lane = 1 + (f & 1);
        const int who[2] = { 0, lane };
        for (int j = 0; j < 2; ++j) {
            const int k = who[j];
            CHECK(gs_renderer_calc_view(r[k], &p) == GS_OK && gs_target_clear(rt[k]) == GS_OK && gs_renderer_draw(r[k], &p, rt[k]) == GS_OK);
        }
        gs_frame_stats st0, st1;
        CHECK(gs_renderer_frame_stats(r[0], &st0) == GS_OK && gs_renderer_frame_stats(r[lane], &st1) == GS_OK);
        CHECK(st0.sort_mode == GS_SORT_VISIBLE && st0.visible_splats > 100 && st0.visible_splats == st1.visible_splats && st0.tile_pairs == st1.tile_pairs);
        CHECK(gs_target_download(rt[0], img0, sizeof img0) == GS_OK && gs_target_download(rt[lane], img1, sizeof img1) == GS_OK);
        CHECK(memcmp(img0, img1, sizeof img0) == 0);
        uint32_t c0 = 0, c1 = 0;
        CHECK(gs_renderer_download_visible_order(r[0], vis0, N, &c0) == GS_OK && gs_renderer_download_visible_order(r[lane], vis1, N, &c1) == GS_OK);
        CHECK(c0 == st0.visible_splats && c0 == c1 && memcmp(vis0, vis1, c0 * sizeof(uint32_t)) == 0);
        /* the frame drawn on a lane inside the library: the same target bits (gs_target_download waits on the context's stream, which the draw made wait for
         * the lane's blend), the same drawn order */
        CHECK(gs_target_download(rtl[f & 1], img1, sizeof img1) == GS_OK && memcmp(img0, img1, sizeof img0) == 0);
        CHECK(gs_renderer_download_visible_order(rl, vis1, N, &c1) == GS_OK && c1 == c0 && memcmp(vis0, vis1, c0 * sizeof(uint32_t)) == 0);
    }
    CHECK(gs_renderer_download_order(rl, ord1, N) == GS_OK && gs_renderer_download_order(r[0], ord0, N) == GS_OK && memcmp(ord0, ord1, sizeof ord0) == 0);
    CHECK(gs_renderer_set_frames_in_flight(rl, 1) == GS_OK && gs_renderer_frames_in_flight(rl, &lanes_set, &lanes_active) == GS_OK && lanes_set == 1 && lanes_active == 0);
    CHECK(gs_renderer_destroy(rl) == GS_OK && gs_target_destroy(rtl[0]) == GS_OK && gs_target_destroy(rtl[1]) == GS_OK);
    uint32_t rows = 0, limit = 0; uint64_t cons = 0;
    CHECK(gs_renderer_sort_history(r[2], &rows, &limit, &cons) == GS_OK && limit == 2 && rows <= 2 && cons >= 2);
    CHECK(gs_renderer_sort_history(r[0], &rows, &limit, &cons) == GS_OK && limit == 128 && rows <= FRAMES && cons <= 1);      /* (download_order above consolidated once) */
    CHECK(gs_renderer_download_order(r[0], ord0, N) == GS_OK);   /* the reference's whole buffer: the recorded sorts carried out on all N */
    for (int k = 1; k < 3; ++k) { CHECK(gs_renderer_download_order(r[k], ord1, N) == GS_OK); CHECK(memcmp(ord0, ord1, sizeof ord0) == 0); }
    CHECK(gs_renderer_set_sort_mode(r[0], GS_SORT_FULL) == GS_OK && gs_renderer_download_order(r[0], ord1, N) == GS_OK && memcmp(ord0, ord1, sizeof ord0) == 0);
    for (int k = 2; k >= 0; --k) { CHECK(gs_renderer_destroy(r[k]) == GS_OK); CHECK(gs_target_destroy(rt[k]) == GS_OK); }
    CHECK(gs_asset_destroy(asset) == GS_OK);
    CHECK(gs_context_destroy(ctx2) == GS_OK && gs_context_destroy(ctx1) == GS_OK && gs_context_destroy(ctx) == GS_OK);
    printf("abi_smoke ok (context: GPU present; %d frames on three renderers, two of them sharing the asset across contexts, and on one with two frames in flight inside the library: same bits, same order)\n", FRAMES);
    return 0;
}
