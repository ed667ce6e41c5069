/* Plain-C consumer of include/gsplat_c.h: links libgsplat_hip.so like a non-C++ host would and walks the parts of the ABI
 * that need no GPU (version, error strings, argument validation, the native importer).  Built and run by tests/test_abi.py. */
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

    /* with a GPU the asset could now be handed to gs_asset_create; without one the context must fail loudly */
    gs_context* ctx = NULL;
    const int32_t rc = gs_context_create(0, NULL, &ctx);
    CHECK(rc == GS_OK || rc == GS_ERR_NO_DEVICE);
    if (rc == GS_OK) CHECK(gs_context_destroy(ctx) == GS_OK);
    printf("abi_smoke ok (context: %s)\n", rc == GS_OK ? "GPU present" : "no device, as reported");
    return 0;
}
