// ref_compute.cpp -- harness around the reference's compute kernels (SplatUtilities.compute:37-252 + GaussianSplatting.hlsl),
// whose TEXT is included below from the scratch directory gen_ref.py filled.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
// The harness does what Unity's C# does for these kernels: bind the buffers (GaussianSplatRenderer.cs:447-525 SetAssetDataOnCS),
// set the uniforms (:579-639) and run one thread per id (DispatchCompute).
#include "hlsl_compat.h"
#include "../../include/gsplat_c.h"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace hlsl {
thread_local bool g_discarded = false;
namespace cs {
#define SHADER_STAGE_COMPUTE 1
#include "SplatUtilities_37_252.inc"
}  // namespace cs
}  // namespace hlsl

using namespace hlsl;
using namespace hlsl::cs;

static_assert(sizeof(SplatChunkInfo) == 64, "SplatChunkInfo is 64 B (GaussianSplatAsset.cs:231-240)");
static_assert(sizeof(SplatViewData) == 40, "SplatViewData is 40 B (GaussianSplatRenderer.cs:249 kGpuViewDataSize)");

namespace {
std::vector<GaussianCutoutShaderData> g_cutouts;
const uint32_t g_zero_word = 0;
}

extern "C" {

int32_t gsr_fused(void) { return REF_FUSED; }

// SetAssetDataOnCS (GaussianSplatRenderer.cs:447-465) + _SplatFormat / _SplatChunkCount (:502-504).  color_format 3 (BC7) is
// the texture unit's job: hand the texture over block-decoded as GS_COLOR_NORM8X4.
int32_t gsr_cs_bind_asset(const gs_asset_desc* d) {
    if (d->color_format > 2) return -1;
    _SplatPos.p = (const uint8_t*)d->pos_data; _SplatPos.size = d->pos_size;
    _SplatOther.p = (const uint8_t*)d->other_data; _SplatOther.size = d->other_size;
    _SplatSH.p = (const uint8_t*)d->sh_data; _SplatSH.size = d->sh_size;
    const uint32_t bpp = d->color_format == 0 ? 16u : (d->color_format == 1 ? 8u : 4u);
    _SplatColor.p = (const uint8_t*)d->color_data; _SplatColor.format = d->color_format;
    _SplatColor.width = 2048; _SplatColor.height = (uint32_t)(d->color_size / (2048ull * bpp));
    const uint64_t chunks = (d->chunk_data && d->chunk_size) ? d->chunk_size / 64 : 0;
    _SplatChunks.p = (const uint8_t*)d->chunk_data; _SplatChunks.count = chunks;
    _SplatChunkCount = (uint)chunks;
    _SplatFormat = d->pos_format | (d->scale_format << 8) | (d->sh_format << 16);
    _SplatCount = d->splat_count;
    return 0;
}

// CalcViewData's uniforms (GaussianSplatRenderer.cs:597-606), UNITY_MATRIX_VP / UNITY_MATRIX_P, cutouts (:742-764), deleted bits.
void gsr_cs_set_frame(const gs_frame_params* P, const gs_cutout* cutouts, uint32_t cutout_count, const uint32_t* deleted_bits, uint64_t deleted_bytes) {
    _MatrixMV = float4x4(P->matrix_mv);
    _MatrixObjectToWorld = float4x4(P->matrix_object_to_world);
    _MatrixWorldToObject = float4x4(P->matrix_world_to_object);
    unity_MatrixVP = float4x4(P->matrix_vp);
    glstate_matrix_projection = float4x4();
    glstate_matrix_projection._m00 = P->proj_m00;
    glstate_matrix_projection._m11 = P->proj_m11;
    _VecScreenParams = float4(P->screen_w, P->screen_h, 0, 0);
    _VecWorldSpaceCameraPos = float4(P->cam_pos_world[0], P->cam_pos_world[1], P->cam_pos_world[2], 0);
    _SplatScale = P->splat_scale;
    _SplatOpacityScale = P->opacity_scale;
    _SHOrder = P->sh_order;
    _SHOnly = P->sh_only;
    g_cutouts.assign(cutout_count, GaussianCutoutShaderData());
    for (uint32_t i = 0; i < cutout_count; ++i) { g_cutouts[i].mat = float4x4(cutouts[i].matrix); g_cutouts[i].typeAndFlags = cutouts[i].type_and_flags; }
    _SplatCutouts.p = (const uint8_t*)g_cutouts.data(); _SplatCutouts.count = cutout_count;
    _SplatCutoutsCount = cutout_count;
    _SplatBitsValid = deleted_bits ? 1u : 0u;
    _SplatDeletedBits.p = deleted_bits ? (const uint8_t*)deleted_bits : (const uint8_t*)&g_zero_word;
    _SplatDeletedBits.size = deleted_bits ? deleted_bytes : 4;
}

void gsr_cs_set_indices(uint32_t* order, uint32_t n) {
    _SplatSortKeys.p = order; _SplatSortKeys.count = n; _SplatCount = n;
    for (uint32_t i = 0; i < n; ++i) CSSetIndices(uint3(i, 0, 0));
}

// SortPoints (GaussianSplatRenderer.cs:612-633): _MatrixMV = worldToCam' * model, one thread per sorted position
void gsr_cs_calc_distances(uint32_t* order, const float* matrix_sort, uint32_t* keys, uint32_t n) {
    _MatrixMV = float4x4(matrix_sort);
    _SplatSortKeys.p = order; _SplatSortKeys.count = n;
    _SplatSortDistances.p = keys; _SplatSortDistances.count = n;
    _SplatCount = n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) CSCalcDistances(uint3((uint)i, 0, 0));
}

void gsr_cs_calc_view(void* view_out, uint32_t n) {
    _SplatViewData.p = (SplatViewData*)view_out; _SplatViewData.count = n;
    _SplatCount = n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) CSCalcViewData(uint3((uint)i, 0, 0));
}

// LoadSplatData(idx) in the layout of the oracle's gso_decode_splat: pos3, rot4, scale3, opacity, col3, sh 15x3
void gsr_cs_decode_all(float* out, uint32_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const SplatData s = LoadSplatData((uint)i);
        float* o = out + i * 59;
        *o++ = s.pos.x; *o++ = s.pos.y; *o++ = s.pos.z;
        *o++ = s.rot.x; *o++ = s.rot.y; *o++ = s.rot.z; *o++ = s.rot.w;
        *o++ = s.scale.x; *o++ = s.scale.y; *o++ = s.scale.z;
        *o++ = s.opacity;
        *o++ = s.sh.col.x; *o++ = s.sh.col.y; *o++ = s.sh.col.z;
        const half3* sh = &s.sh.sh1;
        for (int k = 0; k < 15; ++k) { *o++ = sh[k].x; *o++ = sh[k].y; *o++ = sh[k].z; }
    }
}

void gsr_cs_pixel_index(uint32_t idx, uint32_t* xy) { const uint3 c = SplatIndexToPixelIndex(idx); xy[0] = c.x; xy[1] = c.y; }
uint32_t gsr_cs_sortable_uint(float f) { return FloatToSortableUint(f); }
int32_t gsr_cs_is_splat_cut(const float* pos) { return IsSplatCut(float3(pos[0], pos[1], pos[2])) ? 1 : 0; }

// the encoder-side helpers GaussianSplatting.hlsl carries (used by the editor kernels; handy cross-checks of the codec tests)
void gsr_cs_pack_smallest3(const float* q, float* out4) { const float4 r = PackSmallest3Rotation(float4(q[0], q[1], q[2], q[3])); for (int k = 0; k < 4; ++k) out4[k] = r.d[k]; }
uint32_t gsr_cs_encode_quat_norm10(const float* v) { return EncodeQuatToNorm10(float4(v[0], v[1], v[2], v[3])); }
void gsr_cs_decode_rotation(uint32_t enc, float* out4) { const float4 r = DecodeRotation(DecodePacked_10_10_10_2(enc)); for (int k = 0; k < 4; ++k) out4[k] = r.d[k]; }
uint32_t gsr_cs_encode_morton(uint32_t x, uint32_t y) { return EncodeMorton2D_16x16(uint2(x, y)); }

}  // extern "C"

// intermediate stages of CSCalcViewData's covariance path for one splat (debugging aid of tests/test_ref_parity.py):
// out = CalcMatrixFromRotationScale (9), cov3d0/1 * splatScale^2 (6), CalcCovariance2D (3), DecomposeCovariance (4)
extern "C" void gsr_cs_cov_stages(uint32_t idx, float* out22) {
    const SplatData splat = LoadSplatData(idx);
    const float3x3 m = CalcMatrixFromRotationScale(splat.rot, splat.scale);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out22[i * 3 + j] = m.m[i][j];
    float3 cov3d0, cov3d1;
    CalcCovariance3D(m, cov3d0, cov3d1);
    const float splatScale2 = _SplatScale * _SplatScale;
    cov3d0 *= splatScale2;
    cov3d1 *= splatScale2;
    for (int k = 0; k < 3; ++k) { out22[9 + k] = cov3d0.d[k]; out22[12 + k] = cov3d1.d[k]; }
    const float3 cov2d = CalcCovariance2D(splat.pos, cov3d0, cov3d1, _MatrixMV, UNITY_MATRIX_P, _VecScreenParams);
    for (int k = 0; k < 3; ++k) out22[15 + k] = cov2d.d[k];
    float2 v1, v2;
    DecomposeCovariance(cov2d, v1, v2);
    out22[18] = v1.x; out22[19] = v1.y; out22[20] = v2.x; out22[21] = v2.y;
}
