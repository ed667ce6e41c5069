// ref_render.cpp -- harness around the reference's splat vertex + fragment shader (RenderGaussianSplats.shader:14-109, TEXT
// included below from gen_ref.py's scratch directory).  TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// The shader text is the reference's.  What the GPU's fixed-function units do around it is restated HERE, as the D3D11
// rules Unity runs it under, and is therefore our reading, not the reference's code:
//   * DrawProcedural(6 indices {0,1,2, 1,3,2}, N instances) in instance order (GaussianSplatRenderer.cs:156-166,410-412);
//   * a vertex with a NaN position discards the primitive (the shader's own comment, :44);
//   * all four vertices of an instance share z and w, so depth clipping keeps or drops the whole quad; expressed through the
//     view depth w against the camera's [near, far] (exactly the clip volume of a perspective projection);
//   * viewport: px = (x/w * 0.5 + 0.5) * W, py = (0.5 - 0.5 * y/w) * H, pixel centres at +0.5, rows top-down (D3D);
//   * the two triangles of the quad tile a parallelogram; the interpolated TEXCOORD0 of a pixel is the affine function of
//     its centre that takes the vertex values (evaluated in float64, no sub-pixel snapping);
//   * output merger: Blend OneMinusDstAlpha One (:11) into R16G16B16A16_SFloat (GaussianSplatRenderer.cs:194), each blend
//     the exactly computed src * (1 - dst.a) + dst rounded once to nearest-even half.
#include "hlsl_compat.h"
#include "../../include/gsplat_c.h"

namespace hlsl {
namespace rs {
float4 _ScreenParams;        // UnityShaderVariables.cginc (auto-included by Unity in every CGPROGRAM)
#include "RenderGaussianSplats.inc"
}  // namespace rs
}  // namespace hlsl

using namespace hlsl;
using namespace hlsl::rs;

namespace {
inline uint16_t f64tof16(double d) {                 // round-to-nearest-even, via the float path only when exact
    // src*(1-A)+dst is formed in double (53 bits >> the 11 of a half): round once to half
    uint64_t u; std::memcpy(&u, &d, 8);
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    const int ebits = (int)((u >> 52) & 0x7ffu);
    const uint64_t mant = u & ((1ull << 52) - 1ull);
    if (ebits == 0x7ff) return (uint16_t)(sign | (mant ? 0x7e00u : 0x7c00u));
    if (ebits == 0) return sign;
    const int e = ebits - 1023;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);
    const uint64_t sig = mant | (1ull << 52);
    const int shift = 42 + (e < -14 ? (-14 - e) : 0);
    if (shift >= 64) return sign;
    uint64_t q = sig >> shift;
    const uint64_t rem = sig & ((1ull << shift) - 1ull), half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (q & 1ull))) q++;
    const uint32_t h = (e < -14) ? (uint32_t)q : (uint32_t)(((uint32_t)(e + 14) << 10) + q);
    return (uint16_t)(sign | (h >= 0x7c00u ? 0x7c00u : h));
}
struct Quad { double p0x, p0y, e1x, e1y, e2x, e2y, det; float q0x, q0y, q1x, q1y, q2x, q2y; half4 col; int x0, x1, y0, y1; bool live; };
}

extern "C" {

void gsr_rs_bind(const void* view, const uint32_t* order, uint32_t n, float W, float H) {
    _SplatViewData.p = (const uint8_t*)view; _SplatViewData.count = n;
    _OrderBuffer.p = (const uint8_t*)order; _OrderBuffer.count = n;
    _ScreenParams = float4(W, H, 1.0f + 1.0f / W, 1.0f + 1.0f / H);
    _SplatBitsValid = 0;                                                   // no selection (editor feature)
    _CameraTargetTexture_TexelSize = float4(1.0f / W, 1.0f / H, W, H);     // rendering into a render texture, not the backbuffer
}

// vert(vtxID, instID): out = col rgba, pos xy, vertex xyzw
void gsr_rs_vert(uint32_t vtxID, uint32_t instID, float* out10) {
    const v2f o = vert(vtxID, instID);
    for (int k = 0; k < 4; ++k) out10[k] = o.col.d[k];
    out10[4] = o.pos.x; out10[5] = o.pos.y;
    for (int k = 0; k < 4; ++k) out10[6 + k] = o.vertex.d[k];
}

// frag(i): returns 1 when the fragment is discarded
int32_t gsr_rs_frag(const float* pos2, const float* col4, float* out4) {
    v2f i;
    i.pos = float2(pos2[0], pos2[1]);
    i.col = half4(col4[0], col4[1], col4[2], col4[3]);
    g_discarded = false;
    const half4 r = frag(i);
    for (int k = 0; k < 4; ++k) out4[k] = r.d[k];
    return g_discarded ? 1 : 0;
}

// DrawProcedural of all n instances into rt (W x H x 4 halfs, D3D rows top-down; not cleared here).
void gsr_rs_draw(uint16_t* rt, uint32_t W, uint32_t H, float near_clip, float far_clip) {
    const uint32_t n = (uint32_t)_OrderBuffer.count;
    std::vector<Quad> quads(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Quad& Q = quads[i]; Q.live = false;
        const v2f v0 = vert(0, (uint)i), v1 = vert(1, (uint)i), v2 = vert(2, (uint)i), v3 = vert(3, (uint)i);
        bool finite = true;
        const v2f* vs[4] = {&v0, &v1, &v2, &v3};
        for (int k = 0; k < 4; ++k) for (int c = 0; c < 4; ++c) finite = finite && std::isfinite(vs[k]->vertex.d[c]);
        if (!finite) continue;                                             // NaN vertex: primitive discarded
        const double w = v0.vertex.w;
        if (!(w > 0.0) || !(w >= near_clip && w <= far_clip)) continue;    // depth clip of the whole quad
        auto sx = [&](const v2f& v) { return ((double)v.vertex.x / w * 0.5 + 0.5) * W; };
        auto sy = [&](const v2f& v) { return (0.5 - 0.5 * ((double)v.vertex.y / w)) * H; };
        Q.p0x = sx(v0); Q.p0y = sy(v0);
        Q.e1x = sx(v1) - Q.p0x; Q.e1y = sy(v1) - Q.p0y;
        Q.e2x = sx(v2) - Q.p0x; Q.e2y = sy(v2) - Q.p0y;
        Q.det = Q.e1x * Q.e2y - Q.e1y * Q.e2x;
        if (!(Q.det != 0.0) || !std::isfinite(Q.det)) continue;            // zero-area quad: no pixel
        Q.q0x = v0.pos.x; Q.q0y = v0.pos.y; Q.q1x = v1.pos.x; Q.q1y = v1.pos.y; Q.q2x = v2.pos.x; Q.q2y = v2.pos.y;
        Q.col = v0.col;
        const double xs[4] = {Q.p0x, Q.p0x + Q.e1x, Q.p0x + Q.e2x, sx(v3)}, ys[4] = {Q.p0y, Q.p0y + Q.e1y, Q.p0y + Q.e2y, sy(v3)};
        double xmin = xs[0], xmax = xs[0], ymin = ys[0], ymax = ys[0];
        for (int k = 1; k < 4; ++k) { xmin = std::fmin(xmin, xs[k]); xmax = std::fmax(xmax, xs[k]); ymin = std::fmin(ymin, ys[k]); ymax = std::fmax(ymax, ys[k]); }
        const double fx0 = std::fmax(std::floor(xmin - 0.5), 0.0), fx1 = std::fmin(std::ceil(xmax - 0.5), (double)W - 1.0);
        const double fy0 = std::fmax(std::floor(ymin - 0.5), 0.0), fy1 = std::fmin(std::ceil(ymax - 0.5), (double)H - 1.0);
        if (!(fx0 <= fx1 && fy0 <= fy1)) continue;
        Q.x0 = (int)fx0; Q.x1 = (int)fx1; Q.y0 = (int)fy0; Q.y1 = (int)fy1;
        Q.live = true;
    }
    const int bands = (int)((H + 7) / 8);
#pragma omp parallel for schedule(dynamic, 1)
    for (int band = 0; band < bands; ++band) {
        const int by0 = band * 8, by1 = std::min((int)H - 1, by0 + 7);
        for (uint32_t i = 0; i < n; ++i) {
            const Quad& Q = quads[i];
            if (!Q.live || Q.y1 < by0 || Q.y0 > by1) continue;
            for (int y = std::max(Q.y0, by0); y <= std::min(Q.y1, by1); ++y)
                for (int x = Q.x0; x <= Q.x1; ++x) {
                    const double dx = (x + 0.5) - Q.p0x, dy = (y + 0.5) - Q.p0y;
                    const double s = (dx * Q.e2y - dy * Q.e2x) / Q.det, t = (Q.e1x * dy - Q.e1y * dx) / Q.det;
                    if (!(s >= 0.0 && s <= 1.0 && t >= 0.0 && t <= 1.0)) continue;
                    v2f in;
                    in.col = Q.col;
                    in.pos = float2((float)(Q.q0x + s * ((double)Q.q1x - Q.q0x) + t * ((double)Q.q2x - Q.q0x)),
                                    (float)(Q.q0y + s * ((double)Q.q1y - Q.q0y) + t * ((double)Q.q2y - Q.q0y)));
                    g_discarded = false;
                    const half4 src = frag(in);
                    if (g_discarded) continue;
                    uint16_t* px = rt + ((size_t)y * W + x) * 4;
                    const double oneMinusDstA = 1.0 - (double)f16tof32(px[3]);
                    for (int c = 0; c < 4; ++c) px[c] = f64tof16((double)src.d[c] * oneMinusDstA + (double)f16tof32(px[c]));
                }
        }
    }
}

}  // extern "C"
