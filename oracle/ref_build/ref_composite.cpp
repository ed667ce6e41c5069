// ref_composite.cpp -- harness around the reference's composite pass (GaussianComposite.shader:14-41, TEXT included below from
// gen_ref.py's scratch directory; GammaToLinearSpace comes from the Unity stand-in, unity_stub/UnityCG.cginc).
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Fixed function restated here: one full-screen triangle = one fragment per pixel at
// SV_Position = pixel centre; Blend SrcAlpha OneMinusSrcAlpha (:11) of all four channels over the camera target (`bg`); a
// blend factor of exactly 0 contributes 0 whatever the source holds (D3D11 functional spec, output merger: 0 * x = 0 also
// for NaN / INF) -- that is what turns the frag's 0/0 of an untouched pixel into "background".
#include "hlsl_compat.h"

namespace hlsl {
namespace comp {
#include "GaussianComposite.inc"
}  // namespace comp
}  // namespace hlsl

using namespace hlsl;
using namespace hlsl::comp;

extern "C" {

void gsr_comp_vert(uint32_t vtxID, float* out4) { const v2f o = vert(vtxID); for (int k = 0; k < 4; ++k) out4[k] = o.vertex.d[k]; }

void gsr_comp_resolve(const uint16_t* rt, uint32_t W, uint32_t H, const float* bg, float* out32f) {
    _GaussianSplatRT.p = (const uint8_t*)rt; _GaussianSplatRT.format = 1; _GaussianSplatRT.width = W; _GaussianSplatRT.height = H;
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            v2f i;
            i.vertex = float4((float)x + 0.5f, (float)y + 0.5f, 1.0f, 1.0f);
            const half4 src = frag(i);
            float* o = out32f + ((size_t)y * W + x) * 4;
            const float sa = src.a, da = 1.0f - src.a;
            for (int c = 0; c < 4; ++c) {
                const float s = (sa == 0.0f) ? 0.0f : src.d[c] * sa;
                const float d = (da == 0.0f) ? 0.0f : bg[c] * da;
                o[c] = s + d;
            }
        }
}

}  // extern "C"
