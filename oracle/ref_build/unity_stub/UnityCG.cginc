// Stand-in for Unity's built-in shader library (UnityCG.cginc / UnityShaderVariables.cginc, Unity 2022.3.47f1): NOT part of
// /root/reference, so what the reference's shaders use from it is restated here (SURVEY.md section 8c "arithmetic that
// lives outside /root/reference"):
//   UNITY_MATRIX_VP / UNITY_MATRIX_P  -> the per-camera matrices Unity binds as unity_MatrixVP / glstate_matrix_projection
//   _ScreenParams                     -> (width, height, 1 + 1/width, 1 + 1/height) of the current target
//   GammaToLinearSpace                -> Unity's cubic approximation of the sRGB decode
#ifndef UNITY_CG_STUB_INCLUDED
#define UNITY_CG_STUB_INCLUDED
float4x4 unity_MatrixVP;
float4x4 glstate_matrix_projection;
float4 _ScreenParams;
#define UNITY_MATRIX_VP unity_MatrixVP
#define UNITY_MATRIX_P glstate_matrix_projection
inline half3 GammaToLinearSpace(half3 sRGB)
{
    return sRGB * (sRGB * (sRGB * 0.305306011f + 0.682171111f) + 0.012522878f);
}
#endif
