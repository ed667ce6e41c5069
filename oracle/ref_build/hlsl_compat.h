// hlsl_compat.h -- just enough of the HLSL language surface, as C++17, to compile the TEXT of the reference's shaders
// (package/Shaders/GaussianSplatting.hlsl, SplatUtilities.compute, RenderGaussianSplats.shader, GaussianComposite.shader)
// on the host.  TEST INFRASTRUCTURE ONLY (oracle/_ref): never linked or loaded by the product.
//
// What is the reference's and what is ours:
//   * every statement of the shader functions is the reference's own text, read from /root/reference at build time by
//     gen_ref.py (which only applies the syntactic rewrites listed in its header: semantics, [numthreads], out/inout,
//     (T)0 casts, the `f` suffix on literals);
//   * this header supplies what HLSL leaves to the language / the GPU compiler: vector + matrix types with swizzles, the
//     resource types (ByteAddressBuffer, StructuredBuffer, Texture2D), and the INTRINSICS (mul, dot, lerp, normalize, rcp,
//     exp, f16tof32 ...).  HLSL does not fix the evaluation order or the contraction of an intrinsic, so there are two
//     builds that span that freedom:
//         REF_FUSED = 0 ("strict"):  every intrinsic is evaluated with separately rounded IEEE operations in source
//                                    order, x / c is a true division, exp is the correctly rounded e^x; compiled with
//                                    -ffp-contract=off.
//         REF_FUSED = 1 ("fused"):   what a GPU shader compiler does: dot / mul as a mad chain, lerp as one mad,
//                                    normalize = v * (1 / sqrt(dot)), exp(x) = exp2(x * log2 e) (DXC lowers exp to the
//                                    base-2 DXIL Exp), and the host compiler is allowed to contract the expressions of the
//                                    reference text (-ffp-contract=fast -freciprocal-math -mfma).
//     tests/test_ref_parity.py compares the oracle with both.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#ifndef REF_FUSED
#define REF_FUSED 0
#endif

namespace hlsl {

typedef unsigned int uint;
typedef float half;        // desktop D3D / DXC without -enable-16bit-types: half is float

template <class T, int N> struct vec;

// ---- swizzles: a view of some components of the parent's storage (a union member of the parent) ----
// (V = vec<T, sizeof...(I)> is a template TYPE argument so that the hidden-friend operators of V are found by ADL on a swizzle)
template <class V, class T, int N, int... I>
struct Swz {
    T d[N];
    static constexpr int M = (int)sizeof...(I);
    static constexpr int hlsl_size = M;
    T get(int k) const { const int idx[] = {I...}; return d[idx[k]]; }
    operator V() const { V r; const int idx[] = {I...}; for (int k = 0; k < M; ++k) r.d[k] = d[idx[k]]; return r; }
    Swz& operator=(const V& v) { const V c = v; const int idx[] = {I...}; for (int k = 0; k < M; ++k) d[idx[k]] = c.d[k]; return *this; }
    Swz& operator=(const Swz& o) { return *this = (V)o; }
    Swz& operator+=(const V& v) { return *this = (V)(*this) + v; }
    Swz& operator-=(const V& v) { return *this = (V)(*this) - v; }
    Swz& operator*=(const V& v) { return *this = (V)(*this) * v; }
    Swz& operator/=(const V& v) { return *this = (V)(*this) / v; }
};

#define HLSL_VEC_COMMON(N)                                                                                              \
    static constexpr int hlsl_size = N;                                                                                 \
    T get(int k) const { return d[k]; }                                                                                 \
    T& operator[](int i) { return d[i]; }                                                                               \
    const T& operator[](int i) const { return d[i]; }                                                                   \
    vec() { for (int k = 0; k < N; ++k) d[k] = T(); }                                                                   \
    vec(T s) { for (int k = 0; k < N; ++k) d[k] = s; }                                                                  \
    vec(const vec& o) { for (int k = 0; k < N; ++k) d[k] = o.d[k]; }                                                    \
    template <class U, std::enable_if_t<!std::is_same<U, T>::value, int> = 0>                                           \
    explicit vec(const vec<U, N>& o) { for (int k = 0; k < N; ++k) d[k] = (T)o.d[k]; }                                  \
    vec& operator=(const vec& o) { for (int k = 0; k < N; ++k) d[k] = o.d[k]; return *this; }                           \
    friend vec operator-(const vec& a) { vec r; for (int k = 0; k < N; ++k) r.d[k] = -a.d[k]; return r; }               \
    HLSL_VEC_BINOP(N, +) HLSL_VEC_BINOP(N, -) HLSL_VEC_BINOP(N, *) HLSL_VEC_BINOP(N, /)                                 \
    HLSL_VEC_CMP(N, <) HLSL_VEC_CMP(N, <=) HLSL_VEC_CMP(N, >) HLSL_VEC_CMP(N, >=) HLSL_VEC_CMP(N, ==) HLSL_VEC_CMP(N, !=)

#define HLSL_VEC_BINOP(N, OP)                                                                                           \
    friend vec operator OP(const vec& a, const vec& b) { vec r; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] OP b.d[k]; return r; } \
    friend vec operator OP(const vec& a, T b) { vec r; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] OP b; return r; }    \
    friend vec operator OP(T a, const vec& b) { vec r; for (int k = 0; k < N; ++k) r.d[k] = a OP b.d[k]; return r; }    \
    vec& operator OP##=(const vec& b) { for (int k = 0; k < N; ++k) d[k] = d[k] OP b.d[k]; return *this; }              \
    vec& operator OP##=(T b) { for (int k = 0; k < N; ++k) d[k] = d[k] OP b; return *this; }

#define HLSL_VEC_CMP(N, OP)                                                                                             \
    friend vec<bool, N> operator OP(const vec& a, const vec& b) { vec<bool, N> r; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] OP b.d[k]; return r; } \
    friend vec<bool, N> operator OP(const vec& a, T b) { vec<bool, N> r; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] OP b; return r; }

template <class T> struct vec<T, 2> {
    union {
        T d[2];
        struct { T x, y; };
        struct { T r, g; };
        Swz<vec<T, 2>, T, 2, 0, 1> xy; Swz<vec<T, 2>, T, 2, 1, 0> yx;
    };
    HLSL_VEC_COMMON(2)
    vec(T a, T b) { d[0] = a; d[1] = b; }
};

template <class T> struct vec<T, 3> {
    union {
        T d[3];
        struct { T x, y, z; };
        struct { T r, g, b; };
        Swz<vec<T, 2>, T, 3, 0, 1> xy;
        Swz<vec<T, 3>, T, 3, 0, 1, 2> xyz, rgb;
        Swz<vec<T, 3>, T, 3, 1, 2, 0> yzx; Swz<vec<T, 3>, T, 3, 2, 0, 1> zxy;
    };
    HLSL_VEC_COMMON(3)
    vec(T a, T b_, T c) { d[0] = a; d[1] = b_; d[2] = c; }
    // float3(float2-like, z), incl. the truncating int3(float2 swizzle, int) of GaussianComposite.shader:37
    template <class A, std::enable_if_t<A::hlsl_size == 2, int> = 0>
    vec(const A& a, T c) { d[0] = (T)a.get(0); d[1] = (T)a.get(1); d[2] = c; }
};

template <class T> struct vec<T, 4> {
    union {
        T d[4];
        struct { T x, y, z, w; };
        struct { T r, g, b, a; };
        Swz<vec<T, 2>, T, 4, 0, 1> xy; Swz<vec<T, 2>, T, 4, 2, 3> zw;
        Swz<vec<T, 3>, T, 4, 0, 1, 2> xyz, rgb;
        Swz<vec<T, 4>, T, 4, 0, 1, 2, 3> xyzw;
        Swz<vec<T, 4>, T, 4, 3, 0, 1, 2> wxyz; Swz<vec<T, 4>, T, 4, 0, 3, 1, 2> xwyz; Swz<vec<T, 4>, T, 4, 0, 1, 3, 2> xywz;
        Swz<vec<T, 4>, T, 4, 1, 2, 3, 0> yzwx; Swz<vec<T, 4>, T, 4, 0, 2, 3, 1> xzwy;
        Swz<vec<T, 4>, T, 4, 3, 3, 3, 3> wwww; Swz<vec<T, 4>, T, 4, 0, 1, 2, 0> xyzx; Swz<vec<T, 4>, T, 4, 3, 3, 3, 0> wwwx;
        Swz<vec<T, 4>, T, 4, 1, 2, 0, 1> yzxy; Swz<vec<T, 4>, T, 4, 2, 0, 1, 1> zxyy; Swz<vec<T, 4>, T, 4, 2, 0, 1, 2> zxyz;
        Swz<vec<T, 4>, T, 4, 1, 2, 0, 2> yzxz;
    };
    HLSL_VEC_COMMON(4)
    vec(T a_, T b_, T c, T e) { d[0] = a_; d[1] = b_; d[2] = c; d[3] = e; }
    template <class A, std::enable_if_t<A::hlsl_size == 3, int> = 0>
    vec(const A& v, T e) { d[0] = (T)v.get(0); d[1] = (T)v.get(1); d[2] = (T)v.get(2); d[3] = e; }
    template <class A, std::enable_if_t<A::hlsl_size == 2, int> = 0>
    vec(const A& v, T c, T e) { d[0] = (T)v.get(0); d[1] = (T)v.get(1); d[2] = c; d[3] = e; }
};

typedef vec<float, 2> float2; typedef vec<float, 3> float3; typedef vec<float, 4> float4;
typedef float2 half2; typedef float3 half3; typedef float4 half4;
typedef vec<uint, 2> uint2; typedef vec<uint, 3> uint3; typedef vec<uint, 4> uint4;
typedef vec<int, 2> int2; typedef vec<int, 3> int3; typedef vec<int, 4> int4;
typedef vec<bool, 2> bool2; typedef vec<bool, 3> bool3; typedef vec<bool, 4> bool4;

static_assert(sizeof(float2) == 8 && sizeof(float3) == 12 && sizeof(float4) == 16 && sizeof(uint2) == 8, "HLSL layout");

// ---- matrices (row-major storage, _mRC element names) ----
struct float4x4;
struct float3x3 {
    union {
        float m[3][3];
        struct { float _m00, _m01, _m02, _m10, _m11, _m12, _m20, _m21, _m22; };
    };
    float3x3() { std::memset(m, 0, sizeof(m)); }
    float3x3(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
        m[0][0] = a; m[0][1] = b; m[0][2] = c; m[1][0] = d; m[1][1] = e; m[1][2] = f; m[2][0] = g; m[2][1] = h; m[2][2] = i;
    }
    explicit float3x3(const float4x4& o);        // (float3x3)M: the upper-left 3x3
};
struct float4x4 {
    union {
        float m[4][4];
        struct { float _m00, _m01, _m02, _m03, _m10, _m11, _m12, _m13, _m20, _m21, _m22, _m23, _m30, _m31, _m32, _m33; };
    };
    float4x4() { std::memset(m, 0, sizeof(m)); }
    explicit float4x4(const float* rowMajor16) { std::memcpy(m, rowMajor16, sizeof(m)); }
};
inline float3x3::float3x3(const float4x4& o) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = o.m[i][j]; }

// =====================================================================================================================
// intrinsics
// =====================================================================================================================
inline float asfloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float asfloat(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint asuint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }

// f16tof32 / f32tof16: IEEE binary16, round-to-nearest-even, subnormals kept (D3D11 functional spec 3.2.2 / 22.13.1-2)
inline float f16tof32(uint h) {
    h &= 0xffffu;
    const uint sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, mant = h & 0x3ffu;
    if (e == 0) { const float v = std::ldexp((float)mant, -24); return sign ? -v : v; }        // zero / subnormal: exact
    if (e == 31) return asfloat(sign | 0x7f800000u | (mant << 13));
    return asfloat(sign | ((e + 112u) << 23) | (mant << 13));
}
inline uint f32tof16(float f) {
    const uint x = asuint(f), sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return sign | 0x7e00u;
    if (ax >= 0x477ff000u) return sign | 0x7c00u;                      // rounds to >= 65520: infinity
    if (ax < 0x33000000u) return sign;                                 // < 2^-25: zero (2^-25 itself ties to even = 0)
    if (ax < 0x38800000u) {                                            // subnormal half: value / 2^-24, RTNE
        const float q = std::nearbyint(std::ldexp(asfloat(ax), 24));   // default rounding mode = nearest even; exact scaling
        return sign | (uint)q;
    }
    uint r = ax - 0x38000000u;
    const uint rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return sign | r;
}

inline float sqrt(float x) { return std::sqrt(x); }
inline float abs(float x) { return std::fabs(x); }
inline float sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
inline float min(float a, float b) { return std::fmin(a, b); }
inline float max(float a, float b) { return std::fmax(a, b); }
inline float clamp(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }
inline float round(float x) { return std::nearbyint(x); }            // DXIL Round_ne
inline float rcp(float x) { return 1.0f / x; }
#if REF_FUSED
inline float exp(float x) { const float y = x * 1.44269504088896340736f; return (float)std::exp2((double)y); }
inline float lerp(float a, float b, float t) { return std::fmaf(t, b - a, a); }
inline float hl_dot2(float ax, float ay, float bx, float by) { return std::fmaf(ay, by, ax * bx); }
inline float hl_dot3(float ax, float ay, float az, float bx, float by, float bz) { return std::fmaf(az, bz, std::fmaf(ay, by, ax * bx)); }
inline float hl_dot4(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
    // the w term seeds the chain: with w = 1 (every mul(M, float4(p, 1)) of the path) this is m0*x + m1*y + m2*z + m3 as a mad chain
    return std::fmaf(az, bz, std::fmaf(ay, by, std::fmaf(ax, bx, aw * bw)));
}
#else
inline float exp(float x) { return (float)std::exp((double)x); }
inline float lerp(float a, float b, float t) { const float d = b - a; const float p = t * d; return a + p; }
inline float hl_dot2(float ax, float ay, float bx, float by) { const float p0 = ax * bx, p1 = ay * by; return p0 + p1; }
inline float hl_dot3(float ax, float ay, float az, float bx, float by, float bz) { const float p0 = ax * bx, p1 = ay * by, p2 = az * bz; const float s = p0 + p1; return s + p2; }
inline float hl_dot4(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
    const float p0 = ax * bx, p1 = ay * by, p2 = az * bz, p3 = aw * bw; const float s = p0 + p1; const float t = s + p2; return t + p3;
}
#endif

inline float dot(const float2& a, const float2& b) { return hl_dot2(a.x, a.y, b.x, b.y); }
inline float dot(const float3& a, const float3& b) { return hl_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
inline float dot(const float4& a, const float4& b) { return hl_dot4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w); }
inline float3 cross(const float3& a, const float3& b) {
    return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline bool all(const bool2& v) { return v.d[0] && v.d[1]; }
inline bool all(const bool3& v) { return v.d[0] && v.d[1] && v.d[2]; }
inline bool all(const bool4& v) { return v.d[0] && v.d[1] && v.d[2] && v.d[3]; }

#define HLSL_VEC_FUNCS(V, N)                                                                                            \
    inline V abs(const V& a) { V r; for (int k = 0; k < N; ++k) r.d[k] = abs(a.d[k]); return r; }                       \
    inline V min(const V& a, const V& b) { V r; for (int k = 0; k < N; ++k) r.d[k] = min(a.d[k], b.d[k]); return r; }   \
    inline V max(const V& a, const V& b) { V r; for (int k = 0; k < N; ++k) r.d[k] = max(a.d[k], b.d[k]); return r; }   \
    inline V saturate(const V& a) { V r; for (int k = 0; k < N; ++k) r.d[k] = saturate(a.d[k]); return r; }             \
    inline V lerp(const V& a, const V& b, const V& t) { V r; for (int k = 0; k < N; ++k) r.d[k] = lerp(a.d[k], b.d[k], t.d[k]); return r; } \
    inline V lerp(const V& a, const V& b, float t) { V r; for (int k = 0; k < N; ++k) r.d[k] = lerp(a.d[k], b.d[k], t); return r; } \
    inline float length(const V& a) { return sqrt(dot(a, a)); }                                                         \
    inline V normalize(const V& a) { return REF_FUSED ? a * (1.0f / sqrt(dot(a, a))) : a / sqrt(dot(a, a)); }
HLSL_VEC_FUNCS(float2, 2)
HLSL_VEC_FUNCS(float3, 3)
HLSL_VEC_FUNCS(float4, 4)

inline float3 mul(const float3x3& M, const float3& v) {
    float3 r;
    for (int i = 0; i < 3; ++i) r.d[i] = hl_dot3(M.m[i][0], M.m[i][1], M.m[i][2], v.x, v.y, v.z);
    return r;
}
inline float4 mul(const float4x4& M, const float4& v) {
    float4 r;
    for (int i = 0; i < 4; ++i) r.d[i] = hl_dot4(M.m[i][0], M.m[i][1], M.m[i][2], M.m[i][3], v.x, v.y, v.z, v.w);
    return r;
}
inline float3x3 mul(const float3x3& A, const float3x3& B) {
    float3x3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = hl_dot3(A.m[i][0], A.m[i][1], A.m[i][2], B.m[0][j], B.m[1][j], B.m[2][j]);
    return r;
}
inline float3x3 transpose(const float3x3& A) {
    float3x3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = A.m[j][i];
    return r;
}

// =====================================================================================================================
// resources.  Out-of-bounds reads return 0 (D3D robust buffer access).
// =====================================================================================================================
struct ByteAddressBuffer {
    const uint8_t* p = nullptr; uint64_t size = 0;
    uint Load(uint a) const { uint v = 0; if ((uint64_t)a + 4 <= size) std::memcpy(&v, p + a, 4); return v; }
    uint2 Load2(uint a) const { return uint2(Load(a), Load(a + 4)); }
    uint3 Load3(uint a) const { return uint3(Load(a), Load(a + 4), Load(a + 8)); }
    uint4 Load4(uint a) const { return uint4(Load(a), Load(a + 4), Load(a + 8), Load(a + 12)); }
};
typedef ByteAddressBuffer RWByteAddressBuffer;     // the path only reads its RW byte buffers

template <class T> struct StructuredBuffer {
    const uint8_t* p = nullptr; uint64_t count = 0;
    T operator[](uint i) const { T t{}; if (i < count) std::memcpy((void*)&t, p + (uint64_t)i * sizeof(T), sizeof(T)); return t; }
};
template <class T> struct RWStructuredBuffer {
    T* p = nullptr; uint64_t count = 0;
    T& operator[](uint i) { static thread_local T sink; return i < count ? p[i] : sink; }
};

// Texture2D.Load: the texture unit's format conversion (not reference text).  format: 0 Float32x4, 1 Float16x4, 2 Norm8x4
// (R8G8B8A8_UNorm: c / 255).  BC7 assets are handed over already block-decoded to RGBA8 by the test (that decoder is
// pinned against Pillow in tests/test_bc7.py).
struct Texture2D {
    const uint8_t* p = nullptr; uint32_t format = 0, width = 0, height = 0;
    float4 texel(uint x, uint y) const {
        if (x >= width || y >= height) return float4(0, 0, 0, 0);
        const uint64_t t = (uint64_t)y * width + x;
        if (format == 0) { float4 r; std::memcpy(r.d, p + t * 16, 16); return r; }
        if (format == 1) { uint lo, hi; std::memcpy(&lo, p + t * 8, 4); std::memcpy(&hi, p + t * 8 + 4, 4); return float4(f16tof32(lo), f16tof32(lo >> 16), f16tof32(hi), f16tof32(hi >> 16)); }
        uint e; std::memcpy(&e, p + t * 4, 4);
        const float c[4] = { (float)(e & 255u), (float)((e >> 8) & 255u), (float)((e >> 16) & 255u), (float)(e >> 24) };
#if REF_FUSED
        const float k = 1.0f / 255.0f;
        return float4(c[0] * k, c[1] * k, c[2] * k, c[3] * k);
#else
        return float4(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f, c[3] / 255.0f);
#endif
    }
    float4 Load(const uint3& c) const { return texel(c.x, c.y); }
    float4 Load(const int3& c) const { return (c.x < 0 || c.y < 0) ? float4(0, 0, 0, 0) : texel((uint)c.x, (uint)c.y); }
};

// `discard;` in a pixel shader
extern thread_local bool g_discarded;
#define discard do { ::hlsl::g_discarded = true; return half4(0, 0, 0, 0); } while (0)

template <class T> inline T hlsl_zero() { return T{}; }

}  // namespace hlsl
