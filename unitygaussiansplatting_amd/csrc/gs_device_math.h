// gs_device_math.h -- per-splat arithmetic of the gfx950 kernels (decode, sort key, view data).
//
// Everything here is plain IEEE fp32 with explicit fmaf() and is compiled with -ffp-contract=off, so the
// numbers are a function of the source only ("canonical arithmetic").  The functions are
// __host__ __device__ so that tests/test_host_math.py can compile this header with g++ and compare it with
// the oracle bit for bit on the CPU box; the shipped library only ever calls them from device code.
//
// Reference lines restated (paths relative to /root/reference/package/Shaders/):
//   GaussianSplatting.hlsl :5-11 InvSquareCentered01, :29-53 CalcMatrixFromRotationScale/CalcCovariance3D,
//   :56-90 CalcCovariance2D, :113-127,183-194 Morton texel address, :130-179 ShadeSH, :219-229 DecodeRotation,
//   :261-300 DecodePacked_*, :325-421 LoadUShort/LoadUInt/LoadAndDecodeVector/LoadSplatPos, :428-608 LoadSplatData
//   SplatUtilities.compute :52-57 FloatToSortableUint, :107-162 DecomposeCovariance, :189-252 CSCalcViewData
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#define GS_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define GS_HD inline
#endif

#ifndef GS_CULL_EXTENT
#define GS_CULL_EXTENT 2.8285f     // 2 sqrt(2), rounded up: see the early cull in CalcViewGeom
#endif

namespace gsm {

struct AssetView {              // device (or, in the host test, host) pointers to the five blobs
    const uint8_t* pos;
    const uint8_t* other;
    const uint8_t* color;
    const uint8_t* sh;
    const uint8_t* chunk;
    uint32_t n, posFmt, scaleFmt, colorFmt, shFmt, chunkCount;
};

struct FrameConsts {            // gs_frame_params flattened for kernel argument passing
    float mv[12];               // rows 0..2 of _MatrixMV
    float o2w[12];              // rows 0..2 of _MatrixObjectToWorld
    float w2o[12];              // rows 0..2 of _MatrixWorldToObject (3x3 part used)
    float vp[16];               // UNITY_MATRIX_VP
    float limX, limY, focal;    // 1.3 tanFovX, 1.3 tanFovY, W P00 / 2 of CalcCovariance2D: per-frame constants, see FrameConstsFromProjection
    float cullKx, cullKy, cullMx, cullMy;   // whole-chunk frustum cull (ChunkOutside), see FrameConstsChunkCull
    uint32_t cullOn;
    float screenW, screenH;
    float camx, camy, camz;
    float splatScale, opacityScale;
    uint32_t shOrder, shOnly;
    float nearClip, farClip;
};

// Edit state CSCalcViewData consults (SplatUtilities.compute:91-105): _SplatDeletedBits / _SplatBitsValid and _SplatCutouts
struct EditView {
    const uint32_t* deletedBits;    // ceil(n/32) words, or null (_SplatBitsValid = 0)
    const uint32_t* cutouts;        // cutoutCount x 17 dwords: float4x4 (rows of 4) + typeAndFlags
    uint32_t cutoutCount;
};

// The camera-only part of CalcCovariance2D (GaussianSplatting.hlsl:62-72), evaluated once per frame on the host with the
// same fp32 operations the shader performs per splat (aspect = P00/P11; tanFovX = 1/P00; tanFovY = 1/(P11 aspect) -- which is
// 1/P00 again, a quirk of the reference that is kept; focal = W P00 / 2): three IEEE divisions less per splat.
GS_HD void FrameConstsFromProjection(FrameConsts& c, float p00, float p11, float screenW) {
    const float aspect = p00 / p11;
    const float tanFovX = 1.0f / p00;
    const float tanFovY = 1.0f / (p11 * aspect);
    c.limX = 1.3f * tanFovX;
    c.limY = 1.3f * tanFovY;
    c.focal = screenW * p00 / 2.0f;
}

struct ViewData { float pos[4]; float axis1[2]; float axis2[2]; uint32_t color[2]; };   // 40 B SplatViewData

// ---- bit casts / half ---------------------------------------------------------------------------
GS_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
GS_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
GS_HD float f16tof32(uint32_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __half2float(__ushort_as_half((unsigned short)(h & 0xffffu)));
#else
    h &= 0xffffu;
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    return u2f(x);
#endif
}
GS_HD uint32_t f32tof16(float f) {     // round to nearest even
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__half_as_ushort(__float2half_rn(f));
#else
    uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return sign | 0x7e00u;
    if (x >= 0x477ff000u) return sign | 0x7c00u;
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return sign;
        const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u, shift = 126u - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return sign | r;
    }
    uint32_t r = x - 0x38000000u;
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return sign | r;
#endif
}

// ---- scalar helpers -----------------------------------------------------------------------------
GS_HD float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }
GS_HD float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
GS_HD bool finite32(float x) { return fabsf(x) <= 3.4028234663852886e38f; }
GS_HD float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
GS_HD float dot3f(float ax, float ay, float az, float bx, float by, float bz) { return fmaf(az, bz, fmaf(ay, by, ax * bx)); }
GS_HD float dot2f(float ax, float ay, float bx, float by) { return fmaf(ay, by, ax * bx); }
// row r of mul(M, float4(v,1)) for a matrix stored as rows of 4
GS_HD float mrow(const float* m, int r, float x, float y, float z) {
    return fmaf(m[r * 4 + 2], z, fmaf(m[r * 4 + 1], y, fmaf(m[r * 4 + 0], x, m[r * 4 + 3])));
}
GS_HD float mrow3(const float* m, int r, float x, float y, float z) {
    return fmaf(m[r * 4 + 2], z, fmaf(m[r * 4 + 1], y, m[r * 4 + 0] * x));
}

GS_HD uint32_t FloatToSortableUint(float f) {
    const uint32_t fu = f2u(f);
    const uint32_t mask = (uint32_t)(-(int32_t)(fu >> 31)) | 0x80000000u;
    return fu ^ mask;
}

// ---- raw loads (2-byte aligned addresses, stitched from aligned dwords like the HLSL) -------------
GS_HD uint32_t ld32a(const uint8_t* p, uint64_t a) { return *(const uint32_t*)(p + a); }            // a % 4 == 0
GS_HD uint32_t LoadUInt(const uint8_t* p, uint64_t a) {
    const uint64_t aa = a & ~(uint64_t)3;
    uint32_t v = ld32a(p, aa);
    if (a != aa) { const uint32_t v1 = ld32a(p, aa + 4); v = (v >> 16) | (v1 << 16); }
    return v;
}
GS_HD uint32_t LoadUShort(const uint8_t* p, uint64_t a) {
    const uint64_t aa = a & ~(uint64_t)3;
    uint32_t v = ld32a(p, aa);
    if (a != aa) v >>= 16;
    return v & 0xffffu;
}

#define GS_R63 (1.0f / 63.0f)
#define GS_R31 (1.0f / 31.0f)
#define GS_R2047 (1.0f / 2047.0f)
#define GS_R1023 (1.0f / 1023.0f)
#define GS_R65535 (1.0f / 65535.0f)
#define GS_R255 (1.0f / 255.0f)

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
// (r, g) of a colour as a 2-vector: on the device its fp32 multiplies / fmas are ONE packed instruction (v_pk_mul_f32 /
// v_pk_fma_f32 run two IEEE operations per issue slot on gfx950) -- same results as the scalar forms, element by element.
#if defined(__clang__)
typedef float F2 __attribute__((ext_vector_type(2)));
#else       // a host compiler without vector extensions (the header is also compiled by g++ in tests): the same arithmetic, element by element
struct F2 { float x, y; };
inline F2 operator*(F2 a, F2 b) { return F2{ a.x * b.x, a.y * b.y }; }
inline F2 operator-(F2 a, F2 b) { return F2{ a.x - b.x, a.y - b.y }; }
inline F2 operator-(F2 a) { return F2{ -a.x, -a.y }; }
inline F2& operator+=(F2& a, F2 b) { a.x += b.x; a.y += b.y; return a; }
#endif
struct RGB { F2 rg; float b; };
GS_HD F2 fma2(F2 a, F2 b, F2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return F2{ fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y) };
#endif
}

GS_HD V3 Dec_6_5_5(uint32_t e) { return { (float)(e & 63) * GS_R63, (float)((e >> 6) & 31) * GS_R31, (float)((e >> 11) & 31) * GS_R31 }; }
GS_HD V3 Dec_5_6_5(uint32_t e) { return { (float)(e & 31) * GS_R31, (float)((e >> 5) & 63) * GS_R63, (float)((e >> 11) & 31) * GS_R31 }; }
GS_HD V3 Dec_11_10_11(uint32_t e) { return { (float)(e & 2047) * GS_R2047, (float)((e >> 11) & 1023) * GS_R1023, (float)((e >> 21) & 2047) * GS_R2047 }; }
GS_HD V3 Dec_16_16_16(uint32_t e0, uint32_t e1) { return { (float)(e0 & 65535) * GS_R65535, (float)(e0 >> 16) * GS_R65535, (float)(e1 & 65535) * GS_R65535 }; }

GS_HD uint32_t vecStride(uint32_t fmt) { return fmt == 0 ? 12u : (fmt == 1 ? 6u : (fmt == 2 ? 4u : 2u)); }

GS_HD V3 LoadVec(const uint8_t* buf, uint64_t a, uint32_t fmt) {
    if (fmt == 0) return { u2f(LoadUInt(buf, a)), u2f(LoadUInt(buf, a + 4)), u2f(LoadUInt(buf, a + 8)) };
    if (fmt == 1) return Dec_16_16_16(LoadUInt(buf, a), LoadUShort(buf, a + 4));
    if (fmt == 2) return Dec_11_10_11(LoadUInt(buf, a));
    return Dec_6_5_5(LoadUShort(buf, a));
}

// The same values as LoadVec for a compile-time format, with every load unconditional (a load inside a branch is waited for
// at the end of the branch, which serialises a streaming kernel's loads): unaligned 6- and 2-byte records take the two aligned
// dwords around them and select (the blobs carry >= 4 readable bytes after the last record).
template <int FMT> struct RawVec { uint32_t d[FMT == 0 ? 3 : (FMT == 1 ? 2 : 1)]; };
template <int FMT> GS_HD RawVec<FMT> LoadRawT(const uint8_t* buf, uint64_t a) {            // the loads alone (so that a caller can issue many before it decodes)
    RawVec<FMT> r;
    if (FMT == 0) { r.d[0] = ld32a(buf, a); r.d[FMT == 0 ? 1 : 0] = ld32a(buf, a + 4); r.d[FMT == 0 ? 2 : 0] = ld32a(buf, a + 8); return r; }
    const uint64_t aa = a & ~(uint64_t)3;
    r.d[0] = ld32a(buf, aa);
    if (FMT == 1) r.d[FMT == 1 ? 1 : 0] = ld32a(buf, aa + 4);
    return r;
}
template <int FMT> GS_HD V3 DecodeRawT(const RawVec<FMT>& r, uint64_t a) {
    if (FMT == 0) return { u2f(r.d[0]), u2f(r.d[FMT == 0 ? 1 : 0]), u2f(r.d[FMT == 0 ? 2 : 0]) };
    if (FMT == 2) return Dec_11_10_11(r.d[0]);
    const bool odd = (a & 2u) != 0;
    if (FMT == 3) return Dec_6_5_5(odd ? (r.d[0] >> 16) : (r.d[0] & 0xffffu));
    const uint32_t d0 = r.d[0], d1 = r.d[FMT == 1 ? 1 : 0];
    return Dec_16_16_16(odd ? ((d0 >> 16) | (d1 << 16)) : d0, odd ? (d1 >> 16) : (d1 & 0xffffu));
}
template <int FMT> GS_HD V3 LoadVecT(const uint8_t* buf, uint64_t a) { return DecodeRawT<FMT>(LoadRawT<FMT>(buf, a), a); }
template <int FMT> GS_HD uint32_t vecStrideT() { return FMT == 0 ? 12u : (FMT == 1 ? 6u : (FMT == 2 ? 4u : 2u)); }
// chunk de-normalisation of a decoded position (LoadSplatPos, GaussianSplatting.hlsl:394-421)
GS_HD V3 ChunkLerpPos(const AssetView& a, V3 p, uint32_t ci) {
    if (ci < a.chunkCount) {
        const uint8_t* c = a.chunk + (uint64_t)ci * 64;
        p.x = lerpf(u2f(ld32a(c, 16)), u2f(ld32a(c, 20)), p.x);
        p.y = lerpf(u2f(ld32a(c, 24)), u2f(ld32a(c, 28)), p.y);
        p.z = lerpf(u2f(ld32a(c, 32)), u2f(ld32a(c, 36)), p.z);
    }
    return p;
}
template <int FMT> GS_HD V3 LoadSplatPosChunkT(const AssetView& a, uint32_t idx, uint32_t ci) {
    return ChunkLerpPos(a, LoadVecT<FMT>(a.pos, (uint64_t)idx * vecStrideT<FMT>()), ci);
}

struct ChunkRaw { uint32_t w[16]; };    // colR,colG,colB,colA, posX(2),posY(2),posZ(2), sclX..Z, shR..B
GS_HD ChunkRaw LoadChunk(const uint8_t* chunk, uint32_t ci) {
    ChunkRaw c;
    const uint32_t* p = (const uint32_t*)(chunk + (uint64_t)ci * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) c.w[k] = p[k];     // uniform per 256-splat block: becomes one scalar s_load_dwordx16
    return c;
}

// LoadSplatPos (GaussianSplatting.hlsl:394-421); ci = idx >> 8, passed separately so that a caller whose workgroup is aligned
// to the 256-splat chunks can hand in a wave-uniform value (the ChunkInfo loads then become scalar loads)
GS_HD V3 LoadSplatPosChunk(const AssetView& a, uint32_t idx, uint32_t ci) {
    V3 p = LoadVec(a.pos, (uint64_t)idx * vecStride(a.posFmt), a.posFmt);
    if (ci < a.chunkCount) {
        const uint8_t* c = a.chunk + (uint64_t)ci * 64;
        p.x = lerpf(u2f(ld32a(c, 16)), u2f(ld32a(c, 20)), p.x);
        p.y = lerpf(u2f(ld32a(c, 24)), u2f(ld32a(c, 28)), p.y);
        p.z = lerpf(u2f(ld32a(c, 32)), u2f(ld32a(c, 36)), p.z);
    }
    return p;
}
GS_HD V3 LoadSplatPos(const AssetView& a, uint32_t idx) { return LoadSplatPosChunk(a, idx, idx >> 8); }

// CSCalcDistances body (SplatUtilities.compute:76-81): key of the splat `origIdx` under sort-matrix row 2
GS_HD uint32_t SortKeyOf(const V3& p, float m20, float m21, float m22, float m23) {
    const float z = fmaf(m22, p.z, fmaf(m21, p.y, fmaf(m20, p.x, m23)));
    return FloatToSortableUint(z);
}
GS_HD uint32_t SortKey(const AssetView& a, uint32_t origIdx, float m20, float m21, float m22, float m23) {
    return SortKeyOf(LoadSplatPos(a, origIdx), m20, m21, m22, m23);
}

GS_HD void SplatIndexToPixelIndex(uint32_t idx, uint32_t& x, uint32_t& y) {
    uint32_t t = idx;
    t = (t & 0xFF) | ((t & 0xFE) << 7);
    t &= 0x5555;
    t = (t ^ (t >> 1)) & 0x3333;
    t = (t ^ (t >> 2)) & 0x0f0f;
    const uint32_t tile = idx >> 8;
    x = (tile & 127u) * 16 + (t & 0xF);
    y = (tile >> 7) * 16 + (t >> 8);
}

GS_HD float InvSquareCentered01(float x) {
    x -= 0.5f;
    x *= 0.5f;
    x = sqrtf(fabsf(x)) * sgn(x);
    return x + 0.5f;
}

GS_HD V4 DecodeRotation(uint32_t enc) {
    const float px = (float)(enc & 1023) * GS_R1023, py = (float)((enc >> 10) & 1023) * GS_R1023, pz = (float)((enc >> 20) & 1023) * GS_R1023;
    const uint32_t idx = enc >> 30;
    const float SQRT2 = 1.41421356237f, INV_SQRT2 = 0.70710678118f;
    const float qx = fmaf(px, SQRT2, -INV_SQRT2), qy = fmaf(py, SQRT2, -INV_SQRT2), qz = fmaf(pz, SQRT2, -INV_SQRT2);
    const float qw = sqrtf(1.0f - sat(dot3f(qx, qy, qz, qx, qy, qz)));
    V4 q = { qx, qy, qz, qw };
    if (idx == 0) q = { qw, qx, qy, qz };
    if (idx == 1) q = { qx, qw, qy, qz };
    if (idx == 2) q = { qx, qy, qw, qz };
    return q;
}

// SH coefficient k (1..15) of the splat whose SH record starts at `sp`, before chunk de-normalisation
GS_HD V3 LoadSH(const uint8_t* sp, uint32_t shFormat, int k) {
    if (shFormat == 0) return { u2f(ld32a(sp, (k - 1) * 12)), u2f(ld32a(sp, (k - 1) * 12 + 4)), u2f(ld32a(sp, (k - 1) * 12 + 8)) };
    if (shFormat == 2) return Dec_11_10_11(ld32a(sp, (k - 1) * 4));
    if (shFormat == 3) return Dec_5_6_5(LoadUShort(sp, (uint64_t)(k - 1) * 2));
    // fp16 (Float16 and Cluster* tables)
    return { f16tof32(LoadUShort(sp, (uint64_t)(k - 1) * 6)), f16tof32(LoadUShort(sp, (uint64_t)(k - 1) * 6 + 2)), f16tof32(LoadUShort(sp, (uint64_t)(k - 1) * 6 + 4)) };
}

// the same coefficient as (rg, b); the 5.6.5 decode's multiplies are packed
GS_HD RGB LoadSHrgb(const uint8_t* sp, uint32_t shFormat, int k) {
    if (shFormat == 3) {
        const uint32_t e = LoadUShort(sp, (uint64_t)(k - 1) * 2);
        return { F2{ (float)(e & 31), (float)((e >> 5) & 63) } * F2{ GS_R31, GS_R63 }, (float)((e >> 11) & 31) * GS_R31 };
    }
    const V3 v = LoadSH(sp, shFormat, k);
    return { F2{ v.x, v.y }, v.z };
}

GS_HD uint32_t shStrideOf(uint32_t shFormat) { return shFormat == 0 ? 192u : (shFormat == 2 ? 60u : (shFormat == 3 ? 32u : 96u)); }

#define GS_SH_C1 0.4886025f

// Where the raw (still chunk-normalised) SH coefficients of a splat come from: straight from the asset blob here; the
// view kernel substitutes a reader of its LDS-staged copy (gs_view.hip).  begin() receives the record's address.
struct SHFromBlob {
    const uint8_t* sp; uint32_t fmt;
    GS_HD void begin(const uint8_t* p, uint32_t f) { sp = p; fmt = f; }
    GS_HD V3 load(int k) const { return LoadSH(sp, fmt, k); }
    GS_HD RGB load_rgb(int k) const { return LoadSHrgb(sp, fmt, k); }
};

// Whole-chunk cull (per-frame path only).  A chunk's 256 splats lie in the box [posMin, posMax] of its ChunkInfo and are no
// larger than its scale maximum, so if all 8 corners of the box are on the outer side of one frustum plane -- pushed out
// by a bound on the footprint radius -- no splat of the chunk can be drawn and its workgroup stops before decoding anything.
// Footprint half extent <= 2 sqrt(2) sqrt(2 lambda1), lambda1 <= (|T0|^2 + |T1|^2) 1.1 s^2 smax^2 + 0.6 (as in CalcViewGeom's early cull),
// |T0|^2 + |T1|^2 <= (focal / w)^2 G with G = 2 (|mv0|^2 + |mv1|^2 + (limX^2 + limY^2) |mv2|^2)  =>  half extent <=
// K focal smax / w + 3.1 px with K = 2 sqrt(2) sqrt(2.2 G s^2).  "Right of the screen" (cx - extent > W) becomes the LINEAR test
// x - w (1 + 13.2 / W) - (2 K focal / W) smax > 0 on clip coordinates (slack: K is taken 5 % larger, 6.6 px instead of 3.1);
// likewise left / top / bottom; the depth planes are the rasteriser's own centre-depth rule.  Needs clip.w = |view z|
// (a perspective projection whose last row is (0, 0, -1, 0)); the host checks that and switches the cull off otherwise.
GS_HD void FrameConstsChunkCull(FrameConsts& c) {
    const float g0 = dot3f(c.mv[0], c.mv[1], c.mv[2], c.mv[0], c.mv[1], c.mv[2]), g1 = dot3f(c.mv[4], c.mv[5], c.mv[6], c.mv[4], c.mv[5], c.mv[6]);
    const float g2 = dot3f(c.mv[8], c.mv[9], c.mv[10], c.mv[8], c.mv[9], c.mv[10]);
    const float G = 2.0f * (g0 + g1 + (c.limX * c.limX + c.limY * c.limY) * g2);
    const float K = (GS_CULL_EXTENT * 1.05f) * sqrtf(2.2f * G * (c.splatScale * c.splatScale));
    c.cullKx = 2.0f * K * c.focal / c.screenW;
    c.cullKy = 2.0f * K * c.focal / c.screenH;
    c.cullMx = 1.0f + 13.2f / c.screenW;
    c.cullMy = 1.0f + 13.2f / c.screenH;
    // clip.w of a point must equal -(mv row 2).point: compare row 3 of vp * o2w with -row 2 of mv
    bool ok = true;
    for (int k = 0; k < 4; ++k) {
        float r = 0.0f;
        for (int j = 0; j < 3; ++j) r += c.vp[12 + j] * c.o2w[j * 4 + k];
        if (k == 3) r += c.vp[15];
        const float want = -c.mv[8 + k];
        if (!(fabsf(r - want) <= 1.0e-4f * (fabsf(want) + 1.0f))) ok = false;
    }
    if (!(K > 0.0f) || !(K < 3.0e30f)) ok = false;
    c.cullOn = ok ? 1u : 0u;
}

// corner: 0..7 (bit 0 = x max, bit 1 = y max, bit 2 = z max).  planes[p] = this corner is on the outer side of plane p
// (0 right, 1 left, 2 above-or-below +y, 3 the other y side, 4 nearer than near, 5 beyond far).
GS_HD uint32_t ChunkCornerOutside(const AssetView& a, const FrameConsts& P, uint32_t chunkIdx, uint32_t corner) {
    const uint8_t* ck = a.chunk + (uint64_t)chunkIdx * 64;
    const float px = u2f(ld32a(ck, 16 + ((corner & 1u) ? 4 : 0))), py = u2f(ld32a(ck, 24 + ((corner & 2u) ? 4 : 0))), pz = u2f(ld32a(ck, 32 + ((corner & 4u) ? 4 : 0)));
    float sm = fmaxf(fmaxf(f16tof32(ld32a(ck, 40) >> 16), f16tof32(ld32a(ck, 44) >> 16)), f16tof32(ld32a(ck, 48) >> 16));
    sm *= sm; sm *= sm; sm *= sm;                                   // the asset stores scale^(1/8)
    const float wx = mrow(P.o2w, 0, px, py, pz), wy = mrow(P.o2w, 1, px, py, pz), wz = mrow(P.o2w, 2, px, py, pz);
    const float x = mrow(P.vp, 0, wx, wy, wz), y = mrow(P.vp, 1, wx, wy, wz), w = mrow(P.vp, 3, wx, wy, wz);
    const float cxs = P.cullKx * sm, cys = P.cullKy * sm;
    uint32_t m = 0;
    if (x - w * P.cullMx - cxs > 0.0f) m |= 1u;
    if (-x - w * P.cullMx - cxs > 0.0f) m |= 2u;
    if (y - w * P.cullMy - cys > 0.0f) m |= 4u;
    if (-y - w * P.cullMy - cys > 0.0f) m |= 8u;
    if (w < P.nearClip) m |= 16u;
    if (w > P.farClip) m |= 32u;
    return m;
}
GS_HD bool ChunkOutside(const AssetView& a, const FrameConsts& P, uint32_t chunkIdx) {       // scalar form (host tests)
    if (!P.cullOn || chunkIdx >= a.chunkCount) return false;
    uint32_t all = 63u;
    for (uint32_t c = 0; c < 8; ++c) all &= ChunkCornerOutside(a, P, chunkIdx, c);
    return all != 0u;
}


// ---- BC7 (BPTC) block decode: ColorFormat.BC7 = GraphicsFormat.RGBA_BC7_UNorm (GaussianSplatAsset.cs:56,169), the colour
// texture of the VeryLow preset (GaussianSplatAssetCreator.cs:198,893-910), sampled by _SplatColor.Load in LoadSplatData.
// The texture unit decodes the 16-byte block of the 4x4 texels around the splat's texel; here: one block load + the decode
// of that ONE texel.  Format: Khronos Data Format Specification, "BPTC"; the partition / anchor tables are those of
// unitygaussiansplatting_amd/bc7.py, extracted from and cross-checked against an independent decoder (Pillow) by
// tests/test_bc7.py.  Returns r | g << 8 | b << 16 | a << 24.
GS_HD uint32_t bc7_bits(uint64_t lo, uint64_t hi, uint32_t pos, uint32_t n) {      // n <= 8 bits at bit `pos` of the 128-bit block
    uint64_t v;
    if (pos >= 64u) v = hi >> (pos - 64u);
    else v = (lo >> pos) | (pos ? (hi << (64u - pos)) : 0ull);
    return (uint32_t)v & ((1u << n) - 1u);
}
GS_HD uint32_t DecodeBC7Texel(const uint8_t* block, uint32_t texel) {
        static constexpr uint16_t kP2[64] = { 0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80, 0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000, 0xf710, 0x8e, 0x7100, 0x8ce, 0x8c, 0x7310, 0x3100, 0x8cce, 0x88c, 0x3110, 0x6666, 0x366c, 0x17e8, 0xff0, 0x718e, 0x399c, 0xaaaa, 0xf0f0, 0x5a5a, 0x33cc, 0x3c3c, 0x55aa, 0x9696, 0xa55a, 0x73ce, 0x13c8, 0x324c, 0x3bdc, 0x6996, 0xc33c, 0x9966, 0x660, 0x272, 0x4e4, 0x4e40, 0x2720, 0xc936, 0x936c, 0x39c6, 0x639c, 0x9336, 0x9cc6, 0x817e, 0xe718, 0xccf0, 0xfcc, 0x7744, 0xee22 };
        static constexpr uint32_t kP3[64] = { 0xaa685050, 0x6a5a5040, 0x5a5a4200, 0x5450a0a8, 0xa5a50000, 0xa0a05050, 0x5555a0a0, 0x5a5a5050, 0xaa550000, 0xaa555500, 0xaaaa5500, 0x90909090, 0x94949494, 0xa4a4a4a4, 0xa9a59450, 0x2a0a4250, 0xa5945040, 0xa425054, 0xa5a5a500, 0x55a0a0a0, 0xa8a85454, 0x6a6a4040, 0xa4a45000, 0x1a1a0500, 0x50a4a4, 0xaaa59090, 0x14696914, 0x69691400, 0xa08585a0, 0xaa821414, 0x50a4a450, 0x6a5a0200, 0xa9a58000, 0x5090a0a8, 0xa8a09050, 0x24242424, 0xaa5500, 0x24924924, 0x24499224, 0x50a50a50, 0x500aa550, 0xaaaa4444, 0x66660000, 0xa5a0a5a0, 0x50a050a0, 0x69286928, 0x44aaaa44, 0x66666600, 0xaa444444, 0x54a854a8, 0x95809580, 0x96969600, 0xa85454a8, 0x80959580, 0xaa141414, 0x96960000, 0xaaaa1414, 0xa05050a0, 0xa0a5a5a0, 0x96000000, 0x40804080, 0xa9a8a9a8, 0xaaaaaa44, 0x2a4a5254 };
        static constexpr uint8_t kA2[64] = { 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2, 15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6, 6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15 };
        static constexpr uint8_t kA3a[64] = { 3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3, 3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15, 8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15, 3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3 };
        static constexpr uint8_t kA3b[64] = { 15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8, 15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8, 15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8, 15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8 };
        // per mode: subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, endpoint p-bits, shared p-bits,
        // index bits, secondary index bits -- packed 4 bits each, lowest nibble first
        static constexpr uint64_t kMode[8] = { 0x301040043ull, 0x310060062ull, 0x200050063ull, 0x201070062ull, 0x3200651201ull, 0x2200870201ull, 0x401770001ull, 0x201550062ull };
        static constexpr uint8_t kW2[4] = { 0, 21, 43, 64 };
        static constexpr uint8_t kW3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
        static constexpr uint8_t kW4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
    const uint64_t lo = (uint64_t)ld32a(block, 0) | ((uint64_t)ld32a(block, 4) << 32), hi = (uint64_t)ld32a(block, 8) | ((uint64_t)ld32a(block, 12) << 32);
    const uint32_t m8 = (uint32_t)lo & 255u;
    if (m8 == 0u) return 0u;                                              // reserved mode: the block decodes to zero
    uint32_t mode = 0;
    while (!((m8 >> mode) & 1u)) ++mode;
    const uint64_t mi = kMode[mode];
    const uint32_t ns = (uint32_t)(mi & 15), pb = (uint32_t)(mi >> 4) & 15, rb = (uint32_t)(mi >> 8) & 15, isb = (uint32_t)(mi >> 12) & 15,
                   cb = (uint32_t)(mi >> 16) & 15, ab = (uint32_t)(mi >> 20) & 15, epb = (uint32_t)(mi >> 24) & 15, spb = (uint32_t)(mi >> 28) & 15,
                   ib = (uint32_t)(mi >> 32) & 15, ib2 = (uint32_t)(mi >> 36) & 15;
    uint32_t pos = mode + 1u;
    const uint32_t shape = bc7_bits(lo, hi, pos, pb); pos += pb;
    const uint32_t rot = bc7_bits(lo, hi, pos, rb); pos += rb;
    const uint32_t isel = bc7_bits(lo, hi, pos, isb); pos += isb;
    uint32_t s = 0, an1 = 16u, an2 = 16u;                                 // subset of the texel, anchors of subsets 1 and 2
    if (ns == 2u) { s = (kP2[shape] >> texel) & 1u; an1 = kA2[shape]; }
    else if (ns == 3u) { s = (kP3[shape] >> (2u * texel)) & 3u; an1 = kA3a[shape]; an2 = kA3b[shape]; }
    const uint32_t ne = 2u * ns, e0 = 2u * s, e1 = e0 + 1u;
    uint32_t c0[4], c1[4];
#pragma unroll
    for (uint32_t ch = 0; ch < 3u; ++ch) {
        c0[ch] = bc7_bits(lo, hi, pos + (ch * ne + e0) * cb, cb);
        c1[ch] = bc7_bits(lo, hi, pos + (ch * ne + e1) * cb, cb);
    }
    pos += 3u * ne * cb;
    c0[3] = 255u; c1[3] = 255u;
    if (ab) { c0[3] = bc7_bits(lo, hi, pos + e0 * ab, ab); c1[3] = bc7_bits(lo, hi, pos + e1 * ab, ab); pos += ne * ab; }
    uint32_t cbits = cb, abits = ab;
    if (epb) {
        const uint32_t p0 = bc7_bits(lo, hi, pos + e0, 1), p1 = bc7_bits(lo, hi, pos + e1, 1);
        pos += ne;
#pragma unroll
        for (uint32_t ch = 0; ch < 3u; ++ch) { c0[ch] = (c0[ch] << 1) | p0; c1[ch] = (c1[ch] << 1) | p1; }
        if (ab) { c0[3] = (c0[3] << 1) | p0; c1[3] = (c1[3] << 1) | p1; abits++; }
        cbits++;
    } else if (spb) {
        const uint32_t p = bc7_bits(lo, hi, pos + s, 1);
        pos += ns;
#pragma unroll
        for (uint32_t ch = 0; ch < 3u; ++ch) { c0[ch] = (c0[ch] << 1) | p; c1[ch] = (c1[ch] << 1) | p; }
        cbits++;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < 3u; ++ch) {                                 // left-align to 8 bits, replicate the top bits
        c0[ch] <<= 8u - cbits; c0[ch] |= c0[ch] >> cbits;
        c1[ch] <<= 8u - cbits; c1[ch] |= c1[ch] >> cbits;
    }
    if (ab) { c0[3] <<= 8u - abits; c0[3] |= c0[3] >> abits; c1[3] <<= 8u - abits; c1[3] |= c1[3] >> abits; }
    // index of this texel: ib bits per texel, one bit less for each subset's anchor texel
    const uint32_t before = (texel > 0u ? 1u : 0u) + (texel > an1 ? 1u : 0u) + (texel > an2 ? 1u : 0u);
    const bool isAnchor = texel == 0u || texel == an1 || texel == an2;
    uint32_t i1 = bc7_bits(lo, hi, pos + texel * ib - before, ib - (isAnchor ? 1u : 0u));
    pos += 16u * ib - ns;
    uint32_t ci = i1, cbt = ib, ai = i1, abt = ib;
    if (ib2) {
        const uint32_t i2 = bc7_bits(lo, hi, pos + texel * ib2 - (texel > 0u ? 1u : 0u), ib2 - (texel == 0u ? 1u : 0u));
        ai = i2; abt = ib2;
        if (isel) { ci = i2; cbt = ib2; ai = i1; abt = ib; }
    }
    const uint32_t wc = cbt == 2u ? kW2[ci] : (cbt == 3u ? kW3[ci] : kW4[ci]);
    const uint32_t wa = abt == 2u ? kW2[ai] : (abt == 3u ? kW3[ai] : kW4[ai]);
    uint32_t px[4];
#pragma unroll
    for (uint32_t ch = 0; ch < 3u; ++ch) px[ch] = ((64u - wc) * c0[ch] + wc * c1[ch] + 32u) >> 6;
    px[3] = ((64u - wa) * c0[3] + wa * c1[3] + 32u) >> 6;
    if (rot == 1u) { const uint32_t t = px[3]; px[3] = px[0]; px[0] = t; }
    else if (rot == 2u) { const uint32_t t = px[3]; px[3] = px[1]; px[1] = t; }
    else if (rot == 3u) { const uint32_t t = px[3]; px[3] = px[2]; px[2] = t; }
    return px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
}

// _SplatColor.Load(SplatIndexToPixelIndex(idx)) (GaussianSplatting.hlsl:456-458): the texel as the texture unit returns it,
// before the chunk de-normalisation
GS_HD V4 LoadColorTexel(const AssetView& a, uint32_t idx) {
    uint32_t tx, ty;
    SplatIndexToPixelIndex(idx, tx, ty);
    const uint64_t texel = (uint64_t)ty * 2048 + tx;
    if (a.colorFmt == 0) {
        const uint8_t* c = a.color + texel * 16;
        return { u2f(ld32a(c, 0)), u2f(ld32a(c, 4)), u2f(ld32a(c, 8)), u2f(ld32a(c, 12)) };
    }
    if (a.colorFmt == 1) {
        const uint32_t lo = ld32a(a.color, texel * 8), hi = ld32a(a.color, texel * 8 + 4);
        return { f16tof32(lo), f16tof32(lo >> 16), f16tof32(hi), f16tof32(hi >> 16) };
    }
    uint32_t e;
    if (a.colorFmt == 3) e = DecodeBC7Texel(a.color + ((uint64_t)(ty >> 2) * 512 + (tx >> 2)) * 16, (ty & 3u) * 4u + (tx & 3u));   // RGBA_BC7_UNorm, 512 blocks per row
    else e = ld32a(a.color, texel * 4);
    return { (float)(e & 255) * GS_R255, (float)((e >> 8) & 255) * GS_R255, (float)((e >> 16) & 255) * GS_R255, (float)(e >> 24) * GS_R255 };
}

// splat.sh.col of LoadSplatData: the de-normalised DC colour (GaussianSplatting.hlsl:456-458,590-596), what the debug point
// shader shows (GaussianDebugRenderPoints.shader:48)
GS_HD V3 LoadSplatBaseColor(const AssetView& a, uint32_t idx) {
    V4 col = LoadColorTexel(a, idx);
    const uint32_t ci = idx >> 8;
    if (ci < a.chunkCount) {
        const uint8_t* c = a.chunk + (uint64_t)ci * 64;
        col.x = lerpf(f16tof32(ld32a(c, 0)), f16tof32(ld32a(c, 0) >> 16), col.x);
        col.y = lerpf(f16tof32(ld32a(c, 4)), f16tof32(ld32a(c, 4) >> 16), col.y);
        col.z = lerpf(f16tof32(ld32a(c, 8)), f16tof32(ld32a(c, 8) >> 16), col.z);
    }
    return { col.x, col.y, col.z };
}
// GaussianDebugRenderPoints.shader:49-54 (_DisplayIndex): a colour that encodes the splat's index
GS_HD V3 DebugIndexColor(uint32_t idx, uint32_t count) {
    const float f = (float)idx / (float)count;
    const float r = f * 100.0f, g = f * 10.0f;
    return { r - floorf(r), g - floorf(g), f };
}

// ---- debug box modes (GaussianDebugRenderBoxes.shader): every box is an affine image of the cube [-1,1]^3,
// world = c + B l.  Per pixel the rasteriser's result is restated as a ray / box intersection in the box's own space
// (l = Binv (p - c)): the ray through the pixel centre is p(t) = o + t d with d scaled so that t IS the view depth, the slab
// test gives its entry / exit parameters, and the face the rasteriser draws is the entry face for a box whose transform
// keeps the winding (the cube's triangles are wound so that its outside faces are the back faces: with `Cull Front` the
// faces turned towards the camera are drawn) and the exit face for a mirrored one.  Depth clipping and the depth test use t.
struct BoxRec { float inv[9]; float lo[3]; float r, g, b, a; };      // 64 B: Binv rows, Binv (o - c), colour; a < 0 marks a mirrored box (its alpha is |a|)

// inverse of a 3x3 (rows of 3) by cofactors; returns the determinant.  A singular / non-finite box is not drawn.
GS_HD float Inverse3(const float* b, float* inv) {
    const float c00 = fmaf(b[4], b[8], -(b[5] * b[7])), c01 = fmaf(b[5], b[6], -(b[3] * b[8])), c02 = fmaf(b[3], b[7], -(b[4] * b[6]));
    const float det = fmaf(b[2], c02, fmaf(b[1], c01, b[0] * c00));
    const float r = 1.0f / det;
    inv[0] = c00 * r; inv[1] = fmaf(b[2], b[7], -(b[1] * b[8])) * r; inv[2] = fmaf(b[1], b[5], -(b[2] * b[4])) * r;
    inv[3] = c01 * r; inv[4] = fmaf(b[0], b[8], -(b[2] * b[6])) * r; inv[5] = fmaf(b[2], b[3], -(b[0] * b[5])) * r;
    inv[6] = c02 * r; inv[7] = fmaf(b[1], b[6], -(b[0] * b[7])) * r; inv[8] = fmaf(b[0], b[4], -(b[1] * b[3])) * r;
    return det;
}

// per-frame ray set-up from the frame constants: R0 = VP row 0 / P00, R1 = VP row 1 / P11 (the camera's right / up axes in
// world space), R2 = VP row 3 (its forward axis: clip.w = view depth).  World direction of the ray through pixel centre
// (px + 0.5, py + 0.5): d = ax R0 + ay R1 + R2, ax = ndc.x / P00, ay = ndc.y / P11 -- so that view depth along the ray = t.
struct RayConsts { float r0[3], r1[3], r2[3]; float ox, oy, oz; float p00, p11, W, H; };
GS_HD void RayConstsFromFrame(RayConsts& rc, const float* vp, float p00, float p11, float camx, float camy, float camz, float W, float H) {
    for (int k = 0; k < 3; ++k) { rc.r0[k] = vp[k] / p00; rc.r1[k] = vp[4 + k] / p11; rc.r2[k] = vp[12 + k]; }
    rc.ox = camx; rc.oy = camy; rc.oz = camz; rc.p00 = p00; rc.p11 = p11; rc.W = W; rc.H = H;
}
GS_HD void PixelRay(const RayConsts& rc, int px, int py, float d[3]) {
    const float ndcx = (((float)px + 0.5f) / rc.W) * 2.0f - 1.0f;
    const float ndcy = 1.0f - (((float)py + 0.5f) / rc.H) * 2.0f;
    const float ax = ndcx / rc.p00, ay = ndcy / rc.p11;
    for (int k = 0; k < 3; ++k) d[k] = fmaf(ay, rc.r1[k], fmaf(ax, rc.r0[k], rc.r2[k]));
}
// view depth of the drawn face of box `inv/lo` along the ray with box-space direction ld, or a negative value if the pixel is
// not covered.  NaNs (a ray parallel to a slab it starts on) compare false: not covered.
GS_HD float BoxFaceDepth(const float* lo, const float* ld, bool mirrored) {
    float tmin = -3.4028234663852886e38f, tmax = 3.4028234663852886e38f;
    for (int k = 0; k < 3; ++k) {
        const float t1 = (-1.0f - lo[k]) / ld[k], t2 = (1.0f - lo[k]) / ld[k];
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    if (!(tmin <= tmax)) return -1.0f;
    const float t = mirrored ? tmax : tmin;
    return (t > 0.0f) ? t : -1.0f;
}

// LoadSplatData's rotation, scale (de-normalised) and opacity (GaussianSplatting.hlsl:428-470,565-603) for the box mode
GS_HD void LoadSplatRotScaleOpacity(const AssetView& a, uint32_t idx, V4& q, V3& scale, float& opacity) {
    uint32_t otherStride = 4 + vecStride(a.scaleFmt);
    if (a.shFmt > 3) otherStride += 2;
    const uint64_t otherAddr = (uint64_t)idx * otherStride;
    q = DecodeRotation(LoadUInt(a.other, otherAddr));
    scale = LoadVec(a.other, otherAddr + 4, a.scaleFmt);
    V4 col = LoadColorTexel(a, idx);
    const uint32_t ci = idx >> 8;
    if (ci < a.chunkCount) {
        const uint8_t* c = a.chunk + (uint64_t)ci * 64;
        scale.x = lerpf(f16tof32(ld32a(c, 40)), f16tof32(ld32a(c, 40) >> 16), scale.x);
        scale.y = lerpf(f16tof32(ld32a(c, 44)), f16tof32(ld32a(c, 44) >> 16), scale.y);
        scale.z = lerpf(f16tof32(ld32a(c, 48)), f16tof32(ld32a(c, 48) >> 16), scale.z);
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
        col.w = InvSquareCentered01(lerpf(f16tof32(ld32a(c, 12)), f16tof32(ld32a(c, 12) >> 16), col.w));
    }
    opacity = col.w;
}

// Builds the record of one box from its centre c and matrix B (world = c + B l, rows of 3); returns false if it cannot be drawn.
// Also the conservative pixel bounding box of its projection (the whole screen if a corner is not in front of the camera).
GS_HD bool BuildBox(const float c[3], const float B[9], const RayConsts& rc, const float* vp, float r, float g, float bl, float al,
                    BoxRec& rec, int& x0, int& x1, int& y0, int& y1) {
    const float det = Inverse3(B, rec.inv);
    bool ok = finite32(det) && det != 0.0f;
    for (int k = 0; k < 9; ++k) ok = ok && finite32(rec.inv[k]);
    if (!ok || !(al > 0.0f)) return false;
    const float dx = rc.ox - c[0], dy = rc.oy - c[1], dz = rc.oz - c[2];
    for (int k = 0; k < 3; ++k) rec.lo[k] = fmaf(rec.inv[k * 3 + 2], dz, fmaf(rec.inv[k * 3 + 1], dy, rec.inv[k * 3] * dx));
    rec.r = r; rec.g = g; rec.b = bl; rec.a = det < 0.0f ? -al : al;
    float minx = 3.0e38f, maxx = -3.0e38f, miny = 3.0e38f, maxy = -3.0e38f;
    bool behind = false;
    for (int corner = 0; corner < 8; ++corner) {
        const float lx = (corner & 1) ? 1.0f : -1.0f, ly = (corner & 2) ? 1.0f : -1.0f, lz = (corner & 4) ? 1.0f : -1.0f;
        const float wx = c[0] + fmaf(B[2], lz, fmaf(B[1], ly, B[0] * lx)), wy = c[1] + fmaf(B[5], lz, fmaf(B[4], ly, B[3] * lx)),
                    wz = c[2] + fmaf(B[8], lz, fmaf(B[7], ly, B[6] * lx));
        const float cx = mrow(vp, 0, wx, wy, wz), cy = mrow(vp, 1, wx, wy, wz), cw = mrow(vp, 3, wx, wy, wz);
        if (!(cw > 1.0e-6f)) { behind = true; continue; }
        const float sx = fmaf(0.5f * (cx / cw), rc.W, 0.5f * rc.W), sy = fmaf(-0.5f * (cy / cw), rc.H, 0.5f * rc.H);
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx); miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    if (behind || !(finite32(minx) && finite32(maxx) && finite32(miny) && finite32(maxy))) { x0 = 0; y0 = 0; x1 = (int)rc.W - 1; y1 = (int)rc.H - 1; return true; }
    const float fx0 = fmaxf(floorf(minx) - 1.0f, 0.0f), fx1 = fminf(ceilf(maxx) + 1.0f, rc.W - 1.0f);
    const float fy0 = fmaxf(floorf(miny) - 1.0f, 0.0f), fy1 = fminf(ceilf(maxy) + 1.0f, rc.H - 1.0f);
    if (!(fx0 <= fx1 && fy0 <= fy1)) return false;
    x0 = (int)fx0; x1 = (int)fx1; y0 = (int)fy0; y1 = (int)fy1;
    return true;
}

// IsSplatCut (SplatUtilities.compute:164-187); pos is the object-space position
GS_HD bool IsSplatCut(const EditView& e, float px, float py, float pz) {
    bool finalCut = false;
    for (uint32_t i = 0; i < e.cutoutCount; ++i) {
        const uint32_t* c = e.cutouts + i * 17u;                   // wave-uniform address: scalar loads
        const uint32_t tf = c[16];
        const uint32_t type = tf & 0xFFu;
        if (type == 0xFFu) continue;                               // invalid/null cutout, ignore
        const bool invert = (tf & 0xFF00u) != 0;
        float m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = u2f(c[k]);
        const float cx = mrow(m, 0, px, py, pz), cy = mrow(m, 1, px, py, pz), cz = mrow(m, 2, px, py, pz);
        if (type == 0u) { if (dot3f(cx, cy, cz, cx, cy, cz) <= 1.0f) return invert; }
        if (type == 1u) { if (fabsf(cx) <= 1.0f && fabsf(cy) <= 1.0f && fabsf(cz) <= 1.0f) return invert; }
        finalCut = finalCut || !invert;
    }
    return finalCut;
}

// CSCalcViewData for one splat (SplatUtilities.compute:189-252), in two halves so that a caller which only needs the
// splats that reach the screen can stop after the geometry:
//   CalcViewGeom   LoadSplatData minus SH, clip position, deleted bit / cutouts, 3D -> 2D covariance, axes, opacity
//   CalcViewColor  view direction, ShadeSH, colour packing
// CalcViewDataT = both, i.e. the reference's kernel; the arithmetic is the same whichever way it is called.
struct ViewPartial {
    ViewData view;              // pos, axis1/2 final after CalcViewGeom; color[1] low half = f16 opacity
    V4 col;                     // de-normalised colour (col.w = opacity before _SplatOpacityScale)
    V3 shMin, shMax;
    float wx, wy, wz;           // centerWorldPos
    uint64_t otherEnd;          // byte address one past the splat's `other` record (the u16 SH index sits just before it)
    bool shLerp;
    bool front;                 // clip.w > 0 after deletion / cutouts: the rest of the record is meaningful
    bool culled;                // allowCull only: the splat cannot reach the screen and the geometry was not finished
};

// outputsOnlyIfDrawn: the caller reads vp (beyond front / culled) only for a splat that passes PrepareSplat -- the per-frame kernel --
// so the record is not zero-filled first (a dozen register moves at every divergent exit of a VALU-bound kernel).
GS_HD void CalcViewGeom(const AssetView& a, const FrameConsts& P, const EditView& E, uint32_t idx, ViewPartial& vp, bool allowCull = false,
                        bool outputsOnlyIfDrawn = false) {
    ViewData& view = vp.view;
    if (!outputsOnlyIfDrawn) {
        view.pos[0] = view.pos[1] = view.pos[2] = view.pos[3] = 0.0f;
        view.axis1[0] = view.axis1[1] = view.axis2[0] = view.axis2[1] = 0.0f;
        view.color[0] = view.color[1] = 0u;
        vp.shLerp = false; vp.otherEnd = 0u;
        vp.shMin = { 0, 0, 0 }; vp.shMax = { 0, 0, 0 }; vp.col = { 0, 0, 0, 0 };
    }
    vp.front = false; vp.culled = false;

    // ---- LoadSplatData: position first (needed for the early out)
    V3 pos = LoadVec(a.pos, (uint64_t)idx * vecStride(a.posFmt), a.posFmt);
    const uint32_t ci = idx >> 8;
    const bool chunked = ci < a.chunkCount;
    ChunkRaw ck;
    if (chunked) {
        ck = LoadChunk(a.chunk, ci);
        pos.x = lerpf(u2f(ck.w[4]), u2f(ck.w[5]), pos.x);
        pos.y = lerpf(u2f(ck.w[6]), u2f(ck.w[7]), pos.y);
        pos.z = lerpf(u2f(ck.w[8]), u2f(ck.w[9]), pos.z);
    }
    const float wx = mrow(P.o2w, 0, pos.x, pos.y, pos.z), wy = mrow(P.o2w, 1, pos.x, pos.y, pos.z), wz = mrow(P.o2w, 2, pos.x, pos.y, pos.z);
    vp.wx = wx; vp.wy = wy; vp.wz = wz;
    view.pos[0] = mrow(P.vp, 0, wx, wy, wz);
    view.pos[1] = mrow(P.vp, 1, wx, wy, wz);
    view.pos[2] = mrow(P.vp, 2, wx, wy, wz);
    view.pos[3] = mrow(P.vp, 3, wx, wy, wz);
    // deleted? (:204-214) / cutouts (:216-220): centerClipPos.w = 0, the rest of the clip position is kept
    if (E.deletedBits && ((E.deletedBits[idx >> 5] >> (idx & 31u)) & 1u)) view.pos[3] = 0.0f;
    if (E.cutoutCount && IsSplatCut(E, pos.x, pos.y, pos.z)) view.pos[3] = 0.0f;
    if (view.pos[3] <= 0.0f) return;                               // behindCam (:223), literally: a NaN w is not "behind" (it is never drawn: PrepareSplat)
    vp.front = true;

    // ---- scale (needed first: the early cull below bounds the footprint with it)
    uint32_t otherStride = 4 + vecStride(a.scaleFmt);
    if (a.shFmt > 3) otherStride += 2;
    const uint64_t otherAddr = (uint64_t)idx * otherStride;
    vp.otherEnd = otherAddr + otherStride;
    V3 scale = LoadVec(a.other, otherAddr + 4, a.scaleFmt);
    if (chunked) {
        scale.x = lerpf(f16tof32(ck.w[10]), f16tof32(ck.w[10] >> 16), scale.x);
        scale.y = lerpf(f16tof32(ck.w[11]), f16tof32(ck.w[11] >> 16), scale.y);
        scale.z = lerpf(f16tof32(ck.w[12]), f16tof32(ck.w[12] >> 16), scale.z);
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
        scale.x *= scale.x; scale.y *= scale.y; scale.z *= scale.z;
    }
    const float ss2 = P.splatScale * P.splatScale;

    // ---- CalcCovariance2D, first half: the 2x3 matrix T = J * W
    float vx = mrow(P.mv, 0, pos.x, pos.y, pos.z), vy = mrow(P.mv, 1, pos.x, pos.y, pos.z);
    const float vz = mrow(P.mv, 2, pos.x, pos.y, pos.z);
    const float limX = P.limX, limY = P.limY, focal = P.focal;
    // the four divisions by viewPos.z and the two by its square share ONE reciprocal: rz = 1 / z, 1 / z^2 = rz * rz -- what a
    // shader compiler makes of them, and bit for bit what oracle/_ref's fused build of the reference text computes
    const float rz = 1.0f / vz;
    vx = fminf(fmaxf(vx * rz, -limX), limX) * vz;
    vy = fminf(fmaxf(vy * rz, -limY), limY) * vz;
    const float rzz = rz * rz;
    const float J00 = focal * rz, J02 = -(focal * vx) * rzz;
    const float J11 = J00, J12 = -(focal * vy) * rzz;
    const float T00 = fmaf(J02, P.mv[8], J00 * P.mv[0]), T01 = fmaf(J02, P.mv[9], J00 * P.mv[1]), T02 = fmaf(J02, P.mv[10], J00 * P.mv[2]);
    const float T10 = fmaf(J12, P.mv[8], J11 * P.mv[4]), T11 = fmaf(J12, P.mv[9], J11 * P.mv[5]), T12 = fmaf(J12, P.mv[10], J11 * P.mv[6]);

    // ---- early cull (only for callers that do not need the record of a splat that cannot be drawn).  Exactly as
    // PrepareSplat: centre depth outside [near, far] => clipped.  Conservatively: the quad's half extent along x is
    // 2 (|axis1.x| + |axis2.x|) <= 2 sqrt(2) sqrt(axis1.x^2 + axis2.x^2) (Cauchy-Schwarz) <= 2 sqrt(2) sqrt(2 lambda1) (the axes are
    // orthogonal, |axis_k|^2 = 2 lambda_k <= 2 lambda1), the same along y, and lambda1 <= trace(cov2d) <= (|T0|^2 + |T1|^2) lambda_max(Sigma) + 0.6 with
    // lambda_max(Sigma) <= splatScale^2 |R|^2 max(scale)^2  (|R|^2 <= 1.1 for a 10.10.10.2 quaternion); a centre farther than
    // that (+5 %, + 2 px) outside the screen cannot put a fragment on it.  NaNs compare false and take the full path.
    if (allowCull) {
        const float w = view.pos[3];
        if (!(w >= P.nearClip && w <= P.farClip)) { vp.culled = true; return; }
        const float smax = fmaxf(fmaxf(fabsf(scale.x), fabsf(scale.y)), fabsf(scale.z));
        const float t2 = dot3f(T00, T01, T02, T00, T01, T02) + dot3f(T10, T11, T12, T10, T11, T12);
        const float lam = fmaf(t2 * (1.1f * ss2), smax * smax, 0.6f);
        // (a bound with 5 % + 2 px of slack: the GPU's 1-ulp v_sqrt_f32 does, without the ~10-instruction correctly-rounded fix-up)
#if defined(__HIP_DEVICE_COMPILE__)
        const float rad = fmaf(GS_CULL_EXTENT * 1.05f, __builtin_amdgcn_sqrtf(2.0f * lam), 2.0f);
#else
        const float rad = fmaf(GS_CULL_EXTENT * 1.05f, sqrtf(2.0f * lam), 2.0f);
#endif
        const float invw = 1.0f / w;
        const float cx = fmaf(0.5f * (view.pos[0] * invw), P.screenW, 0.5f * P.screenW);
        const float cy = fmaf(-0.5f * (view.pos[1] * invw), P.screenH, 0.5f * P.screenH);
        if (cx - rad > P.screenW || cx + rad < 0.0f || cy - rad > P.screenH || cy + rad < 0.0f) { vp.culled = true; return; }
    }

    // ---- rotation
    const V4 q = DecodeRotation(LoadUInt(a.other, otherAddr));

    // ---- colour texel
    V4 col = LoadColorTexel(a, idx);

    if (chunked) {
        col.x = lerpf(f16tof32(ck.w[0]), f16tof32(ck.w[0] >> 16), col.x);
        col.y = lerpf(f16tof32(ck.w[1]), f16tof32(ck.w[1] >> 16), col.y);
        col.z = lerpf(f16tof32(ck.w[2]), f16tof32(ck.w[2] >> 16), col.z);
        col.w = lerpf(f16tof32(ck.w[3]), f16tof32(ck.w[3] >> 16), col.w);
        col.w = InvSquareCentered01(col.w);
        vp.shMin = { f16tof32(ck.w[13]), f16tof32(ck.w[14]), f16tof32(ck.w[15]) };
        vp.shMax = { f16tof32(ck.w[13] >> 16), f16tof32(ck.w[14] >> 16), f16tof32(ck.w[15] >> 16) };
        vp.shLerp = a.shFmt > 0 && a.shFmt <= 3;
    } else {
        // a chunk-less asset (VeryHigh fp32, or a lossy-format asset created without a chunk blob): CalcViewColor reads these
        // whatever the record's zero-fill above did (outputsOnlyIfDrawn skips it)
        vp.shLerp = false; vp.shMin = { 0, 0, 0 }; vp.shMax = { 0, 0, 0 };
    }
    vp.col = col;

    // ---- CalcMatrixFromRotationScale + CalcCovariance3D
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    // contraction of the nine entries = oracle/_ref's fused build (g++ -ffp-contract=fast of GaussianSplatting.hlsl:40-44): the
    // first product of every sum / difference is the fused one; x*x takes the fusion in r11 and r22, which leaves r00's sum plain
    const float r00 = fmaf(-2.0f, y * y + z * z, 1.0f), r01 = 2.0f * fmaf(x, y, -(w * z)), r02 = 2.0f * fmaf(x, z, w * y);
    const float r10 = 2.0f * fmaf(x, y, w * z), r11 = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f), r12 = 2.0f * fmaf(y, z, -(w * x));
    const float r20 = 2.0f * fmaf(x, z, -(w * y)), r21 = 2.0f * fmaf(y, z, w * x), r22 = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
    const float m00 = r00 * scale.x, m01 = r01 * scale.y, m02 = r02 * scale.z;
    const float m10 = r10 * scale.x, m11 = r11 * scale.y, m12 = r12 * scale.z;
    const float m20 = r20 * scale.x, m21 = r21 * scale.y, m22 = r22 * scale.z;
    const float c00 = dot3f(m00, m01, m02, m00, m01, m02) * ss2, c01 = dot3f(m00, m01, m02, m10, m11, m12) * ss2, c02 = dot3f(m00, m01, m02, m20, m21, m22) * ss2;
    const float c11 = dot3f(m10, m11, m12, m10, m11, m12) * ss2, c12 = dot3f(m10, m11, m12, m20, m21, m22) * ss2, c22 = dot3f(m20, m21, m22, m20, m21, m22) * ss2;

    // ---- CalcCovariance2D, second half: cov = T * Sigma * T^T (+ the 0.3 px low-pass)
    // VT[i][j] = sum_k V[i][k] * T[j][k]
    const float VT00 = dot3f(c00, c01, c02, T00, T01, T02), VT01 = dot3f(c00, c01, c02, T10, T11, T12);
    const float VT10 = dot3f(c01, c11, c12, T00, T01, T02), VT11 = dot3f(c01, c11, c12, T10, T11, T12);
    const float VT20 = dot3f(c02, c12, c22, T00, T01, T02), VT21 = dot3f(c02, c12, c22, T10, T11, T12);
    const float cov00 = dot3f(T00, T01, T02, VT00, VT10, VT20) + 0.3f;
    const float cov01 = dot3f(T00, T01, T02, VT01, VT11, VT21);
    const float cov11 = dot3f(T10, T11, T12, VT01, VT11, VT21) + 0.3f;

    // ---- DecomposeCovariance (#else branch)
    const float mid = 0.5f * (cov00 + cov11);
    const float hx = (cov00 - cov11) / 2.0f;
    const float radius = sqrtf(dot2f(hx, cov01, hx, cov01));
    const float lambda1 = mid + radius;
    const float lambda2 = fmaxf(mid - radius, 0.1f);
    float dvx = cov01, dvy = lambda1 - cov00;
    const float invLen = 1.0f / sqrtf(dot2f(dvx, dvy, dvx, dvy));
    dvx *= invLen; dvy *= invLen;
    dvy = -dvy;
    const float s1 = fminf(sqrtf(2.0f * lambda1), 4096.0f), s2 = fminf(sqrtf(2.0f * lambda2), 4096.0f);
    view.axis1[0] = s1 * dvx; view.axis1[1] = s1 * dvy;
    view.axis2[0] = s2 * dvy; view.axis2[1] = s2 * (-dvx);

    // ---- opacity (the colour's alpha half; rgb is added by CalcViewColor)
    const float al = fminf(col.w * P.opacityScale, 65000.0f);
    view.color[1] = f32tof16(al);
}

// SH coefficients are consumed in order sh1..sh15 by three fmaf chains (degree 1, 2, 3), so they are decoded
// one at a time instead of being held in 45 registers.
template <class SHSource>
GS_HD void CalcViewColor(const AssetView& a, const FrameConsts& P, uint32_t idx, ViewPartial& vp, SHSource& shsrc) {
    ViewData& view = vp.view;
    const V4 col = vp.col;
    const V3 shMin = vp.shMin, shMax = vp.shMax;
    const bool shLerp = vp.shLerp;
    const float wx = vp.wx, wy = vp.wy, wz = vp.wz;
    // ---- view direction in object space, SH basis
    const float dwx = P.camx - wx, dwy = P.camy - wy, dwz = P.camz - wz;
    float ox = mrow3(P.w2o, 0, dwx, dwy, dwz), oy = mrow3(P.w2o, 1, dwx, dwy, dwz), oz = mrow3(P.w2o, 2, dwx, dwy, dwz);
    const float invN = 1.0f / sqrtf(dot3f(ox, oy, oz, ox, oy, oz));
    ox *= invN; oy *= invN; oz *= invN;
    const float dx = -ox, dy = -oy, dz = -oz;

    const bool onlySH = P.shOnly != 0;
    // (r, g) travel as a 2-vector (packed fp32 instructions on the device), b alone; every channel sees exactly the
    // operations of ShadeSH in the same order
    F2 rg = onlySH ? F2{ 0.5f, 0.5f } : F2{ col.x, col.y };
    float b = onlySH ? 0.5f : col.z;
    if (P.shOrder >= 1) {
        uint32_t shIndex = idx;
        if (a.shFmt > 3) shIndex = LoadUShort(a.other, vp.otherEnd - 2);
        shsrc.begin(a.sh + (uint64_t)shIndex * shStrideOf(a.shFmt), a.shFmt);
        const F2 minRG = { shMin.x, shMin.y }, dRG = F2{ shMax.x, shMax.y } - minRG;      // lerp(a, b, t) = fma(t, b - a, a)
        const float minB = shMin.z, dB = shMax.z - shMin.z;
        auto SHK = [&](int k) -> RGB {
            RGB s = shsrc.load_rgb(k);
            if (shLerp) { s.rg = fma2(s.rg, dRG, minRG); s.b = fmaf(s.b, dB, minB); }
            return s;
        };
        auto MUL = [](float w, const RGB& s) -> RGB { return { F2{ w, w } * s.rg, w * s.b }; };
        auto FMA = [](float w, const RGB& s, const RGB& acc) -> RGB { return { fma2(F2{ w, w }, s.rg, acc.rg), fmaf(w, s.b, acc.b) }; };
        {   // degree 1: res += C1 * (-sh1*y + sh2*z - sh3*x)
            const RGB s1v = SHK(1), s2v = SHK(2), s3v = SHK(3);
            RGB t = MUL(dy, RGB{ -s1v.rg, -s1v.b });
            t = FMA(dz, s2v, t);
            t = FMA(dx, RGB{ -s3v.rg, -s3v.b }, t);
            rg = fma2(F2{ GS_SH_C1, GS_SH_C1 }, t.rg, rg); b = fmaf(GS_SH_C1, t.b, b);
        }
        if (P.shOrder >= 2) {
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            const float xy = dx * dy, yz = dy * dz, xz = dx * dz;
            {
                const float b4 = 1.0925484f * xy, b5 = -1.0925484f * yz, b6 = 0.3153916f * (fmaf(2.0f, zz, -xx) - yy),
                            b7 = -1.0925484f * xz, b8 = 0.5462742f * (xx - yy);
                RGB acc = MUL(b4, SHK(4));
                acc = FMA(b5, SHK(5), acc);
                acc = FMA(b6, SHK(6), acc);
                acc = FMA(b7, SHK(7), acc);
                acc = FMA(b8, SHK(8), acc);
                rg += acc.rg; b += acc.b;
            }
            if (P.shOrder >= 3) {
                const float b9 = (-0.5900436f * dy) * fmaf(3.0f, xx, -yy);
                const float b10 = (2.8906114f * xy) * dz;
                const float b11 = (-0.4570458f * dy) * (fmaf(4.0f, zz, -xx) - yy);
                const float b12 = (0.3731763f * dz) * (fmaf(2.0f, zz, -3.0f * xx) - 3.0f * yy);
                const float b13 = (-0.4570458f * dx) * (fmaf(4.0f, zz, -xx) - yy);
                const float b14 = (1.4453057f * dz) * (xx - yy);
                const float b15 = (-0.5900436f * dx) * fmaf(-3.0f, yy, xx);
                RGB acc = MUL(b9, SHK(9));
                acc = FMA(b10, SHK(10), acc);
                acc = FMA(b11, SHK(11), acc);
                acc = FMA(b12, SHK(12), acc);
                acc = FMA(b13, SHK(13), acc);
                acc = FMA(b14, SHK(14), acc);
                acc = FMA(b15, SHK(15), acc);
                rg += acc.rg; b += acc.b;
            }
        }
    }
    float r = rg.x, g = rg.y;
    r = fmaxf(r, 0.0f); g = fmaxf(g, 0.0f); b = fmaxf(b, 0.0f);
    view.color[0] = (f32tof16(r) << 16) | f32tof16(g);
    view.color[1] = (f32tof16(b) << 16) | (view.color[1] & 0xffffu);
}

template <class SHSource>
GS_HD ViewData CalcViewDataT(const AssetView& a, const FrameConsts& P, const EditView& E, uint32_t idx, SHSource& shsrc) {
    ViewPartial vp;
    CalcViewGeom(a, P, E, idx, vp);
    if (vp.front) CalcViewColor(a, P, idx, vp, shsrc);
    return vp.view;
}

GS_HD ViewData CalcViewData(const AssetView& a, const FrameConsts& P, const EditView& E, uint32_t idx) {
    SHFromBlob src;
    return CalcViewDataT(a, P, E, idx, src);
}

// ---- compositor set-up: which splats are drawn, where, and which pixels (hence tiles) they can touch -----------
// Restates the vertex stage of RenderGaussianSplats.shader:35-77 plus the fixed-function clipping it relies
// on (DESIGN.md "compositor semantics"); must stay in step with prepare() in oracle/gs_oracle.cpp.
GS_HD void PixRange(float c, float e, float size, int& lo, int& hi) {
    float flo = ceilf((c - e) - 0.5f), fhi = floorf((c + e) - 0.5f);
    flo = fmaxf(flo, 0.0f); fhi = fminf(fhi, size - 1.0f);
    if (!(flo <= fhi)) { lo = 1; hi = 0; return; }
    lo = (int)flo; hi = (int)fhi;
}

// ---- the fragment's discard decision, made identical on the CPU and the GPU ---------------------------------------
// frag() discards alpha < 1/255 (RenderGaussianSplats.shader:100).  alpha = saturate(exp(power) * a) goes through the one
// operation of the frame that is not bit-identical on both sides (the GPU's exp2 unit vs a correctly rounded exp2, <= 1 ulp),
// so a fragment whose alpha lands within that ulp of 1/255 would be kept on one side and discarded on the other.  Instead,
// when alpha comes out within kAlphaWindow / 2 ulps of 1/255 (about 10^-6 of the fragments) BOTH sides recompute it from
// Exp2Det, an exp2 made of fp32 operations only (the same bits everywhere), and decide on that.  Outside the window the native
// alpha is at least 8 ulps away from the threshold: no 1-ulp difference can change the decision.
constexpr uint32_t kAlphaThresholdBits = 0x3B808081u;                  // 1.0f / 255.0f
constexpr uint32_t kAlphaWindow = 16u;                                 // [1/255 - 8 ulp, 1/255 + 8 ulp)
constexpr uint32_t kAlphaWindowLo = kAlphaThresholdBits - kAlphaWindow / 2u;
// 2^y for y in (-126, 0]: n = rint(y), 2^(y - n) by the degree-7 Taylor polynomial of 2^f on [-0.5, 0.5] (|error| < 1 ulp), exact scaling
GS_HD float Exp2Det(float y) {
    const float n = rintf(y);
    const float f = y - n;
    float p = 1.525273380405984e-05f;
    p = fmaf(p, f, 1.5403530393381608e-04f);
    p = fmaf(p, f, 1.3333558146428443e-03f);
    p = fmaf(p, f, 9.618129107628477e-03f);
    p = fmaf(p, f, 5.550410866482158e-02f);
    p = fmaf(p, f, 2.402265069591007e-01f);
    p = fmaf(p, f, 6.931471805599453e-01f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}
// alpha of a fragment given the native alpha (= saturate(exp2(y) * a) with whatever exp2 the machine has), y = power * log2(e)
// and the splat's opacity a: returns the alpha to blend with, `live` = the fragment is not discarded.
GS_HD float DecideAlpha(float alphaNative, float y, float a, bool& live) {
    const uint32_t u = f2u(alphaNative) - kAlphaWindowLo;
    if (u < kAlphaWindow) {
        const float ad = fminf(fmaxf(Exp2Det(y) * a, 0.0f), 1.0f);
        live = ad >= u2f(kAlphaThresholdBits);
        return ad;
    }
    live = (int32_t)u >= (int32_t)kAlphaWindow;
    return alphaNative;
}

// ln(x) for a positive normal x from fp32 operations only (bit manipulation, one division, explicit fmaf): the same bits
// on the host and on the device, unlike logf().  |error| < 1e-6 (atanh series of the mantissa reduced to [0.707, 1.414)).
GS_HD float LogDet(float x) {
    const uint32_t u = f2u(x);
    float e = (float)((int)(u >> 23) - 127);
    float m = u2f((u & 0x7fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m *= 0.5f; e += 1.0f; }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    const float p = fmaf(t2, fmaf(t2, fmaf(t2, 1.0f / 7.0f, 0.2f), 1.0f / 3.0f), 1.0f);
    return fmaf(e, 0.69314718f, (2.0f * t) * p);
}

// Can the splat put a live fragment on a pixel centre of the square block of pixel centres [bc - half, bc + half]^2 ?
// A fragment is live iff |q1| <= 2, |q2| <= 2 and exp(-(q1^2+q2^2)) a >= 1/255, q_k = u_k . (p - c)  (frag, :79-106).
// q_k is linear in p, so min |q_k| over the block is closed form (separating-axis test along u_1 and u_2); the block is
// rejected when even those minima violate one of the three conditions.  Conservative: never rejects a live fragment
// (r2 carries the slack for the fp32 rounding of this test and of the kernel's exp).
GS_HD bool BlockMayTouch(float bcx, float bcy, float half, float cx, float cy, float u1x, float u1y, float u2x, float u2y, float r2) {
    const float dx = bcx - cx, dy = bcy - cy;
    const float d1 = fabsf(fmaf(dy, u1y, dx * u1x)) - half * (fabsf(u1x) + fabsf(u1y));
    const float d2 = fabsf(fmaf(dy, u2y, dx * u2x)) - half * (fabsf(u2x) + fabsf(u2y));
    const float m1 = fmaxf(d1, 0.0f), m2 = fmaxf(d2, 0.0f);
    return (m1 <= 2.001f) && (m2 <= 2.001f) && (fmaf(m2, m2, m1 * m1) <= r2);
}

struct SplatFootprint {
    float cx, cy;               // centre in pixels, y down
    int x0, x1, y0, y1;         // inclusive PIXEL rect of the tight footprint, clamped to the screen (x0 > x1: nothing to draw); the
                                // compositor's tile rectangle is this shifted by the tile shape the draw picks (x >> log2 tile width ...)
};
// The 8-byte per-splat rectangle calc_view leaves for the binning: x = x0 | y0 << 16, y = (x1 + 1) | (y1 + 1) << 16 (pixels; targets are
// at most 65535 pixels wide / high); all zero = not drawn.
GS_HD void PackPixelRect(const SplatFootprint& fp, uint32_t& rx, uint32_t& ry) {
    rx = (uint32_t)fp.x0 | ((uint32_t)fp.y0 << 16);
    ry = (uint32_t)(fp.x1 + 1) | ((uint32_t)(fp.y1 + 1) << 16);
}

GS_HD bool PrepareSplat(const ViewData& v, float W, float H, float nearClip, float farClip, SplatFootprint& fp) {
    fp.x0 = 1; fp.x1 = 0; fp.y0 = 1; fp.y1 = 0; fp.cx = 0.0f; fp.cy = 0.0f;
    const float w = v.pos[3];
    const float a = f16tof32(v.color[1]);
    // the rejections are evaluated as ONE predicate (bitwise &: no short-circuit): a chain of early returns compiles to
    // nested divergent branches that each re-materialise the "culled" outputs
    const int drawable = (int)(w > 0.0f) & (int)(w >= nearClip) & (int)(w <= farClip) & (int)finite32(v.axis1[0]) & (int)finite32(v.axis1[1]) &
                         (int)finite32(v.axis2[0]) & (int)finite32(v.axis2[1]) & (int)(a >= 1.0f / 255.0f);
    if (!drawable) return false;
    const float invw = 1.0f / w;
    const float cx = fmaf(0.5f * (v.pos[0] * invw), W, 0.5f * W);
    const float cy = fmaf(-0.5f * (v.pos[1] * invw), H, 0.5f * H);
    const float a1x = v.axis1[0], a1y = v.axis1[1], a2x = v.axis2[0], a2y = v.axis2[1];
    // the oracle's prepare() rejects a splat whose 1 / |axis_k|^2 is not finite (the blend divides by it).  |axis|^2 = d is a sum of two
    // squares (never negative), and fl(1 / d) is finite exactly when d is +inf or 2^-128 < d < inf and not NaN (1 / 2^-128 = 2^128
    // overflows, the next denormal up does not): an integer range test on the bits instead of two IEEE divisions (~22 VALU each)
    const float d1 = dot2f(a1x, a1y, a1x, a1y), d2 = dot2f(a2x, a2y, a2x, a2y);
    const int inv1ok = (int)((f2u(d1) - 0x00200001u) <= (0x7f800000u - 0x00200001u)), inv2ok = (int)((f2u(d2) - 0x00200001u) <= (0x7f800000u - 0x00200001u));
    fp.cx = cx; fp.cy = cy;                                         // (only read when the splat turns out visible)
    if (!((int)finite32(cx) & (int)finite32(cy) & inv1ok & inv2ok)) return false;
    const float exr = 2.0f * (fabsf(a1x) + fabsf(a2x)), eyr = 2.0f * (fabsf(a1y) + fabsf(a2y));
    const float slack = 0.01f;
    int x0, x1, y0, y1;
    PixRange(fp.cx, exr + slack, W, x0, x1);
    PixRange(fp.cy, eyr + slack, H, y0, y1);
    if (x0 > x1 || y0 > y1) return false;
    const float r2 = fmaf(LogDet(255.0f * a), 1.0001f, 1.0e-3f);
    const float rr = sqrtf(fmaxf(r2, 0.0f));
    const float exe = rr * sqrtf(dot2f(a1x, a2x, a1x, a2x)), eye = rr * sqrtf(dot2f(a1y, a2y, a1y, a2y));
    PixRange(fp.cx, fminf(exr, exe) + slack, W, x0, x1);
    PixRange(fp.cy, fminf(eyr, eye) + slack, H, y0, y1);
    if (x0 > x1 || y0 > y1) return true;         // drawn by the reference, but every fragment is below 1/255
    fp.x0 = x0; fp.x1 = x1; fp.y0 = y0; fp.y1 = y1;
    return true;
}

} // namespace gsm
