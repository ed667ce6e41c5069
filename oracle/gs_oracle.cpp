// gs_oracle.cpp -- CPU restatement of the reference's per-frame splat render path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker / reported CPU baseline.  The product path
// (unitygaussiansplatting_amd + libgsplat_hip.so) never links, imports or falls back to it.
//
// *** PARITY PINNED TO THE REFERENCE'S OWN SHADER TEXT (round 3). ***  The reference (aras-p/UnityGaussianSplatting) is HLSL +
// Unity C#; Unity / dxc / dotnet do not exist here and it ships no unit tests or golden vectors for this path other than
// full-scene PNGs of INRIA models that are not available offline.  But the shader maths is
// dependency-free: oracle/ref_build/ compiles the TEXT of GaussianSplatting.hlsl, SplatUtilities.compute:37-252, the vertex +
// fragment shader of RenderGaussianSplats.shader and the fragment shader of GaussianComposite.shader, read from /root/reference
// at build time, as C++ (oracle/_ref/libgs_ref_{strict,fused,fused_clang}.so; `make -C oracle ref`), and
// tests/test_ref_parity.py holds this file against it: the canonical arithmetic below IS the "fused"
// build of that text bit for bit -- sort keys, every field of LoadSplatData in every format, the whole 40-byte view record,
// the fragment's alpha and discard -- and whole frames rasterised through the reference's vert + frag match within the
// framebuffer bar.  What remains outside the reference tree (Unity's GammaToLinearSpace, the fixed-function raster / blend
// rules, f16 conversion) is restated in oracle/ref_build and listed in DESIGN.md.
//
// Each function cites the reference lines it restates (paths relative to /root/reference/package/).
//
// Build: see oracle/Makefile  (g++ -O2 -fopenmp -ffp-contract=off; no fast-math).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gsplat_c.h"

namespace {

// ---------------------------------------------------------------------------------------------
// small vector helpers; every multiply-add is an explicit fmaf (canonical arithmetic)
// ---------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline float dot2(float ax, float ay, float bx, float by) { return fmaf(ay, by, ax * bx); }
// Sensitivity switches (tests only; gso_set_canon).  HLSL leaves two evaluations to the GPU compiler: whether lerp's
// a + t*(b-a) is contracted into an FMA, and whether x / (2^k-1) is a true IEEE division or a multiply by a reciprocal.
// The canonical forms (flags = 0: fused lerp, rounded reciprocal) are what the HIP kernels implement; bit 0 selects the
// unfused lerp of SURVEY.md Appendix B, bit 1 the IEEE division of Appendix A, so that tests/test_canon_sensitivity.py can
// MEASURE what the choice moves.
int g_canon = 0;
inline float lerpf(float a, float b, float t) {                                          // HLSL lerp: a + t*(b-a)
    if (g_canon & 1) { const float d = b - a; const float m = t * d; return a + m; }     // three roundings (-ffp-contract=off)
    return fmaf(t, b - a, a);
}
inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
inline float signf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
// mul(M, float4(v,1)) row r of a row-major 4x4
inline float mul_row(const float* m, int r, f3 v) {
    return fmaf(m[r * 4 + 2], v.z, fmaf(m[r * 4 + 1], v.y, fmaf(m[r * 4 + 0], v.x, m[r * 4 + 3])));
}
// mul((float3x3)M, v) row r
inline float mul3_row(const float* m, int r, f3 v) {
    return fmaf(m[r * 4 + 2], v.z, fmaf(m[r * 4 + 1], v.y, m[r * 4 + 0] * v.x));
}

// ---- IEEE half conversion, round-to-nearest-even (HLSL f32tof16 / f16tof32; Burst math.f32tof16) ----
inline uint16_t f32tof16(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);             // NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            // >= 65520 -> inf (also inf)
    if (x < 0x38800000u) {                                              // < 2^-14: half subnormal or zero
        if (x < 0x33000000u) return (uint16_t)sign;                     // < 2^-25 -> 0
        const uint32_t e = x >> 23;                                     // 102..112
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;                 // 24-bit significand
        const uint32_t shift = 126u - e;                                // 14..24: result = m >> shift (RTNE)
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = x - 0x38000000u;                                       // rebias exponent 127 -> 15
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}
inline float f16tof32(uint32_t h) {
    h &= 0xffffu;
    const uint32_t sign = (h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {                                                          // subnormal: normalise
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; s++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4);
    return f;
}
inline float round_f16(float f) { return f16tof32(f32tof16(f)); }

// One blend of the render-target unit into an RGBA16F target (RenderGaussianSplats.shader:10-12, "Blend OneMinusDstAlpha
// One"): dst' = RTNE_f16(src * t + dst) with the sum of products evaluated EXACTLY and rounded ONCE to the storage
// format.  (The fragment's src = rgb*alpha and t = 1 - dst.a are fp32 values.)  src*t is exact in double (24 x 24 bits),
// fma() makes the double result correctly rounded at 53 bits, and 53 bits are so much wider than the 11 of a half that
// the second rounding below can only differ from a single rounding when the 53-bit result is an exact fp16 tie that
// the infinitely precise value misses by < 2^-40 relative -- it does not occur for these operand widths.
inline uint16_t f64tof16(double d) {
    uint64_t u; std::memcpy(&u, &d, 8);
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    const int ebits = (int)((u >> 52) & 0x7ffu);
    const uint64_t mant = u & ((1ull << 52) - 1ull);
    if (ebits == 0x7ff) return (uint16_t)(sign | (mant ? 0x7e00u : 0x7c00u));
    if (ebits == 0) return sign;                                          // 0 or a double subnormal: far below half range
    const int e = ebits - 1023;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);
    const uint64_t sig = mant | (1ull << 52);                             // 53-bit significand, value = sig * 2^(e-52)
    const int shift = 42 + (e < -14 ? (-14 - e) : 0);                     // keep 11 bits (normal) or fewer (subnormal)
    if (shift >= 64) return sign;
    uint64_t q = sig >> shift;
    const uint64_t rem = sig & ((1ull << shift) - 1ull), half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (q & 1ull))) q++;
    const uint32_t h = (e < -14) ? (uint32_t)q : (uint32_t)(((uint32_t)(e + 14) << 10) + q);
    return (uint16_t)(sign | (h >= 0x7c00u ? 0x7c00u : h));
}
inline float blend_f16(float src, float t, float dst) { return f16tof32(f64tof16(std::fma((double)src, (double)t, (double)dst))); }

// ---------------------------------------------------------------------------------------------
// asset view
// ---------------------------------------------------------------------------------------------
struct Asset {
    uint32_t n, posFmt, scaleFmt, colorFmt, shFmt, chunkCount;
    const uint8_t *pos, *other, *color, *sh, *chunk;
};

Asset make_asset(const gs_asset_desc* d) {
    Asset a;
    a.n = d->splat_count; a.posFmt = d->pos_format; a.scaleFmt = d->scale_format;
    a.colorFmt = d->color_format; a.shFmt = d->sh_format;
    a.pos = (const uint8_t*)d->pos_data; a.other = (const uint8_t*)d->other_data;
    a.color = (const uint8_t*)d->color_data; a.sh = (const uint8_t*)d->sh_data;
    a.chunk = (const uint8_t*)d->chunk_data;
    a.chunkCount = (d->chunk_data && d->chunk_size) ? (uint32_t)(d->chunk_size / 64) : 0;   // GaussianSplatRenderer.cs:504
    return a;
}

inline uint32_t load_u32(const uint8_t* p, uint64_t byteAddr) { uint32_t v; std::memcpy(&v, p + byteAddr, 4); return v; }
inline float load_f32(const uint8_t* p, uint64_t byteAddr) { float v; std::memcpy(&v, p + byteAddr, 4); return v; }

// GaussianSplatting.hlsl:325-343 LoadUShort / LoadUInt (2-byte aligned addresses stitched from aligned dwords;
// on a byte-addressable CPU this is a plain unaligned little-endian load)
inline uint32_t LoadUShort(const uint8_t* p, uint64_t a) { uint16_t v; std::memcpy(&v, p + a, 2); return v; }
inline uint32_t LoadUInt(const uint8_t* p, uint64_t a) { return load_u32(p, a); }

// GaussianSplatting.hlsl:261-300 DecodePacked_*; division by (2^bits-1) is the canonical multiply by the
// fp32-rounded reciprocal (DESIGN.md canonical arithmetic #2)
constexpr float R63 = 1.0f / 63.0f, R31 = 1.0f / 31.0f, R2047 = 1.0f / 2047.0f, R1023 = 1.0f / 1023.0f,
                R65535 = 1.0f / 65535.0f, R255 = 1.0f / 255.0f;
// field / (2^bits - 1): canonical = multiply by the rounded reciprocal; g_canon bit 1 = IEEE division (sensitivity only)
inline float unorm(uint32_t field, float k, float rk) { return (g_canon & 2) ? (float)field / k : (float)field * rk; }
inline f3 DecodePacked_6_5_5(uint32_t e) { return { unorm(e & 63, 63.0f, R63), unorm((e >> 6) & 31, 31.0f, R31), unorm((e >> 11) & 31, 31.0f, R31) }; }
inline f3 DecodePacked_5_6_5(uint32_t e) { return { unorm(e & 31, 31.0f, R31), unorm((e >> 5) & 63, 63.0f, R63), unorm((e >> 11) & 31, 31.0f, R31) }; }
inline f3 DecodePacked_11_10_11(uint32_t e) { return { unorm(e & 2047, 2047.0f, R2047), unorm((e >> 11) & 1023, 1023.0f, R1023), unorm((e >> 21) & 2047, 2047.0f, R2047) }; }
inline f3 DecodePacked_16_16_16(uint32_t e0, uint32_t e1) { return { unorm(e0 & 65535, 65535.0f, R65535), unorm((e0 >> 16) & 65535, 65535.0f, R65535), unorm(e1 & 65535, 65535.0f, R65535) }; }

inline uint32_t vec_stride(uint32_t fmt) { return fmt == 0 ? 12u : fmt == 1 ? 6u : fmt == 2 ? 4u : 2u; }

// GaussianSplatting.hlsl:346-392 LoadAndDecodeVector
inline f3 LoadAndDecodeVector(const uint8_t* buf, uint64_t addrU, uint32_t fmt) {
    if (fmt == 0) return { load_f32(buf, addrU), load_f32(buf, addrU + 4), load_f32(buf, addrU + 8) };
    if (fmt == 1) return DecodePacked_16_16_16(LoadUInt(buf, addrU), LoadUShort(buf, addrU + 4));
    if (fmt == 2) return DecodePacked_11_10_11(LoadUInt(buf, addrU));
    return DecodePacked_6_5_5(LoadUShort(buf, addrU));
}

struct Chunk {          // GaussianSplatting.hlsl:196-202 SplatChunkInfo
    uint32_t colR, colG, colB, colA;
    float posX[2], posY[2], posZ[2];
    uint32_t sclX, sclY, sclZ;
    uint32_t shR, shG, shB;
};
static_assert(sizeof(Chunk) == 64, "ChunkInfo is 64 B");
inline Chunk load_chunk(const Asset& a, uint32_t ci) { Chunk c; std::memcpy(&c, a.chunk + (uint64_t)ci * 64, 64); return c; }

// GaussianSplatting.hlsl:394-421 LoadSplatPosValue / LoadSplatPos
inline f3 LoadSplatPos(const Asset& a, uint32_t idx) {
    f3 pos = LoadAndDecodeVector(a.pos, (uint64_t)idx * vec_stride(a.posFmt), a.posFmt);
    const uint32_t chunkIdx = idx / 256;
    if (chunkIdx < a.chunkCount) {
        const Chunk c = load_chunk(a, chunkIdx);
        pos.x = lerpf(c.posX[0], c.posX[1], pos.x);
        pos.y = lerpf(c.posY[0], c.posY[1], pos.y);
        pos.z = lerpf(c.posZ[0], c.posZ[1], pos.z);
    }
    return pos;
}

// GaussianSplatting.hlsl:113-127,183-194  DecodeMorton2D_16x16 / SplatIndexToPixelIndex
inline void SplatIndexToPixelIndex(uint32_t idx, uint32_t& x, uint32_t& y) {
    uint32_t t = idx;
    t = (t & 0xFF) | ((t & 0xFE) << 7);
    t &= 0x5555;
    t = (t ^ (t >> 1)) & 0x3333;
    t = (t ^ (t >> 2)) & 0x0f0f;
    const uint32_t mx = t & 0xF, my = t >> 8;
    const uint32_t width = 2048 / 16;
    idx >>= 8;
    x = (idx % width) * 16 + mx;
    y = (idx / width) * 16 + my;
}

// GaussianSplatting.hlsl:5-11
inline float InvSquareCentered01(float x) {
    x -= 0.5f;
    x *= 0.5f;
    x = sqrtf(fabsf(x)) * signf(x);
    return x + 0.5f;
}

// GaussianSplatting.hlsl:219-229 DecodeRotation(DecodePacked_10_10_10_2(enc)).  round(pq.w*3) == the 2-bit field.
inline f4 DecodeRotation(uint32_t enc) {
    const float px = unorm(enc & 1023, 1023.0f, R1023), py = unorm((enc >> 10) & 1023, 1023.0f, R1023), pz = unorm((enc >> 20) & 1023, 1023.0f, R1023);
    const uint32_t idx = (enc >> 30) & 3;
    const float SQRT2 = 1.41421356237f, INV_SQRT2 = 0.70710678118f;
    const float qx = fmaf(px, SQRT2, -INV_SQRT2), qy = fmaf(py, SQRT2, -INV_SQRT2), qz = fmaf(pz, SQRT2, -INV_SQRT2);
    const float qw = sqrtf(1.0f - saturatef(dot3({qx, qy, qz}, {qx, qy, qz})));
    f4 q = {qx, qy, qz, qw};
    if (idx == 0) q = {qw, qx, qy, qz};          // q.wxyz
    if (idx == 1) q = {qx, qw, qy, qz};          // q.xwyz
    if (idx == 2) q = {qx, qy, qw, qz};          // q.xywz
    return q;
}


// ---------------------------------------------------------------------------------------------
// BC7 (BPTC) texel fetch: what `_SplatColor.Load` returns for ColorFormat.BC7 (RGBA_BC7_UNorm; GaussianSplatAsset.cs:56,169).
// The block is decoded by a sequential bit reader into all 16 texels (a different structure from the kernel's single-texel
// decoder, gs_device_math.h); both are checked against Pillow's decoder by tests/test_bc7.py.  Tables: Khronos Data Format
// Specification "BPTC" (partition shapes as per-texel subset numbers, anchors of the 2nd / 3rd subset).
// ---------------------------------------------------------------------------------------------
const uint16_t BC7_P2[64] = { 0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80, 0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000, 0xf710, 0x8e, 0x7100, 0x8ce, 0x8c, 0x7310, 0x3100, 0x8cce, 0x88c, 0x3110, 0x6666, 0x366c, 0x17e8, 0xff0, 0x718e, 0x399c, 0xaaaa, 0xf0f0, 0x5a5a, 0x33cc, 0x3c3c, 0x55aa, 0x9696, 0xa55a, 0x73ce, 0x13c8, 0x324c, 0x3bdc, 0x6996, 0xc33c, 0x9966, 0x660, 0x272, 0x4e4, 0x4e40, 0x2720, 0xc936, 0x936c, 0x39c6, 0x639c, 0x9336, 0x9cc6, 0x817e, 0xe718, 0xccf0, 0xfcc, 0x7744, 0xee22 };
const uint32_t BC7_P3[64] = { 0xaa685050, 0x6a5a5040, 0x5a5a4200, 0x5450a0a8, 0xa5a50000, 0xa0a05050, 0x5555a0a0, 0x5a5a5050, 0xaa550000, 0xaa555500, 0xaaaa5500, 0x90909090, 0x94949494, 0xa4a4a4a4, 0xa9a59450, 0x2a0a4250, 0xa5945040, 0xa425054, 0xa5a5a500, 0x55a0a0a0, 0xa8a85454, 0x6a6a4040, 0xa4a45000, 0x1a1a0500, 0x50a4a4, 0xaaa59090, 0x14696914, 0x69691400, 0xa08585a0, 0xaa821414, 0x50a4a450, 0x6a5a0200, 0xa9a58000, 0x5090a0a8, 0xa8a09050, 0x24242424, 0xaa5500, 0x24924924, 0x24499224, 0x50a50a50, 0x500aa550, 0xaaaa4444, 0x66660000, 0xa5a0a5a0, 0x50a050a0, 0x69286928, 0x44aaaa44, 0x66666600, 0xaa444444, 0x54a854a8, 0x95809580, 0x96969600, 0xa85454a8, 0x80959580, 0xaa141414, 0x96960000, 0xaaaa1414, 0xa05050a0, 0xa0a5a5a0, 0x96000000, 0x40804080, 0xa9a8a9a8, 0xaaaaaa44, 0x2a4a5254 };
const uint8_t BC7_A2[64] = { 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2, 15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6, 6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15 };
const uint8_t BC7_A3A[64] = { 3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3, 3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15, 8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15, 3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3 };
const uint8_t BC7_A3B[64] = { 15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8, 15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8, 15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8, 15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8 };
const int BC7_MODE[8][10] = {   // subsets, partition bits, rotation, index selection, colour bits, alpha bits, endpoint p-bits, shared p-bits, index bits, 2nd index bits
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0} };
const int BC7_W2[4] = {0, 21, 43, 64}, BC7_W3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, BC7_W4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

struct BitReader {
    const uint8_t* p; int pos;
    int get(int n) { int v = 0; for (int k = 0; k < n; ++k, ++pos) v |= ((p[pos >> 3] >> (pos & 7)) & 1) << k; return v; }
};

void bc7_decode_block(const uint8_t* block, uint8_t out[16][4]) {
    std::memset(out, 0, 64);
    int mode = 0;
    while (mode < 8 && !((block[0] >> mode) & 1)) ++mode;
    if (mode == 8) return;                                     // reserved: zeros
    const int* M = BC7_MODE[mode];
    const int ns = M[0], cb = M[4], ab = M[5], ib = M[8], ib2 = M[9];
    BitReader br{block, mode + 1};
    const int shape = br.get(M[1]), rot = br.get(M[2]), isel = br.get(M[3]);
    int ep[6][4];
    for (int ch = 0; ch < 3; ++ch) for (int e = 0; e < 2 * ns; ++e) ep[e][ch] = br.get(cb);
    for (int e = 0; e < 2 * ns; ++e) ep[e][3] = ab ? br.get(ab) : 255;
    int cbits = cb, abits = ab;
    if (M[6]) { for (int e = 0; e < 2 * ns; ++e) { const int pb = br.get(1); for (int ch = 0; ch < (ab ? 4 : 3); ++ch) ep[e][ch] = (ep[e][ch] << 1) | pb; } cbits++; if (ab) abits++; }
    else if (M[7]) { for (int sub = 0; sub < ns; ++sub) { const int pb = br.get(1); for (int e = 2 * sub; e < 2 * sub + 2; ++e) for (int ch = 0; ch < 3; ++ch) ep[e][ch] = (ep[e][ch] << 1) | pb; } cbits++; }
    for (int e = 0; e < 2 * ns; ++e) {
        for (int ch = 0; ch < 3; ++ch) { const int x = ep[e][ch] << (8 - cbits); ep[e][ch] = x | (x >> cbits); }
        if (ab) { const int x = ep[e][3] << (8 - abits); ep[e][3] = x | (x >> abits); }
    }
    int subset[16], anchor[3] = {0, -1, -1};
    for (int t = 0; t < 16; ++t) subset[t] = ns == 1 ? 0 : (ns == 2 ? (BC7_P2[shape] >> t) & 1 : (BC7_P3[shape] >> (2 * t)) & 3);
    if (ns == 2) anchor[1] = BC7_A2[shape];
    if (ns == 3) { anchor[1] = BC7_A3A[shape]; anchor[2] = BC7_A3B[shape]; }
    int i1[16], i2[16];
    for (int t = 0; t < 16; ++t) i1[t] = br.get(t == anchor[subset[t]] ? ib - 1 : ib);
    for (int t = 0; t < 16; ++t) i2[t] = ib2 ? br.get(t == 0 ? ib2 - 1 : ib2) : 0;
    auto weight = [](int bits, int i) { return bits == 2 ? BC7_W2[i] : (bits == 3 ? BC7_W3[i] : BC7_W4[i]); };
    for (int t = 0; t < 16; ++t) {
        const int* e0 = ep[2 * subset[t]]; const int* e1 = ep[2 * subset[t] + 1];
        int ci = i1[t], cbt = ib, ai = ib2 ? i2[t] : i1[t], abt = ib2 ? ib2 : ib;
        if (isel) { std::swap(ci, ai); std::swap(cbt, abt); }
        const int wc = weight(cbt, ci), wa = weight(abt, ai);
        int px[4];
        for (int ch = 0; ch < 3; ++ch) px[ch] = ((64 - wc) * e0[ch] + wc * e1[ch] + 32) >> 6;
        px[3] = ((64 - wa) * e0[3] + wa * e1[3] + 32) >> 6;
        if (rot) std::swap(px[3], px[rot - 1]);
        for (int ch = 0; ch < 4; ++ch) out[t][ch] = (uint8_t)px[ch];
    }
}

struct SplatData {      // GaussianSplatting.hlsl:134-137,209-216
    f3 pos; f4 rot; f3 scale; float opacity;
    f3 col; f3 sh[15];
};

// GaussianSplatting.hlsl:428-608 LoadSplatData
SplatData LoadSplatData(const Asset& a, uint32_t idx) {
    SplatData s;
    uint32_t cx, cy;
    SplatIndexToPixelIndex(idx, cx, cy);
    const uint32_t scaleFmt = a.scaleFmt, shFormat = a.shFmt;
    uint32_t otherStride = 4 + vec_stride(scaleFmt);
    if (shFormat > 3) otherStride += 2;
    const uint64_t otherAddr = (uint64_t)idx * otherStride;
    uint32_t shStride = 0;
    if (shFormat == 0) shStride = 192;
    else if (shFormat == 1 || shFormat > 3) shStride = 96;
    else if (shFormat == 2) shStride = 60;
    else if (shFormat == 3) shStride = 32;

    s.pos = LoadAndDecodeVector(a.pos, (uint64_t)idx * vec_stride(a.posFmt), a.posFmt);
    s.rot = DecodeRotation(LoadUInt(a.other, otherAddr));
    s.scale = LoadAndDecodeVector(a.other, otherAddr + 4, scaleFmt);
    // _SplatColor.Load(coord): texel fetch with the texture's own format conversion
    f4 col;
    const uint64_t texel = (uint64_t)cy * 2048 + cx;
    if (a.colorFmt == 0) {
        col = { load_f32(a.color, texel * 16), load_f32(a.color, texel * 16 + 4), load_f32(a.color, texel * 16 + 8), load_f32(a.color, texel * 16 + 12) };
    } else if (a.colorFmt == 1) {
        const uint32_t lo = load_u32(a.color, texel * 8), hi = load_u32(a.color, texel * 8 + 4);
        col = { f16tof32(lo), f16tof32(lo >> 16), f16tof32(hi), f16tof32(hi >> 16) };
    } else if (a.colorFmt == 3) {                               // RGBA_BC7_UNorm: 16-byte blocks of 4x4 texels, 512 blocks per row
        uint8_t px[16][4];
        bc7_decode_block(a.color + ((uint64_t)(cy >> 2) * 512 + (cx >> 2)) * 16, px);
        const uint8_t* t = px[(cy & 3) * 4 + (cx & 3)];
        col = { unorm(t[0], 255.0f, R255), unorm(t[1], 255.0f, R255), unorm(t[2], 255.0f, R255), unorm(t[3], 255.0f, R255) };
    } else {
        const uint32_t e = load_u32(a.color, texel * 4);        // R8G8B8A8_UNorm: x/255
        col = { unorm(e & 255, 255.0f, R255), unorm((e >> 8) & 255, 255.0f, R255), unorm((e >> 16) & 255, 255.0f, R255), unorm(e >> 24, 255.0f, R255) };
    }

    uint32_t shIndex = idx;
    if (shFormat > 3) shIndex = LoadUShort(a.other, otherAddr + otherStride - 2);
    const uint64_t shOffset = (uint64_t)shIndex * shStride;
    if (shFormat == 0) {
        for (int k = 0; k < 15; ++k)
            s.sh[k] = { load_f32(a.sh, shOffset + k * 12), load_f32(a.sh, shOffset + k * 12 + 4), load_f32(a.sh, shOffset + k * 12 + 8) };
    } else if (shFormat == 1 || shFormat > 3) {
        for (int k = 0; k < 15; ++k)
            s.sh[k] = { f16tof32(LoadUShort(a.sh, shOffset + k * 6)), f16tof32(LoadUShort(a.sh, shOffset + k * 6 + 2)), f16tof32(LoadUShort(a.sh, shOffset + k * 6 + 4)) };
    } else if (shFormat == 2) {
        for (int k = 0; k < 15; ++k) s.sh[k] = DecodePacked_11_10_11(load_u32(a.sh, shOffset + k * 4));
    } else {
        for (int k = 0; k < 15; ++k) s.sh[k] = DecodePacked_5_6_5(LoadUShort(a.sh, shOffset + k * 2));
    }

    const uint32_t chunkIdx = idx / 256;
    if (chunkIdx < a.chunkCount) {                                      // :565-603
        const Chunk c = load_chunk(a, chunkIdx);
        const f3 sclMin = { f16tof32(c.sclX), f16tof32(c.sclY), f16tof32(c.sclZ) };
        const f3 sclMax = { f16tof32(c.sclX >> 16), f16tof32(c.sclY >> 16), f16tof32(c.sclZ >> 16) };
        const f4 colMin = { f16tof32(c.colR), f16tof32(c.colG), f16tof32(c.colB), f16tof32(c.colA) };
        const f4 colMax = { f16tof32(c.colR >> 16), f16tof32(c.colG >> 16), f16tof32(c.colB >> 16), f16tof32(c.colA >> 16) };
        const f3 shMin = { f16tof32(c.shR), f16tof32(c.shG), f16tof32(c.shB) };
        const f3 shMax = { f16tof32(c.shR >> 16), f16tof32(c.shG >> 16), f16tof32(c.shB >> 16) };
        s.pos = { lerpf(c.posX[0], c.posX[1], s.pos.x), lerpf(c.posY[0], c.posY[1], s.pos.y), lerpf(c.posZ[0], c.posZ[1], s.pos.z) };
        s.scale = { lerpf(sclMin.x, sclMax.x, s.scale.x), lerpf(sclMin.y, sclMax.y, s.scale.y), lerpf(sclMin.z, sclMax.z, s.scale.z) };
        for (int r = 0; r < 3; ++r) { s.scale.x *= s.scale.x; s.scale.y *= s.scale.y; s.scale.z *= s.scale.z; }   // ^8
        col = { lerpf(colMin.x, colMax.x, col.x), lerpf(colMin.y, colMax.y, col.y), lerpf(colMin.z, colMax.z, col.z), lerpf(colMin.w, colMax.w, col.w) };
        col.w = InvSquareCentered01(col.w);
        if (shFormat > 0 && shFormat <= 3)
            for (int k = 0; k < 15; ++k)
                s.sh[k] = { lerpf(shMin.x, shMax.x, s.sh[k].x), lerpf(shMin.y, shMax.y, s.sh[k].y), lerpf(shMin.z, shMax.z, s.sh[k].z) };
    }
    s.opacity = col.w;
    s.col = { col.x, col.y, col.z };
    return s;
}

// SplatUtilities.compute:52-57
inline uint32_t FloatToSortableUint(float f) {
    uint32_t fu; std::memcpy(&fu, &f, 4);
    const uint32_t mask = (uint32_t)(-(int32_t)(fu >> 31)) | 0x80000000u;
    return fu ^ mask;
}

// GaussianSplatting.hlsl:130-179 ShadeSH (half == float on desktop)
const float SH_C1 = 0.4886025f;
const float SH_C2[5] = { 1.0925484f, -1.0925484f, 0.3153916f, -1.0925484f, 0.5462742f };
const float SH_C3[7] = { -0.5900436f, 2.8906114f, -0.4570458f, 0.3731763f, -0.4570458f, 1.4453057f, -0.5900436f };

inline float sh_channel(float col, const float* sh /*15 coeffs of one channel, stride 3*/, float x, float y, float z, int shOrder, bool onlySH) {
    auto S = [&](int k) { return sh[(k - 1) * 3]; };   // S(1)..S(15)
    float res = onlySH ? 0.5f : col;
    if (shOrder >= 1) {
        float t = (-S(1)) * y;
        t = fmaf(S(2), z, t);
        t = fmaf(-S(3), x, t);
        res = fmaf(SH_C1, t, res);
        if (shOrder >= 2) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            float a = (SH_C2[0] * xy) * S(4);
            a = fmaf(SH_C2[1] * yz, S(5), a);
            a = fmaf(SH_C2[2] * (fmaf(2.0f, zz, -xx) - yy), S(6), a);
            a = fmaf(SH_C2[3] * xz, S(7), a);
            a = fmaf(SH_C2[4] * (xx - yy), S(8), a);
            res += a;
            if (shOrder >= 3) {
                float b = ((SH_C3[0] * y) * fmaf(3.0f, xx, -yy)) * S(9);
                b = fmaf((SH_C3[1] * xy) * z, S(10), b);
                b = fmaf((SH_C3[2] * y) * (fmaf(4.0f, zz, -xx) - yy), S(11), b);
                b = fmaf((SH_C3[3] * z) * (fmaf(2.0f, zz, -3.0f * xx) - 3.0f * yy), S(12), b);
                b = fmaf((SH_C3[4] * x) * (fmaf(4.0f, zz, -xx) - yy), S(13), b);
                b = fmaf((SH_C3[5] * z) * (xx - yy), S(14), b);
                b = fmaf((SH_C3[6] * x) * fmaf(-3.0f, yy, xx), S(15), b);
                res += b;
            }
        }
    }
    return fmaxf(res, 0.0f);
}

struct ViewData {           // GaussianSplatting.hlsl:610-615  (40 bytes)
    float pos[4];
    float axis1[2], axis2[2];
    uint32_t color[2];
};
static_assert(sizeof(ViewData) == 40, "SplatViewData is 40 B");

// SplatUtilities.compute:164-187 IsSplatCut over _SplatCutouts[0.._SplatCutoutsCount)
bool IsSplatCut(const gs_cutout* cutouts, uint32_t count, f3 pos) {
    bool finalCut = false;
    for (uint32_t i = 0; i < count; ++i) {
        const gs_cutout& cutData = cutouts[i];
        const uint32_t type = cutData.type_and_flags & 0xFFu;
        if (type == 0xFFu) continue;                                   // invalid/null cutout, ignore
        const bool invert = (cutData.type_and_flags & 0xFF00u) != 0;
        const f3 cutoutPos = { mul_row(cutData.matrix, 0, pos), mul_row(cutData.matrix, 1, pos), mul_row(cutData.matrix, 2, pos) };
        if (type == 0u) { if (dot3(cutoutPos, cutoutPos) <= 1.0f) return invert; }                                          // ellipsoid
        if (type == 1u) { if (fabsf(cutoutPos.x) <= 1.0f && fabsf(cutoutPos.y) <= 1.0f && fabsf(cutoutPos.z) <= 1.0f) return invert; }   // box
        finalCut |= !invert;
    }
    return finalCut;
}

thread_local float* g_stage = nullptr;   // gso_cov_stages: where CalcViewDataOne drops its intermediate covariance stages

// SplatUtilities.compute:189-252 CSCalcViewData for one splat
ViewData CalcViewDataOne(const Asset& a, const gs_frame_params& P, uint32_t idx, const gs_cutout* cutouts = nullptr, uint32_t cutoutCount = 0,
                         const uint32_t* deletedBits = nullptr) {
    const SplatData splat = LoadSplatData(a, idx);
    ViewData view; std::memset(&view, 0, sizeof(view));

    const f3 centerWorldPos = { mul_row(P.matrix_object_to_world, 0, splat.pos), mul_row(P.matrix_object_to_world, 1, splat.pos), mul_row(P.matrix_object_to_world, 2, splat.pos) };
    float clip[4] = { mul_row(P.matrix_vp, 0, centerWorldPos), mul_row(P.matrix_vp, 1, centerWorldPos), mul_row(P.matrix_vp, 2, centerWorldPos), mul_row(P.matrix_vp, 3, centerWorldPos) };
    // deleted? (:204-214, _SplatBitsValid = deletedBits != null)
    if (deletedBits) {
        const uint32_t wordIdx = idx / 32, bitIdx = idx & 31;
        if (deletedBits[wordIdx] & (1u << bitIdx)) clip[3] = 0.0f;
    }
    // cutouts (:216-220)
    if (IsSplatCut(cutouts, cutoutCount, splat.pos)) clip[3] = 0.0f;
    for (int k = 0; k < 4; ++k) view.pos[k] = clip[k];
    const bool behindCam = clip[3] <= 0.0f;                        // SplatUtilities.compute:223, literally (a NaN w is not "behind")
    if (behindCam) return view;

    // CalcMatrixFromRotationScale (GaussianSplatting.hlsl:29-46): mul(mr, diag(scale))
    const float x = splat.rot.x, y = splat.rot.y, z = splat.rot.z, w = splat.rot.w;
    // (contraction as oracle/_ref's fused build of :40-44 evaluates it: the first product of a sum is the fused one; x*x takes the
    //  fusion in the [1][1] and [2][2] entries, which leaves the [0][0] sum plain)
    float mr[3][3] = {
        { fmaf(-2.0f, y * y + z * z, 1.0f),      2.0f * fmaf(x, y, -(w * z)),          2.0f * fmaf(x, z, w * y) },
        { 2.0f * fmaf(x, y, w * z),            fmaf(-2.0f, fmaf(x, x, z * z), 1.0f), 2.0f * fmaf(y, z, -(w * x)) },
        { 2.0f * fmaf(x, z, -(w * y)),         2.0f * fmaf(y, z, w * x),            fmaf(-2.0f, fmaf(x, x, y * y), 1.0f) } };
    float M[3][3];
    for (int i = 0; i < 3; ++i) { M[i][0] = mr[i][0] * splat.scale.x; M[i][1] = mr[i][1] * splat.scale.y; M[i][2] = mr[i][2] * splat.scale.z; }
    // CalcCovariance3D (:48-53): sig = M * M^T, 6 unique
    auto sig = [&](int i, int j) { return fmaf(M[i][2], M[j][2], fmaf(M[i][1], M[j][1], M[i][0] * M[j][0])); };
    const float splatScale2 = P.splat_scale * P.splat_scale;
    const float c00 = sig(0, 0) * splatScale2, c01 = sig(0, 1) * splatScale2, c02 = sig(0, 2) * splatScale2;
    const float c11 = sig(1, 1) * splatScale2, c12 = sig(1, 2) * splatScale2, c22 = sig(2, 2) * splatScale2;

    if (g_stage) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) g_stage[i * 3 + j] = M[i][j];
                   g_stage[9] = c00; g_stage[10] = c01; g_stage[11] = c02; g_stage[12] = c11; g_stage[13] = c12; g_stage[14] = c22; }

    // CalcCovariance2D (:56-90)
    f3 viewPos = { mul_row(P.matrix_mv, 0, splat.pos), mul_row(P.matrix_mv, 1, splat.pos), mul_row(P.matrix_mv, 2, splat.pos) };
    const float aspect = P.proj_m00 / P.proj_m11;
    const float tanFovX = 1.0f / P.proj_m00;
    const float tanFovY = 1.0f / (P.proj_m11 * aspect);
    const float limX = 1.3f * tanFovX, limY = 1.3f * tanFovY;
    // the divisions by viewPos.z (:67-68,73-74) share one reciprocal and 1 / z^2 = rz * rz: what oracle/_ref's fused build (and a
    // shader compiler's rcp) makes of them
    const float rz = 1.0f / viewPos.z;
    viewPos.x = fminf(fmaxf(viewPos.x * rz, -limX), limX) * viewPos.z;
    viewPos.y = fminf(fmaxf(viewPos.y * rz, -limY), limY) * viewPos.z;
    const float focal = P.screen_w * P.proj_m00 / 2.0f;
    const float rzz = rz * rz;
    const float J00 = focal * rz, J02 = -(focal * viewPos.x) * rzz;
    const float J11 = focal * rz, J12 = -(focal * viewPos.y) * rzz;
    const float* W = P.matrix_mv;
    // T = J * W (rows 0,1; J's zero entries dropped)
    float T[2][3];
    for (int j = 0; j < 3; ++j) {
        T[0][j] = fmaf(J02, W[2 * 4 + j], J00 * W[0 * 4 + j]);
        T[1][j] = fmaf(J12, W[2 * 4 + j], J11 * W[1 * 4 + j]);
    }
    const float V[3][3] = { { c00, c01, c02 }, { c01, c11, c12 }, { c02, c12, c22 } };
    // cov = T * (V * T^T)
    float VT[3][2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) VT[i][j] = fmaf(V[i][2], T[j][2], fmaf(V[i][1], T[j][1], V[i][0] * T[j][0]));
    auto covf = [&](int i, int j) { return fmaf(T[i][2], VT[2][j], fmaf(T[i][1], VT[1][j], T[i][0] * VT[0][j])); };
    const float cov00 = covf(0, 0) + 0.3f, cov01 = covf(0, 1), cov11 = covf(1, 1) + 0.3f;

    // DecomposeCovariance (SplatUtilities.compute:149-159, the live #else branch)
    const float diag1 = cov00, diag2 = cov11, offDiag = cov01;
    const float mid = 0.5f * (diag1 + diag2);
    const float hx = (diag1 - diag2) / 2.0f;
    const float radius = sqrtf(dot2(hx, offDiag, hx, offDiag));
    const float lambda1 = mid + radius;
    const float lambda2 = fmaxf(mid - radius, 0.1f);
    float dvx = offDiag, dvy = lambda1 - diag1;
    const float invLen = 1.0f / sqrtf(dot2(dvx, dvy, dvx, dvy));        // normalize(): 0,0 -> NaN (splat vanishes)
    dvx *= invLen; dvy *= invLen;
    dvy = -dvy;
    const float maxSize = 4096.0f;
    const float s1 = fminf(sqrtf(2.0f * lambda1), maxSize), s2 = fminf(sqrtf(2.0f * lambda2), maxSize);
    view.axis1[0] = s1 * dvx;  view.axis1[1] = s1 * dvy;
    view.axis2[0] = s2 * dvy;  view.axis2[1] = s2 * (-dvx);
    if (g_stage) { g_stage[15] = cov00; g_stage[16] = cov01; g_stage[17] = cov11;
                   g_stage[18] = view.axis1[0]; g_stage[19] = view.axis1[1]; g_stage[20] = view.axis2[0]; g_stage[21] = view.axis2[1]; }

    // view direction + SH (SplatUtilities.compute:241-248)
    const f3 worldViewDir = { P.cam_pos_world[0] - centerWorldPos.x, P.cam_pos_world[1] - centerWorldPos.y, P.cam_pos_world[2] - centerWorldPos.z };
    f3 objViewDir = { mul3_row(P.matrix_world_to_object, 0, worldViewDir), mul3_row(P.matrix_world_to_object, 1, worldViewDir), mul3_row(P.matrix_world_to_object, 2, worldViewDir) };
    const float invN = 1.0f / sqrtf(dot3(objViewDir, objViewDir));
    objViewDir = { objViewDir.x * invN, objViewDir.y * invN, objViewDir.z * invN };
    const float dx = -objViewDir.x, dy = -objViewDir.y, dz = -objViewDir.z;     // ShadeSH: dir *= -1
    const float* shp = &splat.sh[0].x;
    const bool onlySH = P.sh_only != 0;
    const float r = sh_channel(splat.col.x, shp + 0, dx, dy, dz, (int)P.sh_order, onlySH);
    const float g = sh_channel(splat.col.y, shp + 1, dx, dy, dz, (int)P.sh_order, onlySH);
    const float b = sh_channel(splat.col.z, shp + 2, dx, dy, dz, (int)P.sh_order, onlySH);
    const float al = fminf(splat.opacity * P.opacity_scale, 65000.0f);
    view.color[0] = ((uint32_t)f32tof16(r) << 16) | f32tof16(g);
    view.color[1] = ((uint32_t)f32tof16(b) << 16) | f32tof16(al);
    return view;
}

// ---------------------------------------------------------------------------------------------
// rasteriser restatement: RenderGaussianSplats.shader:35-108 + fixed-function state :10-12
// ---------------------------------------------------------------------------------------------
struct Prepared {
    float cx, cy;           // splat centre in pixels (y down)
    float a1x, a1y, a2x, a2y;
    float u1x, u1y, u2x, u2y; // axis_k / |axis_k|^2
    float r, g, b, a;       // colour as the vertex shader unpacks it (f16 -> f32)
    float w;                // view depth of the centre (clip.w): the depth every fragment of the quad is tested with
    int x0, x1, y0, y1;     // pixel rect of the quad's bounding box, clamped to the screen (x0>x1 => nothing)
    int bx0, bx1, by0, by1; // pixel rect of the *tight* footprint used by the shipped binning kernel (bx0 > bx1 => nothing)
    int tx0, tx1, ty0, ty1; // ... as a rectangle of tiles of the shape set by gso_set_tile_shape (default 16x16)
    bool valid;
};
// The compositor tile of the build under test (a performance parameter of the product: 16x16, 32x16 or 32x32 pixels).  Only the
// (tile, splat) pair COUNT depends on it; frames, records and pixel rectangles do not.
int g_tile_wl = 4, g_tile_hl = 4;

// ln(x), x positive normal, from fp32 operations only (same bits on any IEEE machine; the binning's footprint must not
// depend on a libm): mantissa reduced to [0.707, 1.414), atanh series.  |error| < 1e-6.
inline float log_det(float x) {
    uint32_t u; std::memcpy(&u, &x, 4);
    float e = (float)((int)(u >> 23) - 127);
    const uint32_t mu = (u & 0x7fffffu) | 0x3f800000u;
    float m; std::memcpy(&m, &mu, 4);
    if (m > 1.41421356f) { m *= 0.5f; e += 1.0f; }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    const float p = fmaf(t2, fmaf(t2, fmaf(t2, 1.0f / 7.0f, 0.2f), 1.0f / 3.0f), 1.0f);
    return fmaf(e, 0.69314718f, (2.0f * t) * p);
}

inline bool finitef(float v) { return std::isfinite(v); }

// HLSL exp(x).  DXC lowers it to the DXIL Exp opcode, which is base 2: exp(x) = exp2(x * log2(e)), the product rounded to
// fp32 -- and that first rounding, not the exp2 unit, is the larger part of the result's error for the |x| of a gaussian's
// tail (|x| * 2^-24 relative).  Canonical form: the fp32 product, then the correctly rounded exp2.
// What remains between this and a GPU is the <= 1 ulp of its exp2 instruction (v_exp_f32 on gfx950).
inline float exp_canon(float x) {
    const float y = x * 1.44269504088896340736f;
    return (float)std::exp2((double)y);
}

// The discard decision of frag() (RenderGaussianSplats.shader:100), made identical on the CPU and the GPU: when the alpha comes
// out within 8 ulps of 1/255 it is recomputed from an exp2 built from fp32 operations only and the decision is taken on that
// (gs_device_math.h: Exp2Det / DecideAlpha, restated here; tests/test_host_math.py keeps the two in step).  Outside the window
// this is exactly saturate(exp(power) * a) >= 1/255.
constexpr uint32_t kAlphaThresholdBits = 0x3B808081u, kAlphaWindow = 16u, kAlphaWindowLo = kAlphaThresholdBits - kAlphaWindow / 2u;
inline float exp2_det(float y) {
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
inline float fragment_alpha(float power, float a, bool windowed, bool& live) {
    const float y = power * 1.44269504088896340736f;
    float alpha = saturatef((float)std::exp2((double)y) * a);
    uint32_t bits; std::memcpy(&bits, &alpha, 4);
    if (windowed && bits - kAlphaWindowLo < kAlphaWindow) alpha = saturatef(exp2_det(y) * a);
    live = alpha >= 1.0f / 255.0f;
    return alpha;
}

// Shared definition of "is this splat drawn at all, and where": see DESIGN.md "compositor semantics".
Prepared prepare(const ViewData& v, const gs_frame_params& P) {
    Prepared p; std::memset(&p, 0, sizeof(p));
    p.x0 = 1; p.x1 = 0; p.tx0 = 1; p.tx1 = 0; p.bx0 = 1; p.bx1 = 0;
    const float W = P.screen_w, H = P.screen_h;
    const float w = v.pos[3];
    if (!(w > 0.0f)) return p;                                   // vert: behindCam -> NaN vertex -> primitive discarded
    if (!(w >= P.near_clip && w <= P.far_clip)) return p;        // all 4 vertices share z/w: depth-clipped as a whole
    if (!(finitef(v.axis1[0]) && finitef(v.axis1[1]) && finitef(v.axis2[0]) && finitef(v.axis2[1]))) return p;
    p.r = f16tof32(v.color[0] >> 16); p.g = f16tof32(v.color[0]); p.b = f16tof32(v.color[1] >> 16); p.a = f16tof32(v.color[1]);
    if (!(p.a >= 1.0f / 255.0f)) return p;                       // alpha = saturate(e*a) <= a < 1/255: every fragment discards
    p.w = w;
    const float invw = 1.0f / w;
    p.cx = fmaf(0.5f * (v.pos[0] * invw), W, 0.5f * W);          // (0.5 + 0.5*ndc.x) * W
    p.cy = fmaf(-0.5f * (v.pos[1] * invw), H, 0.5f * H);         // (0.5 - 0.5*ndc.y) * H   (image rows top-down)
    if (!(finitef(p.cx) && finitef(p.cy))) return p;
    p.a1x = v.axis1[0]; p.a1y = v.axis1[1]; p.a2x = v.axis2[0]; p.a2y = v.axis2[1];
    const float inv1 = 1.0f / dot2(p.a1x, p.a1y, p.a1x, p.a1y);
    const float inv2 = 1.0f / dot2(p.a2x, p.a2y, p.a2x, p.a2y);
    if (!(finitef(inv1) && finitef(inv2))) return p;
    p.u1x = p.a1x * inv1; p.u1y = p.a1y * inv1; p.u2x = p.a2x * inv2; p.u2y = p.a2y * inv2;
    // quad = c + qx*axis1 + qy*axis2, q in [-2,2]^2  ->  bounding box half extents
    const float exr = 2.0f * (fabsf(p.a1x) + fabsf(p.a2x));
    const float eyr = 2.0f * (fabsf(p.a1y) + fabsf(p.a2y));
    auto pix_range = [](float c, float e, float size, int& lo, int& hi) {
        float flo = ceilf((c - e) - 0.5f), fhi = floorf((c + e) - 0.5f);
        flo = fmaxf(flo, 0.0f); fhi = fminf(fhi, size - 1.0f);
        if (!(flo <= fhi)) { lo = 1; hi = 0; return; }
        lo = (int)flo; hi = (int)fhi;
    };
    const float slack = 0.01f;
    pix_range(p.cx, exr + slack, W, p.x0, p.x1);
    pix_range(p.cy, eyr + slack, H, p.y0, p.y1);
    if (p.x0 > p.x1 || p.y0 > p.y1) { p.x0 = 1; p.x1 = 0; return p; }
    // tight footprint of the shipped binning: quad  INTERSECT  {exp(-|q|^2)*a >= 1/255} = disc |q|^2 <= ln(255 a)
    const float r2 = fmaf(log_det(255.0f * p.a), 1.0001f, 1.0e-3f);
    const float rr = sqrtf(fmaxf(r2, 0.0f));
    const float exe = rr * sqrtf(dot2(p.a1x, p.a2x, p.a1x, p.a2x));
    const float eye = rr * sqrtf(dot2(p.a1y, p.a2y, p.a1y, p.a2y));
    int bx0, bx1, by0, by1;
    pix_range(p.cx, fminf(exr, exe) + slack, W, bx0, bx1);
    pix_range(p.cy, fminf(eyr, eye) + slack, H, by0, by1);
    if (bx0 <= bx1 && by0 <= by1) {
        p.bx0 = bx0; p.bx1 = bx1; p.by0 = by0; p.by1 = by1;
        p.tx0 = bx0 >> g_tile_wl; p.tx1 = bx1 >> g_tile_wl; p.ty0 = by0 >> g_tile_hl; p.ty1 = by1 >> g_tile_hl;
    }
    p.valid = true;
    return p;
}

} // namespace

// =================================================================================================
// C interface (loaded with ctypes by tests / bench cpu_baseline)
// =================================================================================================
extern "C" {

int32_t gso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void gso_set_num_threads(int32_t n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

void gso_set_canon(int32_t flags) { g_canon = flags; }
int32_t gso_get_canon(void) { return g_canon; }

// half conversion exposed for the f16 known-answer tests
uint16_t gso_f32tof16(float f) { return f32tof16(f); }
float gso_f16tof32(uint16_t h) { return f16tof32(h); }
uint16_t gso_f64tof16(double d) { return f64tof16(d); }
float gso_blend_f16(float src, float t, float dst) { return blend_f16(src, t, dst); }

// SplatUtilities.compute:59-67 CSSetIndices
void gso_set_indices(uint32_t* order, uint32_t n) { for (uint32_t i = 0; i < n; ++i) order[i] = i; }

// SplatUtilities.compute:69-82 CSCalcDistances; `matrix_sort` is the _MatrixMV of GaussianSplatRenderer.cs:629
void gso_calc_distances(const gs_asset_desc* d, const uint32_t* order, const float* matrix_sort, uint32_t* keys) {
    const Asset a = make_asset(d);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)a.n; ++i) {
        const uint32_t origIdx = order[i];
        const f3 pos = LoadSplatPos(a, origIdx);
        const float z = mul_row(matrix_sort, 2, pos);
        keys[i] = FloatToSortableUint(z);
    }
}

// GpuSorting.Dispatch semantics (GpuSorting.cs:142-198; DeviceRadixSort.hlsl): stable ascending sort of
// (key, payload) pairs, comparing the low `key_bits` bits.  Implemented as a plain LSD counting sort.
void gso_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t n, uint32_t key_bits) {
    std::vector<uint32_t> k2(n), v2(n);
    uint32_t *ks = keys, *vs = vals, *kd = k2.data(), *vd = v2.data();
    for (uint32_t shift = 0; shift < key_bits; shift += 8) {
        uint64_t hist[257] = {0};
        const uint32_t bits = std::min(8u, key_bits - shift), mask = (1u << bits) - 1u;
        for (uint32_t i = 0; i < n; ++i) hist[((ks[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (uint32_t i = 0; i < n; ++i) { const uint64_t p = hist[(ks[i] >> shift) & mask]++; kd[p] = ks[i]; vd[p] = vs[i]; }
        std::swap(ks, kd); std::swap(vs, vd);
    }
    if (ks != keys) { std::memcpy(keys, ks, (size_t)n * 4); std::memcpy(vals, vs, (size_t)n * 4); }
}

// independent second opinion for the sort tests: std::stable_sort on indices
void gso_stable_sort_reference(const uint32_t* keys, uint32_t* perm_out, uint32_t n, uint32_t key_bits) {
    std::iota(perm_out, perm_out + n, 0u);
    const uint32_t mask = key_bits >= 32 ? 0xffffffffu : ((1u << key_bits) - 1u);
    std::stable_sort(perm_out, perm_out + n, [&](uint32_t x, uint32_t y) { return (keys[x] & mask) < (keys[y] & mask); });
}

// LoadSplatData exposed for the codec known-answer tests: out[0..58] =
// pos3, rot4 (xyzw), scale3, opacity, col3, sh 15x3
void gso_decode_splat(const gs_asset_desc* d, uint32_t idx, float* out) {
    const Asset a = make_asset(d);
    const SplatData s = LoadSplatData(a, idx);
    float* o = out;
    *o++ = s.pos.x; *o++ = s.pos.y; *o++ = s.pos.z;
    *o++ = s.rot.x; *o++ = s.rot.y; *o++ = s.rot.z; *o++ = s.rot.w;
    *o++ = s.scale.x; *o++ = s.scale.y; *o++ = s.scale.z;
    *o++ = s.opacity;
    *o++ = s.col.x; *o++ = s.col.y; *o++ = s.col.z;
    for (int k = 0; k < 15; ++k) { *o++ = s.sh[k].x; *o++ = s.sh[k].y; *o++ = s.sh[k].z; }
}
void gso_decode_all(const gs_asset_desc* d, float* out /* n x 59 */) {
    const Asset a = make_asset(d);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)a.n; ++i) gso_decode_splat(d, (uint32_t)i, out + i * 59);
}
void gso_bc7_decode_block(const uint8_t* block, uint8_t* out64) { uint8_t px[16][4]; bc7_decode_block(block, px); std::memcpy(out64, px, 64); }
void gso_pixel_index(uint32_t idx, uint32_t* xy) { SplatIndexToPixelIndex(idx, xy[0], xy[1]); }

// SplatUtilities.compute:189-252 CSCalcViewData over all splats; deleted_bits may be null (_SplatBitsValid = 0)
void gso_calc_view_ex(const gs_asset_desc* d, const gs_frame_params* P, const gs_cutout* cutouts, uint32_t cutout_count,
                      const uint32_t* deleted_bits, void* view_out) {
    const Asset a = make_asset(d);
    ViewData* out = (ViewData*)view_out;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)a.n; ++i) out[i] = CalcViewDataOne(a, *P, (uint32_t)i, cutouts, cutout_count, deleted_bits);
}
// the covariance path's intermediate stages of one splat (layout of oracle/ref_build's gsr_cs_cov_stages)
void gso_cov_stages(const gs_asset_desc* d, const gs_frame_params* P, uint32_t idx, float* out22) {
    const Asset a = make_asset(d);
    g_stage = out22;
    (void)CalcViewDataOne(a, *P, idx);
    g_stage = nullptr;
}
void gso_calc_view(const gs_asset_desc* d, const gs_frame_params* P, void* view_out) { gso_calc_view_ex(d, P, nullptr, 0, nullptr, view_out); }

// tile shape (pixels) the pair count of gso_draw* refers to: 16x16 (default), 32x16 or 32x32 -- what gs_frame_stats.tile_w/h reports
void gso_set_tile_shape(int32_t tile_w, int32_t tile_h) {
    g_tile_wl = tile_w >= 32 ? 5 : 4; g_tile_hl = tile_h >= 32 ? 5 : 4;
}

// prepare() of every splat in index order, in the layout of gs_renderer_download_raster_records (include/gsplat_c.h):
// recs N x 8 u32 (written only for splats that reach a tile), rects N x 2 u32, vis ceil(N/64) u64.
void gso_raster_records(const void* view_in, uint32_t n, const gs_frame_params* P, uint32_t* recs, uint32_t* rects, uint64_t* vis) {
    const ViewData* view = (const ViewData*)view_in;
    const int64_t words = ((int64_t)n + 63) / 64;
#pragma omp parallel for schedule(static)
    for (int64_t wd = 0; wd < words; ++wd) {
        uint64_t bits = 0;
        for (int64_t i = wd * 64; i < std::min<int64_t>((wd + 1) * 64, n); ++i) {
            const Prepared p = prepare(view[i], *P);
            const bool visible = p.valid && p.tx0 <= p.tx1;
            rects[i * 2] = rects[i * 2 + 1] = 0u;
            std::memset(recs + i * 8, 0, 32);
            if (!visible) continue;
            bits |= 1ull << (i & 63);
            rects[i * 2] = (uint32_t)p.bx0 | ((uint32_t)p.by0 << 16);                         // inclusive pixel rectangle, +1 on the far corner
            rects[i * 2 + 1] = (uint32_t)(p.bx1 + 1) | ((uint32_t)(p.by1 + 1) << 16);
            const float f[6] = { p.cx, p.cy, p.a1x, p.a1y, p.a2x, p.a2y };
            std::memcpy(recs + i * 8, f, 24);
            recs[i * 8 + 6] = view[i].color[0]; recs[i * 8 + 7] = view[i].color[1];
        }
        vis[wd] = bits;
    }
}

// RenderGaussianSplats.shader:79-108 frag() for an unselected splat, exposed for tests/test_ref_parity.py: q = the interpolated
// i.pos, col = i.col (rgb, opacity).  Returns 1 for a discarded fragment, else out4 = (rgb * alpha, alpha).
// windowed = 0: the canonical arithmetic alone (what oracle/_ref's fused build computes); 1: with the deterministic decision inside
// the 16-ulp window around 1/255, as gso_draw and the HIP blend evaluate it.
int32_t gso_fragment(const float* q, const float* col, float* out4, int32_t windowed) {
    const float power = -fmaf(q[1], q[1], q[0] * q[0]);
    bool live;
    const float alpha = fragment_alpha(power, col[3], windowed != 0, live);
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0f;
    if (!live) return 1;
    out4[0] = col[0] * alpha; out4[1] = col[1] * alpha; out4[2] = col[2] * alpha; out4[3] = alpha;
    return 0;
}

// the native (un-windowed) alpha and y = power * log2(e) of a fragment, for tests/test_host_math.py
float gso_fragment_native(const float* q, float a, float* y_out) {
    const float power = -fmaf(q[1], q[1], q[0] * q[0]);
    *y_out = power * 1.44269504088896340736f;
    bool live;
    return fragment_alpha(power, a, false, live);
}

// The DrawProcedural of GaussianSplatRenderer.cs:156-166 with RenderGaussianSplats.shader, executed splat by
// splat in order[] (instance order), "Blend OneMinusDstAlpha One" into an RGBA16F target (rt, W*H*4 halfs,
// row 0 = top).  mode 0: the ROP rounds to fp16 after every blend; mode 1: fp32 accumulation, a pixel stops
// once 1-A < 1/4096 (the shipped "fast" mode), rounded to fp16 once at the end.
// tile_pairs_out (optional) = number of (tile, splat) overlaps of the shipped binning's footprint, tiles of gso_set_tile_shape.
// Parallel over row bands; each band walks all splats in order, so the result is independent of thread count.
// `win` = {x0, y0, x1, y1} inclusive pixel window (fragments outside are not evaluated; rt is still the full W x H target),
// so a 50 M-splat / 4K frame can be checked on a crop in seconds.  The pair / visible counts always cover the whole screen.
static int32_t draw_impl(const ViewData* view, const uint32_t* order, uint32_t n, const gs_frame_params* P, int32_t mode,
                         uint16_t* rt, uint64_t* tile_pairs_out, uint32_t* visible_out, const int win[4], const float* scene_depth) {
    const int W = (int)P->screen_w, H = (int)P->screen_h;
    const int wx0 = std::max(0, win[0]), wy0 = std::max(0, win[1]), wx1 = std::min(W - 1, win[2]), wy1 = std::min(H - 1, win[3]);
    std::vector<Prepared> prep(n);
    uint64_t pairs = 0; uint32_t visible = 0;
#pragma omp parallel for schedule(static) reduction(+ : pairs, visible)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        prep[i] = prepare(view[order[i]], *P);
        if (prep[i].valid && prep[i].tx0 <= prep[i].tx1) {
            pairs += (uint64_t)(prep[i].tx1 - prep[i].tx0 + 1) * (uint64_t)(prep[i].ty1 - prep[i].ty0 + 1);
            visible++;
        }
    }
    if (tile_pairs_out) *tile_pairs_out = pairs;
    if (visible_out) *visible_out = visible;
    if (wx0 > wx1 || wy0 > wy1) return 0;

    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    // Row bands; every band gets the list of draw positions whose quad touches it, in draw order (a counting sort over the
    // bands), so the work is O(n + overlaps) whatever the band count and the result is independent of the thread count.
    const int winH = wy1 - wy0 + 1;
    const int bands = std::max(1, std::min(winH, nthreads * 4));
    auto band_of = [&](int y) { return (int)(((int64_t)(y - wy0) * bands) / winH); };          // y in [wy0, wy1]
    std::vector<uint64_t> start((size_t)bands + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const Prepared& p = prep[i];
        if (!p.valid || p.x0 > p.x1 || p.x1 < wx0 || p.x0 > wx1) continue;
        const int y0 = std::max(p.y0, wy0), y1 = std::min(p.y1, wy1);
        if (y0 > y1) continue;
        for (int b = band_of(y0); b <= band_of(y1); ++b) start[(size_t)b + 1]++;
    }
    for (int b = 0; b < bands; ++b) start[(size_t)b + 1] += start[b];
    std::vector<uint32_t> list(start[bands]);
    {
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (uint32_t i = 0; i < n; ++i) {
            const Prepared& p = prep[i];
            if (!p.valid || p.x0 > p.x1 || p.x1 < wx0 || p.x0 > wx1) continue;
            const int y0 = std::max(p.y0, wy0), y1 = std::min(p.y1, wy1);
            if (y0 > y1) continue;
            for (int b = band_of(y0); b <= band_of(y1); ++b) list[cur[b]++] = i;
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int band = 0; band < bands; ++band) {
        // rows of this band: the y with band_of(y) == band
        int yb0 = wy0 + (int)(((int64_t)band * winH + bands - 1) / bands), yb1 = wy0 + (int)(((int64_t)(band + 1) * winH + bands - 1) / bands) - 1;
        if (yb0 > yb1) continue;
        const int rows = yb1 - yb0 + 1;
        std::vector<float> acc((size_t)rows * W * 4);
        for (int y = 0; y < rows; ++y)
            for (int x = wx0; x <= wx1; ++x)
                for (int c = 0; c < 4; ++c) acc[((size_t)y * W + x) * 4 + c] = f16tof32(rt[((size_t)(yb0 + y) * W + x) * 4 + c]);
        for (uint64_t li = start[band]; li < start[(size_t)band + 1]; ++li) {
            const Prepared& p = prep[list[li]];
            const int y0 = std::max(p.y0, yb0), y1 = std::min(p.y1, yb1);
            const int x0 = std::max(p.x0, wx0), x1 = std::min(p.x1, wx1);
            for (int py = y0; py <= y1; ++py) {
                const float dy = ((float)py + 0.5f) - p.cy;
                float* row = &acc[(size_t)(py - yb0) * W * 4];
                for (int px = x0; px <= x1; ++px) {
                    const float dx = ((float)px + 0.5f) - p.cx;
                    // interpolated quad coordinate (i.pos of the v2f): q = [axis1 axis2]^-1 * delta, axes orthogonal
                    const float q1 = fmaf(dy, p.u1y, dx * p.u1x);
                    const float q2 = fmaf(dy, p.u2y, dx * p.u2x);
                    if (!(fabsf(q1) <= 2.0f && fabsf(q2) <= 2.0f)) continue;     // outside the quad
                    // depth test against the scene (ZTest LEqual, ZWrite Off: RenderGaussianSplats.shader:10 + the camera's depth
                    // attachment, GaussianSplatRenderer.cs:195): all four vertices share the centre's depth, so the whole quad
                    // has the view depth clip.w; scene_depth holds the opaque scene's view depth per pixel
                    if (scene_depth && !(p.w <= scene_depth[(size_t)py * W + px])) continue;
                    float* d = row + (size_t)px * 4;
                    if (mode == 1 && (1.0f - d[3]) < (1.0f / 4096.0f)) continue; // fast mode: pixel finished
                    const float power = -fmaf(q2, q2, q1 * q1);                 // frag: -dot(i.pos, i.pos)
                    bool live;
                    const float alpha = fragment_alpha(power, p.a, true, live);  // saturate(exp(power) * a), discard below 1/255
                    if (!live) continue;                                         // discard
                    const float t = 1.0f - d[3];                                 // OneMinusDstAlpha
                    const float sr = p.r * alpha, sg = p.g * alpha, sb = p.b * alpha;   // fragment output (fp32): rgb*alpha, alpha
                    if (mode == 0) {        // exact: one RTNE to the fp16 storage format per blend
                        d[0] = blend_f16(sr, t, d[0]); d[1] = blend_f16(sg, t, d[1]); d[2] = blend_f16(sb, t, d[2]); d[3] = blend_f16(alpha, t, d[3]);
                    } else {                // fast: fp32 accumulation, rounded once at the end of the draw
                        d[0] = fmaf(sr, t, d[0]); d[1] = fmaf(sg, t, d[1]); d[2] = fmaf(sb, t, d[2]); d[3] = fmaf(alpha, t, d[3]);
                    }
                }
            }
        }
        for (int y = 0; y < rows; ++y)
            for (int x = wx0; x <= wx1; ++x)
                for (int c = 0; c < 4; ++c) rt[((size_t)(yb0 + y) * W + x) * 4 + c] = f32tof16(acc[((size_t)y * W + x) * 4 + c]);
    }
    return 0;
}

int32_t gso_draw(const void* view_in, const uint32_t* order, uint32_t n, const gs_frame_params* P, int32_t mode,
                 uint16_t* rt, uint64_t* tile_pairs_out, uint32_t* visible_out) {
    const int win[4] = { 0, 0, (int)P->screen_w - 1, (int)P->screen_h - 1 };
    return draw_impl((const ViewData*)view_in, order, n, P, mode, rt, tile_pairs_out, visible_out, win, nullptr);
}
// window = {x0, y0, x1, y1} inclusive, or NULL for the whole target; scene_depth = W*H view depths of the opaque scene, or NULL
int32_t gso_draw_ex(const void* view_in, const uint32_t* order, uint32_t n, const gs_frame_params* P, int32_t mode,
                    uint16_t* rt, uint64_t* tile_pairs_out, uint32_t* visible_out, const int32_t* window, const float* scene_depth) {
    int win[4] = { 0, 0, (int)P->screen_w - 1, (int)P->screen_h - 1 };
    if (window) for (int k = 0; k < 4; ++k) win[k] = window[k];
    return draw_impl((const ViewData*)view_in, order, n, P, mode, rt, tile_pairs_out, visible_out, win, scene_depth);
}

// RenderMode.DebugPoints / DebugPointIndices (GaussianDebugRenderPoints.shader:28-63; GaussianSplatRenderer.cs:126-131,148-161):
// instanced quads in splat-INDEX order, `size` pixels wide around the projected centre, opaque (no blend), ZWrite On with the
// default ZTest LEqual (a later instance at the same depth overwrites).  Depth is compared as the view depth clip.w
// (monotonic in the depth-buffer value of a perspective camera).  Sequential over splats: the reference order.
void gso_draw_debug_points(const gs_asset_desc* d, const gs_frame_params* P, int32_t display_index, float size, uint16_t* rt, const float* scene_depth) {
    const Asset a = make_asset(d);
    const int W = (int)P->screen_w, H = (int)P->screen_h;
    std::vector<float> z((size_t)W * H, INFINITY);
    if (scene_depth) std::memcpy(z.data(), scene_depth, (size_t)W * H * 4);
    const float h = 0.5f * size;
    for (uint32_t idx = 0; idx < a.n; ++idx) {
        const SplatData sp = LoadSplatData(a, idx);
        const f3 wp = { mul_row(P->matrix_object_to_world, 0, sp.pos), mul_row(P->matrix_object_to_world, 1, sp.pos), mul_row(P->matrix_object_to_world, 2, sp.pos) };
        const float cxc = mul_row(P->matrix_vp, 0, wp), cyc = mul_row(P->matrix_vp, 1, wp), w = mul_row(P->matrix_vp, 3, wp);
        if (!(w >= P->near_clip && w <= P->far_clip)) continue;           // the quad lies at one depth: clipped as a whole
        const float invw = 1.0f / w;
        const float cx = fmaf(0.5f * (cxc * invw), (float)W, 0.5f * (float)W), cy = fmaf(-0.5f * (cyc * invw), (float)H, 0.5f * (float)H);
        if (!(std::isfinite(cx) && std::isfinite(cy))) continue;
        f3 col = { saturatef(sp.col.x), saturatef(sp.col.y), saturatef(sp.col.z) };
        if (display_index) {
            const float f = (float)idx / (float)a.n;
            const float r = f * 100.0f, g = f * 10.0f;
            col = { r - floorf(r), g - floorf(g), f };
        }
        for (int py = 0; py < H; ++py) {
            const float fy = (float)py + 0.5f;
            if (!(fy >= cy - h && fy < cy + h)) continue;                  // top-left rule
            for (int px = 0; px < W; ++px) {
                const float fx = (float)px + 0.5f;
                if (!(fx >= cx - h && fx < cx + h)) continue;
                float& zz = z[(size_t)py * W + px];
                if (!(w <= zz)) continue;                                  // ZTest LEqual
                zz = w;                                                    // ZWrite On
                uint16_t* o = rt + ((size_t)py * W + px) * 4;
                o[0] = f32tof16(col.x); o[1] = f32tof16(col.y); o[2] = f32tof16(col.z); o[3] = f32tof16(1.0f);
            }
        }
    }
}

// RenderMode.DebugBoxes / DebugChunkBounds (GaussianDebugRenderBoxes.shader:37-97; GaussianSplatRenderer.cs:126-131,156-166).
// vert: a cube [-1,1]^3 per instance.  Splat boxes: instance -> _OrderBuffer[instance]; M = (float3x3)unity_ObjectToWorld *
// CalcMatrixFromRotationScale(rot, scale * _SplatScale); worldPos = ObjectToWorld(pos) + mul(M, localPos) * 2; colour
// saturate(col), alpha saturate(opacity * _SplatOpacityScale).  Chunk boxes: corners lerp(posMin, posMax, {0,1}^3) through
// ObjectToWorld, palette colour, alpha 0.1, instance = chunk index.  frag: (rgb * a, a), Blend OneMinusDstAlpha One, ZWrite Off,
// ZTest LEqual, Cull Front -- and the cube's 36 indices (GaussianSplatRenderer.cs:410-418) wind its outside faces counter-
// clockwise, which Unity treats as back faces: the faces turned TOWARDS the camera are the ones drawn (the far ones for a
// mirrored transform).  The rasteriser's coverage / depth per pixel centre is restated as a ray / box intersection in the box's
// own space: ray through the pixel centre p(t) = o + t d with d = ax R0 + ay R1 + R2 (R0 = VP row 0 / P00,
// R1 = VP row 1 / P11, R2 = VP row 3), so that t is the view depth; l(t) = Binv (p(t) - c); slab test against [-1,1]^3.
struct BoxO { float inv[9], lo[3], r, g, b, a; bool mirrored; int x0, x1, y0, y1; bool ok; };

static float inverse3(const float* b, float* inv) {
    const float c00 = fmaf(b[4], b[8], -(b[5] * b[7])), c01 = fmaf(b[5], b[6], -(b[3] * b[8])), c02 = fmaf(b[3], b[7], -(b[4] * b[6]));
    const float det = fmaf(b[2], c02, fmaf(b[1], c01, b[0] * c00));
    const float r = 1.0f / det;
    inv[0] = c00 * r; inv[1] = fmaf(b[2], b[7], -(b[1] * b[8])) * r; inv[2] = fmaf(b[1], b[5], -(b[2] * b[4])) * r;
    inv[3] = c01 * r; inv[4] = fmaf(b[0], b[8], -(b[2] * b[6])) * r; inv[5] = fmaf(b[2], b[3], -(b[0] * b[5])) * r;
    inv[6] = c02 * r; inv[7] = fmaf(b[1], b[6], -(b[0] * b[7])) * r; inv[8] = fmaf(b[0], b[4], -(b[1] * b[3])) * r;
    return det;
}

int32_t gso_draw_debug_boxes(const gs_asset_desc* d, const uint32_t* order, const gs_frame_params* P, int32_t chunks, int32_t mode,
                             uint16_t* rt, const float* scene_depth) {
    const Asset a = make_asset(d);
    const int W = (int)P->screen_w, H = (int)P->screen_h;
    const float Wf = P->screen_w, Hf = P->screen_h;
    const uint32_t count = chunks ? a.chunkCount : a.n;
    if (count == 0) return 0;
    float R0[3], R1[3], R2[3];
    for (int k = 0; k < 3; ++k) { R0[k] = P->matrix_vp[k] / P->proj_m00; R1[k] = P->matrix_vp[4 + k] / P->proj_m11; R2[k] = P->matrix_vp[12 + k]; }
    const float ox = P->cam_pos_world[0], oy = P->cam_pos_world[1], oz = P->cam_pos_world[2];
    std::vector<BoxO> boxes(count);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        BoxO& bx = boxes[i];
        bx.ok = false;
        const uint32_t idx = chunks ? (uint32_t)i : order[i];          // draw position i -> instance
        float c[3], B[9], r, g, b, al;
        const float* o2w = P->matrix_object_to_world;
        if (!chunks) {
            const SplatData sp = LoadSplatData(a, idx);
            const float sx = sp.scale.x * P->splat_scale, sy = sp.scale.y * P->splat_scale, sz = sp.scale.z * P->splat_scale;
            for (int k = 0; k < 3; ++k) c[k] = mul_row(o2w, k, sp.pos);
            const float x = sp.rot.x, y = sp.rot.y, z = sp.rot.z, w = sp.rot.w;
            const float m1[9] = { fmaf(-2.0f, fmaf(z, z, y * y), 1.0f) * sx, (2.0f * fmaf(-w, z, x * y)) * sy, (2.0f * fmaf(w, y, x * z)) * sz,
                                  (2.0f * fmaf(w, z, x * y)) * sx, fmaf(-2.0f, fmaf(z, z, x * x), 1.0f) * sy, (2.0f * fmaf(-w, x, y * z)) * sz,
                                  (2.0f * fmaf(-w, y, x * z)) * sx, (2.0f * fmaf(w, x, y * z)) * sy, fmaf(-2.0f, fmaf(y, y, x * x), 1.0f) * sz };
            for (int ii = 0; ii < 3; ++ii)
                for (int j = 0; j < 3; ++j) B[ii * 3 + j] = fmaf(o2w[ii * 4 + 2], m1[6 + j], fmaf(o2w[ii * 4 + 1], m1[3 + j], o2w[ii * 4] * m1[j])) * 2.0f;
            r = saturatef(sp.col.x); g = saturatef(sp.col.y); b = saturatef(sp.col.z);
            al = saturatef(sp.opacity * P->opacity_scale);
        } else {
            const Chunk ck = load_chunk(a, idx);
            const float mn[3] = { ck.posX[0], ck.posY[0], ck.posZ[0] }, mx[3] = { ck.posX[1], ck.posY[1], ck.posZ[1] };
            float mid[3], half[3];
            for (int k = 0; k < 3; ++k) { mid[k] = (mn[k] + mx[k]) * 0.5f; half[k] = (mx[k] - mn[k]) * 0.5f; }
            for (int k = 0; k < 3; ++k) c[k] = mul_row(o2w, k, f3{ mid[0], mid[1], mid[2] });
            for (int ii = 0; ii < 3; ++ii)
                for (int j = 0; j < 3; ++j) B[ii * 3 + j] = o2w[ii * 4 + j] * half[j];
            const float t = (float)idx / (float)count;
            r = fmaf(0.5f, cosf(6.28318f * (t + 0.0f)), 0.5f); g = fmaf(0.5f, cosf(6.28318f * (t + 0.33f)), 0.5f); b = fmaf(0.5f, cosf(6.28318f * (t + 0.67f)), 0.5f);
            al = 0.1f;
        }
        const float det = inverse3(B, bx.inv);
        bool ok = std::isfinite(det) && det != 0.0f;
        for (int k = 0; k < 9; ++k) ok = ok && std::isfinite(bx.inv[k]);
        if (!ok || !(al > 0.0f)) continue;
        const float dx = ox - c[0], dy = oy - c[1], dz = oz - c[2];
        for (int k = 0; k < 3; ++k) bx.lo[k] = fmaf(bx.inv[k * 3 + 2], dz, fmaf(bx.inv[k * 3 + 1], dy, bx.inv[k * 3] * dx));
        bx.r = r; bx.g = g; bx.b = b; bx.a = al; bx.mirrored = det < 0.0f;
        bx.x0 = 0; bx.y0 = 0; bx.x1 = W - 1; bx.y1 = H - 1;           // every pixel is tested (the GPU's tile rectangle is only a speed-up)
        bx.ok = true;
    }
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const int bands = std::max(1, std::min(H, nthreads * 4));
#pragma omp parallel for schedule(dynamic, 1)
    for (int band = 0; band < bands; ++band) {
        const int yb0 = (int)((int64_t)H * band / bands), yb1 = (int)((int64_t)H * (band + 1) / bands) - 1;
        for (int py = yb0; py <= yb1; ++py)
            for (int px = 0; px < W; ++px) {
                const float ndcx = (((float)px + 0.5f) / Wf) * 2.0f - 1.0f;
                const float ndcy = 1.0f - (((float)py + 0.5f) / Hf) * 2.0f;
                const float ax = ndcx / P->proj_m00, ay = ndcy / P->proj_m11;
                float dir[3];
                for (int k = 0; k < 3; ++k) dir[k] = fmaf(ay, R1[k], fmaf(ax, R0[k], R2[k]));
                uint16_t* dst = rt + ((size_t)py * W + px) * 4;
                float acc[4] = { f16tof32(dst[0]), f16tof32(dst[1]), f16tof32(dst[2]), f16tof32(dst[3]) };
                const float sceneZ = scene_depth ? scene_depth[(size_t)py * W + px] : 0.0f;
                for (uint32_t i = 0; i < count; ++i) {
                    const BoxO& bx = boxes[i];
                    if (!bx.ok) continue;
                    float ld[3];
                    for (int k = 0; k < 3; ++k) ld[k] = fmaf(bx.inv[k * 3 + 2], dir[2], fmaf(bx.inv[k * 3 + 1], dir[1], bx.inv[k * 3] * dir[0]));
                    float tmin = -3.4028234663852886e38f, tmax = 3.4028234663852886e38f;
                    for (int k = 0; k < 3; ++k) {
                        const float t1 = (-1.0f - bx.lo[k]) / ld[k], t2 = (1.0f - bx.lo[k]) / ld[k];
                        tmin = fmaxf(tmin, fminf(t1, t2));
                        tmax = fminf(tmax, fmaxf(t1, t2));
                    }
                    if (!(tmin <= tmax)) continue;
                    const float t = bx.mirrored ? tmax : tmin;                     // the face turned towards the camera (far face if mirrored)
                    if (!(t > 0.0f) || !(t >= P->near_clip && t <= P->far_clip)) continue;
                    if (scene_depth && !(t <= sceneZ)) continue;
                    if (mode == 1 && (1.0f - acc[3]) < (1.0f / 4096.0f)) continue;
                    const float tt = 1.0f - acc[3];
                    const float src[4] = { bx.r * bx.a, bx.g * bx.a, bx.b * bx.a, bx.a };
                    for (int ch = 0; ch < 4; ++ch) acc[ch] = mode == 0 ? blend_f16(src[ch], tt, acc[ch]) : fmaf(src[ch], tt, acc[ch]);
                }
                for (int ch = 0; ch < 4; ++ch) dst[ch] = f32tof16(acc[ch]);
            }
    }
    return 0;
}

// GaussianComposite.shader:25-39 with "Blend SrcAlpha OneMinusSrcAlpha" onto a constant background.
// UnityCG.cginc GammaToLinearSpace: c*(c*(c*0.305306011+0.682171111)+0.012522878).
void gso_resolve(const uint16_t* rt, uint32_t W, uint32_t H, const float* bg, float* out32f, uint8_t* out8) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)W * H; ++i) {
        const float C[3] = { f16tof32(rt[i * 4 + 0]), f16tof32(rt[i * 4 + 1]), f16tof32(rt[i * 4 + 2]) };
        const float A = f16tof32(rt[i * 4 + 3]);
        float o[4];
        if (!(A > 0.0f)) { o[0] = bg[0]; o[1] = bg[1]; o[2] = bg[2]; o[3] = bg[3]; }
        else {
            const float invA = 1.0f / A;
            for (int c = 0; c < 3; ++c) {
                const float s = C[c] * invA;
                const float lin = s * fmaf(s, fmaf(s, 0.305306011f, 0.682171111f), 0.012522878f);
                o[c] = fmaf(A, lin - bg[c], bg[c]);
            }
            o[3] = fmaf(A, A - bg[3], bg[3]);                            // the blend state has no separate alpha factors: A*A + bg.a*(1-A)
        }
        if (out32f) for (int c = 0; c < 4; ++c) out32f[i * 4 + c] = o[c];
        if (out8) {
            for (int c = 0; c < 3; ++c) {       // linear -> sRGB 8-bit (what an R8G8B8A8_SRGB target stores)
                const float l = saturatef(o[c]);
                const float s = (l <= 0.0031308f) ? 12.92f * l : fmaf(1.055f, powf(l, 1.0f / 2.4f), -0.055f);
                out8[i * 4 + c] = (uint8_t)floorf(fmaf(saturatef(s), 255.0f, 0.5f));
            }
            out8[i * 4 + 3] = (uint8_t)floorf(fmaf(saturatef(o[3]), 255.0f, 0.5f));
        }
    }
}

} // extern "C"
