// gs_import.cpp -- native importer: raw (PLY-domain) splats -> the five GaussianSplatAsset blobs.
//
// Host C++ (multi-threaded), no device code: asset creation is an import-time step, not part of the per-frame path
// (SURVEY.md section 8f "next #1").  Restates, relative to /root/reference/package/:
//   Editor/Utils/GaussianFileReader.cs :211-233   LinearizeData (normalise + swizzle rotation, exp(scale), sigmoid(opacity), SH0 -> colour)
//   Runtime/GaussianUtils.cs           :9-95      Sigmoid, SH0ToColor, SquareCentered01, PackSmallest3Rotation, MortonEncode3
//   Editor/GaussianSplatAssetCreator.cs :362-429  bounds + Morton reorder, :520-658 chunk bounds / normalisation,
//                                       :705-758  Encode* / EmitEncodedVector (truncating (uint)(v * (k + 0.5f))),
//                                       :776-805  other data, :863-932 colour texture (16x16 Morton tiles), :934-1037 SH items
// Arithmetic is plain IEEE fp32, one rounding per operation (-ffp-contract=off), and the two transcendental steps are
// written out (exp as Cephes-style range reduction + polynomial, x^(1/8) as three correctly rounded square roots) so that
// the bytes are a function of this source only: unitygaussiansplatting_amd/creator.py performs the same operations with
// numpy and tests/test_import.py requires the two to agree bit for bit, including the SH palette of the Cluster* formats and
// the mode-6 BC7 colour blocks (both deterministic by construction: no RNG, fixed summation orders).
#include <algorithm>
#include <cstdio>
#include <string>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include <zlib.h>

#include "../../include/gsplat_c.h"

namespace gs {
int32_t fail(int32_t code, const char* what);
}

namespace {

constexpr uint32_t kChunk = 256, kTexWidth = 2048;

// f(a, b) over [0, n) in contiguous ranges, one per hardware thread; ranges whose thread cannot be created run on the caller.
// An exception thrown inside a worker (e.g. bad_alloc of a per-range vector) is carried back and rethrown on the calling
// thread after all workers have joined, where the entry point's catch turns it into an error code (an exception escaping
// a std::thread would call std::terminate).
template <class F> void parallel_for(size_t n, size_t grain, F f) {
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = std::min(hw, (n + grain - 1) / grain);
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    std::mutex mu;
    std::exception_ptr first;
    auto guarded = [&](size_t a, size_t b) {
        try { f(a, b); }
        catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    };
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        const size_t a = t * per, b = std::min(n, a + per);
        if (a >= b) continue;
        try { th.emplace_back([=, &guarded] { guarded(a, b); }); }
        catch (...) { guarded(a, b); }            // std::system_error (thread limit) / bad_alloc: do the range here
    }
    for (auto& x : th) x.join();
    if (first) std::rethrow_exception(first);
}

// PackSmallest3Rotation (GaussianUtils.cs:46-76): q = xyzw, unit; out = the three smallest components in 0..1 + index / 3
inline void pack_smallest3(const float qv[4], float out[4]) {
    int idx = 0;                                                                            // first max wins
    float best = std::fabs(qv[0]);
    for (int c = 1; c < 4; ++c) if (std::fabs(qv[c]) > best) { best = std::fabs(qv[c]); idx = c; }
    float t[4] = { qv[0], qv[1], qv[2], qv[3] };
    if (idx == 0) { t[0] = qv[1]; t[1] = qv[2]; t[2] = qv[3]; t[3] = qv[0]; }
    if (idx == 1) { t[0] = qv[0]; t[1] = qv[2]; t[2] = qv[3]; t[3] = qv[1]; }
    if (idx == 2) { t[0] = qv[0]; t[1] = qv[1]; t[2] = qv[3]; t[3] = qv[2]; }
    const float sg = t[3] >= 0.0f ? 1.0f : -1.0f;
    for (int c = 0; c < 3; ++c) out[c] = ((t[c] * sg) * 1.41421354f) * 0.5f + 0.5f;
    out[3] = (float)idx / 3.0f;
}

inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// exp(x), deterministic: n = rint(x log2 e), r = x - n ln2 (two-part), degree-5 polynomial (Cephes expf), scale by 2^n
inline float exp_det(float x) {
    x = std::fmin(std::fmax(x, -87.0f), 88.0f);
    const float n = std::nearbyintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * (r * r) + r;
    p = p + 1.0f;
    return p * u2f((uint32_t)((int)n + 127) << 23);
}

inline float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float SquareCentered01(float x) { x -= 0.5f; x = x * (x * sgn(x)); return x * 2.0f + 0.5f; }
inline float sat(float v) { return std::fmin(std::fmax(v, 0.0f), 1.0f); }
inline uint32_t q(float v, float k) { return (uint32_t)(int64_t)(v * k); }          // truncating (uint)(v * (max + 0.5f))

inline uint16_t f32tof16(float f) {      // round to nearest even
    uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u, shift = 126u - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = x - 0x38000000u;
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

inline uint64_t part1by2(uint64_t x) {      // GaussianUtils.cs:81-90
    x &= 0x1fffffull;
    x = (x ^ (x << 32)) & 0x1f00000000ffffull;
    x = (x ^ (x << 16)) & 0x1f0000ff0000ffull;
    x = (x ^ (x << 8)) & 0x100f00f00f00f00full;
    x = (x ^ (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x << 2)) & 0x1249249249249249ull;
    return x;
}

inline void texel_of(uint32_t idx, uint32_t& px, uint32_t& py) {       // GaussianSplatAssetCreator.cs:863-871
    uint32_t t = idx;
    t = (t & 0xFF) | ((t & 0xFE) << 7);
    t &= 0x5555;
    t = (t ^ (t >> 1)) & 0x3333;
    t = (t ^ (t >> 2)) & 0x0f0f;
    const uint32_t tile = idx >> 8, width = kTexWidth / 16;
    px = (tile % width) * 16 + (t & 0xF);
    py = (tile / width) * 16 + (t >> 8);
}

uint32_t vec_size(uint32_t f) { return f == 0 ? 12u : (f == 1 ? 6u : (f == 2 ? 4u : 2u)); }
uint32_t color_size(uint32_t f) { return f == 0 ? 16u : (f == 1 ? 8u : (f == 2 ? 4u : 1u)); }            // BC7: 16-byte blocks of 4x4 texels
uint32_t sh_size(uint32_t f) { return f == 0 ? 192u : (f == 2 ? 60u : (f == 3 ? 32u : 96u)); }          // Float16 and the Cluster* table items: 96 B
uint32_t sh_clusters(uint32_t f) { return f == 4 ? 65536u : (f == 5 ? 32768u : (f == 6 ? 16384u : (f == 7 ? 8192u : (f == 8 ? 4096u : 0u)))); }
uint64_t pad8(uint64_t n) { return (n + 7) / 8 * 8; }

void emit_vec(const float v[3], uint8_t* out, uint32_t fmt) {           // EmitEncodedVector :727-758 (saturating)
    if (fmt == 0) { memcpy(out, v, 12); return; }
    const float x = sat(v[0]), y = sat(v[1]), z = sat(v[2]);
    if (fmt == 1) {
        const uint16_t e[3] = { (uint16_t)q(x, 65535.5f), (uint16_t)q(y, 65535.5f), (uint16_t)q(z, 65535.5f) };
        memcpy(out, e, 6);
    } else if (fmt == 2) {
        const uint32_t e = q(x, 2047.5f) | (q(y, 1023.5f) << 11) | (q(z, 2047.5f) << 21);
        memcpy(out, &e, 4);
    } else {
        const uint16_t e = (uint16_t)(q(x, 63.5f) | (q(y, 31.5f) << 6) | (q(z, 31.5f) << 11));
        memcpy(out, &e, 2);
    }
}

// ---- BC7 mode-6 block encoder (bc7.py encode_texture_mode6, operation for operation in fp32): the colour texture of the
// VeryLow preset.  EditorUtility.CompressTexture's output (GaussianSplatAssetCreator.cs:900-903) cannot be reproduced; any
// conforming BC7 stream decodes the same way.  px: 16 texels x RGBA scaled to 0..255.
void bc7_encode_mode6(const float px[16][4], uint8_t out[16]) {
    float lo[4], hi[4];
    for (int c = 0; c < 4; ++c) { lo[c] = hi[c] = px[0][c]; for (int t = 1; t < 16; ++t) { lo[c] = std::fmin(lo[c], px[t][c]); hi[c] = std::fmax(hi[c], px[t][c]); } }
    auto quant = [](const float v[4], int q[4], int& pbit) {
        float berr = 0.0f;
        for (int p = 0; p < 2; ++p) {
            float qq[4], err = 0.0f;
            for (int c = 0; c < 4; ++c) {
                qq[c] = std::fmin(std::fmax(std::floor((v[c] - (float)p) / 2.0f + 0.5f), 0.0f), 127.0f);
                const float d = (qq[c] * 2.0f + (float)p) - v[c];
                const float d2 = d * d;
                err = c == 0 ? d2 : err + d2;
            }
            if (p == 0 || err < berr) { berr = err; pbit = p; for (int c = 0; c < 4; ++c) q[c] = (int)qq[c]; }
        }
    };
    int q0[4], q1[4], p0, p1;
    quant(lo, q0, p0); quant(hi, q1, p1);
    float e0[4], axis[4], den = 0.0f;
    for (int c = 0; c < 4; ++c) { e0[c] = (float)(q0[c] * 2 + p0); axis[c] = (float)(q1[c] * 2 + p1) - e0[c]; const float a2 = axis[c] * axis[c]; den = c == 0 ? a2 : den + a2; }
    static const float W[16] = { 0 / 64.0f, 4 / 64.0f, 9 / 64.0f, 13 / 64.0f, 17 / 64.0f, 21 / 64.0f, 26 / 64.0f, 30 / 64.0f, 34 / 64.0f, 38 / 64.0f, 43 / 64.0f, 47 / 64.0f, 51 / 64.0f, 55 / 64.0f, 60 / 64.0f, 64 / 64.0f };
    int idx[16];
    for (int t = 0; t < 16; ++t) {
        float num = 0.0f;
        for (int c = 0; c < 4; ++c) { const float m = (px[t][c] - e0[c]) * axis[c]; num = c == 0 ? m : num + m; }
        const float tt = num / std::fmax(den, 1.0e-6f);
        int best = 0; float bd = std::fabs(tt - W[0]);
        for (int k = 1; k < 16; ++k) { const float d = std::fabs(tt - W[k]); if (d < bd) { bd = d; best = k; } }
        idx[t] = best;
    }
    if (idx[0] >= 8) {                                        // anchor: texel 0's index must have its top bit clear
        for (int t = 0; t < 16; ++t) idx[t] = 15 - idx[t];
        for (int c = 0; c < 4; ++c) std::swap(q0[c], q1[c]);
        std::swap(p0, p1);
    }
    unsigned __int128 v = (unsigned __int128)1 << 6;
    int pos = 7;
    for (int c = 0; c < 4; ++c) { v |= (unsigned __int128)q0[c] << pos; pos += 7; v |= (unsigned __int128)q1[c] << pos; pos += 7; }
    v |= (unsigned __int128)p0 << pos; pos += 1;
    v |= (unsigned __int128)p1 << pos; pos += 1;
    v |= (unsigned __int128)idx[0] << pos; pos += 3;
    for (int t = 1; t < 16; ++t) { v |= (unsigned __int128)idx[t] << pos; pos += 4; }
    memcpy(out, &v, 16);
}

// ---- SH palette for the Cluster* formats (GaussianSplatAssetCreator.cs:476-518 / KMeansClustering.cs).  The reference's
// mini-batch k-means (k-means++ seeding, its own RNG, Burst) only decides WHICH palette a file gets; the data layout -- fp16
// table of K means + a u16 index per splat -- is what the renderer reads.  This is creator.py's ClusterSHs, step for step, so
// that the two importers emit the same bytes: stride-sampled seeds and training subset, 4 Lloyd iterations on the subset
// (assignment by the smallest |c|^2 - 2 x.c in double, first minimum; means = double sums in point order, rounded to fp32),
// then one assignment pass over all splats.  x: n x 45.  The assignment is the importer's one GEMM-shaped loop (n x K x 45).
void assign_clusters(const float* x, size_t n, const std::vector<float>& means, uint32_t K, uint32_t* out) {
    std::vector<double> mt((size_t)45 * K), c2(K);                            // means transposed: the inner loop runs over clusters
    for (uint32_t j = 0; j < K; ++j) {
        double sq = 0.0;
        for (int k = 0; k < 45; ++k) { const double m = means[(size_t)j * 45 + k]; mt[(size_t)k * K + j] = m; sq += m * m; }
        c2[j] = sq;
    }
    parallel_for(n, 64, [&](size_t a, size_t b) {
        std::vector<double> dot(K);
        for (size_t i = a; i < b; ++i) {
            std::fill(dot.begin(), dot.end(), 0.0);
            for (int k = 0; k < 45; ++k) {
                const double xv = x[i * 45 + k];
                const double* m = &mt[(size_t)k * K];
                for (uint32_t j = 0; j < K; ++j) dot[j] += xv * m[j];
            }
            uint32_t best = 0; double bd = c2[0] - 2.0 * dot[0];
            for (uint32_t j = 1; j < K; ++j) { const double d = c2[j] - 2.0 * dot[j]; if (d < bd) { bd = d; best = j; } }
            out[i] = best;
        }
    });
}

void cluster_shs(const float* x, size_t n, uint32_t K, std::vector<float>& means, std::vector<uint32_t>& index) {
    means.resize((size_t)K * 45);
    for (uint32_t j = 0; j < K; ++j) memcpy(&means[(size_t)j * 45], x + (((uint64_t)j * n) / K) * 45, 180);
    const size_t S = 200000;
    std::vector<float> subBuf;
    const float* sub = x;
    size_t ns = n;
    if (n > S) {
        subBuf.resize(S * 45);
        for (size_t i = 0; i < S; ++i) memcpy(&subBuf[i * 45], x + (((uint64_t)i * n) / S) * 45, 180);
        sub = subBuf.data(); ns = S;
    }
    std::vector<uint32_t> idx(ns), start(K + 1), sorted(ns);
    for (int it = 0; it < 4; ++it) {
        assign_clusters(sub, ns, means, K, idx.data());
        std::fill(start.begin(), start.end(), 0u);                           // counting sort: the points of a cluster in point order
        for (size_t i = 0; i < ns; ++i) start[idx[i] + 1]++;
        for (uint32_t j = 0; j < K; ++j) start[j + 1] += start[j];
        { std::vector<uint32_t> cur(start.begin(), start.end() - 1); for (size_t i = 0; i < ns; ++i) sorted[cur[idx[i]]++] = (uint32_t)i; }
        parallel_for(K, 16, [&](size_t ja, size_t jb) {
            for (size_t j = ja; j < jb; ++j) {
                const uint32_t cnt = start[j + 1] - start[j];
                if (!cnt) continue;                                          // an empty cluster keeps its mean
                double sum[45] = {0};
                for (uint32_t t = start[j]; t < start[j + 1]; ++t) { const float* p = sub + (size_t)sorted[t] * 45; for (int k = 0; k < 45; ++k) sum[k] += (double)p[k]; }
                for (int k = 0; k < 45; ++k) means[j * 45 + k] = (float)(sum[k] / (double)cnt);
            }
        });
    }
    index.resize(n);
    assign_clusters(x, n, means, K, index.data());
}

struct Splat { float pos[3], dc0[3], sh[45], opacity, scale[3], rot[4]; };      // linearised

void sizes_of(uint32_t n, const gs_import_formats& f, uint64_t s[5]) {
    const uint64_t texH = ((uint64_t)(n + kTexWidth - 1) / kTexWidth + 15) / 16 * 16;
    const bool chunks = !(f.pos_format == 0 && f.scale_format == 0 && f.color_format == 0 && f.sh_format == 0);
    s[0] = pad8((uint64_t)n * vec_size(f.pos_format));
    s[1] = pad8((uint64_t)n * (4 + vec_size(f.scale_format) + (f.sh_format > 3 ? 2u : 0u)));              // + u16 SH table index
    s[2] = (uint64_t)kTexWidth * std::max<uint64_t>(texH, 16) * color_size(f.color_format);
    s[3] = (uint64_t)(f.sh_format > 3 ? sh_clusters(f.sh_format) : n) * sh_size(f.sh_format);
    s[4] = chunks ? (uint64_t)((n + kChunk - 1) / kChunk) * 64 : 0;
}

} // namespace

extern "C" {

int32_t gs_import_blob_sizes(uint32_t splat_count, const gs_import_formats* f, uint64_t sizes[5]) {
    if (!f || !sizes || splat_count == 0) return gs::fail(GS_ERR_INVALID_ARGUMENT, "null argument / no splats");
    if (f->pos_format > 3 || f->scale_format > 3 || f->color_format > 3 || f->sh_format > 8) return gs::fail(GS_ERR_INVALID_ARGUMENT, "format enum out of range");
    if (f->sh_format > GS_SH_NORM6 && sh_clusters(f->sh_format) >= splat_count)
        return gs::fail(GS_ERR_INVALID_ARGUMENT, "Cluster* SH needs more splats than table entries (the reference falls back to unclustered data there)");
    sizes_of(splat_count, *f, sizes);
    return GS_OK;
}

static int32_t import_encode_impl(const gs_import_input* in, const gs_import_formats* f, void* const blobs[5], const uint64_t sizes[5],
                                  float bounds_min[3], float bounds_max[3]);

int32_t gs_import_encode(const gs_import_input* in, const gs_import_formats* f, void* const blobs[5], const uint64_t sizes[5],
                         float bounds_min[3], float bounds_max[3]) {
    try { return import_encode_impl(in, f, blobs, sizes, bounds_min, bounds_max); }      // nothing may throw across the C boundary
    catch (const std::bad_alloc&) { return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    catch (...) { return gs::fail(GS_ERR_INVALID_ARGUMENT, "unexpected failure in the importer"); }
}

static int32_t import_encode_impl(const gs_import_input* in, const gs_import_formats* f, void* const blobs[5], const uint64_t sizes[5],
                                  float bounds_min[3], float bounds_max[3]) {
    if (!in || !f || !blobs || !sizes) return gs::fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    uint64_t need[5];
    const int32_t rc = gs_import_blob_sizes(in->splat_count, f, need);
    if (rc != GS_OK) return rc;
    if (!in->pos || !in->dc0 || !in->sh || !in->opacity || !in->scale || !in->rot) return gs::fail(GS_ERR_INVALID_ARGUMENT, "an input array is null");
    for (int k = 0; k < 5; ++k)
        if (need[k] && (!blobs[k] || sizes[k] < need[k])) return gs::fail(GS_ERR_INVALID_ARGUMENT, "an output blob is null or too small");
    const uint32_t n = in->splat_count;
    const bool useChunks = need[4] != 0;
    std::vector<Splat> s;
    try { s.resize(n); } catch (...) { return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }

    // ---- LinearizeData (GaussianFileReader.cs:211-233) or plain copy
    parallel_for(n, 1 << 14, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            Splat& o = s[i];
            memcpy(o.pos, in->pos + i * 3, 12);
            memcpy(o.sh, in->sh + i * 45, 180);
            if (!f->linearize) {
                memcpy(o.dc0, in->dc0 + i * 3, 12); memcpy(o.scale, in->scale + i * 3, 12); memcpy(o.rot, in->rot + i * 4, 16);
                o.opacity = in->opacity[i];
                continue;
            }
            for (int c = 0; c < 3; ++c) {
                o.dc0[c] = in->dc0[i * 3 + c] * 0.2820948f + 0.5f;                         // SH0ToColor
                o.scale[c] = std::fabs(exp_det(in->scale[i * 3 + c]));
            }
            o.opacity = 1.0f / (1.0f + exp_det(-in->opacity[i]));                           // Sigmoid
            const float* w = in->rot + i * 4;                                               // (w, x, y, z)
            const float len = std::sqrt(((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]) + w[3] * w[3]);
            const float qv[4] = { w[1] / len, w[2] / len, w[3] / len, w[0] / len };         // normalize(wxyz).yzwx
            pack_smallest3(qv, o.rot);
        }
    });

    // ---- bounds + Morton reorder (GaussianSplatAssetCreator.cs:362-429)
    float bmin[3] = { s[0].pos[0], s[0].pos[1], s[0].pos[2] }, bmax[3] = { bmin[0], bmin[1], bmin[2] };
    for (uint32_t i = 1; i < n; ++i)
        for (int c = 0; c < 3; ++c) { bmin[c] = std::fmin(bmin[c], s[i].pos[c]); bmax[c] = std::fmax(bmax[c], s[i].pos[c]); }
    if (bounds_min) memcpy(bounds_min, bmin, 12);
    if (bounds_max) memcpy(bounds_max, bmax, 12);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (f->morton) {
        std::vector<uint64_t> code(n);
        const float inv[3] = { 1.0f / (bmax[0] - bmin[0]), 1.0f / (bmax[1] - bmin[1]), 1.0f / (bmax[2] - bmin[2]) };
        parallel_for(n, 1 << 15, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i) {
                uint64_t ip[3];
                for (int c = 0; c < 3; ++c) ip[c] = (uint64_t)(uint32_t)(int64_t)(((s[i].pos[c] - bmin[c]) * inv[c]) * 2097151.0f);
                code[i] = (part1by2(ip[2]) << 2) | (part1by2(ip[1]) << 1) | part1by2(ip[0]);
            }
        });
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });   // (code, index)
    }

    // ---- SH palette for the Cluster* formats: clustered on the raw SH vectors of the reordered splats, before chunking (:286-291)
    const bool clustered = f->sh_format > GS_SH_NORM6;
    std::vector<float> shMeans;
    std::vector<uint32_t> shIndex;
    if (clustered) {
        std::vector<float> x((size_t)n * 45);
        parallel_for(n, 1 << 14, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) memcpy(&x[i * 45], s[order[i]].sh, 180); });
        cluster_shs(x.data(), n, sh_clusters(f->sh_format), shMeans, shIndex);
    }
    // BC7: the colour texels are gathered as floats first and compressed block by block afterwards
    const bool bc7 = f->color_format == GS_COLOR_BC7;
    const uint64_t texH = need[2] / kTexWidth;                        // 1 byte per texel for BC7
    std::vector<float> tex;
    if (bc7) tex.assign((size_t)kTexWidth * texH * 4, 0.0f);

    // ---- chunk bounds and normalisation (CalcChunkDataJob :520-638), per chunk of 256 reordered splats
    uint8_t* posOut = (uint8_t*)blobs[0];
    uint8_t* othOut = (uint8_t*)blobs[1];
    uint8_t* colOut = (uint8_t*)blobs[2];
    uint8_t* shOut = (uint8_t*)blobs[3];
    uint8_t* chkOut = (uint8_t*)blobs[4];
    memset(posOut, 0, need[0]); memset(othOut, 0, need[1]); memset(colOut, 0, need[2]); memset(shOut, 0, need[3]);
    const uint32_t nchunks = (n + kChunk - 1) / kChunk;
    const uint32_t posSz = vec_size(f->pos_format), sclSz = vec_size(f->scale_format), othSz = 4 + sclSz + (clustered ? 2u : 0u),
                   colSz = color_size(f->color_format), shSz = sh_size(f->sh_format);
    parallel_for(nchunks, 16, [&](size_t ca, size_t cb) {
        std::vector<Splat> c(kChunk);
        for (size_t ci = ca; ci < cb; ++ci) {
            const uint32_t first = (uint32_t)ci * kChunk, cnt = std::min(kChunk, n - first);
            for (uint32_t k = 0; k < cnt; ++k) c[k] = s[order[first + k]];
            if (useChunks) {
                const float inf = std::numeric_limits<float>::infinity();
                float pmin[3] = { inf, inf, inf }, pmax[3] = { -inf, -inf, -inf }, smin[3] = { inf, inf, inf }, smax[3] = { -inf, -inf, -inf };
                float cmin[4] = { inf, inf, inf, inf }, cmax[4] = { -inf, -inf, -inf, -inf }, hmin[3] = { inf, inf, inf }, hmax[3] = { -inf, -inf, -inf };
                for (uint32_t k = 0; k < cnt; ++k) {
                    Splat& o = c[k];
                    for (int d = 0; d < 3; ++d) o.scale[d] = std::sqrt(std::sqrt(std::sqrt(o.scale[d])));        // s^(1/8)
                    o.opacity = SquareCentered01(o.opacity);
                    for (int d = 0; d < 3; ++d) {
                        pmin[d] = std::fmin(pmin[d], o.pos[d]); pmax[d] = std::fmax(pmax[d], o.pos[d]);
                        smin[d] = std::fmin(smin[d], o.scale[d]); smax[d] = std::fmax(smax[d], o.scale[d]);
                        cmin[d] = std::fmin(cmin[d], o.dc0[d]); cmax[d] = std::fmax(cmax[d], o.dc0[d]);
                    }
                    cmin[3] = std::fmin(cmin[3], o.opacity); cmax[3] = std::fmax(cmax[3], o.opacity);
                    for (int j = 0; j < 15; ++j)
                        for (int d = 0; d < 3; ++d) { hmin[d] = std::fmin(hmin[d], o.sh[j * 3 + d]); hmax[d] = std::fmax(hmax[d], o.sh[j * 3 + d]); }
                }
                for (int d = 0; d < 3; ++d) {                                             // make sure the ranges are not empty (:592-595)
                    pmax[d] = std::fmax(pmax[d], pmin[d] + 1.0e-5f); smax[d] = std::fmax(smax[d], smin[d] + 1.0e-5f);
                    hmax[d] = std::fmax(hmax[d], hmin[d] + 1.0e-5f);
                }
                for (int d = 0; d < 4; ++d) cmax[d] = std::fmax(cmax[d], cmin[d] + 1.0e-5f);
                uint32_t w[16];
                for (int d = 0; d < 4; ++d) w[d] = (uint32_t)f32tof16(cmin[d]) | ((uint32_t)f32tof16(cmax[d]) << 16);
                for (int d = 0; d < 3; ++d) { w[4 + 2 * d] = f2u(pmin[d]); w[5 + 2 * d] = f2u(pmax[d]); }
                for (int d = 0; d < 3; ++d) w[10 + d] = (uint32_t)f32tof16(smin[d]) | ((uint32_t)f32tof16(smax[d]) << 16);
                for (int d = 0; d < 3; ++d) w[13 + d] = (uint32_t)f32tof16(hmin[d]) | ((uint32_t)f32tof16(hmax[d]) << 16);
                memcpy(chkOut + ci * 64, w, 64);
                for (uint32_t k = 0; k < cnt; ++k) {                                      // normalise with the fp32 (un-rounded) bounds (:613-637)
                    Splat& o = c[k];
                    for (int d = 0; d < 3; ++d) {
                        o.pos[d] = (o.pos[d] - pmin[d]) / (pmax[d] - pmin[d]);
                        o.scale[d] = (o.scale[d] - smin[d]) / (smax[d] - smin[d]);
                        o.dc0[d] = (o.dc0[d] - cmin[d]) / (cmax[d] - cmin[d]);
                    }
                    o.opacity = (o.opacity - cmin[3]) / (cmax[3] - cmin[3]);
                    for (int j = 0; j < 15; ++j)
                        for (int d = 0; d < 3; ++d) o.sh[j * 3 + d] = (o.sh[j * 3 + d] - hmin[d]) / (hmax[d] - hmin[d]);
                }
            }
            for (uint32_t k = 0; k < cnt; ++k) {
                const Splat& o = c[k];
                const uint32_t i = first + k;
                emit_vec(o.pos, posOut + (size_t)i * posSz, f->pos_format);
                // other: rotation 10.10.10.2 + scale (:776-805)
                const uint32_t enc = q(o.rot[0], 1023.5f) | (q(o.rot[1], 1023.5f) << 10) | (q(o.rot[2], 1023.5f) << 20) | (q(o.rot[3], 3.5f) << 30);
                memcpy(othOut + (size_t)i * othSz, &enc, 4);
                emit_vec(o.scale, othOut + (size_t)i * othSz + 4, f->scale_format);
                if (clustered) { const uint16_t si = (uint16_t)shIndex[i]; memcpy(othOut + (size_t)i * othSz + 4 + sclSz, &si, 2); }   // u16 table index (:797-801)
                // colour texel (:873-932)
                uint32_t px, py;
                texel_of(i, px, py);
                uint8_t* t = colOut + ((size_t)py * kTexWidth + px) * colSz;
                const float col[4] = { o.dc0[0], o.dc0[1], o.dc0[2], o.opacity };
                if (bc7) memcpy(&tex[((size_t)py * kTexWidth + px) * 4], col, 16);
                else if (f->color_format == 0) memcpy(t, col, 16);
                else if (f->color_format == 1) { uint16_t h[4]; for (int d = 0; d < 4; ++d) h[d] = f32tof16(col[d]); memcpy(t, h, 8); }
                else { const uint32_t e = q(sat(col[0]), 255.5f) | (q(sat(col[1]), 255.5f) << 8) | (q(sat(col[2]), 255.5f) << 16) | (q(sat(col[3]), 255.5f) << 24); memcpy(t, &e, 4); }
                // SH item (:934-1037); clustered assets store the table instead (below)
                if (clustered) continue;
                uint8_t* sp = shOut + (size_t)i * shSz;
                if (f->sh_format == 0) memcpy(sp, o.sh, 180);
                else if (f->sh_format == 1) { uint16_t h[45]; for (int d = 0; d < 45; ++d) h[d] = f32tof16(o.sh[d]); memcpy(sp, h, 90); }
                else if (f->sh_format == 2) {
                    for (int j = 0; j < 15; ++j) {       // CreateSHDataJob does not saturate
                        const uint32_t e = q(o.sh[j * 3], 2047.5f) | (q(o.sh[j * 3 + 1], 1023.5f) << 11) | (q(o.sh[j * 3 + 2], 2047.5f) << 21);
                        memcpy(sp + j * 4, &e, 4);
                    }
                } else {
                    for (int j = 0; j < 15; ++j) {
                        const uint16_t e = (uint16_t)(q(o.sh[j * 3], 31.5f) | (q(o.sh[j * 3 + 1], 63.5f) << 5) | (q(o.sh[j * 3 + 2], 31.5f) << 11));
                        memcpy(sp + j * 2, &e, 2);
                    }
                }
            }
        }
    });
    if (clustered) {                                                  // ConvertSHClustersJob (:443-474): SHTableItemFloat16 per cluster
        const uint32_t K = sh_clusters(f->sh_format);
        parallel_for(K, 256, [&](size_t a, size_t b) {
            for (size_t j = a; j < b; ++j) { uint16_t h[48] = {0}; for (int d = 0; d < 45; ++d) h[d] = f32tof16(shMeans[j * 45 + d]); memcpy(shOut + j * 96, h, 96); }
        });
    }
    if (bc7) {                                                        // EditorUtility.CompressTexture(tex, BC7, 100) (:900-903), every block in mode 6
        const size_t bw = kTexWidth / 4, bh = texH / 4;
        parallel_for(bw * bh, 1024, [&](size_t a, size_t b) {
            for (size_t blk = a; blk < b; ++blk) {
                const size_t by = blk / bw, bx = blk % bw;
                float px[16][4];
                for (int t = 0; t < 16; ++t) {
                    const float* src = &tex[(((by * 4 + (size_t)(t >> 2)) * kTexWidth) + bx * 4 + (size_t)(t & 3)) * 4];
                    for (int c = 0; c < 4; ++c) px[t][c] = std::fmin(std::fmax(src[c], 0.0f), 1.0f) * 255.0f;
                }
                bc7_encode_mode6(px, colOut + blk * 16);
            }
        });
    }
    return GS_OK;
}

// ---- PLY reader: PLYFileReader.cs:25-76 (header), GaussianFileReader.cs:80-208 (PLYDataToSplats + ReorderSHs) ----------
struct gs_ply {
    uint32_t count = 0;
    std::vector<float> pos, dc0, sh, opacity, scale, rot;
};

static bool read_line(FILE* f, std::string& line) {        // PLYFileReader.ReadLine: up to '\n', a trailing '\r' dropped
    line.clear();
    int c;
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        line.push_back((char)c);
    }
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return any;
}

static int32_t ply_open_impl(const char* path, gs_ply** out, uint32_t* splat_count);

int32_t gs_ply_open(const char* path, gs_ply** out, uint32_t* splat_count) {
    try { return ply_open_impl(path, out, splat_count); }
    catch (const std::bad_alloc&) { return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    catch (...) { return gs::fail(GS_ERR_INVALID_ASSET, "unexpected failure while reading the PLY file"); }
}

static int32_t ply_open_impl(const char* path, gs_ply** out, uint32_t* splat_count) {
    if (!path || !out) return gs::fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return gs::fail(GS_ERR_INVALID_ASSET, "PLY file cannot be opened");
    struct Attr { std::string name; int type; int offset; };                    // type: 1 float, 2 double, 3 uchar, 0 none
    std::vector<Attr> attrs;
    long long count = 0;
    int stride = 0;
    bool le = false;
    std::string line;
    for (int li = 0; li < 9000; ++li) {
        if (!read_line(f, line) || line == "end_header" || line.empty()) break;
        std::vector<std::string> tok;
        size_t a = 0;
        for (;;) { const size_t b = line.find(' ', a); tok.push_back(line.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; }
        if (tok.size() == 3 && tok[0] == "format" && tok[1] == "binary_little_endian" && tok[2] == "1.0") le = true;
        if (tok.size() == 3 && tok[0] == "element" && tok[1] == "vertex") count = atoll(tok[2].c_str());
        if (tok.size() == 3 && tok[0] == "property") {
            const int type = tok[1] == "float" ? 1 : (tok[1] == "double" ? 2 : (tok[1] == "uchar" ? 3 : 0));
            attrs.push_back({ tok[2], type, stride });
            stride += type == 1 ? 4 : (type == 2 ? 8 : (type == 3 ? 1 : 0));
        }
    }
    if (!le) { fclose(f); return gs::fail(GS_ERR_INVALID_ASSET, "PLY not supported: needs to be binary, little endian PLY format"); }
    if (count <= 0 || count > (1ll << 30) || stride <= 0) { fclose(f); return gs::fail(GS_ERR_INVALID_ASSET, "PLY has no vertices"); }
    auto find = [&](const std::string& nm) { for (auto& at : attrs) if (at.name == nm && at.type == 1) return at.offset; return -1; };
    const char* required[] = { "x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3" };
    for (const char* r : required)
        if (find(r) < 0) { fclose(f); return gs::fail(GS_ERR_INVALID_ASSET, "PLY file is probably not a Gaussian Splat file (a required float property is missing)"); }
    gs_ply* p = new (std::nothrow) gs_ply();
    if (!p) { fclose(f); return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    std::vector<uint8_t> body;
    try {
        body.resize((size_t)count * stride);
        p->count = (uint32_t)count;
        p->pos.resize((size_t)count * 3); p->dc0.resize((size_t)count * 3); p->sh.assign((size_t)count * 45, 0.0f);
        p->opacity.resize(count); p->scale.resize((size_t)count * 3); p->rot.resize((size_t)count * 4);
    } catch (...) { fclose(f); delete p; return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    const size_t got = fread(body.data(), 1, body.size(), f);
    fclose(f);
    if (got != body.size()) { delete p; return gs::fail(GS_ERR_INVALID_ASSET, "PLY read error: fewer data bytes than the header announces"); }
    const int ox[3] = { find("x"), find("y"), find("z") }, od[3] = { find("f_dc_0"), find("f_dc_1"), find("f_dc_2") };
    const int os[3] = { find("scale_0"), find("scale_1"), find("scale_2") }, orr[4] = { find("rot_0"), find("rot_1"), find("rot_2"), find("rot_3") };
    const int oo = find("opacity");
    int orest[45];
    for (int k = 0; k < 45; ++k) orest[k] = find("f_rest_" + std::to_string(k));
    const int strideC = stride;
    parallel_for((size_t)count, 1 << 14, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            const uint8_t* v = body.data() + i * strideC;
            auto ld = [&](int off) { float x = 0.0f; if (off >= 0) memcpy(&x, v + off, 4); return x; };
            for (int c = 0; c < 3; ++c) { p->pos[i * 3 + c] = ld(ox[c]); p->dc0[i * 3 + c] = ld(od[c]); p->scale[i * 3 + c] = ld(os[c]); }
            for (int c = 0; c < 4; ++c) p->rot[i * 4 + c] = ld(orr[c]);
            p->opacity[i] = ld(oo);
            // ReorderSHs: the file holds 15 R, 15 G, 15 B; the splat wants 15 x (r, g, b)
            for (int j = 0; j < 15; ++j)
                for (int c = 0; c < 3; ++c) p->sh[i * 45 + j * 3 + c] = ld(orest[c * 15 + j]);
        }
    });
    *out = p;
    if (splat_count) *splat_count = p->count;
    return GS_OK;
}

// ---- SPZ input (Niantic / Scaniverse; Editor/Utils/SPZFileReader.cs): a gzip stream of  header(16 B: "NGSP", version 2,
// numPoints, shLevel | fractBits << 8 | flags << 16)  then, for all points, the arrays  positions (3 x 24-bit fixed point),
// alphas (u8), colours (3 x u8), scales (3 x u8), rotations (3 x u8), SH (shCoeffs x 3 x u8).  UnpackDataJob (:141-205)
// turns them into ALREADY LINEAR InputSplatData (scale = |exp(b/16 - 10)|, opacity = b/255, dc0 = SH0ToColor((b/255 - 0.5)/0.15),
// rot = PackSmallest3Rotation(normalize(xyz, w))), so the handle's arrays go to gs_import_encode with linearize = 0.
static bool gunzip_file(FILE* f, std::vector<uint8_t>& out, size_t limit) {
    std::vector<uint8_t> in(1 << 16);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;       // gzip wrapper
    int rc = Z_OK;
    out.resize(1 << 20);
    size_t have = 0;
    while (rc != Z_STREAM_END) {
        if (zs.avail_in == 0) {
            zs.avail_in = (uInt)fread(in.data(), 1, in.size(), f);
            zs.next_in = in.data();
            if (zs.avail_in == 0) break;
        }
        if (have == out.size()) { if (out.size() >= limit) { inflateEnd(&zs); return false; } out.resize(std::min(limit, out.size() * 2)); }
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
        const size_t before = zs.avail_out;
        rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { inflateEnd(&zs); return false; }
    }
    inflateEnd(&zs);
    out.resize(have);
    return true;
}

static int32_t spz_open_impl(const char* path, gs_ply** out, uint32_t* splat_count) {
    if (!path || !out) return gs::fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ file cannot be opened");
    std::vector<uint8_t> raw;
    const bool ok = gunzip_file(f, raw, (size_t)10'000'000 * 64 + 64);
    fclose(f);
    if (!ok) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error: not a gzip stream");
    if (raw.size() < 16) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, failed to read header");
    uint32_t h[4];
    memcpy(h, raw.data(), 16);
    if (h[0] != 0x5053474eu) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, header magic unexpected");
    if (h[1] != 2u) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, header version unexpected");
    const int64_t n = (int32_t)h[2];
    const int shLevel = (int)(h[3] & 0xFF), fractBits = (int)((h[3] >> 8) & 0xFF);
    if (n < 1 || n > 10'000'000) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, out of range splat count");     // 10M hardcoded in SPZ code
    if (shLevel < 0 || shLevel > 3) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, out of range SH level");
    if (fractBits < 0 || fractBits > 24) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, out of range fractional bits");
    const int shCoeffs = shLevel == 1 ? 3 : (shLevel == 2 ? 8 : (shLevel == 3 ? 15 : 0));
    const size_t N = (size_t)n;
    const size_t oPos = 16, oAlpha = oPos + N * 9, oCol = oAlpha + N, oScale = oCol + N * 3, oRot = oScale + N * 3, oSh = oRot + N * 3, end = oSh + N * 3 * shCoeffs;
    if (raw.size() < end) return gs::fail(GS_ERR_INVALID_ASSET, "SPZ read error, file smaller than it should be");
    gs_ply* p = new (std::nothrow) gs_ply();
    if (!p) return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation");
    try {
        p->count = (uint32_t)n;
        p->pos.resize(N * 3); p->dc0.resize(N * 3); p->sh.assign(N * 45, 0.0f);
        p->opacity.resize(N); p->scale.resize(N * 3); p->rot.resize(N * 4);
    } catch (...) { delete p; return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    const float fractScale = 1.0f / (float)(1 << fractBits);
    const uint8_t* d = raw.data();
    const size_t shLen = N * 3 * shCoeffs;
    parallel_for(N, 1 << 14, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            for (int c = 0; c < 3; ++c) {
                const uint8_t* q = d + oPos + (i * 3 + c) * 3;
                int32_t fx = (int32_t)q[0] | ((int32_t)q[1] << 8) | ((int32_t)q[2] << 16);
                if (fx & 0x800000) fx |= (int32_t)0xff000000;                      // sign extension
                p->pos[i * 3 + c] = (float)fx * fractScale;
                p->scale[i * 3 + c] = std::fabs(exp_det((float)d[oScale + i * 3 + c] / 16.0f - 10.0f));         // LinearScale
                p->dc0[i * 3 + c] = (((float)d[oCol + i * 3 + c] / 255.0f - 0.5f) / 0.15f) * 0.2820948f + 0.5f;    // SH0ToColor
            }
            p->opacity[i] = (float)d[oAlpha + i] / 255.0f;
            float q4[4];
            for (int c = 0; c < 3; ++c) q4[c] = (float)d[oRot + i * 3 + c] * (1.0f / 127.5f) - 1.0f;
            const float sq = (q4[0] * q4[0] + q4[1] * q4[1]) + q4[2] * q4[2];
            q4[3] = std::sqrt(std::fmax(0.0f, 1.0f - sq));
            const float inv = 1.0f / std::sqrt(((q4[0] * q4[0] + q4[1] * q4[1]) + q4[2] * q4[2]) + q4[3] * q4[3]);   // math.normalize
            const float qv[4] = { q4[0] * inv, q4[1] * inv, q4[2] * inv, q4[3] * inv };
            pack_smallest3(qv, &p->rot[i * 4]);
            // UnpackSH: the job reads FIFTEEN coefficients starting at index * shCoeffs * 3 whatever the level (:178-193), i.e. for
            // levels < 3 it runs on into the following splats' bytes; restated as is, with bytes past the array reading as 128 (= 0)
            const size_t base = i * 3 * (size_t)shCoeffs;
            for (int k = 0; k < 45; ++k) {
                const size_t at = base + (size_t)k;
                const float bv = at < shLen ? (float)d[oSh + at] : 128.0f;
                p->sh[i * 45 + k] = (bv - 128.0f) / 128.0f;
            }
        }
    });
    *out = p;
    if (splat_count) *splat_count = p->count;
    return GS_OK;
}

int32_t gs_spz_open(const char* path, gs_ply** out, uint32_t* splat_count) {
    try { return spz_open_impl(path, out, splat_count); }
    catch (const std::bad_alloc&) { return gs::fail(GS_ERR_OUT_OF_MEMORY, "host allocation"); }
    catch (...) { return gs::fail(GS_ERR_INVALID_ASSET, "unexpected failure while reading the SPZ file"); }
}

int32_t gs_ply_arrays(const gs_ply* ply, gs_import_input* out) {
    if (!ply || !out) return gs::fail(GS_ERR_INVALID_ARGUMENT, "null argument");
    out->splat_count = ply->count;
    out->pos = ply->pos.data(); out->dc0 = ply->dc0.data(); out->sh = ply->sh.data();
    out->opacity = ply->opacity.data(); out->scale = ply->scale.data(); out->rot = ply->rot.data();
    return GS_OK;
}

int32_t gs_ply_close(gs_ply* ply) {
    delete ply;
    return GS_OK;
}

} // extern "C"
