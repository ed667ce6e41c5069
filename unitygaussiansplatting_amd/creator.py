"""GaussianSplatAssetCreator: PLY / raw splat arrays -> GaussianSplatAsset blobs (numpy, host side).

Follows the encoding rules of the reference importer so that the bytes are, by construction, what
the decoder in the hot path expects:
  /root/reference/package/Editor/GaussianSplatAssetCreator.cs
      :189-228 quality presets, :362-429 bounds + Morton reorder, :520-658 chunk bounds/normalise,
      :705-758 Encode*/EmitEncodedVector (truncating ``(uint)(v * (k + 0.5f))``),
      :776-805 other data (rot 10.10.10.2 + scale), :863-932 colour texture (Morton 16x16 texel order),
      :934-1037 SH table items
  /root/reference/package/Editor/Utils/GaussianFileReader.cs :91-155 PLY attributes, :186-208 SH
      reorder, :211-233 LinearizeData;  PLYFileReader.cs:25-76 header parse
  /root/reference/package/Runtime/GaussianUtils.cs :9-95 Sigmoid, SH0ToColor, SquareCentered01,
      PackSmallest3Rotation, MortonEncode3

This is an import-time tool (SURVEY.md section 8f "next #1"), not part of the per-frame path.  SH
clustering (Cluster* formats) produces the reference's data layout (fp16 table of K means + a u16 index per splat
in the `other` record, GaussianSplatAssetCreator.cs:476-518,776-805) with a plain seeded mini-batch k-means; the
reference's own KMeansClustering.cs (batched, Burst) differs in its iteration schedule, which only changes which
palette is found, not how it is stored or decoded.
"""
from __future__ import annotations

import io
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from .asset import (CHUNK_DTYPE, CalcTextureSize, ColorFormat, GaussianSplatAsset, GetColorSize,
                    GetVectorSize, SHFormat, VectorFormat, kChunkSize, kTextureWidth)

f32 = np.float32

# --------------------------------------------------------------------------------------------------
# Quality presets: GaussianSplatAssetCreator.cs:189-228
# --------------------------------------------------------------------------------------------------
QUALITY = {
    "VeryLow": (VectorFormat.Norm11, VectorFormat.Norm6, ColorFormat.BC7, SHFormat.Cluster4k),
    "Low": (VectorFormat.Norm11, VectorFormat.Norm6, ColorFormat.Norm8x4, SHFormat.Cluster16k),
    "Medium": (VectorFormat.Norm11, VectorFormat.Norm11, ColorFormat.Norm8x4, SHFormat.Norm6),
    "High": (VectorFormat.Norm16, VectorFormat.Norm16, ColorFormat.Float16x4, SHFormat.Norm11),
    "VeryHigh": (VectorFormat.Float32, VectorFormat.Float32, ColorFormat.Float32x4, SHFormat.Float32),
}

PLY_ATTRS = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
             + [f"f_rest_{i}" for i in range(45)]
             + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
assert len(PLY_ATTRS) == 62   # InputSplatData = 62 floats (GaussianFileReader.cs:17-26,154)


@dataclass
class InputSplatData:
    """Struct-of-arrays form of GaussianFileReader.cs:17-26 (after ReorderSHs: sh is [N,15,3] rgb-interleaved)."""
    pos: np.ndarray       # [N,3] f32
    dc0: np.ndarray       # [N,3] f32   raw f_dc (before LinearizeData) or colour (after)
    sh: np.ndarray        # [N,15,3] f32
    opacity: np.ndarray   # [N] f32     logit (before) or 0..1 (after)
    scale: np.ndarray     # [N,3] f32   log-scale (before) or linear (after)
    rot: np.ndarray       # [N,4] f32   PLY rot_0..3 = (w,x,y,z) (before) or packed smallest-3 + index/3 (after)

    def __len__(self) -> int:
        return len(self.pos)

    def take(self, idx: np.ndarray) -> "InputSplatData":
        return InputSplatData(self.pos[idx], self.dc0[idx], self.sh[idx], self.opacity[idx], self.scale[idx], self.rot[idx])


# --------------------------------------------------------------------------------------------------
# PLY IO (binary little endian, float properties) -- PLYFileReader.cs:25-76, GaussianFileReader.cs:80-183
# --------------------------------------------------------------------------------------------------
def WritePLY(path: str, splats: InputSplatData) -> None:
    """Writes raw (pre-LinearizeData) splats in the INRIA layout: f_rest is channel-major (15 R, 15 G, 15 B)."""
    n = len(splats)
    arr = np.zeros((n, 62), dtype="<f4")
    arr[:, 0:3] = splats.pos
    arr[:, 6:9] = splats.dc0
    arr[:, 9:54] = splats.sh.transpose(0, 2, 1).reshape(n, 45)   # [N,3,15] channel-major
    arr[:, 54] = splats.opacity
    arr[:, 55:58] = splats.scale
    arr[:, 58:62] = splats.rot
    hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    hdr += [f"property float {a}" for a in PLY_ATTRS] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(arr.tobytes())


def ReadPLY(path: str) -> InputSplatData:
    """PLYFileReader.ReadFile + PLYDataToSplats + ReorderSHs (not yet linearised)."""
    with open(path, "rb") as f:
        data = f.read()
    bio = io.BytesIO(data)
    count, attrs, got_le = 0, [], False
    size_of = {"float": 4, "double": 8, "uchar": 1}
    for _ in range(9000):
        line = bio.readline()
        if not line:
            break
        line = line.rstrip(b"\n").rstrip(b"\r").decode("utf-8")
        if line == "end_header" or len(line) == 0:
            break
        tok = line.split(" ")
        if len(tok) == 3 and tok[0] == "format" and tok[1] == "binary_little_endian" and tok[2] == "1.0":
            got_le = True
        if len(tok) == 3 and tok[0] == "element" and tok[1] == "vertex":
            count = int(tok[2])
        if len(tok) == 3 and tok[0] == "property":
            attrs.append((tok[2], tok[1] if tok[1] in size_of else "none"))
    if not got_le:
        raise IOError(f"PLY {path} not supported: needs to be binary, little endian PLY format")
    np_t = {"float": "<f4", "double": "<f8", "uchar": "u1"}
    dt = np.dtype([(nm, np_t[t]) for nm, t in attrs if t != "none"])
    body = np.frombuffer(data, dtype=dt, count=count, offset=bio.tell())
    required = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                "rot_0", "rot_1", "rot_2", "rot_3"]
    fl = {nm for nm, t in attrs if t == "float"}
    missing = [r for r in required if r not in fl]
    if missing:
        raise IOError("PLY file is probably not a Gaussian Splat file? Missing properties: " + ",".join(missing))
    col = lambda nm: (body[nm].astype(f32) if nm in fl else np.zeros(count, f32))
    rest = np.stack([col(f"f_rest_{i}") for i in range(45)], axis=1)          # channel-major [N,45]
    sh = rest.reshape(count, 3, 15).transpose(0, 2, 1).copy()                  # ReorderSHs -> [N,15,3]
    return InputSplatData(
        pos=np.stack([col("x"), col("y"), col("z")], 1),
        dc0=np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1),
        sh=sh, opacity=col("opacity"),
        scale=np.stack([col("scale_0"), col("scale_1"), col("scale_2")], 1),
        rot=np.stack([col("rot_0"), col("rot_1"), col("rot_2"), col("rot_3")], 1))


# --------------------------------------------------------------------------------------------------
# GaussianUtils.cs
# --------------------------------------------------------------------------------------------------
def ExpDet(x):
    """exp(x) from fp32 operations only, one rounding each (range reduction by ln2 in two parts + Cephes' degree-5 polynomial,
    scaled by 2^n through the exponent bits): the same bits here and in the native importer (csrc/gs_import.cpp: exp_det),
    whatever libm / SIMD exp numpy was built with.  |rel. error| < 2^-22."""
    x = np.clip(np.asarray(x, f32), f32(-87.0), f32(88.0))
    n = np.rint(x * f32(1.44269504)).astype(f32)
    r = (x - n * f32(0.693359375)).astype(f32)
    r = (r - n * f32(-2.12194440e-4)).astype(f32)
    p = np.full_like(r, f32(1.9875691500e-4))
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        p = (p * r + f32(c)).astype(f32)
    p = (p * (r * r) + r).astype(f32)
    p = (p + f32(1.0)).astype(f32)
    scale = ((n.astype(np.int32) + 127) << 23).astype(np.uint32).view(f32)
    return (p * scale).astype(f32)


def Sigmoid(v):                      # GaussianUtils.cs:9-12
    v = np.asarray(v, f32)
    return (f32(1) / (f32(1) + ExpDet(-v))).astype(f32)


def SH0ToColor(dc0):                 # GaussianUtils.cs:14-18
    return (np.asarray(dc0, f32) * f32(0.2820948) + f32(0.5)).astype(f32)


def SquareCentered01(x):             # GaussianUtils.cs:25-30
    x = np.asarray(x, f32) - f32(0.5)
    x = x * (x * np.sign(x).astype(f32))
    return (x * f32(2.0) + f32(0.5)).astype(f32)


def InvSquareCentered01(x):          # GaussianUtils.cs:32-38
    x = (np.asarray(x, f32) - f32(0.5)) * f32(0.5)
    x = np.sqrt(np.abs(x)).astype(f32) * np.sign(x).astype(f32)
    return (x + f32(0.5)).astype(f32)


def PackSmallest3Rotation(q):        # GaussianUtils.cs:46-76 ; q is [N,4] xyzw
    q = np.asarray(q, f32)
    index = np.argmax(np.abs(q), axis=1)        # first max wins, like the strict '>' chain
    out = q.copy()
    m = index == 0; out[m] = q[m][:, [1, 2, 3, 0]]
    m = index == 1; out[m] = q[m][:, [0, 2, 3, 1]]
    m = index == 2; out[m] = q[m][:, [0, 1, 3, 2]]
    sgn = np.where(out[:, 3:4] >= 0, f32(1), f32(-1)).astype(f32)
    three = out[:, :3] * sgn
    three = (three * f32(np.sqrt(2.0))) * f32(0.5) + f32(0.5)
    return np.concatenate([three.astype(f32), (index.astype(f32) / f32(3.0))[:, None]], axis=1).astype(f32)


def _part1by2(x):                    # GaussianUtils.cs:81-90
    x = x.astype(np.uint64) & np.uint64(0x1fffff)
    x = (x ^ (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    x = (x ^ (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    x = (x ^ (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    x = (x ^ (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    x = (x ^ (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    return x


def MortonEncode3(v):                # GaussianUtils.cs:92-95 ; v [N,3] uint
    return (_part1by2(v[:, 2]) << np.uint64(2)) | (_part1by2(v[:, 1]) << np.uint64(1)) | _part1by2(v[:, 0])


def DecodeMorton2D_16x16(t):         # GaussianUtils.cs:98-105
    t = np.asarray(t, np.uint32)
    t = (t & 0xFF) | ((t & 0xFE) << 7)
    t &= 0x5555
    t = (t ^ (t >> 1)) & 0x3333
    t = (t ^ (t >> 2)) & 0x0f0f
    return t & 0xF, t >> 8


def SplatIndexToTextureIndex(idx):   # GaussianSplatAssetCreator.cs:863-871
    idx = np.asarray(idx, np.uint32)
    x, y = DecodeMorton2D_16x16(idx)
    width = kTextureWidth // 16
    t = idx >> 8
    px = (t % width) * 16 + x
    py = (t // width) * 16 + y
    return (py.astype(np.int64) * kTextureWidth + px.astype(np.int64))


# --------------------------------------------------------------------------------------------------
# GaussianFileReader.LinearizeData (:211-233)
# --------------------------------------------------------------------------------------------------
def LinearizeData(s: InputSplatData) -> InputSplatData:
    wxyz = s.rot.astype(f32)
    nrm = np.sqrt(np.sum(wxyz * wxyz, axis=1, keepdims=True, dtype=f32)).astype(f32)
    q = (wxyz / nrm)[:, [1, 2, 3, 0]]                   # NormalizeSwizzleRotation: normalize(wxyz).yzwx
    return InputSplatData(
        pos=s.pos.astype(f32), dc0=SH0ToColor(s.dc0), sh=s.sh.astype(f32),
        opacity=Sigmoid(s.opacity), scale=np.abs(ExpDet(s.scale.astype(f32))).astype(f32),
        rot=PackSmallest3Rotation(q))


def ReorderMorton(s: InputSplatData):
    """GaussianSplatAssetCreator.cs:362-429.  Returns (reordered splats, boundsMin, boundsMax)."""
    bmin = s.pos.min(axis=0).astype(f32)
    bmax = s.pos.max(axis=0).astype(f32)
    inv = (f32(1.0) / (bmax - bmin)).astype(f32)
    p = ((s.pos - bmin) * inv * f32((1 << 21) - 1)).astype(f32)
    ip = p.astype(np.uint32)
    code = MortonEncode3(ip)
    order = np.argsort(code, kind="stable")            # (code, index) lexicographic == stable by code
    return s.take(order), bmin, bmax


# --------------------------------------------------------------------------------------------------
# Encoders: GaussianSplatAssetCreator.cs:705-758
# --------------------------------------------------------------------------------------------------
def _sat(v):
    return np.clip(np.asarray(v, f32), f32(0), f32(1))


def _q(v, k):
    # truncating cast, through int64 like the native importer's (uint32_t)(int64_t)(v * k): a NaN (the zero quaternion's normalize()) becomes 0 in both --
    # the reference's own (uint)(NaN * 1023.5f) is unspecified, so this is a definition, not a restatement
    with np.errstate(invalid="ignore"):
        return (v.astype(f32) * f32(k + 0.5)).astype(np.int64).astype(np.uint32)


def EncodeFloat3ToNorm16(v):
    v = _sat(v)
    return _q(v[:, 0], 65535).astype(np.uint64) | (_q(v[:, 1], 65535).astype(np.uint64) << np.uint64(16)) | \
        (_q(v[:, 2], 65535).astype(np.uint64) << np.uint64(32))


def EncodeFloat3ToNorm11(v, saturate=True):
    v = _sat(v) if saturate else np.asarray(v, f32)
    return _q(v[:, 0], 2047) | (_q(v[:, 1], 1023) << 11) | (_q(v[:, 2], 2047) << 21)


def EncodeFloat3ToNorm655(v):
    v = _sat(v)
    return (_q(v[:, 0], 63) | (_q(v[:, 1], 31) << 6) | (_q(v[:, 2], 31) << 11)).astype(np.uint16)


def EncodeFloat3ToNorm565(v, saturate=False):
    v = _sat(v) if saturate else np.asarray(v, f32)
    return (_q(v[:, 0], 31) | (_q(v[:, 1], 63) << 5) | (_q(v[:, 2], 31) << 11)).astype(np.uint16)


def EncodeQuatToNorm10(v):
    return _q(v[:, 0], 1023) | (_q(v[:, 1], 1023) << 10) | (_q(v[:, 2], 1023) << 20) | (_q(v[:, 3], 3) << 30)


def EmitEncodedVector(v, fmt: VectorFormat) -> np.ndarray:
    """Returns [N, size] uint8 (EmitEncodedVector, :727-758)."""
    n = len(v)
    fmt = VectorFormat(fmt)
    if fmt == VectorFormat.Float32:
        return np.ascontiguousarray(v.astype("<f4")).view(np.uint8).reshape(n, 12)
    if fmt == VectorFormat.Norm16:
        enc = EncodeFloat3ToNorm16(v)
        out = np.zeros((n, 6), np.uint8)
        out[:, 0:4] = (enc & np.uint64(0xffffffff)).astype("<u4").view(np.uint8).reshape(n, 4)
        out[:, 4:6] = (enc >> np.uint64(32)).astype("<u2").view(np.uint8).reshape(n, 2)
        return out
    if fmt == VectorFormat.Norm11:
        return EncodeFloat3ToNorm11(v).astype("<u4").view(np.uint8).reshape(n, 4)
    return EncodeFloat3ToNorm655(v).astype("<u2").view(np.uint8).reshape(n, 2)


def _f32tof16_pair(lo, hi):
    lo16 = np.asarray(lo, f32).astype(np.float16).view(np.uint16).astype(np.uint32)
    hi16 = np.asarray(hi, f32).astype(np.float16).view(np.uint16).astype(np.uint32)
    return lo16 | (hi16 << 16)


def _pad8(b: np.ndarray) -> np.ndarray:
    n = (len(b) + 7) // 8 * 8                     # NextMultipleOf(dataLen, 8)  (:812-816, :835-843)
    if n == len(b):
        return b
    out = np.zeros(n, np.uint8)
    out[:len(b)] = b
    return out


# --------------------------------------------------------------------------------------------------
# ClusterSHs (:476-518): K-means over the 45-D SH vectors -> fp16 palette + per-splat index
# --------------------------------------------------------------------------------------------------
def ClusterSHs(sh: np.ndarray, count: int, iterations: int = 4, sample: int = 200_000):
    """sh [N,15,3] f32 -> (means [count,45] f32, indices [N] int).  Deterministic (no RNG), so that the native importer
    (csrc/gs_import.cpp cluster_shs) emits the same palette: seeds and the training subset are stride samples, `iterations`
    Lloyd steps on the subset (nearest mean by |c|^2 - 2 x.c in float64, first minimum; new mean = float64 sum of its points
    in point order / count, rounded to fp32; an empty cluster keeps its mean), then one assignment pass over all splats."""
    n = len(sh)
    x = np.ascontiguousarray(sh.reshape(n, 45), f32)
    means = x[(np.arange(count, dtype=np.int64) * n) // count].copy()

    def assign(pts, cen):
        out = np.empty(len(pts), np.int64)
        c64 = cen.astype(np.float64)
        c2 = (c64 * c64).sum(1)
        for i in range(0, len(pts), 8192):
            blk = pts[i:i + 8192].astype(np.float64)
            d = c2[None, :] - 2.0 * (blk @ c64.T)
            out[i:i + 8192] = d.argmin(1)
        return out

    sub = x if n <= sample else x[(np.arange(sample, dtype=np.int64) * n) // sample]
    for _ in range(iterations):
        idx = assign(sub, means)
        sums = np.zeros((count, 45), np.float64)
        np.add.at(sums, idx, sub.astype(np.float64))              # unbuffered: strictly in point order
        cnt = np.bincount(idx, minlength=count)
        nz = cnt > 0
        means[nz] = (sums[nz] / cnt[nz, None].astype(np.float64)).astype(f32)
    return means, assign(x, means)


# --------------------------------------------------------------------------------------------------
# CreateAsset
# --------------------------------------------------------------------------------------------------
def CreateAssetFromSplats(raw: InputSplatData, quality: str = "Medium", *, formatPos=None, formatScale=None,
                          formatColor=None, formatSH=None, name: str = "asset", morton: bool = True,
                          linearize: bool = True) -> GaussianSplatAsset:
    """GaussianSplatAssetCreator.CreateAsset (:247-340) minus the Unity asset database."""
    fp, fs, fc, fsh = QUALITY[quality]
    fp = VectorFormat(fp if formatPos is None else formatPos)
    fs = VectorFormat(fs if formatScale is None else formatScale)
    fc = ColorFormat(fc if formatColor is None else formatColor)
    fsh = SHFormat(fsh if formatSH is None else formatSH)
    s = LinearizeData(raw) if linearize else raw
    if morton:
        s, bmin, bmax = ReorderMorton(s)
    else:
        bmin, bmax = s.pos.min(0).astype(f32), s.pos.max(0).astype(f32)
    n = len(s)
    pos, scale, dc0, opacity, sh = (s.pos.astype(f32).copy(), s.scale.astype(f32).copy(), s.dc0.astype(f32).copy(),
                                    s.opacity.astype(f32).copy(), s.sh.astype(f32).copy())

    # isUsingChunks (:54-58): any lossy format
    use_chunks = not (fp == VectorFormat.Float32 and fs == VectorFormat.Float32 and
                      fc == ColorFormat.Float32x4 and fsh == SHFormat.Float32)
    chunk_bytes = None
    if use_chunks:                                     # CalcChunkDataJob (:520-638)
        nchunks = (n + kChunkSize - 1) // kChunkSize
        scale = np.sqrt(np.sqrt(np.sqrt(scale))).astype(f32)       # math.pow(s, 1/8) as three correctly rounded roots (as csrc/gs_import.cpp)
        opacity = SquareCentered01(opacity)
        pad = nchunks * kChunkSize - n

        def cmin_cmax(a):                              # a: [N, C] -> per-chunk min/max [nchunks, C]
            c = a.shape[1]
            lo = np.concatenate([a, np.full((pad, c), np.inf, f32)]).reshape(nchunks, kChunkSize, c).min(1)
            hi = np.concatenate([a, np.full((pad, c), -np.inf, f32)]).reshape(nchunks, kChunkSize, c).max(1)
            hi = np.maximum(hi, (lo + f32(1.0e-5)).astype(f32)).astype(f32)
            return lo.astype(f32), hi

        pmin, pmax = cmin_cmax(pos)
        smin, smax = cmin_cmax(scale)
        col4 = np.concatenate([dc0, opacity[:, None]], 1)
        cmin, cmax = cmin_cmax(col4)
        shlo = sh.min(axis=1)
        shhi = sh.max(axis=1)                          # one shared rgb min/max over all 15 coefficients
        hmin, _ = cmin_cmax(shlo)
        hmax0 = np.concatenate([shhi, np.full((pad, 3), -np.inf, f32)]).reshape(nchunks, kChunkSize, 3).max(1).astype(f32)
        hmax = np.maximum(hmax0, (hmin + f32(1.0e-5)).astype(f32)).astype(f32)      # chunkMaxshs = max(chunkMaxshs, chunkMinshs + 1e-5) :595

        chunks = np.zeros(nchunks, CHUNK_DTYPE)
        for k, nm in enumerate(("posX", "posY", "posZ")):
            chunks[nm][:, 0] = pmin[:, k]
            chunks[nm][:, 1] = pmax[:, k]
        for k, nm in enumerate(("sclX", "sclY", "sclZ")):
            chunks[nm] = _f32tof16_pair(smin[:, k], smax[:, k])
        for k, nm in enumerate(("colR", "colG", "colB", "colA")):
            chunks[nm] = _f32tof16_pair(cmin[:, k], cmax[:, k])
        for k, nm in enumerate(("shR", "shG", "shB")):
            chunks[nm] = _f32tof16_pair(hmin[:, k], hmax[:, k])
        chunk_bytes = chunks.view(np.uint8).reshape(-1).copy()

        ci = np.arange(n) // kChunkSize                # normalise with the fp32 (un-rounded) bounds (:613-637)
        pos = ((pos - pmin[ci]) / (pmax[ci] - pmin[ci])).astype(f32)
        scale = ((scale - smin[ci]) / (smax[ci] - smin[ci])).astype(f32)
        dc0 = ((dc0 - cmin[ci, :3]) / (cmax[ci, :3] - cmin[ci, :3])).astype(f32)
        opacity = ((opacity - cmin[ci, 3]) / (cmax[ci, 3] - cmin[ci, 3])).astype(f32)
        sh = ((sh - hmin[ci][:, None, :]) / (hmax[ci] - hmin[ci])[:, None, :]).astype(f32)

    # positions (:760-774, :807-827)
    pos_bytes = _pad8(EmitEncodedVector(pos, fp).reshape(-1))

    # SH palette for the Cluster* formats: raw (not chunk-normalised) SH vectors (:286-291, :476-518)
    sh_means = sh_index = None
    if fsh > SHFormat.Norm6:
        from .asset import GetSHCount
        k = GetSHCount(fsh, n)
        if k >= n:
            raise ValueError(f"{fsh.name} needs more than {k} splats (the reference falls back to unclustered data there)")
        sh_means, sh_index = ClusterSHs(s.sh.astype(f32), k)

    # other: rot + scale (+ u16 SH index) (:776-805)
    rot_enc = EncodeQuatToNorm10(s.rot.astype(f32)).astype("<u4").view(np.uint8).reshape(n, 4)
    parts = [rot_enc, EmitEncodedVector(scale, fs)]
    if sh_index is not None:
        parts.append(sh_index.astype("<u2").view(np.uint8).reshape(n, 2))
    oth = np.concatenate(parts, axis=1)
    oth_bytes = _pad8(np.ascontiguousarray(oth).reshape(-1))

    # colour texture (:873-932)
    w, h = CalcTextureSize(n)
    tex = np.zeros((w * h, 4), f32)
    tex[SplatIndexToTextureIndex(np.arange(n, dtype=np.uint32))] = np.concatenate([dc0, opacity[:, None]], 1)
    if fc == ColorFormat.Float32x4:
        col_bytes = tex.astype("<f4").view(np.uint8).reshape(-1)
    elif fc == ColorFormat.Float16x4:
        col_bytes = tex.astype("<f2").view(np.uint8).reshape(-1)
    elif fc == ColorFormat.BC7:
        # EditorUtility.CompressTexture(tex, BC7, 100) (:900-903).  Unity's compressor cannot be reproduced; every block is
        # encoded in mode 6 here (bc7.py) -- a conforming stream that any BC7 decoder, ours included, reads like Unity's
        from . import bc7
        col_bytes = bc7.encode_texture_mode6(tex.reshape(h, w, 4))
    else:
        p = _sat(tex)
        enc = _q(p[:, 0], 255) | (_q(p[:, 1], 255) << 8) | (_q(p[:, 2], 255) << 16) | (_q(p[:, 3], 255) << 24)
        col_bytes = enc.astype("<u4").view(np.uint8).reshape(-1)
    assert len(col_bytes) == w * h * GetColorSize(fc)

    # SH (:934-1037)
    if sh_means is not None:                           # ConvertSHClustersJob (:443-474): SHTableItemFloat16 per cluster
        item = np.zeros((len(sh_means), 48), "<f2")
        item[:, :45] = sh_means.astype(np.float16)
        sh_bytes = item.view(np.uint8).reshape(-1)
    elif fsh == SHFormat.Float32:
        item = np.zeros((n, 48), "<f4")
        item[:, :45] = sh.reshape(n, 45)
        sh_bytes = item.view(np.uint8).reshape(-1)
    elif fsh == SHFormat.Float16:
        item = np.zeros((n, 48), "<f2")
        item[:, :45] = sh.reshape(n, 45).astype(np.float16)
        sh_bytes = item.view(np.uint8).reshape(-1)
    elif fsh == SHFormat.Norm11:
        enc = EncodeFloat3ToNorm11(sh.reshape(n * 15, 3), saturate=False)    # CreateSHDataJob does not saturate
        sh_bytes = enc.astype("<u4").view(np.uint8).reshape(-1)
    else:
        enc = EncodeFloat3ToNorm565(sh.reshape(n * 15, 3)).reshape(n, 15)
        item = np.zeros((n, 16), "<u2")
        item[:, :15] = enc
        sh_bytes = item.view(np.uint8).reshape(-1)

    a = GaussianSplatAsset(splatCount=n, posFormat=fp, scaleFormat=fs, shFormat=fsh, colorFormat=fc,
                           posData=np.ascontiguousarray(pos_bytes), otherData=np.ascontiguousarray(oth_bytes),
                           colorData=np.ascontiguousarray(col_bytes), shData=np.ascontiguousarray(sh_bytes),
                           chunkData=chunk_bytes, boundsMin=tuple(map(float, bmin)), boundsMax=tuple(map(float, bmax)),
                           name=name)
    a.dataHash = a.ComputeDataHash()
    a.Validate()
    return a


def CreateAssetFromSplatsNative(raw: InputSplatData, quality: str = "Medium", *, formatPos=None, formatScale=None, formatColor=None,
                                formatSH=None, name: str = "asset", morton: bool = True, linearize: bool = True) -> GaussianSplatAsset:
    """The same asset through the native importer of libgsplat_hip.so (gs_import_encode, csrc/gs_import.cpp: multi-threaded
    host C++).  Bit-identical to CreateAssetFromSplats for every format and preset (tests/test_import.py), Cluster* palettes and BC7 included."""
    import ctypes as C
    from . import _lib
    from ._abi import gs_import_formats, gs_import_input
    fp, fs, fc, fsh = QUALITY[quality]
    fp = VectorFormat(fp if formatPos is None else formatPos)
    fs = VectorFormat(fs if formatScale is None else formatScale)
    fc = ColorFormat(fc if formatColor is None else formatColor)
    fsh = SHFormat(fsh if formatSH is None else formatSH)
    n = len(raw)
    arrs = [np.ascontiguousarray(a, f32) for a in (raw.pos, raw.dc0, raw.sh.reshape(n, 45), raw.opacity, raw.scale, raw.rot)]
    inp = gs_import_input(n, *[a.ctypes.data for a in arrs])
    fmt = gs_import_formats(int(fp), int(fs), int(fc), int(fsh), int(bool(linearize)), int(bool(morton)))
    sizes = (C.c_uint64 * 5)()
    _lib.check(_lib.lib().gs_import_blob_sizes(n, C.byref(fmt), sizes), "gs_import_blob_sizes")
    blobs = [np.zeros(int(sz), np.uint8) if sz else None for sz in sizes]
    ptrs = (C.c_void_p * 5)(*[b.ctypes.data if b is not None else None for b in blobs])
    bmin, bmax = (C.c_float * 3)(), (C.c_float * 3)()
    _lib.check(_lib.lib().gs_import_encode(C.byref(inp), C.byref(fmt), ptrs, sizes, bmin, bmax), "gs_import_encode")
    a = GaussianSplatAsset(splatCount=n, posFormat=fp, scaleFormat=fs, shFormat=fsh, colorFormat=fc, posData=blobs[0], otherData=blobs[1],
                           colorData=blobs[2], shData=blobs[3], chunkData=blobs[4], boundsMin=tuple(bmin), boundsMax=tuple(bmax), name=name)
    a.dataHash = a.ComputeDataHash()
    a.Validate()
    return a


# --------------------------------------------------------------------------------------------------
# SPZ (Niantic / Scaniverse), Editor/Utils/SPZFileReader.cs
# --------------------------------------------------------------------------------------------------
SPZ_MAGIC = 0x5053474e


def _spz_sh_coeffs(level: int) -> int:
    return {0: 0, 1: 3, 2: 8, 3: 15}.get(level, 0)


def WriteSPZ(path: str, raw: InputSplatData, fract_bits: int = 12, sh_level: int = 3) -> None:
    """Test-fixture writer: raw (PLY-domain) splats quantised the way the SPZ format stores them (24-bit fixed-point positions,
    log-scale bytes (s + 10) * 16, sigmoid opacity bytes, colour bytes (dc0 * 0.15 + 0.5) * 255, unit-quaternion xyz bytes with
    w >= 0, SH bytes sh * 128 + 128), gzip-compressed."""
    import gzip
    import struct
    n = len(raw)
    k = _spz_sh_coeffs(sh_level)
    pos = np.clip(np.rint(raw.pos.astype(np.float64) * (1 << fract_bits)), -(1 << 23), (1 << 23) - 1).astype(np.int64) & 0xffffff
    pos_b = np.stack([(pos >> s) & 0xff for s in (0, 8, 16)], axis=-1).astype(np.uint8).reshape(n, 9)
    alpha = np.clip(np.rint(255.0 / (1.0 + np.exp(-raw.opacity.astype(np.float64)))), 0, 255).astype(np.uint8)
    col = np.clip(np.rint((raw.dc0.astype(np.float64) * 0.15 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    scale = np.clip(np.rint((raw.scale.astype(np.float64) + 10.0) * 16.0), 0, 255).astype(np.uint8)
    q = raw.rot.astype(np.float64)                                   # (w, x, y, z)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    q = np.where(q[:, :1] < 0, -q, q)
    rot = np.clip(np.rint(q[:, 1:4] * 127.5 + 127.5), 0, 255).astype(np.uint8)
    sh = np.clip(np.rint(raw.sh[:, :k, :].astype(np.float64) * 128.0 + 128.0), 0, 255).astype(np.uint8).reshape(n, k * 3)
    body = struct.pack("<IIII", SPZ_MAGIC, 2, n, sh_level | (fract_bits << 8)) + pos_b.tobytes() + alpha.tobytes() + col.tobytes() + \
        scale.tobytes() + rot.tobytes() + sh.tobytes()
    with open(path, "wb") as f:
        f.write(gzip.compress(body, compresslevel=1))


def ReadSPZ(path: str) -> InputSplatData:
    """SPZFileReader.ReadFile (:66-139) + UnpackDataJob (:141-205): ALREADY LINEAR splats (use linearize=False)."""
    import gzip
    import struct
    with open(path, "rb") as f:
        try:
            data = gzip.decompress(f.read())
        except Exception as e:
            raise IOError(f"SPZ {path} read error: not a gzip stream") from e
    if len(data) < 16:
        raise IOError(f"SPZ {path} read error, failed to read header")
    magic, version, n, word = struct.unpack_from("<IIiI", data, 0)
    if magic != SPZ_MAGIC:
        raise IOError(f"SPZ {path} read error, header magic unexpected {magic}")
    if version != 2:
        raise IOError(f"SPZ {path} read error, header version unexpected {version}")
    sh_level, fract_bits = word & 0xff, (word >> 8) & 0xff
    if n < 1 or n > 10_000_000:
        raise IOError(f"SPZ {path} read error, out of range splat count {n}")
    if sh_level > 3:
        raise IOError(f"SPZ {path} read error, out of range SH level {sh_level}")
    if fract_bits > 24:
        raise IOError(f"SPZ {path} read error, out of range fractional bits {fract_bits}")
    k = _spz_sh_coeffs(sh_level)
    o, parts = 16, []
    for sz in (n * 9, n, n * 3, n * 3, n * 3, n * 3 * k):
        parts.append(np.frombuffer(data, np.uint8, count=min(sz, max(0, len(data) - o)), offset=min(o, len(data))))
        o += sz
    if len(data) < o:
        raise IOError(f"SPZ {path} read error, file smaller than it should be")
    ppos, palpha, pcol, pscale, prot, psh = parts
    b = ppos.reshape(n, 3, 3).astype(np.int32)
    fx = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    fx = np.where(fx & 0x800000, fx - (1 << 24), fx)
    pos = fx.astype(f32) * f32(1.0 / (1 << fract_bits))
    scale = np.abs(ExpDet(pscale.reshape(n, 3).astype(f32) / f32(16.0) - f32(10.0))).astype(f32)
    xyz = prot.reshape(n, 3).astype(f32) * f32(1.0 / 127.5) - f32(1.0)
    sq = ((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]).astype(f32)
    w = np.sqrt(np.maximum(f32(0.0), f32(1.0) - sq)).astype(f32)
    q = np.concatenate([xyz, w[:, None]], 1).astype(f32)
    l2 = (((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3]).astype(f32)
    q = (q * (f32(1.0) / np.sqrt(l2).astype(f32))[:, None]).astype(f32)
    col = ((pcol.reshape(n, 3).astype(f32) / f32(255.0) - f32(0.5)) / f32(0.15)).astype(f32)
    # UnpackSH reads 15 coefficients from index * shCoeffs * 3 whatever the level (:178-193): for levels < 3 it runs on into the
    # following splats' bytes (restated as is; bytes past the end of the array read as 128 = 0.0)
    at = (np.arange(n, dtype=np.int64) * (3 * k))[:, None] + np.arange(45, dtype=np.int64)[None, :]
    padded = np.concatenate([psh, np.full(45, 128, np.uint8)])
    shb = padded[np.minimum(at, len(psh) + 44 if len(psh) else 44)] if len(psh) else np.full((n, 45), 128, np.uint8)
    shb = np.where(at < len(psh), shb, 128)
    sh = ((shb.astype(f32) - f32(128.0)) / f32(128.0)).reshape(n, 15, 3).astype(f32)
    return InputSplatData(pos=pos, dc0=SH0ToColor(col), sh=sh, opacity=(palpha.astype(f32) / f32(255.0)).astype(f32), scale=scale,
                          rot=PackSmallest3Rotation(q))


def _ReadNativeHandle(open_fn_name: str, path: str) -> InputSplatData:
    import ctypes as C
    from . import _lib
    from ._abi import gs_import_input
    l = _lib.lib()
    h, cnt = C.c_void_p(), C.c_uint32()
    _lib.check(getattr(l, open_fn_name)(path.encode("utf-8"), C.byref(h), C.byref(cnt)), open_fn_name)
    try:
        inp = gs_import_input()
        _lib.check(l.gs_ply_arrays(h, C.byref(inp)), "gs_ply_arrays")
        n = cnt.value
        get = lambda ptr, k: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n * k,)).copy()
        return InputSplatData(pos=get(inp.pos, 3).reshape(n, 3), dc0=get(inp.dc0, 3).reshape(n, 3), sh=get(inp.sh, 45).reshape(n, 15, 3),
                              opacity=get(inp.opacity, 1), scale=get(inp.scale, 3).reshape(n, 3), rot=get(inp.rot, 4).reshape(n, 4))
    finally:
        l.gs_ply_close(h)


def ReadSPZNative(path: str) -> InputSplatData:
    """ReadSPZ through the native reader (gs_spz_open, csrc/gs_import.cpp)."""
    return _ReadNativeHandle("gs_spz_open", path)


def ReadPLYNative(path: str) -> InputSplatData:
    """ReadPLY through the native reader (gs_ply_open / gs_ply_arrays, csrc/gs_import.cpp); the arrays are copied out."""
    import ctypes as C
    from . import _lib
    from ._abi import gs_import_input
    h, n = C.c_void_p(), C.c_uint32()
    _lib.check(_lib.lib().gs_ply_open(path.encode(), C.byref(h), C.byref(n)), "gs_ply_open")
    try:
        arr = gs_import_input()
        _lib.check(_lib.lib().gs_ply_arrays(h, C.byref(arr)), "gs_ply_arrays")
        cnt = n.value
        get = lambda ptr, k: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(cnt * k,)).copy()
        return InputSplatData(pos=get(arr.pos, 3).reshape(cnt, 3), dc0=get(arr.dc0, 3).reshape(cnt, 3), sh=get(arr.sh, 45).reshape(cnt, 15, 3),
                              opacity=get(arr.opacity, 1), scale=get(arr.scale, 3).reshape(cnt, 3), rot=get(arr.rot, 4).reshape(cnt, 4))
    finally:
        _lib.lib().gs_ply_close(h)


kCamerasJson = "cameras.json"


def LoadJsonCamerasFile(curPath: str, doImport: bool = True):
    """GaussianSplatAssetCreator.cs:1068-1118: look for cameras.json in the input file's directory and its ancestors; every entry's
    `rotation` is a VIEW matrix whose columns are the camera's axes (axisx = column 0 ...), the y and z axes are negated, and the
    field of view is the constant 25 the reference stores (its "@TODO": callers set the real one, GaussianSplatValidator.cs:56-58)."""
    import json
    import os
    from .asset import CameraInfo
    if not doImport:
        return None
    while True:
        d = os.path.dirname(curPath)
        if not d or not os.path.isdir(d):
            return None
        camerasPath = os.path.join(d, kCamerasJson)
        if os.path.isfile(camerasPath):
            break
        if d == curPath:
            return None
        curPath = d
    with open(camerasPath) as f:
        jsonCameras = json.load(f)
    if not jsonCameras:
        return None
    result = []
    for jc in jsonCameras:
        rot = jc["rotation"]
        col = lambda k: np.array([rot[0][k], rot[1][k], rot[2][k]], np.float32)
        axisx, axisy, axisz = col(0), col(1) * np.float32(-1), col(2) * np.float32(-1)
        result.append(CameraInfo(pos=tuple(float(v) for v in jc["position"][:3]), axisX=tuple(float(v) for v in axisx),
                                 axisY=tuple(float(v) for v in axisy), axisZ=tuple(float(v) for v in axisz), fov=25.0))
    return result


def CreateAsset(path: str, quality: str = "Medium", **kw) -> GaussianSplatAsset:
    """.ply / .spz file -> asset (GaussianFileReader.ReadFile :45-71 + CreateAsset): PLY data is linearised, SPZ data already is."""
    importCameras = kw.pop("importCameras", True)
    if path.lower().endswith(".spz"):
        a = CreateAssetFromSplats(ReadSPZ(path), quality, linearize=False, **kw)
    elif path.lower().endswith(".ply"):
        a = CreateAssetFromSplats(ReadPLY(path), quality, **kw)
    else:
        raise IOError(f"File {path} is not a supported format")
    a.cameras = LoadJsonCamerasFile(path, importCameras) or []           # GaussianSplatAssetCreator.cs:274
    return a
