"""GaussianSplatAsset: the chunked / quantised splat data contract, kept byte for byte.

Host-side mirror of the reference's ``GaussianSplatAsset`` ScriptableObject
(/root/reference/package/Runtime/GaussianSplatAsset.cs:13-246).  Same enum names and values,
same size calculators, same five blobs (pos / other / color / sh / chunk).  Nothing here touches
the GPU; the renderer uploads the blobs through the C-ABI (include/gsplat_c.h, gs_asset_create).
"""
from __future__ import annotations

import enum
import hashlib
import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

kCurrentVersion = 2023_10_20   # GaussianSplatAsset.cs:13
kChunkSize = 256               # GaussianSplatAsset.cs:14
kTextureWidth = 2048           # GaussianSplatAsset.cs:15
kMaxSplats = 8_600_000         # GaussianSplatAsset.cs:16 (a reference editor limit; not enforced by the HIP path)


class VectorFormat(enum.IntEnum):
    """GaussianSplatAsset.cs:31-37; must match VECTOR_FMT_* (GaussianSplatting.hlsl:319-323)."""
    Float32 = 0   # 12 bytes: 32F.32F.32F
    Norm16 = 1    # 6 bytes: 16.16.16
    Norm11 = 2    # 4 bytes: 11.10.11
    Norm6 = 3     # 2 bytes: 6.5.5


class ColorFormat(enum.IntEnum):
    """GaussianSplatAsset.cs:51-57."""
    Float32x4 = 0
    Float16x4 = 1
    Norm8x4 = 2
    BC7 = 3


class SHFormat(enum.IntEnum):
    """GaussianSplatAsset.cs:70-81."""
    Float32 = 0
    Float16 = 1
    Norm11 = 2
    Norm6 = 3
    Cluster64k = 4
    Cluster32k = 5
    Cluster16k = 6
    Cluster8k = 7
    Cluster4k = 8


def GetVectorSize(fmt: VectorFormat) -> int:
    """GaussianSplatAsset.cs:39-49 (raises like the C# ArgumentOutOfRangeException)."""
    try:
        return {VectorFormat.Float32: 12, VectorFormat.Norm16: 6, VectorFormat.Norm11: 4, VectorFormat.Norm6: 2}[VectorFormat(fmt)]
    except (KeyError, ValueError):
        raise ValueError(f"fmt out of range: {fmt}")


def GetColorSize(fmt: ColorFormat) -> int:
    """GaussianSplatAsset.cs:58-68."""
    try:
        return {ColorFormat.Float32x4: 16, ColorFormat.Float16x4: 8, ColorFormat.Norm8x4: 4, ColorFormat.BC7: 1}[ColorFormat(fmt)]
    except (KeyError, ValueError):
        raise ValueError(f"fmt out of range: {fmt}")


def GetOtherSizeNoSHIndex(scaleFormat: VectorFormat) -> int:
    """GaussianSplatAsset.cs:130-133: 4 B rotation (10.10.10.2) + scale vector."""
    return 4 + GetVectorSize(scaleFormat)


def GetSHCount(fmt: SHFormat, splatCount: int) -> int:
    """GaussianSplatAsset.cs:135-150."""
    fmt = SHFormat(fmt)
    if fmt <= SHFormat.Norm6:
        return splatCount
    return {SHFormat.Cluster64k: 64, SHFormat.Cluster32k: 32, SHFormat.Cluster16k: 16,
            SHFormat.Cluster8k: 8, SHFormat.Cluster4k: 4}[fmt] * 1024


def CalcTextureSize(splatCount: int) -> Tuple[int, int]:
    """GaussianSplatAsset.cs:152-160: 2048 wide, height rounded up to the 16-texel Morton tile."""
    width = kTextureWidth
    height = max(1, (splatCount + width - 1) // width)
    height = (height + 15) // 16 * 16
    return width, height


_SH_ITEM_SIZE = {SHFormat.Float32: 192, SHFormat.Float16: 96, SHFormat.Norm11: 60, SHFormat.Norm6: 32}


def CalcPosDataSize(splatCount: int, formatPos: VectorFormat) -> int:
    return splatCount * GetVectorSize(formatPos)


def CalcOtherDataSize(splatCount: int, formatScale: VectorFormat) -> int:
    return splatCount * GetOtherSizeNoSHIndex(formatScale)


def CalcColorDataSize(splatCount: int, formatColor: ColorFormat) -> int:
    w, h = CalcTextureSize(splatCount)
    return w * h * GetColorSize(formatColor)


def CalcSHDataSize(splatCount: int, formatSh: SHFormat) -> int:
    """GaussianSplatAsset.cs:187-198."""
    formatSh = SHFormat(formatSh)
    shCount = GetSHCount(formatSh, splatCount)
    if formatSh in _SH_ITEM_SIZE:
        return shCount * _SH_ITEM_SIZE[formatSh]
    return shCount * 96 + splatCount * 2


def CalcChunkDataSize(splatCount: int) -> int:
    return (splatCount + kChunkSize - 1) // kChunkSize * 64


# ChunkInfo: 64 bytes (GaussianSplatAsset.cs:231-237 / GaussianSplatting.hlsl:196-202)
CHUNK_DTYPE = np.dtype([
    ("colR", "<u4"), ("colG", "<u4"), ("colB", "<u4"), ("colA", "<u4"),
    ("posX", "<f4", (2,)), ("posY", "<f4", (2,)), ("posZ", "<f4", (2,)),
    ("sclX", "<u4"), ("sclY", "<u4"), ("sclZ", "<u4"),
    ("shR", "<u4"), ("shG", "<u4"), ("shB", "<u4"),
])
assert CHUNK_DTYPE.itemsize == 64


@dataclass
class CameraInfo:
    """GaussianSplatAsset.cs:239-246."""
    pos: Tuple[float, float, float]
    axisX: Tuple[float, float, float]
    axisY: Tuple[float, float, float]
    axisZ: Tuple[float, float, float]
    fov: float


@dataclass
class GaussianSplatAsset:
    """The five byte blobs + format enums.  Field names follow the C# properties
    (formatVersion, splatCount, posFormat, scaleFormat, shFormat, colorFormat, posData, otherData,
    colorData, shData, chunkData, cameras, boundsMin, boundsMax, dataHash)."""
    splatCount: int = 0
    posFormat: VectorFormat = VectorFormat.Norm11
    scaleFormat: VectorFormat = VectorFormat.Norm11
    shFormat: SHFormat = SHFormat.Norm11
    colorFormat: ColorFormat = ColorFormat.Float32x4
    posData: Optional[np.ndarray] = None      # uint8 arrays
    otherData: Optional[np.ndarray] = None
    colorData: Optional[np.ndarray] = None
    shData: Optional[np.ndarray] = None
    chunkData: Optional[np.ndarray] = None    # None / empty => no chunking (all-fp32 asset)
    cameras: List[CameraInfo] = field(default_factory=list)
    boundsMin: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    boundsMax: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    formatVersion: int = kCurrentVersion
    dataHash: str = ""
    name: str = "asset"

    # ---- reference-shaped helpers -------------------------------------------------------------
    @property
    def chunkCount(self) -> int:
        """_SplatChunkCount as set by GaussianSplatRenderer.cs:504."""
        return 0 if self.chunkData is None or len(self.chunkData) == 0 else len(self.chunkData) // 64

    @property
    def splatFormat(self) -> int:
        """_SplatFormat = pos | scale<<8 | sh<<16 (GaussianSplatRenderer.cs:502)."""
        return int(self.posFormat) | (int(self.scaleFormat) << 8) | (int(self.shFormat) << 16)

    def ComputeDataHash(self) -> str:
        h = hashlib.sha1()
        h.update(np.array([self.splatCount, self.formatVersion], dtype="<u4").tobytes())
        for blob in (self.chunkData, self.posData, self.otherData, self.colorData, self.shData):
            if blob is not None:
                h.update(memoryview(np.ascontiguousarray(blob)))
        h.update(bytes([int(self.colorFormat)]))
        return h.hexdigest()

    def Validate(self) -> None:
        """Size checks the reference leaves implicit in CreateResourcesForAsset
        (GaussianSplatRenderer.cs:373-405).  Raises ValueError on a malformed asset."""
        n = self.splatCount
        if n <= 0:
            raise ValueError("splatCount must be > 0")
        if self.formatVersion != kCurrentVersion:
            raise ValueError("formatVersion mismatch")
        for nm in ("posData", "otherData", "colorData", "shData"):
            if getattr(self, nm) is None:
                raise ValueError(f"{nm} missing")
        other_stride = GetOtherSizeNoSHIndex(self.scaleFormat) + (2 if self.shFormat > SHFormat.Norm6 else 0)
        need = {
            "posData": CalcPosDataSize(n, self.posFormat),
            "otherData": n * other_stride,
            "colorData": CalcColorDataSize(n, self.colorFormat),
            "shData": GetSHCount(self.shFormat, n) * (_SH_ITEM_SIZE.get(SHFormat(self.shFormat), 96)),
        }
        for nm, sz in need.items():
            if len(getattr(self, nm)) < sz:
                raise ValueError(f"{nm} too small: {len(getattr(self, nm))} < {sz}")
        if self.chunkCount and self.chunkCount < (n + kChunkSize - 1) // kChunkSize:
            raise ValueError("chunkData too small")

    def totalBytes(self) -> int:
        return sum(0 if b is None else len(b) for b in
                   (self.posData, self.otherData, self.colorData, self.shData, self.chunkData))

    # ---- on-disk form: <name>_{chk,pos,oth,col,shs}.bytes + <name>.json -----------------------
    # (file suffixes as written by GaussianSplatAssetCreator.cs:301-305; the .asset YAML becomes JSON)
    def Save(self, folder: str) -> str:
        os.makedirs(folder, exist_ok=True)
        base = os.path.join(folder, self.name)
        files = {"_pos.bytes": self.posData, "_oth.bytes": self.otherData,
                 "_col.bytes": self.colorData, "_shs.bytes": self.shData}
        if self.chunkCount:
            files["_chk.bytes"] = self.chunkData
        for suf, blob in files.items():
            np.ascontiguousarray(blob).tofile(base + suf)
        meta = dict(formatVersion=self.formatVersion, splatCount=self.splatCount,
                    posFormat=int(self.posFormat), scaleFormat=int(self.scaleFormat),
                    shFormat=int(self.shFormat), colorFormat=int(self.colorFormat),
                    boundsMin=list(map(float, self.boundsMin)), boundsMax=list(map(float, self.boundsMax)),
                    dataHash=self.dataHash, hasChunks=bool(self.chunkCount),
                    cameras=[dict(pos=list(c.pos), axisX=list(c.axisX), axisY=list(c.axisY),
                                  axisZ=list(c.axisZ), fov=c.fov) for c in self.cameras])
        with open(base + ".json", "w") as f:
            json.dump(meta, f, indent=1)
        return base + ".json"

    @staticmethod
    def Load(json_path: str) -> "GaussianSplatAsset":
        with open(json_path) as f:
            meta = json.load(f)
        base = json_path[:-5]
        rd = lambda suf: np.fromfile(base + suf, dtype=np.uint8)
        a = GaussianSplatAsset(
            splatCount=meta["splatCount"], posFormat=VectorFormat(meta["posFormat"]),
            scaleFormat=VectorFormat(meta["scaleFormat"]), shFormat=SHFormat(meta["shFormat"]),
            colorFormat=ColorFormat(meta["colorFormat"]), posData=rd("_pos.bytes"), otherData=rd("_oth.bytes"),
            colorData=rd("_col.bytes"), shData=rd("_shs.bytes"),
            chunkData=rd("_chk.bytes") if meta.get("hasChunks") else None,
            cameras=[CameraInfo(tuple(c["pos"]), tuple(c["axisX"]), tuple(c["axisY"]), tuple(c["axisZ"]), c["fov"])
                     for c in meta.get("cameras", [])],
            boundsMin=tuple(meta["boundsMin"]), boundsMax=tuple(meta["boundsMax"]),
            formatVersion=meta["formatVersion"], dataHash=meta.get("dataHash", ""),
            name=os.path.basename(base))
        return a
