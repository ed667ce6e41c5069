"""Asset data contract: size calculators, Morton texel addressing, codec known-answer tests (CPU only).

The bytes are produced by unitygaussiansplatting_amd.creator following the reference importer's rules
(GaussianSplatAssetCreator.cs:705-758, 520-658) and decoded by the oracle following the shader
(GaussianSplatting.hlsl:261-300, 428-608): encode -> decode must land within one quantisation step."""
import numpy as np
import pytest

import oracle_lib as O
from common import small_asset
from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import creator, scenes


def test_size_calculators_match_reference_table():
    # bicycle Medium = 24.5 / 49.1 / 24.6 / 196.2 / 1.5 MB = 296.0 MB
    n = 6_131_954
    assert A.CalcPosDataSize(n, A.VectorFormat.Norm11) == n * 4
    assert A.CalcOtherDataSize(n, A.VectorFormat.Norm11) == n * 8
    assert A.CalcTextureSize(n) == (2048, 3008)
    assert A.CalcColorDataSize(n, A.ColorFormat.Norm8x4) == 2048 * 3008 * 4
    assert A.CalcSHDataSize(n, A.SHFormat.Norm6) == n * 32
    assert A.CalcChunkDataSize(n) == 23_953 * 64
    total = (A.CalcPosDataSize(n, 2) + A.CalcOtherDataSize(n, 2) + A.CalcColorDataSize(n, 2) + A.CalcSHDataSize(n, 3) + A.CalcChunkDataSize(n))
    assert abs(total / 1e6 - 296.0) < 0.05
    assert A.CalcSHDataSize(1000, A.SHFormat.Cluster4k) == 4096 * 96 + 1000 * 2
    assert A.GetVectorSize(A.VectorFormat.Norm16) == 6 and A.GetColorSize(A.ColorFormat.BC7) == 1
    with pytest.raises(ValueError):
        A.GetVectorSize(7)


def test_morton_texel_addressing_is_a_bijection():
    # SplatIndexToPixelIndex (hlsl:183-194) o creator's SplatIndexToTextureIndex (:863-871) over 2^20 indices
    idx = np.arange(1 << 20, dtype=np.uint32)
    lin = creator.SplatIndexToTextureIndex(idx)
    assert len(np.unique(lin)) == len(idx)
    xy = np.zeros(2, np.uint32)
    import ctypes as C
    for i in (0, 1, 2, 3, 255, 256, 257, 65535, 65536, 123457, (1 << 20) - 1):
        O.lib().gso_pixel_index(C.c_uint32(i), xy.ctypes.data_as(C.c_void_p))
        assert int(xy[1]) * 2048 + int(xy[0]) == int(lin[i])
    # inside a 16x16 tile the low 8 bits are Morton order: index 0..3 -> (0,0),(1,0),(0,1),(1,1)
    x, y = creator.DecodeMorton2D_16x16(np.arange(4))
    assert list(zip(x.tolist(), y.tolist())) == [(0, 0), (1, 0), (0, 1), (1, 1)]


def test_f16_conversion_exhaustive_against_numpy():
    import ctypes as C
    lib = O.lib()
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    got = np.array([lib.gso_f16tof32(C.c_uint16(int(v))) for v in h[::7]], np.float32)
    r7 = ref[::7]
    nan = np.isnan(r7)                 # NaN payloads do not survive the float -> Python double -> float32 trip
    assert np.array_equal(got.view(np.uint32)[~nan], r7.view(np.uint32)[~nan]) and np.isnan(got[nan]).all()
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf], np.float32)])
    want = f.astype(np.float16).view(np.uint16)
    got = np.array([lib.gso_f32tof16(C.c_float(float(v))) for v in f], np.uint16)
    assert np.array_equal(got, want)


def _linear(n, seed):
    raw = scenes.make_splats(n, seed, 3.0)
    return creator.LinearizeData(raw)


@pytest.mark.parametrize("quality", ["Medium", "High", "VeryHigh"])
def test_codec_roundtrip_within_one_quantisation_step(quality):
    n = 5000
    raw = scenes.make_splats(n, 3, 3.0)
    lin = creator.LinearizeData(raw)
    lin_sorted, _, _ = creator.ReorderMorton(lin)
    a = creator.CreateAssetFromSplats(raw, quality)
    dec = O.Oracle(a).decode_all()            # pos3 rot4 scale3 opacity col3 sh45
    fp, fs, fc, fsh = creator.QUALITY[quality]
    ext = lin_sorted.pos.max(0) - lin_sorted.pos.min(0)
    pos_tol = {A.VectorFormat.Float32: 1e-6, A.VectorFormat.Norm16: 2e-5, A.VectorFormat.Norm11: 1.2e-3}[fp] * float(ext.max())
    assert np.abs(dec[:, 0:3] - lin_sorted.pos).max() <= pos_tol
    # rotation: decoded quaternion equals +-q (10.10.10.2 smallest-three)
    q = raw.rot / np.linalg.norm(raw.rot, axis=1, keepdims=True)
    q = q[:, [1, 2, 3, 0]]
    # apply the same Morton order
    order = np.argsort(creator.MortonEncode3(((lin.pos - lin.pos.min(0)) / (lin.pos.max(0) - lin.pos.min(0)) * np.float32((1 << 21) - 1)).astype(np.float32).astype(np.uint32)), kind="stable")
    q = q[order]
    dq = dec[:, 3:7]
    err = np.minimum(np.abs(dq - q).max(1), np.abs(dq + q).max(1))
    assert err.max() < 4e-3        # 10-bit step on a sqrt(2) range is 1.4e-3 per component; w is derived
    assert np.abs(np.linalg.norm(dq, axis=1) - 1).max() < 2e-3
    # scale: relative error (scale^(1/8) is quantised, then ^8)
    rel = np.abs(dec[:, 7:10] - lin_sorted.scale) / lin_sorted.scale
    assert rel.max() < {A.VectorFormat.Float32: 1e-6, A.VectorFormat.Norm16: 6e-3, A.VectorFormat.Norm11: 3e-2}[fs]   # chunk bounds are stored as fp16 (2^-11 rel.) and the value is raised to the 8th power
    # opacity and colour
    col_tol = {A.ColorFormat.Float32x4: 1e-6, A.ColorFormat.Float16x4: 4e-3, A.ColorFormat.Norm8x4: 3e-2}[fc]
    # opacity is stored as SquareCentered01(o) (zero slope at 0.5), so a step q in storage is ~sqrt(q/2) near o = 0.5
    op_tol = {A.ColorFormat.Float32x4: 1e-6, A.ColorFormat.Float16x4: 2.5e-2, A.ColorFormat.Norm8x4: 7e-2}[fc]
    assert np.abs(dec[:, 10] - lin_sorted.opacity).max() < op_tol
    span = lin_sorted.dc0.max() - lin_sorted.dc0.min()
    assert np.abs(dec[:, 11:14] - lin_sorted.dc0).max() < col_tol * float(span) + 1e-6
    sh = lin_sorted.sh.reshape(n, 45)
    sh_tol = {A.SHFormat.Float32: 1e-7, A.SHFormat.Norm11: 2e-3, A.SHFormat.Norm6: 4e-2}[fsh] * float(sh.max() - sh.min())
    assert np.abs(dec[:, 14:59] - sh).max() <= sh_tol + 1e-7


@pytest.mark.parametrize("fp,fs", [(A.VectorFormat.Norm16, A.VectorFormat.Norm6), (A.VectorFormat.Norm6, A.VectorFormat.Norm16)])
def test_unaligned_strides_decode(fp, fs):
    # 6-byte and 2-byte records put every other splat on a 2-byte-aligned address (hlsl:325-392)
    raw = scenes.make_splats(1031, 9, 2.0)
    a = creator.CreateAssetFromSplats(raw, "Medium", formatPos=fp, formatScale=fs)
    dec = O.Oracle(a).decode_all()
    lin, _, _ = creator.ReorderMorton(creator.LinearizeData(raw))
    tol = {A.VectorFormat.Norm16: 2e-5, A.VectorFormat.Norm6: 4e-2}[fp] * float((lin.pos.max(0) - lin.pos.min(0)).max())
    assert np.abs(dec[:, 0:3] - lin.pos).max() <= tol
    assert np.isfinite(dec).all()


def test_exhaustive_field_decode_norm6_norm11():
    # every 16-bit Norm6 code and a sweep of Norm11 codes decode to field/(2^bits-1) (hlsl:261-291) within 1 ulp
    import ctypes as C
    n = 65536
    pos = np.arange(n, dtype=np.uint16)
    a = A.GaussianSplatAsset(splatCount=n, posFormat=A.VectorFormat.Norm6, scaleFormat=A.VectorFormat.Float32,
                             shFormat=A.SHFormat.Float32, colorFormat=A.ColorFormat.Float32x4,
                             posData=pos.view(np.uint8).copy(), otherData=np.zeros(n * 16, np.uint8),
                             colorData=np.zeros(A.CalcColorDataSize(n, 0), np.uint8), shData=np.zeros(n * 192, np.uint8))
    dec = O.Oracle(a).decode_all()
    want = np.stack([(pos & 63) / 63.0, ((pos >> 6) & 31) / 31.0, ((pos >> 11) & 31) / 31.0], 1)
    assert np.abs(dec[:, 0:3] - want).max() < 1.2e-7
    codes = (np.arange(n, dtype=np.uint64) * 65521 % (1 << 32)).astype(np.uint32)
    a.posFormat = A.VectorFormat.Norm11
    a.posData = codes.view(np.uint8).copy()
    dec = O.Oracle(a).decode_all()
    want = np.stack([(codes & 2047) / 2047.0, ((codes >> 11) & 1023) / 1023.0, ((codes >> 21) & 2047) / 2047.0], 1)
    assert np.abs(dec[:, 0:3] - want).max() < 1.2e-7


def test_ply_roundtrip_and_asset_io(tmp_path):
    raw = scenes.make_splats(777, 4, 2.0)
    p = str(tmp_path / "scene.ply")
    creator.WritePLY(p, raw)
    back = creator.ReadPLY(p)
    for f in ("pos", "dc0", "sh", "opacity", "scale", "rot"):
        assert np.array_equal(getattr(raw, f), getattr(back, f)), f
    a = creator.CreateAsset(p, "Medium", name="io")
    j = a.Save(str(tmp_path))
    b = A.GaussianSplatAsset.Load(j)
    assert b.dataHash == a.dataHash == b.ComputeDataHash()
    for nm in ("posData", "otherData", "colorData", "shData", "chunkData"):
        assert np.array_equal(getattr(a, nm), getattr(b, nm))
    with pytest.raises(IOError):
        bad = tmp_path / "bad.ply"
        bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
        creator.ReadPLY(str(bad))


def test_asset_validate_rejects_truncated_blobs():
    a = small_asset(2000)
    b = A.GaussianSplatAsset(**{k: getattr(a, k) for k in ("splatCount", "posFormat", "scaleFormat", "shFormat", "colorFormat",
                                                          "posData", "otherData", "colorData", "shData", "chunkData")})
    b.Validate()
    b.shData = a.shData[:-8]
    with pytest.raises(ValueError):
        b.Validate()


def test_clustered_sh_layout_and_decode():
    """Cluster* formats: K x 96-byte fp16 palette (SHTableItemFloat16), u16 index appended to every `other` record
    (CreateOtherDataJob, GaussianSplatAssetCreator.cs:776-805); the decoder returns exactly the palette entry."""
    from unitygaussiansplatting_amd import creator, scenes
    raw = scenes.make_splats(6000, 3, 2.0)
    a = creator.CreateAssetFromSplats(raw, "Medium", formatSH=A.SHFormat.Cluster4k)
    n, k = a.splatCount, 4096
    assert len(a.shData) == k * 96 and len(a.otherData) >= n * (4 + 4 + 2)
    oth = np.frombuffer(a.otherData.tobytes()[:n * 10], np.uint8).reshape(n, 10)
    idx = oth[:, 8:10].copy().view("<u2").reshape(n).astype(np.int64)
    assert idx.max() < k and len(np.unique(idx)) > k // 2
    table = np.frombuffer(a.shData.tobytes(), "<f2").reshape(k, 48)[:, :45].astype(np.float32)
    dec = O.Oracle(a).decode_all()                       # pos3 rot4 scale3 opacity1 col3 sh45
    assert dec.shape[1] == 59
    assert np.array_equal(dec[:, 14:59], table[idx])
    with pytest.raises(ValueError):
        creator.CreateAssetFromSplats(scenes.make_splats(3000, 3, 2.0), "Medium", formatSH=A.SHFormat.Cluster4k)
