"""BC7 colour (ColorFormat.BC7 = RGBA_BC7_UNorm, GaussianSplatAsset.cs:56,169; VeryLow preset, GaussianSplatAssetCreator.cs:198).

Three decoders exist here -- bc7.py (plain Python reference), oracle/gs_oracle.cpp (sequential bit reader, all 16 texels) and
gs_device_math.h (the kernels' single-texel decoder, also compiled for the host by this test) -- and one that is NOT ours:
Pillow's.  CPU tests: the partition / anchor tables re-extracted from Pillow equal bc7.py's; all four decoders agree on
random blocks of every mode (incl. the reserved one); the mode-6 encoder round-trips.  GPU test: a VeryLow asset
(BC7 + Cluster4k + Norm6 scale) renders with bit-exact view records and the usual frame tolerance."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from common import RT_TOL, default_camera, rt_err, small_asset, views_equal
from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import bc7, camera, creator, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pil = pytest.importorskip("PIL")


def random_blocks(mode, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        v = int.from_bytes(rng.bytes(16), "little")
        if mode < 8:
            v = (v & ~((1 << (mode + 1)) - 1)) | (1 << mode)
        else:
            v &= ~0xff                                    # reserved: low byte zero -> decodes to zero
        out.append(v.to_bytes(16, "little"))
    return out


def test_tables_match_an_independent_decoder():
    import bc7_tables_from_pil as T
    t = T.extract()
    for k, v in t.items():
        assert list(getattr(bc7, k)) == list(v), k


@pytest.fixture(scope="module")
def host_decoder(tmp_path_factory):
    """gs_device_math.h's DecodeBC7Texel compiled for the host (the same source the kernels compile)."""
    d = tmp_path_factory.mktemp("bc7host")
    src = d / "h.cpp"
    src.write_text('#include "%s/unitygaussiansplatting_amd/csrc/gs_device_math.h"\n'
                   'extern "C" void decode_block(const uint8_t* b, uint8_t* out) { for (uint32_t t = 0; t < 16; ++t) { uint32_t v = gsm::DecodeBC7Texel(b, t); memcpy(out + t * 4, &v, 4); } }\n' % ROOT)
    so = d / "libh.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", str(so), str(src)])
    return C.CDLL(str(so))


@pytest.mark.parametrize("mode", range(9))
def test_all_decoders_agree_with_pillow(mode, host_decoder):
    import bc7_tables_from_pil as T
    blocks = random_blocks(mode, 300, 100 + mode)
    # the reserved mode (low byte zero) decodes to zero in ALL channels (Khronos Data Format Spec / D3D11 functional spec);
    # Pillow returns opaque black for it, so for mode 8 the expectation is the specification, not Pillow
    ref = T.pil_decode(blocks) if mode < 8 else [np.zeros((16, 4), np.uint8)] * len(blocks)
    L = O.lib()
    for b, want in zip(blocks, ref):
        assert np.array_equal(bc7.decode_block(b), want)
        buf = np.frombuffer(b + bytes(16), np.uint8).copy()
        got = np.zeros((16, 4), np.uint8)
        L.gso_bc7_decode_block(buf.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, want), "oracle decoder"
        got2 = np.zeros((16, 4), np.uint8)
        host_decoder.decode_block(buf.ctypes.data_as(C.c_void_p), got2.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got2, want), "kernel decoder (host build)"


def test_mode6_encoder_round_trip_and_asset_decode():
    rng = np.random.default_rng(3)
    # smooth blocks are reproduced to 7.5-bit endpoint precision; a constant block exactly to 8 bits
    base = rng.random((2, 4, 1, 1, 4), dtype=np.float32) * np.float32(0.55)       # + 0.4 of ramps stays below 1: the block is one line
    ramp = np.linspace(0, 1, 4, dtype=np.float32)[None, None, None, :, None] * 0.2
    img = np.clip(base + ramp + ramp.transpose(0, 1, 3, 2, 4), 0, 1) * np.ones((2, 4, 4, 4, 4), np.float32)
    tex = img.transpose(0, 2, 1, 3, 4).reshape(8, 16, 4)
    dec = bc7.decode_texture(bc7.encode_texture_mode6(tex), 16, 8).astype(np.float32) / 255.0
    assert np.abs(dec - tex).max() < 0.03
    const = np.full((4, 4, 4), 0.4, np.float32)
    assert np.abs(bc7.decode_texture(bc7.encode_texture_mode6(const), 4, 4).astype(np.float32) / 255.0 - 0.4).max() <= 1.0 / 255.0
    # a VeryLow asset: the oracle's per-splat colour equals the reference decode of the blob at the splat's texel
    raw = scenes.make_splats(6000, 9, 2.0)
    a = creator.CreateAssetFromSplats(raw, "VeryLow", name="verylow")
    assert a.colorFormat == A.ColorFormat.BC7 and a.shFormat == A.SHFormat.Cluster4k
    w, h = A.CalcTextureSize(a.splatCount)
    assert len(a.colorData) == w * h
    texels = bc7.decode_texture(a.colorData, w, h).reshape(-1, 4)
    idx = creator.SplatIndexToTextureIndex(np.arange(a.splatCount, dtype=np.uint32))
    dec = O.Oracle(a).decode_all()
    chunk = np.frombuffer(a.chunkData, A.CHUNK_DTYPE)
    mn = np.stack([chunk[k].astype(np.uint32) & 0xffff for k in ("colR", "colG", "colB")], 1).astype(np.uint16).view(np.float16).astype(np.float32)
    mx = np.stack([chunk[k].astype(np.uint32) >> 16 for k in ("colR", "colG", "colB")], 1).astype(np.uint16).view(np.float16).astype(np.float32)
    ci = np.arange(a.splatCount) // 256
    t = texels[idx, :3].astype(np.float32) * np.float32(1.0 / 255.0)
    want = np.float32(t) * (mx[ci] - mn[ci]) + mn[ci]
    assert np.allclose(dec[:, 11:14], want, atol=2e-6)


@pytest.mark.gpu
def test_verylow_asset_renders_like_the_oracle(gpu_ctx):
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget
    raw = scenes.make_splats(30_000, 17, 3.0)
    a = creator.CreateAssetFromSplats(raw, "VeryLow", name="verylow")
    cam = default_camera(W=480, H=300, az=15.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    st = r.FrameStats()
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    assert np.array_equal(r.DownloadOrder(), orc.order)
    assert views_equal(r.DownloadView(), orc.calc_view(P))
    ref = orc.draw(P, 0)
    assert rt_err(rt.Download(), ref) <= RT_TOL and st.tile_pairs == orc.pairs(P, st)
    r.OnDisable()
