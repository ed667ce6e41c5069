"""-m gpu: one asserted oracle-parity test per BASELINE.json configuration that fits one GPU.

  C1  synthetic 100 k / 640x360 / Medium, through the real load path: PLY file -> gs_ply_open -> gs_import_encode ->
      gs_asset_create -> frame
  C2  bicycle-sized 6,131,954 / 1200x797 / Medium: the whole frame against the oracle (keys, order, view, per-frame
      raster records, P, visible, RGBA16F target)
  C3  garden-sized 5,834,784 / 1920x1080 / VeryHigh fp32 at FULL size
  C5  the C2 asset from 8 cameras (azimuth k * 45 deg) at 1920x1080, every view the whole frame
  C4  synthetic 50 M / 3840x2160 / Medium: keys / order / view bit-exact, P and visible equal, the WHOLE target -- the only
      configuration with > 2^32-byte blobs, 6,104 sort partitions and 16,200 tiles
  C2 pinned to 32x32 / 16x16 tiles, and C2d (overdraw-heavy) under the automatic policy: two draws, the second at 32x32

Bars: keys, order, 40-byte view records, raster records, P, visible: bit-exact / equal; RGBA16F
target in exact mode: |d| <= 2^-9 * max(1, |c|) per channel (fp16 ulps grow with the value; splat colours may exceed 1).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from common import RT_TOL, rt_err, views_equal
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget

pytestmark = pytest.mark.gpu

TOL = RT_TOL


def rt_close(img, ref, rare=0):
    return rt_err(img, ref, rare=rare), float((img == ref).all(axis=-1).mean())


def cam_of(cfg, az=0.0):
    return camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, az), pixelWidth=cfg.width, pixelHeight=cfg.height,
                         fieldOfView=cfg.fov_y)


def check_raster_records(r, orc, P):
    """rec / rect / visibility of the PER-FRAME calc_view launch (FULL = false: early cull, chunk cull, colour only for visible
    splats) against the oracle's prepare() -- the view-parity tests only ever read the on-demand full kernel."""
    recs, rects, vis = r.DownloadRasterRecords()
    orecs, orects, ovis = orc.raster_records(P)
    assert np.array_equal(vis, ovis), "visibility bits differ"
    assert np.array_equal(rects, orects), "tile rectangles differ"
    m = (np.unpackbits(ovis.view(np.uint8), bitorder="little")[:orc.n]).astype(bool)
    assert np.array_equal(recs[m], orecs[m]), "blend records of visible splats differ"
    return int(m.sum())


def full_frame_vs_oracle(gpu_ctx, a, cfg, azimuths=(0.0,), window=None, check_rt=True, rare=0, tile=None, redraw_expect_tile=None):
    """tile = (w, h): pin the compositor tile (SetTileShape); redraw_expect_tile = (w, h): draw every view a SECOND time under the
    automatic policy and require that draw to have used that tile (adapt_tile_shape reads the first draw's report) -- both draws are
    compared with the oracle."""
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    if tile is not None:
        r.SetTileShape(*tile)
    orc = O.Oracle(a)
    rt = RenderTarget(gpu_ctx, cfg.width, cfg.height)
    out = {}
    for az in azimuths:
        cam = cam_of(cfg, az)
        r.SortPoints(cam)
        r.CalcViewData(cam)
        rt.Clear(); r.Draw(cam, rt)
        st = r.FrameStats()
        P = r.FrameParams(cam)
        orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
        assert np.array_equal(r.DownloadDistances(), orc.keys), "sorted keys differ"
        assert np.array_equal(r.DownloadOrder(), orc.order), "order differs"
        orc.calc_view(P)
        nvis = check_raster_records(r, orc, P)                  # before DownloadView re-runs the full kernel
        img = rt.Download()
        assert views_equal(r.DownloadView(), orc.view), "view records differ"
        if check_rt:
            ref = orc.draw(P, 0, window=window)
            if window is not None:
                x0, y0, x1, y1 = window
                img, ref = img[y0:y1 + 1, x0:x1 + 1], ref[y0:y1 + 1, x0:x1 + 1]
            mx, eq = rt_close(img, ref, rare)
            assert mx <= TOL, f"target max rel-abs {mx} > 2^-9 ({eq:.4f} of the pixels bit-equal)"
            assert O.f16_to_f32(ref)[..., 3].mean() > 0.005           # the crop is not empty
            out["rt_max"], out["rt_equal"] = mx, eq
        else:
            orc.draw(P, 0, window=(0, 0, -1, -1))               # counts only
        assert st.tile_pairs == orc.pairs(P, st) and st.visible_splats == orc.visible == nvis
        if tile is not None:
            assert (st.tile_w, st.tile_h) == tuple(tile)
        out["pairs"], out["visible"] = int(st.tile_pairs), int(st.visible_splats)
        if redraw_expect_tile is not None:
            # the same view again: the policy has seen the first draw's pairs-per-visible-splat ratio by now
            r.CalcViewData(cam)
            rt.Clear(); r.Draw(cam, rt)
            st2 = r.FrameStats()
            assert (st2.tile_w, st2.tile_h) == tuple(redraw_expect_tile), f"second draw used {st2.tile_w}x{st2.tile_h}"
            assert st2.tile_pairs == orc.pairs(P, st2) and st2.visible_splats == nvis
            check_raster_records(r, orc, P)
            img2 = rt.Download()
            if check_rt:
                mx2, eq2 = rt_close(img2, ref, rare)
                assert mx2 <= TOL, f"second draw: target max rel-abs {mx2} > 2^-9 ({eq2:.4f} bit-equal)"
                # blend order per pixel does not depend on the tile: the two draws are the same frame, bit for bit
                assert np.array_equal(img2, img), "the frame changed with the tile shape"
            out["pairs2"] = int(st2.tile_pairs)
    r.OnDisable()
    rt.Dispose()
    return out


def test_c1_ply_to_pixels(gpu_ctx, tmp_path):
    cfg = scenes.CONFIGS["C1"]
    raw = scenes.make_config_splats(cfg)
    path = str(tmp_path / "c1.ply")
    creator.WritePLY(path, raw)
    a = creator.CreateAssetFromSplatsNative(creator.ReadPLYNative(path), cfg.quality, name="C1")     # gs_ply_open + gs_import_encode
    assert a.splatCount == 100_000
    out = full_frame_vs_oracle(gpu_ctx, a, cfg, azimuths=(0.0, 70.0))
    assert out["visible"] > 10_000


@pytest.fixture(scope="module")
def c2_asset():
    cfg = scenes.CONFIGS["C2"]
    return creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2")


def test_c2_full_frame(gpu_ctx, c2_asset):
    cfg = scenes.CONFIGS["C2"]
    assert c2_asset.splatCount == 6_131_954
    full_frame_vs_oracle(gpu_ctx, c2_asset, cfg, azimuths=(0.0, 33.0))


@pytest.mark.parametrize("tile", [(32, 32), (16, 16)])
def test_c2_full_frame_pinned_tile(gpu_ctx, c2_asset, tile):
    """The C2 frame with the compositor tile pinned to the two shapes the automatic policy does not pick for it."""
    cfg = scenes.CONFIGS["C2"]
    full_frame_vs_oracle(gpu_ctx, c2_asset, cfg, azimuths=(0.0,), tile=tile)


def test_c2d_full_frame(gpu_ctx):
    """C2d (C2 with bicycle-like overdraw) at 1200x797 under the AUTOMATIC tile policy: the first draw runs 32x16, its report (17 tiles
    of 16x16 per visible splat) makes adapt_tile_shape pick 32x32 for the second -- the regime the policy exists for, whole frame."""
    cfg = scenes.CONFIGS["C2d"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2d")
    out = full_frame_vs_oracle(gpu_ctx, a, cfg, azimuths=(0.0,), redraw_expect_tile=(32, 32))
    assert out["pairs2"] < out["pairs"]                              # larger tiles, fewer pairs


def test_c5_views(gpu_ctx, c2_asset):
    """BASELINE.json config 5: the C2 asset from the 8 cameras of the orbit (azimuth k * 45 deg) at 1920x1080 -- here all on one GPU,
    every view the whole frame against the oracle (keys, order, view + raster records, P, visible, target)."""
    cfg = scenes.CONFIGS["C5"]
    assert (cfg.width, cfg.height) == (1920, 1080) and cfg.n == c2_asset.splatCount
    out = full_frame_vs_oracle(gpu_ctx, c2_asset, cfg, azimuths=tuple(45.0 * k for k in range(8)))
    assert out["visible"] > 1_000_000


def test_c3_full_size(gpu_ctx):
    cfg = scenes.CONFIGS["C3"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C3")
    assert a.splatCount == 5_834_784 and (a.chunkData is None or len(a.chunkData) == 0)
    # the one frame where a pixel above 2^-9 was measured (1.125 * 2^-9, one pixel of 2 M): the outlier allowance is opted into HERE only
    full_frame_vs_oracle(gpu_ctx, a, cfg, rare=2 + 2 * (cfg.width * cfg.height) // 1_000_000)


@pytest.mark.skipif(os.environ.get("GSPLAT_SKIP_C4") == "1", reason="GSPLAT_SKIP_C4=1")
def test_c4_50m(gpu_ctx):
    cfg = scenes.CONFIGS["C4"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C4")
    assert a.splatCount == 50_000_000
    assert len(a.shData) > 2 ** 30 and len(a.otherData) < 2 ** 32 and a.totalBytes() > 2.4e9
    # the WHOLE 3840x2160 target (round 5; rounds 1-4 compared a centred 512x512 window): first / last tile rows and columns and the
    # tail of the 16,200-tile schedule included.  GSPLAT_C4_WINDOW=1 restores the window (a slow host: the oracle composites 8.3 Mpx).
    window = None
    if os.environ.get("GSPLAT_C4_WINDOW") == "1":
        cx, cy = cfg.width // 2, cfg.height // 2
        window = (cx - 256, cy - 256, cx + 255, cy + 255)
    out = full_frame_vs_oracle(gpu_ctx, a, cfg, window=window, rare=0 if window else 2 + 2 * (cfg.width * cfg.height) // 1_000_000)
    assert out["pairs"] > 10_000_000
