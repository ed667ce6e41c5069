"""-m gpu: CSCalcViewData through the C-ABI vs the oracle: every 40-byte record bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
from common import default_camera, small_asset, views_equal
from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer

pytestmark = pytest.mark.gpu

CASES = [("Medium", {}), ("High", {}), ("VeryHigh", {}),
         ("Medium", dict(formatPos=A.VectorFormat.Norm16, formatScale=A.VectorFormat.Norm6, formatSH=A.SHFormat.Float16, formatColor=A.ColorFormat.Float16x4)),
         ("Medium", dict(formatPos=A.VectorFormat.Norm6, formatScale=A.VectorFormat.Float32, formatSH=A.SHFormat.Norm11, formatColor=A.ColorFormat.Float32x4)),
         ("Medium", dict(formatPos=A.VectorFormat.Float32, formatScale=A.VectorFormat.Norm16)),
         ("Medium", dict(formatSH=A.SHFormat.Cluster16k)),                                         # fp16 palette gathered by a u16 index stored after the scale
         ("Low", dict(formatSH=A.SHFormat.Cluster4k))]                                             # Low preset (Norm6 scale: 8-byte `other` stride), 4k palette


@pytest.mark.parametrize("quality,fmt", CASES)
def test_view_data_bit_exact_all_formats(gpu_ctx, quality, fmt):
    a = small_asset(30_011, 7, quality, **fmt)        # 30011: last chunk / last texture tile partially filled
    tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695))
    r = GaussianSplatRenderer(gpu_ctx, a, tr)
    r.OnEnable()
    orc = O.Oracle(a)
    for (shOrder, shOnly, ss, osc, az) in [(3, False, 1.0, 1.0, 20.0), (2, False, 0.5, 1.0, 100.0), (1, True, 1.0, 4.0, 200.0), (0, False, 2.0, 0.25, 300.0)]:
        r.m_SHOrder, r.m_SHOnly, r.m_SplatScale, r.m_OpacityScale = shOrder, shOnly, ss, osc
        cam = default_camera(W=640, H=360, az=az)
        r.CalcViewData(cam)
        got = r.DownloadView()
        want = orc.calc_view(r.FrameParams(cam))
        g, w = got.view(np.uint32).reshape(-1, 10), want.view(np.uint32).reshape(-1, 10)
        assert np.array_equal(g, w), f"{(g != w).any(axis=1).sum()} records differ (shOrder={shOrder})"
        assert (want["pos"][:, 3] > 0).sum() > 1000
    r.OnDisable()


def test_behind_camera_and_nan_axes_records(gpu_ctx):
    from test_oracle import fp32_point_asset
    pts = np.array([[0, 0, 0], [0.3, 0.1, 0], [0, 0, 9.0], [0, 0, 5.0], [1e30, 0, 0]], np.float32)    # on axis, off axis, behind, at the eye, far away
    a = fp32_point_asset(pts)
    oth = a.otherData.view(np.uint32).reshape(-1, 4).copy()
    oth[0, 1:4] = np.float32(1e-25).view(np.uint32)            # 3D covariance underflows -> exactly isotropic 2D covariance -> NaN axes
    a.otherData = oth.view(np.uint8).reshape(-1)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    cam = camera.Camera(position=(0, 0, 5), target=(0, 0, 0), pixelWidth=64, pixelHeight=64)
    r.CalcViewData(cam)
    got = r.DownloadView()
    want = O.Oracle(a).calc_view(r.FrameParams(cam))
    assert views_equal(got, want)
    assert np.isnan(got["axis1"][0]).all() and got["pos"][2, 3] <= 0 and (got["color"][2] == 0).all()
    r.OnDisable()


def test_asset_hot_reload_and_invalid_assets(gpu_ctx):
    import ctypes as C
    from unitygaussiansplatting_amd import _lib
    from unitygaussiansplatting_amd._abi import make_asset_desc
    a = small_asset(5000, 3, "Medium")
    b = small_asset(7000, 4, "VeryHigh")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    cam = default_camera()
    r.CalcViewData(cam)
    assert r.splatCount == 5000
    r.m_Asset = b
    r.Update()                                   # dataHash changed -> resources rebuilt (GaussianSplatRenderer.cs:641-658)
    assert r.splatCount == 7000
    r.CalcViewData(cam)
    assert np.array_equal(r.DownloadView().view(np.uint32), O.Oracle(b).calc_view(r.FrameParams(cam)).view(np.uint32))
    r.OnDisable()
    assert not r.HasValidRenderSetup
    # malformed descriptions are rejected with an error code, not a crash
    keep = []
    d = make_asset_desc(a, keep)
    h = C.c_void_p()
    d.sh_size -= 64
    assert _lib.lib().gs_asset_create(gpu_ctx._h, C.byref(d), C.byref(h)) == -5
    d = make_asset_desc(a, keep)
    d.color_format = 3                           # BC7 is 1 byte per texel: a Norm8x4 blob is larger than needed and is accepted as BC7 data
    assert _lib.lib().gs_asset_create(gpu_ctx._h, C.byref(d), C.byref(h)) == 0
    assert _lib.lib().gs_asset_destroy(h) == 0
    d = make_asset_desc(a, keep)
    d.color_format = 4
    assert _lib.lib().gs_asset_create(gpu_ctx._h, C.byref(d), C.byref(h)) == -1
    d = make_asset_desc(a, keep)
    d.pos_format = 9
    assert _lib.lib().gs_asset_create(gpu_ctx._h, C.byref(d), C.byref(h)) == -1


def test_view_buffer_every_frame_mode_matches_on_demand(gpu_ctx):
    """gs_renderer_set_view_buffer_mode(1) runs the reference's full CSCalcViewData every frame; the default evaluates
    colours only for splats that reach the screen (and stops early for splats that cannot) and materialises m_GpuView in
    gs_renderer_download_view.  Both must give the same view buffer, the same frame and the same pair / visible counts."""
    from unitygaussiansplatting_amd.renderer import RenderTarget
    a = small_asset(50_000, 9, "Medium")
    cam = default_camera(W=400, H=240, az=70.0, radius=4.0)
    res = []
    for every in (False, True):
        r = GaussianSplatRenderer(gpu_ctx, a)
        r.OnEnable()
        r.SetViewBufferMode(every)
        rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        st = r.FrameStats()
        img = rt.Download().copy()
        view = r.DownloadView().copy()
        rt.Clear(); r.Draw(cam, rt)                      # the on-demand download re-ran calc_view: the draw after it must not change
        assert np.array_equal(rt.Download(), img)
        res.append((img, view, st.tile_pairs, st.visible_splats))
        r.OnDisable(); rt.Dispose()
    assert np.array_equal(res[0][0], res[1][0]) and views_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    assert 0 < res[0][3] < a.splatCount


@pytest.mark.parametrize("shfmt", ["Norm11", "Float16", "Norm6"])
def test_chunkless_lossy_sh_asset_on_the_per_frame_path(gpu_ctx, shfmt):
    """An asset with a lossy SH format but NO chunk blob (gs_asset_create accepts it: _SplatChunkCount = 0, so nothing is chunk-lerped,
    SplatUtilities.compute / GaussianSplatting.hlsl:521-527): the per-frame calc_view launch (which does not zero-fill its per-splat
    record) must see shLerp = false like the full kernel and the oracle do -- records, visibility and the frame itself."""
    from test_gpu_configs import check_raster_records
    from common import RT_TOL, rt_err
    from unitygaussiansplatting_amd.renderer import RenderTarget
    a0 = small_asset(20_000, 9, "VeryHigh", formatSH=getattr(A.SHFormat, shfmt))
    import copy
    a = copy.copy(a0)
    a.chunkData = None
    cam = default_camera(W=400, H=240, az=70.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    st = r.FrameStats()
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    orc.calc_view(P)
    assert check_raster_records(r, orc, P) == st.visible_splats > 1000
    assert rt_err(rt.Download(), orc.draw(P, 0)) <= RT_TOL and st.tile_pairs == orc.pairs(P, st)
    assert np.array_equal(r.DownloadView().view(np.uint32), orc.view.view(np.uint32))
    r.OnDisable(); rt.Dispose()
