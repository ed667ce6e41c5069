"""Scene-depth test of the splat draw and the debug point render modes.

Scene depth: the reference binds the camera's depth buffer next to the splat RT (GaussianSplatRenderer.cs:195) and draws with
ZTest LEqual, ZWrite Off (RenderGaussianSplats.shader:10); every vertex of a splat quad carries the centre's depth (:56-60), so
a fragment survives iff the splat's view depth <= the opaque scene's view depth at the pixel.  CPU: a two-plane known answer
for the oracle; GPU: gs_target_set_scene_depth against the oracle.

Debug points: GaussianDebugRenderPoints.shader through gs_renderer_set_render_mode, against the oracle's sequential
restatement (bit-exact: no transcendental, no blending)."""
import numpy as np
import pytest

import oracle_lib as O
from common import RT_TOL, default_camera, rt_err, small_asset
from test_oracle import fp32_point_asset
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderMode, RenderTarget


def test_oracle_depth_test_two_planes():
    """Two opaque-ish splats on the optical axis at view depths 4 and 6; a scene plane at depth 5 covering the left half of the
    screen hides the far splat there and nothing else; a plane at 3 hides both; depth exactly equal passes (LEqual)."""
    a = fp32_point_asset([[0.0, 0.0, 2.0], [0.0, 0.0, 0.0]])              # camera at z = 6 looking down -z: depths 4 and 6
    cam = camera.Camera(position=(0.0, 0.0, 6.0), pixelWidth=64, pixelHeight=48)
    tr = camera.Transform()
    P = camera.frame_params(cam, tr)
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
    orc.calc_view(P)
    w = orc.view["pos"][:, 3]
    assert np.allclose(w, [4.0, 6.0])
    free = O.f16_to_f32(orc.draw(P, 0))
    near_only = O.f16_to_f32(orc.draw(P, 0, scene_depth=np.full((48, 64), 5.0, np.float32)))
    none = O.f16_to_f32(orc.draw(P, 0, scene_depth=np.full((48, 64), 3.0, np.float32)))
    equal = O.f16_to_f32(orc.draw(P, 0, scene_depth=np.full((48, 64), np.float32(w[1]), np.float32)))
    assert none.max() == 0.0 and np.array_equal(equal, free)
    assert free[24, 32, 3] > near_only[24, 32, 3] > 0.0                    # the far splat adds coverage behind the near one
    half = np.full((48, 64), np.inf, np.float32); half[:, :32] = 5.0
    mixed = O.f16_to_f32(orc.draw(P, 0, scene_depth=half))
    assert np.array_equal(mixed[:, :32], near_only[:, :32]) and np.array_equal(mixed[:, 32:], free[:, 32:])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_scene_depth_matches_oracle(gpu_ctx, mode):
    a = small_asset(40_000, 5, "Medium")
    cam = default_camera(W=400, H=260, az=20.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    r.blendMode = mode
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    # an "opaque scene": a tilted plane through the middle of the splat cloud + a box of nearer depth in one corner
    yy, xx = np.mgrid[0:cam.pixelHeight, 0:cam.pixelWidth].astype(np.float32)
    depth = (5.0 + 0.004 * (xx - 200.0) + 0.002 * (yy - 130.0)).astype(np.float32)
    depth[:60, :80] = 1.0
    depth[200:, 300:] = np.inf
    r.SortPoints(cam); r.CalcViewData(cam)
    rt.SetSceneDepth(depth)
    rt.Clear(); r.Draw(cam, rt)
    st = r.FrameStats()
    img = rt.Download()
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    orc.calc_view(P)
    ref = orc.draw(P, mode, scene_depth=depth)
    free = orc.draw(P, mode)
    assert rt_err(img, ref) <= (RT_TOL if mode == 0 else 4e-3)
    assert st.tile_pairs == orc.pairs(P, st)                                  # binning does not depend on the depth attachment
    assert O.f16_to_f32(img)[:60, :80].max() == 0.0                         # everything is behind the near box
    assert rt_err(img, free) > 0.05                                         # ... and the plane really hides something
    rt.SetSceneDepth(None)                                                  # detached again: the unoccluded frame
    rt.Clear(); r.Draw(cam, rt)
    assert rt_err(rt.Download(), free) <= (RT_TOL if mode == 0 else 4e-3)
    r.OnDisable()


@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["Medium", "VeryHigh"])
def test_gpu_debug_point_modes(gpu_ctx, quality):
    a = small_asset(20_000, 6, quality)
    cam = default_camera(W=320, H=200, az=-35.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    orc = O.Oracle(a)
    P = r.FrameParams(cam)
    depth = np.full((cam.pixelHeight, cam.pixelWidth), np.inf, np.float32)
    depth[:, :100] = 5.5
    for rm, size, sd in ((RenderMode.DebugPoints, 3.0, None), (RenderMode.DebugPointIndices, 3.0, None), (RenderMode.DebugPoints, 1.0, None),
                         (RenderMode.DebugPoints, 7.5, depth)):
        r.m_RenderMode, r.m_PointDisplaySize = rm, size
        rt.SetSceneDepth(sd)
        rt.Clear(); r.Draw(cam, rt)                                         # no sort, no calc_view needed
        img = rt.Download()
        ref = orc.draw_debug_points(P, rm == RenderMode.DebugPointIndices, size, scene_depth=sd)
        assert np.array_equal(img, ref), (rm, size)
        assert (img[..., 3] == 0x3c00).mean() > 0.02
    # back to splats on the same target: the normal path is unaffected
    r.m_RenderMode = RenderMode.Splats
    rt.SetSceneDepth(None)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix)); orc.calc_view(P)
    assert rt_err(rt.Download(), orc.draw(P, 0)) <= RT_TOL
    r.OnDisable()


def _hull_coverage(corners_px, W, H):
    """float64: which pixel centres lie inside the convex hull of the projected corners, and how far from its boundary."""
    from scipy.spatial import ConvexHull
    hull = ConvexHull(corners_px)
    yy, xx = np.mgrid[0:H, 0:W]
    pts = np.stack([xx + 0.5, yy + 0.5], -1).reshape(-1, 2).astype(np.float64)
    d = (hull.equations[:, :2] @ pts.T + hull.equations[:, 2:3]).max(0)          # signed distance to the nearest violated / closest face (<= 0 inside)
    return (d <= 0).reshape(H, W), np.abs(d).reshape(H, W)


def test_oracle_debug_box_is_the_projection_of_the_cube():
    """One splat box seen from outside: the oracle's ray / box coverage equals the convex hull of the 8 projected corners
    (independent float64 geometry; pixels within 0.01 px of the outline excluded), its colour is (rgb * a, a), a box behind an
    occluder at its front face's depth is hidden, and a mirrored object transform (which flips the winding, hence draws the far
    faces) covers the same pixels."""
    a = fp32_point_asset([[0.3, -0.2, 0.0]])                        # rotation ~identity, scale 0.05, colour 0.5, opacity 1
    cam = camera.Camera(position=(0.4, 0.6, 3.0), pixelWidth=160, pixelHeight=120)
    for scale in ((1.0, 1.0, 1.0), (-1.0, 1.0, 1.0)):
        tr = camera.Transform(position=(0.1, 0.0, 0.0), rotation=(0.0, 0.259, 0.0, 0.966), scale=scale)
        P = camera.frame_params(cam, tr, splatScale=2.0)
        orc = O.Oracle(a)
        img = O.f16_to_f32(orc.draw_debug_boxes(P, chunks=False))
        dec = orc.decode_all()[0]
        pos, rot, sc = dec[0:3].astype(np.float64), dec[3:7].astype(np.float64), dec[7:10].astype(np.float64) * 2.0
        x, y, z, w = rot
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        o2w = tr.localToWorldMatrix.astype(np.float64)
        B = 2.0 * o2w[:3, :3] @ (R * sc[None, :])
        c = o2w[:3, :3] @ pos + o2w[:3, 3]
        vp = (cam.projectionMatrix.astype(np.float64) @ cam.worldToCameraMatrix.astype(np.float64))
        corners = []
        for k in range(8):
            l = np.array([1.0 if k & 1 else -1.0, 1.0 if k & 2 else -1.0, 1.0 if k & 4 else -1.0])
            cw = vp @ np.append(c + B @ l, 1.0)
            corners.append(((0.5 + 0.5 * cw[0] / cw[3]) * 160, (0.5 - 0.5 * cw[1] / cw[3]) * 120))
        inside, dist = _hull_coverage(np.array(corners), 160, 120)
        covered = img[..., 3] > 0
        sure = dist > 0.01
        assert covered.sum() > 200 and np.array_equal(covered[sure], inside[sure])
        alpha = np.float32(1.0)
        assert np.allclose(img[covered], [0.5 * alpha, 0.5 * alpha, 0.5 * alpha, alpha], atol=1e-3)
        # the drawn face's depth: an occluder just in front of the box hides it, one just behind its far side does not
        front = float(np.linalg.norm(c - np.asarray(cam.position, np.float64))) - 2.0 * np.linalg.norm(B, 2)
        hidden = orc.draw_debug_boxes(P, chunks=False, scene_depth=np.full((120, 160), front, np.float32))
        shown = orc.draw_debug_boxes(P, chunks=False, scene_depth=np.full((120, 160), front + 4.1 * np.linalg.norm(B, 2), np.float32))
        assert not hidden.any() and np.array_equal(O.f16_to_f32(shown)[..., 3] > 0, covered)
    # camera INSIDE the box: the faces that would be visible are the culled ones -- nothing is drawn
    cam_in = camera.Camera(position=(0.31, -0.19, 0.01), target=(0.0, 0.0, -5.0), pixelWidth=64, pixelHeight=48)
    assert not O.Oracle(a).draw_debug_boxes(camera.frame_params(cam_in, camera.Transform(), splatScale=2.0), chunks=False).any()


@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["Medium", "VeryHigh"])
def test_gpu_debug_box_modes(gpu_ctx, quality):
    a = small_asset(4_000, 6, quality)
    cam = default_camera(W=200, H=128, az=-35.0)
    orc = O.Oracle(a)
    depth = np.full((cam.pixelHeight, cam.pixelWidth), np.inf, np.float32)
    depth[:, :80] = 6.0
    cases = [(RenderMode.DebugBoxes, camera.Transform(), None, 0), (RenderMode.DebugBoxes, camera.Transform(scale=(-1.0, 1.0, 1.2)), depth, 0),
             (RenderMode.DebugBoxes, camera.Transform(), None, 1)]
    if a.chunkData is not None and len(a.chunkData):
        cases += [(RenderMode.DebugChunkBounds, camera.Transform(position=(0.2, 0.0, 0.0)), None, 0), (RenderMode.DebugChunkBounds, camera.Transform(), depth, 0)]
    for rm, tr, sd, mode in cases:
        r = GaussianSplatRenderer(gpu_ctx, a, transform=tr)
        r.OnEnable()
        r.m_SplatScale, r.m_OpacityScale, r.blendMode = 1.5, 0.6, mode
        rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
        r.m_RenderMode = rm
        rt.SetSceneDepth(sd)
        r.SortPoints(cam)                                               # the boxes go through the order buffer; no CalcViewData
        rt.Clear(); r.Draw(cam, rt)
        img = rt.Download()
        orc.reset_order()
        orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
        assert np.array_equal(r.DownloadOrder(), orc.order)
        P = r.FrameParams(cam)
        ref = orc.draw_debug_boxes(P, chunks=rm == RenderMode.DebugChunkBounds, mode=mode, scene_depth=sd)
        e = rt_err(img, ref, RT_TOL if mode == 0 else 4e-3)
        assert e <= (RT_TOL if mode == 0 else 4e-3), (rm, e)
        assert (O.f16_to_f32(ref)[..., 3] > 0).mean() > 0.05
        if sd is not None:
            assert rt_err(img, orc.draw_debug_boxes(P, chunks=rm == RenderMode.DebugChunkBounds, mode=mode)) > 0.01      # the occluder hides something
        # back to splats with the same renderer: the box draw invalidated the per-splat rectangles, calc_view restores them
        r.m_RenderMode = RenderMode.Splats
        rt.SetSceneDepth(None)
        r.blendMode = 0
        from unitygaussiansplatting_amd._lib import GsError
        with pytest.raises(GsError):
            rt.Clear(); r.Draw(cam, rt)
        r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        orc.calc_view(P)
        assert rt_err(rt.Download(), orc.draw(P, 0)) <= RT_TOL
        r.OnDisable(); rt.Dispose()
