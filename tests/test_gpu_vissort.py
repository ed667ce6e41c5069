"""-m gpu: GS_SORT_VISIBLE (cull before sort; csrc/gs_vissort.hip) against the reference's full sort.

The bar (bit-exact, like every order test): the order the mode draws from == the VISIBLE SUBSEQUENCE of the oracle's order buffer --
the oracle sorts all N splats, stably, through the previous order, on every SortPoints (GaussianSplatRenderer.cs:612-639;
SplatUtilities.compute:69-82) -- on every frame of sequences that revisit matrices, hold the camera still and skip sorts
(m_SortNthFrame), on scenes built to tie (lattices, duplicated positions, planes facing the camera); and the frame is bit-identical to
the one the library's own full-sort path composites.  Where the in-kernel fix-up gives up (a run of > 64 equal keys that the history
would have to order) the frame is reported (GS_ERR_TIE_OVERFLOW), the renderer rebuilds the reference's whole order buffer from the
kept matrices and the redraw is exact again."""
import numpy as np
import pytest

import oracle_lib as O
from common import default_camera, small_asset
from test_vissort_model import cams_for, tie_heavy_asset
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd._abi import GS_ERR_TIE_OVERFLOW, GS_SORT_VISIBLE
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget, SortMode
from vissort_model import VisibleSortModel, visible_bits

pytestmark = pytest.mark.gpu


def run_sequence(gpu_ctx, asset, cams, sort_every=1, expect_overflow=False, check_frames=True):
    """Every camera of `cams` is a frame: SortPoints (every sort_every-th frame), CalcViewData, Draw -- on a renderer in visible mode, on a
    second one in full mode, and on the oracle.  Returns (frames on which the mode was still active, overflow seen, max run of equal keys)."""
    rv = GaussianSplatRenderer(gpu_ctx, asset)
    rv.sortMode = SortMode.Visible
    rv.OnEnable()
    rf = GaussianSplatRenderer(gpu_ctx, asset)
    rf.sortMode = SortMode.Full
    rf.OnEnable()
    assert rv.SortModeActive() and not rf.SortModeActive()
    orc = O.Oracle(asset)
    model = VisibleSortModel(asset)
    W, H = cams[0].pixelWidth, cams[0].pixelHeight
    rtv, rtf = RenderTarget(gpu_ctx, W, H), RenderTarget(gpu_ctx, W, H)
    active_frames, overflowed, longest = 0, False, 0
    for k, cam in enumerate(cams):
        if k % sort_every == 0:
            m = camera.sort_matrix(cam, rv.transform.localToWorldMatrix)
            rv.SortPoints(cam); rf.SortPoints(cam); orc.sort(m); model.push(m)
        was_active = rv.SortModeActive()
        rv.CalcViewData(cam); rtv.Clear(); rv.Draw(cam, rtv)
        rf.CalcViewData(cam); rtf.Clear(); rf.Draw(cam, rtf)
        try:
            st = rv.FrameStats()
        except GsError as e:
            assert e.code == GS_ERR_TIE_OVERFLOW and was_active, f"frame {k}: {e}"
            overflowed = True
            assert not rv.SortModeActive()                      # the renderer now sorts all splats ...
            rv.CalcViewData(cam); rtv.Clear(); rv.Draw(cam, rtv)  # ... and the frame is drawn again
            st = rv.FrameStats()
        P = rv.FrameParams(cam)
        orc.calc_view(P)
        vis = visible_bits(orc, P)
        longest = max(longest, model.longest_run(vis))
        want = orc.order[vis[orc.order]]
        if rv.SortModeActive():
            active_frames += 1
            assert st.sort_mode == GS_SORT_VISIBLE and st.tie_exhausted == 0
            got = rv.DownloadVisibleOrder()
            assert len(got) == len(want) == st.visible_splats, f"frame {k}: V = {len(got)} vs {len(want)}"
            assert np.array_equal(got, want), f"frame {k}: visible order differs at {int(np.argmax(got != want))} of {len(want)}"
            assert np.array_equal(model.visible_order(vis), want)
        else:
            assert np.array_equal(rv.DownloadOrder(), orc.order), f"frame {k}: the rebuilt order buffer differs from the reference's"
        assert st.visible_splats == int(vis.sum())
        if check_frames:
            assert np.array_equal(rtv.Download(), rtf.Download()), f"frame {k}: the frame differs from the full-sort path's"
    assert overflowed == expect_overflow
    # the whole reference buffer on demand (one full sort per kept matrix), while the mode stays active
    assert np.array_equal(rv.DownloadOrder(), orc.order)
    for x in (rv, rf):
        x.OnDisable()
    rtv.Dispose(); rtf.Dispose()
    return active_frames, overflowed, longest


def orbit(n, step=3.0, W=320, H=200, **kw):
    return [default_camera(W, H, az=step * k, **kw) for k in range(n)]


@pytest.mark.parametrize("quality", ["Medium", "VeryHigh", "VeryLow"])
def test_orbit_with_revisits_and_still_frames(gpu_ctx, quality):
    """Moving camera, two still frames, a return to an earlier pose (its matrix moves to the front of the history) -- chunk-quantised
    positions (Medium / VeryLow: Norm11 / Norm6 lattices per chunk, many natural ties) and fp32 ones."""
    a = small_asset(60_000, 7, quality)
    cams = orbit(5) + [default_camera(az=12.0)] * 2 + [default_camera(az=3.0), default_camera(az=40.0, elev=35.0), default_camera(az=3.0)]
    active, _, _ = run_sequence(gpu_ctx, a, cams)
    assert active == len(cams)


def test_no_sort_yet_draws_in_index_order(gpu_ctx):
    """Before the first SortPoints the order buffer is CSSetIndices' identity: the visible splats in index order."""
    a = small_asset(20_000, 5, "Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.sortMode = SortMode.Visible
    r.OnEnable()
    rf = GaussianSplatRenderer(gpu_ctx, a)
    rf.sortMode = SortMode.Full
    rf.OnEnable()
    cam = default_camera()
    rt, rt2 = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight), RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    rf.CalcViewData(cam); rt2.Clear(); rf.Draw(cam, rt2)
    got = r.DownloadVisibleOrder()
    assert len(got) == r.FrameStats().visible_splats > 1000 and np.all(np.diff(got.astype(np.int64)) > 0)
    assert np.array_equal(rt.Download(), rt2.Download())
    r.OnDisable(); rf.OnDisable(); rt.Dispose(); rt2.Dispose()


@pytest.mark.parametrize("nth", [3, 5])
def test_sort_nth_frame_uses_the_stale_matrix_on_this_frames_visible_set(gpu_ctx, nth):
    a = small_asset(60_000, 9, "Medium")
    active, _, _ = run_sequence(gpu_ctx, a, orbit(9, step=6.0), sort_every=nth)
    assert active == 9


def test_static_camera_then_a_move(gpu_ctx):
    a = small_asset(40_000, 4, "Medium")
    cams = [default_camera(az=25.0)] * 20 + orbit(3, step=2.0, elev=30.0)
    active, _, _ = run_sequence(gpu_ctx, a, cams, check_frames=False)
    assert active == len(cams)


@pytest.mark.parametrize("kind", ["lattice", "duplicates"])
def test_tie_adversarial_scenes_within_the_fix_up(gpu_ctx, kind):
    """lattice: a yaw-only camera ties whole columns (runs of 5..30, ordered by a wave from the pitched views earlier in the sequence);
    duplicates: every position three times (tied under every matrix: index order)."""
    a = tie_heavy_asset(kind)
    active, _, longest = run_sequence(gpu_ctx, a, cams_for(kind))
    assert active == len(cams_for(kind))
    assert longest >= (5 if kind == "lattice" else 3) and longest <= 64


def test_run_longer_than_a_wave_falls_back_to_the_full_sort(gpu_ctx):
    """planes: two planes facing a z-axis camera = two runs of ~3000 equal keys, which the earlier pitched view ordered: reported, the whole
    order buffer rebuilt from the kept matrices, the redraw exact; the renderer stays on full sorts until ResetOrder."""
    a = tie_heavy_asset("planes")
    active, overflowed, longest = run_sequence(gpu_ctx, a, cams_for("planes"), expect_overflow=True)
    assert overflowed and longest > 64 and 1 <= active < len(cams_for("planes"))


def test_all_equal_keys_from_the_start_need_no_history(gpu_ctx):
    """A plane facing the camera from the FIRST sort on: one giant run, but nothing earlier to order it by -- index order, no overflow."""
    a = tie_heavy_asset("planes")
    z_axis = camera.Camera(position=(0.0, 0.0, 6.0), pixelWidth=320, pixelHeight=200)
    active, overflowed, longest = run_sequence(gpu_ctx, a, [z_axis] * 3)
    assert active == 3 and not overflowed and longest > 64


def test_mode_switches_keep_the_reference_order(gpu_ctx):
    """Visible -> Full hands the reference's order buffer over (rebuilt from the kept matrices); Full -> Visible waits for ResetOrder;
    UploadOrder parks the mode; ResetOrder restarts it."""
    a = small_asset(30_000, 6, "Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.sortMode = SortMode.Visible
    r.OnEnable()
    orc = O.Oracle(a)
    cams = orbit(4, step=5.0)
    rt = RenderTarget(gpu_ctx, 320, 200)
    for cam in cams[:3]:
        r.SortPoints(cam); orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
        r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    r.SetSortMode(SortMode.Full)
    assert not r.SortModeActive() and np.array_equal(r.DownloadOrder(), orc.order)
    r.SortPoints(cams[3]); orc.sort(camera.sort_matrix(cams[3], r.transform.localToWorldMatrix))
    assert np.array_equal(r.DownloadOrder(), orc.order)
    r.SetSortMode(SortMode.Visible)
    assert not r.SortModeActive()                               # the buffer holds full sorts: not until CSSetIndices
    r.ResetOrder(); orc.reset_order()
    assert r.SortModeActive()
    r.SortPoints(cams[1]); orc.sort(camera.sort_matrix(cams[1], r.transform.localToWorldMatrix))
    r.CalcViewData(cams[1]); rt.Clear(); r.Draw(cams[1], rt)
    P = r.FrameParams(cams[1]); orc.calc_view(P); vis = visible_bits(orc, P)
    assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
    r.UploadOrder(orc.order[::-1].copy())
    assert not r.SortModeActive()
    r.OnDisable(); rt.Dispose()


def test_debug_boxes_through_the_visible_order(gpu_ctx):
    from unitygaussiansplatting_amd.renderer import RenderMode
    a = small_asset(3000, 8, "Medium")
    cams = orbit(3, step=20.0, W=160, H=100)
    frames = []
    for mode in (SortMode.Visible, SortMode.Full):
        r = GaussianSplatRenderer(gpu_ctx, a)
        r.sortMode = mode
        r.m_RenderMode = RenderMode.DebugBoxes
        r.OnEnable()
        rt = RenderTarget(gpu_ctx, 160, 100)
        out = []
        for cam in cams:
            r.SortPoints(cam); rt.Clear(); r.Draw(cam, rt); out.append(rt.Download())
        frames.append(out)
        r.OnDisable(); rt.Dispose()
    for x, y in zip(*frames):
        assert np.array_equal(x, y) and x.any()


def test_c2_three_frames(gpu_ctx):
    """Full size: the bench scene (6,131,954 splats, 1200x797), three frames of the bench orbit (0.25 deg / frame)."""
    cfg = scenes.CONFIGS["C2"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2")
    cams = [camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * k), pixelWidth=cfg.width, pixelHeight=cfg.height,
                          fieldOfView=cfg.fov_y) for k in range(3)]
    active, _, longest = run_sequence(gpu_ctx, a, cams)
    assert active == 3 and 2 <= longest <= 64
