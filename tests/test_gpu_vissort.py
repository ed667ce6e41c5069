"""-m gpu: GS_SORT_VISIBLE (cull before sort; csrc/gs_vissort.hip) against the reference's full sort.

The bar (bit-exact, like every order test): the order the mode draws from == the VISIBLE SUBSEQUENCE of the oracle's order buffer --
the oracle sorts all N splats, stably, through the previous order, on EVERY SortPoints, sequentially (GaussianSplatRenderer.cs:612-639;
SplatUtilities.compute:69-82; GpuSorting.cs:142-198) -- on every frame of sequences that revisit matrices, hold the camera still, skip
sorts (m_SortNthFrame), fly in a straight line with a fixed orientation, run past the library's history of sort matrices (default 128
rows; most sequences are also run with a limit of a few rows, so that the base order is consolidated every few frames), on scenes built
to tie (lattices, duplicated and co-located positions, planes facing the camera: runs of thousands of equal keys); and the frame is
bit-identical to the one the library's own full-sort path composites.  There is no case the mode reports instead of ordering (ABI 8)."""
import numpy as np
import pytest

import oracle_lib as O
from common import default_camera, small_asset
from test_vissort_model import cams_for, tie_heavy_asset
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd._abi import GS_SORT_VISIBLE
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget, SortMode
from vissort_model import VisibleSortModel, visible_bits

pytestmark = pytest.mark.gpu


def run_sequence(gpu_ctx, asset, cams, sort_every=1, check_frames=True, history_limit=None, check_order=None, full_order_at=(), use_model=True):
    """Every camera of `cams` is a frame: SortPoints (every sort_every-th frame), CalcViewData, Draw -- on a renderer in visible mode, on a
    second one in full mode, and on the oracle (which sorts on EVERY SortPoints, whether or not the frame is checked).
    check_order: frames whose drawn order is compared (default all); full_order_at: frames after which the whole reference buffer is asked
    for (gs_renderer_download_order: the recorded sorts carried out on all N) and compared.  Returns statistics."""
    rv = GaussianSplatRenderer(gpu_ctx, asset)
    rv.sortMode = SortMode.Visible
    rv.OnEnable()
    if history_limit is not None:
        rv.SetSortHistoryLimit(history_limit)
    rf = GaussianSplatRenderer(gpu_ctx, asset)
    rf.sortMode = SortMode.Full
    rf.OnEnable()
    assert rv.SortModeActive() and not rf.SortModeActive()
    orc = O.Oracle(asset)
    model = VisibleSortModel(asset, depth=10 ** 9) if use_model else None
    W, H = cams[0].pixelWidth, cams[0].pixelHeight
    rtv, rtf = RenderTarget(gpu_ctx, W, H), RenderTarget(gpu_ctx, W, H)
    longest, long_runs, distinct = 0, 0, set()
    for k, cam in enumerate(cams):
        if k % sort_every == 0:
            m = camera.sort_matrix(cam, rv.transform.localToWorldMatrix)
            distinct.add(np.asarray(m, np.float32).reshape(16)[8:12].tobytes())
            rv.SortPoints(cam); rf.SortPoints(cam); orc.sort(m)
            if model:
                model.push(m)
        rv.CalcViewData(cam); rtv.Clear(); rv.Draw(cam, rtv)
        st = rv.FrameStats()                                     # (never raises for the order: there is nothing the mode cannot order)
        assert rv.SortModeActive() and st.sort_mode == GS_SORT_VISIBLE
        long_runs += st.tie_long_runs
        longest = max(longest, st.tie_longest_run)
        if check_order is None or k in check_order:
            P = rv.FrameParams(cam)
            orc.calc_view(P)
            vis = visible_bits(orc, P)
            want = orc.order[vis[orc.order]]
            got = rv.DownloadVisibleOrder()
            assert len(got) == len(want) == st.visible_splats, f"frame {k}: V = {len(got)} vs {len(want)}"
            assert np.array_equal(got, want), f"frame {k}: visible order differs at {int(np.argmax(got != want))} of {len(want)}"
            if model:
                assert np.array_equal(model.visible_order(vis), want)
                longest = max(longest, model.longest_run(vis))
            if check_frames:
                rf.CalcViewData(cam); rtf.Clear(); rf.Draw(cam, rtf)
                assert np.array_equal(rtv.Download(), rtf.Download()), f"frame {k}: the frame differs from the full-sort path's"
        if k in full_order_at:
            assert np.array_equal(rv.DownloadOrder(), orc.order), f"frame {k}: the order buffer (recorded sorts carried out on all N) differs from the reference's"
            assert rv.SortModeActive()
    # the whole reference buffer on demand (one full sort + the chain fix-up over N), while the mode stays set
    assert np.array_equal(rv.DownloadOrder(), orc.order)
    assert np.array_equal(rf.DownloadOrder(), orc.order)
    rows, limit, consolidations = rv.SortHistory()
    for x in (rv, rf):
        x.OnDisable()
    rtv.Dispose(); rtf.Dispose()
    return dict(longest=longest, long_runs=long_runs, consolidations=consolidations, distinct=len(distinct), limit=limit)


def orbit(n, step=3.0, W=320, H=200, **kw):
    return [default_camera(W, H, az=step * k, **kw) for k in range(n)]


def flight(n, start=(0.5, 0.4, 7.0), step=(0.01, -0.02, -0.07), yaw=8.0, pitch=-5.0, W=320, H=200):
    """A straight line with a FIXED orientation: every frame's sort matrix has the same direction and another constant."""
    cy, sy, cp, sp = np.cos(np.radians(yaw)), np.sin(np.radians(yaw)), np.cos(np.radians(pitch)), np.sin(np.radians(pitch))
    R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    R = R @ np.diag([1.0, 1.0, -1.0])                            # (Unity's forward is +Z: looking towards the scene from z > 0)
    return [camera.Camera(position=tuple(np.asarray(start) + k * np.asarray(step)), rotation=R, pixelWidth=W, pixelHeight=H) for k in range(n)]


@pytest.mark.parametrize("quality", ["Medium", "VeryHigh", "VeryLow"])
@pytest.mark.parametrize("limit", [None, 3])
def test_orbit_with_revisits_and_still_frames(gpu_ctx, quality, limit):
    """Moving camera, two still frames, a return to an earlier pose (its matrix moves to the front of the history) -- chunk-quantised
    positions (Medium / VeryLow: Norm11 / Norm6 lattices per chunk, many natural ties) and fp32 ones; with the default history and with
    one of three rows (a consolidation every other frame)."""
    a = small_asset(60_000, 7, quality)
    cams = orbit(5) + [default_camera(az=12.0)] * 2 + [default_camera(az=3.0), default_camera(az=40.0, elev=35.0), default_camera(az=3.0)]
    s = run_sequence(gpu_ctx, a, cams, history_limit=limit)
    assert (s["consolidations"] >= 2) if limit else (s["consolidations"] == 1)      # (1: the download_order at the end)


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
    rf.CalcViewData(cam); rtf = rt2; rtf.Clear(); rf.Draw(cam, rtf)
    got = r.DownloadVisibleOrder()
    assert len(got) == r.FrameStats().visible_splats > 1000 and np.all(np.diff(got.astype(np.int64)) > 0)
    assert np.array_equal(rt.Download(), rt2.Download())
    r.OnDisable(); rf.OnDisable(); rt.Dispose(); rt2.Dispose()


@pytest.mark.parametrize("nth", [3, 5])
def test_sort_nth_frame_uses_the_stale_matrix_on_this_frames_visible_set(gpu_ctx, nth):
    a = small_asset(60_000, 9, "Medium")
    run_sequence(gpu_ctx, a, orbit(9, step=6.0), sort_every=nth)
    run_sequence(gpu_ctx, a, orbit(9, step=6.0), sort_every=nth, history_limit=2)


def test_static_camera_then_a_move(gpu_ctx):
    a = small_asset(40_000, 4, "Medium")
    cams = [default_camera(az=25.0)] * 20 + orbit(3, step=2.0, elev=30.0)
    s = run_sequence(gpu_ctx, a, cams, check_frames=False)
    assert s["distinct"] == 4 and s["consolidations"] == 1


@pytest.mark.parametrize("kind", ["lattice", "duplicates", "colocated"])
@pytest.mark.parametrize("limit", [None, 2])
def test_tie_adversarial_scenes(gpu_ctx, kind, limit):
    """lattice: a yaw-only camera ties whole columns (runs of 5..30, ordered by a wave from the pitched views earlier in the sequence);
    duplicates: every position three times (tied under every matrix: the base order decides); colocated: 150 splats at each of a few
    points -- runs of more than 64 whose members no matrix can separate (the advisor's case: no error, the base order)."""
    a = tie_heavy_asset(kind)
    s = run_sequence(gpu_ctx, a, cams_for(kind), history_limit=limit)
    assert s["longest"] >= (5 if kind == "lattice" else 3)
    if kind == "colocated":
        assert s["long_runs"] > 0 and s["longest"] >= 150


@pytest.mark.parametrize("limit", [None, 2])
def test_runs_of_thousands_are_ordered_in_the_frame_that_meets_them(gpu_ctx, limit):
    """planes: two planes facing a z-axis camera = two runs of ~3000 equal keys, which the earlier pitched view ordered: the workgroup-wide
    sorting network orders them in that very frame (ABI 7 drew it in index order, reported it and fell back to full sorts)."""
    a = tie_heavy_asset("planes")
    s = run_sequence(gpu_ctx, a, cams_for("planes"), history_limit=limit)
    assert s["longest"] > 1000 and s["long_runs"] >= 2


def test_all_equal_keys_from_the_start(gpu_ctx):
    """A plane facing the camera from the FIRST sort on: one giant run, nothing earlier to order it by -- the base (index) order."""
    a = tie_heavy_asset("planes")
    z_axis = camera.Camera(position=(0.0, 0.0, 6.0), pixelWidth=320, pixelHeight=200)
    run_sequence(gpu_ctx, a, [z_axis] * 3)


@pytest.mark.parametrize("quality,limit", [("Medium", None), ("VeryHigh", 8)])
def test_fixed_orientation_flight_past_32_matrices(gpu_ctx, quality, limit):
    """An initial rotation (three pitched / yawed views), then 45 frames of a straight-line flight with a FIXED orientation: 48 distinct
    sort matrices, 45 of them with the same direction -- pairs tied under one of them tend to stay tied under all, so what the initial
    rotation made of them has to survive the whole flight.  (ABI 7 kept 32 rows and then fell back to the index, silently.)
    Medium: Norm11 lattices per chunk; VeryHigh: the 10 x 30 x 10 lattice, exact."""
    a = tie_heavy_asset("lattice", quality=quality)
    pitched = lambda deg: camera.Camera(position=scenes.orbit_eye(6.0, 25.0, deg), pixelWidth=320, pixelHeight=200)
    cams = [pitched(10.0), pitched(70.0), default_camera(az=-20.0, elev=-15.0)] + flight(45)
    s = run_sequence(gpu_ctx, a, cams, history_limit=limit, full_order_at=(40,))
    assert s["distinct"] >= 47 and s["longest"] >= 4
    assert s["consolidations"] >= (5 if limit else 2)


def test_more_sorts_than_the_default_history_holds(gpu_ctx):
    """140 distinct matrices with the default limit (128 rows): the library consolidates by itself, once, and stays exact."""
    a = small_asset(30_000, 11, "Medium")
    cams = orbit(70, step=1.5) + flight(70, start=(1.0, 1.5, 6.5), step=(-0.01, -0.01, -0.03))
    s = run_sequence(gpu_ctx, a, cams, check_frames=False, check_order={0, 60, 126, 127, 128, 129, 139}, use_model=False)
    assert s["distinct"] == 140 and s["limit"] == 128 and s["consolidations"] == 2      # (at the 129th row, and the download at the end)


def test_long_run_met_at_frame_40(gpu_ctx):
    """40 frames of orbit over the planes scene (every one a distinct matrix that orders the planes' splats), then the z-axis camera: two runs
    of ~3000 equal keys whose order is the whole history's."""
    a = tie_heavy_asset("planes")
    z_axis = camera.Camera(position=(0.0, 0.0, 6.0), pixelWidth=320, pixelHeight=200)
    cams = [camera.Camera(position=scenes.orbit_eye(6.0, 20.0, 4.0 * k), pixelWidth=320, pixelHeight=200) for k in range(40)] + [z_axis, z_axis]
    for limit in (None, 16):
        s = run_sequence(gpu_ctx, a, cams, history_limit=limit, check_order={0, 20, 39, 40, 41}, use_model=False)
        assert s["longest"] > 1000


def test_c2_orbit_of_40_frames(gpu_ctx):
    """Full size: the bench scene (6,131,954 splats, 1200x797), 40 frames of the bench orbit (0.25 deg / frame), a history of 16 rows: two
    consolidations of 6 M splats on the way; the drawn order checked on five frames, the whole buffer at frame 24 and at the end."""
    cfg = scenes.CONFIGS["C2"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2")
    cams = [camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * k), pixelWidth=cfg.width, pixelHeight=cfg.height,
                          fieldOfView=cfg.fov_y) for k in range(40)]
    s = run_sequence(gpu_ctx, a, cams, history_limit=16, check_order={0, 1, 17, 33, 39}, full_order_at=(24,), use_model=False)
    assert s["consolidations"] >= 3


def test_c2_three_frames_default_history(gpu_ctx):
    cfg = scenes.CONFIGS["C2"]
    a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2")
    cams = [camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * k), pixelWidth=cfg.width, pixelHeight=cfg.height,
                          fieldOfView=cfg.fov_y) for k in range(3)]
    s = run_sequence(gpu_ctx, a, cams)
    assert 2 <= s["longest"] <= 64


def test_mode_switches_keep_the_reference_order(gpu_ctx):
    """Visible -> Full at frame 40 hands the reference's order buffer over (the recorded sorts carried out on all N); Full -> Visible
    continues from whatever the buffer holds (its ranks end the tie chain); UploadOrder makes the uploaded order the base (a frame
    without SortPoints draws in it); ResetOrder restarts from CSSetIndices' identity."""
    a = tie_heavy_asset("lattice", quality="Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.sortMode = SortMode.Visible
    r.OnEnable()
    orc = O.Oracle(a)
    rt = RenderTarget(gpu_ctx, 320, 200)

    def frame(cam, sort=True):
        if sort:
            r.SortPoints(cam); orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
        r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        P = r.FrameParams(cam); orc.calc_view(P); vis = visible_bits(orc, P)
        return vis

    cams = orbit(20, step=4.0, elev=20.0) + flight(20)
    for cam in cams:
        vis = frame(cam)
    assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
    r.SetSortMode(SortMode.Full)                                 # frame 40
    assert not r.SortModeActive() and np.array_equal(r.DownloadOrder(), orc.order)
    more = orbit(3, step=7.0, elev=-10.0)
    frame(more[0])
    assert np.array_equal(r.DownloadOrder(), orc.order)
    r.SetSortMode(SortMode.Visible)
    assert r.SortModeActive()                                    # the buffer of full sorts is the new base
    vis = frame(more[0], sort=False)                             # no SortPoints yet on this base: drawn in the base order
    assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
    for cam in (more[1], more[2], cams[25], more[1]):
        vis = frame(cam)
        assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
    assert np.array_equal(r.DownloadOrder(), orc.order)
    rev = orc.order[::-1].copy()
    r.UploadOrder(rev); orc.order[:] = rev
    assert r.SortModeActive()
    vis = frame(more[2], sort=False)
    assert np.array_equal(r.DownloadVisibleOrder(), rev[vis[rev]])
    for cam in (more[0], cams[3]):
        vis = frame(cam)
        assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
    d = r.DownloadDistances()                                    # m_GpuSortDistances: the sorted keys of the last SortPoints, all N
    assert np.all(np.diff(d.astype(np.int64)) >= 0) and np.array_equal(r.DownloadOrder(), orc.order)
    r.ResetOrder(); orc.reset_order()
    assert np.array_equal(r.DownloadDistances(), d)              # ... kept across CSSetIndices, as in the reference
    vis = frame(cams[1])
    assert np.array_equal(r.DownloadVisibleOrder(), orc.order[vis[orc.order]])
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


def test_frames_dealt_to_two_renderers_sharing_one_asset(gpu_ctx):
    """Frames in flight: two renderers on two contexts (streams) over ONE copy of the asset's device blobs (ShareResourcesOf), the frames dealt
    alternately, every renderer told every SortPoints matrix (bookkeeping in this mode): each frame is drawn from the reference's order -- the
    visible subsequence of the oracle's buffer -- and is the same bits as the frame of one renderer drawing every frame."""
    from unitygaussiansplatting_amd.renderer import GpuContext
    a = tie_heavy_asset("lattice", quality="Medium")
    seq = GaussianSplatRenderer(gpu_ctx, a)
    seq.sortMode = SortMode.Visible
    seq.OnEnable()
    ctx2 = GpuContext(0)
    lanes = [GaussianSplatRenderer(gpu_ctx, a), GaussianSplatRenderer(ctx2, a)]
    for L in lanes:
        L.sortMode = SortMode.Visible
    lanes[0].OnEnable()
    lanes[1].ShareResourcesOf(lanes[0])
    lanes[1].SetSortHistoryLimit(3)                              # (the lanes need not agree on when they consolidate)
    orc = O.Oracle(a)
    rts = [RenderTarget(gpu_ctx, 320, 200), RenderTarget(ctx2, 320, 200)]
    rt_seq = RenderTarget(gpu_ctx, 320, 200)
    cams = orbit(6, step=5.0, elev=20.0) + flight(6)
    for k, cam in enumerate(cams):
        orc.sort(camera.sort_matrix(cam, seq.transform.localToWorldMatrix))
        seq.SortPoints(cam); seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq)
        for L in lanes:
            L.SortPoints(cam)
        L, rt = lanes[k % 2], rts[k % 2]
        L.CalcViewData(cam); rt.Clear(); L.Draw(cam, rt)
        P = L.FrameParams(cam); orc.calc_view(P); vis = visible_bits(orc, P)
        assert np.array_equal(L.DownloadVisibleOrder(), orc.order[vis[orc.order]]), f"frame {k}"
        assert np.array_equal(rt.Download(), rt_seq.Download()), f"frame {k}"
    for L in lanes:
        assert np.array_equal(L.DownloadOrder(), orc.order)
    lanes[1].OnDisable(); lanes[0].OnDisable(); seq.OnDisable()
    for t in rts + [rt_seq]:
        t.Dispose()


def test_views_in_flight_helper_gives_the_sequential_frames(gpu_ctx):
    """renderer.ViewsInFlight: the 8 cameras of a C5-style batch dealt to two lanes, twice (the second batch continues the sort history): every
    target == the target a single renderer drawing the cameras one after the other produces, bit for bit."""
    from unitygaussiansplatting_amd.renderer import ViewsInFlight
    a = small_asset(40_000, 12, "Medium")
    seq = GaussianSplatRenderer(gpu_ctx, a)
    seq.sortMode = SortMode.Visible
    seq.OnEnable()
    main = GaussianSplatRenderer(gpu_ctx, a)
    main.OnEnable()
    vif = ViewsInFlight(main, lanes=2)
    rt = RenderTarget(gpu_ctx, 320, 200)
    for batch in range(2):
        cams = [default_camera(az=45.0 * k + 7.0 * batch, elev=12.0) for k in range(8)]
        want = []
        for cam in cams:
            seq.SortPoints(cam); seq.CalcViewData(cam); rt.Clear(); seq.Draw(cam, rt)
            want.append(rt.Download())
        got = [t.Download() for t in vif.Render(cams)]
        for k in range(8):
            assert np.array_equal(got[k], want[k]) and want[k].any(), f"batch {batch}, view {k}"
    assert np.array_equal(vif.lanes[1].DownloadOrder(), seq.DownloadOrder())
    vif.Dispose(); main.OnDisable(); seq.OnDisable(); rt.Dispose()


def test_a_run_of_tens_of_thousands(gpu_ctx):
    """60,000 splats on two planes facing the camera, after a pitched view: two runs of ~30,000 equal keys each, ordered by the network through
    global memory in the frame that meets them (slow by design -- milliseconds -- but the reference's order)."""
    a = tie_heavy_asset("planes", n=60_000)
    s = run_sequence(gpu_ctx, a, cams_for("planes")[:3], use_model=False, check_frames=False)
    assert s["longest"] > 20_000 and s["long_runs"] >= 2


def _lane_pair(gpu_ctx, a, lanes=2, limit=3):
    """(one frame at a time, frames in flight inside the library): two renderers in visible mode on the same context and asset"""
    seq = GaussianSplatRenderer(gpu_ctx, a)
    seq.sortMode = SortMode.Visible
    seq.OnEnable()
    lib = GaussianSplatRenderer(gpu_ctx, a)
    lib.sortMode = SortMode.Visible
    lib.OnEnable()
    lib.SetFramesInFlight(lanes)
    assert lib.FramesInFlight() == (lanes, True)
    if limit:
        lib.SetSortHistoryLimit(limit)
    return seq, lib


def test_frames_in_flight_inside_the_library(gpu_ctx):
    """gs_renderer_set_frames_in_flight: ONE renderer, the reference's calls (SortPoints, CalcViewData, Draw, resolve), the frames dealt to lanes on streams of
    their own inside the library.  Every frame's drawn order == the visible subsequence of the oracle's buffer, every frame == the frame of a renderer
    that draws one at a time, bit for bit -- checked frame by frame, then over a burst that is never synchronised in between (three targets in rotation),
    then across a switch to GS_SORT_FULL and back (the lanes take over a base that is no longer the identity), UploadOrder, ResetOrder, and with the
    lanes freed again."""
    a = tie_heavy_asset("lattice", quality="Medium")
    seq, lib = _lane_pair(gpu_ctx, a, lanes=2, limit=3)
    orc = O.Oracle(a)
    W, H = 320, 200
    rt_seq = RenderTarget(gpu_ctx, W, H)
    rts = [RenderTarget(gpu_ctx, W, H) for _ in range(3)]

    def sort_all(cam):
        orc.sort(camera.sort_matrix(cam, seq.transform.localToWorldMatrix)); seq.SortPoints(cam); lib.SortPoints(cam)

    def checked_frame(k, cam, rt):
        sort_all(cam)
        seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq)
        lib.CalcViewData(cam); rt.Clear(); lib.Draw(cam, rt)
        P = lib.FrameParams(cam); orc.calc_view(P); vis = visible_bits(orc, P)
        assert np.array_equal(lib.DownloadVisibleOrder(), orc.order[vis[orc.order]]), f"frame {k}"
        st = lib.FrameStats()
        assert st.sort_mode == GS_SORT_VISIBLE and st.visible_splats == int(vis.sum())
        assert np.array_equal(rt.Download(), rt_seq.Download()), f"frame {k}"

    cams = orbit(7, step=5.0, elev=20.0) + flight(6)
    for k, cam in enumerate(cams):
        checked_frame(k, cam, rts[k % 3])
    assert np.array_equal(lib.DownloadOrder(), orc.order)        # the owner's own bookkeeping answers for the whole buffer
    # a burst without a host sync: 9 frames into 3 targets in rotation, each resolved (on the context's stream: behind its frame's blend) before it is reused
    burst = orbit(9, step=4.0, elev=-10.0)
    want = []
    for cam in burst:
        orc.sort(camera.sort_matrix(cam, seq.transform.localToWorldMatrix)); seq.SortPoints(cam)
        seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq); want.append((rt_seq.Download(), rt_seq.Resolve((0.1, 0.2, 0.3, 1.0))))
    got = []
    for k, cam in enumerate(burst):
        rt = rts[k % 3]
        if k >= 3:
            got.append((rt.Download(), None))                    # (Download blocks on the context's stream only: the frame drawn into rt three frames ago must be there)
        lib.SortPoints(cam); lib.CalcViewData(cam); rt.Clear(); lib.Draw(cam, rt); rt.ResolveAsync((0.1, 0.2, 0.3, 1.0))
    for k in range(6, 9):
        rt = rts[k % 3]
        got.append((rt.Download(), rt.Resolve((0.1, 0.2, 0.3, 1.0))))
    for k, ((g, gr), (w, wr)) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), f"burst frame {k}"
        if gr is not None:
            assert np.array_equal(gr, wr), f"burst frame {k} (resolved)"
    # GS_SORT_FULL for three frames (the lanes idle, the owner sorts all N itself), then back: the lanes' base is the buffer those sorts left
    for x in (seq, lib):
        x.SetSortMode(SortMode.Full)
    assert lib.FramesInFlight() == (2, False)
    for k, cam in enumerate(orbit(3, step=9.0, elev=35.0)):
        sort_all(cam)
        lib.CalcViewData(cam); rts[0].Clear(); lib.Draw(cam, rts[0])
        seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq)
        assert np.array_equal(rts[0].Download(), rt_seq.Download())
    assert np.array_equal(lib.DownloadOrder(), orc.order)
    for x in (seq, lib):
        x.SetSortMode(SortMode.Visible)
    assert lib.FramesInFlight() == (2, True)
    for k, cam in enumerate(flight(5, start=(-0.3, 0.2, 6.0))):
        checked_frame(100 + k, cam, rts[k % 3])
    # an uploaded order, then CSSetIndices
    perm = np.random.default_rng(11).permutation(a.splatCount).astype(np.uint32)
    for x in (seq, lib):
        x.UploadOrder(perm)
    orc.order[:] = perm
    for k, cam in enumerate(orbit(4, step=6.0, elev=5.0)):
        checked_frame(200 + k, cam, rts[k % 3])
    for x in (seq, lib):
        x.ResetOrder()
    orc.order[:] = np.arange(a.splatCount, dtype=np.uint32)
    for k, cam in enumerate(orbit(4, step=-6.0, elev=15.0)):
        checked_frame(300 + k, cam, rts[k % 3])
    assert np.array_equal(lib.DownloadOrder(), orc.order)
    # the lanes freed: the renderer carries on by itself from the same order
    lib.SetFramesInFlight(1)
    assert lib.FramesInFlight() == (1, False)
    for k, cam in enumerate(orbit(3, step=7.0, elev=-5.0)):
        checked_frame(400 + k, cam, rts[k % 3])
    assert np.array_equal(lib.DownloadOrder(), orc.order)
    for x in (lib, seq):
        x.OnDisable()
    for t in rts + [rt_seq]:
        t.Dispose()


def test_library_lanes_get_every_setting(gpu_ctx):
    """Cutouts, deleted bits, the blend mode, a pinned tile shape, the view-buffer mode and a scene-depth attachment reach the lanes -- set before the lanes exist
    and changed while they do: every frame equals the one-at-a-time renderer's, the 40-byte view records the oracle's."""
    from test_cutouts import CUTOUT_SETS
    from unitygaussiansplatting_amd.cutout import shader_data_array
    a = small_asset(30_011, 7, "Medium")
    seq = GaussianSplatRenderer(gpu_ctx, a)
    lib = GaussianSplatRenderer(gpu_ctx, a)
    bits = np.random.default_rng(3).integers(0, 2 ** 32, (a.splatCount + 31) // 32, dtype=np.uint64).astype(np.uint32)
    bits &= np.random.default_rng(5).integers(0, 2 ** 32, len(bits), dtype=np.uint64).astype(np.uint32)
    for x in (seq, lib):
        x.sortMode = SortMode.Visible
        x.OnEnable()
        x.m_Cutouts = CUTOUT_SETS["hole_ellipsoid"]
        x.SetDeletedBits(bits)
        x.UpdateCutoutsBuffer()
    lib.SetFramesInFlight(3)                                     # (made with the settings above in place)
    orc = O.Oracle(a)
    W, H = 320, 200
    rt_seq, rt = RenderTarget(gpu_ctx, W, H), RenderTarget(gpu_ctx, W, H)
    depth = np.full((H, W), 1.0e9, np.float32)
    depth[:, : W // 2] = 4.0                                     # an opaque wall in front of most of the scene on the left half
    steps = [("hole_ellipsoid", True, (0, 0), False), ("crop_box", True, (16, 16), False), ("hole_then_crop", False, (32, 32), True), (None, False, (0, 0), True),
             (None, True, (32, 16), False), ("crop_box", False, (0, 0), False)]
    for k, (name, use_bits, tile, with_depth) in enumerate(steps):
        cam = default_camera(W, H, az=11.0 * k, elev=10.0)
        for x in (seq, lib):
            x.m_Cutouts = CUTOUT_SETS[name] if name else None
            x.SetDeletedBits(bits if use_bits else None)
            x.SetTileShape(*tile)
            x.SortPoints(cam)
        for t in (rt_seq, rt):
            t.SetSceneDepth(depth if with_depth else None)
        seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq)
        lib.CalcViewData(cam); rt.Clear(); lib.Draw(cam, rt)
        assert np.array_equal(rt.Download(), rt_seq.Download()), (k, name)
        arr, n = shader_data_array(lib.m_Cutouts, lib.transform.localToWorldMatrix)
        want = orc.calc_view(lib.FrameParams(cam), arr, n, bits if use_bits else None)
        assert np.array_equal(lib.DownloadView().view(np.uint32), want.view(np.uint32)), (k, name)
        if tile != (0, 0):
            st = lib.FrameStats()
            assert (st.tile_w, st.tile_h) == tile
    for x in (lib, seq):
        x.OnDisable()
    rt.Dispose(); rt_seq.Dispose()

