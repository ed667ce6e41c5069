"""-m gpu: the compositor (bin + pair sort + blend) and the resolve through the C-ABI vs the oracle.

Tolerances: the only non-bit-exact operation is exp2 (v_exp_f32 vs the correctly rounded exp2 of the same
fp32 argument, <= 1 ulp), which can flip an fp16 rounding of the accumulated colour; the discard decision at alpha = 1/255 is
identical on both sides.  exact mode: tests/common.py rt_err <= 2^-9 for EVERY pixel (relative to max(1, |c|); the plain max-abs is
asserted next to it; > 99.99 % of the pixels bit-equal; an outlier allowance exists only as an opt-in, used by the C3 full-size frame
alone); fast mode: <= 4e-3.  Resolved 8-bit image: PSNR >= 50 dB and no pixel off by >= 3/255 (validator metric,
GaussianSplatValidator.cs:159-208)."""
import numpy as np
import pytest

import oracle_lib as O
from common import RT_TOL, default_camera, diff_pixels, psnr8, rt_abs, rt_err, small_asset
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GaussianSplatRenderSystem, RenderTarget

pytestmark = pytest.mark.gpu


def render_both(gpu_ctx, a, cam, mode=0, tr=None, **fields):
    r = GaussianSplatRenderer(gpu_ctx, a, tr)
    for k, v in fields.items():
        setattr(r, k, v)
    r.OnEnable()
    r.blendMode = mode
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    r.SortPoints(cam)
    r.CalcViewData(cam)
    rt.Clear()
    r.Draw(cam, rt)
    st = r.FrameStats()
    img = rt.Download()
    o32, o8 = rt.Resolve((0.1, 0.2, 0.3, 1.0))
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    orc.calc_view(P)
    ref = orc.draw(P, mode)
    r32, r8 = O.resolve(ref, (0.1, 0.2, 0.3, 1.0))
    pairs = orc.pairs(P, st)                 # the oracle's (tile, splat) count for the tile shape the draw used
    r.OnDisable()
    rt.Dispose()
    return dict(img=img, ref=ref, o32=o32, o8=o8, r32=r32, r8=r8, st=st, orc=orc, pairs=pairs)


@pytest.mark.parametrize("W,H", [(640, 360), (333, 217), (16, 16), (17, 9), (1920, 1080)])
@pytest.mark.parametrize("mode", [0, 1])
def test_framebuffer_parity(gpu_ctx, W, H, mode):
    a = small_asset(60_000, 5, "Medium")
    res = render_both(gpu_ctx, a, default_camera(W=W, H=H, az=40.0), mode)
    e = rt_err(res["img"], res["ref"])
    assert e <= (RT_TOL if mode == 0 else 4e-3), e
    if mode == 0:                                   # SURVEY.md section 8c's wording: plain max-abs <= 2^-9 (no value of this scene exceeds 2: same bar)
        assert rt_abs(res["img"], res["ref"]) <= RT_TOL, rt_abs(res["img"], res["ref"])
    assert res["st"].tile_pairs == res["pairs"] and res["st"].visible_splats == res["orc"].visible
    assert (res["img"] == res["ref"]).all(axis=2).mean() > 0.995
    assert psnr8(res["o8"], res["r8"]) >= 50.0 and diff_pixels(res["o8"], res["r8"]) == 0
    # resolved float image = lerp(bg, GammaToLinearSpace(C/A), A): the un-premultiply divides the target's tolerance by A,
    # so the bound is the target's 2^-8 over the smallest A that matters (0.5) -- the 8-bit validator metric above is the bar
    assert np.abs(res["o32"] - res["r32"]).max() <= 2.0 ** -7
    tw, th = res["st"].tile_w, res["st"].tile_h
    assert (tw, th) in ((16, 16), (32, 16), (32, 32)) and res["st"].tiles_x == (W + tw - 1) // tw and res["st"].tiles_y == (H + th - 1) // th


@pytest.mark.parametrize("W,H", [(640, 360), (333, 217), (33, 17)])
@pytest.mark.parametrize("mode", [0, 1])
def test_tile_shapes_give_the_same_frame(gpu_ctx, W, H, mode):
    """The compositor tile (16x16, 32x16, 32x32 pixels: gs_renderer_set_tile_shape) is a performance parameter: every pixel blends the
    same splats in the same order whatever tile it belongs to, so the frames are BIT-identical across shapes (both blend modes: the
    fast mode's per-pixel early stop does not depend on the tile either); only the (tile, splat) pair count changes, and it equals
    the oracle's count for that shape."""
    a = small_asset(60_000, 5, "Medium")
    cam = default_camera(W=W, H=H, az=40.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    r.blendMode = mode
    rt = RenderTarget(gpu_ctx, W, H)
    r.SortPoints(cam)
    r.CalcViewData(cam)                               # once: the per-splat pixel rectangles do not depend on the tile shape
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    orc.calc_view(P)
    frames, pairs = [], []
    ref = orc.draw(P, mode)
    for shape in ((16, 16), (32, 16), (32, 32), (0, 0)):
        r.SetTileShape(*shape)
        for _ in range(2):                            # the second draw runs with the schedule made from the first one's tile costs
            rt.Clear()
            r.Draw(cam, rt)
            st = r.FrameStats()
            frames.append(rt.Download())
            want = shape if shape != (0, 0) else r.TileShape(W, H)
            assert (st.tile_w, st.tile_h) == want and st.tiles_x == (W + want[0] - 1) // want[0] and st.tiles_y == (H + want[1] - 1) // want[1]
            assert st.tile_pairs == orc.pairs(P, want) and st.visible_splats == orc.visible
            pairs.append(int(st.tile_pairs))
    assert rt_err(frames[0], ref) <= (RT_TOL if mode == 0 else 4e-3)
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
    assert pairs[0] >= pairs[2] >= pairs[4]           # larger tiles, fewer pairs
    with pytest.raises(GsError):
        r.SetTileShape(64, 64)
    r.OnDisable()
    rt.Dispose()


@pytest.mark.parametrize("quality", ["High", "VeryHigh"])
def test_framebuffer_parity_other_formats(gpu_ctx, quality):
    a = small_asset(40_000, 8, quality)
    res = render_both(gpu_ctx, a, default_camera(W=500, H=300, az=-30.0), 0, m_SplatScale=1.5, m_OpacityScale=0.7, m_SHOrder=2)
    assert rt_err(res["img"], res["ref"]) <= RT_TOL
    assert res["st"].tile_pairs == res["pairs"]


def test_everything_culled_and_single_splat(gpu_ctx):
    from test_oracle import fp32_point_asset
    a = fp32_point_asset([[0.3, 0.1, 0.0]])
    cam = camera.Camera(position=(0, 0, 5), target=(0, 0, 0), pixelWidth=128, pixelHeight=96)
    res = render_both(gpu_ctx, a, cam)
    assert res["st"].visible_splats == 1 and res["st"].tile_pairs == res["pairs"] >= 1
    assert np.abs(O.f16_to_f32(res["img"]) - O.f16_to_f32(res["ref"])).max() <= 2.0 ** -10
    back = camera.Camera(position=(0, 0, 5), target=(0, 0, 10), pixelWidth=128, pixelHeight=96)      # looking away
    res = render_both(gpu_ctx, a, back)
    assert res["st"].visible_splats == 0 and res["st"].tile_pairs == 0 and not res["img"].any()
    assert np.allclose(res["o32"], [0.1, 0.2, 0.3, 1.0])


def test_pair_buffer_overflow_is_reported_and_recovered(gpu_ctx):
    # 4000 huge splats at 1080p need far more than the initial 4M pairs: the frame reports GS_ERR_PAIR_OVERFLOW, the
    # buffer is grown, and the re-drawn frame is correct
    from unitygaussiansplatting_amd import creator, scenes
    raw = scenes.make_splats(4000, 2, 1.0, logscale_mu=-1.0, logscale_sigma=0.2)
    a = creator.CreateAssetFromSplats(raw, "Medium")
    cam = default_camera(W=1920, H=1080, az=0.0, radius=3.0)
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    rt = RenderTarget(gpu_ctx, 1920, 1080)
    r.SetTileShape(16, 16)                       # the premise (more than the initial 4 M pairs) is a 16x16-tile count
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    gpu_ctx.Synchronize()
    pairs, cap = r.PollPairs()                   # the non-blocking report: this draw was truncated
    assert pairs > cap == (1 << 22)
    with pytest.raises(GsError) as e:
        r.FrameStats()
    assert e.value.code == -6
    rt.Clear(); r.Draw(cam, rt)
    st = r.FrameStats()
    assert st.tile_pairs > (1 << 22) and st.pair_capacity >= st.tile_pairs
    orc = O.Oracle(a)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    P = r.FrameParams(cam)
    orc.calc_view(P)
    ref = orc.draw(P, 0)
    assert st.tile_pairs == orc.pairs(P, st)
    assert rt_err(rt.Download(), ref) <= RT_TOL
    r.OnDisable()


def test_two_objects_render_order_and_accumulating_draw(gpu_ctx):
    # GatherSplatsForCamera (GaussianSplatRenderer.cs:89-102): higher m_RenderOrder first, then nearer first; each object's
    # draw blends UNDER what is already in the target
    a1, a2 = small_asset(8000, 31, "Medium"), small_asset(8000, 32, "Medium")
    sysm = GaussianSplatRenderSystem()
    r1 = GaussianSplatRenderer(gpu_ctx, a1, camera.Transform(position=(0.0, 0.0, 1.5)))
    r2 = GaussianSplatRenderer(gpu_ctx, a2, camera.Transform(position=(0.0, 0.0, -1.5)))
    for r in (r1, r2):
        r.OnEnable()
        GaussianSplatRenderSystem.instance.UnregisterSplat(r)
        sysm.RegisterSplat(r)
    cam = default_camera(W=320, H=200, az=0.0)
    rt = RenderTarget(gpu_ctx, 320, 200)

    def oracle_frame(order):
        ref = np.zeros((200, 320, 4), np.uint16)
        for r, a in order:
            orc = O.Oracle(a)
            orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
            P = r.FrameParams(cam)
            orc.calc_view(P)
            orc.draw(P, 0, ref)
        return ref
    sysm.OnPreCullCamera(cam, rt)
    assert [g for g in sysm.m_ActiveSplats] == [r1, r2]              # r1 is nearer to the camera at z=+6
    img = rt.Download()
    assert rt_err(img, oracle_frame([(r1, a1), (r2, a2)])) <= RT_TOL
    r2.m_RenderOrder = 5                                              # now r2 is drawn first (on top)
    sysm.OnPreCullCamera(cam, rt)
    assert sysm.m_ActiveSplats == [r2, r1]
    img2 = rt.Download()
    assert rt_err(img2, oracle_frame([(r2, a2), (r1, a1)])) <= RT_TOL
    assert (img != img2).any()
    for r in (r1, r2):
        r.OnDisable()


def test_sort_every_nth_frame(gpu_ctx):
    a = small_asset(20_000, 6, "Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    sysm = GaussianSplatRenderSystem()
    GaussianSplatRenderSystem.instance.UnregisterSplat(r)
    sysm.RegisterSplat(r)
    r.m_SortNthFrame = 3
    rt = RenderTarget(gpu_ctx, 160, 100)
    orders = []
    for f in range(4):
        sysm.OnPreCullCamera(default_camera(W=160, H=100, az=60.0 * f), rt)
        orders.append(r.DownloadOrder())
    assert np.array_equal(orders[0], orders[1]) and np.array_equal(orders[1], orders[2])     # frames 1,2 reuse frame 0's order
    assert not np.array_equal(orders[2], orders[3])                                          # frame 3 re-sorts
    r.OnDisable()


def test_determinism_run_twice(gpu_ctx):
    a = small_asset(60_000, 5, "Medium")
    cam = default_camera(W=640, H=360, az=75.0)
    x = render_both(gpu_ctx, a, cam)
    y = render_both(gpu_ctx, a, cam)
    assert np.array_equal(x["img"], y["img"]) and np.array_equal(x["o8"], y["o8"])


def test_overlapped_sort_queue_gives_identical_frames(gpu_ctx):
    """gs_context_set_overlap(1): SortPoints runs on the context's second queue concurrently with CalcViewData and is joined
    by the draw / by any readback.  Orders, keys and frames must equal the serial ones bit for bit, over several frames
    (stateful previous order), including a readback issued while the sort is still un-joined."""
    a = small_asset(120_000, 5, "Medium")
    frames = {}
    for overlap in (False, True):
        gpu_ctx.SetOverlap(overlap)
        r = GaussianSplatRenderer(gpu_ctx, a)
        r.OnEnable()
        rt = RenderTarget(gpu_ctx, 640, 360)
        out = []
        for f in range(4):
            cam = default_camera(W=640, H=360, az=30.0 + 25.0 * f)
            r.SortPoints(cam)
            if f == 2:
                out.append(r.DownloadOrder().copy())          # readback before the join point
            r.CalcViewData(cam)
            rt.Clear()
            r.Draw(cam, rt)
            out += [rt.Download().copy(), r.DownloadOrder().copy(), r.DownloadDistances().copy(), r.FrameStats().tile_pairs]
        frames[overlap] = out
        r.OnDisable()
        rt.Dispose()
    gpu_ctx.SetOverlap(False)
    for x, y in zip(frames[False], frames[True]):
        assert np.array_equal(x, y)


def test_overlapped_sort_with_frames_in_flight(gpu_ctx):
    """The pipelined form of the overlap: K frames are recorded back to back without a host sync, so frame k+1's depth sort
    (second queue, released by frame k's bin_emit through evOrderFree) runs beside frame k's pair sort / blend / resolve and
    frame k+1's calc_view.  Every frame goes to its own target; all targets, the final order / keys and P must equal the
    serial run's bit for bit -- a sort that overwrote order[] under a bin_emit still reading it, or a calc_view that
    overtook a blend, would show here.  1.5 M splats so that the kernels are long enough to overlap for real."""
    a = creator.CreateAssetFromSplatsNative(scenes.make_splats(1_500_000, 11, 3.0), "Medium", name="inflight")
    K = 6
    results = {}
    for overlap in (False, True):
        gpu_ctx.SetOverlap(overlap)
        r = GaussianSplatRenderer(gpu_ctx, a)
        r.OnEnable()
        rts = [RenderTarget(gpu_ctx, 960, 540) for _ in range(K)]
        cams = [default_camera(W=960, H=540, az=10.0 + 40.0 * f) for f in range(K)]
        prepared = [(r.SortMatrix(c), r.FrameParams(c)) for c in cams]
        for rep in range(2):                                   # the first round sizes the pair buffer
            for f in range(K):
                m16, p = prepared[f]
                r.SortPointsPrepared(m16); r.CalcViewDataPrepared(p); rts[f].Clear(); r.DrawPrepared(p, rts[f]); rts[f].ResolveAsync((0, 0, 0, 1))
            st = r.FrameStats()
            if rep == 0:
                r.ReservePairs(int(st.tile_pairs * 2) + (1 << 20))
        results[overlap] = [t.Download().copy() for t in rts] + [r.DownloadOrder().copy(), r.DownloadDistances().copy(), st.tile_pairs]
        r.OnDisable()
        for t in rts:
            t.Dispose()
    gpu_ctx.SetOverlap(False)
    for x, y in zip(results[False], results[True]):
        assert np.array_equal(x, y)
    assert np.abs(results[True][0].view(np.float16).astype(np.float32)).sum() > 0


def test_tile_schedule_survives_changing_targets(gpu_ctx):
    """The blend's tile schedule of a draw is made by the extra workgroup of that draw's bin_emit from the costs the PREVIOUS draw's
    blend left, and those are only a hint for the same tile count; a renderer that alternates between targets of different
    sizes (and between splat and debug-box draws, which leave no costs) must fall back to the schedule kernel and still draw
    every tile exactly once: every frame equals the frame a fresh renderer draws."""
    a = small_asset(60_000, 9, "Medium")
    sizes = [(640, 360), (320, 200), (640, 360), (640, 360), (1280, 720), (320, 200), (320, 200)]
    def fresh(W, H, az):
        r = GaussianSplatRenderer(gpu_ctx, a); r.OnEnable()
        rt = RenderTarget(gpu_ctx, W, H)
        cam = default_camera(W=W, H=H, az=az)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        img = rt.Download().copy()
        r.OnDisable(); rt.Dispose()
        return img
    r = GaussianSplatRenderer(gpu_ctx, a); r.OnEnable()
    rts = {}
    for f, (W, H) in enumerate(sizes):
        rt = rts.setdefault((W, H), RenderTarget(gpu_ctx, W, H))
        cam = default_camera(W=W, H=H, az=20.0 + 15.0 * f)
        r.ResetOrder()                                          # same starting order as the fresh renderer (ties)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        assert np.array_equal(rt.Download(), fresh(W, H, 20.0 + 15.0 * f)), f"frame {f} ({W}x{H})"
    r.OnDisable()
    for rt in rts.values():
        rt.Dispose()


def test_profiling_ring_reports_stage_and_frame_times(gpu_ctx):
    """gs_renderer_set_profiling + gs_renderer_frame_times / gs_renderer_stage_times: one GPU duration per profiled frame,
    stage means that add up to about a frame, and the ring resets after stage_times."""
    a = small_asset(120_000, 5, "Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    rt = RenderTarget(gpu_ctx, 640, 360)
    r.SetProfiling(6)                                   # a ring: the slot after the last recorded frame is being reused, so keep spare slots
    for f in range(4):
        cam = default_camera(W=640, H=360, az=30.0 + 5.0 * f)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    ft = r.FrameTimes()
    st = r.StageTimes()
    assert len(ft) == 4 and (ft > 0).all() and (ft < 50.0).all()
    parts = st.calc_distances_ms + st.sort_ms + st.calc_view_ms + st.bin_ms + st.pair_sort_ms + st.blend_ms
    assert st.frames == 4 and 0 < parts <= 1.05 * float(ft.mean()) + 0.05 and abs(st.total_ms - parts) < 1e-4
    assert st.onesweep_depth_ms > 0 and st.onesweep_pairs_ms > 0 and 1 <= st.onesweep_pair_launches <= 3
    assert len(r.FrameTimes()) == 0                     # stage_times reset the ring
    r.SetProfiling(0)
    r.OnDisable(); rt.Dispose()
