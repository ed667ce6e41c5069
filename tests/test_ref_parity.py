"""The oracle against the reference's own shader text (oracle/_ref, see tests/ref_lib.py and oracle/ref_build/).

HLSL leaves the evaluation of its intrinsics and the contraction of a*b+c to the shader compiler, so "the reference's result"
is a small family; oracle/_ref holds three members.  What is asserted:
  * everything that involves no rounding freedom is BIT-EQUAL in every build: Morton / texel addressing, sortable keys of a
    float, the packed-field decoders, the stable order given the keys;
  * the oracle's canonical arithmetic IS the fused g++ build bit for bit for: sort keys and order,
    every field of LoadSplatData in every format, the clip position and the packed colour of the 40-byte view record, the
    fragment's alpha / discard decision; and the oracle's alternative-reading switches (gso_set_canon(3)) reproduce the
    STRICT build's decoders bit for bit;
  * where the oracle's contraction choices differ from a compiler's (the 2x2 covariance: axes) the difference is bounded far
    inside DESIGN.md section 5.1's sensitivity figures, and no larger than the builds differ among themselves;
  * whole frames: the reference's vertex + fragment shader under D3D raster rules reproduce the oracle's frame within the
    framebuffer bar (2^-9), and its composite fragment the oracle's resolve."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R
from common import RT_TOL, rt_diff, small_asset, views_equal
from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import camera, cutout

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402
from test_golden import load as load_golden  # noqa: E402

FORMAT_CASES = [("Medium", {}), ("High", {}), ("VeryHigh", {}), ("Low", {}), ("VeryLow", {}),
                ("Medium", dict(formatPos=A.VectorFormat.Norm16, formatScale=A.VectorFormat.Norm6, formatSH=A.SHFormat.Float16, formatColor=A.ColorFormat.Float16x4)),
                ("Medium", dict(formatPos=A.VectorFormat.Norm6, formatScale=A.VectorFormat.Float32, formatSH=A.SHFormat.Norm11, formatColor=A.ColorFormat.Float32x4)),
                ("Medium", dict(formatPos=A.VectorFormat.Float32, formatScale=A.VectorFormat.Norm16, formatSH=A.SHFormat.Cluster16k))]
TR = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695))


@pytest.fixture(autouse=True)
def _canon0():
    O.lib().gso_set_canon(0)
    yield
    O.lib().gso_set_canon(0)


def ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


def cam_for(az=25.0, W=320, H=200):
    from common import default_camera
    return default_camera(W=W, H=H, az=az)


@pytest.mark.parametrize("which", R.BUILDS)
def test_integer_paths_are_bit_equal_in_every_build(which):
    L = R.lib(which)
    rng = np.random.default_rng(1)
    # FloatToSortableUint (SplatUtilities.compute:52-57) == the definition, incl. +-0, denormals, +-inf, NaN payloads
    f = np.concatenate([np.array([0.0, -0.0, 1e-45, -1e-45, np.inf, -np.inf, 1.0, -1.0], np.float32),
                        rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    fu = f.view(np.uint32)
    want = fu ^ np.where(fu >> 31, np.uint32(0xffffffff), np.uint32(0x80000000))
    got = np.array([L.gsr_cs_sortable_uint(C.c_float(float(v))) for v in f], np.uint32)
    nan = np.isnan(f)                                              # a NaN's payload does not survive the float argument of ctypes
    assert np.array_equal(got[~nan], want[~nan])
    # SplatIndexToPixelIndex == the oracle's, and EncodeMorton2D_16x16 inverts DecodeMorton2D_16x16
    xy_r, xy_o = np.zeros(2, np.uint32), np.zeros(2, np.uint32)
    for idx in list(range(0, 1024)) + [int(v) for v in rng.integers(0, 1 << 24, 2000)]:
        L.gsr_cs_pixel_index(C.c_uint32(idx), xy_r.ctypes.data_as(C.c_void_p))
        O.lib().gso_pixel_index(C.c_uint32(idx), xy_o.ctypes.data_as(C.c_void_p))
        assert np.array_equal(xy_r, xy_o)
        assert L.gsr_cs_encode_morton(C.c_uint32(int(xy_r[0]) & 15), C.c_uint32(int(xy_r[1]) & 15)) == (idx & 255)


@pytest.mark.parametrize("quality,fmt", FORMAT_CASES)
def test_canonical_arithmetic_is_the_fused_build_of_the_reference_text(quality, fmt):
    a = small_asset(30_011, 7, quality, **fmt)
    orc, ref = O.Oracle(a), R.Ref(a, "fused")
    assert np.array_equal(ref.decode_all().view(np.uint32), orc.decode_all().view(np.uint32)), "LoadSplatData differs"
    for (shOrder, shOnly, ss, osc, az) in [(3, False, 1.0, 1.0, 20.0), (2, False, 0.5, 1.0, 100.0), (1, True, 1.0, 4.0, 200.0), (0, False, 2.0, 0.25, 300.0)]:
        cam = cam_for(az, 640, 360)
        ms = camera.sort_matrix(cam, TR.localToWorldMatrix)
        orc.reset_order(); ref.set_indices()
        assert np.array_equal(ref.calc_distances(ms), orc.calc_distances(ms)), "CSCalcDistances keys differ"
        P = camera.frame_params(cam, TR, ss, osc, shOrder, shOnly)
        vo, vr = orc.calc_view(P), ref.calc_view(P)
        live = vo["pos"][:, 3] > 0
        assert live.sum() > 1000
        R.assert_view_is_the_fused_build(vr, vo)
        if ss == 1.0 and shOrder == 3:                      # another compiler's contraction choices: the same records up to the axes
            vc = R.Ref(a, "fused_clang").calc_view(P)
            assert np.array_equal(vc["pos"].view(np.uint32), vo["pos"].view(np.uint32))
            assert (vc["color"] != vo["color"]).any(axis=1).mean() <= 1e-3
            R.assert_axes_close(vc, vo, live)


@pytest.mark.parametrize("quality,fmt", FORMAT_CASES[:4] + FORMAT_CASES[5:])
def test_alternative_readings_are_the_strict_build(quality, fmt):
    """gso_set_canon(3) (unfused lerp, IEEE division: SURVEY.md appendix A/B) reproduces the strict build's decoders exactly."""
    a = small_asset(30_011, 7, quality, **fmt)
    orc, ref = O.Oracle(a), R.Ref(a, "strict")
    O.lib().gso_set_canon(3)
    do, dr = orc.decode_all(), ref.decode_all()
    cols = [0, 1, 2, 7, 8, 9, 10, 11, 12, 13] + list(range(14, 59))             # everything but the rotation (its dot is a mad chain in the oracle)
    assert np.array_equal(dr[:, cols].view(np.uint32), do[:, cols].view(np.uint32))
    assert ulps(dr[:, 3:7], do[:, 3:7]).max() <= 512 and np.abs(dr[:, 3:7] - do[:, 3:7]).max() <= 2e-6   # q.w = sqrt(1 - |xyz|^2) near 0


@pytest.mark.parametrize("which", R.BUILDS)
@pytest.mark.parametrize("quality", ["Medium", "High", "VeryHigh"])
def test_every_build_is_inside_the_sensitivity_bounds(which, quality):
    """DESIGN.md section 5.1 measured what the alternative readings move; the real alternative evaluations of the reference text
    must not move more: keys by a few ulp, few order entries, axes by << 0.05 px, colours by <= 1 half ulp, P by <= 2."""
    a = small_asset(30_011, 7, quality)
    orc, ref = O.Oracle(a), R.Ref(a, which)
    cam = cam_for(33.0, 640, 360)
    ms = camera.sort_matrix(cam, TR.localToWorldMatrix)
    ko, kr = orc.calc_distances(ms).astype(np.int64), ref.calc_distances(ms).astype(np.int64)
    assert np.abs(ko - kr).max() <= 8 and (ko != kr).mean() <= 0.5
    oo = O.sort_pairs(orc.keys, orc.order)[1]
    orr = O.sort_pairs(ref.keys, ref.order)[1]
    moved = oo != orr
    assert moved.mean() <= 5e-3                                                     # section 5.1: 30 of 100 k for the unfused lerp alone
    P = camera.frame_params(cam, TR)
    vo, vr = orc.calc_view(P), ref.calc_view(P)
    live = (vo["pos"][:, 3] > 0) & (vr["pos"][:, 3] > 0)
    assert (vo["pos"][:, 3] > 0).sum() - live.sum() <= 2
    assert np.abs(vr["pos"][live] - vo["pos"][live]).max() <= 1e-4 * np.abs(vo["pos"][live]).max()
    R.assert_axes_close(vr, vo, live)                                                 # section 5.1: 0.041 px for the unfused lerp
    h = lambda v, sh: O.f16_to_f32(((v >> sh) & 0xffff).astype(np.uint16))
    for word, sh in ((0, 16), (0, 0), (1, 16), (1, 0)):
        co, cr = h(vo["color"][live, word], sh), h(vr["color"][live, word], sh)
        assert np.abs(co - cr).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(co)).max() + 1e-7                # <= 1 half ulp
    # the members of the family differ among themselves as much as the oracle differs from them
    if which != "fused":
        v2 = R.Ref(a, "fused").calc_view(P)
        ok = live & (v2["pos"][:, 3] > 0)
        ax = lambda v: np.concatenate([v["axis1"][ok], v["axis2"][ok]], axis=1)
        spread = np.nanquantile(np.abs(ax(vr) - ax(v2)), 0.999)
        mine = np.nanquantile(np.abs(ax(vo) - ax(v2)), 0.999)
        assert mine <= max(4.0 * spread, 1e-4), (mine, spread)


def test_sort_order_from_the_reference_keys_is_the_oracles_on_the_golden_scenes():
    for name in sorted(G.CASES):
        z, a = load_golden(name)
        cam, tr = G.camera_for(name), G.transform_for(name)
        ref = R.Ref(a, "fused")
        ref.set_indices()
        keys = ref.calc_distances(camera.sort_matrix(cam, tr.localToWorldMatrix))
        k, order = O.sort_pairs(keys, ref.order)
        assert np.array_equal(k, z["keys"]) and np.array_equal(order, z["order"])


def test_cutouts_and_deleted_bits_through_the_reference_kernel():
    a = small_asset(20_000, 5, "Medium")
    cam = cam_for(40.0)
    P = camera.frame_params(cam, TR)
    cuts = [cutout.GaussianCutout(cutout.Type.Ellipsoid, False, camera.Transform(position=(0.5, 0.2, 0.0), scale=(1.5, 1.0, 2.0))),
            None,
            cutout.GaussianCutout(cutout.Type.Box, True, camera.Transform(position=(-0.5, 0.0, 0.3), rotation=(0.0, 0.3827, 0.0, 0.9239), scale=(1.0, 2.0, 1.0)))]
    arr, n = cutout.shader_data_array(cuts, TR.localToWorldMatrix)
    deleted = np.random.default_rng(3).integers(0, 2 ** 32, (a.splatCount + 31) // 32, dtype=np.uint64).astype(np.uint32) & np.uint32(0x11111111)
    orc, ref = O.Oracle(a), R.Ref(a, "fused")
    vo = orc.calc_view(P, arr, n, deleted).copy()
    vr = ref.calc_view(P, arr, n, deleted)
    R.assert_view_is_the_fused_build(vr, vo)
    cut = vo["pos"][:, 3] == 0
    assert 0.2 < cut.mean() < 0.95
    plain = orc.calc_view(P)
    assert (plain["pos"][:, 3] == 0).sum() == 0


@pytest.mark.parametrize("which", R.BUILDS)
def test_fragment_alpha_and_discard(which):
    """frag() of RenderGaussianSplats.shader vs the oracle's fragment: bit-equal alpha and the same discard decision in the fused
    builds (exp = exp2(x * log2 e)); <= 16 ulp and decisions that differ only within 2e-6 of 1/255 in the
    strict build (correctly rounded e^x: the fp32 product x * log2 e of the other reading is off by up to |x| 2^-24)."""
    rng = np.random.default_rng(9)
    n = 20000
    q = rng.uniform(-2.2, 2.2, (n, 2)).astype(np.float32)
    col = np.concatenate([rng.uniform(0, 1.5, (n, 3)), rng.uniform(0.0, 1.2, (n, 1))], axis=1).astype(np.float32)
    # a third of the fragments right at the discard threshold: alpha = exp(-|q|^2) * a ~ 1/255
    k = n // 3
    col[:k, 3] = (np.exp((q[:k].astype(np.float64) ** 2).sum(axis=1)) / 255.0 * (1.0 + rng.uniform(-3e-7, 3e-7, k))).astype(np.float32)
    L = O.lib()
    out_o, out_r = np.zeros(4, np.float32), np.zeros(4, np.float32)
    flips = worst = 0
    for i in range(n):
        do = L.gso_fragment(q[i].ctypes.data_as(C.c_void_p), col[i].ctypes.data_as(C.c_void_p), out_o.ctypes.data_as(C.c_void_p), C.c_int32(0))
        dr, out_r = R.fragment(which, q[i], col[i])
        if which != "strict":
            assert do == dr and np.array_equal(out_o.view(np.uint32), out_r.view(np.uint32)), (i, q[i], col[i])
        elif do != dr:
            flips += 1
            alpha = float(np.exp(-float((q[i].astype(np.float64) ** 2).sum())) * col[i, 3])
            assert abs(alpha * 255.0 - 1.0) <= 2e-6, (i, alpha)
        elif not do:
            worst = max(worst, int(ulps(out_o[3:], out_r[3:]).max()))
    # the oracle's DRAW takes the decision inside a 16-ulp window around 1/255 on a deterministic exp2 (so that the GPU takes the same
    # one): there, and only there, it may leave the reference arithmetic -- by a few ulps of alpha
    if which == "fused":
        out_w = np.zeros(4, np.float32)
        moved = 0
        for i in range(n):
            do = L.gso_fragment(q[i].ctypes.data_as(C.c_void_p), col[i].ctypes.data_as(C.c_void_p), out_o.ctypes.data_as(C.c_void_p), C.c_int32(0))
            dw = L.gso_fragment(q[i].ctypes.data_as(C.c_void_p), col[i].ctypes.data_as(C.c_void_p), out_w.ctypes.data_as(C.c_void_p), C.c_int32(1))
            if do != dw or not np.array_equal(out_o.view(np.uint32), out_w.view(np.uint32)):
                moved += 1
                alpha = float(np.exp(-float((q[i].astype(np.float64) ** 2).sum())) * col[i, 3])
                assert abs(alpha * 255.0 - 1.0) <= 1.5e-6, (i, alpha)
                if not do and not dw:
                    assert ulps(out_o[3:], out_w[3:]).max() <= 4
        assert 0 < moved <= k
    assert worst <= 16 and flips <= k // 4          # exp2(fl(x log2 e)) vs the correctly rounded e^x: |x| 2^-24 relative from the rounded product


@pytest.mark.parametrize("which", ["strict", "fused"])
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_golden_frames_through_the_reference_shaders(which, name):
    """Sort keys -> stable sort -> CSCalcViewData with the y-flipped projection Unity binds for a render texture -> vert + frag
    + Blend OneMinusDstAlpha One under D3D raster rules -> rows flipped back: the oracle's golden frame within the framebuffer
    bar, most pixels bit-equal; then the composite fragment + its blend state: the oracle's resolve."""
    z, a = load_golden(name)
    cam, tr = G.camera_for(name), G.transform_for(name)
    ref = R.Ref(a, which)
    ref.set_indices()
    keys = ref.calc_distances(camera.sort_matrix(cam, tr.localToWorldMatrix))
    _, ref.order = O.sort_pairs(keys, ref.order)
    P = camera.frame_params(cam, tr)
    ref.calc_view(R.flipped(P))
    W, H = cam.pixelWidth, cam.pixelHeight
    rt = ref.draw(W, H, P.near_clip, P.far_clip)[::-1].copy()
    e = rt_diff(rt, z["rt_exact"]).max(axis=-1)
    assert e.max() <= RT_TOL, e.max() / RT_TOL
    assert (e == 0).mean() >= 0.9
    bg = (0.1, 0.2, 0.3, 1.0)
    o32, _ = O.resolve(z["rt_exact"], bg)
    r32 = R.resolve(which, z["rt_exact"], bg)
    assert np.abs(r32 - o32).max() <= 2e-6


def test_frame_through_the_reference_shaders_larger_scene():
    a = small_asset(20_000, 5, "Medium")
    cam = cam_for(25.0)
    tr = camera.Transform()
    orc, ref = O.Oracle(a), R.Ref(a, "fused")
    ms = camera.sort_matrix(cam, tr.localToWorldMatrix)
    orc.sort(ms)
    ref.set_indices()
    _, ref.order = O.sort_pairs(ref.calc_distances(ms), ref.order)
    assert np.array_equal(ref.order, orc.order)
    P = camera.frame_params(cam, tr)
    orc.calc_view(P)
    want = orc.draw(P, 0)
    ref.calc_view(R.flipped(P))
    got = ref.draw(cam.pixelWidth, cam.pixelHeight, P.near_clip, P.far_clip)[::-1].copy()
    e = rt_diff(got, want).max(axis=-1)
    assert e.max() <= RT_TOL and (e == 0).mean() >= 0.95, (e.max() / RT_TOL, (e == 0).mean())


@pytest.mark.parametrize("key", ["C1", "C2", "C3", "C2d"])
def test_bench_scenes_are_pinned_too(key):
    """The workloads bench.py measures (scenes.CONFIGS: their scale / opacity / position distributions, presets, cameras and target
    sizes), 60 k splats of each: keys and the whole view record from the fused build of the reference's text are the oracle's bit for bit
    at the bench's own camera, a quarter turn on and from inside the cloud -- so the parity the bench lines report against the oracle is
    parity with the reference's arithmetic on that data, not only on tests/common.py's small assets."""
    from unitygaussiansplatting_amd import creator, scenes
    cfg = scenes.CONFIGS[key]
    a = creator.CreateAssetFromSplats(scenes.make_config_splats(cfg, 0 if key == "C1" else 60_000), cfg.quality, name=key)
    orc, ref = O.Oracle(a), R.Ref(a, "fused")
    assert np.array_equal(ref.decode_all().view(np.uint32), orc.decode_all().view(np.uint32)), "LoadSplatData differs"
    tr = camera.Transform()
    drawn = 0
    for az, radius in [(0.0, cfg.eye_radius), (90.25, cfg.eye_radius), (200.0, 0.15 * cfg.eye_radius)]:
        cam = camera.Camera(position=scenes.orbit_eye(radius, cfg.eye_elev_deg, az), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
        ms = camera.sort_matrix(cam, tr.localToWorldMatrix)
        orc.reset_order(); ref.set_indices()
        assert np.array_equal(ref.calc_distances(ms), orc.calc_distances(ms)), "CSCalcDistances keys differ"
        P = camera.frame_params(cam, tr)
        vo, vr = orc.calc_view(P), ref.calc_view(P)
        R.assert_view_is_the_fused_build(vr, vo)
        drawn += int((vo["pos"][:, 3] > 0).sum())
    assert drawn > 30_000


@pytest.mark.parametrize("key", ["C2", "C2d"])
def test_bench_scene_frame_through_the_reference_shaders(key):
    """A frame of a bench scene's sample (40 k splats, the bench camera, a quarter of the target's size) rasterised through the
    reference's own vert + frag: the oracle's frame within the framebuffer bar.  C2d's splats are large (17 tiles per visible splat at
    full size): thousands of blends per pixel, the case where a per-blend rounding difference would accumulate."""
    from unitygaussiansplatting_amd import creator, scenes
    cfg = scenes.CONFIGS[key]
    a = creator.CreateAssetFromSplats(scenes.make_config_splats(cfg, 40_000), cfg.quality, name=key)
    W, H = cfg.width // 4, cfg.height // 4
    cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.0), pixelWidth=W, pixelHeight=H, fieldOfView=cfg.fov_y)
    tr = camera.Transform()
    orc, ref = O.Oracle(a), R.Ref(a, "fused")
    ms = camera.sort_matrix(cam, tr.localToWorldMatrix)
    orc.sort(ms)
    ref.set_indices()
    _, ref.order = O.sort_pairs(ref.calc_distances(ms), ref.order)
    assert np.array_equal(ref.order, orc.order)
    P = camera.frame_params(cam, tr)
    orc.calc_view(P)
    want = orc.draw(P, 0)
    ref.calc_view(R.flipped(P))
    got = ref.draw(W, H, P.near_clip, P.far_clip)[::-1].copy()
    e = rt_diff(got, want).max(axis=-1)
    assert (want.view(np.uint16)[..., 3] != 0).mean() > 0.2                 # the frame is not empty
    assert e.max() <= RT_TOL and (e == 0).mean() >= 0.9, (e.max() / RT_TOL, (e == 0).mean())
