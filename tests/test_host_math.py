"""The kernels' arithmetic header (csrc/gs_device_math.h) compiled for the HOST must agree with the oracle bit for bit.

This runs on the CPU box: it catches any divergence between the HIP kernels' op order and the oracle's canonical
arithmetic before a GPU is involved.  (The GPU tests then check the same thing through the C-ABI on the device.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from common import default_camera, small_asset
from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd._abi import VIEW_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", so,
                           os.path.join(HERE, "host_math_harness.cpp")])
    lib = C.CDLL(so)
    lib.hm_f16tof32.restype = C.c_float
    lib.hm_f16tof32.argtypes = [C.c_uint32]
    lib.hm_f32tof16.restype = C.c_uint32
    lib.hm_f32tof16.argtypes = [C.c_float]
    return lib


CASES = [("Medium", {}), ("High", {}), ("VeryHigh", {}),
         ("Medium", dict(formatPos=A.VectorFormat.Norm16, formatScale=A.VectorFormat.Norm6, formatSH=A.SHFormat.Float16, formatColor=A.ColorFormat.Float16x4)),
         ("Medium", dict(formatPos=A.VectorFormat.Norm6, formatScale=A.VectorFormat.Float32, formatSH=A.SHFormat.Norm11, formatColor=A.ColorFormat.Float32x4)),
         ("Medium", dict(formatSH=A.SHFormat.Cluster4k)),                                          # fp16 palette + u16 index in `other` (6-byte stride: unaligned dwords)
         ("Medium", dict(formatScale=A.VectorFormat.Norm6, formatSH=A.SHFormat.Cluster4k))]


@pytest.mark.parametrize("quality,fmt", CASES)
def test_keys_and_view_data_bit_exact(hm, quality, fmt):
    a = small_asset(6000, 7, quality, **fmt)
    cam = default_camera(az=33.0)
    tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695), scale=(1.0, 1.0, 1.0))
    orc = O.Oracle(a)
    ms = np.ascontiguousarray(camera.sort_matrix(cam, tr.localToWorldMatrix), np.float32).reshape(16)
    k0 = orc.calc_distances(ms).copy()
    k1 = np.zeros_like(k0)
    hm.hm_calc_distances(C.byref(orc.desc), orc.order.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), k1.ctypes.data_as(C.c_void_p))
    assert np.array_equal(k0, k1)
    for shOrder, shOnly, ss, os_ in [(3, False, 1.0, 1.0), (2, False, 0.7, 1.0), (1, True, 1.0, 3.0), (0, False, 2.0, 0.3)]:
        P = camera.frame_params(cam, tr, ss, os_, shOrder, shOnly)
        v0 = orc.calc_view(P).copy()
        v1 = np.zeros(a.splatCount, VIEW_DTYPE)
        hm.hm_calc_view(C.byref(orc.desc), C.byref(P), v1.ctypes.data_as(C.c_void_p))
        assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), (shOrder, shOnly)
        assert (v0["pos"][:, 3] > 0).sum() > a.splatCount // 2


def test_half_conversion_matches(hm):
    lib = O.lib()
    rng = np.random.default_rng(1)
    f = (rng.standard_normal(5000) * 10.0 ** rng.integers(-9, 6, 5000)).astype(np.float32)
    for v in list(f) + [0.0, 65504.0, 65520.0, 5.96e-8, 2.98e-8, 1e-10, -3e-5]:
        assert hm.hm_f32tof16(C.c_float(float(v))) == lib.gso_f32tof16(C.c_float(float(v)))
    for h in range(0, 0x7c00, 37):
        assert hm.hm_f16tof32(h) == lib.gso_f16tof32(C.c_uint16(h))


def test_footprints_match_oracle_prepare(hm):
    a = small_asset(6000, 7, "Medium")
    cam = default_camera(W=333, H=217, az=12.0)
    tr = camera.Transform()
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
    v = orc.calc_view(P)
    orc.tile = (16, 16)                                  # the harness reports the header's pixel rectangles as 16x16 tiles
    orc.draw(P)
    out = np.zeros((a.splatCount, 5), np.int32)
    cxy = np.zeros((a.splatCount, 2), np.float32)
    tiles = np.zeros(a.splatCount, np.uint32)
    hm.hm_prepare_ex(v.ctypes.data_as(C.c_void_p), C.c_uint32(a.splatCount), C.byref(P), out.ctypes.data_as(C.c_void_p), cxy.ctypes.data_as(C.c_void_p),
                     tiles.ctypes.data_as(C.c_void_p))
    w = np.maximum(out[:, 1] - out[:, 0] + 1, 0).astype(np.int64)
    h = np.maximum(out[:, 3] - out[:, 2] + 1, 0).astype(np.int64)
    assert int(tiles.sum()) == orc.tile_pairs
    assert int((tiles > 0).sum()) == orc.visible
    assert (tiles == w * h).all()


def _live_pixels(v, P, i):
    """Pixel-exact liveness of splat i (the fragment test of the blend kernel / oracle), fp32, over its quad bounding box."""
    f = np.float32
    W, H = int(P.screen_w), int(P.screen_h)
    w = v["pos"][i, 3]
    a1, a2 = v["axis1"][i], v["axis2"][i]
    al = np.uint16(v["color"][i, 1] & 0xffff).view(np.float16).astype(f)
    cx = f(f(f(0.5) * f(v["pos"][i, 0] * f(f(1) / w))) * f(W) + f(0.5 * W))
    cy = f(f(f(-0.5) * f(v["pos"][i, 1] * f(f(1) / w))) * f(H) + f(0.5 * H))
    u1 = a1 * f(f(1) / (a1 * a1).sum(dtype=f)); u2 = a2 * f(f(1) / (a2 * a2).sum(dtype=f))
    ex = f(2) * (abs(a1[0]) + abs(a2[0])) + f(1); ey = f(2) * (abs(a1[1]) + abs(a2[1])) + f(1)
    x0, x1 = max(int(np.floor(cx - ex)), 0), min(int(np.ceil(cx + ex)), W - 1)
    y0, y1 = max(int(np.floor(cy - ey)), 0), min(int(np.ceil(cy + ey)), H - 1)
    if x0 > x1 or y0 > y1:
        return None
    xs = (np.arange(x0, x1 + 1, dtype=f) + f(0.5))[None, :] - cx
    ys = (np.arange(y0, y1 + 1, dtype=f) + f(0.5))[:, None] - cy
    q1 = xs * u1[0] + ys * u1[1]; q2 = xs * u2[0] + ys * u2[1]
    alpha = np.clip(np.exp(-(q1 * q1 + q2 * q2), dtype=f) * al, 0, 1)
    live = (np.abs(q1) <= 2) & (np.abs(q2) <= 2) & (alpha >= f(1 / 255))
    return x0, y0, live, (cx, cy, u1, u2, al)


def test_block_culling_never_drops_a_live_fragment(hm):
    """BlockMayTouch (the 8x8-quadrant test of the blend kernel's ballot; also checked at 16x16) is conservative w.r.t. the
    pixel-exact fragment test, and PrepareSplat's tile rectangle contains every tile that owns a live pixel."""
    hm.hm_block_may_touch.argtypes = [C.c_float] * 10
    hm.hm_log_det.restype = C.c_float
    hm.hm_log_det.argtypes = [C.c_float]
    for x in [1.0, 1.0001, 1.5, 2.0, 2.7182817, 10.0, 255.0, 16575000.0, 0.999]:
        assert abs(hm.hm_log_det(x) - np.log(np.float64(np.float32(x)))) < 2e-6
    a = small_asset(20000, 7, "Medium")
    cam = default_camera(W=333, H=217, az=12.0)
    tr = camera.Transform()
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    v = orc.calc_view(P)
    n = a.splatCount
    out = np.zeros((n, 5), np.int32); cxy = np.zeros((n, 2), np.float32); tiles = np.zeros(n, np.uint32)
    vp = v.ctypes.data_as(C.c_void_p)
    hm.hm_prepare_ex(vp, C.c_uint32(n), C.byref(P), out.ctypes.data_as(C.c_void_p), cxy.ctypes.data_as(C.c_void_p), tiles.ctypes.data_as(C.c_void_p))
    checked_tiles = dropped_tiles = checked_quads = dropped_quads = 0
    for i in np.flatnonzero(out[:, 4] == 1)[:4000]:
        lp = _live_pixels(v, P, i)
        if lp is None:
            continue
        x0, y0, live, (cx, cy, u1, u2, al) = lp
        ys, xs = np.nonzero(live)
        tx0, tx1, ty0, ty1 = out[i, :4]
        if len(xs) == 0:
            continue
        assert tx0 <= tx1, "a splat with live pixels has an empty footprint"
        r2 = np.float32(np.float32(hm.hm_log_det(np.float32(255.0) * al)) * np.float32(1.0001) + np.float32(1e-3))
        live_tiles = set(zip(((xs + x0) >> 4).tolist(), ((ys + y0) >> 4).tolist()))
        for (tx, ty) in live_tiles:
            assert tx0 <= tx <= tx1 and ty0 <= ty <= ty1
        for ty in range(ty0, ty1 + 1):
            for tx in range(tx0, tx1 + 1):
                hit = hm.hm_block_may_touch(tx * 16 + 8.0, ty * 16 + 8.0, 7.5, cx, cy, u1[0], u1[1], u2[0], u2[1], r2)
                checked_tiles += 1
                dropped_tiles += 0 if hit else 1
                if (tx, ty) in live_tiles:
                    assert hit, f"splat {i}: tile ({tx},{ty}) has a live pixel but fails the block test"
        live_quads = set(zip(((xs + x0) >> 3).tolist(), ((ys + y0) >> 3).tolist()))
        for qy in range((y0 >> 3), ((y0 + live.shape[0] - 1) >> 3) + 1):
            for qx in range((x0 >> 3), ((x0 + live.shape[1] - 1) >> 3) + 1):
                hit = hm.hm_block_may_touch(qx * 8 + 4.0, qy * 8 + 4.0, 3.5, cx, cy, u1[0], u1[1], u2[0], u2[1], r2)
                checked_quads += 1
                dropped_quads += 0 if hit else 1
                if (qx, qy) in live_quads:
                    assert hit, f"splat {i}: quadrant ({qx},{qy}) has a live pixel but was culled"
    assert checked_tiles > 3000 and dropped_tiles > 0.05 * checked_tiles
    assert checked_quads > 10000 and dropped_quads > 0.1 * checked_quads


@pytest.mark.parametrize("quality", ["Medium", "VeryHigh"])
def test_early_cull_never_drops_a_drawable_splat(hm, quality):
    """CalcViewGeom(allowCull) -- the per-frame path -- may only give up on splats whose full-path footprint is empty, and
    must leave the geometry of every other splat bit-identical.  Cameras inside the cloud, grazing, zoomed in (huge
    footprints), zoomed out, with splat scale 0.1..2 and a scaled/rotated object."""
    a = small_asset(40_000, 11, quality, extent=4.0)
    orc = O.Oracle(a)
    culled_total = drawable_total = 0
    cases = [((0.0, 0.0, 6.0), (0, 0, 0), 39.1, 1.0, (1, 1, 1)), ((0.3, 0.2, 0.5), (2.0, 0.1, -1.0), 60.0, 2.0, (1, 1, 1)),
             ((0.0, 0.0, 1.2), (0, 0, 0), 10.0, 2.0, (1.5, 0.7, 1.0)), ((5.0, 4.0, 5.0), (0, 0, 0), 75.0, 0.1, (1, 1, 1)),
             ((0.05, 3.9, 0.0), (0, 0, 0), 90.0, 1.0, (1, 1, 1)), ((-2.0, 0.5, 2.5), (3.0, 0.5, 2.4), 25.0, 1.3, (0.5, 0.5, 2.0))]
    for k, (eye, target, fov, ss, objscale) in enumerate(cases):
        cam = camera.Camera(position=eye, target=target, fieldOfView=fov, pixelWidth=333 + 64 * k, pixelHeight=217 + 31 * k,
                            nearClipPlane=0.3, farClipPlane=(8.0 if k == 3 else 1000.0))
        tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695), scale=objscale)
        P = camera.frame_params(cam, tr, ss, 1.0, 3, False)
        out = np.zeros((a.splatCount, 2), np.uint32)
        hm.hm_cull_check(C.byref(orc.desc), C.byref(P), out.ctypes.data_as(C.c_void_p))
        culled = (out[:, 0] & 1).astype(bool)
        assert not (out[:, 0] & 2).any(), "geometry of an unculled splat differs from the full path"
        assert not (culled & (out[:, 1] > 0)).any(), f"case {k}: {int((culled & (out[:, 1] > 0)).sum())} drawable splats were culled"
        culled_total += int(culled.sum()); drawable_total += int((out[:, 1] > 0).sum())
    assert culled_total > 20_000 and drawable_total > 20_000          # both outcomes are exercised


def test_chunk_cull_never_drops_a_drawable_splat(hm):
    """ChunkOutside (whole 256-splat chunks skipped by the per-frame calc_view) may only fire for chunks in which no splat
    has a non-empty footprint in the full path; and it must fire often enough to matter.  Same adversarial cameras as the
    per-splat early cull, plus an orthographic-like projection for which the cull must switch itself off."""
    hm.hm_chunk_cull.restype = C.c_uint32
    a = small_asset(60_000, 11, "Medium", extent=4.0)
    orc = O.Oracle(a)
    nch = (a.splatCount + 255) // 256
    culled_chunks = drawable_total = 0
    cases = [((0.0, 0.0, 6.0), (0, 0, 0), 39.1, 1.0, (1, 1, 1)), ((0.3, 0.2, 0.5), (2.0, 0.1, -1.0), 60.0, 2.0, (1, 1, 1)),
             ((0.0, 0.0, 1.2), (0, 0, 0), 10.0, 2.0, (1.5, 0.7, 1.0)), ((5.0, 4.0, 5.0), (0, 0, 0), 75.0, 0.1, (1, 1, 1)),
             ((0.05, 3.9, 0.0), (0, 0, 0), 90.0, 1.0, (1, 1, 1)), ((-2.0, 0.5, 2.5), (3.0, 0.5, 2.4), 25.0, 1.3, (0.5, 0.5, 2.0)),
             ((9.0, 1.0, 0.0), (20.0, 1.0, 0.0), 39.1, 1.0, (1, 1, 1))]                       # looking away from the scene
    for k, (eye, target, fov, ss, objscale) in enumerate(cases):
        cam = camera.Camera(position=eye, target=target, fieldOfView=fov, pixelWidth=333 + 64 * k, pixelHeight=217 + 31 * k,
                            nearClipPlane=0.3, farClipPlane=(8.0 if k == 3 else 1000.0))
        tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695), scale=objscale)
        P = camera.frame_params(cam, tr, ss, 1.0, 3, False)
        out = np.zeros((a.splatCount, 2), np.uint32)
        hm.hm_cull_check(C.byref(orc.desc), C.byref(P), out.ctypes.data_as(C.c_void_p))
        cc = np.zeros(nch, np.uint8)
        assert hm.hm_chunk_cull(C.byref(orc.desc), C.byref(P), cc.ctypes.data_as(C.c_void_p)) == 1
        per_splat = np.repeat(cc, 256)[:a.splatCount].astype(bool)
        bad = per_splat & (out[:, 1] > 0)
        assert not bad.any(), f"case {k}: {int(bad.sum())} drawable splats sit in culled chunks"
        culled_chunks += int(cc.sum()); drawable_total += int((out[:, 1] > 0).sum())
    assert culled_chunks > 0.3 * nch * len(cases) / 2 and drawable_total > 20_000
    # a projection whose clip.w is not the view depth: the cull must disable itself
    P = camera.frame_params(camera.Camera(), camera.Transform())
    P.matrix_vp[12:16] = [0.0, 0.0, 0.0, 1.0]
    cc = np.zeros(nch, np.uint8)
    assert hm.hm_chunk_cull(C.byref(orc.desc), C.byref(P), cc.ctypes.data_as(C.c_void_p)) == 0 and not cc.any()


def test_discard_decision_is_identical_whatever_the_exp2_unit_rounds(hm):
    """The fragment's alpha near 1/255: the header's DecideAlpha (what the blend kernel runs) fed with the correctly rounded native
    alpha AND with that alpha +-1 ulp (any exp2 unit within 1 ulp) takes the oracle's decision every time, and inside the window the
    alpha is the oracle's bit for bit; Exp2Det is within 1 ulp of 2^y and identical on both sides by construction (fp32 operations)."""
    import ctypes as C
    hm.hm_exp2_det.restype = C.c_float; hm.hm_exp2_det.argtypes = [C.c_float]
    hm.hm_decide_alpha.restype = C.c_float; hm.hm_decide_alpha.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]
    rng = np.random.default_rng(4)
    ys = rng.uniform(-30.0, 0.0, 4000).astype(np.float32)
    got = np.array([hm.hm_exp2_det(C.c_float(float(y))) for y in ys], np.float32)
    want = np.exp2(ys.astype(np.float64))
    assert (np.abs(got.astype(np.float64) - want) / want).max() < 1.2e-7
    L = O.lib()
    out = np.zeros(4, np.float32)
    T = np.float32(1.0) / np.float32(255.0)
    n_in_window = 0
    for _ in range(6000):
        q = rng.uniform(-2.0, 2.0, 2).astype(np.float32)
        power = np.float32(-(np.float32(q[1] * q[1]) + np.float32(q[0]) * np.float32(q[0])))          # not the fused form: only used to aim near the threshold
        a = np.float32(min(1.0, float(np.exp(-float(power))) / 255.0 * (1.0 + rng.uniform(-2e-6, 2e-6))))
        col = np.array([0.3, 0.5, 0.7, a], np.float32)
        dead = L.gso_fragment(q.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int32(1))
        # the oracle's own native alpha and y (canonical forms), then the header's decision with the native alpha nudged by -1, 0, +1 ulp
        L.gso_fragment_native.restype = C.c_float
        y = C.c_float()
        native = np.float32(L.gso_fragment_native(q.ctypes.data_as(C.c_void_p), C.c_float(float(a)), C.byref(y)))
        for nudge in (-1, 0, 1):
            an = np.uint32(int(native.view(np.uint32)) + nudge).view(np.float32) if native > 0 else native
            live = C.c_int32()
            alpha = np.float32(hm.hm_decide_alpha(C.c_float(float(an)), y, C.c_float(float(a)), C.byref(live)))
            assert (live.value == 0) == bool(dead), (q, a, nudge)
            u, u0 = int(an.view(np.uint32)) - 0x3B808079, int(native.view(np.uint32)) - 0x3B808079
            if 0 <= u < 16 and 0 <= u0 < 16:                  # both sides inside the window: the same deterministic alpha
                n_in_window += 1
                if not dead:
                    assert alpha.view(np.uint32) == out[3].view(np.uint32)
            elif not dead:                                     # one side at the window's edge: a few ulps apart, same decision (above)
                assert abs(int(alpha.view(np.uint32)) - int(out[3].view(np.uint32))) <= 4
    assert n_in_window > 500


def test_axis_length_guard_equals_the_division_it_replaces():
    """PrepareSplat's `1 / |axis|^2 is finite` as an integer range test on the bits of d = |axis|^2 (gs_device_math.h): equal to the
    IEEE division for every class of d a sum of two squares can take -- zero, denormals around 2^-128 (where 1/d starts to
    overflow), normals, +inf, NaN."""
    edge = np.array([0x00000000, 0x00000001, 0x001fffff, 0x00200000, 0x00200001, 0x00200002, 0x007fffff, 0x00800000, 0x3f800000,
                     0x7f7fffff, 0x7f800000, 0x7f800001, 0x7fc00000, 0x7fffffff], np.uint32)
    rnd = np.random.default_rng(2).integers(0, 0x7fffffff, 200000, dtype=np.uint64).astype(np.uint32)
    bits = np.concatenate([edge, rnd, np.arange(0x001ffff0, 0x00200010, dtype=np.uint32)])
    d = bits.view(np.float32)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        want = np.isfinite(np.float32(1.0) / d)
    got = (bits - np.uint32(0x00200001)) <= np.uint32(0x7f800000 - 0x00200001)
    assert np.array_equal(got, want)
