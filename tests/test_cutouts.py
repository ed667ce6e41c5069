"""Cutouts and deleted bits inside CSCalcViewData (SplatUtilities.compute:164-187, 204-220; GaussianCutout.cs:24-40).

CPU: known-answer tests of the oracle's IsSplatCut against the shader's decision table, and the kernels' arithmetic header
compiled for the host against the oracle, bit for bit.  GPU (-m gpu): the same through the C-ABI, plus a composite."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from common import RT_TOL, default_camera, rt_err, small_asset, views_equal
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd._abi import VIEW_DTYPE, gs_cutout
from unitygaussiansplatting_amd.cutout import GaussianCutout, Type, shader_data_array

HERE = os.path.dirname(os.path.abspath(__file__))


def _decoded_positions(orc):
    return orc.decode_all()[:, 0:3].astype(np.float64)


def _expected_cut(pos, cutouts, renderer_matrix):
    """IsSplatCut written independently in float64 numpy (decision table of SplatUtilities.compute:164-187)."""
    n = len(pos)
    decided = np.zeros(n, bool)
    result = np.zeros(n, bool)
    final = np.zeros(n, bool)
    for c in cutouts:
        if c is None or not c.isActiveAndEnabled:
            continue
        m = c.transform.worldToLocalMatrix.astype(np.float64) @ np.asarray(renderer_matrix, np.float64)
        q = pos @ m[:3, :3].T + m[:3, 3]
        inside = (q * q).sum(1) <= 1.0 if c.m_Type == Type.Ellipsoid else (np.abs(q) <= 1.0).all(1)
        take = inside & ~decided
        result[take] = c.m_Invert
        decided |= take
        if not c.m_Invert:
            final[~decided] = True
    return np.where(decided, result, final)


def _margin_ok(pos, cutouts, renderer_matrix, eps=1e-4):
    """Splats not within eps of any cutout surface (fp32 vs fp64 may legitimately disagree there)."""
    ok = np.ones(len(pos), bool)
    for c in cutouts:
        if c is None or not c.isActiveAndEnabled:
            continue
        m = c.transform.worldToLocalMatrix.astype(np.float64) @ np.asarray(renderer_matrix, np.float64)
        q = pos @ m[:3, :3].T + m[:3, 3]
        d = (q * q).sum(1) - 1.0 if c.m_Type == Type.Ellipsoid else np.abs(q).max(1) - 1.0
        ok &= np.abs(d) > eps
    return ok


CUTOUT_SETS = {
    "crop_box": [GaussianCutout(Type.Box, False, camera.Transform(position=(0.2, 0.0, -0.1), scale=(1.5, 1.0, 2.0)))],
    "hole_ellipsoid": [GaussianCutout(Type.Ellipsoid, True, camera.Transform(position=(0.5, 0.3, 0.0), rotation=(0.0, 0.3826834, 0.0, 0.9238795), scale=(1.2, 0.6, 0.9)))],
    "hole_then_crop": [GaussianCutout(Type.Ellipsoid, True, camera.Transform(scale=(0.8, 0.8, 0.8))),
                       None,
                       GaussianCutout(Type.Box, False, camera.Transform(scale=(2.0, 2.0, 2.0))),
                       GaussianCutout(Type.Box, False, camera.Transform(position=(3, 3, 3)), isActiveAndEnabled=False)],
    "crop_then_hole": [GaussianCutout(Type.Box, False, camera.Transform(scale=(2.0, 2.0, 2.0))),
                       GaussianCutout(Type.Ellipsoid, True, camera.Transform(scale=(0.8, 0.8, 0.8)))],
    "only_null": [None],
}


@pytest.mark.parametrize("name", list(CUTOUT_SETS))
def test_oracle_is_splat_cut_decision_table(name):
    cuts = CUTOUT_SETS[name]
    a = small_asset(20000, 5, "VeryHigh")
    tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695))
    cam = default_camera(az=40.0)
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    base = orc.calc_view(P).copy()
    arr, n = shader_data_array(cuts, tr.localToWorldMatrix)
    v = orc.calc_view(P, arr, n).copy()
    pos = _decoded_positions(orc)
    want_cut = _expected_cut(pos, cuts, tr.localToWorldMatrix)
    ok = _margin_ok(pos, cuts, tr.localToWorldMatrix)
    got_cut = (v["pos"][:, 3] == 0.0) & (base["pos"][:, 3] != 0.0)
    front = base["pos"][:, 3] > 0
    assert np.array_equal(got_cut[ok & front], want_cut[ok & front])
    if name != "only_null":
        assert 100 < want_cut[front].sum() < front.sum() - 100          # the case is not vacuous
    else:
        assert not want_cut.any()
    # a cut splat keeps clip.xyz, gets w = 0 and an all-zero remainder; an uncut splat is untouched
    cutm = v["pos"][:, 3] != base["pos"][:, 3]
    assert np.array_equal(v["pos"][cutm, :3], base["pos"][cutm, :3])
    assert (v["axis1"][cutm] == 0).all() and (v["axis2"][cutm] == 0).all() and (v["color"][cutm] == 0).all()
    assert np.array_equal(v[~cutm].view(np.uint32), base[~cutm].view(np.uint32))


def test_oracle_crop_then_hole_order_matters():
    """The first cutout containing the splat decides: inside the crop box first => kept even inside the later hole."""
    a = small_asset(20000, 5, "VeryHigh")
    tr = camera.Transform()
    cam = default_camera()
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    a1, n1 = shader_data_array(CUTOUT_SETS["hole_then_crop"], tr.localToWorldMatrix)
    a2, n2 = shader_data_array(CUTOUT_SETS["crop_then_hole"], tr.localToWorldMatrix)
    w1 = orc.calc_view(P, a1, n1)["pos"][:, 3].copy()
    w2 = orc.calc_view(P, a2, n2)["pos"][:, 3].copy()
    assert ((w1 == 0) & (w2 > 0)).sum() > 50


def test_oracle_deleted_bits():
    a = small_asset(5003, 3, "Medium")
    cam = default_camera()
    tr = camera.Transform()
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    base = orc.calc_view(P).copy()
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2 ** 32, (a.splatCount + 31) // 32, dtype=np.uint64).astype(np.uint32)
    v = orc.calc_view(P, deleted_bits=bits).copy()
    deleted = ((bits[np.arange(a.splatCount) >> 5] >> (np.arange(a.splatCount) & 31).astype(np.uint32)) & 1).astype(bool)
    assert (v["pos"][deleted, 3] == 0).all() and (v["color"][deleted] == 0).all()
    assert np.array_equal(v[~deleted].view(np.uint32), base[~deleted].view(np.uint32))
    zero = orc.calc_view(P, deleted_bits=np.zeros_like(bits)).copy()
    assert np.array_equal(zero.view(np.uint32), base.view(np.uint32))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hmc") / "libhm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", so,
                           os.path.join(HERE, "host_math_harness.cpp")])
    return C.CDLL(so)


@pytest.mark.parametrize("name", ["hole_then_crop", "hole_ellipsoid"])
def test_kernel_header_matches_oracle_with_cutouts_and_deleted_bits(hm, name):
    a = small_asset(6000, 7, "Medium")
    tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695))
    cam = default_camera(az=33.0)
    orc = O.Oracle(a)
    P = camera.frame_params(cam, tr)
    arr, n = shader_data_array(CUTOUT_SETS[name], tr.localToWorldMatrix)
    bits = np.random.default_rng(9).integers(0, 2 ** 32, (a.splatCount + 31) // 32, dtype=np.uint64).astype(np.uint32)
    bits &= np.random.default_rng(10).integers(0, 2 ** 32, len(bits), dtype=np.uint64).astype(np.uint32)       # ~25 % deleted
    v0 = orc.calc_view(P, arr, n, bits).copy()
    v1 = np.zeros(a.splatCount, VIEW_DTYPE)
    hm.hm_calc_view_ex(C.byref(orc.desc), C.byref(P), arr, C.c_uint32(n), bits.ctypes.data_as(C.c_void_p), v1.ctypes.data_as(C.c_void_p))
    assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    assert 500 < (v0["pos"][:, 3] > 0).sum() < a.splatCount - 500


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["Medium", "VeryHigh"])
def test_gpu_view_and_frame_with_cutouts_and_deleted_bits(gpu_ctx, quality):
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget
    a = small_asset(30_011, 7, quality)
    tr = camera.Transform(position=(0.1, -0.2, 0.3), rotation=(0.1, 0.2, 0.05, 0.9695))
    r = GaussianSplatRenderer(gpu_ctx, a, tr)
    r.OnEnable()
    orc = O.Oracle(a)
    cam = default_camera(W=320, H=200, az=20.0)
    rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
    P = r.FrameParams(cam)
    bits = np.random.default_rng(3).integers(0, 2 ** 32, (a.splatCount + 31) // 32, dtype=np.uint64).astype(np.uint32)
    bits &= np.random.default_rng(5).integers(0, 2 ** 32, len(bits), dtype=np.uint64).astype(np.uint32)
    for name, use_bits in [("hole_then_crop", False), ("hole_ellipsoid", True), ("crop_box", True), ("only_null", False), (None, True), (None, False)]:
        r.m_Cutouts = CUTOUT_SETS[name] if name else None
        r.SetDeletedBits(bits if use_bits else None)
        arr, n = shader_data_array(r.m_Cutouts, tr.localToWorldMatrix)
        r.SortPoints(cam)
        r.CalcViewData(cam)
        rt.Clear()
        r.Draw(cam, rt)
        got = r.DownloadView()
        want = orc.calc_view(P, arr, n, bits if use_bits else None)
        assert views_equal(got, want), (name, use_bits)
        orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
        ref = orc.draw(P, 0)
        e = rt_err(rt.Download(), ref)
        assert e <= RT_TOL, (name, use_bits, e)
        st = r.FrameStats()
        assert st.tile_pairs == orc.pairs(P, st)
    r.OnDisable()


@pytest.mark.gpu
def test_gpu_cutout_argument_validation(gpu_ctx):
    from unitygaussiansplatting_amd import _lib
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer
    a = small_asset(5000, 3, "Medium")
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    l = _lib.lib()
    arr = (gs_cutout * 65)()
    assert l.gs_renderer_set_cutouts(r._r_h, arr, 65) == -1
    assert l.gs_renderer_set_cutouts(r._r_h, None, 3) == -1
    assert l.gs_renderer_set_cutouts(r._r_h, None, 0) == 0
    w = np.zeros(7, np.uint32)
    assert l.gs_renderer_set_deleted_bits(r._r_h, w.ctypes.data, 7) == -1      # wrong word count
    r.OnDisable()
