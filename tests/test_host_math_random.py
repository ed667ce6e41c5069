"""CPU suite: the kernels' arithmetic header (csrc/gs_device_math.h) compiled for the HOST, on the seeded random walk the GPU is held to the oracle on
(tests/test_gpu_random_parity.py: presets, extents, cameras outside / inside the cloud / grazing / far, 10-110 degree fields of view, targets from 1x1 to
1280x720 incl. 48x1024 and 1024x48, rotated / non-uniformly scaled / mirrored transforms, _SplatScale, _SplatOpacityScale, _SHOrder, _SHOnly):

    sort keys, 40-byte view records          == the oracle's, bit for bit                                     (SplatUtilities.compute:69-82, 189-252)
    CalcViewGeom(allowCull), the per-frame path    gives up only on splats whose full-path footprint is empty; every other splat's geometry bit-identical
    ChunkCornerOutside, the whole-chunk cull       fires only for 256-splat chunks without a drawable splat

The two culls are SAFETY properties of this implementation (the reference has no cull before its rasteriser): a cull that dropped a drawable splat would be a
wrong frame nobody flags, so they are walked over many more cameras than a GPU budget allows.  Eight seeds in the suite; GSPLAT_HOSTMATH_SEEDS=n adds n more
(3,000 were run once in this container = 9,024 cameras: all passed)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_random_parity import _case
from test_host_math import hm  # noqa: F401  (the fixture: builds tests/host_math_harness.cpp)
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd._abi import VIEW_DTYPE

_SEEDS = list(range(1, 9)) + [100 + k for k in range(int(os.environ.get("GSPLAT_HOSTMATH_SEEDS", "0")))]


@pytest.mark.parametrize("seed", _SEEDS)
def test_kernel_arithmetic_on_a_random_case(hm, seed):  # noqa: F811
    c = _case(seed)
    if c["quality"] in ("Low", "VeryLow"):                       # (BC7 / Cluster16k through the host harness: test_host_math.py's fixed cases; here the walk is about cameras)
        c["quality"] = "Medium"
    n = min(c["n"], 20_011)
    raw = scenes.make_splats(n, c["asset_seed"], c["extent"])
    a = creator.CreateAssetFromSplatsNative(raw, c["quality"], name=f"rnd{seed}")
    orc = O.Oracle(a)
    tr = c["tr"] or camera.Transform()
    f = c["fields"]
    what = {k: c[k] for k in ("quality", "extent", "kind", "fields")} | {"n": n, "size": (c["cam"].pixelWidth, c["cam"].pixelHeight), "transform": c["tr"]}
    ms = np.ascontiguousarray(camera.sort_matrix(c["cam"], tr.localToWorldMatrix), np.float32).reshape(16)
    k0 = orc.calc_distances(ms).copy()
    k1 = np.zeros_like(k0)
    hm.hm_calc_distances(C.byref(orc.desc), orc.order.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), k1.ctypes.data_as(C.c_void_p))
    assert np.array_equal(k0, k1), f"seed {seed}: keys; {what}"
    P = camera.frame_params(c["cam"], tr, f["m_SplatScale"], f["m_OpacityScale"], f["m_SHOrder"], f["m_SHOnly"])
    v0 = orc.calc_view(P).copy()
    v1 = np.zeros(a.splatCount, VIEW_DTYPE)
    hm.hm_calc_view(C.byref(orc.desc), C.byref(P), v1.ctypes.data_as(C.c_void_p))
    g, w = v1.view(np.uint32).reshape(-1, 10), v0.view(np.uint32).reshape(-1, 10)
    same = (g == w) | (np.isnan(g.view(np.float32)) & np.isnan(w.view(np.float32)) & (np.arange(10) < 8))      # NaN = NaN in the float fields (a NaN axis means "not drawn")
    assert same.all(), f"seed {seed}: {int((~same).any(axis=1).sum())} view records differ; {what}"
    # the culls: the case's own camera, then two that look at random points (not at the cloud's centre) with random near / far planes
    rng = np.random.default_rng(9_000 + seed)
    hm.hm_chunk_cull.restype = C.c_uint32
    nch = (a.splatCount + 255) // 256
    e = c["extent"]
    cams = [c["cam"]] + [camera.Camera(position=tuple(rng.uniform(-1.5, 1.5, 3) * e), target=tuple(rng.uniform(-1.0, 1.0, 3) * e),
                                       fieldOfView=float(rng.choice([10.0, 39.0965, 75.0, 110.0])), pixelWidth=c["cam"].pixelWidth, pixelHeight=c["cam"].pixelHeight,
                                       nearClipPlane=float(rng.choice([0.01, 0.3, 1.0])) * max(e, 1.0) / 3.0, farClipPlane=float(rng.choice([2.0, 8.0, 1000.0])) * max(e, 1.0))
                         for _ in range(2)]
    for k, cm in enumerate(cams):
        Pk = camera.frame_params(cm, tr, f["m_SplatScale"], f["m_OpacityScale"], f["m_SHOrder"], f["m_SHOnly"])
        out = np.zeros((a.splatCount, 2), np.uint32)
        hm.hm_cull_check(C.byref(orc.desc), C.byref(Pk), out.ctypes.data_as(C.c_void_p))
        culled = (out[:, 0] & 1).astype(bool)
        assert not (out[:, 0] & 2).any(), f"seed {seed} camera {k}: geometry of an unculled splat differs from the full path; {what}"
        assert not (culled & (out[:, 1] > 0)).any(), f"seed {seed} camera {k}: {int((culled & (out[:, 1] > 0)).sum())} drawable splats were culled early; {what}"
        cc = np.zeros(nch, np.uint8)
        hm.hm_chunk_cull(C.byref(orc.desc), C.byref(Pk), cc.ctypes.data_as(C.c_void_p))
        per_splat = np.repeat(cc, 256)[:a.splatCount].astype(bool)
        bad = per_splat & (out[:, 1] > 0)
        assert not bad.any(), f"seed {seed} camera {k}: {int(bad.sum())} drawable splats sit in culled chunks; {what}"
