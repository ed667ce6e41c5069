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
         ("Medium", dict(formatPos=A.VectorFormat.Norm6, formatScale=A.VectorFormat.Float32, formatSH=A.SHFormat.Norm11, formatColor=A.ColorFormat.Float32x4))]


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
    orc.draw(P)
    out = np.zeros((a.splatCount, 5), np.int32)
    cxy = np.zeros((a.splatCount, 2), np.float32)
    hm.hm_prepare(v.ctypes.data_as(C.c_void_p), C.c_uint32(a.splatCount), C.byref(P), out.ctypes.data_as(C.c_void_p), cxy.ctypes.data_as(C.c_void_p))
    w = np.maximum(out[:, 1] - out[:, 0] + 1, 0).astype(np.int64)
    h = np.maximum(out[:, 3] - out[:, 2] + 1, 0).astype(np.int64)
    assert int((w * h).sum()) == orc.tile_pairs
    assert int(((w * h) > 0).sum()) == orc.visible
