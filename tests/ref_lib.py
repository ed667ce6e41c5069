"""ctypes binding of oracle/_ref/libgs_ref_*.so: the reference's OWN shader text (GaussianSplatting.hlsl,
SplatUtilities.compute:37-252, RenderGaussianSplats.shader, GaussianComposite.shader) compiled for the host by
oracle/ref_build (gen_ref.py + hlsl_compat.h).  Test infrastructure: only tests/ may load it, and only to check the oracle.

Three builds of the same text (oracle/Makefile `ref`):  strict (separately rounded IEEE operations),  fused (intrinsics as
mad chains + host-compiler contraction, g++),  fused_clang (the same with clang's contraction choices)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from unitygaussiansplatting_amd import asset as A
from unitygaussiansplatting_amd import bc7
from unitygaussiansplatting_amd._abi import VIEW_DTYPE, gs_frame_params, make_asset_desc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle", "_ref")
REFERENCE = "/root/reference"
BUILDS = ("strict", "fused", "fused_clang")


def build() -> None:
    """(Re)build oracle/_ref when the reference tree is here; the prebuilt libraries travel to machines without it."""
    if os.path.isdir(os.path.join(REFERENCE, "package", "Shaders")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)


_libs: dict = {}


def lib(which: str):
    if which not in _libs:
        build()
        path = os.path.join(_DIR, f"libgs_ref_{which}.so")
        if not os.path.exists(path):
            if which == "fused_clang":
                pytest.skip("no clang++ on the machine that built oracle/_ref")
            if not os.path.isdir(REFERENCE):
                pytest.skip("oracle/_ref was not built and there is no reference tree to build it from")
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        L.gsr_cs_sortable_uint.restype = C.c_uint32
        L.gsr_cs_sortable_uint.argtypes = [C.c_float]
        L.gsr_cs_encode_morton.restype = C.c_uint32
        L.gsr_cs_encode_quat_norm10.restype = C.c_uint32
        _libs[which] = L
    return _libs[which]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# The oracle's canonical arithmetic is BIT-equal to the `fused` build as g++ 11.4 contracts the shader text: a
# property of one compiler's contraction choices.  oracle/Makefile records the compiler that built oracle/_ref in _ref/COMPILER; with
# another one the fused build is a different -- equally legitimate -- member of the family, and the tests that assert bit-equality
# fall back to the bounds of DESIGN.md section 5.1 (what `fused_clang` is held to) instead of failing without a product change.
PINNED_COMPILER = "g++ (Ubuntu 11.4.0"


def fused_compiler() -> str:
    lib("fused")
    path = os.path.join(_DIR, "COMPILER")
    if os.path.exists(path):
        return open(path).read().strip()
    try:                                   # an _ref built before the compiler was recorded: it was built in this image
        return subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0].strip()
    except Exception:
        return "unknown"


def fused_is_pinned() -> bool:
    return fused_compiler().startswith(PINNED_COMPILER)


def assert_axes_close(vr, vo, live):
    """Axis offsets in pixels.  The 2x2 eigen-decomposition is ill-conditioned for a nearly isotropic screen covariance (the axes of a
    circle may point anywhere), so an ulp of difference in cov2d can turn the axes of such a splat by a fraction of a degree without
    changing the ellipse.  Asserted: the ELLIPSE (axis1 axis1^T + axis2 axis2^T, what the fragments see) agrees to rounding noise for
    every splat, and the axes themselves for all but the ill-conditioned tail (DESIGN.md section 5.1 measured 0.041 px there)."""
    cat = lambda v: np.concatenate([v["axis1"][live], v["axis2"][live]], axis=1).astype(np.float64)
    ar, ao = cat(vr), cat(vo)
    assert np.array_equal(np.isnan(ar), np.isnan(ao))
    ok = ~np.isnan(ao).any(axis=1)
    ar, ao = ar[ok], ao[ok]
    shape = lambda a: np.stack([a[:, 0] ** 2 + a[:, 2] ** 2, a[:, 0] * a[:, 1] + a[:, 2] * a[:, 3], a[:, 1] ** 2 + a[:, 3] ** 2], axis=1)
    sr, so = shape(ar), shape(ao)
    rel = np.abs(sr - so).max(axis=1) / np.abs(so).max(axis=1)
    assert np.quantile(rel, 0.999) <= 2e-5 and rel.max() <= 2e-3, (rel.max(), np.quantile(rel, 0.999))
    d = np.abs(ar - ao).max(axis=1)
    assert np.quantile(d, 0.999) <= 1e-3 and np.median(d) <= 1e-5 and d.max() <= 0.5, (d.max(), np.quantile(d, 0.999), np.median(d))




def assert_view_is_the_fused_build(v_fused, v_other):
    """40-byte SplatViewData records of the `fused` build (v_fused) vs the oracle's / the HIP kernels' (v_other): bit-equal when _ref was
    built by the pinned compiler; with another compiler (another, equally legitimate, set of contraction choices) the bounds DESIGN.md
    section 5.1 holds `fused_clang` to: clip position bit-equal, packed colour within half an ulp in < 1e-3 of the records, axes close."""
    if fused_is_pinned():
        # Bit for bit -- every field of every record -- with ONE measured exception class: the packed colour of about one record in
        # 300,000 differs by one half-precision ulp in one channel (C1 from inside the cloud: 1 of 3 x 100,000; none in the 8 x 4 x
        # 30,011 records of the format matrix nor in the C2 / C3 / C2d samples).  It appears with _SHOrder = 3 only: g++ vectorises the
        # float3 sum of the seven degree-3 terms per component and contracts one component's chain differently from the in-order
        # fmaf chain the canon uses for all three; the fp32 results differ by an ulp in a quarter of the records and one in ~2^13 of
        # those crosses an fp16 rounding boundary.  Allowed: <= 1e-5 of the records, colour only, one half ulp; everything else equal.
        g, w = v_fused.view(np.uint32).reshape(-1, 10), v_other.view(np.uint32).reshape(-1, 10)
        gf, wf = g[:, :8].view(np.float32), w[:, :8].view(np.float32)
        same8 = (g[:, :8] == w[:, :8]) | (np.isnan(gf) & np.isnan(wf))          # NaN = NaN (a NaN axis only means "not drawn")
        assert same8.all(), "clip position / axes differ from the fused build of the reference text"
        bad = np.flatnonzero((g[:, 8:] != w[:, 8:]).any(axis=1))
        assert len(bad) <= max(1, int(1e-5 * len(g))), f"{len(bad)} packed colours differ from the fused build of the reference text"
        for i in bad:
            hg = g[i, 8:].view(np.uint16).astype(np.int32); hw = w[i, 8:].view(np.uint16).astype(np.int32)
            assert np.abs(hg - hw).max() <= 1 and (hg != hw).sum() == 1, (i, g[i], w[i])
        return
    print(f"oracle/_ref was built by `{fused_compiler()}` (pinned: {PINNED_COMPILER}...): asserting the section 5.1 bounds instead of bit-equality")
    assert np.array_equal(v_fused["pos"].view(np.uint32), v_other["pos"].view(np.uint32))
    assert (v_fused["color"] != v_other["color"]).any(axis=1).mean() <= 1e-3
    assert_axes_close(v_fused, v_other, v_other["pos"][:, 3] > 0)


def flipped(P: gs_frame_params) -> gs_frame_params:
    """The matrices Unity binds when it renders into a texture on D3D / Vulkan / Metal (GL.GetGPUProjectionMatrix(proj, true)):
    projection row 1 negated, so clip.y points down and the D3D viewport stores the picture bottom row first."""
    F = gs_frame_params.from_buffer_copy(bytes(P))
    for k in range(4):
        F.matrix_vp[4 + k] = -P.matrix_vp[4 + k]
    F.proj_m11 = -P.proj_m11
    return F


class Ref:
    """The reference kernels bound to one asset (SetAssetDataOnCS)."""

    def __init__(self, asset, which: str):
        self.L = lib(which)
        self.which = which
        self.n = asset.splatCount
        self._keep = []
        a = asset
        if a.colorFormat == A.ColorFormat.BC7:            # the texture unit decodes BC7; hand the texels over as R8G8B8A8
            h = len(a.colorData) // (2048 // 4 * 16) * 4
            rgba = bc7.decode_texture(a.colorData, 2048, h)
            a = A.GaussianSplatAsset(splatCount=a.splatCount, posFormat=a.posFormat, scaleFormat=a.scaleFormat, colorFormat=A.ColorFormat.Norm8x4,
                                     shFormat=a.shFormat, posData=a.posData, otherData=a.otherData, colorData=rgba.reshape(-1), shData=a.shData,
                                     chunkData=a.chunkData, name=a.name)
        self.desc = make_asset_desc(a, self._keep)
        self.order = np.arange(self.n, dtype=np.uint32)
        self.keys = np.zeros(self.n, np.uint32)
        self.view = np.zeros(self.n, VIEW_DTYPE)

    def bind(self):
        assert self.L.gsr_cs_bind_asset(C.byref(self.desc)) == 0

    def set_indices(self):
        self.L.gsr_cs_set_indices(_p(self.order), C.c_uint32(self.n))

    def calc_distances(self, matrix_sort):
        self.bind()
        m = np.ascontiguousarray(matrix_sort, np.float32).reshape(16)
        self.L.gsr_cs_calc_distances(_p(self.order), _p(m), _p(self.keys), C.c_uint32(self.n))
        return self.keys

    def decode_all(self):
        self.bind()
        out = np.zeros((self.n, 59), np.float32)
        self.L.gsr_cs_decode_all(_p(out), C.c_uint32(self.n))
        return out

    def calc_view(self, P: gs_frame_params, cutouts=None, cutout_count: int = 0, deleted_bits=None):
        self.bind()
        db = np.ascontiguousarray(deleted_bits, np.uint32) if deleted_bits is not None else None
        self.L.gsr_cs_set_frame(C.byref(P), cutouts if cutout_count else None, C.c_uint32(cutout_count),
                                _p(db) if db is not None else None, C.c_uint64(db.nbytes if db is not None else 0))
        self.L.gsr_cs_calc_view(_p(self.view), C.c_uint32(self.n))
        return self.view

    def draw(self, W: int, H: int, near: float, far: float, rt=None):
        """DrawProcedural of self.view through self.order under the D3D rules of oracle/ref_build/ref_render.cpp."""
        self.L.gsr_rs_bind(_p(self.view), _p(self.order), C.c_uint32(self.n), C.c_float(W), C.c_float(H))
        if rt is None:
            rt = np.zeros((H, W, 4), np.uint16)
        self.L.gsr_rs_draw(_p(rt), C.c_uint32(W), C.c_uint32(H), C.c_float(near), C.c_float(far))
        return rt


def fragment(which: str, q, col):
    out = np.zeros(4, np.float32)
    d = lib(which).gsr_rs_frag(_p(np.asarray(q, np.float32)), _p(np.asarray(col, np.float32)), _p(out))
    return int(d), out


def resolve(which: str, rt: np.ndarray, bg):
    H, W, _ = rt.shape
    out = np.zeros((H, W, 4), np.float32)
    lib(which).gsr_comp_resolve(_p(np.ascontiguousarray(rt)), C.c_uint32(W), C.c_uint32(H), _p(np.asarray(bg, np.float32)), _p(out))
    return out
