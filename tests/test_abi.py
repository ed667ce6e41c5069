"""The C-ABI library loads on a box without a GPU and exports every symbol include/gsplat_c.h declares (CPU only;
no compute calls)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from unitygaussiansplatting_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "gsplat_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 35
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gsplat_c.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_error_strings():
    lib = _lib.lib()
    assert lib.gs_abi_version() == 9
    for code in range(0, -10, -1):
        assert lib.gs_error_string(code) not in (None, b"", b"unknown error")
    assert lib.gs_error_string(-99) == b"unknown error"


def test_struct_layouts_match_the_header():
    assert C.sizeof(_abi.gs_asset_desc) == 6 * 4 + 5 * 16
    assert C.sizeof(_abi.gs_frame_params) == (64 + 2 + 2 + 3 + 2 + 2 + 2) * 4
    assert C.sizeof(_abi.gs_frame_stats) == 56
    assert C.sizeof(_abi.gs_cutout) == 68            # GaussianCutout.ShaderData: float4x4 + uint
    assert C.sizeof(_abi.gs_stage_times) == 60
    assert _abi.VIEW_DTYPE.itemsize == 40


def test_argument_validation_without_a_device():
    lib = _lib.lib()
    assert lib.gs_asset_create(None, None, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_create(None, None, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_sort(None, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_target_create(None, 16, 16, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_sorter_create(None, 16, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_set_cutouts(None, None, 0) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_set_deleted_bits(None, None, 0) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_set_sort_mode(None, _abi.GS_SORT_VISIBLE) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_download_visible_order(None, None, 0, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_set_sort_history_limit(None, 8) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_sort_history(None, None, None, None) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_target_create(None, 65536, 16, None) == _abi.GS_ERR_INVALID_ARGUMENT     # pixel rectangles are packed in 16-bit fields
    assert lib.gs_context_destroy(None) == 0 and lib.gs_asset_destroy(None) == 0      # destroying null is a no-op
    if not os.path.exists("/dev/kfd"):
        h = C.c_void_p()
        assert lib.gs_context_create(0, None, C.byref(h)) == _abi.GS_ERR_NO_DEVICE      # fails loudly, no CPU fallback
        assert not h


def test_automatic_tile_shape_is_host_logic():
    """gs_renderer_tile_shape(NULL, w, h) = the compositor tile a fresh renderer would pick for a w x h target (host code, no device):
    16x16 below 2^18 pixels, 32x16 from there on (32x32 only after a draw reported > 6 tiles per visible splat: tests/test_gpu_draw.py);
    GSPLAT_TILE pins it for the whole process, which is why the pinned cases run in a child process."""
    lib = _lib.lib()
    if os.environ.get("GSPLAT_TILE"):
        pytest.skip("GSPLAT_TILE pins the shape")
    tw, th = C.c_uint32(0), C.c_uint32(0)
    for (w, h), want in {(320, 200): (16, 16), (511, 512): (16, 16), (512, 512): (32, 16), (1200, 797): (32, 16), (1920, 1080): (32, 16),
                         (3840, 2160): (32, 16), (65535, 65535): (32, 16), (1, 1): (16, 16)}.items():
        assert lib.gs_renderer_tile_shape(None, w, h, C.byref(tw), C.byref(th)) == 0
        assert (tw.value, th.value) == want, (w, h, tw.value, th.value)
    assert lib.gs_renderer_tile_shape(None, 64, 64, None, C.byref(th)) == _abi.GS_ERR_INVALID_ARGUMENT
    assert lib.gs_renderer_set_tile_shape(None, 16, 16) == _abi.GS_ERR_INVALID_ARGUMENT
    code = ("import ctypes as C; from unitygaussiansplatting_amd import _lib; l = _lib.lib(); a, b = C.c_uint32(), C.c_uint32();"
            "assert l.gs_renderer_tile_shape(None, 320, 200, C.byref(a), C.byref(b)) == 0; print(a.value, b.value)")
    for pin, want in (("32x32", "32 32"), ("16x16", "16 16"), ("nonsense", "16 16")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSPLAT_TILE=pin, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.split("\n")[0].strip() == want, (pin, out.stdout, out.stderr[-500:])


def test_product_code_never_references_the_oracle():
    pkg = os.path.join(ROOT, "unitygaussiansplatting_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cs")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                # comments may cite the oracle; code may not bind it: its symbols (gso_*), its library, its binding module
                assert "gso_" not in txt and "oracle_lib" not in txt and "libgs_oracle" not in txt, f
                # ... nor oracle/_ref, the reference's own shader text compiled for the host (symbols gsr_*, tests/ref_lib.py)
                assert "gsr_" not in txt and "ref_lib" not in txt and "libgs_ref" not in txt and "hlsl_compat" not in txt, f
                assert not re.search(r'#include\s*"[^"]*oracle', txt), f


def test_header_is_plain_c99_and_the_dotnet_binding_covers_it():
    """include/gsplat_c.h must compile as C99 (it is the boundary a non-C++ host binds), and the shipped P/Invoke file must
    declare every function the header does -- no more, no fewer."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "gsplat_c.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", hdr])
    names = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", open(hdr).read()))
    cs = open(os.path.join(root, "unitygaussiansplatting_amd", "dotnet", "GaussianSplatNative.cs")).read()
    assert set(re.findall(r"extern\s+\w+\s+(gs_[a-z0-9_]+)\s*\(", cs)) == names
    # ... and its error / sort-mode enums carry the header's values (CamelCase of the header's names)
    htxt = open(hdr).read()
    camel = lambda n: "".join(w.capitalize() for w in n.split("_"))
    want_err = {camel(n): int(v) for n, v in re.findall(r"\bGS_ERR_([A-Z_]+)\s*=\s*(-?\d+)", htxt)}
    want_err["Ok"] = 0
    have_err = {n: int(v) for n, v in re.findall(r"(\w+)\s*=\s*(-?\d+)", re.search(r"enum Error \{([^}]*)\}", cs).group(1))}
    assert have_err == want_err
    want_mode = {camel(n): int(v) for n, v in re.findall(r"\bGS_SORT_([A-Z_]+)\s*=\s*(\d+)", htxt)}
    have_mode = {n: int(v) for n, v in re.findall(r"(\w+)\s*=\s*(\d+)", re.search(r"enum SortMode \{([^}]*)\}", cs).group(1))}
    assert have_mode == want_mode and want_mode == {"Full": 0, "Visible": 1}


def _run_plain_c_program(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _lib.lib()                                     # makes sure the library is built
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "abi_smoke.c"), "-o", exe,
                           "-L" + libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "abi_smoke ok" in out.stdout
    return out.stdout


def test_plain_c_program_links_and_runs(tmp_path):
    """tests/abi_smoke.c: a C99 translation unit that includes the header, links libgsplat_hip.so and uses the ABI."""
    _run_plain_c_program(tmp_path)


@pytest.mark.gpu
def test_plain_c_program_renders_through_the_abi(tmp_path):
    """The same program on a GPU box: frames through nothing but the C-ABI -- importer -> gs_asset_create -> three renderers in GS_SORT_VISIBLE,
    two of them on contexts of their own sharing the asset with the frames dealt alternately (history limit 2): same bits, same order."""
    assert "GPU present" in _run_plain_c_program(tmp_path)
