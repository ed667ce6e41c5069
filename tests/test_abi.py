"""The C-ABI library loads on a box without a GPU and exports every symbol include/gsplat_c.h declares (CPU only;
no compute calls)."""
import ctypes as C
import os
import re

import numpy as np

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
    assert lib.gs_abi_version() == 6
    for code in range(0, -9, -1):
        assert lib.gs_error_string(code) not in (None, b"", b"unknown error")
    assert lib.gs_error_string(-99) == b"unknown error"


def test_struct_layouts_match_the_header():
    assert C.sizeof(_abi.gs_asset_desc) == 6 * 4 + 5 * 16
    assert C.sizeof(_abi.gs_frame_params) == (64 + 2 + 2 + 3 + 2 + 2 + 2) * 4
    assert C.sizeof(_abi.gs_frame_stats) == 40
    assert C.sizeof(_abi.gs_cutout) == 68            # GaussianCutout.ShaderData: float4x4 + uint
    assert C.sizeof(_abi.gs_stage_times) == 56
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
    assert lib.gs_context_destroy(None) == 0 and lib.gs_asset_destroy(None) == 0      # destroying null is a no-op
    if not os.path.exists("/dev/kfd"):
        h = C.c_void_p()
        assert lib.gs_context_create(0, None, C.byref(h)) == _abi.GS_ERR_NO_DEVICE      # fails loudly, no CPU fallback
        assert not h


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


def test_plain_c_program_links_and_runs(tmp_path):
    """tests/abi_smoke.c: a C99 translation unit that includes the header, links libgsplat_hip.so and uses the ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _lib.lib()                                     # makes sure the library is built
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "abi_smoke.c"), "-o", exe,
                           "-L" + libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "abi_smoke ok" in out.stdout
