"""Builds libgsplat_hip.so (gfx950) in-tree with hipcc.  `python -m unitygaussiansplatting_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgsplat_hip.so")
SOURCES = ["gs_api.hip", "gs_sort.hip", "gs_vissort.hip", "gs_view.hip", "gs_raster.hip", "gs_comm.hip", "gs_import.cpp"]
HEADERS = ["gs_common.h", "gs_device_math.h", os.path.join("..", "..", "include", "gsplat_c.h")]
# -ffp-contract=off: the kernels' arithmetic is written with explicit fmaf(); nothing else may fuse, so that
# results match the oracle's canonical arithmetic bit for bit.
# -fno-slp-vectorize: v_pk_*_f32 is no faster than two scalar VALU ops on gfx950 and the packing costs v_movs (and hides
# the fp16 mixed-precision FMA patterns of the blend kernel).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """A/B builds for kernel tuning: unitygaussiansplatting_amd/variants/<name>.so with extra -D flags (select with GSPLAT_LIB)."""
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    cmd = [hipcc()] + FLAGS + ["-D" + d for d in defines] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-lz", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-lz", "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
