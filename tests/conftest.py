import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _has_gpu():
        # the oracle counts (tile, splat) pairs for the tile shape the library composites a target of that size with
        # (gs_renderer_tile_shape with a null renderer = the automatic choice; tests that override it set Oracle.tile themselves)
        import ctypes as C
        import oracle_lib
        from unitygaussiansplatting_amd import _lib

        def hook(W, H):
            w, h = C.c_uint32(), C.c_uint32()
            _lib.check(_lib.lib().gs_renderer_tile_shape(None, int(W), int(H), C.byref(w), C.byref(h)), "gs_renderer_tile_shape")
            return w.value, h.value
        oracle_lib.tile_shape_hook = hook


def _has_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gpu_ctx():
    from unitygaussiansplatting_amd.renderer import GpuContext
    ctx = GpuContext(0)
    yield ctx
    ctx.Dispose()
