"""Loads libgsplat_hip.so (the C-ABI of include/gsplat_c.h) with ctypes.

There is no CPU fallback: if the library is missing or a call fails, GsError is raised."""
from __future__ import annotations

import ctypes as C
import os

from ._abi import (gs_asset_desc, gs_cutout, gs_frame_params, gs_frame_stats, gs_import_formats, gs_import_input,
                   gs_stage_times)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPLAT_LIB") or os.path.join(_HERE, "libgsplat_hip.so")   # GSPLAT_LIB: A/B a variant build


class GsError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: gs_error {code}" + (f" ({detail})" if detail else ""))


_lib = None

# name -> (restype, argtypes); every symbol include/gsplat_c.h declares
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "gs_abi_version": (C.c_int32, []),
    "gs_error_string": (C.c_char_p, [C.c_int32]),
    "gs_last_error_string": (C.c_char_p, []),
    "gs_context_create": (C.c_int32, [C.c_int32, _P, _PP]),
    "gs_context_destroy": (C.c_int32, [_P]),
    "gs_context_synchronize": (C.c_int32, [_P]),
    "gs_context_set_shared_gpu": (C.c_int32, [_P, C.c_int32]),
    "gs_context_set_overlap": (C.c_int32, [_P, C.c_int32]),
    "gs_context_device_info": (C.c_int32, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "gs_asset_create": (C.c_int32, [_P, C.POINTER(gs_asset_desc), _PP]),
    "gs_asset_destroy": (C.c_int32, [_P]),
    "gs_asset_splat_count": (C.c_int32, [_P, C.POINTER(C.c_uint32)]),
    "gs_asset_device_blobs": (C.c_int32, [_P, C.c_void_p * 5, C.c_uint64 * 5]),
    "gs_asset_info": (C.c_int32, [_P, C.c_uint32 * 6]),
    "gs_comm_unique_id": (C.c_int32, [C.c_uint8 * 128]),
    "gs_comm_create": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_uint8 * 128, _PP]),
    "gs_comm_destroy": (C.c_int32, [_P]),
    "gs_comm_info": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gs_asset_broadcast": (C.c_int32, [_P, _P, C.c_int32, _PP]),
    "gs_asset_replicate": (C.c_int32, [_P, _P, _PP]),
    "gs_renderer_create": (C.c_int32, [_P, _P, _PP]),
    "gs_renderer_destroy": (C.c_int32, [_P]),
    "gs_renderer_reset_order": (C.c_int32, [_P]),
    "gs_renderer_sort": (C.c_int32, [_P, C.POINTER(C.c_float)]),
    "gs_renderer_calc_view": (C.c_int32, [_P, C.POINTER(gs_frame_params)]),
    "gs_renderer_draw": (C.c_int32, [_P, C.POINTER(gs_frame_params), _P]),
    "gs_renderer_render": (C.c_int32, [_P, C.POINTER(C.c_float), C.POINTER(gs_frame_params), _P, C.c_int32]),
    "gs_renderer_set_cutouts": (C.c_int32, [_P, C.POINTER(gs_cutout), C.c_uint32]),
    "gs_renderer_set_deleted_bits": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_renderer_set_view_buffer_mode": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_set_blend_mode": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_set_render_mode": (C.c_int32, [_P, C.c_int32, C.c_float]),
    "gs_renderer_set_tile_shape": (C.c_int32, [_P, C.c_uint32, C.c_uint32]),
    "gs_renderer_tile_shape": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gs_renderer_set_profiling": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_set_kernel_timing": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_reserve_pairs": (C.c_int32, [_P, C.c_uint64]),
    "gs_renderer_poll_pairs": (C.c_int32, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gs_renderer_download_order": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_renderer_download_distances": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_renderer_upload_order": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_renderer_set_sort_mode": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_sort_mode": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gs_renderer_set_sort_history_limit": (C.c_int32, [_P, C.c_uint32]),
    "gs_renderer_sort_history": (C.c_int32, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "gs_renderer_set_frames_in_flight": (C.c_int32, [_P, C.c_int32]),
    "gs_renderer_frames_in_flight": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gs_renderer_download_visible_order": (C.c_int32, [_P, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    "gs_renderer_download_view": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_renderer_download_raster_records": (C.c_int32, [_P, _P, _P, _P]),
    "gs_renderer_frame_stats": (C.c_int32, [_P, C.POINTER(gs_frame_stats)]),
    "gs_renderer_frame_times": (C.c_int32, [_P, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "gs_renderer_stage_times": (C.c_int32, [_P, C.POINTER(gs_stage_times)]),
    "gs_target_create": (C.c_int32, [_P, C.c_uint32, C.c_uint32, _PP]),
    "gs_target_destroy": (C.c_int32, [_P]),
    "gs_target_clear": (C.c_int32, [_P]),
    "gs_target_set_scene_depth": (C.c_int32, [_P, _P, C.c_int32]),
    "gs_target_download": (C.c_int32, [_P, _P, C.c_size_t]),
    "gs_target_resolve": (C.c_int32, [_P, C.POINTER(C.c_float), _P, _P]),
    "gs_target_device_ptr": (C.c_int32, [_P, _PP, _PP]),
    "gs_target_set_profiling": (C.c_int32, [_P, C.c_int32]),
    "gs_target_resolve_time": (C.c_int32, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "gs_import_blob_sizes": (C.c_int32, [C.c_uint32, C.POINTER(gs_import_formats), C.c_uint64 * 5]),
    "gs_import_encode": (C.c_int32, [C.POINTER(gs_import_input), C.POINTER(gs_import_formats), C.c_void_p * 5, C.c_uint64 * 5,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gs_ply_open": (C.c_int32, [C.c_char_p, _PP, C.POINTER(C.c_uint32)]),
    "gs_spz_open": (C.c_int32, [C.c_char_p, _PP, C.POINTER(C.c_uint32)]),
    "gs_ply_arrays": (C.c_int32, [_P, C.POINTER(gs_import_input)]),
    "gs_ply_close": (C.c_int32, [_P]),
    "gs_sorter_create": (C.c_int32, [_P, C.c_uint32, _PP]),
    "gs_sorter_destroy": (C.c_int32, [_P]),
    "gs_sorter_dispatch": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32]),
    "gs_sorter_sort_host": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32]),
}


def lib():
    """The loaded library.  Raises GsError if it has not been built (python -m unitygaussiansplatting_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsError(-2, "load", f"{LIB_PATH} is missing: build it with `python -m unitygaussiansplatting_amd.build`")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("GSPLAT_LIB") and not hasattr(l, name):
                continue                   # an A/B variant build of an older ABI: the missing entry point fails when it is called
            fn = getattr(l, name)          # AttributeError here = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, where: str) -> None:
    if code != 0:
        detail = lib().gs_last_error_string()
        raise GsError(code, where, (detail or b"").decode("utf-8", "replace"))
