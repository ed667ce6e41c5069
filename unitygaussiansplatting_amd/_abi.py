"""ctypes mirror of include/gsplat_c.h (structs + enums only; loading the library is in _lib.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

GS_OK = 0
GS_ERR_INVALID_ARGUMENT = -1
GS_ERR_HIP = -2
GS_ERR_UNSUPPORTED_FORMAT = -3
GS_ERR_OUT_OF_MEMORY = -4
GS_ERR_INVALID_ASSET = -5
GS_ERR_PAIR_OVERFLOW = -6
GS_ERR_SORT_TIMEOUT = -7
GS_ERR_NO_DEVICE = -8
GS_ERR_COMM = -9
GS_SORT_FULL = 0
GS_SORT_VISIBLE = 1


class gs_asset_desc(C.Structure):
    _fields_ = [
        ("splat_count", C.c_uint32), ("pos_format", C.c_uint32), ("scale_format", C.c_uint32),
        ("color_format", C.c_uint32), ("sh_format", C.c_uint32), ("memory_kind", C.c_uint32),
        ("pos_data", C.c_void_p), ("pos_size", C.c_uint64),
        ("other_data", C.c_void_p), ("other_size", C.c_uint64),
        ("color_data", C.c_void_p), ("color_size", C.c_uint64),
        ("sh_data", C.c_void_p), ("sh_size", C.c_uint64),
        ("chunk_data", C.c_void_p), ("chunk_size", C.c_uint64),
    ]


class gs_frame_params(C.Structure):
    _fields_ = [
        ("matrix_mv", C.c_float * 16), ("matrix_object_to_world", C.c_float * 16),
        ("matrix_world_to_object", C.c_float * 16), ("matrix_vp", C.c_float * 16),
        ("proj_m00", C.c_float), ("proj_m11", C.c_float),
        ("screen_w", C.c_float), ("screen_h", C.c_float),
        ("cam_pos_world", C.c_float * 3),
        ("splat_scale", C.c_float), ("opacity_scale", C.c_float),
        ("sh_order", C.c_uint32), ("sh_only", C.c_uint32),
        ("near_clip", C.c_float), ("far_clip", C.c_float),
    ]


class gs_cutout(C.Structure):
    _fields_ = [("matrix", C.c_float * 16), ("type_and_flags", C.c_uint32)]


class gs_import_input(C.Structure):
    _fields_ = [("splat_count", C.c_uint32), ("pos", C.c_void_p), ("dc0", C.c_void_p), ("sh", C.c_void_p),
                ("opacity", C.c_void_p), ("scale", C.c_void_p), ("rot", C.c_void_p)]


class gs_import_formats(C.Structure):
    _fields_ = [("pos_format", C.c_uint32), ("scale_format", C.c_uint32), ("color_format", C.c_uint32), ("sh_format", C.c_uint32),
                ("linearize", C.c_uint32), ("morton", C.c_uint32)]


class gs_frame_stats(C.Structure):
    _fields_ = [("tile_pairs", C.c_uint64), ("pair_capacity", C.c_uint64), ("visible_splats", C.c_uint32),
                ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("sort_error", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("sort_mode", C.c_uint32), ("tie_long_runs", C.c_uint32), ("tie_longest_run", C.c_uint32)]


class gs_stage_times(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("calc_distances_ms", "sort_ms", "calc_view_ms", "bin_ms", "pair_sort_ms",
                                         "blend_ms", "resolve_ms", "total_ms")] + [("frames", C.c_uint32)] + \
               [("onesweep_depth_ms", C.c_float), ("onesweep_pairs_ms", C.c_float), ("onesweep_pair_launches", C.c_uint32),
                ("onesweep_depth_kernel_ms", C.c_float), ("onesweep_pairs_kernel_ms", C.c_float), ("onesweep_depth_launches", C.c_uint32)]


VIEW_DTYPE = np.dtype([("pos", "<f4", (4,)), ("axis1", "<f4", (2,)), ("axis2", "<f4", (2,)), ("color", "<u4", (2,))])
assert VIEW_DTYPE.itemsize == 40     # kGpuViewDataSize, GaussianSplatRenderer.cs:371


def make_asset_desc(asset, keepalive: list) -> gs_asset_desc:
    """Host-memory gs_asset_desc for a unitygaussiansplatting_amd.asset.GaussianSplatAsset.
    `keepalive` receives the contiguous arrays whose pointers the struct borrows."""
    d = gs_asset_desc()
    d.splat_count = asset.splatCount
    d.pos_format, d.scale_format = int(asset.posFormat), int(asset.scaleFormat)
    d.color_format, d.sh_format = int(asset.colorFormat), int(asset.shFormat)
    d.memory_kind = 0
    for nm, blob in (("pos", asset.posData), ("other", asset.otherData), ("color", asset.colorData),
                     ("sh", asset.shData), ("chunk", asset.chunkData)):
        if blob is None or len(blob) == 0:
            setattr(d, nm + "_data", None)
            setattr(d, nm + "_size", 0)
            continue
        arr = np.ascontiguousarray(blob, dtype=np.uint8)
        keepalive.append(arr)
        setattr(d, nm + "_data", arr.ctypes.data)
        setattr(d, nm + "_size", arr.nbytes)
    return d
