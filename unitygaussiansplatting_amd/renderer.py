"""Host-side mirror of the reference's renderer API over the C-ABI (include/gsplat_c.h).

Same class, field and method names as /root/reference/package/Runtime/GaussianSplatRenderer.cs
(GaussianSplatRenderSystem :17-212, GaussianSplatRenderer :214-680) and GpuSorting.cs, minus the Unity
plumbing (MonoBehaviour, CommandBuffer, materials): where the C# records a dispatch into a CommandBuffer,
these methods enqueue the corresponding HIP kernels on the context's stream.  A .NET host would be the same
thin layer over [DllImport("gsplat_hip")] (unitygaussiansplatting_amd/dotnet/GaussianSplatNative.cs).
"""
from __future__ import annotations

import atexit
import ctypes as C
import enum
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._abi import (GS_ERR_PAIR_OVERFLOW, GS_SORT_FULL, GS_SORT_VISIBLE, VIEW_DTYPE, gs_frame_params, gs_frame_stats, gs_stage_times,
                   make_asset_desc)
from ._lib import GsError, check
from .asset import GaussianSplatAsset, kCurrentVersion
from .camera import Camera, Transform, frame_params, sort_matrix
from .cutout import GaussianCutout, shader_data_array


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# Contexts still alive at interpreter exit are disposed (children first) BEFORE the HIP runtime's own static
# destructors run; a late __del__ calling into a torn-down runtime aborts the process.
_live_contexts = weakref.WeakSet()


@atexit.register
def _dispose_all_contexts():
    for ctx in list(_live_contexts):
        try:
            ctx.Dispose()
        except Exception:
            pass


class RenderMode(enum.IntEnum):          # GaussianSplatRenderer.RenderMode (:217-223)
    Splats = 0
    DebugPoints = 1
    DebugPointIndices = 2
    DebugBoxes = 3
    DebugChunkBounds = 4


class SortMode(enum.IntEnum):            # gs_sort_mode: what SortPoints sorts (no counterpart in the reference, whose sort precedes its cull)
    Full = GS_SORT_FULL                  # all N splats, every SortPoints, like GaussianSplatRenderer.SortPoints (:612-639)
    Visible = GS_SORT_VISIBLE            # cull first: only the splats CalcViewData found visible, inside Draw


class GpuContext:
    """One GPU + one HIP stream (the analogue of Unity's graphics device + render thread)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._h = C.c_void_p()
        check(_lib.lib().gs_context_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)), "gs_context_create")
        self.device = device
        self._children = weakref.WeakSet()      # targets / sorters / renderers: disposed before the context is
        _live_contexts.add(self)

    def _adopt(self, child) -> None:
        self._children.add(child)

    def Synchronize(self) -> None:
        check(_lib.lib().gs_context_synchronize(self._h), "gs_context_synchronize")

    def SetSharedGpu(self, shared) -> None:
        """Other kernels that wait on their own workgroups (another process using the library, another context sorting or binning at the same time) may share
        the GPU: gs_context_set_shared_gpu -- the full sort's XCD-dealt gather pass and a persistent grid's static first round give way to the dependency-ordered
        forms.  True / False pin it, None = automatic (shared while this process holds more than one context on the device)."""
        check(_lib.lib().gs_context_set_shared_gpu(self._h, -1 if shared is None else int(bool(shared))), "gs_context_set_shared_gpu")

    def SetOverlap(self, enabled: bool) -> None:
        """Run SortPoints concurrently with CalcViewData on the context's second queue (default off: no gain on MI355X)."""
        check(_lib.lib().gs_context_set_overlap(self._h, int(bool(enabled))), "gs_context_set_overlap")

    def DeviceInfo(self) -> Tuple[str, int, int]:
        name = C.create_string_buffer(256)
        cus = C.c_int32()
        mem = C.c_uint64()
        check(_lib.lib().gs_context_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)), "gs_context_device_info")
        return name.value.decode(), cus.value, mem.value

    def Dispose(self) -> None:
        if self._h:
            for ch in list(self._children):     # a child must never outlive its context (it holds a raw pointer to it)
                ch.Dispose()
            _lib.lib().gs_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class RenderTarget:
    """The _GaussianSplatRT temporary: RGBA16F, premultiplied, cleared to 0 (GaussianSplatRenderer.cs:194-196)."""

    def __init__(self, ctx: GpuContext, width: int, height: int):
        self.ctx, self.width, self.height = ctx, width, height
        self._h = C.c_void_p()
        check(_lib.lib().gs_target_create(ctx._h, width, height, C.byref(self._h)), "gs_target_create")
        ctx._adopt(self)

    def Clear(self) -> None:
        check(_lib.lib().gs_target_clear(self._h), "gs_target_clear")

    def SetSceneDepth(self, depth: Optional[np.ndarray]) -> None:
        """The camera's depth attachment (GaussianSplatRenderer.cs:195): H x W float32 VIEW depths of the opaque scene, or None."""
        if depth is None:
            check(_lib.lib().gs_target_set_scene_depth(self._h, None, 0), "gs_target_set_scene_depth")
            return
        d = np.ascontiguousarray(depth, np.float32)
        assert d.shape == (self.height, self.width)
        check(_lib.lib().gs_target_set_scene_depth(self._h, d.ctypes.data, 0), "gs_target_set_scene_depth")

    def Download(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.uint16)
        check(_lib.lib().gs_target_download(self._h, out.ctypes.data, out.nbytes), "gs_target_download")
        return out

    def Resolve(self, background=(0.0, 0.0, 0.0, 0.0), want8: bool = True):
        """GaussianComposite.shader onto a constant background: (linear RGBA float32, sRGB RGBA8)."""
        bg = np.asarray(background, np.float32)
        o32 = np.empty((self.height, self.width, 4), np.float32)
        o8 = np.empty((self.height, self.width, 4), np.uint8) if want8 else None
        check(_lib.lib().gs_target_resolve(self._h, _fptr(bg), o32.ctypes.data, o8.ctypes.data if want8 else None), "gs_target_resolve")
        return o32, o8

    def ResolveAsync(self, background=(0.0, 0.0, 0.0, 0.0)) -> None:
        bg = np.asarray(background, np.float32)
        check(_lib.lib().gs_target_resolve(self._h, _fptr(bg), None, None), "gs_target_resolve")

    def SetProfiling(self, enabled: bool) -> None:
        check(_lib.lib().gs_target_set_profiling(self._h, int(bool(enabled))), "gs_target_set_profiling")

    def ResolveTime(self) -> Tuple[float, int]:
        """(mean GPU ms, count) of the resolves since the last call (hipEvents on the context's stream); blocks."""
        ms, cnt = C.c_float(), C.c_int32()
        check(_lib.lib().gs_target_resolve_time(self._h, C.byref(ms), C.byref(cnt)), "gs_target_resolve_time")
        return ms.value, cnt.value

    def Dispose(self) -> None:
        if self._h:
            _lib.lib().gs_target_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class GpuSorting:
    """GpuSorting.cs: stable ascending (uint key, uint payload) device sort.  `SupportResources.Load(count)`
    is the constructor; `Dispatch` sorts device buffers in place, `DispatchHost` round-trips numpy arrays."""

    def __init__(self, ctx: GpuContext, count: int):
        self.ctx, self.count = ctx, count
        self._h = C.c_void_p()
        check(_lib.lib().gs_sorter_create(ctx._h, count, C.byref(self._h)), "gs_sorter_create")
        ctx._adopt(self)

    @property
    def Valid(self) -> bool:
        return bool(self._h)

    def Dispatch(self, keys_dev: int, values_dev: int, count: int, key_bits: int = 32) -> None:
        check(_lib.lib().gs_sorter_dispatch(self._h, C.c_void_p(keys_dev), C.c_void_p(values_dev), count, key_bits), "gs_sorter_dispatch")

    def DispatchHost(self, keys: np.ndarray, values: np.ndarray, key_bits: int = 32):
        k = np.ascontiguousarray(keys, np.uint32).copy()
        v = np.ascontiguousarray(values, np.uint32).copy()
        check(_lib.lib().gs_sorter_sort_host(self._h, k.ctypes.data, v.ctypes.data, len(k), key_bits), "gs_sorter_sort_host")
        return k, v

    def Dispose(self) -> None:
        if self._h:
            _lib.lib().gs_sorter_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class GaussianSplatRenderer:
    """GaussianSplatRenderer component (GaussianSplatRenderer.cs:214-680), render path only."""

    def __init__(self, ctx: GpuContext, asset: Optional[GaussianSplatAsset] = None, transform: Optional[Transform] = None):
        self.ctx = ctx
        self.transform = transform or Transform()
        # serialized fields, :225-244
        self.m_Asset: Optional[GaussianSplatAsset] = asset
        self.m_RenderOrder = 0
        self.m_SplatScale = 1.0
        self.m_OpacityScale = 1.0
        self.m_SHOrder = 3
        self.m_SHOnly = False
        self.m_SortNthFrame = 1
        self.m_RenderMode = RenderMode.Splats                                # :241
        self.m_PointDisplaySize = 3.0                                        # :242
        self.m_Cutouts: Optional[List[Optional[GaussianCutout]]] = None      # :244
        self.m_FrameCounter = 0
        self.blendMode = 0            # 0 exact (fp16 ROP rounding), 1 fast (fp32 accumulate)
        # applied to the native renderer when its resources are created (SetSortMode changes it later).  GSPLAT_SORT_MODE=visible makes the visible-only
        # mode the default of this host layer -- how the whole GPU test suite is run through it (profiles/r05_pytest_gpu_visible.log)
        self.sortMode = SortMode.Visible if os.environ.get("GSPLAT_SORT_MODE", "") == "visible" else SortMode.Full
        self.framesInFlight = max(1, int(os.environ.get("GSPLAT_FRAMES_IN_FLIGHT", "1") or 1))     # SetFramesInFlight
        self._asset_h = C.c_void_p()
        self._r_h = C.c_void_p()
        self._keep: list = []
        self.m_SplatCount = 0
        self.m_PrevAsset = None
        self.m_PrevHash = None
        self.m_Registered = False
        ctx._adopt(self)

    def Dispose(self) -> None:
        self.DisposeResourcesForAsset()

    # -- properties ---------------------------------------------------------------------------------------
    @property
    def asset(self):
        return self.m_Asset

    @property
    def splatCount(self) -> int:
        return self.m_SplatCount

    @property
    def HasValidAsset(self) -> bool:          # :361-368
        a = self.m_Asset
        return (a is not None and a.splatCount > 0 and a.formatVersion == kCurrentVersion and a.posData is not None
                and a.otherData is not None and a.shData is not None and a.colorData is not None)

    @property
    def HasValidRenderSetup(self) -> bool:    # :369
        return bool(self._r_h)

    # -- lifetime -------------------------------------------------------------------------------------------
    def CreateResourcesForAsset(self) -> None:      # :373-424
        if not self.HasValidAsset:
            return
        a = self.m_Asset
        self._keep = []
        desc = make_asset_desc(a, self._keep)
        check(_lib.lib().gs_asset_create(self.ctx._h, C.byref(desc), C.byref(self._asset_h)), "gs_asset_create")
        self._keep = []                              # data was copied to the GPU
        check(_lib.lib().gs_renderer_create(self.ctx._h, self._asset_h, C.byref(self._r_h)), "gs_renderer_create")
        self.m_SplatCount = a.splatCount
        if self.sortMode != SortMode.Full:
            check(_lib.lib().gs_renderer_set_sort_mode(self._r_h, int(self.sortMode)), "gs_renderer_set_sort_mode")
        if self.framesInFlight > 1:
            check(_lib.lib().gs_renderer_set_frames_in_flight(self._r_h, int(self.framesInFlight)), "gs_renderer_set_frames_in_flight")

    def ShareResourcesOf(self, other: "GaussianSplatRenderer") -> None:
        """A second renderer over the SAME device blobs as `other` (no copy; `other` must outlive this one), on this renderer's own context =
        its own stream: frames / views in flight side by side.  In SortMode.Visible tell every such renderer every SortPoints matrix and let
        each draw the frames dealt to it."""
        assert other._asset_h and not self._r_h
        self.m_Asset = other.m_Asset
        self._asset_h, self._asset_borrowed = other._asset_h, True
        check(_lib.lib().gs_renderer_create(self.ctx._h, self._asset_h, C.byref(self._r_h)), "gs_renderer_create")
        self.m_SplatCount = other.m_SplatCount
        self.m_PrevAsset, self.m_PrevHash = other.m_PrevAsset, other.m_PrevHash
        if self.sortMode != SortMode.Full:
            check(_lib.lib().gs_renderer_set_sort_mode(self._r_h, int(self.sortMode)), "gs_renderer_set_sort_mode")

    def DisposeResourcesForAsset(self) -> None:     # :527-565
        l = _lib.lib()
        if self._r_h:
            l.gs_renderer_destroy(self._r_h)
            self._r_h = C.c_void_p()
        if self._asset_h:
            if not getattr(self, "_asset_borrowed", False):
                l.gs_asset_destroy(self._asset_h)
            self._asset_h = C.c_void_p()
            self._asset_borrowed = False
        self.m_SplatCount = 0

    def OnEnable(self) -> None:                     # :475-485
        self.m_FrameCounter = 0
        self.EnsureSorterAndRegister()
        self.CreateResourcesForAsset()
        self.m_PrevAsset = self.m_Asset
        self.m_PrevHash = self.m_Asset.dataHash if self.m_Asset else None

    def OnDisable(self) -> None:                    # :567-577
        self.DisposeResourcesForAsset()
        GaussianSplatRenderSystem.instance.UnregisterSplat(self)
        self.m_Registered = False

    def EnsureSorterAndRegister(self) -> None:      # :461-473
        if not self.m_Registered:
            GaussianSplatRenderSystem.instance.RegisterSplat(self)
            self.m_Registered = True

    def Update(self) -> None:                       # :641-658 asset hot-reload keyed on dataHash
        cur = self.m_Asset.dataHash if self.m_Asset else None
        if self.m_PrevAsset is not self.m_Asset or self.m_PrevHash != cur:
            self.m_PrevAsset, self.m_PrevHash = self.m_Asset, cur
            self.DisposeResourcesForAsset()
            self.CreateResourcesForAsset()

    def ActivateCamera(self, index: int, mainCam: Camera) -> None:     # :660-680
        """Pose `mainCam` as camera `index` of the asset's cameras.json: parented to this renderer's transform with
        localPosition = cam.pos and localRotation = LookRotation(cam.axisZ, cam.axisY), then unparented keeping its world pose,
        scale one.  World position = localToWorld * pos.  World rotation is Unity's Transform.rotation of a child: parent.rotation *
        ScaleMulQuat(parent.localScale, localRotation) -- the local rotation conjugated by the SIGNS of the parent's scale, which
        is what makes the importer's negated y / z camera axes come out looking at the scene under the sample scene's mirrored
        (scale z = -1) splat object (GSTestScene.unity:363-365)."""
        from .camera import look_rotation, quat_to_mat3
        if mainCam is None or self.m_Asset is None or not self.m_Asset.cameras:
            return
        ci = self.m_Asset.cameras[index]
        o2w = self.transform.localToWorldMatrix.astype(np.float64)
        pos = o2w @ np.array([ci.pos[0], ci.pos[1], ci.pos[2], 1.0])
        mainCam.position = tuple(float(v) for v in pos[:3])
        sgn = np.diag(np.sign(np.asarray(self.transform.scale, np.float64)))
        mainCam.rotation = quat_to_mat3(self.transform.rotation) @ (sgn @ look_rotation(ci.axisZ, ci.axisY) @ sgn)

    # -- per frame ------------------------------------------------------------------------------------------
    def FrameParams(self, cam: Camera) -> gs_frame_params:
        return frame_params(cam, self.transform, self.m_SplatScale, self.m_OpacityScale, self.m_SHOrder, self.m_SHOnly)

    def ResetOrder(self) -> None:                   # CSSetIndices, :434-438
        check(_lib.lib().gs_renderer_reset_order(self._r_h), "gs_renderer_reset_order")

    def SortPoints(self, cam: Camera, matrix: Optional[np.ndarray] = None) -> None:     # :612-639
        if matrix is None:
            matrix = self.transform.localToWorldMatrix
        m = np.ascontiguousarray(sort_matrix(cam, matrix), np.float32).reshape(16)
        check(_lib.lib().gs_renderer_sort(self._r_h, _fptr(m)), "gs_renderer_sort")

    def UpdateCutoutsBuffer(self) -> None:          # :742-764 (+ _SplatCutoutsCount, :508)
        arr, n = shader_data_array(self.m_Cutouts, self.transform.localToWorldMatrix)
        check(_lib.lib().gs_renderer_set_cutouts(self._r_h, arr, n), "gs_renderer_set_cutouts")

    def SetDeletedBits(self, bits: Optional[np.ndarray]) -> None:
        """The m_GpuEditDeleted buffer (:269,779): one bit per splat, ceil(N/32) uint32 words; None = no edit buffers
        (_SplatBitsValid = 0).  The editing tools that fill it in the reference (EditDeleteSelected ...) are out of scope."""
        if bits is None:
            check(_lib.lib().gs_renderer_set_deleted_bits(self._r_h, None, 0), "gs_renderer_set_deleted_bits")
            return
        w = np.ascontiguousarray(bits, np.uint32)
        check(_lib.lib().gs_renderer_set_deleted_bits(self._r_h, w.ctypes.data, len(w)), "gs_renderer_set_deleted_bits")

    def CalcViewData(self, cam: Camera) -> None:    # :579-610
        self.UpdateCutoutsBuffer()                  # SetAssetDataOnCS, :507
        p = self.FrameParams(cam)
        check(_lib.lib().gs_renderer_calc_view(self._r_h, C.byref(p)), "gs_renderer_calc_view")

    def Draw(self, cam: Camera, rt: RenderTarget) -> None:     # the DrawProcedural of :156-166
        check(_lib.lib().gs_renderer_set_blend_mode(self._r_h, int(self.blendMode)), "gs_renderer_set_blend_mode")
        check(_lib.lib().gs_renderer_set_render_mode(self._r_h, int(self.m_RenderMode), float(self.m_PointDisplaySize)), "gs_renderer_set_render_mode")
        p = self.FrameParams(cam)
        check(_lib.lib().gs_renderer_draw(self._r_h, C.byref(p), rt._h), "gs_renderer_draw")

    # -- the same three calls with constants the host has already built (a render loop that prepares its per-camera
    #    structs once, like Unity's own camera does, pays three ctypes calls per frame here) -----------------------------
    def SortMatrix(self, cam: Camera, matrix: Optional[np.ndarray] = None) -> np.ndarray:
        if matrix is None:
            matrix = self.transform.localToWorldMatrix
        return np.ascontiguousarray(sort_matrix(cam, matrix), np.float32).reshape(16)

    def SortPointsPrepared(self, m16: np.ndarray) -> None:
        check(_lib.lib().gs_renderer_sort(self._r_h, _fptr(m16)), "gs_renderer_sort")

    def CalcViewDataPrepared(self, p: gs_frame_params) -> None:
        check(_lib.lib().gs_renderer_calc_view(self._r_h, C.byref(p)), "gs_renderer_calc_view")

    def DrawPrepared(self, p: gs_frame_params, rt: RenderTarget) -> None:
        check(_lib.lib().gs_renderer_draw(self._r_h, C.byref(p), rt._h), "gs_renderer_draw")

    # -- parity / measurement hooks ---------------------------------------------------------------------------
    def SetViewBufferMode(self, every_frame: bool) -> None:
        """True: run the reference's full CSCalcViewData every frame (m_GpuView written); False (default): colours only for
        splats that reach the screen, m_GpuView materialised by DownloadView on demand."""
        check(_lib.lib().gs_renderer_set_view_buffer_mode(self._r_h, int(bool(every_frame))), "gs_renderer_set_view_buffer_mode")

    def SetSortMode(self, mode: SortMode) -> None:
        """SortMode.Visible: cull before sorting -- SortPoints only records its matrix and Draw sorts the splats CalcViewData found visible
        (same frame, same order among the drawn splats; include/gsplat_c.h gs_renderer_set_sort_mode for the conditions)."""
        self.sortMode = SortMode(mode)
        if self._r_h:
            check(_lib.lib().gs_renderer_set_sort_mode(self._r_h, int(self.sortMode)), "gs_renderer_set_sort_mode")

    def SetFramesInFlight(self, frames: int) -> None:
        """Frames (or the views of a batch of cameras) in flight INSIDE the library, behind this one renderer: `frames` lanes on streams of their own, dealt
        round-robin at every CalcViewData while SortMode.Visible draws splats -- one frame's latency-bound kernels under another's blend; the calls and the
        frames stay what they are with one frame at a time (gs_renderer_set_frames_in_flight).  1 = off (default)."""
        self.framesInFlight = int(frames)
        if self._r_h:
            check(_lib.lib().gs_renderer_set_frames_in_flight(self._r_h, int(frames)), "gs_renderer_set_frames_in_flight")

    def FramesInFlight(self):
        """(lanes set, True while the next CalcViewData goes to a lane)"""
        f, a = C.c_int32(), C.c_int32()
        check(_lib.lib().gs_renderer_frames_in_flight(self._r_h, C.byref(f), C.byref(a)), "gs_renderer_frames_in_flight")
        return f.value, bool(a.value)

    def SortModeActive(self) -> bool:
        """True while the visible-only path is what the next Draw uses (ABI 8: whenever the mode is set -- whatever the order buffer holds
        is the base its sorts start from)."""
        m, a = C.c_int32(), C.c_int32()
        check(_lib.lib().gs_renderer_sort_mode(self._r_h, C.byref(m), C.byref(a)), "gs_renderer_sort_mode")
        return bool(a.value)

    def SetSortHistoryLimit(self, rows: int) -> None:
        """SortMode.Visible: sort matrices recorded before the library carries them out on all N (2..128, default 128).  Same drawn order either way."""
        check(_lib.lib().gs_renderer_set_sort_history_limit(self._r_h, int(rows)), "gs_renderer_set_sort_history_limit")

    def SortHistory(self):
        """(matrices recorded since the base order, limit, consolidations so far)"""
        rows, limit, cons = C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(_lib.lib().gs_renderer_sort_history(self._r_h, C.byref(rows), C.byref(limit), C.byref(cons)), "gs_renderer_sort_history")
        return rows.value, limit.value, cons.value

    def DownloadVisibleOrder(self) -> np.ndarray:
        """SortMode.Visible: the depth-ordered indices of the visible splats (the visible subsequence of the reference's _OrderBuffer)."""
        out = np.empty(self.m_SplatCount, np.uint32)
        cnt = C.c_uint32()
        check(_lib.lib().gs_renderer_download_visible_order(self._r_h, out.ctypes.data, len(out), C.byref(cnt)), "gs_renderer_download_visible_order")
        return out[:cnt.value].copy()

    def SetProfiling(self, frames: int) -> None:
        """frames = 0 off; > 0: ring of per-frame hipEvent sets, averaged by StageTimes()."""
        check(_lib.lib().gs_renderer_set_profiling(self._r_h, int(frames)), "gs_renderer_set_profiling")

    def SetTileShape(self, tile_w: int = 0, tile_h: int = 0) -> None:
        """The compositor's tile, pixels: 16x16, 32x16 or 32x32; 0, 0 = automatic (per target size).  A performance knob: the frame
        is bit-identical whatever the shape."""
        check(_lib.lib().gs_renderer_set_tile_shape(self._r_h, int(tile_w), int(tile_h)), "gs_renderer_set_tile_shape")

    def TileShape(self, width: int, height: int) -> Tuple[int, int]:
        """(tile_w, tile_h) a draw into a width x height target composites with."""
        w, h = C.c_uint32(), C.c_uint32()
        check(_lib.lib().gs_renderer_tile_shape(self._r_h, int(width), int(height), C.byref(w), C.byref(h)), "gs_renderer_tile_shape")
        return w.value, h.value

    def SetKernelTiming(self, enabled: bool) -> None:
        """With profiling on: Onesweep launches carry their own start/stop timestamps (StageTimes().onesweep_*_kernel_ms); perturbs the stage brackets."""
        check(_lib.lib().gs_renderer_set_kernel_timing(self._r_h, int(bool(enabled))), "gs_renderer_set_kernel_timing")

    def PollPairs(self) -> Tuple[int, int]:
        """Non-blocking: (pair count of the most recent draw that started compositing, pair capacity); count > capacity = that draw was
        truncated (its farthest pairs dropped)."""
        p, c = C.c_uint64(), C.c_uint64()
        check(_lib.lib().gs_renderer_poll_pairs(self._r_h, C.byref(p), C.byref(c)), "gs_renderer_poll_pairs")
        return p.value, c.value

    def ReservePairs(self, n: int) -> None:
        check(_lib.lib().gs_renderer_reserve_pairs(self._r_h, n), "gs_renderer_reserve_pairs")

    def DownloadOrder(self) -> np.ndarray:
        out = np.empty(self.m_SplatCount, np.uint32)
        check(_lib.lib().gs_renderer_download_order(self._r_h, out.ctypes.data, len(out)), "gs_renderer_download_order")
        return out

    def UploadOrder(self, order: np.ndarray) -> None:
        o = np.ascontiguousarray(order, np.uint32)
        check(_lib.lib().gs_renderer_upload_order(self._r_h, o.ctypes.data, len(o)), "gs_renderer_upload_order")

    def DownloadDistances(self) -> np.ndarray:
        out = np.empty(self.m_SplatCount, np.uint32)
        check(_lib.lib().gs_renderer_download_distances(self._r_h, out.ctypes.data, len(out)), "gs_renderer_download_distances")
        return out

    def DownloadView(self) -> np.ndarray:
        out = np.empty(self.m_SplatCount, VIEW_DTYPE)
        check(_lib.lib().gs_renderer_download_view(self._r_h, out.ctypes.data, out.nbytes), "gs_renderer_download_view")
        return out

    def DownloadRasterRecords(self):
        """(recs N x 8 u32, rects N x 2 u32, vis_mask ceil(N/64) u64) as the per-frame calc_view launch left them."""
        n = self.m_SplatCount
        recs = np.empty((n, 8), np.uint32)
        rects = np.empty((n, 2), np.uint32)
        vis = np.empty((n + 63) // 64, np.uint64)
        check(_lib.lib().gs_renderer_download_raster_records(self._r_h, recs.ctypes.data, rects.ctypes.data, vis.ctypes.data),
              "gs_renderer_download_raster_records")
        return recs, rects, vis

    def FrameStats(self) -> gs_frame_stats:
        """Blocks.  Raises GsError(GS_ERR_PAIR_OVERFLOW) if the frame overflowed the pair buffer (it is grown: draw again)."""
        s = gs_frame_stats()
        check(_lib.lib().gs_renderer_frame_stats(self._r_h, C.byref(s)), "gs_renderer_frame_stats")
        return s

    def FrameTimes(self, capacity: int = 4096) -> np.ndarray:
        """GPU duration (ms) of every frame in the profiling ring; call before StageTimes (which resets the ring)."""
        out = np.zeros(capacity, np.float32)
        cnt = C.c_int32()
        check(_lib.lib().gs_renderer_frame_times(self._r_h, _fptr(out), capacity, C.byref(cnt)), "gs_renderer_frame_times")
        return out[:cnt.value].copy()

    def StageTimes(self) -> gs_stage_times:
        t = gs_stage_times()
        check(_lib.lib().gs_renderer_stage_times(self._r_h, C.byref(t)), "gs_renderer_stage_times")
        return t

    def __del__(self):
        try:
            self.DisposeResourcesForAsset()
        except Exception:
            pass


class ViewsInFlight:
    """Several cameras (or consecutive frames) of ONE splat object side by side on one GPU (no counterpart in the reference, which records one camera
    after the other into one command buffer): `lanes` renderers on contexts (= HIP streams) of their own over one copy of the asset, all in
    SortMode.Visible, the views dealt round-robin.  SortPoints is bookkeeping in that mode, so every lane is told every view's matrix in the order the
    reference would sort them and each draws its views from the reference's order buffer: every target holds the bits a single renderer drawing
    the views one after the other would produce (tests/test_gpu_vissort.py), while one view's latency-bound sort / binning kernels run under
    another's VALU-bound blend (-20 % per view with two lanes)."""

    def __init__(self, renderer: "GaussianSplatRenderer", lanes: int = 2):
        assert lanes >= 1 and renderer.HasValidRenderSetup
        renderer.SetSortMode(SortMode.Visible)
        self.lanes = [renderer]
        for _ in range(lanes - 1):
            c = GpuContext(renderer.ctx.device)
            r = GaussianSplatRenderer(c, renderer.m_Asset)
            r.transform, r.sortMode = renderer.transform, SortMode.Visible
            r.ShareResourcesOf(renderer)
            self.lanes.append(r)
        self._targets = {}
        self._next = 0

    def _sync_fields(self):
        src = self.lanes[0]
        for r in self.lanes[1:]:
            for f in ("m_SplatScale", "m_OpacityScale", "m_SHOrder", "m_SHOnly", "m_RenderMode", "m_PointDisplaySize", "m_Cutouts", "blendMode", "transform"):
                setattr(r, f, getattr(src, f))

    def Target(self, lane: int, slot: int, W: int, H: int) -> RenderTarget:
        key = (lane, slot, W, H)
        if key not in self._targets:
            self._targets[key] = RenderTarget(self.lanes[lane].ctx, W, H)
        return self._targets[key]

    def Render(self, cams: Sequence[Camera], sort: bool = True) -> List[RenderTarget]:
        """SortPoints + CalcViewData + Draw for every camera, in order; returns one target per camera (owned by this object, reused by the next call
        with the same sizes: download or resolve them before calling again).  Nothing is synchronised here."""
        self._sync_fields()
        out = []
        for k, cam in enumerate(cams):
            if sort:
                for r in self.lanes:                         # every lane learns every matrix, in the reference's order
                    r.SortPoints(cam)
            li = self._next % len(self.lanes)
            self._next += 1
            r, rt = self.lanes[li], self.Target(li, k, cam.pixelWidth, cam.pixelHeight)
            r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
            out.append(rt)
        return out

    def Synchronize(self) -> None:
        for r in self.lanes:
            r.ctx.Synchronize()

    def Dispose(self) -> None:
        for t in self._targets.values():
            t.Dispose()
        self._targets = {}
        for r in self.lanes[1:]:
            r.DisposeResourcesForAsset()
        self.lanes = self.lanes[:1]


class GaussianSplatRenderSystem:
    """GaussianSplatRenderSystem (GaussianSplatRenderer.cs:17-212): per-camera gather, order, sort, draw, composite."""
    instance: "GaussianSplatRenderSystem" = None  # type: ignore

    def __init__(self):
        self.m_Splats: List[GaussianSplatRenderer] = []
        self.m_ActiveSplats: List[GaussianSplatRenderer] = []
        # A frame whose (tile, splat) pairs overflow the pair buffer drops its farthest pairs; the library notices (and grows
        # the buffer) within the pipeline depth, but THAT frame is truncated.  With strictPairs OnPreCullCamera reads the frame
        # statistics (BLOCKING: it serialises host and GPU) and renders the frame again after an overflow, so what it returns is
        # always complete.  Off by default: a render loop keeps the GPU queue full and a scene's first frames grow the buffer.
        self.strictPairs = False
        # Without strictPairs a truncated frame is still SIGNALLED: after the draws OnPreCullCamera polls (without blocking) what the
        # most recent draws that reached the GPU reported; True = some renderer's latest report exceeds its pair capacity, i.e. a frame
        # of the last few was drawn without its farthest pairs (the library grows the buffer at the next draw).
        self.lastFrameTruncated = False

    def RegisterSplat(self, r: GaussianSplatRenderer) -> None:      # :25-36
        if r not in self.m_Splats:
            self.m_Splats.append(r)

    def UnregisterSplat(self, r: GaussianSplatRenderer) -> None:    # :38-70
        if r in self.m_Splats:
            self.m_Splats.remove(r)

    def GatherSplatsForCamera(self, cam: Camera) -> bool:           # :73-105
        self.m_ActiveSplats = [gs for gs in self.m_Splats if gs.HasValidAsset and gs.HasValidRenderSetup]
        if not self.m_ActiveSplats:
            return False
        w2c = cam.worldToCameraMatrix.astype(np.float64)

        def cam_z(gs):                                              # camTr.InverseTransformPoint(pos).z (Unity: +z forward)
            p = np.append(np.asarray(gs.transform.position, np.float64), 1.0)
            return -float((w2c @ p)[2])
        # orderB.CompareTo(orderA), then posA.z.CompareTo(posB.z): stable sort on (-order, z)
        self.m_ActiveSplats.sort(key=lambda gs: (-gs.m_RenderOrder, cam_z(gs)))
        return True

    def SortAndRenderSplats(self, cam: Camera, rt: RenderTarget) -> None:     # :108-169
        for gs in self.m_ActiveSplats:
            if gs.m_FrameCounter % gs.m_SortNthFrame == 0:
                gs.SortPoints(cam, gs.transform.localToWorldMatrix)
            gs.m_FrameCounter += 1
            gs.CalcViewData(cam)
            gs.Draw(cam, rt)

    def OnPreCullCamera(self, cam: Camera, rt: RenderTarget, background=None):     # :187-211
        """Clear the splat RT, sort/calc/draw every active renderer, then (optionally) composite onto `background`."""
        if not self.GatherSplatsForCamera(cam):
            return None
        rt.Clear()
        self.SortAndRenderSplats(cam, rt)
        self.lastFrameTruncated = False
        for gs in self.m_ActiveSplats:
            if gs.m_RenderMode == RenderMode.Splats:
                pairs, cap = gs.PollPairs()
                self.lastFrameTruncated |= pairs > cap
        if self.strictPairs:
            overflowed = False
            for gs in self.m_ActiveSplats:
                if gs.m_RenderMode != RenderMode.Splats:
                    continue
                try:
                    gs.FrameStats()
                except GsError as e:
                    if e.code != GS_ERR_PAIR_OVERFLOW:
                        raise
                    overflowed = True                         # the buffer has been grown by the call
            if overflowed:                                     # same frame again: the order is already sorted, only the draws repeat
                rt.Clear()
                for gs in self.m_ActiveSplats:
                    gs.CalcViewData(cam)
                    gs.Draw(cam, rt)
            self.lastFrameTruncated = False                    # what is returned is complete
        if background is not None:
            return rt.Resolve(background)
        return None


GaussianSplatRenderSystem.instance = GaussianSplatRenderSystem()
