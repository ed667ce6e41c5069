"""Multi-view rendering across GPUs: one camera per GPU, shared splat buffers broadcast once.

The reference is single-GPU, single-process; this layer is new.  The path shards
naturally by *view*: every GPU holds a replica of the immutable asset blobs and runs the whole per-camera path
(sort -> view data -> composite) on its own camera, so there is NO per-frame collective.  The only exchange is
at load time: rank `root` broadcasts the five GaussianSplatAsset blobs (pos / other / color / sh / chunk) to
every rank.  Two transports:
  * `Comm` -- the product path: the library's own RCCL communicator (gs_comm_* / gs_asset_broadcast of include/gsplat_c.h,
    ncclBroadcast per blob on the context's stream); only the 128-byte unique id travels over a host channel;
  * `broadcast_asset` -- torch.distributed tensors adopted without a copy (gs_asset_desc.memory_kind = 1): what the
    world-size-2 gloo tests exercise on CPU boxes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .asset import ColorFormat, GaussianSplatAsset, SHFormat, VectorFormat, kCurrentVersion

BLOB_NAMES = ("posData", "otherData", "colorData", "shData", "chunkData")


class Comm:
    """gs_comm: one RCCL rank bound to a GpuContext.  `uid` = Comm.UniqueId() of ONE rank, handed to all ranks by the host."""

    def __init__(self, ctx, nranks: int, rank: int, uid: bytes):
        from . import _lib
        from ._lib import check
        assert len(uid) == 128
        self.ctx, self.nranks, self.rank = ctx, nranks, rank
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(_lib.lib().gs_comm_create(ctx._h, nranks, rank, buf, C.byref(self._h)), "gs_comm_create")
        ctx._adopt(self)

    @staticmethod
    def UniqueId() -> bytes:
        from . import _lib
        from ._lib import check
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().gs_comm_unique_id(buf), "gs_comm_unique_id")
        return bytes(buf)

    def BroadcastAsset(self, renderer, root: int = 0) -> None:
        """CreateResourcesForAsset across ranks: on `root` the renderer's m_Asset (host blobs) is uploaded as usual, then the
        device blobs are broadcast; every rank ends with an asset handle + renderer, and m_Asset describing it."""
        from . import _lib
        from ._abi import make_asset_desc
        from ._lib import check
        l = _lib.lib()
        src = C.c_void_p()
        root_error = None
        if self.rank == root:
            try:
                keep: list = []
                desc = make_asset_desc(renderer.m_Asset, keep)
                check(l.gs_asset_create(self.ctx._h, C.byref(desc), C.byref(src)), "gs_asset_create")
            except Exception as e:            # still enter the collective (with no asset): every rank then fails together instead of
                root_error = e                # waiting for a header that never comes
                src = C.c_void_p()
        out = C.c_void_p()
        rc = l.gs_asset_broadcast(self._h, src, root, C.byref(out))
        if root_error is not None:
            raise root_error
        check(rc, "gs_asset_broadcast")
        renderer._asset_h = out
        info = (C.c_uint32 * 6)()
        check(l.gs_asset_info(out, info), "gs_asset_info")
        if self.rank != root:
            sizes = (C.c_uint64 * 5)()
            ptrs = (C.c_void_p * 5)()
            check(l.gs_asset_device_blobs(out, ptrs, sizes), "gs_asset_device_blobs")
            renderer.m_Asset = asset_from_meta(dict(splatCount=int(info[0]), posFormat=int(info[1]), scaleFormat=int(info[2]), colorFormat=int(info[3]),
                                                    shFormat=int(info[4]), formatVersion=kCurrentVersion,
                                                    dataHash="", name="broadcast", boundsMin=(0, 0, 0), boundsMax=(0, 0, 0), sizes=[int(x) for x in sizes]))
        check(l.gs_renderer_create(self.ctx._h, renderer._asset_h, C.byref(renderer._r_h)), "gs_renderer_create")
        renderer.m_SplatCount = int(info[0])
        renderer.m_PrevAsset, renderer.m_PrevHash = renderer.m_Asset, (renderer.m_Asset.dataHash if renderer.m_Asset else None)

    def Dispose(self) -> None:
        if self._h:
            from . import _lib
            _lib.lib().gs_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


def assign_views(num_views: int, world_size: int) -> List[List[int]]:
    """View k is rendered by rank k % world_size (C5: 8 views -> 8 GPUs, one each)."""
    return [[v for v in range(num_views) if v % world_size == rk] for rk in range(world_size)]


def asset_meta(asset: GaussianSplatAsset) -> dict:
    return dict(splatCount=asset.splatCount, posFormat=int(asset.posFormat), scaleFormat=int(asset.scaleFormat),
                colorFormat=int(asset.colorFormat), shFormat=int(asset.shFormat), formatVersion=asset.formatVersion,
                dataHash=asset.dataHash, name=asset.name, boundsMin=tuple(asset.boundsMin), boundsMax=tuple(asset.boundsMax),
                sizes=[0 if getattr(asset, nm) is None else int(len(getattr(asset, nm))) for nm in BLOB_NAMES])


def asset_from_meta(meta: dict, blobs: Optional[Sequence[Optional[np.ndarray]]] = None) -> GaussianSplatAsset:
    a = GaussianSplatAsset(splatCount=meta["splatCount"], posFormat=VectorFormat(meta["posFormat"]),
                           scaleFormat=VectorFormat(meta["scaleFormat"]), shFormat=SHFormat(meta["shFormat"]),
                           colorFormat=ColorFormat(meta["colorFormat"]), formatVersion=meta["formatVersion"],
                           dataHash=meta["dataHash"], name=meta["name"], boundsMin=meta["boundsMin"], boundsMax=meta["boundsMax"])
    if blobs is not None:
        for nm, b in zip(BLOB_NAMES, blobs):
            setattr(a, nm, b)
    else:                                   # placeholders so HasValidAsset-style checks see "data lives on the GPU"
        for nm, sz in zip(BLOB_NAMES, meta["sizes"]):
            setattr(a, nm, np.zeros(0, np.uint8) if sz else None)
    return a


def broadcast_asset(asset: Optional[GaussianSplatAsset], torch, dist, rank: int, world: int, device, root: int = 0):
    """Returns (meta, [5 uint8 tensors on `device` or None]).  On `root`, `asset` must be given; elsewhere it is ignored.
    One broadcast per blob (296 MB for the bicycle-sized Medium asset; 5 large messages, not thousands of small ones)."""
    meta_box = [asset_meta(asset) if rank == root else None]
    if dist is not None and world > 1:
        dist.broadcast_object_list(meta_box, src=root)
    meta = meta_box[0]
    blobs = []
    for nm, sz in zip(BLOB_NAMES, meta["sizes"]):
        if sz == 0:
            blobs.append(None)
            continue
        # +16 bytes of zero padding: the 2-byte-aligned dword stitching of the decoder may touch the dword after the last record
        t = torch.zeros(sz + 16, dtype=torch.uint8, device=device)
        if rank == root:
            src = np.ascontiguousarray(getattr(asset, nm), dtype=np.uint8)
            t[:sz].copy_(torch.from_numpy(src))
        if dist is not None and world > 1:
            dist.broadcast(t, src=root)
        blobs.append(t)
    return meta, blobs


def attach_device_asset(renderer, meta: dict, blobs) -> None:
    """CreateResourcesForAsset for blobs that already live on the renderer's GPU (borrowed, not copied)."""
    from . import _lib
    from ._abi import gs_asset_desc
    from ._lib import check
    d = gs_asset_desc()
    d.splat_count = meta["splatCount"]
    d.pos_format, d.scale_format = meta["posFormat"], meta["scaleFormat"]
    d.color_format, d.sh_format = meta["colorFormat"], meta["shFormat"]
    d.memory_kind = 1
    for nm, t, sz in zip(("pos", "other", "color", "sh", "chunk"), blobs, meta["sizes"]):
        setattr(d, nm + "_data", None if t is None else t.data_ptr())
        # pos / other / sh: a borrowed blob must DECLARE the readable bytes behind its last record (gs_asset_create checks for >= 4);
        # broadcast_asset's tensors carry 16 bytes of zero padding for exactly that
        pad = 16 if nm in ("pos", "other", "sh") and t is not None and int(t.numel()) >= sz + 16 else 0
        setattr(d, nm + "_size", 0 if t is None else sz + pad)
    renderer._device_blobs = blobs            # keep the tensors alive as long as the renderer
    check(_lib.lib().gs_asset_create(renderer.ctx._h, C.byref(d), C.byref(renderer._asset_h)), "gs_asset_create")
    check(_lib.lib().gs_renderer_create(renderer.ctx._h, renderer._asset_h, C.byref(renderer._r_h)), "gs_renderer_create")
    renderer.m_SplatCount = meta["splatCount"]
    renderer.m_PrevAsset, renderer.m_PrevHash = renderer.m_Asset, meta["dataHash"]
