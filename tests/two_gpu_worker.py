"""Worker of tests/test_parallel.py::test_two_gpus_broadcast_and_render_their_own_views: one rank = one process = one GPU.
usage: two_gpu_worker.py <rank> <nranks> <uid hex> <out .npy>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def view_camera(k: int):
    from common import default_camera
    return default_camera(az=45.0 * k, elev=12.0)


def main():
    rank, nranks, uid, out = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4]
    from common import small_asset
    from unitygaussiansplatting_amd import parallel
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget
    ctx = GpuContext(rank)                                       # one GPU per rank
    asset = small_asset(20000, 5, "Medium") if rank == 0 else None      # only the root holds the host blobs
    r = GaussianSplatRenderer(ctx, asset)
    comm = parallel.Comm(ctx, nranks, rank, uid)
    comm.BroadcastAsset(r, root=0)                               # gs_asset_create on the root, ncclBroadcast per blob over xGMI, gs_renderer_create everywhere
    cam = view_camera(rank)
    rt = RenderTarget(ctx, cam.pixelWidth, cam.pixelHeight)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    r.FrameStats()
    np.save(out, rt.Download())
    r.DisposeResourcesForAsset(); rt.Dispose(); comm.Dispose()


if __name__ == "__main__":
    main()
