"""N > 1 host logic on CPU: world_size-2 gloo run of the view-parallel path (asset broadcast + camera sharding).

There is no CPU product renderer, so each rank renders its assigned views with the ORACLE from the blobs it
*received*; rank 0 compares them with single-process renders.  This covers what bench.py does at N > 1 except the
HIP kernels themselves (covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from common import small_asset
    from unitygaussiansplatting_amd import camera, parallel, scenes
    asset = small_asset(3000, 21, "Medium") if rank == 0 else None
    meta, blobs = parallel.broadcast_asset(asset, torch, dist, rank, world, torch.device("cpu"))
    sizes = meta["sizes"]
    host = [None if t is None else t[:sz].numpy() for t, sz in zip(blobs, sizes)]
    local = parallel.asset_from_meta(meta, host)
    local.Validate()
    assert local.ComputeDataHash() == meta["dataHash"]            # every rank holds the same bytes
    mine = parallel.assign_views(n_views, world)[rank]
    orc = O.Oracle(local)
    tr = camera.Transform()
    for v in mine:
        cam = camera.Camera(position=scenes.orbit_eye(6.0, 10.0, 45.0 * v), pixelWidth=96, pixelHeight=64)
        orc.reset_order()
        orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
        P = camera.frame_params(cam, tr)
        orc.calc_view(P)
        np.save(os.path.join(out_dir, f"view{v}.npy"), orc.draw(P))
    dist.barrier()
    dist.destroy_process_group()


def test_assign_views():
    from unitygaussiansplatting_amd import parallel
    assert parallel.assign_views(8, 8) == [[k] for k in range(8)]
    assert parallel.assign_views(8, 2) == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert parallel.assign_views(3, 4) == [[0], [1], [2], []]
    assert sorted(sum(parallel.assign_views(13, 4), [])) == list(range(13))


def test_two_rank_gloo_broadcast_and_view_sharding(tmp_path):
    import torch.multiprocessing as mp
    n_views, world = 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_views, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from common import small_asset
    from unitygaussiansplatting_amd import camera, scenes
    orc = O.Oracle(small_asset(3000, 21, "Medium"))
    tr = camera.Transform()
    for v in range(n_views):
        cam = camera.Camera(position=scenes.orbit_eye(6.0, 10.0, 45.0 * v), pixelWidth=96, pixelHeight=64)
        orc.reset_order()
        orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
        P = camera.frame_params(cam, tr)
        orc.calc_view(P)
        assert np.array_equal(orc.draw(P), np.load(tmp_path / f"view{v}.npy")), v


def test_bench_gpus_2_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` with no launcher spawns its own ranks; on a box with fewer GPUs it must exit non-zero and
    print no result line (it used to run one rank silently and report n_gpus = 1)."""
    import subprocess
    if os.path.exists("/dev/kfd"):
        import torch
        if torch.cuda.device_count() >= 2:
            pytest.skip("this box really has 2 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "1", "--warmup", "1",
                        "--cpu-baseline", "off"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert '"n_gpus"' not in p.stdout
    assert "GPU" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_rccl_asset_broadcast_single_rank(gpu_ctx):
    """gs_comm_create / gs_asset_broadcast with nranks = 1 (librccl called directly from the library): the asset that comes
    out renders the same frame as one created by gs_asset_create; argument validation of the comm entry points."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import default_camera, small_asset
    from unitygaussiansplatting_amd import _lib, parallel
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget
    a = small_asset(20000, 5, "Medium")
    cam = default_camera()
    frames = []
    for use_comm in (False, True):
        r = GaussianSplatRenderer(gpu_ctx, a)
        comm = None
        if use_comm:
            comm = parallel.Comm(gpu_ctx, 1, 0, parallel.Comm.UniqueId())
            nr, rk = C.c_int32(), C.c_int32()
            assert _lib.lib().gs_comm_info(comm._h, C.byref(nr), C.byref(rk)) == 0 and (nr.value, rk.value) == (1, 0)
            comm.BroadcastAsset(r, root=0)
        else:
            r.CreateResourcesForAsset()
        rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        frames.append((rt.Download(), r.DownloadOrder()))
        r.DisposeResourcesForAsset()
        rt.Dispose()
        if comm is not None:
            out = C.c_void_p()
            assert _lib.lib().gs_asset_broadcast(comm._h, None, 0, C.byref(out)) == -1          # the root must pass an asset
            assert _lib.lib().gs_asset_broadcast(comm._h, None, 3, C.byref(out)) == -1          # root out of range
            comm.Dispose()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    bad = C.c_void_p()
    uid = (C.c_uint8 * 128).from_buffer_copy(parallel.Comm.UniqueId())
    assert _lib.lib().gs_comm_create(gpu_ctx._h, 2, 5, uid, C.byref(bad)) == -1                  # rank out of range
