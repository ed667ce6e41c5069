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


@pytest.mark.parametrize("config,want_views", [("C5", [[0, 2, 4, 6], [1, 3, 5, 7]]), ("C2", [[0], [1]])])
def test_bench_two_ranks_dry_run_under_gloo(config, want_views):
    """bench.py's N > 1 control flow without GPUs (`--dry-run`: no device work): the self-spawn under torch.distributed.run on 127.0.0.1,
    the gloo host group, the per-rank view assignment, the max-over-ranks reduction and the schema of the N > 1 JSON line."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--config", config, "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 4
    assert out["views_per_rank"] == want_views
    assert out["ms_per_step_per_rank"] == [1.0, 2.0] and out["ms_per_step"] == 2.0      # the pretend regions: rank k takes (1 + k) ms per step; MAX over ranks
    assert out["scaling"] == ("strong" if config == "C5" else "weak")
    assert out["config"]["host_group"] == "gloo" and "rccl_ranks" in out["config"]
    for k in ("metric", "value", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "roofline_blend", "roofline_streaming", "cpu_baseline"):
        assert k in out


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


@pytest.mark.gpu
def test_two_gpus_broadcast_and_render_their_own_views(gpu_ctx, tmp_path):
    """The multi-GPU path itself, on the day a node with >= 2 GPUs runs this suite (skipped on the 1-GPU boxes): two processes, one GPU each,
    gs_comm_create(nranks = 2) + gs_asset_broadcast over RCCL / xGMI, each rank renders its own camera -- the frame of rank k == the frame a
    single GPU renders from camera k."""
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (no multi-GPU node was available in any round: no scaling curve was measured)")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import small_asset
    from two_gpu_worker import view_camera
    from unitygaussiansplatting_amd import parallel
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget
    uid = parallel.Comm.UniqueId().hex()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_gpu_worker.py"), str(k), "2", uid, str(tmp_path / f"view{k}.npy")], env=env)
             for k in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a = small_asset(20000, 5, "Medium")
    for k in range(2):
        cam = view_camera(k)
        r = GaussianSplatRenderer(gpu_ctx, a)
        r.OnEnable()
        rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
        r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
        assert np.array_equal(np.load(tmp_path / f"view{k}.npy"), rt.Download()), f"view {k}"
        r.OnDisable(); rt.Dispose()


@pytest.mark.gpu
def test_asset_replica_through_the_receive_half_of_the_broadcast(gpu_ctx):
    """gs_asset_replicate = the non-root branch of gs_asset_broadcast (header -> padded allocations -> blobs -> view) with a
    device copy in place of ncclBroadcast: the replica (on a second context of the same GPU) renders the same frame, owns its
    blobs (the source can be destroyed first) and reports the source's description."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import default_camera, small_asset
    from unitygaussiansplatting_amd import _lib
    from unitygaussiansplatting_amd._lib import check
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget
    l = _lib.lib()
    cam = default_camera()
    for quality in ("Medium", "VeryHigh"):                       # with and without a chunk blob
        a = small_asset(20000, 5, quality)
        r0 = GaussianSplatRenderer(gpu_ctx, a)
        r0.CreateResourcesForAsset()
        rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
        r0.SortPoints(cam); r0.CalcViewData(cam); rt.Clear(); r0.Draw(cam, rt)
        want, want_order = rt.Download(), r0.DownloadOrder()
        ctx2 = GpuContext(0)
        rep = C.c_void_p()
        check(l.gs_asset_replicate(ctx2._h, r0._asset_h, C.byref(rep)), "gs_asset_replicate")
        i0, i1 = (C.c_uint32 * 6)(), (C.c_uint32 * 6)()
        check(l.gs_asset_info(r0._asset_h, i0), "info"); check(l.gs_asset_info(rep, i1), "info")
        assert list(i0) == list(i1)
        r0.DisposeResourcesForAsset(); rt.Dispose()              # the replica does not borrow from the source
        r1 = GaussianSplatRenderer(ctx2, a)
        r1._asset_h = rep
        check(l.gs_renderer_create(ctx2._h, rep, C.byref(r1._r_h)), "gs_renderer_create")
        r1.m_SplatCount = a.splatCount
        r1.m_PrevAsset, r1.m_PrevHash = a, a.dataHash
        rt2 = RenderTarget(ctx2, cam.pixelWidth, cam.pixelHeight)
        r1.SortPoints(cam); r1.CalcViewData(cam); rt2.Clear(); r1.Draw(cam, rt2)
        assert np.array_equal(rt2.Download(), want) and np.array_equal(r1.DownloadOrder(), want_order)
        r1.DisposeResourcesForAsset(); rt2.Dispose()
    bad = C.c_void_p()
    assert l.gs_asset_replicate(None, None, C.byref(bad)) == -1


@pytest.mark.gpu
def test_torch_transport_at_world_one_adopts_borrowed_blobs(gpu_ctx):
    """The documented second transport (torch.distributed tensors adopted without a copy, memory_kind = 1) at world = 1 on the GPU:
    broadcast_asset + attach_device_asset give the frame of a plain upload (the borrowed-blob size rule of gs_asset_create
    used to reject exactly these tensors)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import default_camera, small_asset
    from unitygaussiansplatting_amd import parallel
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget
    cam = default_camera()
    for quality in ("Medium", "VeryHigh", "Low"):
        a = small_asset(20000, 5, quality)
        frames = []
        for borrowed in (False, True):
            r = GaussianSplatRenderer(gpu_ctx, a)
            if borrowed:
                meta, blobs = parallel.broadcast_asset(a, torch, None, 0, 1, torch.device("cuda:0"))
                torch.cuda.synchronize()
                parallel.attach_device_asset(r, meta, blobs)
            else:
                r.CreateResourcesForAsset()
            rt = RenderTarget(gpu_ctx, cam.pixelWidth, cam.pixelHeight)
            r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
            frames.append(rt.Download())
            r.DisposeResourcesForAsset(); rt.Dispose()
        assert np.array_equal(frames[0], frames[1]), quality
