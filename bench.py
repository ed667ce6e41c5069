#!/usr/bin/env python3
"""bench.py -- frames/s of the splat render hot path on MI355X (BASELINE.json metric).

One "step" = one frame of the reference's per-camera path for one GaussianSplatRenderer:
    SortPoints (CSCalcDistances + device radix sort)  ->  CalcViewData (CSCalcViewData)
    ->  clear RT  ->  draw all splats front-to-back  ->  composite (GaussianComposite)
on the configuration BASELINE.json quotes its metric on: bicycle-*sized* synthetic scene, 6,131,954 splats,
Medium asset (296 MB), 1200x797 (SURVEY.md section 8d "C2"; the real INRIA model is not available offline).
All inputs are resident in HBM before the timed region.  The camera orbits by 0.25 degrees per frame.

    python bench.py [--gpus N --steps K --warmup W] [--config C2] [--blend exact|fast] [--cpu-baseline auto|off] [--sort-mode both|full|visible]
(--config C2d: C2 with bicycle-like overdraw, ~18 tiles per visible splat -- a non-headline stress of the composite stage.)

Sort modes (round 5): `full` = SortPoints as the reference runs it (all N splats keyed and sorted every frame); `visible` = GS_SORT_VISIBLE,
cull first and sort the splats that are drawn (same frame, same order among the drawn splats: include/gsplat_c.h).  By default BOTH are
measured back to back over the same frames; the headline (`value`, `ms_per_step`, `config.sort_mode`) is the visible-only mode IF this run's
own cross-check holds on its last frame -- the visible order is the visible subsequence of the order buffer the full mode holds and the two
frames are bit-identical -- otherwise the full mode; `modes` carries both.  --headline full|visible pins it.

N > 1: view-parallel, one rank per GPU.  Launched either by the driver (python -m torch.distributed.run ... bench.py --gpus N:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or bare (`python bench.py --gpus N`: the script re-executes
itself under torch.distributed.run on 127.0.0.1 and fails loudly if the box has fewer than N GPUs).  Rank 0 builds the
asset and uploads it; the five blobs are broadcast once with the library's own RCCL communicator (gs_comm_create /
gs_asset_broadcast: ncclBroadcast per blob over xGMI; the 128-byte unique id travels through torch.distributed); every
rank then renders its own camera (azimuth rank*45 deg) with no per-frame collective.  torch.distributed (backend
"nccl" = RCCL) provides the barriers and the max-over-ranks reduction of the timed region.
value = all ranks' splats*frames / max-over-ranks time.  --config C5: the C2 asset, 8 cameras at 1920x1080 dealt
round-robin to the ranks (view k -> rank k % N), a step = every rank renders each of its views once.

Prints ONE JSON line on rank 0 (see DESIGN.md "measurement" for the byte formulas behind `roofline`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0   # the same guide's measured float4-copy rate (79 % of the spec peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C2d", "C3", "C4", "C5"])
    ap.add_argument("--splats", type=int, default=0, help="override the splat count (debugging only; result is labelled)")
    ap.add_argument("--blend", default="exact", choices=["exact", "fast"])
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--sort-nth-frame", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=5, help="the timed region (exactly --steps frames) is run this many times back to back; ms_per_step / value are the MEDIAN region")
    ap.add_argument("--broadcast", action="store_true", help="go through gs_comm_create / gs_asset_broadcast even with one rank")
    ap.add_argument("--sort-mode", default="both", choices=["both", "full", "visible"], help="which depth-sort modes to measure")
    ap.add_argument("--headline", default="auto", choices=["auto", "full", "visible"], help="which measured mode `value` reports (auto: visible if its cross-check holds)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: the rank / view / reduction logic and the JSON schema only (CPU test of the N > 1 path, gloo)")
    return ap.parse_args()


def stage_bytes(n, P, vis, W, H, asset, passes_pair, mode="full", tie_fix=True):
    """ALGORITHMIC bytes per launch of each stage (DESIGN.md "measurement"): the compulsory
    traffic of the algorithm as shipped, every buffer counted once per pass over it."""
    from unitygaussiansplatting_amd.asset import GetVectorSize, GetOtherSizeNoSHIndex, GetColorSize
    b_pos = GetVectorSize(asset.posFormat)
    sh_item = {0: 192, 1: 96, 2: 60, 3: 32}.get(int(asset.shFormat), 96)
    chunk = 64.0 / 256.0 if asset.chunkCount else 0.0
    b_asset = b_pos + GetOtherSizeNoSHIndex(asset.scaleFormat) + GetColorSize(asset.colorFormat) + sh_item + chunk
    out = {
        # pos/rot/scale/colour/chunk of every splat in, 8-B tile rect + 1 visibility bit out; the SH record is read and the
        # 32-B blend record written only for splats that reach the screen (the 40-B m_GpuView record is materialised on demand)
        "calc_view": n * (b_asset - sh_item + 8 + 0.125 + 1.0 / 64) + vis * (sh_item + 32),   # (+ 1 flag byte per wave of 64 splats for the binning)
        "pair_sort": P * 16 * passes_pair + P * 4,               # Onesweep passes over the pairs + tile-range scan of the keys
        "blend": P * (4 + 32) + W * H * 16,                      # pair index + record per pair, RT read + write
        "resolve": W * H * (8 + 16),                             # RGBA16F in, float RGBA out (the 8-bit sRGB image is written only on request)
    }
    if mode == "full":
        out["calc_distances"] = n * (b_pos + chunk + 4)          # CSCalcDistances' arithmetic in index order: pos in, key out (+ the digit histograms)
        out["sort"] = n * (16 * 4 - 4)                           # 4 Onesweep passes x 16 B/key (the first reads prev order + gathered key instead of key + payload;
                                                                 # the last writes only the order: the sorted keys are materialised on demand)
        out["bin"] = n * (4 + 1.0 / 64) + vis * 8 + P * 8        # order + the wave's visibility byte per position, rect per visible splat, (tile, splat) pairs out
    else:
        # GS_SORT_VISIBLE: only the V visible splats are keyed, sorted and binned
        out["calc_distances"] = n / 8.0 + vis * (b_pos + chunk + 8)     # visibility bits of all N; position in, (key, index) out per visible splat
        out["sort"] = vis * 16 * 4 + (vis * 4 if tie_fix else 0)       # 4 plain passes x 16 B/key over V (+ the fix-up's pass over the sorted keys)
        # vis_count: order + the rectangle gather in, rectangle + local offset by position out; vis_offsets: offsets in / out; vis_emit: offset, index,
        # rectangle in, (tile, splat) pairs out
        out["bin"] = vis * (4 + 8 + 12) + vis * 8 + vis * 16 + P * 8
    return out


def respawn_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run on this node."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


VALU_PEAK_GWI = 687.0     # G wave-instructions / s, whole chip, plain v_fma_f32 at 8 waves per SIMD: MEASURED on MI355X, profiles/r05_valu_issue.txt


def load_stored(name, config, P=None):
    """profiles/<name>: figures that need rocprofv3 passes of their own (PMC counters), STORED by the builder's measurement call.  A configuration may
    hold several entries -- "<config>" and "<config>@<label>", collected on different frames (the default run; the driver's `--steps 20 --warmup 5`):
    the one whose pair count is closest to this frame's P is returned (the caller still refuses to use it beyond 2 %)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path)).get("configs", {})
    except Exception:
        return {}
    cands = [v for k, v in d.items() if k == config or k.startswith(config + "@")]
    if not cands:
        return {}
    if P is None:
        return d.get(config, cands[0])
    return min(cands, key=lambda v: abs((v.get("tile_pairs_P") or 0) - P))


def dry_run(args, rank, world):
    """--dry-run: the N > 1 control flow without a GPU -- gloo group, view assignment, max-over-ranks reduction, per-rank gather, the JSON
    line's schema -- so that a CPU test can hold bench.py's multi-rank path (tests/test_parallel.py)."""
    import torch
    import torch.distributed as dist
    from unitygaussiansplatting_amd import parallel, scenes
    cfg = scenes.CONFIGS[args.config]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    num_views = 8 if args.config == "C5" else world
    my_views = parallel.assign_views(num_views, world)[rank]
    el = 1e-3 * (1 + rank) * args.steps                       # a pretend region: rank k takes (1 + k) ms per step
    t = torch.tensor([el], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    views = [None] * world
    if world > 1:
        dist.barrier()
        dist.all_gather(per_rank, t)
        dist.all_gather_object(views, my_views)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    else:
        per_rank, views = [t.clone()], [my_views]
    if rank == 0:
        n = args.splats or cfg.n
        print(json.dumps({"metric": f"Msplats/s rendered (sort+view+composite+resolve), {cfg.label}; ms/frame in ms_per_step", "dry_run": True,
                          "value": round(n * args.steps * num_views / float(t.item()) / 1e6, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(float(t.item()) / args.steps * 1e3, 4),
                          "ms_per_step_per_rank": [round(float(x.item()) / args.steps * 1e3, 4) for x in per_rank], "views_per_rank": views,
                          "higher_is_better": True, "scaling": "strong" if args.config == "C5" else "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": cfg.label, "views": num_views, "rccl_ranks": 0, "host_group": "gloo"},
                          "roofline": None, "roofline_blend": None, "roofline_streaming": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world > 1:
            args.gpus = world
        else:
            raise SystemExit(f"bench.py --gpus {args.gpus}: WORLD_SIZE is {world}; launch {args.gpus} ranks (or run it bare, it spawns them itself)")
    if args.dry_run:
        return dry_run(args, rank, world)

    # torch is plumbing here (barrier, the max-over-ranks reduction, moving the 128-byte RCCL id); its first import on a
    # fresh box takes a minute or two, so it is overlapped with building the synthetic scene.
    holder = {}

    def _imp():
        import torch
        holder["torch"] = torch
    th = threading.Thread(target=_imp)
    th.start()

    from unitygaussiansplatting_amd import camera, creator, scenes
    from unitygaussiansplatting_amd import _lib
    from unitygaussiansplatting_amd._lib import GsError, check
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode

    cfg = scenes.CONFIGS[args.config]
    t_build = time.perf_counter()
    asset = None
    if rank == 0:
        raw = scenes.make_config_splats(cfg, args.splats)
        asset = creator.CreateAssetFromSplatsNative(raw, cfg.quality, name=cfg.key)      # gs_import_encode: same bytes as the numpy importer, ~8x faster
        del raw
    th.join()
    torch = holder["torch"]
    t_build = time.perf_counter() - t_build
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: this box exposes {torch.cuda.device_count()} GPU(s); refusing to report a multi-GPU number")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # The host-side group is GLOO: it carries the barriers, the max-over-ranks reduction and the 128-byte RCCL id.  The ONE RCCL
        # communicator of the process is the library's own (gs_comm_create: ncclCommInitRank), so RCCL sees exactly `world` ranks once.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- asset residency: rank 0 uploads the blobs (gs_asset_create); with N > 1 they are broadcast once through the
    #      library's own RCCL communicator (gs_comm_create + gs_asset_broadcast; only the unique id goes through torch) ----
    from unitygaussiansplatting_amd import parallel
    ctx = GpuContext(local_rank)
    r = GaussianSplatRenderer(ctx, asset)
    r.m_SortNthFrame = args.sort_nth_frame
    t0 = time.perf_counter()
    comm = None
    rccl_ranks = 0
    if world > 1 or args.broadcast:
        uid = [parallel.Comm.UniqueId() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        comm = parallel.Comm(ctx, world, rank, uid[0])
        comm.BroadcastAsset(r, root=0)                 # gs_asset_create on rank 0, ncclBroadcast per blob, gs_renderer_create everywhere
        nr, rk = C.c_int32(), C.c_int32()
        check(_lib.lib().gs_comm_info(comm._h, C.byref(nr), C.byref(rk)), "gs_comm_info")      # what the communicator itself says
        rccl_ranks = int(nr.value)
    else:
        r.CreateResourcesForAsset()
    ctx.Synchronize()
    t_bcast = time.perf_counter() - t0
    r.blendMode = 0 if args.blend == "exact" else 1
    W, H = cfg.width, cfg.height
    n = r.splatCount
    asset_bytes = sum(int(len(getattr(r.m_Asset, nm))) for nm in parallel.BLOB_NAMES if getattr(r.m_Asset, nm) is not None) if rank == 0 else 0

    # views of this rank: one camera per rank (azimuth rank*45 deg), or C5's 8 cameras dealt round-robin
    num_views = 8 if args.config == "C5" else world
    my_views = parallel.assign_views(num_views, world)[rank]
    rts = [RenderTarget(ctx, W, H) for _ in my_views]
    rt = rts[0]

    def cam_at(frame, view=None):
        az = (my_views[0] if view is None else view) * 45.0 + 0.25 * frame
        return camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, az), pixelWidth=W, pixelHeight=H,
                             fieldOfView=cfg.fov_y)

    # Per-camera constants (the sort matrix and gs_frame_params) are built before the loops, as a host engine has them from
    # its camera; the loop body is the five C-ABI calls of one frame.  (Building them in numpy costs ~0.5 ms per frame,
    # which would make this Python harness, not the GPU, the bottleneck.)
    total_frames = max(args.warmup, 1) + args.steps + 2
    prepared = []
    for i in range(total_frames):
        per_view = []
        for v in my_views:
            cam = cam_at(i, v)
            per_view.append((r.SortMatrix(cam), r.FrameParams(cam)))
        prepared.append(per_view)
    r.UpdateCutoutsBuffer()
    check(_lib.lib().gs_renderer_set_blend_mode(r._r_h, int(r.blendMode)), "gs_renderer_set_blend_mode")
    bg = np.asarray((0.0, 0.0, 0.0, 1.0), np.float32)
    bgp = bg.ctypes.data_as(C.POINTER(C.c_float))
    lib_ = _lib.lib()

    def frame(i, cam=None):
        for (m16, p), t in zip(prepared[i], rts):
            if i % r.m_SortNthFrame == 0:
                r.SortPointsPrepared(m16)
            r.CalcViewDataPrepared(p)
            t.Clear()
            r.DrawPrepared(p, t)
            check(lib_.gs_target_resolve(t._h, bgp, None, None), "gs_target_resolve")

    def full_sync():
        ctx.Synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- timed region: exactly K frames, nothing but the frame's C-ABI calls between the barriers ----------------
    per_rank_s = []

    def run_region(first, gather=False):
        full_sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            frame(first + k)
        ctx.Synchronize()                                   # every kernel of the region (the context's own stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64)
            if gather:
                parts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, tmax)
                per_rank_s[:] = [float(x.item()) for x in parts]
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    # a device-to-device copy ceiling measured on this GPU BEFORE the timed regions (SURVEY.md section 8d asks for it next to the 8 TB/s
    # spec; boxes differ by up to 8 %): 512 MiB read + 512 MiB written, 10 times
    copy_ceiling = None
    if rank == 0:
        try:
            src = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
            dst = torch.empty_like(src)
            dst.copy_(src); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize()
            copy_ceiling = round(10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:
            copy_ceiling = None

    def measure(mode):
        """One sort mode: warm-up (sizes the pair buffer: an overflowing frame grows it and is re-run), `repeats` un-instrumented regions of
        exactly K frames, then the same K frames with the per-stage hipEvents, then once more with the Onesweep launches' own timestamps."""
        r.ResetOrder()                                       # both modes start from CSSetIndices' order (the visible-only mode needs it)
        r.SetSortMode(SortMode.Visible if mode == "visible" else SortMode.Full)
        if mode == "visible" and not r.SortModeActive():
            raise SystemExit("bench.py: the visible-only sort mode did not become active")
        fi = 0
        first_frame_ms = None
        for _ in range(max(args.warmup, 1)):
            if fi == 0:                       # the very first frame (identity order: keys in Morton order, cold buffers), timed on its own
                ctx.Synchronize()
                t_first = time.perf_counter()
            frame(fi)
            if fi == 0:
                ctx.Synchronize()
                first_frame_ms = (time.perf_counter() - t_first) * 1e3
            try:
                r.FrameStats()
            except GsError as e:
                if e.code != -6:
                    raise
                frame(fi)
                r.FrameStats()
            fi += 1
        # the orbit over the timed region may need more pairs than the warm-up saw: leave 50 % headroom
        st = r.FrameStats()
        r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))
        # The un-instrumented region -- exactly K frames between barrier + synchronize -- is run `repeats` times back to back over the same
        # frames; ms_per_step and value are the MEDIAN region's (min / max / every region reported next to them): one 14 ms region alone is a
        # coin-flip inside +-4 % on these boxes.
        # Python's cyclic collector is off while frames are timed (as timeit does): with torch imported a full collection walks a few hundred
        # thousand objects -- tens of milliseconds -- and when it strikes at the head of a region, while the queue is still empty, the GPU waits
        # for the host (measured: the fifth 50-frame region of C2 / C3 took 10-40 ms longer in five of six runs; profiles/r04_bench_regions_c2.json).
        gc.collect()
        gc.disable()
        regions = [run_region(fi, gather=True) for _ in range(max(args.repeats, 1))]
        per_rank = list(per_rank_s)
        # ---- the same K frames again with the per-stage hipEvents recorded (14 per frame, on the stream each kernel is
        #      launched on).  The events themselves cost ~50 us of a 0.6 ms frame (every record is a barrier + signal packet
        #      between two kernels), so the headline time comes from the regions above and the per-kernel durations from this one.
        r.SetProfiling(min((args.steps + 1) * len(my_views), 1024))       # a ring: one spare slot so that the first frame's events are not recycled
        rt.SetProfiling(True)                           # the composite is a target method: its own event pairs on the same stream
        elapsed_instr = run_region(fi)
        resolve_ms, _ = rt.ResolveTime()
        rt.SetProfiling(False)
        st = r.FrameStats()                   # raises if the last frame overflowed / a sort spin expired
        frame_ms = r.FrameTimes()             # per-frame GPU durations of the instrumented pass
        stage = r.StageTimes()
        # ---- and a third pass in which every Onesweep launch carries its OWN start / stop timestamps (hipExtLaunchKernelGGL events = the
        #      dispatch packet's completion signal, what rocprofv3 --kernel-trace reports): the streaming kernel's launch duration without
        #      kernel boundaries or event packets.  A pass of its own because the timestamped launches perturb the stage brackets.
        r.SetProfiling(min((args.steps + 1) * len(my_views), 1024))
        r.SetKernelTiming(True)
        run_region(fi)
        r.FrameStats()
        stage_k = r.StageTimes()
        r.SetKernelTiming(False)
        gc.enable()
        r.SetProfiling(0)
        return dict(mode=mode, fi=fi, regions=regions, per_rank=per_rank, elapsed=float(np.median(regions)), elapsed_instr=elapsed_instr, resolve_ms=resolve_ms, st=st,
                    frame_ms=frame_ms, stage=stage, stage_k=stage_k, first_frame_ms=first_frame_ms, active=bool(r.SortModeActive()))

    modes = ["full", "visible"] if args.sort_mode == "both" else [args.sort_mode]
    res = {m: measure(m) for m in modes}

    # ---- the run's own cross-check of the visible-only mode, on the last frame of the orbit: the order it draws from must be the visible
    #      subsequence of the order buffer the reference-shaped mode holds (rebuilt by the library from the same sort matrices when the mode is
    #      switched back), and the two frames must be the same bits
    cross = None
    if "visible" in res:
        last = res["visible"]["fi"] + args.steps - 1
        m16, p = prepared[last][-1]
        r.SortPointsPrepared(m16); r.CalcViewDataPrepared(p); rts[-1].Clear(); r.DrawPrepared(p, rts[-1])
        st_v = r.FrameStats()
        img_v = rts[-1].Download()
        vis_order = r.DownloadVisibleOrder()
        _, _, vbits = r.DownloadRasterRecords()
        mask = np.unpackbits(vbits.view(np.uint8), bitorder="little")[:n].astype(bool)
        r.SetSortMode(SortMode.Full)                         # the library rebuilds the order buffer the reference would hold (one full sort per kept matrix)
        full_order = r.DownloadOrder()
        r.CalcViewDataPrepared(p); rts[-1].Clear(); r.DrawPrepared(p, rts[-1])
        r.FrameStats()
        img_f = rts[-1].Download()
        cross = {"visible_order_is_subsequence_of_full_order": bool(np.array_equal(vis_order, full_order[mask[full_order]])),
                 "frames_bit_identical": bool(np.array_equal(img_v, img_f)), "visible": int(len(vis_order)), "tie_long_runs": int(st_v.tie_long_runs),
                 "note": "GPU-internal: GS_SORT_VISIBLE against the library's own reference-shaped full sort on the orbit's last frame (the oracle check is parity_vs_oracle)"}
        cross["ok"] = cross["visible_order_is_subsequence_of_full_order"] and cross["frames_bit_identical"]
    if world > 1 and cross is not None:
        ok = [None] * world
        dist.all_gather_object(ok, bool(cross["ok"]))
        cross["ok_all_ranks"] = all(ok)
    headline = args.headline
    if headline == "auto":
        headline = "visible" if ("visible" in res and cross is not None and cross.get("ok_all_ranks", cross["ok"])) else ("full" if "full" in res else modes[0])
    if headline not in res:
        headline = modes[0]
    R = res[headline]
    stage, stage_k, st, resolve_ms, frame_ms = R["stage"], R["stage_k"], R["st"], R["resolve_ms"], R["frame_ms"]
    elapsed = R["elapsed"]
    ms_per_step = elapsed / args.steps * 1e3
    msplats = n * args.steps * num_views / elapsed / 1e6

    if rank == 0:
        P = int(st.tile_pairs)
        numTiles = st.tiles_x * st.tiles_y
        passes_pair = 1 if numTiles <= 256 else (2 if numTiles <= 65536 else 3)
        vis = int(st.visible_splats)
        vmode = headline == "visible"
        sb = stage_bytes(n, P, vis, W, H, r.m_Asset, passes_pair, headline)
        times = {"calc_distances": stage.calc_distances_ms, "sort": stage.sort_ms, "calc_view": stage.calc_view_ms,
                 "bin": stage.bin_ms, "pair_sort": stage.pair_sort_ms, "blend": stage.blend_ms, "resolve": resolve_ms}
        stages = {}
        for k, ms in times.items():
            gbs = sb[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            stages[k] = {"ms": round(ms, 4), "alg_MB": round(sb[k] / 1e6, 2), "GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        # Kernels by total time per frame.  The Onesweep launches of a frame (4 depth-sort passes + the pair-sort passes) are one kernel, timed by
        # the launches' OWN start / stop timestamps (gs_stage_times.onesweep_*_kernel_ms = what rocprofv3 --kernel-trace reports; the hipEventRecord
        # brackets around them include kernel boundaries and barrier packets and read 8-10 % high).  The other kernels are one launch (or, for the
        # visible-only binning, three) per stage bracket.
        key_kernel = "visible_keys_kernel" if vmode else "sort_keys_kernel"
        bin_kernel = "vis_count+vis_offsets+vis_emit" if vmode else "bin_emit_kernel"
        launches = {"onesweep_kernel": 4 + int(stage.onesweep_pair_launches), "blend_kernel": 1, "calc_view_kernel": 1, bin_kernel: 3 if vmode else 1, key_kernel: 1}
        sweep_ms = stage_k.onesweep_depth_kernel_ms + stage_k.onesweep_pairs_kernel_ms
        if not sweep_ms > 0:
            sweep_ms = stage.onesweep_depth_ms + stage.onesweep_pairs_ms
        ktime = {"onesweep_kernel": sweep_ms, "blend_kernel": stage.blend_ms, "calc_view_kernel": stage.calc_view_ms, bin_kernel: stage.bin_ms, key_kernel: stage.calc_distances_ms}
        depth_keys = vis if vmode else n
        kbytes = {"onesweep_kernel": depth_keys * (16 * 4 - (0 if vmode else 4)) + P * 16 * passes_pair, "blend_kernel": sb["blend"], "calc_view_kernel": sb["calc_view"],
                  bin_kernel: sb["bin"], key_kernel: sb["calc_distances"]}
        frame_bytes = sum(sb.values())
        tj = load_stored("hbm_traffic.json", args.config + ("_visible" if vmode else ""), P)
        vj = load_stored("valu_insts.json", args.config, P)

        def roof_hbm(k):
            """HBM roofline of kernel k: algorithmic bytes per launch / its mean launch duration against the 8 TB/s spec peak."""
            k_ms = ktime[k] / launches[k]
            k_bytes = kbytes[k] / launches[k]
            ach = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            # `traffic` (HBM bytes per launch from the PMC counters) cannot be collected inside this run -- counter collection needs rocprofv3
            # passes of their own -- so it is the STORED figure of profiles/hbm_traffic.json for this configuration and sort mode, and only if the
            # frame it was collected on had this frame's pair count within 2 % (the bytes scale with P); null otherwise.
            traffic, traffic_source = None, "not collected for this configuration"
            stored_P = tj.get("tile_pairs_P")
            if k in tj.get("kernels", {}):
                if stored_P and abs(stored_P - P) <= 0.02 * P:
                    traffic = tj["kernels"][k]["hbm_bytes_per_launch"]
                    traffic_source = f"STORED, not measured in this run: profiles/hbm_traffic.json ({tj.get('source', '')}; collected at P = {stored_P})"
                else:
                    traffic_source = f"stored figure not used: it was collected at P = {stored_P}, this frame has P = {P}"
            return {"bound": "hbm", "kernel": k, "launches_per_frame": launches[k], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": round(ach / HBM_ACHIEVABLE_GBS, 4),
                    "alg_bytes_per_launch": int(k_bytes), "avg_launch_ms": round(k_ms, 4)}

        def roof(k):
            """A VALU-bound kernel (blend, calc_view: SQ counters, profiles/) is priced against the VALU issue roof MEASURED on this part
            (scripts/probes/valu_issue.hip -> profiles/r05_valu_issue.txt: 687 G wave-instructions / s for plain fp32 VALU on the whole chip); its wave
            instructions per launch are the STORED SQ_INSTS_VALU of profiles/valu_insts.json (counter collection needs a rocprofv3 pass of its own).
            The HBM figures the contract asks for stay in `hbm`."""
            h = roof_hbm(k)
            if k not in ("blend_kernel", "calc_view_kernel"):
                return h
            wi = vj.get("kernels", {}).get(k, {}).get("valu_wave_insts")
            stored_P = vj.get("tile_pairs_P")
            usable = wi and (k != "blend_kernel" or (stored_P and abs(stored_P - P) <= 0.02 * P))
            k_ms = ktime[k] / launches[k]
            ach = (wi / (k_ms * 1e-3) / 1e9) if (usable and k_ms > 0) else None
            return {"bound": "valu", "kernel": k, "launches_per_frame": launches[k], "achieved": round(ach, 1) if ach else None, "peak": VALU_PEAK_GWI,
                    "unit": "G wave-instructions/s", "frac": round(ach / VALU_PEAK_GWI, 4) if ach else None,
                    "valu_wave_insts": int(wi) if usable else None,
                    "valu_source": (f"STORED, not measured in this run: profiles/valu_insts.json ({vj.get('source', '')})" if usable else
                                    "no SQ_INSTS_VALU stored for this configuration / pair count"),
                    "peak_source": "MEASURED on MI355X: scripts/probes/valu_issue.hip, profiles/r05_valu_issue.txt (v_fma_f32, 8 waves per SIMD; v_pk_fma_f32 443, v_exp_f32 / v_fma_mixlo_f16 290: a kernel of such instructions tops out lower)",
                    "traffic": h["traffic"], "avg_launch_ms": h["avg_launch_ms"], "hbm": h}

        # `roofline` = the kernel with the largest total time per frame, as the contract asks; `roofline_blend` and `roofline_streaming` (the
        # bandwidth-type kernel with the largest time: the Onesweep launches) are ALWAYS emitted under fixed keys so that rounds can be compared.
        dom = max(ktime, key=lambda k: ktime[k])
        roofline = roof(dom)
        roofline_blend = roof("blend_kernel")
        stream_dom = max((k for k in ktime if k not in ("blend_kernel", "calc_view_kernel")), key=lambda k: ktime[k])
        roofline_streaming = roof(stream_dom)
        roofline.update({"frames_averaged": int(stage.frames),
                    "instrumented_ms_per_step": round(R["elapsed_instr"] / args.steps * 1e3, 4),
                    "instrumented_frame_gpu_ms": ({"median": round(float(np.median(frame_ms)), 4), "p95": round(float(np.percentile(frame_ms, 95)), 4),
                                                   "max": round(float(frame_ms.max()), 4), "frames": int(len(frame_ms))} if len(frame_ms) else None),
                    "timing": "onesweep_kernel: the launches' own start/stop timestamps (hipExtLaunchKernelGGL events = rocprofv3's kernel durations) from a third pass over the same K frames; `stages` and the other kernels: hipEventRecord brackets on the launching stream from a second pass (the events add ~50 us/frame, so ms_per_step is timed without either)",
                    "onesweep_bracketed_ms_per_frame": round(stage.onesweep_depth_ms + stage.onesweep_pairs_ms, 4),
                    "kernel_ms_per_frame": {k: round(v, 4) for k, v in ktime.items()},
                    # (a step renders every view of this rank once: C5 on one GPU = 8 frames per step)
                    "whole_frame": {"alg_MB": round(frame_bytes / 1e6, 1), "frames_per_step": len(my_views),
                                    "GBps": round(frame_bytes * len(my_views) / (ms_per_step * 1e-3) / 1e9, 1),
                                    "hbm_frac": round(frame_bytes * len(my_views) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}})
        roofline["measured_copy_ceiling_GBps"] = copy_ceiling             # measured before the timed regions

        cpu = None
        parity = None
        if world == 1 and args.cpu_baseline == "auto" and n <= 10_000_000:
            # (C5: the LAST of the 8 views is the one timed on the CPU and checked)
            cpu, parity = cpu_baseline(asset, r, rts[-1], cam_at(R["fi"] + args.steps - 1, my_views[-1]), n, W, H, r.blendMode, "visible" in res)

        def mode_summary(x):
            s_, k_ = x["stage"], x["stage_k"]
            return {"ms_per_step": round(x["elapsed"] / args.steps * 1e3, 4), "value_Msplats_s": round(n * args.steps * num_views / x["elapsed"] / 1e6, 2),
                    "regions_ms_per_step": [round(v / args.steps * 1e3, 4) for v in x["regions"]], "active": x["active"],
                    "tile_pairs_P": int(x["st"].tile_pairs), "visible_splats": int(x["st"].visible_splats),
                    "stages_ms": {"calc_distances": round(s_.calc_distances_ms, 4), "sort": round(s_.sort_ms, 4), "calc_view": round(s_.calc_view_ms, 4),
                                  "bin": round(s_.bin_ms, 4), "pair_sort": round(s_.pair_sort_ms, 4), "blend": round(s_.blend_ms, 4), "resolve": round(x["resolve_ms"], 4)},
                    "onesweep_depth_kernel_ms": round(k_.onesweep_depth_kernel_ms, 4), "onesweep_pairs_kernel_ms": round(k_.onesweep_pairs_kernel_ms, 4),
                    "first_frame_ms": round(x["first_frame_ms"], 3) if x["first_frame_ms"] is not None else None}

        ref_msplats = 6_131_954 / 6.8e-3 / 1e6      # 6.8 ms/frame, RTX 3080 Ti, real bicycle scene
        regions = R["regions"]
        out = {
            "metric": f"Msplats/s rendered (sort+view+composite+resolve), {cfg.label}; ms/frame in ms_per_step",
            "value": round(msplats, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_regions": {"median": round(ms_per_step, 4), "min": round(min(regions) / args.steps * 1e3, 4), "max": round(max(regions) / args.steps * 1e3, 4),
                                    "all": [round(x / args.steps * 1e3, 4) for x in regions],
                                    "note": f"{len(regions)} back-to-back regions of exactly {args.steps} steps each (barrier + synchronize on both sides); value and ms_per_step are the median region"},
            "ms_per_step_per_rank": [round(x / args.steps * 1e3, 4) for x in R["per_rank"]] if R["per_rank"] else None,
            "higher_is_better": True, "scaling": "strong" if args.config == "C5" else "weak",
            # BASELINE.md's only number (6.8 ms/frame, RTX 3080 Ti) is for the REAL bicycle scene; this is the synthetic stand-in of
            # the same size, so the ratio is context, not a like-for-like comparison (config.baseline_note)
            "vs_baseline": round(msplats / num_views / ref_msplats, 3) if args.config == "C2" and not args.splats else None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg.label + (f" [splat count overridden to {n}]" if args.splats else ""),
                       "sort_mode": headline,
                       "sort_mode_note": ("visible = GS_SORT_VISIBLE: cull first, key + sort + bin the V visible splats only, ties ordered by the sort-matrix history -- the same frame "
                                          "and the same order among the drawn splats as the reference's full sort (tests/test_gpu_vissort.py; this run's cross-check: `sort_mode_cross_check`); "
                                          "full = SortPoints as the reference runs it; both are in `modes`"),
                       "splats": n, "resolution": [W, H], "asset_MB": round(asset_bytes / 1e6, 1), "views": num_views, "views_per_rank": len(my_views),
                       "blend": args.blend, "sort_nth_frame": args.sort_nth_frame, "view_buffer": "on demand (gs_renderer_download_view)",
                       "sort_queue_overlap": os.environ.get("GSPLAT_OVERLAP", "0") == "1", "tile_pairs_P": P, "tile": f"{st.tile_w}x{st.tile_h}", "visible_splats": int(st.visible_splats),
                       "parallelism": (f"view-parallel x{world} (one camera per GPU, asset broadcast once by gs_asset_broadcast = ncclBroadcast per blob)" if world > 1 else "single GPU"),
                       "rccl_ranks": rccl_ranks, "host_group": ("gloo" if world > 1 else None),
                       "baseline_note": "vs_baseline = per-view Msplats/s / 901.8 (reference: 6.8 ms/frame on RTX 3080 Ti with the REAL INRIA bicycle, whose overdraw is far higher than this synthetic scene's: context only)"},
            "modes": {m: mode_summary(x) for m, x in res.items()},
            "sort_mode_cross_check": cross,
            "first_frame_ms": round(R["first_frame_ms"], 3) if R["first_frame_ms"] is not None else None,
            "roofline": roofline, "roofline_blend": roofline_blend, "roofline_streaming": roofline_streaming, "stages": stages, "cpu_baseline": cpu, "parity_vs_oracle": parity,
            "setup_s": {"scene_build": round(t_build, 1), "asset_broadcast": round(t_bcast, 3)},
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
    if comm is not None:
        comm.Dispose()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(asset, r, rt, cam, n, W, H, mode, check_visible):
    """The oracle (CPU restatement of the reference shaders, oracle/gs_oracle.cpp) timed on the host cores for one
    whole frame of the same workload, and used as the checker for the GPU frame of the same camera -- in the reference-shaped
    full-sort mode and (check_visible) in the visible-only mode."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from unitygaussiansplatting_amd import camera
    from unitygaussiansplatting_amd.renderer import SortMode
    orc = O.Oracle(asset)
    cores = int(O.lib().gso_num_threads())
    P = r.FrameParams(cam)
    ms = camera.sort_matrix(cam, r.transform.localToWorldMatrix)
    orc.sort(ms)                                    # warm the page cache / thread pool; also the previous-frame order
    t0 = time.perf_counter()
    orc.sort(ms)
    t1 = time.perf_counter()
    orc.calc_view(P)
    t2 = time.perf_counter()
    ref = orc.draw(P, mode)
    t3 = time.perf_counter()
    O.resolve(ref, (0, 0, 0, 1))
    t4 = time.perf_counter()
    total = t4 - t0
    # same camera on the GPU, then compare
    r.SetSortMode(SortMode.Full)
    r.ResetOrder()
    r.SortPoints(cam); r.SortPoints(cam)
    r.CalcViewData(cam)
    rt.Clear()
    r.Draw(cam, rt)
    st_par = r.FrameStats()
    img = rt.Download()
    a, b = O.f16_to_f32(img), O.f16_to_f32(ref)
    d = np.abs(a - b)
    order_equal = bool(np.array_equal(r.DownloadOrder(), orc.order))
    from common import RT_TOL, rt_diff, rt_err
    e = rt_diff(img, ref).max(axis=-1)
    parity = {"order_bit_exact": order_equal, "rt_max_abs": float(d.max()), "rt_mean_abs": float(d.mean()),
              "rt_pixels_bit_equal": float((img == ref).all(axis=2).mean()),
              "rt_max_rel": float(e.max()), "rt_pixels_over_2^-9": int((e > RT_TOL).sum()), "within_bar": bool(rt_err(img, ref) <= RT_TOL),      # every pixel, no outlier allowance
              "tile_pairs_equal": bool(int(st_par.tile_pairs) == int(orc.pairs(P, st_par)))}
    if check_visible:
        # the visible-only mode on the same two sorts: its order against the visible subsequence of the ORACLE's order buffer, its frame against the oracle's
        _, _, vbits = orc.raster_records(P)
        mask = np.unpackbits(vbits.view(np.uint8), bitorder="little")[:n].astype(bool)
        r.ResetOrder()
        r.SetSortMode(SortMode.Visible)
        r.SortPoints(cam); r.SortPoints(cam)
        r.CalcViewData(cam)
        rt.Clear()
        r.Draw(cam, rt)
        r.FrameStats()
        img_v = rt.Download()
        parity["visible_mode"] = {"order_is_visible_subsequence_of_oracle_order": bool(np.array_equal(r.DownloadVisibleOrder(), orc.order[mask[orc.order]])),
                                  "within_bar": bool(rt_err(img_v, ref) <= RT_TOL), "frame_bit_identical_to_full_mode": bool(np.array_equal(img_v, img))}
        r.SetSortMode(SortMode.Full)
    cpu = {"value": round(n / total / 1e6, 3), "unit": "Msplats/s", "cores": cores, "kind": "port",
           "sample": f"1 whole frame of the same workload ({n} splats, {W}x{H}): sort {t1 - t0:.2f}s + view {t2 - t1:.2f}s + "
                     f"composite {t3 - t2:.2f}s + resolve {t4 - t3:.2f}s = {total:.2f}s on {cores} OpenMP threads",
           "ms_per_frame": round(total * 1e3, 1)}
    return cpu, parity


if __name__ == "__main__":
    main()
