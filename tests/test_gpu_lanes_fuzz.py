"""-m gpu: random API sequences through a renderer with frames in flight inside the library (gs_renderer_set_frames_in_flight) against a renderer that
draws one frame at a time and against the oracle's order buffer.

The hand-written sequences of test_gpu_vissort.py walk the transitions one after the other; this one interleaves them at random (seeded): frames into
three targets in rotation with and without a sort, with and without a resolve, read back at once or only when the target comes round again; the sort
mode switched both ways; UploadOrder / ResetOrder; the number of lanes changed (1 = freed) between any two frames; the history limit, the tile shape,
cutouts and deleted bits changed under the lanes; the order buffer, the drawn order and the frame statistics asked for at any point.  The bar is the
usual one: every frame == the one-at-a-time renderer's bits, every order == the reference's (the oracle sorts all N, stably, on every SortPoints:
GaussianSplatRenderer.cs:612-639, SplatUtilities.compute:69-82, GpuSorting.cs:142-198)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from common import default_camera
from test_cutouts import CUTOUT_SETS
from test_vissort_model import tie_heavy_asset
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd._abi import GS_SORT_VISIBLE
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, RenderTarget, SortMode
from vissort_model import visible_bits

pytestmark = pytest.mark.gpu



def _camera_pool(rng, W, H, n=24):
    """orbit positions (some revisited exactly: a recurring matrix moves to the front of the history), a few dolly steps with a fixed orientation
    (same direction row, another offset) and one camera that sees almost nothing"""
    cams = [default_camera(W, H, az=float(rng.uniform(0, 360)), elev=float(rng.uniform(-25, 35)), radius=float(rng.uniform(4.5, 7.5))) for _ in range(n)]
    cams += [default_camera(W, H, az=40.0, elev=10.0, radius=6.0 + 0.4 * k) for k in range(4)]
    cams.append(default_camera(W, H, az=10.0, elev=5.0, radius=60.0))
    return cams


# GSPLAT_FUZZ_SEEDS=n: a campaign of n further seeds (scripts/r06_call24.sh ran 24 of them once; the suite keeps three)
_CASES = [(1, 40_000, 320, 200), (2, 40_000, 320, 200), (3, 300_000, 640, 400)] + \
         [(10 + k, 40_000 if k % 3 else 200_000, 320 if k % 3 else 640, 200 if k % 3 else 400) for k in range(int(os.environ.get("GSPLAT_FUZZ_SEEDS", "0")))]


@pytest.mark.parametrize("seed,splats,W,H", _CASES)
def test_random_call_sequences_with_frames_in_flight(gpu_ctx, seed, splats, W, H):
    rng = np.random.default_rng(1000 + seed)
    a = tie_heavy_asset("lattice", n=splats, quality="Medium")
    n = a.splatCount
    cur_bits = [None]
    seq, lib = GaussianSplatRenderer(gpu_ctx, a), GaussianSplatRenderer(gpu_ctx, a)
    for x in (seq, lib):
        x.sortMode = SortMode.Visible
        x.OnEnable()
        x.SetSortHistoryLimit(4)
    lib.SetFramesInFlight(2)
    orc = O.Oracle(a)
    rt_seq = RenderTarget(gpu_ctx, W, H)
    rts = [RenderTarget(gpu_ctx, W, H) for _ in range(3)]
    pending = {}                                                 # target index -> (frame number, the bits the one-at-a-time renderer drew, resolved or None)
    cams = _camera_pool(rng, W, H)
    mode = SortMode.Visible
    bits_pool = [None] + [(np.random.default_rng(50 + k).integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint64)
                           & np.random.default_rng(60 + k).integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint64)).astype(np.uint32) for k in range(2)]
    cut_pool = [None, "hole_ellipsoid", "crop_box"]
    bg = (0.1, 0.2, 0.3, 1.0)
    frames = drawn_visible = 0
    last_cam, last_was_visible, plain_frame = None, False, True
    log = []

    def settle(k):
        if k in pending:
            f, want, want_res = pending.pop(k)
            assert np.array_equal(rts[k].Download(), want), f"seed {seed}: frame {f} (target {k}) differs from the one-at-a-time frame; ops: {log[-12:]}"
            if want_res is not None:
                got = rts[k].Resolve(bg)
                assert np.array_equal(got[0], want_res[0]) and np.array_equal(got[1], want_res[1]), f"seed {seed}: frame {f} resolved; ops: {log[-12:]}"

    for step in range(90):
        op = rng.choice(["frame", "mode", "upload", "reset", "lanes", "limit", "order", "visorder", "cutouts", "bits", "tile", "stats"],
                        p=[0.56, 0.05, 0.03, 0.03, 0.07, 0.03, 0.05, 0.05, 0.04, 0.03, 0.03, 0.03])
        log.append(str(op))
        if op == "frame":
            cam = cams[int(rng.integers(len(cams)))] if (last_cam is None or rng.random() > 0.12) else last_cam      # (sometimes the camera does not move)
            k = int(rng.integers(3))
            sort = rng.random() < 0.8
            resolve = rng.random() < 0.5
            settle(k)                                            # the frame drawn into this target earlier must still be there, whole, before it is overwritten
            if sort:
                m = camera.sort_matrix(cam, seq.transform.localToWorldMatrix)
                orc.sort(m); seq.SortPoints(cam); lib.SortPoints(cam)
            seq.CalcViewData(cam); rt_seq.Clear(); seq.Draw(cam, rt_seq)
            want = rt_seq.Download()
            want_res = rt_seq.Resolve(bg) if resolve else None
            lib.CalcViewData(cam); rts[k].Clear(); lib.Draw(cam, rts[k])
            if resolve:
                rts[k].ResolveAsync(bg)
            pending[k] = (frames, want, want_res)
            if rng.random() < 0.4:
                settle(k)
            frames += 1
            last_cam, last_was_visible = cam, mode == SortMode.Visible
            plain_frame = lib.m_Cutouts is None and cur_bits[0] is None      # (what the frame was drawn with: cutouts take effect at CalcViewData)
            drawn_visible += int(last_was_visible)
            log[-1] = f"frame{frames - 1}:t{k}{'s' if sort else ''}{'r' if resolve else ''}"
        elif op == "mode":
            mode = SortMode.Full if mode == SortMode.Visible else SortMode.Visible
            for x in (seq, lib):
                x.SetSortMode(mode)
            lanes, active = lib.FramesInFlight()
            assert active == (lanes > 1 and mode == SortMode.Visible)
            last_was_visible = False
        elif op == "upload":
            perm = rng.permutation(n).astype(np.uint32)
            for x in (seq, lib):
                x.UploadOrder(perm)
            orc.order[:] = perm
            last_was_visible = False                             # (the per-frame questions are about a frame drawn from the order that was just replaced)
        elif op == "reset":
            for x in (seq, lib):
                x.ResetOrder()
            orc.order[:] = np.arange(n, dtype=np.uint32)
            last_was_visible = False
        elif op == "lanes":
            want_lanes = int(rng.choice([1, 2, 3]))
            lib.SetFramesInFlight(want_lanes)
            assert lib.FramesInFlight() == (want_lanes, want_lanes > 1 and mode == SortMode.Visible)
            log[-1] = f"lanes{want_lanes}"
            last_was_visible = False                             # (the per-frame questions go to the lane that drew last: it may be gone)
        elif op == "limit":
            rows = int(rng.integers(2, 7))
            for x in (seq, lib):
                x.SetSortHistoryLimit(rows)
        elif op == "order":
            assert np.array_equal(lib.DownloadOrder(), orc.order), f"seed {seed}: the order buffer differs from the reference's after {log[-12:]}"
        elif op == "visorder":
            if last_was_visible:
                P = lib.FrameParams(last_cam)
                got = lib.DownloadVisibleOrder()
                assert np.array_equal(got, seq.DownloadVisibleOrder()), f"seed {seed}: drawn order differs from the one-at-a-time renderer's after {log[-12:]}"
                if plain_frame:
                    orc.calc_view(P)
                    vis = visible_bits(orc, P)
                    assert np.array_equal(got, orc.order[vis[orc.order]]), f"seed {seed}: drawn order differs from the oracle's visible subsequence after {log[-12:]}"
        elif op == "cutouts":
            name = cut_pool[int(rng.integers(len(cut_pool)))]
            for x in (seq, lib):
                x.m_Cutouts = CUTOUT_SETS[name] if name else None
        elif op == "bits":
            cur_bits[0] = bits_pool[int(rng.integers(len(bits_pool)))]
            for x in (seq, lib):
                x.SetDeletedBits(cur_bits[0])
        elif op == "tile":
            t = [(0, 0), (16, 16), (32, 16), (32, 32)][int(rng.integers(4))]
            for x in (seq, lib):
                x.SetTileShape(*t)
        elif op == "stats":
            if last_was_visible:
                s1, s2 = seq.FrameStats(), lib.FrameStats()
                assert s2.sort_mode == GS_SORT_VISIBLE
                assert (s1.visible_splats, s1.tile_pairs) == (s2.visible_splats, s2.tile_pairs), f"seed {seed}: frame statistics differ after {log[-12:]}"
    for k in range(3):
        settle(k)
    assert np.array_equal(lib.DownloadOrder(), orc.order) and np.array_equal(seq.DownloadOrder(), orc.order)
    assert frames >= 30 and drawn_visible >= 10, (frames, drawn_visible)
    for x in (lib, seq):
        x.OnDisable()
    for t in rts + [rt_seq]:
        t.Dispose()

