"""CPU: the SEMANTICS of GS_SORT_VISIBLE -- the visible subsequence of the reference's order buffer (oracle: every SortPoints is a stable
sort of all N through the previous order, GaussianSplatRenderer.cs:612-639) equals the visible splats sorted by the lexicographic
chain (key under the most recent matrix, ..., key under the oldest, index) with repeated matrices counted once -- the rule
csrc/gs_vissort.hip's tie fix-up and gs_api.hip's matrix history implement (tests/vissort_model.py restates it in numpy).
The GPU side of the same statement is tests/test_gpu_vissort.py."""
import numpy as np
import pytest

import oracle_lib as O
from common import default_camera
from unitygaussiansplatting_amd import camera, creator, scenes
from vissort_model import VisibleSortModel, visible_bits


def tie_heavy_asset(kind: str, n: int = 6000, seed: int = 3, quality: str = "VeryHigh"):
    """Positions arranged so that MANY splats share a sort key (VeryHigh: fp32, exact; Medium: the same geometry on Norm11 lattices per chunk)."""
    raw = scenes.make_splats(n, seed, 3.0)
    rng = np.random.default_rng(seed)
    pos = raw.pos.copy()
    if kind == "lattice":            # columns of equal (x, z): a yaw-only camera ties each column
        g = np.stack(np.meshgrid(np.arange(10), np.arange(30), np.arange(10), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        pos = ((g[rng.integers(0, len(g), n)] - np.array([4.5, 14.5, 4.5], np.float32)) * np.float32(0.25)).astype(np.float32)
    elif kind == "planes":           # two planes facing a camera on the z axis
        pos[:, 2] = np.where(rng.random(n) < 0.5, np.float32(-0.5), np.float32(0.75))
    elif kind == "duplicates":       # every position three times
        base = pos[: n // 3]
        pos = np.concatenate([base, base, base, pos[: n - 3 * (n // 3)]])[:n]
    elif kind == "colocated":        # 150 splats at each of 8 points (runs of > 64 equal keys that NO matrix separates), the rest random
        pts = pos[:8].copy()
        pos[: 8 * 150] = np.repeat(pts, 150, axis=0)[rng.permutation(8 * 150)]
    raw = creator.InputSplatData(pos.astype(np.float32), raw.dc0, raw.sh, raw.opacity, raw.scale, raw.rot)
    return creator.CreateAssetFromSplats(raw, quality, name=f"ties_{kind}_{quality}")


def cams_for(kind: str):
    z_axis = camera.Camera(position=(0.0, 0.0, 6.0), pixelWidth=320, pixelHeight=200)
    yaw = lambda deg: camera.Camera(position=scenes.orbit_eye(6.0, 0.0, deg), pixelWidth=320, pixelHeight=200)
    pitched = lambda deg: camera.Camera(position=scenes.orbit_eye(6.0, 25.0, deg), pixelWidth=320, pixelHeight=200)
    if kind == "lattice":
        return [pitched(10.0), yaw(20.0), yaw(20.0), pitched(40.0), yaw(20.0), yaw(33.0), pitched(10.0)]
    if kind == "planes":
        return [pitched(15.0), z_axis, z_axis, yaw(8.0), z_axis]
    return [default_camera(az=a) for a in (0.0, 5.0, 5.0, 10.0, 0.0, 15.0)]


@pytest.mark.parametrize("kind", ["lattice", "planes", "duplicates", "colocated", "random"])
def test_visible_subsequence_of_the_full_order_is_the_chain_sort(kind):
    a = tie_heavy_asset(kind)
    orc = O.Oracle(a)
    model = VisibleSortModel(a)
    l2w = np.eye(4, dtype=np.float32)
    rng = np.random.default_rng(11)
    longest = 0
    for k, cam in enumerate(cams_for(kind)):
        m = camera.sort_matrix(cam, l2w)
        orc.sort(m)
        model.push(m)
        P = camera.frame_params(cam, camera.Transform())
        orc.calc_view(P)
        for visible in (visible_bits(orc, P), rng.random(a.splatCount) < 0.4):     # the frame's own visibility, and an arbitrary subset
            want = orc.order[visible[orc.order]]
            assert np.array_equal(model.visible_order(visible), want), f"{kind}: frame {k}"
            longest = max(longest, model.longest_run(visible))
    assert len(model.hist) < len(cams_for(kind))                     # the sequences repeat matrices: kept once
    if kind == "lattice":
        assert 5 <= longest                                          # columns of tied splats: the history decides
    if kind in ("planes", "colocated"):
        assert longest > 64                                          # beyond a wave: the workgroup-wide sorting network on the GPU


@pytest.mark.parametrize("kind", ["lattice", "planes", "duplicates", "random"])
@pytest.mark.parametrize("every", [1, 2, 3])
def test_consolidation_is_one_stable_sort_plus_the_chain_fix_up(kind, every):
    """What gs_api.hip's vis_consolidate does every `limit` sorts: the reference's buffer after sorts M_1 .. M_k of a base B == ONE stable
    sort of B by M_k, then every run of equal keys re-ordered by (key under M_k-1, ..., key under M_1, position) -- and the visible chain
    sort on the new base (ranks in it instead of indices) keeps giving the visible subsequence of the reference's buffer."""
    a = tie_heavy_asset(kind)
    orc = O.Oracle(a)
    model = VisibleSortModel(a, depth=10 ** 9)
    l2w = np.eye(4, dtype=np.float32)
    rng = np.random.default_rng(5)
    cams = cams_for(kind) + cams_for(kind)[::-1]
    for k, cam in enumerate(cams):
        m = camera.sort_matrix(cam, l2w)
        orc.sort(m)
        model.push(m)
        if k % every == every - 1:
            model.consolidate()
            assert np.array_equal(model.base, orc.order), f"{kind}: consolidation at sort {k}"
            assert len(model.hist) == 1
        visible = rng.random(a.splatCount) < 0.4
        assert np.array_equal(model.visible_order(visible), orc.order[visible[orc.order]]), f"{kind}: frame {k}"


def test_truncated_history_is_exact_unless_a_dropped_matrix_decides():
    """With only `depth` matrices kept the chain can differ from the reference's only on pairs that stay tied under ALL kept matrices
    (here: a planar scene seen from the same axis with translations only) -- and the model / the library count exactly those."""
    a = tie_heavy_asset("planes", n=3000)
    orc = O.Oracle(a)
    full, short = VisibleSortModel(a, depth=32), VisibleSortModel(a, depth=2)
    l2w = np.eye(4, dtype=np.float32)
    cams = [camera.Camera(position=scenes.orbit_eye(6.0, 25.0, 15.0), pixelWidth=320, pixelHeight=200)] + \
           [camera.Camera(position=(0.0, 0.0, 6.0 + 0.5 * k), target=(0.0, 0.0, -100.0), pixelWidth=320, pixelHeight=200) for k in range(3)]
    vis = np.ones(a.splatCount, bool)
    for cam in cams:
        m = camera.sort_matrix(cam, l2w)
        orc.sort(m); full.push(m); short.push(m)
    assert np.array_equal(full.visible_order(vis), orc.order)
    assert short.dropped == 2 and not np.array_equal(short.visible_order(vis), orc.order)      # the pitched view (dropped) ordered the planes' splats


def bitonic_pairs(L: int):
    """The compare-exchange schedule of gs_vissort.hip's tie_sort_long_run, restated: the all-ascending bitonic network over 2^lp >= L positions --
    per merge level lk a FLIP step (position o of the lower half of every block of 2^lk meets its mirror image in the upper half), then half-cleaners
    of strides 2^(lk-2) .. 1 -- with every pair whose upper position does not exist (>= L) left out.  Yields the steps as lists of (x, y), x < y."""
    lp = 0
    while (1 << lp) < L:
        lp += 1
    half = (1 << (lp - 1)) if lp else 0
    for lk in range(1, lp + 1):
        step = []
        for p in range(half):
            blk, o = p >> (lk - 1), p & ((1 << (lk - 1)) - 1)
            x, y = (blk << lk) + o, (blk << lk) + (1 << lk) - 1 - o
            if y < L:
                step.append((x, y))
        yield step
        for lj in range(lk - 2, -1, -1):
            step = []
            for p in range(half):
                x = ((p >> lj) << (lj + 1)) | (p & ((1 << lj) - 1))
                y = x + (1 << lj)
                if y < L:
                    step.append((x, y))
            yield step


@pytest.mark.parametrize("L", list(range(1, 70)) + [127, 128, 129, 150, 255, 257, 1000, 3001])
def test_the_long_run_network_sorts_any_length(L):
    """Positions beyond the run act as +infinity that never moves (every exchange puts the smaller element at the lower position), so the network
    for the next power of two sorts ANY length with the missing pairs simply skipped; within a step no position occurs twice (the kernel's threads
    exchange concurrently between two barriers)."""
    rng = np.random.default_rng(L)
    for trial in range(3):
        v = rng.permutation(L) if trial else np.arange(L)[::-1].copy()
        for step in bitonic_pairs(L):
            flat = [i for xy in step for i in xy]
            assert len(flat) == len(set(flat)) and all(x < y < L for x, y in step)
            for x, y in step:
                if v[y] < v[x]:
                    v[x], v[y] = v[y], v[x]
        assert np.array_equal(v, np.arange(L)), L
