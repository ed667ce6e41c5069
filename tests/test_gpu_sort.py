"""-m gpu: the Onesweep sort through the C-ABI vs the oracle (bit-exact), plus size-independent properties at full size."""
import numpy as np
import pytest

import oracle_lib as O
from common import default_camera, small_asset
from unitygaussiansplatting_amd import camera
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuSorting

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 255, 256, 4095, 4096, 4097, 8191, 100_000, 1_000_003])
def test_sorter_random_keys(gpu_ctx, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    vals = rng.permutation(n).astype(np.uint32)
    s = GpuSorting(gpu_ctx, n)
    k, v = s.DispatchHost(keys, vals)
    ko, vo = O.sort_pairs(keys, vals)
    assert np.array_equal(k, ko) and np.array_equal(v, vo)
    s.Dispose()


@pytest.mark.parametrize("n,pattern", [(25_000_019, "random"), (12_582_912, "sparse_digits"), (8192 * 32 * 3 + 1, "random")])
def test_sorter_many_partitions_multi_round(gpu_ctx, n, pattern):
    """More partitions than resident workgroups (3052 / 1536 of 8192 keys vs 768 slots): the persistent grid takes several
    tickets per workgroup, the look-back crosses many 32-partition groups and uses INCLUSIVE anchors of finished rounds;
    "sparse_digits" leaves most digits empty in most partitions (their look-back is skipped); the third size ends exactly
    one key past a group boundary."""
    rng = np.random.default_rng(n)
    if pattern == "random":
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    else:
        keys = (rng.integers(0, 3, n).astype(np.uint32) << 24) | (rng.integers(0, 2, n).astype(np.uint32) << 9) | rng.integers(0, 5, n).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    s = GpuSorting(gpu_ctx, n)
    k, v = s.DispatchHost(keys, vals)
    ko, vo = O.sort_pairs(keys, vals)
    assert np.array_equal(k, ko) and np.array_equal(v, vo)
    s.Dispose()


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "low_byte_only", "high_byte_only", "sorted", "reversed", "max_keys"])
def test_sorter_ties_and_adversarial_patterns(gpu_ctx, kind):
    n = 300_001
    rng = np.random.default_rng(5)
    keys = {
        "all_equal": np.full(n, 0x12345678, np.uint32),
        "two_values": rng.integers(0, 2, n).astype(np.uint32) * np.uint32(0x80000001),
        "low_byte_only": rng.integers(0, 256, n).astype(np.uint32),
        "high_byte_only": rng.integers(0, 256, n).astype(np.uint32) << 24,
        "sorted": np.sort(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)),
        "reversed": np.sort(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32))[::-1].copy(),
        "max_keys": np.where(rng.random(n) < 0.5, np.uint32(0xffffffff), np.uint32(0xfffffffe)).astype(np.uint32),   # collide with the tail padding value
    }[kind]
    vals = np.arange(n, dtype=np.uint32)
    s = GpuSorting(gpu_ctx, n)
    k, v = s.DispatchHost(keys, vals)
    perm = O.stable_sort_reference(keys)
    assert np.array_equal(k, keys[perm]) and np.array_equal(v, perm)      # stability: ties keep input order
    s.Dispose()


@pytest.mark.parametrize("bits", [1, 7, 8, 9, 12, 16, 17, 24, 31])
def test_sorter_key_bits(gpu_ctx, bits):
    n = 70_001
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)      # high bits are noise that must be ignored
    vals = np.arange(n, dtype=np.uint32)
    s = GpuSorting(gpu_ctx, n)
    k, v = s.DispatchHost(keys, vals, bits)
    ko, vo = O.sort_pairs(keys, vals, bits)
    assert np.array_equal(k, ko) and np.array_equal(v, vo)
    s.Dispose()


def test_sorter_reuse_many_times(gpu_ctx):
    # the look-back status array is reused with a fresh epoch per pass: many sorts on one sorter, varying sizes
    s = GpuSorting(gpu_ctx, 50_000)
    rng = np.random.default_rng(0)
    for it in range(40):
        n = int(rng.integers(1, 50_001))
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k, v = s.DispatchHost(keys, vals)
        ko, vo = O.sort_pairs(keys, vals)
        assert np.array_equal(k, ko) and np.array_equal(v, vo), it
    s.Dispose()


def test_depth_keys_signs_zeros_denormals(gpu_ctx):
    from test_oracle import fp32_point_asset
    rng = np.random.default_rng(2)
    z = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 10, np.zeros(50, np.float32), -np.zeros(50, np.float32),
                        (rng.standard_normal(200) * 1e-41).astype(np.float32), np.array([np.inf, -np.inf, 3.4e38, -3.4e38], np.float32)])
    rng.shuffle(z)
    a = fp32_point_asset(np.stack([rng.standard_normal(len(z)).astype(np.float32), np.zeros_like(z), z], 1))
    r = GaussianSplatRenderer(gpu_ctx, a)
    r.OnEnable()
    cam = camera.Camera(position=(0, 0, 5), target=(0, 0, 0), pixelWidth=64, pixelHeight=64)
    orc = O.Oracle(a)
    for frame in range(2):
        r.SortPoints(cam)
        orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
        assert np.array_equal(r.DownloadOrder(), orc.order)
        assert np.array_equal(r.DownloadDistances(), orc.keys)
    r.OnDisable()


@pytest.mark.parametrize("quality", ["Medium", "High", "VeryHigh"])
def test_renderer_sort_is_bit_exact_over_frames(gpu_ctx, quality):
    a = small_asset(50_000, 5, quality)
    r = GaussianSplatRenderer(gpu_ctx, a, camera.Transform(position=(0.2, 0.0, -0.1), rotation=(0.0, 0.2588, 0.0, 0.9659)))
    r.OnEnable()
    orc = O.Oracle(a)
    # the first frame sorts from identity order, later frames from the previous order (stateful tie-breaking)
    for frame, az in enumerate([0.0, 0.5, 90.0, 180.0, 180.0]):
        cam = default_camera(az=az)
        r.SortPoints(cam)
        orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
        assert np.array_equal(r.DownloadOrder(), orc.order), frame
    # a non-identity starting order supplied by the caller, then reset
    perm = np.random.default_rng(1).permutation(a.splatCount).astype(np.uint32)
    r.UploadOrder(perm)
    orc.order[:] = perm
    cam = default_camera(az=10.0)
    r.SortPoints(cam)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    assert np.array_equal(r.DownloadOrder(), orc.order)
    # m_GpuSortDistances stays the SORTED keys of the last sort when the order buffer is overwritten afterwards (the last depth pass
    # skips the key write and the keys are rebuilt through order[] on demand: they must be rebuilt before order[] changes)
    sorted_keys = orc.keys.copy()
    r.ResetOrder()
    assert np.array_equal(r.DownloadOrder(), np.arange(a.splatCount, dtype=np.uint32))
    assert np.array_equal(r.DownloadDistances(), sorted_keys)
    r.SortPoints(cam)
    orc.order[:] = np.arange(a.splatCount, dtype=np.uint32)
    orc.sort(camera.sort_matrix(cam, r.transform.localToWorldMatrix))
    r.UploadOrder(perm)
    assert np.array_equal(r.DownloadDistances(), orc.keys) and np.array_equal(r.DownloadOrder(), perm)
    r.OnDisable()


def test_every_sort_test_in_the_other_pass_shape():
    """The sort picks one of two pass shapes by the expected key count (gs_sort.hip: 16 keys per thread / three workgroups per CU up to
    24 M keys, 20 keys per thread / two workgroups per CU above).  The order may not depend on it: this file again, in a process that
    pins shape B for every size (1 key ... 25 M keys, ties, every key width, the renderer's gather pass), and once more with A pinned
    (sizes above the threshold take B by themselves in the run above)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    for shape in ("b", "a"):
        env = dict(os.environ, GSPLAT_SORT_SHAPE=shape, PYTHONPATH=os.path.dirname(here) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "not other_pass_shape"],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"shape {shape}:\n" + r.stdout[-3000:] + r.stderr[-2000:]
