"""bench.py host logic that needs no GPU: the STORED counter files (PMC traffic, SQ_INSTS_VALU: rocprofv3 passes of their own) are keyed per set of
frames, and a bench line only pairs them with a frame whose pair count matches."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# last-frame pair counts of the headline workload (C2): the default run (`--steps 50 --warmup 10`) and the driver's (`--steps 20 --warmup 5`)
P_DEFAULT, P_DRIVER = 5207981, 5735437


def test_stored_counters_exist_for_the_default_and_the_driver_frames():
    b = _bench()
    for P in (P_DEFAULT, P_DRIVER):
        t = b.load_stored("hbm_traffic.json", "C2_visible", P)
        v = b.load_stored("valu_insts.json", "C2", P)
        assert t.get("tile_pairs_P") == P and v.get("tile_pairs_P") == P, (P, t.get("tile_pairs_P"), v.get("tile_pairs_P"))
        for k in ("blend_kernel", "calc_view_kernel", "onesweep_kernel"):
            assert t["kernels"][k]["hbm_bytes_per_launch"] > 0
        assert v["kernels"]["blend_kernel"]["valu_wave_insts"] > 5e7
    # the reference-shaped full-sort mode (`--headline full`) on the driver's frames
    assert b.load_stored("hbm_traffic.json", "C2", P_DRIVER).get("tile_pairs_P") == P_DRIVER


def test_entry_selection_is_by_pair_count_and_never_crosses_configurations(tmp_path, monkeypatch):
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"configs": {"C2": {"tile_pairs_P": 100}, "C2@x": {"tile_pairs_P": 200}, "C2d": {"tile_pairs_P": 150}, "C2_visible": {"tile_pairs_P": 151}}},
              open(prof / "f.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.load_stored("f.json", "C2", 149)["tile_pairs_P"] == 100
    assert b.load_stored("f.json", "C2", 151)["tile_pairs_P"] == 200        # "C2d" / "C2_visible" are other configurations, not entries of "C2"
    assert b.load_stored("f.json", "C2d", 1)["tile_pairs_P"] == 150
    assert b.load_stored("f.json", "C2")["tile_pairs_P"] == 100             # no pair count given: the plain key
    assert b.load_stored("f.json", "C3", 1) == {} and b.load_stored("missing.json", "C2", 1) == {}
