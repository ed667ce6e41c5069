"""bench.py host logic that needs no GPU: how the per-dispatch records of the run's own rocprofv3 --pmc children are folded into per-launch figures
(the timed region's frames only, template instantiations of a kernel together), and the algorithmic byte formulas of the stages."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_names_are_folded_over_template_arguments():
    b = _bench()
    assert b.short_kernel("void gs::(anonymous namespace)::blend_kernel<0, false, 5, 4>(unsigned int const*, float)") == "blend_kernel"
    assert b.short_kernel("gs::(anonymous namespace)::onesweep_kernel<8, false, 8>(unsigned int const*)") == "onesweep_kernel"
    assert b.short_kernel("void gs::tie_fix_kernel<1>(gsm::AssetView)") == "tie_fix_kernel"


def test_counter_rows_of_the_timed_frames_only():
    """A child runs W warm-up frames and then the K timed ones; per kernel the mean is over the dispatches of the last K frames."""
    b = _bench()
    W, K = 3, 4
    rows = []
    did = 0
    for f in range(W + K):
        for inst, val in (("blend_kernel<0, false, 5, 4>", 100.0 if f < W else 10.0 + f), ("onesweep_kernel<8, false, 8>", 1.0), ("onesweep_kernel<6, false, 16>", 3.0)):
            did += 1
            rows.append({"Kernel_Name": f"void gs::(anonymous namespace)::{inst}(int)", "Counter_Name": "FETCH_SIZE", "Counter_Value": str(val), "Dispatch_Id": str(did)})
    rows.append({"Kernel_Name": "void gs::set_indices_kernel(unsigned int*, unsigned int)", "Counter_Name": "FETCH_SIZE", "Counter_Value": "7", "Dispatch_Id": "0"})
    rows.append({"Kernel_Name": "void gs::x_kernel(int)", "Counter_Name": "OTHER", "Counter_Value": "9", "Dispatch_Id": "999"})
    k = b.fold_counter_rows({}, rows[::-1], "FETCH_SIZE", W + K, K)          # (order in the file does not matter: sorted by dispatch id)
    assert k["blend_kernel"]["dispatches"] == K and abs(k["blend_kernel"]["FETCH_SIZE_KiB"] - (10.0 + (3 + 4 + 5 + 6) / 4.0)) < 1e-9
    assert k["onesweep_kernel"]["dispatches"] == 2 * K and k["onesweep_kernel"]["FETCH_SIZE_KiB"] == 2.0
    assert k["set_indices_kernel"]["dispatches"] == 1 and "x_kernel" not in k
    k = b.fold_counter_rows(k, [dict(r, Counter_Name="SQ_INSTS_VALU") for r in rows[:3 * (W + K)]], "SQ_INSTS_VALU", W + K, K)
    assert "SQ_INSTS_VALU" in k["blend_kernel"] and "FETCH_SIZE_KiB" in k["blend_kernel"]


def test_stage_bytes_follow_the_survey_formulas():
    from unitygaussiansplatting_amd.asset import GaussianSplatAsset  # noqa: F401  (the module imports)
    from common import small_asset
    b = _bench()
    a = small_asset(20000, 5, "Medium")
    n, P, V, W, H = 1_000_000, 900_000, 300_000, 1200, 797
    full = b.stage_bytes(n, P, V, W, H, a, 2, "full")
    vis = b.stage_bytes(n, P, V, W, H, a, 2, "visible")
    assert full["blend"] == vis["blend"] == P * 36 + W * H * 8               # 4 + 32 B per pair, the RGBA16F target written once
    assert full["sort"] == n * 60 and vis["sort"] == V * 68
    assert abs(full["calc_distances"] - n * 8.25) < 1 and vis["calc_distances"] < full["calc_distances"]
    assert b.VALU_SPEC_GWI == 256 * 4 * 2.4 / 2


def test_headline_is_gated_on_the_end_of_orbit_check():
    """`value` reports a visible-only mode only if ITS end-of-orbit check holds: the oracle's replay when the run has one, else the cross-check against the
    full mode; a failed check falls back loudly (the failed modes are returned for the warning), never silently."""
    b = _bench()
    el = {"full": 0.62, "reference_shaped": 0.67, "visible": 0.56, "visible_in_flight": 0.44}
    modes = list(el)
    ok = {"visible_mode": {"ok": True}, "visible_in_flight": {"ok": True}}
    assert b.pick_headline("auto", el, ok, None, modes)[0] == "visible_in_flight"
    m, why, failed = b.pick_headline("auto", el, {"visible_mode": {"ok": True}, "visible_in_flight": {"ok": False}}, None, modes)
    assert m == "visible" and failed == ["visible_in_flight"] and "oracle" in why
    m, why, failed = b.pick_headline("auto", el, {"visible_mode": {"ok": False}, "visible_in_flight": {"ok": False}}, None, modes)
    assert m == "full" and set(failed) == {"visible", "visible_in_flight"} and "FAILED" in why
    # no oracle in the run (N > 1, 50 M splats): the GPU-internal cross-check, every rank's
    cross = {"ok": True, "ok_all_ranks": False, "in_flight": {"ok": True, "ok_all_ranks": True}}
    m, why, failed = b.pick_headline("auto", el, None, cross, modes)
    assert m == "visible_in_flight" and failed == ["visible"] and "cross-check" in why
    assert b.pick_headline("auto", {"full": 0.6}, None, None, ["full"])[0] == "full"
    assert b.pick_headline("visible", el, None, None, modes)[:2] == ("visible", "pinned by --headline")
    assert b.pick_headline("visible", {"full": 0.6}, None, None, ["full"])[0] == "full"
    # the sort logs of two modes describe the same sequence once immediate repetitions are dropped
    import numpy as np
    a, c = np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32) + 1
    assert b.same_sorts([a, a, c], [a, c, c]) and not b.same_sorts([a, c], [c, a]) and not b.same_sorts([a, c, a], [a, c])
