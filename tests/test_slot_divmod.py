"""bin_emit turns output slot o of a tile rectangle w tiles wide into (row, column) = (o // w, o % w) with an fp32 reciprocal
estimate and one correction step instead of a u32 division (gs_raster.hip: slot_to_xy).  The device's v_rcp_f32 is accurate
to 1 ulp, not correctly rounded, so this CPU model checks the claim the kernel relies on for EVERY admissible reciprocal: with
the correctly rounded 1/w perturbed by -1, 0 and +1 ulp, floor(fl(o * rcp)) is within 1 of o // w for all o < 2^20 and the
correction makes it exact.  (Slots >= 2^20 take the integer division in the kernel.)"""
import numpy as np
import pytest


def slot_to_xy_model(o: np.ndarray, w: np.ndarray, ulp_shift: int):
    inv = (np.float32(1.0) / w.astype(np.float32)).astype(np.float32)
    if ulp_shift:
        inv = np.nextafter(inv, np.float32(np.inf if ulp_shift > 0 else -np.inf)).astype(np.float32)
    q = (o.astype(np.float32) * inv).astype(np.float32).astype(np.int64)          # v_cvt_u32_f32 truncates
    r = o.astype(np.int64) - q * w.astype(np.int64)
    lo = r < 0
    hi = r >= w
    return q - lo + hi, r + np.where(lo, w, 0) - np.where(hi, w, 0), q


@pytest.mark.parametrize("ulp_shift", [-1, 0, 1])
def test_reciprocal_divmod_is_exact_below_2_pow_20(ulp_shift):
    rng = np.random.default_rng(11 + ulp_shift)
    # every width up to 4096 against adversarial slots (multiples of w and their neighbours, the top of the range) + random pairs
    ws = np.arange(1, 4097, dtype=np.int64)
    cases_o, cases_w = [], []
    for k in (1, 2, 3, 255, 256, 257, 4095, 65535):
        base = ws * k
        for d in (-1, 0, 1):
            cases_o.append(base + d); cases_w.append(ws)
    top = (1 << 20) - 1
    cases_o.append(np.full_like(ws, top)); cases_w.append(ws)
    cases_o.append((top // ws) * ws); cases_w.append(ws)
    cases_o.append((top // ws) * ws - 1); cases_w.append(ws)
    w_r = rng.integers(1, 1 << 16, 2_000_000)
    o_r = rng.integers(0, 1 << 20, 2_000_000)
    cases_o.append(o_r); cases_w.append(w_r)
    o = np.concatenate(cases_o); w = np.concatenate(cases_w)
    keep = (o >= 0) & (o < (1 << 20))
    o, w = o[keep], w[keep]
    ty, tx, q_est = slot_to_xy_model(o, w, ulp_shift)
    assert np.abs(q_est - o // w).max() <= 1, "the estimate must be within one of the quotient"
    assert np.array_equal(ty, o // w) and np.array_equal(tx, o % w)
