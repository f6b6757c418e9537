"""Shared helpers for the parity tests."""
import numpy as np


def ulp_diff(got, exp):
    """|got - exp| in units of the fp32 spacing at exp; NaN==NaN counts as 0."""
    got = np.asarray(got, np.float32)
    exp = np.asarray(exp, np.float32)
    both_nan = np.isnan(got) & np.isnan(exp)
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    sp = np.spacing(np.abs(np.where(np.isfinite(exp), exp, 1)).astype(np.float32)).astype(np.float64)
    out = err / sp
    out[both_nan] = 0
    return out


def assert_bits_equal(got, exp, what=""):
    got = np.ascontiguousarray(got)
    exp = np.ascontiguousarray(exp)
    assert got.shape == exp.shape, f"{what}: shape {got.shape} vs {exp.shape}"
    if got.dtype == np.float32:
        g, e = got.view(np.uint32), exp.astype(np.float32).view(np.uint32)
        nan_ok = np.isnan(got) & np.isnan(exp)
        bad = (g != e) & ~nan_ok & ~((got == 0) & (exp == 0))  # +0 / -0 compare equal (torch.eq semantics)
    else:
        bad = got != exp
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} words differ, first at {np.argwhere(bad)[0]}: {got[tuple(np.argwhere(bad)[0])]} vs {exp[tuple(np.argwhere(bad)[0])]}"


def assert_ulp(got, exp, max_ulp, what="", abs_floor=0.0):
    d = ulp_diff(got, exp)
    if abs_floor:
        d = np.where(np.abs(np.asarray(got, np.float64) - np.asarray(exp, np.float64)) <= abs_floor, 0, d)
    assert np.nanmax(d) <= max_ulp, f"{what}: max {np.nanmax(d):.1f} ulp > {max_ulp}"
    assert (np.isnan(got) == np.isnan(exp)).all(), f"{what}: NaN pattern differs"
