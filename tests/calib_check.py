"""Clip-pick / type-pick parity against the reference's OWN per-candidate scores.

`tests/golden/*_traces.npz` (make_golden.py, round 2) hold, for every complete `TensorQuantizer(x)` calibration
recorded from the reference's Python, the [ncand, rows] matrix of `mse_loss` values its final `search_mse` saw and the
per-type sums its type selection compared.  A replacement may pick a different clip candidate than the reference only
where the REFERENCE's scores of the two candidates tie within its own fp32 reduction noise; these helpers
assert exactly that, row by row, instead of allowing a percentage of rows to differ.
"""
import numpy as np

NEAR_TIE_RTOL = 2e-5      # relative gap of two reference MSEs below which either candidate is an acceptable pick


def ratios_of(lo, hi, step):
    """fl32(i * 0.01) for the reference's `range(lo, hi, step)` (AQ:298-300 / OQ:206-207)."""
    return np.asarray([np.float32(i * 0.01) for i in range(int(lo), int(hi), int(step))], dtype=np.float32)


def reference_pick(trace):
    """Index of the reference's pick per row: best starts at 1e10, strict '<', ascending candidates (AQ:299-306)."""
    ncand, rows = trace.shape
    best = np.full(rows, np.float32(1e10), dtype=np.float32)
    pick = np.full(rows, -1, dtype=np.int64)
    for c in range(ncand):
        better = trace[c] < best
        pick[better] = c
        best[better] = trace[c][better]
    return pick, best


def check_alpha_picks(key, got_alpha, ref_alpha, trace, ratios, xmax_rtol=2e-6):
    """Every row: the candidate we picked is the reference's, or one whose REFERENCE score is within NEAR_TIE_RTOL of
    the reference's best.  Also pins x_max (abs-max / 3-sigma rule): alpha / ratio must agree with the reference's.
    Returns the boolean mask of rows whose pick is identical to the reference's."""
    got = np.asarray(got_alpha, dtype=np.float32).reshape(-1)
    ref = np.asarray(ref_alpha, dtype=np.float32).reshape(-1)
    trace = np.asarray(trace, dtype=np.float32)
    assert trace.shape == (ratios.size, ref.size), (key, trace.shape, ratios.size, ref.size)
    pick, best = reference_pick(trace)
    same = np.zeros(ref.size, dtype=bool)
    for r in range(ref.size):
        if pick[r] < 0:                       # no candidate qualified: alpha stays x_max
            assert np.isclose(got[r], ref[r], rtol=xmax_rtol), (key, r, got[r], ref[r])
            same[r] = True
            continue
        xmax = np.float64(ref[r]) / np.float64(ratios[pick[r]])
        if xmax == 0.0 or not np.isfinite(xmax):
            assert got[r] == ref[r] or (np.isnan(got[r]) and np.isnan(ref[r])), (key, r)
            same[r] = True
            continue
        rel = np.abs(np.float64(got[r]) / xmax - ratios.astype(np.float64))
        c = int(np.argmin(rel))
        tol = (4 * xmax_rtol + 3e-7) * ratios[c]        # 3e-7: the reference's alpha is itself a rounded product
        assert rel[c] <= tol, (key, r, "alpha is not x_max times a candidate ratio", got[r], xmax)
        if c == pick[r]:
            same[r] = True
            continue
        gap = (np.float64(trace[c, r]) - np.float64(best[r])) / np.float64(best[r])
        assert gap <= NEAR_TIE_RTOL, (key, r, "picked candidate %d, reference %d, reference MSE gap %.3g" % (c, pick[r], gap))
    return same


def check_type_pick(key, got_mode, ref_mode, types, type_sums):
    """A different winning type is acceptable only when the reference's own summed scores of the two tie."""
    if got_mode == ref_mode:
        return
    sums = dict(zip(types, [float(v) for v in type_sums]))
    assert got_mode in sums and ref_mode in sums, (key, got_mode, ref_mode, types)
    gap = abs(sums[got_mode] - sums[ref_mode]) / sums[ref_mode]
    assert gap <= NEAR_TIE_RTOL, (key, "type %s vs reference %s, reference sums differ by %.3g" % (got_mode, ref_mode, gap))
