"""Clip-pick / type-pick parity against the reference's OWN per-candidate scores.

`tests/golden/*_traces.npz` (make_golden.py, round 2) hold, for every complete `TensorQuantizer(x)` calibration
recorded from the reference's Python, the [ncand, rows] matrix of `mse_loss` values its final `search_mse` saw and the
per-type sums its type selection compared.  A replacement may pick a different clip candidate than the reference only
where the REFERENCE's scores of the two candidates tie within its own fp32 reduction noise; these helpers
assert exactly that, row by row, instead of allowing a percentage of rows to differ -- with the noise MEASURED on the
reference (round 6: tests/golden/*_traces64.npz) rather than guessed, and a ledger of how much of it was used.
"""
import glob
import os

import numpy as np


def reference_score_noise():
    """max |fp32 score - fp64 score| / score over EVERY score the reference computed while the trace fixtures were recorded
    (tests/golden/*_traces64.npz, make_golden.py round 6: the same float32 element terms, the mean taken in float64).  This
    is the reference's own reduction noise -- how far one of its MSE values can sit from the number it stands for.
    Returns (max, number of scores compared)."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    worst, n = 0.0, 0
    for f64 in sorted(glob.glob(os.path.join(gold, "*_traces64.npz"))):
        d64 = np.load(f64)
        srcs = [np.load(p) for p in (f64.replace("_traces64", "_traces"), f64.replace("_traces64", "")) if os.path.exists(p)]
        for k in d64.files:
            k32 = k[:-2]                                        # "...trace64" -> "...trace"
            src = next((d for d in srcs if k32 in d.files), None)
            assert src is not None, (f64, k)
            a, b = src[k32].astype(np.float64).reshape(d64[k].shape), d64[k]
            ok = np.isfinite(a) & np.isfinite(b) & (b != 0)
            if ok.any():
                worst = max(worst, float((np.abs(a[ok] - b[ok]) / np.abs(b[ok])).max()))
                n += int(ok.sum())
    assert n > 100000, "the *_traces64.npz fixtures are missing"
    return worst, n


REFERENCE_SCORE_NOISE, _N_SCORES = reference_score_noise()
# Two correct implementations may pick different candidates a (the reference: argmin of its noisy fp32 scores s32) and c
# (ours: argmin of sums formed in float64, s64) only when  s32(c) - s32(a) <= |s32(c) - s64(c)| + |s64(a) - s32(a)|, i.e.
# within TWICE the reference's reduction noise.  Round 5 used 2e-5 (20 x SURVEY 8c's 1e-6); measured on the fixtures the
# noise is 3.0e-7, so the rule is 5.9e-7 -- tighter than the survey's.
NEAR_TIE_RTOL = 2.0 * REFERENCE_SCORE_NOISE

# every call of check_alpha_picks / check_type_pick leaves a line here: (key, rows, identical picks, largest certified gap)
LEDGER = []


def ledger_summary(prefix=""):
    """(rows, identical picks, fraction, largest certified gap) over the ledger entries whose key starts with `prefix`."""
    rows = [e for e in LEDGER if e[0].startswith(prefix)]
    n, same = sum(e[1] for e in rows), sum(e[2] for e in rows)
    return n, same, (same / n if n else 1.0), max([e[3] for e in rows] + [0.0])


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
    max_gap = 0.0
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
        assert gap <= NEAR_TIE_RTOL, (key, r, "picked candidate %d, reference %d, reference MSE gap %.3g (allowed %.3g)" % (
            c, pick[r], gap, NEAR_TIE_RTOL))
        max_gap = max(max_gap, float(gap))
    LEDGER.append((str(key), int(ref.size), int(same.sum()), max_gap))
    return same


def check_type_pick(key, got_mode, ref_mode, types, type_sums):
    """A different winning type is acceptable only when the reference's own summed scores of the two tie."""
    if got_mode == ref_mode:
        LEDGER.append(("type:" + str(key), 1, 1, 0.0))
        return
    sums = dict(zip(types, [float(v) for v in type_sums]))
    assert got_mode in sums and ref_mode in sums, (key, got_mode, ref_mode, types)
    gap = abs(sums[got_mode] - sums[ref_mode]) / sums[ref_mode]
    assert gap <= NEAR_TIE_RTOL, (key, "type %s vs reference %s, reference sums differ by %.3g" % (got_mode, ref_mode, gap))
    LEDGER.append(("type:" + str(key), 1, 0, float(gap)))
