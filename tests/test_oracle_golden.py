"""The CPU oracle (oracle/antq_oracle.c + .py) against the golden vectors captured from the
reference's own Python (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from calib_check import check_alpha_picks, check_type_pick, ratios_of
from conftest import golden


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


# ---------------------------------------------------------------- grids
def test_ant_grid_restatement(oracle):
    g = golden("ant_grids.npz")
    n = 0
    for k in g.files:
        inv = k.startswith("INVALID_")
        t, b, s = (k[8:] if inv else k).split("_")
        bit, signed = int(b[1:]), s == "s"
        if inv:
            with pytest.raises(Exception):
                oracle.ant_grid(t, bit, signed)
            continue
        with np.errstate(all="ignore"):
            mine = oracle.ant_grid(t, bit, signed)
        if t == "apot":
            # torch.sort is not stable: the relative order of +0.0 / -0.0 is unspecified
            assert np.array_equal(mine, g[k]), k
        else:
            assert same_f32(mine, g[k]), k
        n += 1
    assert n > 100


def test_olive_grid_restatement(oracle):
    g = golden("olive_grids.npz")
    for k in g.files:
        if k.startswith("INVALID_"):
            continue
        t, b, s = k.split("_")
        fn = {"int": oracle.olive_int_value, "flint": oracle.olive_flint_value, "outlier": oracle.olive_outlier_value}[t]
        assert same_f32(fn(int(b[1:]), s == "s"), g[k]), k


# ---------------------------------------------------------------- a1: the scan
def test_nearest_matches_reference_traces(oracle):
    gr = golden("ant_grids.npz")
    n = golden("ant_nearest.npz")
    keys = sorted({k[:-2] for k in n.files if k.endswith("_x") and not k.startswith("f64_")})
    assert len(keys) >= 10
    for k in keys:
        z, idx = oracle.nearest(n[k + "_x"], gr[k])
        assert same_f32(z, n[k + "_z"]), k
        assert np.array_equal(idx.astype(np.int16), n[k + "_idx"]), k
    z, idx = oracle.nearest(n["f64_flint_b4_s_x"], gr["flint_b4_s"].astype(np.float64))
    assert z.dtype == np.float64 and np.array_equal(z[~np.isnan(z)], n["f64_flint_b4_s_z"][~np.isnan(z)])
    assert np.array_equal(idx.astype(np.int16), n["f64_flint_b4_s_idx"])


def test_nearest_documented_rules(oracle):
    """SURVEY A.3 worked examples."""
    g = golden("ant_grids.npz")["flint_b4_s"]
    z, idx = oracle.nearest(np.float32([0.3125, -0.3125, 7.5, -7.5, np.nan, np.inf, 2e5]), g)
    assert z.tolist()[:4] == [0.625, 0.0, 10.0, -5.0]
    assert idx.tolist() == [9, 8, 15, 1, -1, -1, -1]
    assert z[4:].tolist() == [0.0, 0.0, 0.0]
    o = golden("olive_nearest.npz")
    z, idx = oracle.nearest(np.float32([40, -40]), o["flint_b4_s_grid"])
    assert z.tolist() == [48.0, -48.0] and idx.tolist() == [22, 21]


def test_olive_nearest(oracle):
    o = golden("olive_nearest.npz")
    for k in ("int_b4_s", "int_b4_u", "flint_b4_s", "flint_b4_u"):
        z, idx = oracle.nearest(o[k + "_x"], o[k + "_grid"])
        assert same_f32(z, o[k + "_z"]), k
        assert np.array_equal(idx.astype(np.int16), o[k + "_idx"]), k


# ---------------------------------------------------------------- a4: ANT _forward
def _ant_forward_cases():
    f = golden("ant_forward.npz")
    for k in f.files:
        if k.endswith("_out") and not k.startswith(("c0_", "g16_")):
            yield k[:-4]


def test_ant_forward(oracle):
    f = golden("ant_forward.npz")
    gr = golden("ant_grids.npz")
    n = 0
    for case in _ant_forward_cases():
        sname, t, s, pc = case.rsplit("_", 3)
        x = f[sname + "_x"]
        if s == "u":
            x = np.abs(x)
        x2 = x.reshape(x.shape[0], -1)
        grid = gr["%s_b4_%s" % (t, s)]
        out, idx = oracle.forward(x2, f[case + "_alpha"], grid)
        assert same_f32(out, f[case + "_out"]), case
        assert np.array_equal(idx.reshape(-1).astype(np.int16), f[case + "_idx"]), case
        n += 1
    assert n == 48


def test_ant_forward_int8_per_tensor_and_group16(oracle):
    f = golden("ant_forward.npz")
    gr = golden("ant_grids.npz")
    x = f["c0_x"].reshape(64, -1)
    out, idx = oracle.forward(x, f["c0_int8_pt_alpha"], gr["int_b8_s"])
    assert same_f32(out, f["c0_int8_pt_out"])
    assert np.array_equal(idx.reshape(-1).astype(np.int16), f["c0_int8_pt_idx"])
    # group-16 = per-channel on x.view(-1, 16)
    xg = f["g16_x"].reshape(-1, 16)
    out, idx = oracle.forward(xg, f["g16_flint_alpha"], gr["flint_b4_s"])
    assert same_f32(out, f["g16_flint_out"])


def test_ste_is_identity_on_fixtures(oracle):
    """fl(fl(q-d)+d) == q on realistic data, so out == fl(q*s)."""
    f = golden("ant_forward.npz")
    gr = golden("ant_grids.npz")["flint_b4_s"]
    x = f["w4x768_x"]
    alpha = f["w4x768_flint_s_pc_alpha"]
    out, idx = oracle.forward(x, alpha, gr)
    s = (alpha / np.float32(10.0)).astype(np.float32)[:, None]
    assert same_f32(out, (gr[idx] * s).astype(np.float32))


# ---------------------------------------------------------------- a5: OliVe _forward + OVP
def test_olive_forward(oracle):
    f = golden("olive_forward.npz")
    og = golden("olive_grids.npz")
    n = 0
    for k in f.files:
        if not k.endswith("_out"):
            continue
        case = k[:-4]
        name, t, pc, mode = case.rsplit("_", 3)
        signed = not name.startswith("a6x50")
        s = "s" if signed else "u"
        normal = og["%s_b4_%s" % (t, s)]
        grid = normal if mode == "noout" else np.concatenate([normal, og["outlier_b4_%s" % s]])
        x = f[name + "_x"]
        x2 = x.reshape(x.shape[0], -1)
        out, idx = oracle.forward(x2, f[case + "_alpha"], grid, gmax=float(normal.max()), ovp=(mode == "ovp"))
        assert same_f32(out, f[case + "_out"]), case
        ref_idx = f[case + "_idx"]
        got = idx.reshape(-1)
        keep = got != oracle.IDX_VICTIM          # the reference trace holds the pre-masking index
        assert np.array_equal(got[keep].astype(np.int16), ref_idx[keep]), case
        n += 1
    assert n == 34


def test_ovp_rules_hand_case(oracle):
    """pairs (2k,2k+1): odd victim iff even is outlier; even victim iff odd outlier and even not; odd numel wraps."""
    og = golden("olive_grids.npz")
    normal, outl = og["flint_b4_s"], og["outlier_b4_s"]
    grid = np.concatenate([normal, outl])
    # alpha = 32 -> scale 1: inputs are already in the grid domain
    x = np.float32([[100, 1, 100, -100, 1, 100, 2, 4, 5]])    # 9 elements: last one wraps to element 0
    out, idx = oracle.forward(x, np.float32([32.0]), grid, gmax=32.0, ovp=True)
    assert out.tolist() == [[96.0, 0.0, 96.0, 0.0, 0.0, 96.0, 2.0, 4.0, 0.0]]
    assert idx[0, 1] == idx[0, 3] == idx[0, 4] == idx[0, 8] == oracle.IDX_VICTIM


# ---------------------------------------------------------------- a9/a10: mse + clip search
@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_search_mse_traces(oracle, tree):
    if tree == "ant":
        s = golden("ant_search.npz")
        gr = golden("ant_grids.npz")
        cases = [(n, t, None) for t in ("int", "flint", "pot") for n in ("w", "a", "au")]
    else:
        s = golden("olive_search.npz")
        og = golden("olive_grids.npz")
        cases = [(n, t, m) for t in ("int", "flint") for n in ("w", "a") for m in ("ovp", "noout")]
    for name, t, m in cases:
        if tree == "ant":
            key = "%s_%s" % (name, t)
            x = s["a_x"] if name.startswith("a") else s["w_x"]
            if name == "au":
                x = np.abs(x)
            per_row = name == "w"
            grid = gr["%s_b4_%s" % (t, "u" if name == "au" else "s")]
            gmax, ovp, lb, ub, step = float(grid.max()), False, 75, 150, 1
            xmax = np.abs(x).max(1) if per_row else np.abs(x).max(keepdims=True).reshape(1)
        else:
            key = "%s_%s_%s" % (name, t, m)
            x = s[name + "_x"]
            per_row = name == "w"
            normal = og["%s_b4_s" % t]
            ovp = m == "ovp"
            grid = np.concatenate([normal, og["outlier_b4_s"]]) if ovp else normal
            gmax, lb, ub, step = float(normal.max()), 75, 250, 2
            if ovp:
                xd = x.astype(np.float64)
                if per_row:
                    mean, std = xd.mean(1), xd.std(1, ddof=1)
                else:
                    mean, std = xd.mean(keepdims=True).reshape(1), np.array([xd.std(ddof=1)])
                xmax = np.maximum(np.abs(mean + 3 * std), np.abs(mean - 3 * std))
            else:
                xmax = np.abs(x).max(1) if per_row else np.abs(x).max(keepdims=True).reshape(1)
        xmax = xmax.astype(np.float32)
        best, alpha, trace = oracle.search_mse(x, xmax, lb, ub, step, grid, gmax, ovp, per_row)
        ref = s[key + "_trace"]
        assert trace.shape == ref.shape, key
        np.testing.assert_allclose(trace, ref, rtol=2e-5, atol=1e-12, err_msg=key)
        np.testing.assert_allclose(best.sum(), s[key + "_best_sum"], rtol=2e-5)
        # chosen candidate: identical unless two candidates tie within the reduction tolerance
        ref_alpha = s[key + "_alpha"].reshape(-1)
        close = np.isclose(alpha, ref_alpha, rtol=1e-6)
        if not close.all():
            srt = np.sort(ref, axis=0)
            near_tie = (srt[1] - srt[0]) <= 4e-5 * srt[0]
            assert near_tie[~close].all(), key


_TYPE_ORDER = ("int", "flint", "pot", "float", "float1", "float2", "float3", "float4", "apot")


@pytest.mark.parametrize("which", ["wide", "long"])
def test_full_calibration_wide_fixture_set(oracle, which):
    """a11 + a12 on the CPU: type selection (AQ:328-415, incl. the -floatN searches on float_value(1)), the final clip
    search and the forward, against 90 calibrations recorded from the reference (ant_select_wide.npz).  A pick may
    differ from the reference's only where the reference's OWN scores tie (ant_select_wide_traces.npz)."""
    # ("long": rows of 1024 elements, the shapes on which the HIP path selects the type on ONE read of the tensor)
    sel, tr = golden("ant_select_%s.npz" % which), golden("ant_select_%s_traces.npz" % which)
    n_alpha = n_same = 0
    for k in [str(v) for v in sel["keys"]]:
        name, mode, b, win = k.split("__")
        bit, (lo, up) = int(b[1:]), map(int, win.split("_"))
        x = sel[name + "__x"]
        per_row = name == "w"
        signed = True if per_row else bool(x.min() < 0)                    # update_signed, AQ:72
        assert signed == bool(sel[k + "__signed"]), k
        xmax = (np.abs(x).max(1) if per_row else np.abs(x).max(keepdims=True).reshape(1)).astype(np.float32)
        ref_mode = str(sel[k + "__mode"])
        if bit > 6:
            mode, lo = "int", 95                                              # AQ:482-483, :296-297
        elif mode.startswith("ant-"):
            scores, types = [], []
            for t in _TYPE_ORDER:
                if ("-" + t) not in mode:
                    continue
                g = oracle.ant_float_value(bit, signed, 1) if (t.startswith("float") and t != "float") else oracle.ant_grid(t, bit, signed)
                best, _, _ = oracle.search_mse(x, xmax, lo, up, 1, g, float(np.max(g)), False, per_row)
                scores.append(float(best.astype(np.float32).sum()))
                types.append(t)
            np.testing.assert_allclose(scores, tr[k + "__type_sums"], rtol=2e-5, err_msg=k)
            mode = types[int(np.argmin(scores))]                              # first smallest, like argsort(mse)[0] (AQ:411-412)
            check_type_pick(k, mode, ref_mode, types, tr[k + "__type_sums"])
            mode = ref_mode
        assert mode == ref_mode, k
        grid = oracle.ant_grid(mode, bit, signed)
        g_ref = sel[k + "__grid"]
        assert np.array_equal(grid, g_ref), k                                  # (-0 == +0: apot order)
        gmax = float(np.max(grid))
        best, alpha, trace = oracle.search_mse(x, xmax, lo, up, 1, grid, gmax, False, per_row)
        ref_trace = tr[k + "__trace"]
        np.testing.assert_allclose(trace, ref_trace.reshape(trace.shape), rtol=2e-5, atol=1e-12, err_msg=k)
        ref_alpha = sel[k + "__alpha"].reshape(-1)
        same = check_alpha_picks(k, alpha, ref_alpha, ref_trace, ratios_of(lo, up, 1), xmax_rtol=0.0)
        n_alpha += same.size
        n_same += int(same.sum())
        assert np.array_equal(np.asarray(alpha, np.float32).reshape(-1)[same], ref_alpha[same]), k
        # the forward for ALL rows, on the reference's alpha
        out, _ = oracle.forward(x, ref_alpha if per_row else ref_alpha[:1], grid, gmax, False)
        assert f32_same_rows(out, sel[k + "__out"].reshape(x.shape)), k
    assert n_same >= 0.97 * n_alpha, (n_same, n_alpha)       # (informative: how often the noise flips a pick at all)


@pytest.mark.parametrize("which", ["wide", "long"])
def test_olive_full_calibration_wide_fixture_set(oracle, which):
    """OliVe a10-a12 on the CPU (OQ:189-292): 3-sigma x_max, step-2 clip search with the victim rule inside the loss,
    int / flint selection, final forward -- against olive_select_wide.npz (bits 3..8, outliers on / off, odd numel),
    picks checked against the reference's own scores (olive_select_wide_traces.npz)."""
    sel, tr = golden("olive_select_%s.npz" % which), golden("olive_select_%s_traces.npz" % which)
    n_alpha = n_same = 0
    for k in [str(v) for v in sel["keys"]]:
        name, mode, b, win, om = k.split("__")
        bit, (lo, up) = int(b[1:]), map(int, win.split("_"))
        ovp = om == "ovp"
        x = sel[name + "__x"]
        per_row = name == "w"
        signed = True if per_row else bool(x.min() < 0)
        assert signed == bool(sel[k + "__signed"]), k
        if ovp:
            xd = x.astype(np.float64)
            if per_row:
                mean, std = xd.mean(1), xd.std(1, ddof=1)
            else:
                mean, std = xd.mean(keepdims=True).reshape(1), np.array([xd.std(ddof=1)])
            xmax = np.maximum(np.abs(mean + 3 * std), np.abs(mean - 3 * std)).astype(np.float32)
        else:
            xmax = (np.abs(x).max(1) if per_row else np.abs(x).max(keepdims=True).reshape(1)).astype(np.float32)
        outl = oracle.olive_outlier_value(bit, signed)
        assert np.array_equal(outl, sel[k + "__outliers"]), k

        def full(t):
            n = oracle.olive_grid(t, bit, signed)
            return n, (np.concatenate([n, outl]) if ovp else n)

        ref_mode = str(sel[k + "__mode"])
        if bit > 6:
            mode = "int"
        elif mode.startswith("ant-"):
            scores = []
            for t in ("int", "flint"):
                n, g = full(t)
                best, _, _ = oracle.search_mse(x, xmax, lo, up, 2, g, float(n.max()), ovp, per_row)
                scores.append(float(best.astype(np.float32).sum()))
            np.testing.assert_allclose(scores, tr[k + "__type_sums"], rtol=5e-5, err_msg=k)
            mode = ("int", "flint")[int(np.argmin(scores))]
            check_type_pick(k, mode, ref_mode, ("int", "flint"), tr[k + "__type_sums"])
            mode = ref_mode
        assert mode == ref_mode, k
        normal, grid = full(mode)
        assert np.array_equal(normal, sel[k + "__grid"]), k
        gmax = float(normal.max())
        best, alpha, trace = oracle.search_mse(x, xmax, lo, up, 2, grid, gmax, ovp, per_row)
        ref_alpha = sel[k + "__alpha"].reshape(-1)
        # (the 3-sigma x_max carries the reduction noise of mean / std: alpha agrees to a few ulp, not bit for bit)
        same = check_alpha_picks(k, alpha, ref_alpha, tr[k + "__trace"], ratios_of(lo, up, 2), xmax_rtol=2e-6)
        n_alpha += same.size
        n_same += int(same.sum())
        # the forward (victims included) for the WHOLE tensor, on the reference's alpha
        out, _ = oracle.forward(x, ref_alpha if per_row else ref_alpha[:1], grid, gmax, ovp)
        assert f32_same_rows(out, sel[k + "__out"]), k
    assert n_same >= 0.9 * n_alpha, (n_same, n_alpha)


def f32_same_rows(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    b = np.ascontiguousarray(b, np.float32).reshape(-1)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


# ---------------------------------------------------------------- a14: quant_affine (configs[0])
def test_affine(oracle):
    a = golden("affine.npz")
    x = a["c0_x"].reshape(64, -1)
    for k in (4, 8):
        out, q = oracle.affine(x, k, x.min(), x.max())
        assert same_f32(out, a["c0_k%d_pt_out" % k])
        out, q = oracle.affine(x, k, x.min(1), x.max(1))
        assert same_f32(out, a["c0_k%d_pc_out" % k])
        assert q.min() >= -(1 << (k - 1)) and q.max() <= (1 << (k - 1)) - 1
    l = a["lin_x"]
    assert same_f32(oracle.affine(l, 4, l.min(1), l.max(1))[0], a["lin_k4_pc_out"])
    assert same_f32(oracle.affine(l, 8, l.min(), l.max())[0], a["lin_k8_pt_out"])


def test_bf16_helpers(oracle):
    f = np.float32([1.0, 1.00390625, 1.005859375, -2.5, 3.4e38, np.inf, 1e-40, 0.0, -0.0])
    b = oracle.f32_to_bf16(f)
    import torch
    ref = torch.from_numpy(f).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(b, ref)
    assert np.array_equal(oracle.bf16_to_f32(b).view(np.uint32), b.astype(np.uint32) << 16)


def test_reference_score_noise_sets_the_near_tie_rule():
    """the tolerance under which two clip / type candidates count as tied is not a guess -- it is twice
    the largest |fp32 score - fp64 score| / score over every mse_loss value the reference computed while the trace fixtures
    were recorded (159 612 scores, *_traces64.npz).  SURVEY 8c expected about 1e-6; measured 3.0e-7."""
    import calib_check
    assert calib_check._N_SCORES >= 150000
    assert 1e-8 < calib_check.REFERENCE_SCORE_NOISE < 1e-6, calib_check.REFERENCE_SCORE_NOISE
    assert calib_check.NEAR_TIE_RTOL == 2.0 * calib_check.REFERENCE_SCORE_NOISE < 1e-6
    # the fp64 twins really are twins: same shapes, same picks wherever the fp32 scores do not tie
    tr, t64 = golden("ant_select_traces.npz"), golden("ant_select_traces64.npz")
    n = same = 0
    for k in tr.files:
        if not k.endswith("__trace"):
            continue
        a, b = tr[k], t64[k + "64"]
        assert a.shape == b.shape and b.dtype == np.float64
        same += int((a.argmin(0) == b.argmin(0)).sum())
        n += a.shape[1]
    assert same >= 0.97 * n, (same, n)
