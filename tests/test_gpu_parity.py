"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bar: grid indices bit-exact; dequantised floats bit-identical to the fp32 op sequence of the
reference (bf16 outputs = that result rounded to nearest-even; NaN matches NaN).
"""
import os
import types

import numpy as np
import pytest

import calib_check
from calib_check import check_alpha_picks, check_type_pick, ratios_of


def _pick_report(what, n_same, n_rows, ledger_from):
    """print and assert how the clip picks of an end-to-end fixture suite compare with the reference's:
    identical-pick fraction >= 0.99, every other row certified a tie within calib_check.NEAR_TIE_RTOL (= twice the
    reference's measured reduction noise); the largest gap actually used goes to the log (and to gpurun_out/ for DESIGN)."""
    gaps = [e[3] for e in calib_check.LEDGER[ledger_from:]]
    line = "%s: %d / %d rows with the reference's own pick (%.4f), largest certified gap %.3g of %.3g allowed" % (
        what, n_same, n_rows, n_same / max(n_rows, 1), max(gaps + [0.0]), calib_check.NEAR_TIE_RTOL)
    print(line)
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "calib_picks.log"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert n_same >= 0.99 * n_rows, line

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _args(**kw):
    d = dict(w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False, no_outlier=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def f32_same(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def to_dev(x_np, dev, bf16=False):
    import torch
    if bf16:
        return torch.from_numpy(x_np.view(np.int16)).to(dev).view(torch.bfloat16)
    return torch.from_numpy(x_np).to(dev)


def bf16_bits(t):
    import torch
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def bf16_same(got_bits, ref_bits, oracle):
    g, r = oracle.bf16_to_f32(got_bits).reshape(-1), oracle.bf16_to_f32(ref_bits).reshape(-1)
    return bool(np.all((got_bits.reshape(-1) == ref_bits.reshape(-1)) | (np.isnan(g) & np.isnan(r))))


def run_case(antq_lib, oracle, dev, x, alpha, grid, gmax, per_row, ovp, bf16):
    """One tensor through antq_fakequant against the oracle.  Long rows take the lane kernel or the per-row table kernel
    depending on their length and dtype (knob 5: 0 = always the table kernel, 2 = always the lane kernel): both are checked."""
    _run_case(antq_lib, oracle, dev, x, alpha, grid, gmax, per_row, ovp, bf16)
    vec = (x.shape[1] if per_row else x.size) // (8 if bf16 else 4)
    if vec >= 128 and (x.shape[1] if per_row else x.size) % (8 if bf16 else 4) == 0:
        for knob in (0, 2):
            antq_lib.lib().antq_debug_set(5, knob)
            try:
                _run_case(antq_lib, oracle, dev, x, alpha, grid, gmax, per_row, ovp, bf16)
            finally:
                antq_lib.lib().antq_debug_set(5, 1)


def _run_case(antq_lib, oracle, dev, x, alpha, grid, gmax, per_row, ovp, bf16):
    rows, K = x.shape
    plan = antq_lib.plan_for(grid)
    import torch
    a_t = torch.from_numpy(np.atleast_1d(alpha).astype(np.float32)).to(dev)
    if bf16:
        xb = oracle.f32_to_bf16(x)
        ref, ridx = oracle.forward(xb, alpha, grid, gmax, ovp)
        out, idx = antq_lib.fakequant(to_dev(xb, dev, True), a_t, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
        out2 = antq_lib.fakequant(to_dev(xb, dev, True), a_t, plan, gmax, rows, K, per_row, ovp=ovp)
        assert bf16_same(bf16_bits(out), ref, oracle)
        bad = np.flatnonzero(bf16_bits(out).reshape(-1) != bf16_bits(out2).reshape(-1))
        assert bad.size == 0, ("index output on / off differ", bad[:16], bad.size, bf16_bits(out).reshape(-1)[bad[:8]],
                               bf16_bits(out2).reshape(-1)[bad[:8]], xb.reshape(-1)[bad[:8]])
    else:
        ref, ridx = oracle.forward(x, alpha, grid, gmax, ovp)
        out, idx = antq_lib.fakequant(to_dev(x, dev), a_t, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
        out2 = antq_lib.fakequant(to_dev(x, dev), a_t, plan, gmax, rows, K, per_row, ovp=ovp)
        assert f32_same(out.cpu().numpy(), ref)
        assert f32_same(out2.cpu().numpy(), ref)
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx)


SHAPES = [(16, 4096), (64, 576), (64, 147), (128, 64), (7, 1000), (1, 4099), (512, 16), (3, 7), (1, 1), (9, 2304)]


def make_x(rng, rows, K, unsigned=False, specials=True):
    x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
    f = x.reshape(-1)
    f[::53] *= 9
    if specials and f.size > 16:
        f[5], f[7], f[9], f[11], f[13] = np.nan, np.inf, -3e30, 0.0, -0.0
        f[15] = 1e-41
    return np.abs(x) if unsigned else x


def safe_absmax(x):
    am = np.abs(np.nan_to_num(x, nan=0, posinf=0, neginf=0))
    am[am > 1e10] = 0
    return am


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("gname", ["flint_b4_s", "int_b4_s", "pot_b4_u", "float_b4_s", "int_b8_s", "int_b8_u",
                                   "flint_b6_s", "pot_b6_u", "apot_b4_s", "int_b2_s"])
def test_ant_fakequant_vs_oracle(antq_lib, oracle, dev, gname, bf16):
    rng = np.random.default_rng(sum(map(ord, gname)))
    g = golden("ant_grids.npz")[gname]
    for rows, K in SHAPES:
        x = make_x(rng, rows, K, unsigned=gname.endswith("_u"))
        alpha = (safe_absmax(x).max(1) * 0.9 + 1e-6).astype(np.float32)
        run_case(antq_lib, oracle, dev, x, alpha, g, float(g.max()), True, False, bf16)
        run_case(antq_lib, oracle, dev, x, np.float32(alpha.max()), g, float(g.max()), False, False, bf16)
        # heavy clipping: many |x / s| beyond twice the outermost value, where (q - d) + d is no longer q
        run_case(antq_lib, oracle, dev, x, (alpha * 0.07).astype(np.float32), g, float(g.max()), True, False, bf16)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("t", ["int", "flint"])
@pytest.mark.parametrize("bit,signed", [(4, True), (4, False), (8, True)])
def test_olive_fakequant_ovp_vs_oracle(antq_lib, oracle, dev, t, bit, signed, bf16):
    rng = np.random.default_rng(7)
    O = golden("olive_grids.npz")
    s = "s" if signed else "u"
    gn, go = O["%s_b%d_%s" % (t, bit, s)], O["outlier_b%d_%s" % (bit, s)]
    g = np.concatenate([gn, go])
    for rows, K in SHAPES:
        x = make_x(rng, rows, K, unsigned=not signed)
        m = rng.random((rows, K)) < 0.03
        x[m] *= rng.uniform(8, 64, m.sum()).astype(np.float32)
        f = x.reshape(-1)
        if f.size > 4:
            f[0], f[2], f[3] = 1.5, 1.2, (-1.4 if signed else 1.4)
        alpha = (3 * np.nan_to_num(safe_absmax(x)).std(1) + 1e-6).astype(np.float32)
        run_case(antq_lib, oracle, dev, x, alpha, g, float(gn.max()), True, True, bf16)
        run_case(antq_lib, oracle, dev, x, np.float32(alpha.mean()), g, float(gn.max()), False, True, bf16)
        run_case(antq_lib, oracle, dev, x, alpha, gn, float(gn.max()), True, False, bf16)   # no_outlier


def test_scan_plan_and_odd_scales(antq_lib, oracle, dev):
    """Grids the table cannot hold (scan plan) and scales outside the fast-division domain."""
    rng = np.random.default_rng(3)
    x = make_x(rng, 32, 512)
    g_scan = np.float32([3, 1, 2, 1, 3, -7])
    assert not antq_lib.plan_for(g_scan).is_table
    run_case(antq_lib, oracle, dev, x * 50, np.full(32, 3.0, np.float32), g_scan, 3.0, True, False, False)
    g = golden("ant_grids.npz")["flint_b4_s"]
    for a in (0.0, -0.5, np.inf, np.nan, 1e-30, 1e30, 1e-44):
        alpha = np.full(32, a, np.float32)
        alpha[::2] = 0.07
        with np.errstate(all="ignore"):
            run_case(antq_lib, oracle, dev, x, alpha, g, 10.0, True, False, False)
            run_case(antq_lib, oracle, dev, x, alpha, g, 10.0, True, False, True)
    # huge / tiny inputs around the table's domain edges
    xe = np.float32([[1e3, -1e3, 1e5, 1.1e5, -1.1e5, 1e19, 1e-30, -1e-30, 1e-39, 65536.0, 2 ** 20, 2 ** 20 - 1] * 64])
    run_case(antq_lib, oracle, dev, xe, np.float32([10.0]), g, 10.0, False, False, False)


def test_golden_forward_fixtures_through_the_kernels(antq_lib, oracle, dev):
    """The reference's own outputs (captured by make_golden.py), not just the oracle's."""
    import torch
    f = golden("ant_forward.npz")
    gr = golden("ant_grids.npz")
    n = 0
    for k in f.files:
        if not k.endswith("_out") or k.startswith(("c0_", "g16_")):
            continue
        case = k[:-4]
        sname, t, s, pc = case.rsplit("_", 3)
        x = f[sname + "_x"]
        if s == "u":
            x = np.abs(x)
        rows = x.shape[0] if pc == "pc" else 1
        grid = gr["%s_b4_%s" % (t, s)]
        out, idx = antq_lib.fakequant(to_dev(np.ascontiguousarray(x), dev), to_dev(f[case + "_alpha"], dev),
                                      antq_lib.plan_for(grid), 10.0, rows, x.size // rows, pc == "pc", want_idx=True)
        assert f32_same(out.cpu().numpy(), f[k]), case
        assert np.array_equal(idx.cpu().numpy().reshape(-1), f[case + "_idx"]), case
        n += 1
    assert n == 48
    # configs[0], second leg: INT8 per-tensor on the ResNet-18 conv1-shaped tensor
    out = antq_lib.fakequant(to_dev(f["c0_x"], dev), to_dev(f["c0_int8_pt_alpha"], dev),
                             antq_lib.plan_for(gr["int_b8_s"]), 10.0, 1, f["c0_x"].size, False)
    assert f32_same(out.cpu().numpy(), f["c0_int8_pt_out"])
    # group-16: rows := numel/16, row_len := 16 on the same buffer
    out = antq_lib.fakequant(to_dev(f["g16_x"], dev), to_dev(f["g16_flint_alpha"], dev),
                             antq_lib.plan_for(gr["flint_b4_s"]), 10.0, f["g16_x"].size // 16, 16, True)
    assert f32_same(out.cpu().numpy(), f["g16_flint_out"])

    fo = golden("olive_forward.npz")
    og = golden("olive_grids.npz")
    n = 0
    for k in fo.files:
        if not k.endswith("_out"):
            continue
        case = k[:-4]
        name, t, pc, mode = case.rsplit("_", 3)
        s = "u" if name.startswith("a6x50") else "s"
        normal = og["%s_b4_%s" % (t, s)]
        grid = normal if mode == "noout" else np.concatenate([normal, og["outlier_b4_%s" % s]])
        x = fo[name + "_x"]
        rows = x.shape[0] if pc == "pc" else 1
        out = antq_lib.fakequant(to_dev(np.ascontiguousarray(x), dev), to_dev(fo[case + "_alpha"], dev),
                                 antq_lib.plan_for(grid), float(normal.max()), rows, x.size // rows, pc == "pc",
                                 ovp=(mode == "ovp"))
        assert f32_same(out.cpu().numpy(), fo[k]), case
        n += 1
    assert n == 34


def test_quant_cuda_dropin_operator(antq_lib, oracle, dev):
    """quant_cuda.quant(x, grid) -> (z, idx): same outputs as the reference op, idx all zeros."""
    import torch
    from ant_quantization_amd import quant_cuda
    gr = golden("ant_grids.npz")
    n = golden("ant_nearest.npz")
    for k in sorted({k[:-2] for k in n.files if k.endswith("_x") and not k.startswith("f64_")}):
        x = to_dev(n[k + "_x"], dev)
        z, idx = quant_cuda.quant(x, to_dev(gr[k], dev))
        assert f32_same(z.cpu().numpy(), n[k + "_z"]), k
        assert idx.shape == x.shape and idx.dtype == x.dtype and not idx.any()
        _, j = antq_lib.nearest(x, to_dev(gr[k], dev), want_idx=True)
        assert np.array_equal(j.cpu().numpy(), n[k + "_idx"]), k
    x64 = torch.from_numpy(n["f64_flint_b4_s_x"]).to(dev)
    z, _ = quant_cuda.quant(x64, torch.from_numpy(gr["flint_b4_s"].astype(np.float64)).to(dev))
    zr = n["f64_flint_b4_s_z"]
    zz = z.cpu().numpy()
    assert z.dtype == torch.float64 and np.array_equal(zz[~np.isnan(zr)], zr[~np.isnan(zr)])
    o = golden("olive_nearest.npz")
    for k in ("int_b4_s", "flint_b4_u"):
        z, _ = quant_cuda.quant(to_dev(o[k + "_x"], dev), to_dev(o[k + "_grid"], dev))
        assert f32_same(z.cpu().numpy(), o[k + "_z"]), k
    # a 509-entry grid (OliVe 8-bit + outliers) overflowed the reference's 256-entry LDS array; here it works
    og = golden("olive_grids.npz")
    big = np.concatenate([og["int_b8_s"], og["outlier_b8_s"]])
    xs = (np.random.default_rng(0).standard_normal(5000) * 40).astype(np.float32)
    z, _ = quant_cuda.quant(to_dev(xs, dev), to_dev(big, dev))
    assert f32_same(z.cpu().numpy(), oracle.nearest(xs, big)[0])
    # bf16 extension
    xb = oracle.f32_to_bf16(xs)
    zb = antq_lib.nearest(to_dev(xb, dev, True), to_dev(big, dev))
    assert np.array_equal(bf16_bits(zb), oracle.f32_to_bf16(oracle.nearest(oracle.bf16_to_f32(xb), big)[0]))


def test_affine_kernel_vs_golden(antq_lib, dev):
    import torch
    from ant_quantization_amd.ant.quant_affine import AsymmetricQuantFunction
    a = golden("affine.npz")
    c0 = to_dev(a["c0_x"], dev)
    for k in (4, 8):
        out = AsymmetricQuantFunction.apply(c0, k, c0.min(), c0.max())
        assert f32_same(out.cpu().numpy(), a["c0_k%d_pt_out" % k])
        mn, mx = c0.view(64, -1).min(1).values, c0.view(64, -1).max(1).values
        out = AsymmetricQuantFunction.apply(c0, k, mn, mx)
        assert f32_same(out.cpu().numpy(), a["c0_k%d_pc_out" % k])
    l = to_dev(a["lin_x"], dev)
    assert f32_same(AsymmetricQuantFunction.apply(l, 4, l.min(1).values, l.max(1).values).cpu().numpy(), a["lin_k4_pc_out"])
    assert f32_same(AsymmetricQuantFunction.apply(l, 8, l.min(), l.max()).cpu().numpy(), a["lin_k8_pt_out"])


@pytest.mark.parametrize("bf16", [False, True])
def test_dynamic_absmax_fakequant(antq_lib, oracle, dev, bf16):
    """alpha computed in the kernel (one quant group per wavefront / lane group / two-pass)."""
    rng = np.random.default_rng(11)
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    O = golden("olive_grids.npz")
    g_ol = np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])
    plan_ol = antq_lib.plan_for(g_ol)
    for rows, K in [(64, 4096), (300, 16), (128, 64), (40, 576), (16, 8192), (64, 147), (5, 2048), (1000, 32), (3, 28672),
                    (9, 16384), (7, 5120), (6, 11008), (2, 16392), (2, 65536), (1, 131080), (520, 8), (77, 256), (33, 512)]:
        x = make_x(rng, rows, K, specials=False)
        # OliVe codebook with outlier-victim pairs, alpha = abs-max * ratio (no_outlier-style x_max, OQ:198): same kernels,
        # OVP instantiation -- incl. the single-read paths for 28 672-wide rows (one row per 1024-thread workgroup)
        xo = x.copy()
        xo.reshape(-1)[::61] *= 30
        if bf16:
            xob = oracle.f32_to_bf16(xo)
            a_ol = oracle.absmax(oracle.bf16_to_f32(xob), True, 0.25)
            ref, ridx = oracle.forward(xob, a_ol, g_ol, 32.0, True)
            out, a_dev, idx = antq_lib.fakequant_dynamic(to_dev(xob, dev, True), plan_ol, 32.0, rows, K, ratio=0.25, ovp=True,
                                                         want_idx=True)
            assert bf16_same(bf16_bits(out), ref, oracle), (rows, K, "ovp")
        else:
            a_ol = oracle.absmax(xo, True, 0.25)
            ref, ridx = oracle.forward(xo, a_ol, g_ol, 32.0, True)
            out, a_dev, idx = antq_lib.fakequant_dynamic(to_dev(xo, dev), plan_ol, 32.0, rows, K, ratio=0.25, ovp=True,
                                                         want_idx=True)
            assert f32_same(out.cpu().numpy(), ref), (rows, K, "ovp")
        assert np.array_equal(a_dev.cpu().numpy(), a_ol), (rows, K, "ovp")
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx), (rows, K, "ovp")
        for ratio in (1.0, 0.83):
            if bf16:
                xb = oracle.f32_to_bf16(x)
                xf = oracle.bf16_to_f32(xb)
                alpha = oracle.absmax(xf, True, ratio)
                ref, ridx = oracle.forward(xb, alpha, g)
                out, a_dev, idx = antq_lib.fakequant_dynamic(to_dev(xb, dev, True), plan, 10.0, rows, K, ratio=ratio,
                                                             want_idx=True)
                assert bf16_same(bf16_bits(out), ref, oracle), (rows, K)
            else:
                alpha = oracle.absmax(x, True, ratio)
                ref, ridx = oracle.forward(x, alpha, g)
                out, a_dev, idx = antq_lib.fakequant_dynamic(to_dev(x, dev), plan, 10.0, rows, K, ratio=ratio,
                                                             want_idx=True)
                assert f32_same(out.cpu().numpy(), ref), (rows, K)
            assert np.array_equal(a_dev.cpu().numpy(), alpha), (rows, K)
            assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx), (rows, K)
    # abs-max alone, incl. NaN propagation and the per-tensor atomic path
    x = make_x(rng, 33, 1000)
    got = antq_lib.absmax(to_dev(x, dev), 33, 1000, per_row=True).cpu().numpy()
    ref = oracle.absmax(x, True)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    xc = make_x(rng, 257, 1023, specials=False)
    assert antq_lib.absmax(to_dev(xc, dev), 257, 1023, per_row=False).item() == np.abs(xc).max()


def test_search_sse_vs_oracle_traces(antq_lib, oracle, dev):
    """Fused clip search: per-candidate MSE within reduction tolerance, chosen alpha identical
    (except for near-ties), for ANT (75..150) and OliVe (75..250 step 2, OVP)."""
    import torch
    from ant_quantization_amd import core
    s = golden("ant_search.npz")
    gr = golden("ant_grids.npz")
    for name, t in [("w", "int"), ("w", "flint"), ("w", "pot"), ("a", "flint"), ("au", "int")]:
        x = s["a_x"] if name.startswith("a") else s["w_x"]
        if name == "au":
            x = np.abs(x)
        per_row = name == "w"
        grid = gr["%s_b4_%s" % (t, "u" if name == "au" else "s")]
        xt = to_dev(np.ascontiguousarray(x), dev)
        xmax = core.row_absmax(xt, per_row)
        best, alpha, ratios = core.clip_search(xt, xmax, per_row, 75, 150, 1, antq_lib.plan_for(grid), 10.0)
        key = "%s_%s" % (name, t)
        ref_trace = s[key + "_trace"]
        rows, K = (x.shape[0], x.shape[1]) if per_row else (1, x.size)
        sse = antq_lib.search_sse(xt, rows, K, xmax, per_row, ratios, antq_lib.plan_for(grid), 10.0)
        mse = (sse / K).float().cpu().numpy()
        np.testing.assert_allclose(mse, ref_trace.reshape(mse.shape), rtol=2e-5, atol=1e-12, err_msg=key)
        np.testing.assert_allclose(best.sum().item(), s[key + "_best_sum"], rtol=2e-5)
        check_alpha_picks(key, alpha.cpu().numpy(), s[key + "_alpha"], ref_trace.reshape(mse.shape), ratios_of(75, 150, 1),
                          xmax_rtol=0.0)


_ANT_TYPES = ("int", "flint", "pot", "float", "float1", "float2", "float3", "float4", "apot")


def _check_calibration(antq_lib, dev, q, x, out, k, sel, tr, lo, up, step, ovp, xmax_rtol):
    """One complete calibration against the reference's record of it.  Type and clip picks must be the reference's, or
    candidates the reference's OWN scores rank within its reduction noise of its best (tests/calib_check.py); the
    forward is compared for EVERY row on the reference's alpha, and q's own output wherever its alpha is the same."""
    import torch
    ref_mode = str(sel[k + "__mode"])
    types = [t for t in (_ANT_TYPES if not hasattr(q, "outliers") else ("int", "flint")) if ("-" + t) in k.split("__")[1]]
    check_type_pick(k, q.mode, ref_mode, types, tr[k + "__type_sums"])
    assert bool(q.is_signed) == bool(sel[k + "__signed"]), k
    if q.mode != ref_mode:
        return None                                   # a reference-certified tie between two types: other grid, nothing to compare
    g_got, g_ref = q.quant_grid.cpu().numpy(), sel[k + "__grid"]
    if q.mode == "apot" and g_ref.size >= 32:
        # torch.sort is unstable: the order of apot's +0 / -0 pair is unspecified for >= 32 entries
        assert np.array_equal(g_got, g_ref), k
    else:
        assert f32_same(g_got, g_ref), k
    full = g_ref
    if hasattr(q, "outliers"):
        assert f32_same(q.outliers.cpu().numpy(), sel[k + "__outliers"]), k
        if ovp:
            full = np.concatenate([g_ref, sel[k + "__outliers"]])
    ref_alpha = sel[k + "__alpha"].reshape(-1)
    got_alpha = q.alpha.detach().cpu().numpy().reshape(-1)
    same = check_alpha_picks(k, got_alpha, ref_alpha, tr[k + "__trace"], ratios_of(lo, up, step), xmax_rtol=xmax_rtol)
    per_row = ref_alpha.size > 1 or not q.is_input
    rows = x.shape[0] if per_row else 1
    # the forward for ALL rows (victim pairs included) on the reference's alpha: bit for bit
    fq = antq_lib.fakequant(x.contiguous(), to_dev(ref_alpha, dev), antq_lib.plan_for(full), float(np.max(g_ref)), rows,
                            x.numel() // rows, per_row, ovp=ovp)
    assert f32_same(fq.cpu().numpy(), sel[k + "__out"]), k
    # q's own output: identical bits wherever its alpha is the reference's
    exact = got_alpha == ref_alpha
    got = out.detach().cpu().numpy().reshape(x.shape[0], -1)
    ref_out = sel[k + "__out"].reshape(x.shape[0], -1)
    if per_row and not ovp:
        assert f32_same(got[exact], ref_out[exact]), k
    elif exact.all():
        assert f32_same(got, ref_out), k
    assert q._steady and torch.equal(q(x), out)
    return same


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_quantizer_end_to_end_vs_reference_fixtures(antq_lib, dev, tree, capsys):
    """TensorQuantizer: first call calibrates (type select + clip search), later calls are steady
    state.  Chosen type, alpha and output against what the reference's Python produced."""
    import importlib
    import torch
    qm = importlib.import_module("ant_quantization_amd.%s.quant_modules" % tree)
    if tree == "ant":
        sel, tr = golden("ant_select.npz"), golden("ant_select_traces.npz")
        cases = [(n, m) for n in ("w_gauss", "w_unif", "w_laplace", "x_relu", "x_gelu")
                 for m in ("ant-int-pot-flint", "ant-int-flint", "flint", "int")]
        kw = {}
        fmt = "%s__%s"
    else:
        sel, tr = golden("olive_search.npz"), golden("olive_search_traces.npz")
        cases = [(n, m) for n in ("w", "a") for m in ("ant-int-flint", "flint", "int")]
        kw = dict(w_up=250, a_up=250)
        fmt = "full_%s__%s"
    n_rows = n_same = 0
    led0 = len(calib_check.LEDGER)
    for name, mode in cases:
        k = fmt % (name, mode)
        x_np = sel[name + ("__x" if tree == "ant" else "_x")]
        is_input = name.startswith(("x_", "a"))
        q = qm.TensorQuantizer(mode=mode, bit=4, is_signed=not is_input, is_enable=True, is_input=is_input,
                               args=_args(**kw)).to(dev)
        q.name = "golden"
        x = to_dev(np.ascontiguousarray(x_np), dev)
        if not is_input:
            q.alpha.data = torch.ones(x.shape[0], 1, device=dev)
        out = q(x)
        same = _check_calibration(antq_lib, dev, q, x, out, k, sel, tr, 75, 150 if tree == "ant" else 250,
                                  1 if tree == "ant" else 2, tree == "olive", 0.0 if tree == "ant" else 2e-6)
        assert same is not None, k
        n_rows += same.size
        n_same += int(same.sum())
        np.testing.assert_allclose(q.mse.item(), sel[k + "__mse"], rtol=2e-3)
        assert float(q.has_inited_quant_para) == 1.0
    printed = capsys.readouterr().out
    _pick_report("end to end, %s fixtures" % tree, n_same, n_rows, led0)
    assert "4-bit \t golden," in printed       # the log line format print_result.sh parses


def test_quantized_layers_forward_and_qat_backward(antq_lib, oracle, dev):
    """Conv2dQuantizer / LinearQuantizer run F.conv2d / F.linear on fake-quantised operands;
    the ANT quantiser is trainable (STE): d out/d x = 1, d out/d alpha = sum g*(q-d)/gmax."""
    import torch
    import torch.nn as nn
    from ant_quantization_amd.ant import quant_model, quant_modules as qm, quant_utils
    torch.manual_seed(0)
    quant_utils.set_quantizer(types.SimpleNamespace(mode="flint", wbit=4, abit=4, **vars(_args())))
    net = nn.Sequential(nn.Conv2d(3, 8, 3), nn.ReLU(), nn.Flatten(), nn.Linear(8 * 6 * 6, 10))
    qnet = quant_model.quantize_model(net).to(dev)
    quant_utils.enable_quantization(qnet)
    x = torch.randn(4, 3, 8, 8, device=dev)
    y = qnet(x)
    assert y.shape == (4, 10) and torch.isfinite(y).all()
    conv = qnet[0]
    wq = conv.quant_weight(conv.weight)
    ref, _ = oracle.forward(conv.weight.detach().cpu().numpy().reshape(8, -1),
                            conv.quant_weight.alpha.detach().cpu().numpy().reshape(-1),
                            conv.quant_weight.quant_grid.cpu().numpy())
    assert f32_same(wq.detach().cpu().numpy(), ref)
    # QAT gradients vs the autograd graph of the reference's op sequence (torch ops, fp32)
    lin = qnet[3]
    w = lin.weight.detach().clone().requires_grad_(True)
    alpha = lin.quant_weight.alpha.detach().clone().requires_grad_(True)
    grid = lin.quant_weight.quant_grid
    scale = alpha / grid.max()
    d = w / scale
    q = qm.QuantBase.forward(d.detach(), grid)
    out_ref = ((q - d).detach() + d) * scale
    g = torch.randn_like(out_ref)
    out_ref.backward(g)
    lin.quant_weight.alpha.grad = None
    w2 = lin.weight
    w2.grad = None
    out = lin.quant_weight(w2)
    assert out.requires_grad and torch.equal(out.detach(), out_ref.detach())
    out.backward(g)
    torch.testing.assert_close(w2.grad, w.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(lin.quant_weight.alpha.grad, alpha.grad, rtol=2e-3, atol=1e-5)


def test_full_size_properties_headline_tensor(antq_lib, dev):
    """4096 x 4096 bf16 / fp32 at BASELINE size: properties that need no CPU reference."""
    import torch
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    torch.manual_seed(6)
    x = torch.randn(4096, 4096, device=dev) * 0.02
    alpha = x.abs().amax(1).contiguous()
    out = antq_lib.fakequant(x, alpha, plan, 10.0, 4096, 4096, True)
    # (1) idempotent: quantising the quantised tensor changes nothing
    assert torch.equal(antq_lib.fakequant(out, alpha, plan, 10.0, 4096, 4096, True), out)
    # (2) every output is fl(g * s) for a grid value g: at most 15 distinct magnitudes per row
    s = alpha / torch.tensor(10.0, device=dev)      # tensor / tensor = true division, as in the reference
    gt = torch.from_numpy(g).to(dev)
    allowed = (gt[None, :] * s[:64, None])
    assert all(torch.isin(out[r], allowed[r]).all() for r in range(64))
    # (3) monotone: the quantiser is a non-decreasing step function of x
    xs, order = torch.sort(x[:256], dim=1)
    os_ = torch.gather(out[:256], 1, order)
    assert (os_[:, 1:] >= os_[:, :-1]).all()
    # (4) power-of-two scaling commutes exactly with the whole pipeline
    out8 = antq_lib.fakequant(x * 8, alpha * 8, plan, 10.0, 4096, 4096, True)
    assert torch.equal(out8, out * 8)
    # (5) dynamic == static with alpha = row abs-max; bf16 path = fp32 path on x.float() rounded once
    outd, ad, _ = antq_lib.fakequant_dynamic(x, plan, 10.0, 4096, 4096)
    assert torch.equal(ad, alpha) and torch.equal(outd, out)
    xb = x.bfloat16()
    ab = xb.float().abs().amax(1).contiguous()
    outb = antq_lib.fakequant(xb, ab, plan, 10.0, 4096, 4096, True)
    assert torch.equal(outb, antq_lib.fakequant(xb.float(), ab, plan, 10.0, 4096, 4096, True).bfloat16())
    # (6) group-16 view of the same buffer == per-row on the reshaped tensor
    a16 = x.view(-1, 16).abs().amax(1).contiguous()
    o16 = antq_lib.fakequant(x, a16, plan, 10.0, x.numel() // 16, 16, True)
    o16d, a16d, _ = antq_lib.fakequant_dynamic(x, plan, 10.0, x.numel() // 16, 16)
    assert torch.equal(a16d, a16) and torch.equal(o16d, o16)
    # (7) in place is allowed
    xc = x.clone()
    antq_lib.fakequant(xc, alpha, plan, 10.0, 4096, 4096, True, out=xc)
    assert torch.equal(xc, out)


def test_olive_full_size_pair_invariant(antq_lib, dev):
    """OPT-sized [4096, 4096] tensor with planted outliers: no pair keeps two non-zero outliers-with-victim
    violations, and victims are exactly the partners of outliers."""
    import torch
    O = golden("olive_grids.npz")
    gn, go = O["flint_b4_s"], O["outlier_b4_s"]
    plan = antq_lib.plan_for(np.concatenate([gn, go]))
    torch.manual_seed(4)
    x = torch.randn(4096, 4096, device=dev) * 0.02
    m = torch.rand_like(x) < 0.001
    x[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 64)
    alpha = (3 * x.std(1)).contiguous()
    out, idx = antq_lib.fakequant(x, alpha, plan, 32.0, 4096, 4096, True, ovp=True, want_idx=True)
    pairs = idx.view(-1, 2)
    n_norm = gn.size
    is_out = pairs >= n_norm
    is_vic = pairs == antq_lib.IDX_VICTIM
    assert is_out.any() and is_vic.any()
    assert not (is_out[:, 0] & is_out[:, 1]).any()            # never two outliers in a pair
    assert (is_vic.sum(1) <= 1).all()
    assert (is_vic.any(1) == is_out.any(1)).all()             # a victim iff its partner is an outlier
    assert (out.view(-1, 2)[is_vic] == 0).all()


def test_batched_launch_equals_individual_launches(antq_lib, dev):
    """ResNet-50's 54 weight tensors in ONE launch: per-channel, group-16 and OliVe."""
    import torch
    sys_path = __import__("sys").path
    import os
    sys_path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_configs import resnet50_shapes
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    plan_i = antq_lib.plan_for(golden("ant_grids.npz")["int_b4_s"])
    torch.manual_seed(1)
    shapes = resnet50_shapes()
    assert len(shapes) == 54 and sum(int(np.prod(s)) for s in shapes) == 25502912
    for dtype in (torch.float32, torch.bfloat16):
        ws = [(torch.randn(*s, device=dev) * 0.05).to(dtype) for s in shapes]
        for mode in ("per_channel", "group16"):
            jobs, refs = [], []
            for k, w in enumerate(ws):
                rows, K = (w.shape[0], w.numel() // w.shape[0]) if mode == "per_channel" else (w.numel() // 16, 16)
                a = antq_lib.absmax(w, rows, K)
                p = plan if k % 2 == 0 else plan_i                   # mixed grids inside one launch
                refs.append(antq_lib.fakequant(w, a, p, 10.0, rows, K, True))
                jobs.append((w, torch.zeros_like(w), a, p, 10.0, rows, K, True))
            b = antq_lib.Batch(jobs)
            assert not b.singles                 # conv1 (K = 147, ragged) rides in the same launch
            b.run()
            for j, r in zip(jobs, refs):
                assert torch.equal(j[1], r), (mode, dtype, tuple(j[0].shape))
    O = golden("olive_grids.npz")
    pol = antq_lib.plan_for(np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]))
    ws = [torch.randn(256, 512, device=dev) * 0.02 for _ in range(3)]
    for w in ws:
        w.view(-1)[::301] *= 30
    al = [(3 * w.std(1)).contiguous() for w in ws]
    refs = [antq_lib.fakequant(w, a, pol, 32.0, 256, 512, True, ovp=True) for w, a in zip(ws, al)]
    jobs = [(w, torch.zeros_like(w), a, pol, 32.0, 256, 512, True) for w, a in zip(ws, al)]
    antq_lib.Batch(jobs, ovp=True).run()
    assert all(torch.equal(j[1], r) for j, r in zip(jobs, refs))
    # ragged and odd-numel jobs with outlier-victim pairs (the wrap-around partner) next to vector jobs, one launch
    # (per-group tables of the batched kernel with victim pairs: rows of 16 / 32 / 64 vectors, fp32 and bf16)
    for dt in (torch.float32, torch.bfloat16):
        epl = 4 if dt == torch.float32 else 8
        ws = [(torch.randn(r, v * epl, device=dev) * 0.02).to(dt) for r, v in ((96, 16), (40, 32), (33, 64), (8, 128))]
        for w in ws:
            w.view(-1)[::23] *= 30
        al = [(3 * w.float().std(1)).contiguous() for w in ws]
        refs = [antq_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], True, ovp=True) for w, a in zip(ws, al)]
        jobs = [(w, torch.zeros_like(w), a, pol, 32.0, w.shape[0], w.shape[1], True) for w, a in zip(ws, al)]
        antq_lib.Batch(jobs, ovp=True).run()
        assert all(torch.equal(j[1], r) for j, r in zip(jobs, refs)), dt
    shapes2 = [(7, 33), (64, 147), (256, 512), (5, 27), (1, 4099)]
    ws = [torch.randn(*sh, device=dev) * 0.02 for sh in shapes2]
    for w in ws:
        w.view(-1)[::17] *= 30
    for per_row in (True, False):
        al = [(3 * w.std(1)).contiguous() if per_row else (3 * w.std()).reshape(1) for w in ws]
        refs = [antq_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], per_row, ovp=True) for w, a in zip(ws, al)]
        jobs = [(w, torch.zeros_like(w), a, pol, 32.0, w.shape[0], w.shape[1], per_row) for w, a in zip(ws, al)]
        bt = antq_lib.Batch(jobs, ovp=True)
        assert not bt.singles
        bt.run()
        assert all(torch.equal(j[1], r) for j, r in zip(jobs, refs)), per_row


def _oracle_codes(oracle, ridx, n_normal, ovp, zero_code=None):
    """The packed 4-bit code of every element FROM THE ORACLE'S scan-order indices (OQ:155-179, :311-320): a normal value
    keeps its index, an outlier is its index in the outlier codebook, a victim carries the identifier 15."""
    want = ridx.astype(np.int64).copy()
    if ovp:
        want[ridx >= n_normal] -= n_normal
    want[ridx == oracle.IDX_VICTIM] = 15
    if zero_code is not None:
        want[ridx == oracle.IDX_NONE] = zero_code
    return want


def _nibbles(codes, rows, K):
    import torch
    return torch.stack([(codes & 15), (codes >> 4)], 1).reshape(rows, K).cpu().numpy().astype(np.int64)


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_packed_4bit_codec_roundtrip_equals_fakequant(antq_lib, oracle, dev, dtype_name):
    """The codes ARE the oracle's grid indices (outlier -> index in the outlier codebook, victim -> 15) and
    decode4(encode4(x)) is the oracle's forward output, bit for bit -- ANT and OliVe, fp32 and bf16, incl. an odd
    element count (torch.roll wrap, OQ:313-318).  (The fused kernel is compared with the same oracle elsewhere; here
    nothing on the reference side of an assert comes from the HIP library.)"""
    import torch
    dtype = getattr(torch, dtype_name)
    bf16 = dtype_name == "bfloat16"
    rng = np.random.default_rng(5)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")

    def host(x_np):       # what the kernels see, as the oracle takes it: fp32, or bf16 bits
        return oracle.f32_to_bf16(x_np) if bf16 else x_np

    def same(t, ref):
        return bf16_same(bf16_bits(t), ref, oracle) if bf16 else f32_same(t.cpu().numpy(), ref)

    for rows, K in [(64, 4096), (33, 24), (128, 64), (5, 1000)]:
        x_np = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
        x_np[rng.random((rows, K)) < 0.02] *= 30
        # ANT: flint / int / pot 4-bit, per-row alpha
        for gname in ("flint_b4_s", "int_b4_s", "pot_b4_u"):
            g = G[gname]
            xx_np = np.abs(x_np) if gname.endswith("_u") else x_np
            xh = host(xx_np)
            xf = oracle.bf16_to_f32(xh) if bf16 else xh
            alpha_np = (np.abs(xf).max(1) * np.float32(0.9)).astype(np.float32)
            ref, ridx = oracle.forward(xh, alpha_np, g, 10.0, False)
            xx, alpha = to_dev(xh, dev, bf16), torch.from_numpy(alpha_np).to(dev)
            plan = antq_lib.plan_for(g)
            codes = antq_lib.encode4(xx, alpha, plan, 10.0, rows, K, True)
            assert codes.numel() == rows * K // 2 and codes.dtype == torch.uint8
            assert np.array_equal(_nibbles(codes, rows, K), _oracle_codes(oracle, ridx, 0, False)), (gname, rows, K)
            dec = antq_lib.decode4(codes, alpha, plan, 10.0, rows, K, True, dtype)
            assert same(dec, ref), (gname, rows, K)
        # OliVe: normal + outlier codebooks, outlier-victim pairs, identifier code 15
        for t in ("int", "flint"):
            gn, go = O["%s_b4_s" % t], O["outlier_b4_s"]
            gg, gmax = np.concatenate([gn, go]), float(gn.max())
            xh = host(x_np)
            xf = oracle.bf16_to_f32(xh) if bf16 else xh
            alpha_np = (3 * xf.std(1)).astype(np.float32)
            ref, ridx = oracle.forward(xh, alpha_np, gg, gmax, True)
            assert (ridx == oracle.IDX_VICTIM).any() and (ridx >= gn.size).any()
            x, alpha = to_dev(xh, dev, bf16), torch.from_numpy(alpha_np).to(dev)
            plan = antq_lib.plan_for(gg)
            codes = antq_lib.encode4(x, alpha, plan, gmax, rows, K, True, n_normal=gn.size, ovp=True)
            assert np.array_equal(_nibbles(codes, rows, K), _oracle_codes(oracle, ridx, gn.size, True)), (t, rows, K)
            dec = antq_lib.decode4(codes, alpha, plan, gmax, rows, K, True, dtype, n_normal=gn.size, ovp=True)
            assert same(dec, ref), (t, rows, K)
    # An odd element count (the torch.roll wrap of OQ:313-318) has no packed form -- two codes per byte, eight per
    # 32-bit word: the codec refuses it loudly (row_len % 8) instead of guessing; the fused kernel's wrap rule is pinned
    # against the oracle in test_olive_fakequant_ovp_vs_oracle / the odd-numel golden cases.
    gn, go = O["flint_b4_s"], O["outlier_b4_s"]
    plan = antq_lib.plan_for(np.concatenate([gn, go]))
    xo = to_dev(host((rng.standard_normal((1, 4097)) * 0.02).astype(np.float32)), dev, bf16)
    with pytest.raises(antq_lib.AntqError):
        antq_lib.encode4(xo, torch.tensor([0.06], device=dev), plan, float(gn.max()), 1, 4097, False, n_normal=gn.size, ovp=True)
    # heavy clipping (most |x / s| beyond twice the outermost value), signed zeros, a denormal, magnitudes past the
    # scan's 102400 horizon (-> the zero code), rows whose scale is outside the table path's range (2^-60: literal path)
    for rows, K in [(16, 4096), (9, 40)]:
        x_np = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
        f = x_np.reshape(-1)
        f[3], f[4], f[6], f[11], f[12] = 0.0, -0.0, 1e-41, 3e30, -3e30
        x = torch.from_numpy(x_np).to(dev).to(dtype)
        for gname, ovp in (("flint_b4_s", False), ("int_b4_s", False), ("olive", True)):
            if ovp:
                gn, go = O["flint_b4_s"], O["outlier_b4_s"]
                g, gmax, nn = np.concatenate([gn, go]), float(gn.max()), gn.size
            else:
                g, gmax, nn = G[gname], 10.0, 0
            plan = antq_lib.plan_for(g)
            alpha = (torch.nan_to_num(x.float()).abs().clamp(max=1.0).amax(1) * 0.07).contiguous()
            alpha[1] = 2.0 ** -60
            ref, ridx = antq_lib.fakequant(x, alpha, plan, gmax, rows, K, True, ovp=ovp, want_idx=True)
            assert torch.isfinite(ref.float()).all()
            codes = antq_lib.encode4(x, alpha, plan, gmax, rows, K, True, n_normal=nn, ovp=ovp)
            nib = torch.stack([(codes & 15), (codes >> 4)], 1).reshape(rows, K).to(torch.int16)
            zero_code = int(np.flatnonzero((g[:nn] if ovp else g) == 0)[-1])
            want = ridx.clone()
            want[ridx == antq_lib.IDX_VICTIM] = 15
            want[ridx == antq_lib.IDX_NONE] = zero_code
            if ovp:
                want[ridx >= nn] -= nn
            assert torch.equal(nib, want), (gname, rows, K)
            # the decoder multiplies the code's value by the scale; the reference's (q - d) + d is that value only while
            # |d| <= 2 |q|: compare where the element is not clipped beyond twice the outermost value
            dec = antq_lib.decode4(codes, alpha, plan, gmax, rows, K, True, dtype, n_normal=nn, ovp=ovp)
            near = x.float().abs() <= 1.9 * alpha[:, None] * float(np.abs(g).max()) / gmax
            near[1] = False
            assert torch.equal(dec[near], ref[near]), (gname, rows, K)
            antq_lib.lib().antq_debug_set(4, 0)           # the same through the exact-division encoder
            try:
                assert torch.equal(antq_lib.encode4(x, alpha, plan, gmax, rows, K, True, n_normal=nn, ovp=ovp), codes)
            finally:
                antq_lib.lib().antq_debug_set(4, 1)
    # group-16 view and per-tensor scale
    g = G["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    x = (torch.randn(256, 512, device=dev) * 0.05).to(dtype)
    a16 = antq_lib.absmax(x, x.numel() // 16, 16)
    ref = antq_lib.fakequant(x, a16, plan, 10.0, x.numel() // 16, 16, True)
    dec = antq_lib.decode4(antq_lib.encode4(x, a16, plan, 10.0, x.numel() // 16, 16, True), a16, plan, 10.0,
                           x.numel() // 16, 16, True, dtype)
    assert torch.equal(dec.reshape(256, 512), ref)
    at = x.float().abs().max().reshape(1)
    ref = antq_lib.fakequant(x, at, plan, 10.0, 1, x.numel(), False)
    dec = antq_lib.decode4(antq_lib.encode4(x, at, plan, 10.0, 1, x.numel(), False), at, plan, 10.0, 1, x.numel(), False, dtype)
    assert torch.equal(dec.reshape(256, 512), ref)


def test_ant_outlier_mode_vs_reference(antq_lib, dev, capsys):
    """mode='outlier' (int4 body + int16 outliers by percentile, AQ:417-465) against the reference's outputs: the two
    percentile ends to 1e-6 (np.percentile on the host), the dequantised tensors BIT FOR BIT (round 3: was rtol 2e-6)."""
    import torch
    from ant_quantization_amd.ant import quant_modules as qm
    o = golden("ant_outlier.npz")
    x = to_dev(o["x"], dev)
    for pct in (99, 95):
        for signed in (True, False):
            xx = x if signed else x.abs()
            q = qm.TensorQuantizer(mode="outlier", bit=4, is_signed=signed, is_enable=True, is_input=not signed,
                                   args=_args(percent=float(pct))).to(dev)
            q.name = "golden"
            k = "p%d_%s" % (pct, "s" if signed else "u")
            out = q(xx)
            np.testing.assert_allclose(q.percent_value_int4.item(), o[k + "_p4"], rtol=1e-6)
            np.testing.assert_allclose(q.percent_value_int16.item(), o[k + "_p16"], rtol=1e-6)
            assert f32_same(out.detach().cpu().numpy(), o[k + "_out"]), k            # bit for bit (north_star: <= 1 ULP)
            assert f32_same(q(xx * 0.5).detach().cpu().numpy(), o[k + "_out2"]), k
    capsys.readouterr()


def test_multihead_attention_quantizer(antq_lib, dev):
    """Wrapper math == nn.MultiheadAttention when the quantisers are off; with them on it runs end to end
    and the four quantisers calibrate (weights per-channel, inputs per-tensor)."""
    import torch
    import torch.nn as nn
    from ant_quantization_amd.ant import quant_model, quant_utils
    from ant_quantization_amd.ant.multihead_attention import MultiheadAttentionQuantizer
    torch.manual_seed(0)
    quant_utils.set_quantizer(types.SimpleNamespace(mode="ant-int-flint", wbit=4, abit=4, **vars(_args())))
    for batch_first in (False, True):
        ma = nn.MultiheadAttention(64, 4, batch_first=batch_first).to(dev).eval()
        enc = nn.Sequential(ma)
        qenc = quant_model.quantize_model(enc).to(dev).eval()
        qma = qenc[0]
        assert type(qma) is MultiheadAttentionQuantizer
        assert {"in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias", "in_quant_weight.alpha",
                "out_quant_input.quant_grid"} <= set(qma.state_dict().keys())
        x = torch.randn(10, 3, 64, device=dev) if not batch_first else torch.randn(3, 10, 64, device=dev)
        quant_utils.disable_quantization(qenc)
        ref, ref_w = ma(x, x, x)
        out, w = qma(x, x, x)
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(w, ref_w, rtol=1e-4, atol=1e-5)
        quant_utils.enable_quantization(qenc)
        with torch.no_grad():
            outq, _ = qma(x, x, x)
        assert outq.shape == ref.shape and torch.isfinite(outq).all()
        assert 0 < (outq - ref).abs().mean() < ref.abs().mean()         # quantised, but still the same function
        assert qma.in_quant_weight.alpha.shape == (192, 1) and qma.out_quant_input.alpha.dim() == 0
        assert all(float(t.has_inited_quant_para) == 1.0 for t in (qma.in_quant_weight, qma.in_quant_input,
                                                                     qma.out_quant_weight, qma.out_quant_input))


def test_fp16_io_is_fp32_path_rounded_once(antq_lib, dev):
    """fp16 storage (extension, like bf16): result == fp32 kernel on x.float(), rounded to half once."""
    import torch
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    torch.manual_seed(2)
    for rows, K in [(64, 4096), (16, 576), (128, 64), (512, 16), (9, 147)]:
        x = (torch.randn(rows, K, device=dev) * 0.05).half()
        x.view(-1)[::37] *= 20
        alpha = (x.float().abs().amax(1) * 0.8).contiguous()
        plan = antq_lib.plan_for(G["flint_b4_s"])
        ref = antq_lib.fakequant(x.float(), alpha, plan, 10.0, rows, K, True).half()
        assert torch.equal(antq_lib.fakequant(x, alpha, plan, 10.0, rows, K, True), ref)
        od, ad, _ = antq_lib.fakequant_dynamic(x, plan, 10.0, rows, K, ratio=0.8)
        assert torch.equal(ad, x.float().abs().amax(1) * 0.8) and torch.equal(od, ref)
        gn = O["flint_b4_s"]
        pol = antq_lib.plan_for(np.concatenate([gn, O["outlier_b4_s"]]))
        a3 = (3 * x.float().std(1)).contiguous()
        ref = antq_lib.fakequant(x.float(), a3, pol, 32.0, rows, K, True, ovp=True).half()
        assert torch.equal(antq_lib.fakequant(x, a3, pol, 32.0, rows, K, True, ovp=True), ref)


def test_hf_models_end_to_end_and_checkpoint_roundtrip(antq_lib, dev, capsys):
    """Drop-in at module level: tiny HF BERT (ANT) and GPT-2 (OliVe), calibrate on the first batch, steady state
    afterwards, and a checkpoint saved from the calibrated model restores alpha / grids / bit so that a freshly
    rewritten model reproduces the outputs without calibrating."""
    import torch
    transformers = pytest.importorskip("transformers")
    from transformers import BertConfig, BertModel, GPT2Config, GPT2LMHeadModel
    from ant_quantization_amd.ant import quant_model as aqm, quant_utils as aqu
    from ant_quantization_amd.olive import quant_model as oqm, quant_utils as oqu
    args = _args(mode="ant-int-flint", wbit=4, abit=4)
    torch.manual_seed(0)
    ids = torch.randint(0, 100, (4, 16), device=dev)
    cases = [
        (aqm, aqu, lambda: BertModel(BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                                                 intermediate_size=128, vocab_size=100)), lambda o: o.last_hidden_state),
        (oqm, oqu, lambda: GPT2LMHeadModel(GPT2Config(n_embd=64, n_layer=2, n_head=4, vocab_size=100, n_positions=32,
                                                      bos_token_id=0, eos_token_id=0)), lambda o: o.logits),
    ]
    for qmod, qutil, make, pick in cases:
        qutil.set_quantizer(args)
        torch.manual_seed(1)
        base = make().eval()
        model = qmod.quantize_model(base).to(dev).eval()
        qutil.enable_quantization(model)
        with torch.no_grad():
            y1 = pick(model(ids))          # calibrates every quantiser on this batch
            y2 = pick(model(ids))          # steady state
        assert torch.isfinite(y1).all() and torch.equal(y1, y2)
        with torch.no_grad():
            y_fp = pick(base.to(dev)(ids))
        assert 0 < (y1 - y_fp).abs().mean() < y_fp.abs().mean()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        assert any(k.endswith("quant_weight.quant_grid") for k in sd) and all(
            float(v) == 1.0 for k, v in sd.items() if k.endswith("has_inited_quant_para"))
        torch.manual_seed(2)                # different random weights: everything must come from the checkpoint
        fresh = qmod.quantize_model(make().eval()).to(dev).eval()
        qutil.enable_quantization(fresh)
        qmod.load_ant_state_dict(fresh, sd)
        fresh.load_state_dict(sd, strict=True)
        capsys.readouterr()
        with torch.no_grad():
            y3 = pick(fresh(ids))
        assert "4-bit" not in capsys.readouterr().out      # no calibration line: the checkpoint's state was used
        assert torch.equal(y3, y1)


def test_launch_shapes_and_unordered_launches_are_bit_identical(antq_lib, oracle, dev):
    """Round 3 launch shapes: wavefronts per workgroup (knob 6) and vectors per lane (knob 7) of the lane kernel, the
    batched row-table kernel with 4 / 2 / 1 wavefronts per workgroup, ordinary and unordered launches
    (ANTQ_FLAG_UNORDERED: no barrier bit on the dispatch packet) -- every combination against the oracle."""
    import torch
    rng = np.random.default_rng(77)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    gol = np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])
    knob = antq_lib.lib().antq_debug_set
    try:
        for bf16 in (True, False):
            for rows, K in ((48, 4096), (40, 1000), (3, 8200)):
                x = make_x(rng, rows, K, specials=False)
                x.reshape(-1)[::41] *= 25
                xh = oracle.f32_to_bf16(x) if bf16 else x
                xf = oracle.bf16_to_f32(xh) if bf16 else xh
                xt = to_dev(xh, dev, bf16)
                for ovp, g, gmax in ((False, G["flint_b4_s"], 10.0), (True, gol, 32.0)):
                    alpha = (np.abs(xf).max(1) * np.float32(0.3 if ovp else 0.9)).astype(np.float32)
                    ref, _ = oracle.forward(xh, alpha, g, gmax, ovp)
                    plan, a_t = antq_lib.plan_for(g), torch.from_numpy(alpha).to(dev)
                    bufs = [torch.empty_like(xt) for _ in range(3)]     # (unordered launches write caller-owned buffers)
                    torch.cuda.synchronize()
                    for w in (0, 1, 4):
                        for u in (0, 1, 2, 4):
                            knob(6, w)
                            knob(7, u)
                            for unordered in (False, True):
                                # several back-to-back launches into distinct outputs: unordered ones may overlap
                                outs = [antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, True, ovp=ovp, unordered=unordered,
                                                           out=bufs[i]) for i in range(3)]
                                for o in outs:
                                    ok = bf16_same(bf16_bits(o), ref, oracle) if bf16 else f32_same(o.cpu().numpy(), ref)
                                    assert ok, (bf16, rows, K, ovp, w, u, unordered)
                    knob(7, 0)
                    if K * (2 if bf16 else 4) // 16 >= 128:
                        for w in (4, 2, 1):
                            knob(6, w)
                            out = torch.zeros_like(xt)
                            out2 = torch.zeros_like(xt)
                            bt = antq_lib.Batch([(xt, out, a_t, plan, gmax, rows, K, True), (xt, out2, a_t, plan, gmax, rows, K, True)],
                                                ovp=ovp)
                            bt.run()
                            for o in (out, out2):
                                ok = bf16_same(bf16_bits(o), ref, oracle) if bf16 else f32_same(o.cpu().numpy(), ref)
                                assert ok, ("batch", bf16, rows, K, ovp, w)
                    knob(6, 0)
    finally:
        knob(6, 0)
        knob(7, 0)


def test_olive_three_sigma_statistic_on_one_read(antq_lib, oracle, dev):
    """antq_moments + antq_xmax_3sigma (OQ:193-197, :213-218): the sums against float64 numpy, x_max against the oracle's
    restatement and against torch's own mean / std on the GPU (the reference's ops), per row and per tensor, fp32 and
    bf16, ragged / unaligned rows, a constant row (std 0) and a one-element row (NaN, like torch.std); bit-reproducible."""
    import torch
    rng = np.random.default_rng(31)
    for rows, K in ((64, 4096), (7, 33), (3, 8200), (1, 1), (5, 1), (128, 64)):
        x = (rng.standard_normal((rows, K)) * 0.05 + 0.01).astype(np.float32)
        x.reshape(-1)[::37] *= 12
        if rows > 2 and K > 1:
            x[2] = 0.75                                     # constant row: std exactly 0
        for bf16 in (False, True):
            xh = oracle.f32_to_bf16(x) if bf16 else x
            xf = (oracle.bf16_to_f32(xh) if bf16 else xh).astype(np.float64)
            xt = to_dev(xh, dev, bf16)
            for per_row in (True, False):
                sums = antq_lib.moments(xt, rows, K, per_row)
                ref = xf if per_row else xf.reshape(1, -1)
                np.testing.assert_allclose(sums[:, 0].cpu().numpy(), ref.sum(1), rtol=1e-12, atol=1e-13)
                np.testing.assert_allclose(sums[:, 1].cpu().numpy(), (ref * ref).sum(1), rtol=1e-12, atol=1e-15)
                assert torch.equal(sums, antq_lib.moments(xt, rows, K, per_row))          # fixed order: same bits
                got = antq_lib.xmax_3sigma(xt, rows, K, per_row).cpu().numpy()
                want = oracle.three_sigma(xh, per_row)
                t2 = xt.reshape(rows, -1) if per_row else xt.reshape(1, -1)
                mean, std = t2.mean(dim=-1), t2.std(dim=-1)
                tor = torch.maximum((mean + 3 * std).abs(), (mean - 3 * std).abs()).float().cpu().numpy()
                assert got.shape == want.shape == tor.shape
                nan = np.isnan(want)
                assert np.array_equal(np.isnan(got), nan) and np.array_equal(np.isnan(tor), nan)
                assert nan.all() == (ref.shape[1] == 1)                    # one element: unbiased std is 0 / 0
                tol = 2.0 ** -7 if bf16 else 2e-6            # bf16: one rounding step of the tensor's dtype at most
                np.testing.assert_allclose(got[~nan], want[~nan], rtol=tol)
                np.testing.assert_allclose(got[~nan], tor[~nan], rtol=tol, atol=1e-30)
                if bf16 and (~nan).any():                    # ... and that only where torch's fp32 sum sits on a rounding boundary
                    assert (got[~nan] == tor[~nan]).mean() >= 0.9
    # an unaligned view (storage offset 1 element) goes through the element path: same sums
    base = torch.randn(4097, device=dev)
    v = base[1:]
    s1 = antq_lib.moments(v, 1, 4096, False)
    np.testing.assert_allclose(s1[0, 0].item(), v.double().sum().item(), rtol=1e-12)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_weights_at_rest_forward_is_bit_identical(antq_lib, dev, tree, capsys):
    """quant_utils.set_weights_at_rest(model): the weight quantisers' launches go out unordered (ANTQ_FLAG_UNORDERED); a
    calibrated model's forward must produce the same bits with the flag on and off, also right after other kernels were
    queued on the stream (the activations it does NOT apply to are produced by them)."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4, w_up=150, a_up=150))
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(1024, 2048), nn.GELU(), nn.Linear(2048, 1024), nn.GELU(), nn.Linear(1024, 64))
    model = qmod.quantize_model(net).to(dev).eval()
    qutil.enable_quantization(model)
    x = torch.randn(32, 1024, device=dev)
    with torch.no_grad():
        y0 = model(x)                       # calibration
        y1 = model(x)
        qutil.set_weights_at_rest(model, True, resident=False)
        assert all(m.weights_at_rest for m in model.modules() if hasattr(m, "weights_at_rest"))
        for _ in range(5):
            junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 64, device=dev)      # work in flight on the stream
            y2 = model(x)
            assert torch.equal(y2, y1)
        qutil.set_weights_at_rest(model, False)
        assert torch.equal(model(x), y1) and torch.equal(y0, y1)
        # weights rewritten by work STILL IN FLIGHT (an in-place update, a device-side load_state_dict): the version counter
        # moves, so the first forward afterwards launches ordered -- and later ones, unordered again, agree with it
        qutil.set_weights_at_rest(model, True, resident=False)
        model(x)
        lins = [m for m in model.modules() if isinstance(getattr(m, "weight", None), torch.Tensor) and hasattr(m, "quant_weight")]
        new_w = [torch.randn_like(m.weight) * 0.05 for m in lins]
        ref_model_out = None
        for rep in range(3):
            big = torch.randn(8192, 8192, device=dev)
            for m, w in zip(lins, new_w):
                big = big * 1.0001                                  # a few ms of queued work ahead of the copies
                m.weight.copy_(w * (1.0 + 0.1 * rep))
            y_a = model(x)                                          # first forward after the change: ordered launches
            y_b = model(x)                                          # unordered again
            qutil.set_weights_at_rest(model, False)
            y_ref = model(x)
            qutil.set_weights_at_rest(model, True, resident=False)
            assert torch.equal(y_a, y_ref) and torch.equal(y_b, y_ref), rep
    # a model moved to bf16 AFTER calibration: alpha is a bf16 Parameter now and is converted to float32 on every forward
    # (a kernel in flight right before the launch) -- such launches stay ordered
    mb = qmod.quantize_model(nn.Sequential(nn.Linear(1024, 2048), nn.GELU(), nn.Linear(2048, 64))).to(dev).eval()
    qutil.enable_quantization(mb)
    with torch.no_grad():
        mb(x)
        mb = mb.bfloat16()
        xb = x.bfloat16()
        ref_b = mb(xb)
        qutil.set_weights_at_rest(mb, True, resident=False)
        for _ in range(4):
            junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 64, device=dev)
            assert torch.equal(mb(xb), ref_b)
        qw = [m.quant_weight for m in mb.modules() if hasattr(m, "quant_weight")]
        assert all(q._alpha32 is not None and q._alpha32.dtype == torch.float32 for q in qw)     # (the cached scales are in use)
        for q in qw:                                   # an alpha edit is noticed: copy refreshed, that launch ordered
            q.alpha.mul_(1.25)
        y_new = mb(xb)
        qutil.set_weights_at_rest(mb, False)
        assert torch.equal(mb(xb), y_new) and not torch.equal(y_new, ref_b)
        qutil.set_weights_at_rest(mb, True, resident=False)
        assert torch.equal(mb(xb), y_new)
    # the flag set BEFORE the first (calibrating) forward: calibration writes alpha a moment before the launch
    net2 = nn.Sequential(nn.Linear(1024, 512), nn.GELU(), nn.Linear(512, 64))
    m2 = qmod.quantize_model(net2).to(dev).eval()
    qutil.enable_quantization(m2)
    qutil.set_weights_at_rest(m2, True, resident=False)
    with torch.no_grad():
        z0 = m2(x)
        z1 = m2(x)
        qutil.set_weights_at_rest(m2, False)
        assert torch.equal(m2(x), z1) and torch.equal(z0, z1)
    capsys.readouterr()
    del junk


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_weights_at_rest_under_inference_mode_and_on_temporaries(antq_lib, dev, tree, capsys):
    """(1) a bf16 model run under torch.inference_mode() with weights at rest -- tensors created inside the
    forward have no version counter; the mode must neither raise nor cache such a tensor; (2) a weight quantiser called on
    a TEMPORARY (a fresh tensor object per call whose address the caching allocator recycles) never launches unordered;
    (3) a codebook change without touching weight or alpha makes the next launch ordered."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4, w_up=150, a_up=150))
    torch.manual_seed(5)
    x = torch.randn(16, 1024, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        net = nn.Sequential(nn.Linear(1024, 2048), nn.GELU(), nn.Linear(2048, 64))
        model = qmod.quantize_model(net).to(dev).eval()
        qutil.enable_quantization(model)
        with torch.no_grad():
            model(x)                                    # calibration (fp32)
            model = model.to(dt)
            xd = x.to(dt)
            ref = model(xd)
        qutil.set_weights_at_rest(model, True, resident=False)
        with torch.inference_mode():
            for _ in range(4):
                assert torch.equal(model(xd), ref)
        with torch.no_grad():                           # ... and back outside: still the same bits
            assert torch.equal(model(xd), ref)
        qw = [m.quant_weight for m in model.modules() if hasattr(m, "quant_weight")]
        assert all(not torch.is_inference(q._rest_out) for q in qw)
        if dt != torch.float32:
            assert all(q._alpha32 is None or not torch.is_inference(q._alpha32) for q in qw)
    # (2) temporaries: same address, same version 0, a new object each time -> never at rest
    q = qw[0]
    w = next(m for m in model.modules() if hasattr(m, "quant_weight")).weight
    with torch.no_grad():
        seen = set()
        for _ in range(6):
            tmp = (w * 1.0)                             # a kernel in flight writes tmp right before the quantiser reads it
            seen.add(tmp.data_ptr())
            assert q._at_rest(tmp) is False
            del tmp
        assert len(seen) < 6                            # (the allocator did recycle an address: the case the advice describes)
        assert q._at_rest(w) is False and q._at_rest(w) is True
        # (3) a new plan object / gmax with weight and alpha untouched
        q._gmax = q._gmax * 2.0
        assert q._at_rest(w) is False and q._at_rest(w) is True
        q._gmax = q._gmax / 2.0
    import pickle
    st = pickle.loads(pickle.dumps(q.__getstate__()["_rest_src"]))
    assert st is None
    capsys.readouterr()


def test_far_clipped_elements_keep_the_tables_decision(antq_lib, oracle, dev):
    """Elements clipped beyond twice the outermost grid value (|x / s| >= xlim): the straight-through arithmetic
    ((q - d) + d) * s is no longer q * s out there, but the table's decision still is right -- the row-table kernels redo
    only the arithmetic for them (round 3; the lane kernel since round 2).  Small alphas put 1 ... 60 % of the elements
    there, of both signs, next to NaN / Inf / 1e30 (which must still take the literal scan): per tensor through both
    kernels (knob 5), batched, unordered; ANT signed / unsigned and OliVe with the pair rule; fp32 and bf16."""
    import torch
    rng = np.random.default_rng(41)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    gol = np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])
    for rows, K in ((24, 4096), (6, 8200), (40, 1024)):
        for div in (2.5, 8.0, 40.0, 900.0):
            x = make_x(rng, rows, K, specials=True)
            for bf16 in (False, True):
                for g, gmax, ovp, uns in ((G["flint_b4_s"], 10.0, False, False), (G["flint_b4_u"], 10.0, False, True),
                                          (G["int_b4_s"], 10.0, False, False), (gol, 32.0, True, False)):
                    xx = np.abs(x) if uns else x
                    alpha = (safe_absmax(xx).max(1) / np.float32(div)).astype(np.float32)
                    run_case(antq_lib, oracle, dev, xx, alpha, g, gmax, True, ovp, bf16)
                    # batched + unordered
                    xh = oracle.f32_to_bf16(xx) if bf16 else xx
                    ref, _ = oracle.forward(xh, alpha, g, gmax, ovp)
                    xt, a_t, plan = to_dev(xh, dev, bf16), torch.from_numpy(alpha).to(dev), antq_lib.plan_for(g)
                    ob = torch.zeros_like(xt)
                    antq_lib.Batch([(xt, ob, a_t, plan, gmax, rows, K, True)], ovp=ovp).run()
                    ou = antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, True, ovp=ovp, unordered=True, out=torch.empty_like(xt))
                    for o in (ob, ou):
                        ok = bf16_same(bf16_bits(o), ref, oracle) if bf16 else f32_same(o.cpu().numpy(), ref)
                        assert ok, (rows, K, div, bf16, ovp, uns)


def test_float64_forward_is_the_reference_sequence_around_the_operator(antq_lib, oracle, dev, capsys):
    """A float64 tensor through the module surface: the reference's kernel is dispatched for double and narrows to float
    inside (KQ/quant_kernel.cu:51, :28) while the ops around it stay in double; `core.fake_quant` does the same -- the
    reference's op sequence (AQ:535-551 / OQ:294-330) in float64 around `antq_nearest` -- checked bit for bit against that
    sequence restated in numpy float64 on the oracle's scan, per channel and per tensor, ANT and OliVe with the pair rule;
    then a TensorQuantizer fed a double tensor calibrates (on the float32 image) and returns float64."""
    import torch
    from ant_quantization_amd import core
    rng = np.random.default_rng(64)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    gol = np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])
    for shape in ((16, 256), (7, 33), (64, 3, 7, 7)):
        x = rng.standard_normal(shape) * 0.05
        x.reshape(-1)[::19] *= 25
        xt = torch.from_numpy(x).to(dev)
        for g, gmax, ovp in ((G["flint_b4_s"], 10.0, False), (G["int_b4_s"], 10.0, False), (gol, 32.0, True)):
            plan = antq_lib.plan_for(g)
            for per_channel in (True, False):
                x2 = x.reshape(shape[0], -1)
                alpha = (np.abs(x2).max(1) * 0.8 if per_channel else np.array(np.abs(x).max() * 0.8)).astype(np.float32)
                a_t = torch.from_numpy(alpha.reshape(-1, 1) if per_channel else alpha.reshape(())).to(dev)
                out = core.fake_quant(xt, a_t, plan, gmax, per_channel, ovp=ovp)
                assert out.dtype == torch.float64 and out.shape == xt.shape
                scale = alpha.astype(np.float64).reshape(-1, 1) / gmax if per_channel else alpha.astype(np.float64) / gmax
                d = (x2 / scale).reshape(shape) if per_channel else x / scale
                q = oracle.nearest(d.reshape(-1), g.astype(np.float64))[0]
                if ovp:                                          # OQ:311-320
                    mask = np.abs(q) > 32
                    vo = np.roll(mask, 1)
                    vo[::2] = False
                    ve = np.roll(mask & ~vo, -1)
                    ve[1::2] = False
                    q = q * (~(ve | vo))
                q = q.reshape(shape)
                t = (q - d) + d
                ref = (t.reshape(shape[0], -1) * scale).reshape(shape) if per_channel else t * scale
                got = out.cpu().numpy()
                assert np.array_equal(got.view(np.uint64), np.ascontiguousarray(ref).view(np.uint64)), (shape, ovp, per_channel)
    from ant_quantization_amd.ant import quant_modules as qm
    from ant_quantization_amd.olive import quant_modules as oqm
    w = torch.from_numpy(rng.standard_normal((32, 512)) * 0.03).to(dev)
    for mod, mode, kw in ((qm, "ant-int-flint", {}), (oqm, "ant-int-flint", dict(w_up=250, a_up=250))):
        q = mod.TensorQuantizer(mode=mode, bit=4, is_signed=True, is_enable=True, args=_args(**kw)).to(dev)
        q.name = "f64"
        q.alpha.data = torch.ones(32, 1, device=dev)
        y = q(w)
        assert y.dtype == torch.float64 and torch.isfinite(y).all()
        y32 = q(w.float())                                       # same calibrated state, float32 input: the fused kernel
        np.testing.assert_allclose(y.detach().cpu().numpy(), y32.detach().double().cpu().numpy(), rtol=1e-6, atol=1e-12)
    capsys.readouterr()


def _ref_checkpoint(fx, prefixes, dev, strip):
    """The state dict the reference wrote (tests/golden/*_ckpt.npz, keys 'module.'-prefixed as ImageNet/main.py saves a
    DistributedDataParallel model); later prefixes override earlier ones.  strip: drop the 7 characters the way
    main.py:151-157 does before load_ant_state_dict / load_state_dict."""
    import torch
    check = {}
    for pre in prefixes:
        tag = pre + "sd__"
        for k in fx.files:
            if k.startswith(tag):
                key = k[len(tag):]
                check[key[7:] if strip else key] = torch.from_numpy(np.array(fx[k])).to(dev)
    return check


def _ckpt_nets(tree):
    import torch.nn as nn
    if tree == "ant":
        return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(512, 32), nn.ReLU(),
                             nn.Linear(32, 10))
    from transformers import pytorch_utils
    return nn.Sequential(nn.Linear(64, 128), nn.GELU(), nn.Linear(128, 64), pytorch_utils.Conv1D(32, 64))


def _check_recorded_quantizers(fx, pre, model, dev, weight_of):
    """Every TensorQuantizer of `model` against the tensors the reference's own forward produced: bit for bit."""
    import torch
    for name in [str(n) for n in fx[pre + "quantizers"]]:
        q = model.get_submodule(name)
        if "quant_input" in name:
            inp = torch.from_numpy(fx[pre + "q__%s__in" % name]).to(dev)
        else:
            inp = weight_of(model, name).detach()
        with torch.no_grad():
            out = q(inp)
        assert f32_same(out.cpu().numpy(), fx[pre + "q__%s__out" % name]), (pre, name)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_checkpoint_written_by_the_reference_loads_and_reproduces_its_forward(antq_lib, dev, tree, capsys):
    """N2 as a wire test: a checkpoint written by the REFERENCE'S quant_model / quant_modules (make_golden.py --tree
    *_ckpt: its quantize_model, its calibration, its state_dict()) goes through load_ant_state_dict +
    load_state_dict(strict=True) of a freshly rewritten model (AQ/quant_model.py:151-154, ImageNet/main.py:151-162), with
    the 'module.' prefix stripped and -- loaded into a wrapper holding the model as `.module` -- with it; no calibration
    runs, every quantiser reproduces the reference's recorded tensors bit for bit (weights, the calibration batch, a
    second batch), the model output agrees within GEMM rounding.  ANT: also the mixed-precision checkpoint after the
    reference's set_8_bit_layer_l (a 256-entry quant_grid next to 16-entry ones)."""
    import torch
    import torch.nn as nn
    if tree == "ant":
        from ant_quantization_amd.ant import quant_model as qmod, quant_utils as qutil
        args = _args(mode="ant-int-pot-flint", wbit=4, abit=4)
    else:
        pytest.importorskip("transformers")
        from ant_quantization_amd.olive import quant_model as qmod, quant_utils as qutil
        args = _args(mode="ant-int-flint", wbit=4, abit=4, w_up=250, a_up=250)
    fx = golden("%s_ckpt.npz" % tree)
    qutil.set_quantizer(args)

    def weight_of(model, qname):
        return model.get_submodule(qname.rsplit(".", 1)[0]).weight

    variants = [(["a__"], ["a__", "a2__"])] + ([(["a__", "b__"], ["b__"])] if tree == "ant" else [])
    for prefixes, recorded in variants:
        for strip in (True, False):
            torch.manual_seed(99)                      # other random weights: everything must come from the checkpoint
            model = qmod.quantize_model(_ckpt_nets(tree)).to(dev).eval()
            qutil.enable_quantization(model)
            check = _ref_checkpoint(fx, prefixes, dev, strip)
            holder = model
            if not strip:
                holder = nn.Module()
                holder.module = model
            qmod.load_ant_state_dict(holder, check)
            holder.load_state_dict(check, strict=True)
            capsys.readouterr()
            for pre in recorded:
                _check_recorded_quantizers(fx, pre, model, dev, weight_of)
                with torch.no_grad():
                    y = model(torch.from_numpy(fx[pre + "x"]).to(dev))
                np.testing.assert_allclose(y.cpu().numpy(), fx[pre + "y"], rtol=2e-4, atol=2e-5)
            assert "-bit" not in capsys.readouterr().out           # no calibration line: the checkpoint's state was used
            if prefixes[-1] == "b__":
                assert int(model[3].quant_weight.bit) == 8 and model[3].quant_weight.quant_grid.numel() == 256
                assert int(model[0].quant_weight.bit) == 4 and model[0].quant_weight.quant_grid.numel() == 16


def test_multihead_attention_quantizer_vs_reference_fixture(antq_lib, dev, capsys):
    """N4: the reference's MultiheadAttentionQuantizer (AQ/multihead_attention.py:486-686; quantiser call sites :459,
    :663-668) rewritten, calibrated and run by the reference itself; its checkpoint loaded here.  The four quantisers
    reproduce the recorded tensors bit for bit (in-projection weight, query, out-projection weight, attention output),
    output and averaged attention weights agree within softmax / GEMM rounding.  Then the same from scratch: our own
    calibration on the recorded query picks the reference's alpha and grid."""
    import torch
    import torch.nn as nn
    from ant_quantization_amd.ant import quant_model as qmod, quant_utils as qutil
    from ant_quantization_amd.ant.multihead_attention import MultiheadAttentionQuantizer
    fx = golden("ant_ckpt.npz")
    qutil.set_quantizer(_args(mode="ant-int-pot-flint", wbit=4, abit=4))

    def weight_of(model, qname):
        mha = model.get_submodule(qname.rsplit(".", 1)[0])
        return mha.in_proj_weight if "in_quant" in qname else mha.out_proj_weight

    for pre, bf in (("m__", False), ("mb__", True)):
        torch.manual_seed(7)
        model = qmod.quantize_model(nn.Sequential(nn.MultiheadAttention(64, 4, batch_first=bf))).to(dev).eval()
        assert type(model[0]) is MultiheadAttentionQuantizer
        qutil.enable_quantization(model)
        check = _ref_checkpoint(fx, [pre], dev, True)
        qmod.load_ant_state_dict(model, check)
        model.load_state_dict(check, strict=True)
        capsys.readouterr()
        _check_recorded_quantizers(fx, pre, model, dev, weight_of)
        x = torch.from_numpy(fx[pre + "x"]).to(dev)
        with torch.no_grad():
            y, w = model[0](x, x, x)
        assert "-bit" not in capsys.readouterr().out
        np.testing.assert_allclose(y.cpu().numpy(), fx[pre + "y"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(w.cpu().numpy(), fx[pre + "attn_weights"], rtol=2e-4, atol=2e-6)
        # from scratch: same weights, our calibration on the same query
        torch.manual_seed(7)
        ma = nn.MultiheadAttention(64, 4, batch_first=bf)
        with torch.no_grad():
            ma.in_proj_weight.copy_(torch.from_numpy(fx[pre + "sd__module.0.in_proj_weight"]))
            ma.in_proj_bias.copy_(torch.from_numpy(fx[pre + "sd__module.0.in_proj_bias"]))
            ma.out_proj.weight.copy_(torch.from_numpy(fx[pre + "sd__module.0.out_proj_weight"]))
            ma.out_proj.bias.copy_(torch.from_numpy(fx[pre + "sd__module.0.out_proj_bias"]))
        own = qmod.quantize_model(nn.Sequential(ma)).to(dev).eval()
        qutil.enable_quantization(own)
        with torch.no_grad():
            y2, _ = own[0](x, x, x)
        capsys.readouterr()
        for qn in ("in_quant_weight", "in_quant_input", "out_quant_weight"):
            q = getattr(own[0], qn)
            assert np.array_equal(q.quant_grid.cpu().numpy(), fx[pre + "sd__module.0.%s.quant_grid" % qn]), qn
            np.testing.assert_allclose(q.alpha.detach().cpu().numpy().reshape(-1),
                                       fx[pre + "sd__module.0.%s.alpha" % qn].reshape(-1), rtol=2e-2)
        np.testing.assert_allclose(y2.cpu().numpy(), fx[pre + "y"], rtol=0.2, atol=0.05)


def test_nearest_fast_path_equals_literal_scan(antq_lib, oracle, dev):
    """antq_nearest analyses the (device-resident) grid per workgroup and binary-searches; it must agree with
    the literal scan on every kind of grid: sorted, two sorted runs (OliVe), unsorted, duplicates, hostile."""
    import torch
    rng = np.random.default_rng(12)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    grids_ = [G[k] for k in ("flint_b4_s", "flint_b4_u", "int_b4_s", "pot_b4_u", "apot_b4_s", "int_b8_s", "int_b8_u",
                             "flint_b6_s", "pot_b6_s", "float_b6_u", "int_b2_s")]
    grids_ += [np.concatenate([O["%s_b4_%s" % (t, s)], O["outlier_b4_%s" % s]]) for t in ("int", "flint") for s in "su"]
    grids_ += [np.float32([3, 1, 2, 1, 3, -7]), np.float32([1.0, 1.0000001, 5.0]), np.float32([0.0, 0.0, 0.0]),
               np.float32([5.0]), np.float32([-1.0, 1.0]), np.float32([0, 1e-3, 1e4]),
               np.sort(rng.standard_normal(200).astype(np.float32) * 3), rng.standard_normal(40).astype(np.float32)]
    for g in grids_:
        hi = float(np.abs(g).max()) + 1.0
        x = np.concatenate([
            rng.standard_normal(20000).astype(np.float32) * np.float32(hi / 2),
            rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32),
            np.repeat(g, 3) + np.tile(np.float32([0, 1e-6, -1e-6]), g.size),
            ((g[:-1].astype(np.float64) + g[1:]) / 2).astype(np.float32) if g.size > 1 else np.float32([0]),
            np.float32([0.0, -0.0, 1e5, -1e5, 102400.0, 2e5, np.inf, -np.inf, np.nan, 36000.0, -36000.0, 65535.0])])
        with np.errstate(all="ignore"):
            zr, jr = oracle.nearest(x, g)
        z, j = antq_lib.nearest(to_dev(x, dev), to_dev(g, dev), want_idx=True)
        assert f32_same(z.cpu().numpy(), zr), g[:6]
        assert np.array_equal(j.cpu().numpy().astype(np.int32), jr), g[:6]


@pytest.mark.parametrize("gname", ["flint_b4_s", "int_b4_s", "pot_b4_u", "olive_flint", "olive_int"])
def test_x_domain_thresholds_dense_sweep(antq_lib, oracle, dev, gname):
    """The x-domain row kernel moves every decision threshold into the x domain per row.  Probe it where it
    can go wrong: +-96 ulps around (mid-point * scale) of every pair of adjacent grid values, for 256 rows
    with awkward random scales -- bit-exact values and indices against the oracle, fp32 and bf16."""
    rng = np.random.default_rng(99)
    if gname.startswith("olive"):
        O = golden("olive_grids.npz")
        gn = O[gname.split("_")[1] + "_b4_s"]
        g = np.concatenate([gn, O["outlier_b4_s"]])
        gmax, ovp = float(gn.max()), True
    else:
        g = golden("ant_grids.npz")[gname]
        gmax, ovp = float(g.max()), False
    plan = antq_lib.plan_for(g)
    assert plan.is_table and plan.host[:80].view(np.uint32)[16] == 1      # x-domain eligible
    gs = np.unique(g)
    mids = ((gs[:-1].astype(np.float64) + gs[1:]) / 2)
    rows, K = 256, 4096
    alpha = (np.exp(rng.uniform(np.log(1e-3), np.log(50.0), rows)) * rng.uniform(1.0, 2.0, rows)).astype(np.float32)
    scale = (alpha / np.float32(gmax)).astype(np.float32)
    x = (rng.standard_normal((rows, K)) * (alpha[:, None] / 3)).astype(np.float32)
    per = 2 * 96 + 1
    offs = np.arange(-96, 97, dtype=np.int64)
    for r in range(rows):
        centers = (mids * float(scale[r])).astype(np.float32)
        centers = centers[centers != 0]
        n = min(len(centers), K // per)
        sel = rng.choice(len(centers), n, replace=False)
        for k, ci in enumerate(sel):
            c = centers[ci]
            bits = np.abs(c).view(np.uint32).astype(np.int64) + offs
            vals = bits.astype(np.uint32).view(np.float32) * np.sign(c)
            x[r, k * per:(k + 1) * per] = vals
    if gname.endswith("_u"):
        x = np.abs(x)
    run_case(antq_lib, oracle, dev, x, alpha, g, gmax, True, ovp, False)
    run_case(antq_lib, oracle, dev, x, alpha, g, gmax, True, ovp, True)


def test_c4_llama70b_sized_tensor_sampled_rows(antq_lib, oracle, dev):
    """C4: the largest tensor of the synthetic 70B stack, [28672, 8192] bf16 (470 MB in, 470 MB out),
    OliVe flint-4 with outlier-victim pairs.  The oracle checks 96 rows drawn from the whole height (values and
    indices, bit-exact); the pair invariant and the one-launch batch path are checked on every element."""
    import torch
    O = golden("olive_grids.npz")
    gn, go = O["flint_b4_s"], O["outlier_b4_s"]
    grid = np.concatenate([gn, go])
    plan = antq_lib.plan_for(grid)
    R, K = 28672, 8192
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(R, K, device=dev, generator=gen) * 0.02
    m = torch.rand(x.shape, device=dev, generator=gen) < 0.001
    x[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 64, generator=gen)
    xb = x.bfloat16()
    alpha = (3 * x.std(1)).contiguous()
    del x, m
    out, idx = antq_lib.fakequant(xb, alpha, plan, 32.0, R, K, True, ovp=True, want_idx=True)
    rows = np.unique(np.concatenate([np.arange(16), np.arange(R - 16, R), np.random.default_rng(5).integers(0, R, 64)]))
    rt = torch.from_numpy(rows).to(dev)
    ref, ridx = oracle.forward(bf16_bits(xb[rt]), alpha[rt].cpu().numpy(), grid, 32.0, True)
    assert bf16_same(bf16_bits(out[rt]), ref, oracle)
    assert np.array_equal(idx[rt].cpu().numpy().astype(np.int32), ridx)
    pairs = idx.view(-1, 2)
    is_out, is_vic = pairs >= gn.size, pairs == antq_lib.IDX_VICTIM
    assert is_out.any() and not (is_out[:, 0] & is_out[:, 1]).any()
    assert (is_vic.any(1) == is_out.any(1)).all() and (out.view(-1, 2)[is_vic] == 0).all()
    del pairs, is_out, is_vic, idx
    ob = torch.empty_like(xb)
    antq_lib.Batch([(xb, ob, alpha, plan, 32.0, R, K, True)], ovp=True).run()
    assert torch.equal(ob.view(torch.int16), out.view(torch.int16))


def test_weight_bank_one_launch_for_all_layers(antq_lib, dev):
    """WeightBank: every calibrated weight quantiser of a model served from ONE batched launch -- same bits as the
    per-layer path, refreshed when a weight or alpha changes, bypassed when gradients are wanted."""
    import torch
    from ant_quantization_amd.weight_bank import WeightBank
    from ant_quantization_amd.ant import quant_model as aqm, quant_utils as aqu
    from ant_quantization_amd.olive import quant_model as oqm, quant_utils as oqu

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = torch.nn.Conv2d(3, 16, 3, padding=1)        # K = 27: ragged, element-granular blocks of the same launch
            self.c2 = torch.nn.Conv2d(16, 32, 3, padding=1)       # K = 144
            self.f1 = torch.nn.Linear(32, 512)
            self.f2 = torch.nn.Linear(512, 10)

        def forward(self, x):
            x = torch.relu(self.c2(torch.relu(self.c1(x)))).mean((2, 3))
            return self.f2(torch.relu(self.f1(x)))

    for qmod, qutil, mode in ((aqm, aqu, "ant-int-flint"), (oqm, oqu, "ant-int-flint")):
        qutil.set_quantizer(_args(mode=mode, wbit=4, abit=4))
        torch.manual_seed(3)
        model = qmod.quantize_model(Net()).to(dev).eval()
        qutil.enable_quantization(model)
        x = torch.randn(8, 3, 16, 16, device=dev)
        with torch.no_grad():
            model(x)                       # calibrate
            y_ref = model(x)               # per-layer launches
        bank = WeightBank(model)
        assert len(bank.entries) == 4 and not bank.skipped
        with torch.no_grad():
            y1 = model(x)
            y2 = model(x)
        assert torch.equal(y1, y_ref) and torch.equal(y2, y_ref) and bank.launches == 1
        for e in bank.entries.values():    # each resident buffer == what the quantiser produces on its own
            e["q"]._bank = None
            with torch.no_grad():
                assert torch.equal(e["q"](e["mod"].weight), e["out"])
            e["q"]._bank = bank
        # an optimiser-style in-place update of one weight refreshes the bank once, at the first stale layer
        with torch.no_grad():
            model.f1.weight.mul_(1.01)
            y3 = model(x)
        assert bank.launches == 2
        bank.detach()
        with torch.no_grad():
            assert torch.equal(model(x), y3)
        assert not torch.equal(y3, y_ref)
        # with gradients wanted (ANT QAT) the bank steps aside
        bank = WeightBank(model)
        model(x).sum().backward() if qmod is aqm else None
        assert bank.launches == 0
        if qmod is aqm:
            assert model.f1.weight.grad is not None and model.f1.quant_weight.alpha.grad is not None


def test_entry_points_capture_into_a_hip_graph(antq_lib, dev):
    """The C ABI neither allocates nor synchronises: a per-tensor launch and a batched launch captured into a
    hipGraph replay to the same bits on fresh data in the captured buffers."""
    import torch
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    plan.dev(dev)                                     # table upload happens outside the capture
    torch.manual_seed(11)
    xs = [torch.randn(256, 1024, device=dev) * 0.02 for _ in range(3)]
    al = [x.abs().amax(1).contiguous() for x in xs]
    outs = [torch.empty_like(x) for x in xs]
    one = torch.empty_like(xs[0])
    bt = antq_lib.Batch([(x, o, a, plan, 10.0, 256, 1024, True) for x, a, o in zip(xs, al, outs)])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                        # warm-up on the side stream, as graph capture wants
        bt.run()
        antq_lib.fakequant(xs[0], al[0], plan, 10.0, 256, 1024, True, out=one)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bt.run()
        antq_lib.fakequant(xs[0], al[0], plan, 10.0, 256, 1024, True, out=one)
    for x in xs:                                      # new data, same buffers
        x.mul_(1.7).add_(0.003)
    for x, a in zip(xs, al):
        a.copy_(x.abs().amax(1))
    for o in outs + [one]:
        o.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for x, a, o in zip(xs, al, outs):
        assert torch.equal(o, antq_lib.fakequant(x, a, plan, 10.0, 256, 1024, True))
    assert torch.equal(one, outs[0])
    # unordered launches (ANTQ_FLAG_UNORDERED) inside a capture: whatever the runtime makes of the any-order flag in a graph
    # node, the replay must produce the same bits (weights at rest: x / alpha are not written by the graph)
    bufs = [torch.zeros_like(x) for x in xs]
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for x, a, b in zip(xs, al, bufs):
            antq_lib.fakequant(x, a, plan, 10.0, 256, 1024, True, out=b, unordered=True)
    for b in bufs:
        b.zero_()
    g2.replay()
    torch.cuda.synchronize()
    for b, o in zip(bufs, outs):
        assert torch.equal(b, o)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 12))))
def test_random_grids_fuzz(antq_lib, oracle, dev, seed):
    """Arbitrary codebooks, not just the reference's generators: random size, order, duplicates, signed zeros,
    near-ties, geometric or uniform spacing, with and without entries beyond 32 (OliVe's outlier test).  Whatever
    plan the builder picks (x-domain table, d-domain table, literal scan), values and indices match the oracle."""
    rng = np.random.default_rng(1000 + seed)
    for case in range(6):
        m = int(rng.choice([2, 3, 5, 8, 15, 16, 29, 31, 64, 200]))
        kind = rng.integers(0, 4)
        if kind == 0:
            g = rng.uniform(-40, 40, m)
        elif kind == 1:
            g = np.sign(rng.standard_normal(m)) * np.exp(rng.uniform(np.log(0.05), np.log(400), m))
        elif kind == 2:
            g = np.arange(m) * rng.uniform(0.1, 3.0) + rng.uniform(-20, 0)
        else:
            g = np.round(rng.uniform(-12, 12, m) * 4) / 4
        g = g.astype(np.float32)
        if rng.random() < 0.5:
            g = np.sort(g)
        if rng.random() < 0.4 and m > 3:
            g[rng.integers(0, m)] = g[rng.integers(0, m)]          # duplicate
        if rng.random() < 0.3:
            g[rng.integers(0, m)] = -0.0
        if rng.random() < 0.2 and m > 2:
            i = rng.integers(0, m - 1)
            g[i + 1] = np.nextafter(g[i], np.float32(np.inf))       # two entries one ulp apart
        gmax = float(np.abs(g).max()) if rng.random() < 0.5 else float(max(g.max(), 0.5))
        rows, K = [(8, 1024), (3, 4096), (16, 200), (5, 33), (64, 16), (2, 8192)][case]
        x = make_x(rng, rows, K, specials=bool(case % 2))
        x *= np.float32(rng.uniform(0.5, 60))
        alpha = (safe_absmax(x).max(1) * rng.uniform(0.3, 1.2, rows) + 1e-6).astype(np.float32)
        ovp = bool(rng.random() < 0.5)
        bf16 = bool(rng.random() < 0.5)
        run_case(antq_lib, oracle, dev, x, alpha, g, gmax, True, ovp, bf16)
        run_case(antq_lib, oracle, dev, x, np.float32(alpha.mean()), g, gmax, False, ovp, not bf16)


def _random_grid(rng, m):
    kind = rng.integers(0, 4)
    if kind == 0:
        g = rng.uniform(-40, 40, m)
    elif kind == 1:
        g = np.sign(rng.standard_normal(m)) * np.exp(rng.uniform(np.log(0.05), np.log(400), m))
    elif kind == 2:
        g = np.arange(m) * rng.uniform(0.1, 3.0) + rng.uniform(-20, 0)
    else:
        g = np.round(rng.uniform(-12, 12, m) * 4) / 4
    g = g.astype(np.float32)
    if rng.random() < 0.6:
        g = np.sort(g)
    if rng.random() < 0.4 and m > 3:
        g[rng.integers(0, m)] = g[rng.integers(0, m)]
    if rng.random() < 0.3:
        g[rng.integers(0, m)] = -0.0
    return g


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 8))))
def test_random_grids_fuzz_other_entry_points(antq_lib, oracle, dev, seed):
    """The same arbitrary codebooks through the remaining entry points: antq_nearest (fp32 / fp64 / bf16, up to 1024
    entries), antq_fakequant_dynamic, antq_search_sse + antq_search_pick, antq_fakequant_batch."""
    import torch
    from ant_quantization_amd import core
    rng = np.random.default_rng(5000 + seed)
    # --- nearest
    for m in (int(rng.choice([2, 7, 16, 31, 33, 100, 256, 509, 1024])) for _ in range(4)):
        g = _random_grid(rng, m)
        hi = float(np.abs(g).max()) + 1.0
        x = np.concatenate([rng.standard_normal(6000).astype(np.float32) * np.float32(hi / 2),
                            rng.integers(0, 2 ** 32, 3000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                            np.repeat(g, 2) + np.tile(np.float32([1e-6, -1e-6]), g.size),
                            ((g[:-1].astype(np.float64) + g[1:]) / 2).astype(np.float32)])
        with np.errstate(all="ignore"):
            zr, jr = oracle.nearest(x, g)
            z, j = antq_lib.nearest(to_dev(x, dev), to_dev(g, dev), want_idx=True)
            assert f32_same(z.cpu().numpy(), zr) and np.array_equal(j.cpu().numpy().astype(np.int32), jr), (m, "f32")
            z64 = antq_lib.nearest(to_dev(x.astype(np.float64), dev), to_dev(g.astype(np.float64), dev))
            assert f32_same(z64.cpu().numpy().astype(np.float32), zr), (m, "f64")
            xb = oracle.f32_to_bf16(x)
            zb_ref, _ = oracle.nearest(oracle.bf16_to_f32(xb), g)
            zb = antq_lib.nearest(to_dev(xb, dev, True), to_dev(g, dev))
            assert bf16_same(bf16_bits(zb), oracle.f32_to_bf16(zb_ref), oracle), (m, "bf16")
    # --- dynamic abs-max, clip search, batch
    jobs, refs = [], []
    for case in range(4):
        g = _random_grid(rng, int(rng.choice([3, 8, 15, 16, 29, 64])))
        if not (g.max() > 0):
            g[-1] = 1.5
        gmax = float(g.max())
        plan = antq_lib.plan_for(g)
        rows, K = [(16, 1024), (64, 64), (6, 4096), (33, 200)][case]
        x = make_x(rng, rows, K, specials=False) * np.float32(rng.uniform(0.5, 30))
        ratio = float(np.float32(rng.uniform(0.5, 1.1)))
        alpha = oracle.absmax(x, True, ratio)
        ref, ridx = oracle.forward(x, alpha, g, gmax)
        xt = to_dev(x, dev)
        out, a_dev, idx = antq_lib.fakequant_dynamic(xt, plan, gmax, rows, K, ratio=ratio, want_idx=True)
        assert np.array_equal(a_dev.cpu().numpy(), alpha) and f32_same(out.cpu().numpy(), ref), ("dynamic", case)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx)
        for per_row in (True, False):
            xmax = core.row_absmax(xt, per_row)
            best, al, ratios = core.clip_search(xt, xmax, per_row, 60, 130, 3, plan, gmax)
            rb, ra, trace = oracle.search_mse(x, xmax.cpu().numpy(), 60, 130, 3, g, gmax, False, per_row)
            r_, k_ = (rows, K) if per_row else (1, rows * K)
            sse = antq_lib.search_sse(xt, r_, k_, xmax, per_row, ratios, plan, gmax)
            np.testing.assert_allclose((sse / k_).float().cpu().numpy(), trace, rtol=3e-5, atol=1e-12)
            close = np.isclose(al.cpu().numpy(), ra, rtol=1e-6)
            if not close.all():           # only near-ties may pick a different candidate
                srt = np.sort(trace, axis=0)
                assert ((srt[1] - srt[0]) <= 1e-4 * srt[0])[~close].all(), ("search", case, per_row)
        a_t = torch.from_numpy(alpha).to(dev)
        jobs.append((xt, torch.empty_like(xt), a_t, plan, gmax, rows, K, True))
        refs.append(ref)
    antq_lib.Batch(jobs).run()
    for j, ref in zip(jobs, refs):
        assert f32_same(j[1].cpu().numpy(), ref)


def test_quantizer_end_to_end_wide_fixture_set(antq_lib, dev, capsys):
    """ant_select_wide.npz: 90 complete calibrations recorded from the reference's Python -- the modes the first
    fixture set leaves out (pot, float, float1..4, apot, type lists containing them, incl. the AQ:370-397 quirk),
    bit widths 2..7, two search windows; weights per channel, gelu / relu activations per tensor.  Picks are checked
    against the reference's own per-candidate scores (ant_select_wide_traces.npz), outputs for every row."""
    import torch
    from ant_quantization_amd.ant import quant_modules as qm
    sel, tr = golden("ant_select_wide.npz"), golden("ant_select_wide_traces.npz")
    n_rows = n_same = n_cases = 0
    led0 = len(calib_check.LEDGER)
    for k in [str(v) for v in sel["keys"]]:
        name, mode, b, win = k.split("__")
        bit, (lo, up) = int(b[1:]), map(int, win.split("_"))
        x_np = sel[name + "__x"]
        is_input = name != "w"
        q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                               args=_args(w_low=lo, a_low=lo, w_up=up, a_up=up)).to(dev)
        q.name = "golden"
        x = to_dev(np.ascontiguousarray(x_np), dev)
        if not is_input:
            q.alpha.data = torch.ones(x.shape[0], 1, device=dev)
        out = q(x)
        same = _check_calibration(antq_lib, dev, q, x, out, k, sel, tr, 95 if bit > 6 else lo, up, 1, False, 0.0)
        if same is None:
            continue
        n_cases += 1
        n_rows += same.size
        n_same += int(same.sum())
        np.testing.assert_allclose(q.mse.item(), sel[k + "__mse"], rtol=3e-3, err_msg=k)
    assert n_cases >= 88, n_cases
    capsys.readouterr()
    _pick_report("end to end, ant wide fixture set", n_same, n_rows, led0)


def test_olive_quantizer_end_to_end_wide_fixture_set(antq_lib, dev, capsys):
    """olive_select_wide.npz: OliVe calibrations recorded from the reference's Python -- bit widths 3..8, outliers on
    and off (`no_outlier`), two search windows, per-channel weights, per-tensor activations, an odd-numel tensor.
    Picks against the reference's own scores (olive_select_wide_traces.npz), whole-tensor outputs on its alpha."""
    import torch
    from ant_quantization_amd.olive import quant_modules as qm
    sel, tr = golden("olive_select_wide.npz"), golden("olive_select_wide_traces.npz")
    n_rows = n_same = n_cases = 0
    led0 = len(calib_check.LEDGER)
    for k in [str(v) for v in sel["keys"]]:
        name, mode, b, win, om = k.split("__")
        bit, (lo, up) = int(b[1:]), map(int, win.split("_"))
        x_np = sel[name + "__x"]
        is_input = name != "w"
        q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                               args=_args(w_low=lo, a_low=lo, w_up=up, a_up=up, no_outlier=(om == "noout"))).to(dev)
        q.name = "golden"
        x = to_dev(np.ascontiguousarray(x_np), dev)
        if not is_input:
            q.alpha.data = torch.ones(x.shape[0], 1, device=dev)
        out = q(x)
        same = _check_calibration(antq_lib, dev, q, x, out, k, sel, tr, lo, up, 2, om == "ovp",
                                  2e-6 if om == "ovp" else 0.0)
        if same is None:
            continue
        n_cases += 1
        n_rows += same.size
        n_same += int(same.sum())
        np.testing.assert_allclose(q.mse.item(), sel[k + "__mse"], rtol=5e-3, err_msg=k)
    assert n_cases >= 88, n_cases
    capsys.readouterr()
    _pick_report("end to end, olive wide fixture set", n_same, n_rows, led0)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_quantizer_end_to_end_long_rows_single_read_type_selection(antq_lib, dev, tree, capsys):
    """*_select_long.npz: complete `ant-...` calibrations recorded from the reference on rows of 1024 elements -- the
    shapes on which the quantiser selects the type on ONE read of the tensor (antq_search_sse_multi: 2, 3 and 4 candidate
    codebooks, the installed grid's search reused).  Picks against the reference's own scores, outputs for every row on
    the reference's alpha; and exactly one search launch per calibration."""
    import importlib
    import torch
    qm = importlib.import_module("ant_quantization_amd.%s.quant_modules" % tree)
    sel, tr = golden("%s_select_long.npz" % tree), golden("%s_select_long_traces.npz" % tree)
    calls = {"multi": 0, "single": 0}
    real_m, real_s = antq_lib.search_sse_multi, antq_lib.search_sse

    def cm(*a, **k):
        calls["multi"] += 1
        return real_m(*a, **k)

    def cs(*a, **k):
        calls["single"] += 1
        return real_s(*a, **k)

    antq_lib.search_sse_multi, antq_lib.search_sse = cm, cs
    n_cases = 0
    try:
        for k in [str(v) for v in sel["keys"]]:
            parts = k.split("__")
            name, mode, b, win = parts[:4]
            om = parts[4] if tree == "olive" else "noout"
            bit, (lo, up) = int(b[1:]), map(int, win.split("_"))
            x_np = sel[name + "__x"]
            is_input = name != "w"
            kw = dict(w_low=lo, a_low=lo, w_up=up, a_up=up)
            if tree == "olive":
                kw["no_outlier"] = om == "noout"
            q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                                   args=_args(**kw)).to(dev)
            q.name = "golden"
            x = to_dev(np.ascontiguousarray(x_np), dev)
            if not is_input:
                q.alpha.data = torch.ones(x.shape[0], 1, device=dev)
            calls["multi"] = calls["single"] = 0
            out = q(x)
            # float1-4 all search float_value(1) (AQ:370-397): their installed grid is another one and is searched again
            expect_single = 1 if any(t in mode for t in ("float1", "float2", "float3", "float4")) and "float" in q.mode and q.mode != "float" else 0
            if tree == "olive" and bit >= 5 and om == "ovp":
                # 5-bit codebook + outliers: more than 128 buckets, no per-row table -> the single-read entry declines and the
                # quantiser searches type by type (2 types + the installed grid; for a per-tensor input the installed grid's
                # search is the one a moment ago on this very tensor: core.SearchMemo)
                assert calls == {"multi": 1, "single": 2 if is_input else 3}, (k, calls)
            else:
                assert calls["multi"] >= 1 and calls["single"] == expect_single, (k, calls, q.mode)
            same = _check_calibration(antq_lib, dev, q, x, out, k, sel, tr, lo, up, 1 if tree == "ant" else 2,
                                      tree == "olive" and om == "ovp", 2e-6 if (tree == "olive" and om == "ovp") else 0.0)
            if same is not None:
                n_cases += 1
            np.testing.assert_allclose(q.mse.item(), sel[k + "__mse"], rtol=5e-3, err_msg=k)
    finally:
        antq_lib.search_sse_multi, antq_lib.search_sse = real_m, real_s
    assert n_cases >= len(sel["keys"]) - 1
    capsys.readouterr()


def test_calibration_collectives_single_rank_group(antq_lib, dev, capsys):
    """The reference's DDP syncs inside _init_quant_para (broadcast(mse), all_reduce(alpha)/world, broadcast(grid),
    AQ:520-531) run when a process group exists: with a one-rank RCCL group they must leave the result unchanged."""
    import socket
    import torch
    import torch.distributed as dist
    from ant_quantization_amd.ant import quant_modules as qm
    sel = golden("ant_select.npz")
    x = to_dev(np.ascontiguousarray(sel["w_laplace__x"]), dev)

    def calibrate(mode):
        q = qm.TensorQuantizer(mode=mode, bit=4, is_signed=True, is_enable=True, args=_args()).to(dev)
        q.name = "golden"
        q.alpha.data = torch.ones(x.shape[0], 1, device=dev)
        return q, q(x)

    ref = {m: calibrate(m) for m in ("ant-int-pot-flint", "outlier")}
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        for m, (q0, y0) in ref.items():
            q1, y1 = calibrate(m)
            assert q1.mode == q0.mode and torch.equal(y1, y0)
            assert torch.equal(q1.alpha.data, q0.alpha.data) and torch.equal(q1.quant_grid, q0.quant_grid)
    finally:
        dist.destroy_process_group()
    capsys.readouterr()


def test_c_abi_from_a_torch_free_cpp_host(antq_lib):
    """tests/cabi/cabi_check.cpp: a plain HIP-runtime C++ program (no torch, no Python) drives the C ABI on its own
    stream and buffers -- nearest, fused fake-quant (ANT, per-tensor, OliVe pairs), dynamic alpha, abs-max, the batched
    launch, the 4-bit codec, error codes -- and compares every result bit for bit with the oracle library."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabi", "cabi_check")
    if not os.path.exists(exe):          # normally built by __graft_entry__.build() and shipped with the snapshot
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "ARCH=gfx950"], check=False, timeout=600)
    assert os.path.exists(exe), "tests/cabi/cabi_check not built (run __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "CABI CHECK OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 4))))
def test_random_shapes_fuzz(antq_lib, oracle, dev, seed):
    """Every reference codebook x random shapes (1-element rows, odd K, K = 147, rows that straddle every kernel's
    vector width, odd numel with outlier-victim pairs wrapping to element 0) x per-row / per-tensor x fp32 / bf16."""
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(9000 + seed)
    names = [k for k in G.files if not k.startswith("INVALID")]
    for _ in range(25):
        if rng.random() < 0.5:
            gname = names[rng.integers(0, len(names))]
            g, ovp = G[gname], False
            gmax = float(g.max())
        else:
            t, b, s = ["int", "flint"][rng.integers(0, 2)], int(rng.choice([3, 4, 4, 4, 5, 8])), "su"[rng.integers(0, 2)]
            gn = O["%s_b%d_%s" % (t, b, s)]
            g, gmax, ovp, gname = np.concatenate([gn, O["outlier_b%d_%s" % (b, s)]]), float(gn.max()), bool(rng.random() < 0.8), "olive_" + s
        rows = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 130]))
        K = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 27, 32, 64, 100, 147, 256, 257, 512, 1000, 1024, 2048, 4096, 4100, 8192]))
        rows = min(rows, max(1, 1_000_000 // K))
        x = make_x(rng, rows, K, unsigned=gname.endswith("_u"), specials=bool(rng.random() < 0.5)) * np.float32(rng.uniform(0.2, 40))
        if ovp:
            m = rng.random((rows, K)) < 0.03
            x[m] *= rng.uniform(8, 64, m.sum()).astype(np.float32)
        alpha = (safe_absmax(x).max(1) * rng.uniform(0.3, 1.2, rows) + 1e-6).astype(np.float32)
        per_row = bool(rng.random() < 0.7)
        with np.errstate(all="ignore"):
            run_case(antq_lib, oracle, dev, x, alpha if per_row else np.float32(alpha.mean()), g, gmax, per_row, ovp,
                     bool(rng.random() < 0.5))


def test_alpha_grad_kernel_vs_autograd_of_the_reference_graph(antq_lib, dev):
    """antq_alpha_grad (the fused backward w.r.t. alpha) against torch autograd on the reference's op sequence
    (AQ:535-551), per channel and per tensor, vector and ragged rows, fp32 / bf16 / fp16."""
    import torch
    from ant_quantization_amd import core
    from ant_quantization_amd.ant import quant_modules as qm
    g_np = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g_np)
    grid = torch.from_numpy(g_np).to(dev)
    torch.manual_seed(5)
    for shape, per_channel in (((64, 1024), True), ((8, 3, 3, 3), True), ((300, 16), True), ((5, 4100), True),
                               ((16, 64, 96), False), ((7, 33), False), ((2048, 4096), False)):
        x = (torch.randn(*shape, device=dev) * 0.05)
        if per_channel:
            alpha = (x.reshape(shape[0], -1).abs().amax(1) * 0.9).reshape(shape[0], *([1] * (len(shape) - 1)))
        else:
            alpha = x.abs().max() * 0.8
        go = torch.randn_like(x)
        # the reference graph in double precision is the yardstick for both
        xd, ad = x.double().requires_grad_(True), alpha.double().clone().requires_grad_(True)
        scale = ad / grid.double().max()
        d = xd / scale
        q = qm.QuantBase.forward((x / (alpha / grid.max())).detach(), grid).double()
        out_ref = ((q - d).detach() + d) * scale
        out_ref.backward(go.double())
        xi = x.detach().clone().requires_grad_(True)
        al = alpha.clone().requires_grad_(True)
        out = core.fake_quant(xi, al, plan, 10.0, per_channel)
        out.backward(go)
        assert torch.equal(xi.grad, go)                                  # d out / d x = 1, no clip mask
        torch.testing.assert_close(al.grad.double().reshape(-1), ad.grad.reshape(-1), rtol=2e-3, atol=2e-4)
        # the kernel alone against the same reduction done by torch in float64
        rows, row_len = core.view_rows(x, per_channel)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            xt, gt = x.to(dt), go.to(dt)
            outf = antq_lib.fakequant(xt, alpha.reshape(-1).float().contiguous(), plan, 10.0, rows, row_len, per_channel)
            gsum = antq_lib.alpha_grad(xt, outf, gt, rows, row_len, per_channel)
            term = (gt.float() * (outf.float() - xt.float())).double()
            ref = term.reshape(rows, row_len).sum(1) if per_channel else term.sum().reshape(1)
            mag = term.abs().reshape(rows, row_len).sum(1) if per_channel else term.abs().sum().reshape(1)
            # fp32 partial sums over the 4 / 8 elements of a lane's vector, float64 beyond that
            assert ((gsum - ref).abs() <= 3e-7 * mag + 1e-12).all(), (shape, dt)


def test_weight_bank_mixed_precision_layers_and_dtype_moves(antq_lib, dev):
    """WeightBank on a model with an 8-bit first layer (set_8_bit_layer_n), then moved to bf16: buffers are rebuilt,
    the two dtypes / grids ride in their own batches, results still equal the per-layer path."""
    import torch
    from ant_quantization_amd.weight_bank import WeightBank
    from ant_quantization_amd.ant import quant_model as aqm, quant_utils as aqu
    aqu.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4))
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Linear(96, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 10))
    model = aqm.quantize_model(net).to(dev).eval()
    aqu.enable_quantization(model)
    aqm.set_8_bit_layer_n(model, 1)                      # some layers' quantisers -> 8 bit (forces 'int', AQ:482)
    x = torch.randn(32, 96, device=dev)
    with torch.no_grad():
        model(x)
        y_ref = model(x)
    eight = [m for m in model.modules() if hasattr(m, "quant_weight") and int(m.quant_weight.bit) == 8]
    four = [m for m in model.modules() if hasattr(m, "quant_weight") and int(m.quant_weight.bit) == 4]
    assert eight and four and all(m.quant_weight.mode == "int" and m.quant_weight.quant_grid.numel() == 256 for m in eight)
    bank = WeightBank(model)
    with torch.no_grad():
        assert torch.equal(model(x), y_ref) and bank.launches == 1
    # a dtype move re-creates every parameter: stamps go stale, the bank rebuilds its descriptors once
    model.to(torch.bfloat16)
    xb = x.bfloat16()
    with torch.no_grad():
        yb = model(xb)
        assert bank.launches == 2 and torch.equal(model(xb), yb) and bank.launches == 2
    bank.detach()
    with torch.no_grad():
        assert torch.equal(model(xb), yb)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_steady_state_forward_is_sync_free_and_graph_capturable(antq_lib, dev, tree):
    """After calibration a quantised model's forward does no device->host read (the reference syncs several times
    per quantiser per forward, AQ:470 / :482): the whole forward -- activation quantisers, weight quantisers with and
    without a WeightBank -- captures into a hipGraph (capture would fail on any sync) and replays on new inputs."""
    import importlib
    import torch
    from ant_quantization_amd.weight_bank import WeightBank
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4))
    torch.manual_seed(12)
    net = torch.nn.Sequential(torch.nn.Linear(128, 512), torch.nn.GELU(), torch.nn.Linear(512, 512), torch.nn.GELU(),
                              torch.nn.Linear(512, 64))
    model = qmod.quantize_model(net).to(dev).eval()
    qutil.enable_quantization(model)
    static_x = torch.randn(64, 128, device=dev)
    with torch.no_grad():
        model(static_x)                                  # calibration (this one does read back)
    for use_bank in (False, True):
        bank = WeightBank(model) if use_bank else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                model(static_x)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_y = model(static_x)
        for seed in (1, 2):
            xn = torch.randn(64, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
            static_x.copy_(xn)
            graph.replay()
            torch.cuda.synchronize()
            with torch.no_grad():
                assert torch.equal(static_y, model(xn)), (tree, use_bank, seed)
        if bank is not None:
            bank.detach()


def test_qat_training_loop_updates_weights_and_alpha(antq_lib, dev):
    """ANT QAT (IMG/main.py:197-202): a few optimiser steps through the fused forward + fused alpha gradient -- the loss
    goes down, weights and the per-channel alphas move, and evaluation afterwards uses the updated parameters."""
    import torch
    from ant_quantization_amd.ant import quant_model as aqm, quant_utils as aqu
    aqu.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4))
    torch.manual_seed(21)
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 16))
    teacher = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 16)).to(dev)
    model = aqm.quantize_model(net).to(dev)
    aqu.enable_quantization(model)
    x = torch.randn(512, 64, device=dev)
    with torch.no_grad():
        y = teacher(x)
        model(x)                                                       # calibrate
    w0 = model[0].weight.detach().clone()
    a0 = model[0].quant_weight.alpha.detach().clone()
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    losses = []
    model.train()
    for _ in range(40):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert not torch.equal(model[0].weight, w0) and not torch.equal(model[0].quant_weight.alpha, a0)
    assert torch.isfinite(model[0].quant_weight.alpha).all() and model[0].quant_weight.alpha.grad is not None
    model.eval()
    with torch.no_grad():
        assert float(torch.nn.functional.mse_loss(model(x), y)) < 0.6 * losses[0]


def test_quantizer_follows_external_grid_edits(antq_lib, oracle, dev, capsys):
    """The plan follows the `quant_grid` buffer: an in-place edit after calibration (what the DDP broadcast of AQ:531
    does on the non-zero ranks, or a user patching a codebook) is picked up on the next forward."""
    import torch
    from ant_quantization_amd.ant import quant_modules as qm
    g = golden("ant_grids.npz")
    q = qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=True, args=_args()).to(dev)
    x = torch.randn(16, 256, device=dev) * 0.03
    q.alpha.data = torch.ones(16, 1, device=dev)
    q(x)
    assert f32_same(q.quant_grid.cpu().numpy(), g["flint_b4_s"])
    q.quant_grid.copy_(torch.from_numpy(g["int_b4_s"]).to(dev))          # in place, same buffer
    out = q(x)
    ref, _ = oracle.forward(x.cpu().numpy(), q.alpha.detach().cpu().numpy().reshape(-1), g["int_b4_s"])
    assert f32_same(out.detach().cpu().numpy(), ref)
    capsys.readouterr()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_dynamic_batched_launch_equals_per_tensor_dynamic(antq_lib, dev, dtype_name):
    """ANTQ_FLAG_DYNAMIC: many tensors, alpha = group / row abs-max computed in the kernel, ONE batch -- same alphas and
    the same bits as antq_fakequant_dynamic per tensor (itself oracle-checked), ANT and OliVe pairs, from 1-vector groups
    to 8192-vector rows; what cannot live in registers is refused."""
    import torch
    dtype = getattr(torch, dtype_name)
    epl = 4 if dtype == torch.float32 else 8
    g = golden("ant_grids.npz")["flint_b4_s"]
    O = golden("olive_grids.npz")
    pol = antq_lib.plan_for(np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]))
    plan = antq_lib.plan_for(g)
    torch.manual_seed(14)
    shapes = [(64, 128 * epl), (33, 256 * epl), (17, 384 * epl), (9, 1024 * epl), (40, 200 * epl), (5, 777 * epl),
              (6, 512 * epl), (3, 2048 * epl), (4, 1500 * epl),
              # groups of 1 .. 64 vectors (group-16 bf16 = 2 lanes per group): butterfly over the group's lanes
              (4096, epl), (2048, 2 * epl), (700, 4 * epl), (512, 16 * epl), (129, 32 * epl), (65, 64 * epl),
              # rows of 2049 .. 8192 vectors (C4's 28 672-wide rows): one row per 1024-thread workgroup
              (3, 2049 * epl), (2, 3584 * epl), (3, 4096 * epl), (2, 7168 * epl), (1, 8192 * epl)]
    for p, gmax, ovp in ((plan, 10.0, False), (pol, 32.0, True)):
        xs = [(torch.randn(*s, device=dev) * 0.03).to(dtype) for s in shapes]
        for x in xs:
            x.view(-1)[::97] *= 25
        refs = [antq_lib.fakequant_dynamic(x, p, gmax, x.shape[0], x.shape[1], ovp=ovp) for x in xs]
        outs = [torch.zeros_like(x) for x in xs]
        alphas = [torch.zeros(x.shape[0], dtype=torch.float32, device=dev) for x in xs]
        bt = antq_lib.Batch([(x, o, a, p, gmax, x.shape[0], x.shape[1], True) for x, o, a in zip(xs, outs, alphas)],
                            ovp=ovp, dynamic=True)
        bt.run()
        for (ro, ra, _), o, a, s in zip(refs, outs, alphas, shapes):
            assert torch.equal(a, ra) and torch.equal(o, ro), (s, ovp)
        # nobody wants the scales: no alpha buffers (the kernels skip the store), same values; per tensor and batched
        outs2 = [torch.zeros_like(x) for x in xs]
        antq_lib.Batch([(x, o, None, p, gmax, x.shape[0], x.shape[1], True) for x, o in zip(xs, outs2)], ovp=ovp, dynamic=True).run()
        for (ro, _, _), o, x, s in zip(refs, outs2, xs, shapes):
            assert torch.equal(o, ro), (s, ovp)
            o3, a3, _ = antq_lib.fakequant_dynamic(x, p, gmax, x.shape[0], x.shape[1], ovp=ovp, want_alpha=False)
            assert a3 is None and torch.equal(o3, ro), (s, ovp)
    for bad in [(8, 72 * epl), (8, 8193 * epl)]:      # a small group that is no power of two / too long for the registers
        x = torch.randn(*bad, device=dev).to(dtype)
        with pytest.raises(antq_lib.AntqError):
            antq_lib.Batch([(x, torch.empty_like(x), torch.zeros(8, device=dev), plan, 10.0, bad[0], bad[1], True)], dynamic=True)


def test_affine_vector_kernel_equals_element_kernel(antq_lib, oracle, dev):
    """antq_affine picks a 16-byte vector kernel (exact 5-FMA division) for aligned inputs and an element kernel (IEEE
    division) otherwise: same bits on the same data, across benign and hostile (min, max) ranges, and both equal the
    oracle's restatement of quant_affine.py:95-115."""
    import torch
    torch.manual_seed(17)
    rows, K = 96, 1024
    base = torch.randn(rows * K + 4, device=dev)
    base[::311] *= 50
    for k in (4, 8):
        for scale_x, shift in ((1.0, 0.0), (1e-6, 0.0), (1e4, 3e4), (1e-9, 1.0), (3e18, 0.0)):
            buf = base * scale_x + shift
            xa = buf[:rows * K].view(rows, K)                       # 16-byte aligned -> vector kernel
            xu = buf[1:rows * K + 1]                                # 4-byte offset   -> element kernel
            xu.copy_(xa.reshape(-1).clone())
            xa = xu.clone().view(rows, K)
            for per_row in (False, True):
                if per_row:
                    mn, mx = xa.min(1).values.contiguous(), xa.max(1).values.contiguous()
                else:
                    mn, mx = xa.min().reshape(1), xa.max().reshape(1)
                ov, qv = antq_lib.affine(xa, k, mn, mx, rows, K, per_row, want_q=True)
                oe = antq_lib.affine(xu, k, mn, mx, rows, K, per_row)
                assert torch.equal(ov.view(-1).view(torch.int32), oe.view(torch.int32)), (k, scale_x, shift, per_row)
                ref = oracle.affine(xa.cpu().numpy(), k, mn.cpu().numpy(), mx.cpu().numpy())
                ref_out = ref[0] if isinstance(ref, tuple) else ref
                assert f32_same(ov.cpu().numpy(), ref_out), (k, scale_x, shift, per_row)


def test_nearest_plan_equals_scan_and_cache_follows_grid_edits(antq_lib, oracle, dev):
    """antq_nearest_plan (table lookup on d = x) == the reference scan, fp32 / bf16 / fp16, aligned and ragged; and
    quant_cuda.quant's per-buffer plan cache notices an in-place edit of the grid."""
    import torch
    from ant_quantization_amd import quant_cuda
    rng = np.random.default_rng(31)
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    grids_ = [G["flint_b4_s"], G["int_b8_u"], G["pot_b6_u"], G["apot_b4_s"], np.float32([3, 1, 2, 1, 3, -7]),
              np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])]
    for g in grids_:
        plan = antq_lib.plan_for(g)
        hi = float(np.abs(g).max()) + 1
        for n in (8192, 8191, 33):
            x = np.concatenate([rng.standard_normal(n - 12).astype(np.float32) * np.float32(hi / 2),
                                np.float32([0.0, -0.0, 1e5, -1e5, 102400.0, 2e5, np.inf, -np.inf, np.nan, 36000.0, 1e-30, 65535.0])])
            with np.errstate(all="ignore"):
                zr, jr = oracle.nearest(x, g)
            z, j = antq_lib.nearest_plan(to_dev(x, dev), plan, want_idx=True)
            assert f32_same(z.cpu().numpy(), zr) and np.array_equal(j.cpu().numpy().astype(np.int32), jr), (g[:4], n)
            xb = oracle.f32_to_bf16(x)
            with np.errstate(all="ignore"):
                zb_ref, _ = oracle.nearest(oracle.bf16_to_f32(xb), g)
            zb = antq_lib.nearest_plan(to_dev(xb, dev, True), plan)
            assert bf16_same(bf16_bits(zb), oracle.f32_to_bf16(zb_ref), oracle), (g[:4], n, "bf16")
            xh = torch.from_numpy(x).to(dev).half()
            with np.errstate(all="ignore"):
                zh_ref, _ = oracle.nearest(xh.float().cpu().numpy(), g)
            zh = antq_lib.nearest_plan(xh, plan)
            assert f32_same(zh.float().cpu().numpy(), torch.from_numpy(zh_ref).half().float().numpy()), (g[:4], n, "f16")
    # the drop-in: same buffer, new contents
    gt = torch.from_numpy(G["flint_b4_s"].copy()).to(dev)
    x = torch.randn(4096, device=dev) * 4
    for _ in range(3):          # (third call: through the plan learnt at the second sighting)
        z1, zero = quant_cuda.quant(x, gt)
        assert not zero.any() and f32_same(z1.cpu().numpy(), oracle.nearest(x.cpu().numpy(), G["flint_b4_s"])[0])
    gt.copy_(torch.from_numpy(G["int_b4_s"]).to(dev))
    z2, _ = quant_cuda.quant(x, gt)
    assert f32_same(z2.cpu().numpy(), oracle.nearest(x.cpu().numpy(), G["int_b4_s"])[0])


def test_dropin_operator_never_trusts_a_buffer_identity(antq_lib, oracle, dev):
    """quant_cuda.quant remembers a plan per grid ADDRESS only as a hint that the kernel verifies against the device
    array (antq_nearest_hinted).  What the reference's own calibration does -- `quant_grid.data = int_value()`, then
    flint, pot, the winner again: same object, same `_version`, same numel, addresses recycled by the caching
    allocator -- and in-place edits through `.data` (no version bump either) must always quantise on the values the
    buffer holds NOW."""
    import torch
    from ant_quantization_amd import quant_cuda
    G = golden("ant_grids.npz")
    names = ["int_b4_s", "flint_b4_s", "pot_b4_s", "float_b4_s"]
    x = torch.randn(1 << 14, device=dev) * 4
    xn = x.cpu().numpy()
    refs = {k: oracle.nearest(xn, G[k])[0] for k in names}
    buf = torch.nn.Module()
    buf.register_buffer("quant_grid", torch.ones(16, device=dev))
    v0 = buf.quant_grid._version
    seen_hint = 0
    for it in range(40):
        k = names[(it * 7 + it // 3) % 4]
        buf.quant_grid.data = torch.from_numpy(G[k]).to(dev)          # rebinding: same object, same version, old block freed
        for _ in range(1 + it % 3):
            z, _ = quant_cuda.quant(x, buf.quant_grid)
            assert f32_same(z.cpu().numpy(), refs[k]), (it, k)
        h = quant_cuda._hints.get((buf.quant_grid.data_ptr(), 16, dev.index))
        seen_hint += int(h is not None and h.plan is not None)
    assert buf.quant_grid._version == v0
    assert seen_hint > 0                                               # the table path did engage along the way
    # in-place through .data: same address, same version, other values -- with a plan already believed for the address
    quant_cuda._hints.clear()         # (this address may be a recycled one that has earned a long probation above)
    g = torch.from_numpy(G["flint_b4_s"].copy()).to(dev)
    for _ in range(3):
        quant_cuda.quant(x, g)
    h = quant_cuda._hints[(g.data_ptr(), 16, dev.index)]
    assert h.plan is not None
    v = g._version
    g.data.copy_(torch.from_numpy(G["pot_b4_s"]).to(dev))
    assert g._version == v
    z, _ = quant_cuda.quant(x, g)
    assert f32_same(z.cpu().numpy(), refs["pot_b4_s"])                 # stale belief: the kernel scanned the device values
    torch.cuda.synchronize()
    assert int(h.stale[0]) == 1                                        # ... and told the host, without a sync on the path
    for _ in range(6):
        z, _ = quant_cuda.quant(x, g)
        assert f32_same(z.cpu().numpy(), refs["pot_b4_s"])
    assert h.plan is not None and int(h.stale[0]) == 0 and np.array_equal(h.plan.grid, G["pot_b4_s"])   # relearnt
    # OliVe's caller passes a fresh torch.cat((quant_grid, outliers)) temporary on every call (OQ:303-306)
    O = golden("olive_grids.npz")
    gn, go = to_dev(O["flint_b4_s"], dev), to_dev(O["outlier_b4_s"], dev)
    ref = oracle.nearest(xn * 20, np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]))[0]
    for _ in range(8):
        z, _ = quant_cuda.quant(x * 20, torch.cat((gn, go)))
        assert f32_same(z.cpu().numpy(), ref)
    # ANT's `outlier` mode goes through the quantiser's OWN plan as the hint
    from ant_quantization_amd.ant import quant_modules as qm
    q = qm.TensorQuantizer(mode="outlier", bit=4, is_signed=True, is_enable=True, args=_args(percent=99)).to(dev)
    q.name = "t"
    w = torch.randn(64, 256, device=dev)
    o1 = q(w)
    q.quant_grid.data = torch.from_numpy(G["flint_b4_s"]).to(dev)        # rebinding behind the quantiser's back (int -> flint)
    q._grid_key = q._grid_now()                                        # worst case: the host watch is fooled too
    o2 = q(w)
    scale = q.percent_value_int4 / torch.max(q.quant_grid)
    body = torch.from_numpy(oracle.nearest((w / scale).cpu().numpy().reshape(-1), q.quant_grid.cpu().numpy())[0]).to(dev).view_as(w) * scale
    inl = w.abs() <= q.percent_value_int4
    assert torch.equal(o2[inl], body[inl]) and not torch.equal(o1, o2)


def test_quant_cuda_importable_the_reference_way(antq_lib, oracle, dev, tmp_path):
    """The reference does `import quant_cuda` (AQ/quant_modules.py:7) and calls
    `quant_cuda.quant(x.view(-1), grid.type_as(x))` (AQ:12-18): from a clean interpreter whose sys.path holds only the
    drop-in directory (or the package directory itself), fp32 and fp64, against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "import quant_cuda\n"
        "g = np.load(sys.argv[2])['flint_b4_s']\n"
        "x = np.load(sys.argv[3])\n"
        "for dt in (torch.float32, torch.float64):\n"
        "    xt = torch.from_numpy(x).to('cuda:0').to(dt)\n"
        "    grid = torch.from_numpy(g).to('cuda:0')\n"
        "    for _ in range(3):\n"
        "        z, idx = quant_cuda.quant(xt.view(-1), grid.type_as(xt))\n"
        "    assert z.dtype == dt and z.shape == xt.view(-1).shape and idx.shape == z.shape and not idx.any()\n"
        "    np.save(sys.argv[4] + str(dt)[-2:] + '.npy', z.cpu().numpy())\n")
    x = np.random.default_rng(5).standard_normal((64, 100)).astype(np.float32) * 4
    np.save(tmp_path / "x.npy", x)
    zr, _ = oracle.nearest(x.reshape(-1), golden("ant_grids.npz")["flint_b4_s"])
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for d in (os.path.join(root, "ant_quantization_amd", "dropin"), os.path.join(root, "ant_quantization_amd")):
        out = str(tmp_path / ("z_%s_" % os.path.basename(d)))
        subprocess.check_call([sys.executable, "-c", code, d, os.path.join(root, "tests", "golden", "ant_grids.npz"),
                               str(tmp_path / "x.npy"), out], cwd=str(tmp_path), env=env)
        assert f32_same(np.load(out + "32.npy"), zr)
        assert np.array_equal(np.load(out + "64.npy"), zr.astype(np.float64))


@pytest.mark.parametrize("bf16", [False, True])
def test_row_sharded_odd_numel_ovp_wrap(antq_lib, oracle, dev, bf16):
    """A row-sharded OliVe tensor with an ODD element count: the last element's partner is global element 0
    (torch.roll wrap, OQ:313-318), which another rank owns.  sharding.wrap_flag / apply_wrap_flag (what
    fix_odd_numel_wrap broadcasts and applies) through the real kernels, all ranks emulated on this GPU: the
    concatenated blocks must equal the unsharded oracle result bit for bit."""
    import torch
    from ant_quantization_amd import sharding
    O = golden("olive_grids.npz")
    gn = O["flint_b4_s"]
    grid = np.concatenate([gn, O["outlier_b4_s"]])
    plan = antq_lib.plan_for(grid)
    rng = np.random.default_rng(17)
    rows, K = 9, 33
    for world in (2, 3):
        for first, blk_first in ((0.9, 0.0), (0.001, 0.9), (0.9, 0.9), (0.001, 0.001), (-0.9, 0.9)):
            x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
            alpha = np.full(rows, 0.06, np.float32) + rng.random(rows).astype(np.float32) * 0.01
            blocks = [sharding.row_block(rows, r, world, pair_safe_row_len=K) for r in range(world)]
            x[0, 0], x[-1, -1], x[blocks[-1][0], 0] = first, 0.01, blk_first
            if bf16:
                xs = oracle.f32_to_bf16(x)
                full, _ = oracle.forward(xs, alpha, grid, 32.0, True)
            else:
                xs = x
                full, _ = oracle.forward(x, alpha, grid, 32.0, True)
            outs, xts, ats = [], [], []
            for b, e in blocks:
                xt = to_dev(np.ascontiguousarray(xs[b:e]), dev, bf16)
                at = to_dev(alpha[b:e].copy(), dev)
                outs.append(antq_lib.fakequant(xt, at, plan, 32.0, e - b, K, True, ovp=True))
                xts.append(xt)
                ats.append(at)
            flag = sharding.wrap_flag(outs[0], ats[0], plan, 32.0)
            assert int(flag.item()) == int(abs(first) > 0.5)
            sharding.apply_wrap_flag(xts[-1], outs[-1], ats[-1], plan, 32.0, flag)
            got = torch.cat([o.reshape(-1) for o in outs])
            if bf16:
                assert bf16_same(bf16_bits(got), full.reshape(-1), oracle), (world, first, blk_first)
            else:
                assert f32_same(got.cpu().numpy(), full), (world, first, blk_first)


def test_bench_workload_batched_kernel_vs_oracle(antq_lib, oracle, dev):
    """The launch bench.py times, checked DIRECTLY: the same batch (32 x [4096, 4096] bf16, randn * 0.02 from the device
    generator seeded 6, signed flint-4, alpha = row abs-max, one antq_fakequant_batch launch = k_fq_batch<bf16,false>),
    72 rows of every tensor against the oracle (values), the same rows' indices through the per-tensor kernel (the lane
    kernel with the exact per-element decision), and every element of the batched output against the per-tensor launch;
    then the per-tensor launches through the per-row table kernel too (knob 5 = 0): identical bits."""
    import torch
    from ant_quantization_amd import grids
    g = grids.ant_flint(4, True)
    assert np.array_equal(g, golden("ant_grids.npz")["flint_b4_s"])
    plan = antq_lib.plan_for(g)
    R = K = 4096
    nbuf = 32
    gen = torch.Generator(device=dev)
    gen.manual_seed(6)
    xs, alphas, outs = [], [], []
    for _ in range(nbuf):
        x = (torch.randn(R, K, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
        xs.append(x)
        alphas.append(antq_lib.absmax(x, R, K, per_row=True))
        outs.append(torch.empty_like(x))
    batch = antq_lib.Batch([(xs[i], outs[i], alphas[i], plan, 10.0, R, K, True) for i in range(nbuf)])
    assert not batch.singles
    batch.run()
    rng = np.random.default_rng(6)
    n_rows = 0
    for i in range(nbuf):
        rows = np.unique(np.concatenate([[0, 1, R - 2, R - 1], rng.integers(0, R, 72)]))
        rt = torch.from_numpy(rows).to(dev)
        xa = bf16_bits(xs[i][rt])
        a = alphas[i][rt].cpu().numpy()
        assert np.array_equal(a, oracle.absmax(oracle.bf16_to_f32(xa), True, 1.0))          # the calibrated alpha itself
        ref, ridx = oracle.forward(xa, a, g, 10.0, False)
        assert bf16_same(bf16_bits(outs[i][rt]), ref, oracle), i
        o1, idx = antq_lib.fakequant(xs[i], alphas[i], plan, 10.0, R, K, True, want_idx=True)
        assert np.array_equal(idx[rt].cpu().numpy().astype(np.int32), ridx), i
        assert torch.equal(o1.view(torch.int16), outs[i].view(torch.int16)), i               # batch == per tensor, everywhere
        n_rows += rows.size
    assert n_rows >= 64 * nbuf
    antq_lib.lib().antq_debug_set(5, 0)
    try:
        for i in range(0, nbuf, 8):
            o2 = antq_lib.fakequant(xs[i], alphas[i], plan, 10.0, R, K, True)
            assert torch.equal(o2.view(torch.int16), outs[i].view(torch.int16)), i
    finally:
        antq_lib.lib().antq_debug_set(5, 1)


def _resnet50_weight_shapes():
    s = [(64, 3, 7, 7)]
    inp = 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            s += [(planes, inp, 1, 1), (planes, planes, 3, 3), (planes * 4, planes, 1, 1)]
            if b == 0:
                s.append((planes * 4, inp, 1, 1))
            inp = planes * 4
    s.append((1000, 2048))
    return s


@pytest.mark.parametrize("bf16", [False, True])
def test_c1_resnet50_all_weight_tensors_sampled_against_the_oracle(antq_lib, oracle, dev, bf16):
    """configs[1] at its real size against the ORACLE (not against another HIP launch): all 54 ResNet-50 weight tensors
    (randn * sqrt(2 / fan_out), seed 1), signed flint-4, alpha = abs-max, (a) per output channel -- the
    reference's grouping -- and (b) groups of 16 on the flattened tensor, each as ONE batched launch; >= 64 rows / groups
    of every tensor (all of them when it has fewer) are compared with oracle.forward bit for bit, the rest of the batched
    output with the per-tensor launches."""
    import torch
    from ant_quantization_amd import grids
    g = grids.ant_flint(4, True)
    plan = antq_lib.plan_for(g)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    ws = []
    for sh in _resnet50_weight_shapes():
        fan_out = sh[0] * int(np.prod(sh[2:], dtype=np.int64))
        w = torch.randn(*sh, device=dev, generator=gen) * float(np.sqrt(2.0 / fan_out))
        ws.append(w.to(torch.bfloat16) if bf16 else w)
    assert len(ws) == 54
    rng = np.random.default_rng(101)
    for group in (None, 16):
        views = [w.reshape(w.shape[0], -1) if group is None else w.reshape(-1, group) for w in ws]
        alphas = [antq_lib.absmax(v, v.shape[0], v.shape[1], per_row=True) for v in views]
        outs = [torch.empty_like(v) for v in views]
        bt = antq_lib.Batch([(v, o, a, plan, 10.0, v.shape[0], v.shape[1], True) for v, o, a in zip(views, outs, alphas)])
        bt.run()
        n_checked = 0
        for v, o, a in zip(views, outs, alphas):
            R = v.shape[0]
            rows = np.arange(R) if R <= 256 else np.unique(np.concatenate([[0, 1, R - 2, R - 1], rng.choice(R, 96, replace=False)]))
            rt = torch.from_numpy(rows).to(dev)
            xa = bf16_bits(v[rt]) if bf16 else v[rt].cpu().numpy()
            an = a[rt].cpu().numpy()
            xf = oracle.bf16_to_f32(xa) if bf16 else xa
            assert np.array_equal(an, oracle.absmax(xf, True, 1.0))
            ref, _ = oracle.forward(xa, an, g, 10.0, False)
            got = o[rt]
            ok = bf16_same(bf16_bits(got), ref, oracle) if bf16 else f32_same(got.cpu().numpy(), ref)
            assert ok, (tuple(v.shape), group)
            assert rows.size >= min(R, 64)
            n_checked += rows.size
            o1 = antq_lib.fakequant(v, a, plan, 10.0, v.shape[0], v.shape[1], True)
            assert torch.equal(o1.view(torch.int16) if bf16 else o1.view(torch.int32), o.view(torch.int16) if bf16 else o.view(torch.int32))
        assert n_checked >= 54 * 64


@pytest.mark.parametrize("shape", [(16384, 4096), (4096, 16384), (4096, 4096)])
def test_c3_opt67b_weight_shapes_static_ovp_sampled_against_the_oracle(antq_lib, oracle, dev, shape):
    """configs[3] at its real sizes against the ORACLE: the three OPT-6.7B weight shapes (randn * 0.02 with
    0.1 % of the entries multiplied by U(8, 64), seed 4), OliVe flint-4 + outlier codebook with the outlier-victim pair
    rule, STATIC alpha = the 3-sigma statistic (antq_xmax_3sigma, compared with the oracle's restatement), bf16 and fp32:
    batched launch, ordinary and unordered per-tensor launches; 96 rows of each against oracle.forward (values and -- per
    tensor -- indices incl. victims), everything else against the per-tensor launch."""
    import torch
    O = golden("olive_grids.npz")
    gn, go = O["flint_b4_s"], O["outlier_b4_s"]
    gg, gmax = np.concatenate([gn, go]), float(gn.max())
    plan = antq_lib.plan_for(gg)
    R, K = shape
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    w32 = torch.randn(R, K, device=dev, generator=gen) * 0.02
    m = torch.rand(R, K, device=dev, generator=gen) < 0.001
    w32[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 64, generator=gen)
    rng = np.random.default_rng(104)
    for bf16 in (True, False):
        w = w32.to(torch.bfloat16) if bf16 else w32
        alpha = antq_lib.xmax_3sigma(w, R, K, per_row=True)
        rows = np.unique(np.concatenate([[0, 1, R - 2, R - 1], rng.choice(R, 96, replace=False)]))
        rt = torch.from_numpy(rows).to(dev)
        xa = bf16_bits(w[rt]) if bf16 else w[rt].cpu().numpy()
        an = alpha[rt].cpu().numpy()
        np.testing.assert_allclose(an, oracle.three_sigma(xa, True), rtol=2.0 ** -7 if bf16 else 2e-6)
        ref, ridx = oracle.forward(xa, an, gg, gmax, True)
        assert (ridx == oracle.IDX_VICTIM).any() and (ridx >= gn.size).any()
        out_b = torch.empty_like(w)
        antq_lib.Batch([(w, out_b, alpha, plan, gmax, R, K, True)], ovp=True).run()
        o1, idx = antq_lib.fakequant(w, alpha, plan, gmax, R, K, True, ovp=True, want_idx=True)
        o2 = antq_lib.fakequant(w, alpha, plan, gmax, R, K, True, ovp=True)
        o3 = antq_lib.fakequant(w, alpha, plan, gmax, R, K, True, ovp=True, unordered=True, out=torch.empty_like(w))
        with pytest.raises(antq_lib.AntqError):           # an unordered launch into a buffer the allocator may just have recycled
            antq_lib.fakequant(w, alpha, plan, gmax, R, K, True, ovp=True, unordered=True)
        for o in (out_b, o1, o2, o3):
            ok = bf16_same(bf16_bits(o[rt]), ref, oracle) if bf16 else f32_same(o[rt].cpu().numpy(), ref)
            assert ok, (shape, bf16)
        assert np.array_equal(idx[rt].cpu().numpy().astype(np.int32), ridx)
        iv = (lambda t: t.view(torch.int16)) if bf16 else (lambda t: t.view(torch.int32))
        assert torch.equal(iv(out_b), iv(o2)) and torch.equal(iv(o1), iv(o2)) and torch.equal(iv(o3), iv(o2))


def test_bert_base_real_shapes_weights_and_activations(antq_lib, oracle, dev, capsys):
    """C2 at its real sizes: the 74 nn.Linear weights of BERT-base (49 x [768,768] incl. the pooler,
    12 x [3072,768], 12 x [768,3072], the [2,768] classifier; randn * 0.02, seed 2) calibrated with `ant-int-pot-flint`
    4-bit (w_low 80, w_up 150: ABERT/scripts/cola_ptq.sh:82-89) and the two activation shapes [64,128,768] /
    [64,128,3072] (gelu(randn), seed 3) per tensor through the atomic `PT` search path.  Oracle checks: the clip
    search of sampled rows (scores within reduction noise, the pick the oracle's own or a tie by the ORACLE's scores),
    the steady-state forward of sampled rows / the whole activation bit for bit, and the batched launch over all 74
    weights against the per-layer outputs."""
    import torch
    from ant_quantization_amd.ant import quant_modules as qm
    from calib_check import NEAR_TIE_RTOL
    shapes = [(768, 768)] * 49 + [(3072, 768)] * 12 + [(768, 3072)] * 12 + [(2, 768)]
    gen = torch.Generator(device=dev).manual_seed(2)
    args = _args(w_low=80, a_low=80, w_up=150, a_up=150)
    rng = np.random.default_rng(2)
    jobs, per_layer = [], []
    for li, (r, k) in enumerate(shapes):
        w = torch.randn(r, k, device=dev, generator=gen) * 0.02
        q = qm.TensorQuantizer(mode="ant-int-pot-flint", bit=4, is_signed=True, is_enable=True, args=args).to(dev)
        q.name = "L%d" % li
        q.alpha.data = torch.ones(r, 1, device=dev)
        out = q(w).detach()
        grid = q.quant_grid.cpu().numpy()
        alpha = q.alpha.detach().reshape(-1)
        if li % 6 == 0 or r == 2:             # 14 layers get the oracle's calibration on sampled rows
            rows = np.unique(rng.integers(0, r, 3))
            wn = w[torch.from_numpy(rows).to(dev)].cpu().numpy()
            xmax = np.abs(wn).max(1).astype(np.float32)
            best, oalpha, trace = oracle.search_mse(wn, xmax, 80, 150, 1, grid, 10.0, False, True)
            got = alpha[torch.from_numpy(rows).to(dev)].cpu().numpy()
            for j in range(rows.size):
                if got[j] != oalpha[j]:
                    c = int(np.argmin(np.abs(got[j] / xmax[j] - np.float32(np.arange(80, 150) * 0.01))))
                    assert (trace[c, j] - best[j]) <= NEAR_TIE_RTOL * best[j], (li, rows[j], got[j], oalpha[j])
        rows = np.unique(np.concatenate([[0, r - 1], rng.integers(0, r, 24)]))
        rt = torch.from_numpy(rows).to(dev)
        ref, _ = oracle.forward(w[rt].cpu().numpy(), alpha[rt].cpu().numpy(), grid, 10.0, False)
        assert f32_same(out[rt].cpu().numpy(), ref), li
        ob = torch.empty_like(w)
        jobs.append((w, ob, alpha.contiguous(), antq_lib.plan_for(grid), 10.0, r, k, True))
        per_layer.append(out)
    b = antq_lib.Batch(jobs)
    assert not b.singles
    b.run()
    for j, out in zip(jobs, per_layer):
        assert torch.equal(j[1], out)
    # activations, per tensor
    gen = torch.Generator(device=dev).manual_seed(3)
    for shape in ((64, 128, 768), (64, 128, 3072)):
        x = torch.nn.functional.gelu(torch.randn(*shape, device=dev, generator=gen))
        q = qm.TensorQuantizer(mode="ant-int-pot-flint", bit=4, is_signed=False, is_enable=True, is_input=True, args=args).to(dev)
        q.name = "act"
        out = q(x).detach()
        assert q.is_signed                                                  # gelu output has negatives: AQ:71-73
        grid = q.quant_grid.cpu().numpy()
        a = q.alpha.detach().reshape(1).cpu().numpy()
        xn = x.cpu().numpy().reshape(1, -1)
        ref, _ = oracle.forward(xn, a, grid, 10.0, False)
        assert f32_same(out.cpu().numpy(), ref), shape
        # the per-tensor search (LDS accumulators + one atomic per candidate and workgroup): three candidates' scores
        # against the oracle's forward + mse on the whole tensor, and the pick against its neighbours
        xmax = antq_lib.absmax(x.contiguous(), 1, x.numel(), per_row=False)
        from ant_quantization_amd import core
        ratios = core._ratios(80, 150, 1, dev)
        sse = antq_lib.search_sse(x.contiguous(), 1, x.numel(), xmax, False, ratios, antq_lib.plan_for(grid), 10.0)
        mse = (sse[:, 0] / x.numel()).float().cpu().numpy()
        c_pick = int(np.argmin(mse))
        assert np.float32(xmax.item() * np.float32((80 + c_pick) * 0.01)) == a[0]
        for c in sorted({0, c_pick, max(c_pick - 1, 0), min(c_pick + 1, 69), 69}):
            ac = np.float32(xmax.item()) * np.float32((80 + c) * 0.01)
            oq, _ = oracle.forward(xn, np.float32([ac]), grid, 10.0, False)
            ref_mse = oracle.mse(oq, xn, per_row=False)
            np.testing.assert_allclose(mse[c], np.asarray(ref_mse).reshape(-1)[0], rtol=2e-5, err_msg=str((shape, c)))
    capsys.readouterr()


def test_type_selection_on_one_read_equals_per_type_searches(antq_lib, oracle, dev, capsys):
    """antq_search_sse_multi (every candidate type of a type selection on ONE read of the tensor) against one
    antq_search_sse per type: the same sums (rows held by one wavefront: identical bits; split rows / per-tensor sums:
    fp64 atomics in another order), hence the same picks; and the quantiser's calibration issues exactly one search
    launch for an `ant-` list where it used to issue one per type plus one for the installed grid."""
    import torch
    from ant_quantization_amd import core, grids
    G = golden("ant_grids.npz")
    O = golden("olive_grids.npz")
    ant = [G["int_b4_s"], G["flint_b4_s"], G["pot_b4_s"], G["float_b4_s"]]
    ol = [np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])]
    torch.manual_seed(23)
    cases = [(torch.randn(64, 768, device=dev) * 0.02, True, ant, [10.0] * 4, False, 1),
             (torch.randn(48, 3072, device=dev) * 0.02, True, ant[:3], [10.0] * 3, False, 1),
             (torch.nn.functional.gelu(torch.randn(16, 64, 768, device=dev)), False, ant[:3], [10.0] * 3, False, 1),
             ((torch.randn(128, 4096, device=dev) * 0.02).bfloat16(), True, ol, [28.0, 32.0], True, 2),
             (torch.randn(9, 130000, device=dev).half(), False, ol, [28.0, 32.0], True, 2)]
    for x, per_row, gl, gmaxs, ovp, step in cases:
        x = x.contiguous()
        if ovp:
            x.view(-1)[::211] *= 20
        plans = [antq_lib.plan_for(g) for g in gl]
        rows, row_len = core.view_rows(x, per_row)
        xmax = core.row_absmax(x, per_row)
        ratios = core._ratios(75, 150, step, dev)
        multi = antq_lib.search_sse_multi(x, rows, row_len, xmax, per_row, ratios, plans, gmaxs, ovp=ovp)
        assert multi is not None and multi.shape[0] == len(gl)
        res = core.clip_search_types(x, xmax, per_row, 75, 150, step, plans, gmaxs, ovp=ovp)
        for t, (p, gm) in enumerate(zip(plans, gmaxs)):
            one = antq_lib.search_sse(x, rows, row_len, xmax, per_row, ratios, p, gm, ovp=ovp)
            exact = per_row          # a wavefront walks its row's tasks in order in both kernels: identical sums
            if exact:
                assert torch.equal(multi[t], one), t
            else:
                torch.testing.assert_close(multi[t], one, rtol=1e-12, atol=0)
            b1, a1, _ = core.clip_search(x, xmax, per_row, 75, 150, step, p, gm, ovp=ovp)
            flips = (res[t][1] != a1)
            assert flips.float().mean() <= (0.0 if exact else 0.02), t
    # a shape without the single-read path falls back (None), the quantiser then searches type by type
    small = torch.randn(64, 64, device=dev)
    assert antq_lib.search_sse_multi(small, 64, 64, core.row_absmax(small, True), True, core._ratios(75, 150, 1, dev),
                                     [antq_lib.plan_for(g) for g in ant[:2]], [10.0, 10.0]) is None
    # launch count of a complete calibration
    from ant_quantization_amd.ant import quant_modules as qm
    calls = {"multi": 0, "single": 0}
    real_m, real_s = antq_lib.search_sse_multi, antq_lib.search_sse

    def cm(*a, **k):
        calls["multi"] += 1
        return real_m(*a, **k)

    def cs(*a, **k):
        calls["single"] += 1
        return real_s(*a, **k)

    antq_lib.search_sse_multi, antq_lib.search_sse = cm, cs
    try:
        w = torch.randn(96, 1024, device=dev) * 0.02
        q = qm.TensorQuantizer(mode="ant-int-pot-flint", bit=4, is_signed=True, is_enable=True, args=_args()).to(dev)
        q.name = "n1"
        q.alpha.data = torch.ones(96, 1, device=dev)
        out = q(w)
    finally:
        antq_lib.search_sse_multi, antq_lib.search_sse = real_m, real_s
    assert calls == {"multi": 1, "single": 0}, calls
    # ... and it is the calibration the per-type path produces
    antq_lib.lib().antq_debug_set(2, 0)           # no x-domain path: the multi entry declines, four searches as before
    try:
        q2 = qm.TensorQuantizer(mode="ant-int-pot-flint", bit=4, is_signed=True, is_enable=True, args=_args()).to(dev)
        q2.name = "n1"
        q2.alpha.data = torch.ones(96, 1, device=dev)
        out2 = q2(w)
    finally:
        antq_lib.lib().antq_debug_set(2, 1)
    assert q.mode == q2.mode and torch.equal(q.alpha.detach(), q2.alpha.detach()) and torch.equal(out.detach(), out2.detach())
    capsys.readouterr()


@pytest.mark.gpu
def test_calibration_sums_are_bit_reproducible(antq_lib, dev):
    """Every sum of the clip search / type selection is formed in one fixed order (no floating-point atomics): per row by
    the wavefront that owns the row, per tensor from workgroup partials added in a fixed tree.  Ten runs of every launch
    shape give identical bits -- a near-tied pair of candidates cannot resolve differently from run to run or rank to
    rank -- and the output buffer needs no initialisation (it is filled with NaN here)."""
    import torch
    from ant_quantization_amd import core, grids
    torch.manual_seed(5)
    ratios = core._ratios(75, 150, 1, dev)
    flint = antq_lib.plan_for(grids.ant_flint(4, True))
    int8 = antq_lib.plan_for(grids.ant_int(8, True))
    plans = [antq_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "flint", "pot")]
    real_empty = torch.empty

    def poisoned(*a, **k):
        t = real_empty(*a, **k)
        if t.dtype == torch.float64:
            t.fill_(float("nan"))
        return t

    cases = [(torch.randn(512, 4096, device=dev) * 0.02, True),        # rows of 4 tasks
             (torch.randn(512, 4096, device=dev) * 0.02, False),       # one sum over 2 M elements: 1024 workgroup partials
             ((torch.randn(96, 9216, device=dev) * 0.02).bfloat16(), True),
             ((torch.randn(96, 9216, device=dev) * 0.02).bfloat16(), False),
             (torch.randn(64, 147, device=dev) * 0.1, True),           # ragged rows: the element-granular kernel
             (torch.randn(64, 147, device=dev) * 0.1, False),
             (torch.randn(1, 50000, device=dev), True)]                # a single row
    torch.empty = poisoned
    try:
        for x, per_row in cases:
            rows, row_len = x.shape
            xmax = core.row_absmax(x, per_row)
            for plan in (flint, int8):
                ref = antq_lib.search_sse(x, rows, row_len, xmax, per_row, ratios, plan, 10.0)
                assert torch.isfinite(ref).all()
                for _ in range(9):
                    assert torch.equal(antq_lib.search_sse(x, rows, row_len, xmax, per_row, ratios, plan, 10.0), ref)
            if row_len % 8 == 0 and row_len >= 1024:
                ref = antq_lib.search_sse_multi(x, rows, row_len, xmax, per_row, ratios, plans, [10.0] * 3)
                assert ref is not None and torch.isfinite(ref).all()
                for _ in range(9):
                    assert torch.equal(antq_lib.search_sse_multi(x, rows, row_len, xmax, per_row, ratios, plans, [10.0] * 3), ref)
                # the multi kernel forms the very same sums as one search per type
                for t, p in enumerate(plans):
                    assert torch.equal(ref[t], antq_lib.search_sse(x, rows, row_len, xmax, per_row, ratios, p, 10.0)), (t, per_row)
        # the alpha gradient's whole-tensor sum goes through the same fixed-order reduction
        x = torch.randn(777, 4096, device=dev)
        o = x + torch.randn_like(x) * 0.01
        go = torch.randn_like(x)
        ref = antq_lib.alpha_grad(x, o, go, 777, 4096, per_row=False)
        want = (go.double() * (o - x).double()).sum()
        assert torch.isfinite(ref).all() and abs(float(ref) - float(want)) <= 1e-6 * abs(float(want)) + 1e-9
        for _ in range(9):
            assert torch.equal(antq_lib.alpha_grad(x, o, go, 777, 4096, per_row=False), ref)
        # candidate lists longer than one workgroup's accumulators (128): split over blockIdx.y in every kernel
        r150 = core._ratios(1, 151, 1, dev)
        assert r150.numel() == 150
        for x, per_row in cases:
            rows, row_len = x.shape
            xmax = core.row_absmax(x, per_row)
            full = antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150, flint, 10.0)
            assert torch.isfinite(full).all() and torch.equal(full, antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150, flint, 10.0))
            halves = torch.cat([antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150[:75].contiguous(), flint, 10.0),
                                antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150[75:].contiguous(), flint, 10.0)])
            if per_row and rows > 1:
                # (round 6: rows of 512 elements and more take the sorted-row search, whose split of the elements into
                #  step-function and literal ones follows the SMALLEST scale of the launch's candidates -- with a list that starts
                #  at ratio 0.01 the two halves classify differently and agree to the closed form's rounding, not to the bit;
                #  still the same bits on every run.  The direct kernels, knobs 19 = 20 = 0, are split-independent to the bit.)
                torch.testing.assert_close(full, halves, rtol=1e-7, atol=0)
                antq_lib.lib().antq_debug_set(19, 0)
                antq_lib.lib().antq_debug_set(20, 0)
                try:
                    f0 = antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150, flint, 10.0)
                    h0 = torch.cat([antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150[:75].contiguous(), flint, 10.0),
                                    antq_lib.search_sse(x, rows, row_len, xmax, per_row, r150[75:].contiguous(), flint, 10.0)])
                finally:
                    antq_lib.lib().antq_debug_set(19, 1)
                    antq_lib.lib().antq_debug_set(20, 1)
                assert torch.equal(f0, h0)
                torch.testing.assert_close(full, f0, rtol=2e-7, atol=0)
            else:
                # (an fp32 tensor with one scale and >= 1 M elements takes the sorted search too: an element the first half's
                #  smallest scale makes literal contributes (O - x)^2 in double either way -- equal to the double's rounding)
                torch.testing.assert_close(full, halves, rtol=1e-10, atol=0)
    finally:
        torch.empty = real_empty


@pytest.mark.gpu
def test_more_than_2_pow_32_elements(antq_lib, oracle, dev):
    """Maximum sizes: one tensor of 2^32 + 2^22 bf16 elements (8.6 GB in, 8.6 GB out; every index past 32 bits).  The
    oracle cannot run at this size, so the launch over the whole tensor is compared, bit for bit, with launches over row
    blocks of it (each far below 2^32) -- per row, in 16-element groups (static and abs-max in the kernel) and as a
    batched launch -- and the tail rows against the oracle."""
    import torch
    from ant_quantization_amd import grids
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40e9:
        pytest.skip("needs 40 GB of free HBM")
    K = 4096
    rows = (1 << 20) + 1024                       # 2^32 + 2^22 elements
    x = torch.empty(rows, K, dtype=torch.bfloat16, device=dev)
    blk = 1 << 17
    g = torch.Generator(device=dev).manual_seed(11)
    for r0 in range(0, rows, blk):                # filled block-wise (a 17 GB fp32 temporary otherwise)
        r1 = min(rows, r0 + blk)
        x[r0:r1] = (torch.randn(r1 - r0, K, device=dev, generator=g) * 0.02).bfloat16()
    assert x.numel() > (1 << 32)
    plan = antq_lib.plan_for(grids.ant_flint(4, True))
    out = torch.empty_like(x)

    def blocks_equal(full_fn, block_fn):
        full_fn()
        torch.cuda.synchronize()
        tmp = torch.empty(blk, K, dtype=torch.bfloat16, device=dev)
        for r0 in list(range(0, rows, 8 * blk)) + [rows - blk]:       # a sample of blocks incl. the last (indices > 2^32)
            r1 = min(rows, r0 + blk)
            block_fn(r0, r1, tmp[: r1 - r0])
            assert torch.equal(out[r0:r1], tmp[: r1 - r0]), r0

    # per row
    alpha = antq_lib.absmax(x, rows, K)
    assert torch.equal(alpha[-blk:], antq_lib.absmax(x[-blk:], blk, K))
    blocks_equal(lambda: antq_lib.fakequant(x, alpha, plan, 10.0, rows, K, True, out=out),
                 lambda r0, r1, t: antq_lib.fakequant(x[r0:r1], alpha[r0:r1], plan, 10.0, r1 - r0, K, True, out=t))
    # the last rows against the oracle
    xt = x[-4:].view(torch.int16).cpu().numpy().view(np.uint16)
    ref, _ = oracle.forward(xt, alpha[-4:].cpu().numpy(), grids.ant_flint(4, True), 10.0, False)
    assert np.array_equal(out[-4:].view(torch.int16).cpu().numpy().view(np.uint16), ref)
    # 16-element groups, static alpha and abs-max in the kernel
    G = 16
    ag = antq_lib.absmax(x, x.numel() // G, G)
    blocks_equal(lambda: antq_lib.fakequant(x, ag, plan, 10.0, x.numel() // G, G, True, out=out),
                 lambda r0, r1, t: antq_lib.fakequant(x[r0:r1], ag[r0 * K // G: r1 * K // G], plan, 10.0,
                                                      (r1 - r0) * K // G, G, True, out=t))
    blocks_equal(lambda: antq_lib.fakequant_dynamic(x, plan, 10.0, x.numel() // G, G, out=out, want_alpha=False),
                 lambda r0, r1, t: antq_lib.fakequant_dynamic(x[r0:r1], plan, 10.0, (r1 - r0) * K // G, G, out=t,
                                                              want_alpha=False))
    # one batched launch over the whole tensor as a single job
    b = antq_lib.Batch([(x, out, alpha, plan, 10.0, rows, K, True)])
    blocks_equal(b.run,
                 lambda r0, r1, t: antq_lib.fakequant(x[r0:r1], alpha[r0:r1], plan, 10.0, r1 - r0, K, True, out=t))
    del x, out
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_nearest_256_entry_grid_above_its_top_value(antq_lib, oracle, dev):
    """The binary search of antq_nearest probes sorted[p + step - 1]; with m = 256 (a power of two) and x above the top
    entry that index runs to 2 m - 2.  The LDS behind the table used to be whatever the previous kernel left there: run
    kernels with large LDS images in between and compare every call with the oracle."""
    import torch
    from ant_quantization_amd import grids
    rng = np.random.default_rng(97)
    int8 = antq_lib.plan_for(grids.ant_int(8, True))
    junk = torch.randn(512, 4096, device=dev)
    ja = antq_lib.absmax(junk, 512, 4096)
    for trial in range(12):
        g = np.sort(rng.standard_normal(256).astype(np.float32) * np.float32(3.0))
        top = float(g.max())
        x = np.concatenate([np.float32(top) + np.abs(rng.standard_normal(4000)).astype(np.float32) * np.float32(5.0),
                            rng.standard_normal(4000).astype(np.float32) * np.float32(4.0),
                            np.float32([top, np.nextafter(np.float32(top), np.float32(np.inf)), 60000.0, -60000.0])])
        zr, jr = oracle.nearest(x, g)
        for dt in (np.float32, np.float64):
            antq_lib.fakequant(junk, ja, int8, 10.0, 512, 4096, True)          # leaves a 4 KiB+ table image in LDS
            z, j = antq_lib.nearest(to_dev(x.astype(dt), dev), to_dev(g.astype(dt), dev), want_idx=True)
            assert f32_same(z.cpu().numpy().astype(np.float32), zr), (trial, dt)
            assert np.array_equal(j.cpu().numpy().astype(np.int32), jr), (trial, dt)


@pytest.mark.gpu
def test_empty_inputs(antq_lib, dev):
    """Zero-sized tensors are a no-op everywhere (the reference's launcher returns its freshly allocated, empty z / idx,
    KQ/quant_kernel.cu:42-62): operator, fused entry points, dynamic variant, codec, abs-max, and a batch never sees one."""
    import torch
    from ant_quantization_amd import grids, quant_cuda
    g = grids.ant_flint(4, True)
    plan = antq_lib.plan_for(g)
    gt = torch.from_numpy(g).to(dev)
    for dt in (torch.float32, torch.float64, torch.bfloat16):
        x = torch.empty(0, dtype=dt, device=dev)
        z, idx = quant_cuda.quant(x, gt.to(dt) if dt != torch.bfloat16 else gt)
        assert z.shape == (0,) and z.dtype == dt and idx.numel() == 0
    for dt in (torch.float32, torch.bfloat16):
        x = torch.empty(0, 64, dtype=dt, device=dev)
        a = torch.empty(0, dtype=torch.float32, device=dev)
        out, idx = antq_lib.fakequant(x, a, plan, 10.0, 0, 64, True, want_idx=True)
        assert out.shape == (0, 64) and idx.shape == (0, 64)
        out, alpha, _ = antq_lib.fakequant_dynamic(x, plan, 10.0, 0, 64)
        assert out.shape == (0, 64) and alpha.numel() == 0
        assert antq_lib.absmax(x, 0, 64).numel() == 0
        codes = antq_lib.encode4(x, a, plan, 10.0, 0, 64, True)
        assert codes.numel() == 0
        assert antq_lib.decode4(codes, a, plan, 10.0, 0, 64, True, dt).numel() == 0
    with pytest.raises(antq_lib.AntqError):
        antq_lib.Batch([(torch.empty(0, 64, device=dev), torch.empty(0, 64, device=dev), torch.empty(0, device=dev), plan, 10.0, 0, 64, True)])


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 3))))
def test_fuzz_every_launch_form_long_rows(antq_lib, oracle, dev, seed):
    """Rows long enough for the per-row table kernels (and a few that are not), every reference codebook and -- a fifth
    of the cases -- arbitrary value lists, random lengths / scales (heavy clipping, scales down to 2^-60 and up to 2^40,
    zero, negative, infinite, NaN and denormal alphas) / outliers / specials, fp32 and bf16 -- through EVERY launch form of
    the same arithmetic: ordinary launch (+ indices), unordered launch into a caller-owned buffer, one batched launch of
    all tensors of a dtype (mixed task sizes, rotated and fixed maps in one grid), the in-kernel abs-max form per tensor
    and batched, and the packed 4-bit codec where a 4-bit code exists.  Each against the ORACLE.  ANTQ_FUZZ_SEEDS widens
    it (tools/fuzz_campaign.sh)."""
    import torch
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(31000 + seed)
    names = [k for k in G.files if not k.startswith("INVALID")]
    groups = {}
    for case in range(10):
        bf16 = bool(rng.random() < 0.6)
        pick = rng.random()
        if pick < 0.2:
            # an ARBITRARY value list (random size, order, duplicates, -0, entries beyond 32), with or without the pair rule
            g = _random_grid(rng, int(rng.choice([2, 3, 5, 8, 15, 16, 29, 64])))
            if not (g.max() > 0):
                g[-1] = np.float32(1.5)
            gname, n_normal, ovp = "random", 0, bool(rng.random() < 0.5)
            gmax = float(np.abs(g).max()) if rng.random() < 0.5 else float(max(g.max(), 0.5))
        elif pick < 0.6:
            gname = names[rng.integers(0, len(names))]
            g, ovp, n_normal = G[gname], False, 0
            gmax = float(g.max())
        else:
            t, b, s = ["int", "flint"][rng.integers(0, 2)], int(rng.choice([3, 4, 4, 4, 5, 8])), "su"[rng.integers(0, 2)]
            gn = O["%s_b%d_%s" % (t, b, s)]
            g, gmax, ovp, gname, n_normal = np.concatenate([gn, O["outlier_b%d_%s" % (b, s)]]), float(gn.max()), True, "olive_" + s, gn.size
        epv = 8 if bf16 else 4
        kind = rng.integers(0, 5)
        if kind == 0:
            K = epv * int(rng.choice([128, 192, 256, 384, 512, 1024, 1376, 2048, 3584]))        # whole tasks
        elif kind == 1:
            K = epv * int(rng.integers(128, 2100))                                              # partial last task
        elif kind == 2:
            K = epv * int(rng.integers(1, 128))                                                 # short rows
        elif kind == 3:
            K = int(rng.integers(1, 9000))                                                      # ragged
        else:
            K = epv * int(rng.choice([16, 32, 64, 100, 127, 129, 255, 257]))
        rows = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 33, 64]))
        rows = max(1, min(rows, 600_000 // K))
        x = make_x(rng, rows, K, unsigned=gname.endswith("_u"), specials=bool(rng.random() < 0.3)) * np.float32(rng.uniform(0.2, 40))
        m = rng.random((rows, K)) < 0.02
        x[m] *= rng.uniform(8, 200, int(m.sum())).astype(np.float32)
        xh = oracle.f32_to_bf16(x) if bf16 else x
        xf = oracle.bf16_to_f32(xh) if bf16 else xh
        am = safe_absmax(xf).max(1)
        akind = rng.integers(0, 6)
        if akind == 0:
            alpha = am * np.float32(rng.uniform(0.01, 0.08))                                    # most elements clipped
        elif akind == 1:
            alpha = am * np.float32(2.0 ** rng.integers(-60, 40))                               # outside the table path's range
        else:
            alpha = am * rng.uniform(0.2, 1.3, rows)
        alpha = (alpha + 1e-6).astype(np.float32)
        if akind == 2 and rows > 2:
            alpha[1], alpha[2] = 0.0, -alpha[2]
        if akind == 3 and rows > 3:
            alpha[0], alpha[1], alpha[3] = np.inf, np.nan, np.float32(1e-42)       # (a denormal scale)
        per_row = bool(rng.random() < 0.75)
        a_np = alpha if per_row else np.float32(alpha.mean())
        tag = (seed, case, gname, rows, K, bf16, ovp, per_row, int(akind))
        with np.errstate(all="ignore"):
            ref, ridx = oracle.forward(xh, a_np, g, gmax, ovp)
        plan = antq_lib.plan_for(g)
        xt = to_dev(xh, dev, bf16)
        a_t = torch.from_numpy(np.atleast_1d(a_np).astype(np.float32)).to(dev)
        unaligned = bool(rng.random() < 0.15)          # tensors that start 2 / 4 bytes into a 16-byte line: the element kernels

        def like(t):                                   # an output buffer laid out like the input (aligned or not)
            if not unaligned:
                return torch.zeros_like(t)
            base = torch.zeros(t.numel() + 8, dtype=t.dtype, device=dev)
            return base[1:1 + t.numel()].view(t.shape)

        if unaligned:
            x_un = like(xt)
            x_un.copy_(xt)
            xt = x_un
            assert xt.data_ptr() % 16 != 0

        def same(t, r=ref, xh=xh, a_np=a_np, K=K):
            if bf16_same(bf16_bits(t), r, oracle) if bf16 else f32_same(t.cpu().numpy(), r):
                return True
            got = bf16_bits(t).reshape(-1) if bf16 else t.cpu().numpy().reshape(-1).view(np.uint32)
            want = r.reshape(-1) if bf16 else r.reshape(-1).view(np.uint32)
            bad = np.flatnonzero(got != want)
            print("MISMATCH %d of %d; first at %s: x bits %s alpha %s got %s want %s" % (
                bad.size, got.size, bad[:6], [hex(int(v)) for v in xh.reshape(-1).view(np.uint16 if bf16 else np.uint32)[bad[:6]]],
                np.atleast_1d(a_np)[np.minimum(bad[:6] // K, np.atleast_1d(a_np).size - 1)],
                [hex(int(v)) for v in got[bad[:6]]], [hex(int(v)) for v in want[bad[:6]]]))
            return False

        out, idx = antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
        assert same(out), ("ordered", tag)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx), ("indices", tag)
        bufs = [like(xt) for _ in range(2)]
        torch.cuda.synchronize()
        for b in bufs:
            antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, per_row, ovp=ovp, unordered=True, out=b)
        for b in bufs:
            assert same(b), ("unordered", tag)
        groups.setdefault((bf16, ovp), []).append((xt, like(xt), a_t, plan, gmax, rows, K, per_row, ref, tag))
        # in-kernel abs-max (rows only; the reference's dynamic scale is ratio * max|x|, no specials in it)
        if per_row and not np.isnan(xf).any() and not np.isinf(xf).any():
            ratio = float(np.float32(rng.uniform(0.3, 1.1)))
            a_dyn = oracle.absmax(xf, True, ratio)
            with np.errstate(all="ignore"):
                ref_d, _ = oracle.forward(xh, a_dyn, g, gmax, ovp)
            out_d, a_dev, _ = antq_lib.fakequant_dynamic(xt, plan, gmax, rows, K, ratio=ratio, ovp=ovp)
            assert np.array_equal(a_dev.cpu().numpy(), a_dyn), ("dynamic alpha", tag)
            assert same(out_d, ref_d), ("dynamic", tag)
            # the batched form of the same (ratio 1): row lengths it has no single-read kernel for are refused, not guessed
            a_one = oracle.absmax(xf, True, 1.0)
            o_b, a_b = torch.zeros_like(xt), torch.zeros(rows, dtype=torch.float32, device=dev)
            try:
                bd = antq_lib.Batch([(xt, o_b, a_b, plan, gmax, rows, K, True)], ovp=ovp, dynamic=True)
            except antq_lib.AntqError:
                bd = None
            if bd is not None:
                bd.run()
                with np.errstate(all="ignore"):
                    ref_b, _ = oracle.forward(xh, a_one, g, gmax, ovp)
                assert np.array_equal(a_b.cpu().numpy(), a_one), ("dynamic batch alpha", tag)
                assert same(o_b, ref_b), ("dynamic batch", tag)
        # packed 4-bit codes: exist for <= 16 codes (ANT) / 8 + 8 with the identifier (OliVe 4-bit), whole 32-bit words
        four_bit = (g.size <= 16 and not ovp) or (gname != "random" and ovp and n_normal <= 15 and g.size - n_normal <= 8)
        if four_bit and K % 8 == 0 and (per_row or True):
            try:
                codes = antq_lib.encode4(xt, a_t, plan, gmax, rows, K, per_row, n_normal=n_normal, ovp=ovp)
            except antq_lib.AntqError:
                codes = None                       # no 4-bit code for this codebook: refused, not guessed
            if codes is not None:
                # the codes are the oracle's indices (an element the scan never reaches -- NaN, beyond its horizon -- takes
                # the zero code); the decoder returns value * scale, which is the reference's ((q - d) + d) * s wherever that
                # is finite and the element is not clipped beyond twice the outermost value
                zc = np.flatnonzero((g[:n_normal] if ovp else g) == 0)
                want = _oracle_codes(oracle, ridx, n_normal, ovp, int(zc[-1]) if zc.size else None)
                nib = _nibbles(codes, rows, K)
                scanned = (ridx != oracle.IDX_NONE) | bool(zc.size)
                assert np.array_equal(nib[scanned], want[scanned]), ("codes", tag)
                dec = antq_lib.decode4(codes, a_t, plan, gmax, rows, K, per_row, torch.bfloat16 if bf16 else torch.float32,
                                       n_normal=n_normal, ovp=ovp)
                reff = oracle.bf16_to_f32(ref) if bf16 else ref
                a_rows = np.broadcast_to(np.atleast_1d(a_np).astype(np.float32)[:, None], (rows, K)) if per_row else np.float32(a_np)
                with np.errstate(all="ignore"):
                    near = np.isfinite(reff) & (np.abs(xf) <= 1.9 * np.abs(a_rows) * float(np.abs(g).max()) / gmax)
                got = bf16_bits(dec) if bf16 else dec.cpu().numpy().view(np.uint32)
                wantb = ref if bf16 else ref.view(np.uint32)
                bad = np.flatnonzero((got != wantb) & near)
                # (an arbitrary value list has no exact straight-through step -- (q - d) + d is q only when neighbouring
                #  magnitudes lie within a factor of two, as in every reference codebook: there the CODES are the contract)
                assert bad.size == 0 or gname == "random", ("codec", tag, bad[:8], got.reshape(-1)[bad[:8]], wantb.reshape(-1)[bad[:8]])
    for (bf16, ovp), jobs in groups.items():
        antq_lib.Batch([j[:8] for j in jobs], ovp=ovp).run()
        for j in jobs:
            ref = j[8]
            ok = bf16_same(bf16_bits(j[1]), ref, oracle) if bf16 else f32_same(j[1].cpu().numpy(), ref)
            assert ok, ("batched", j[9])


def test_caller_supplied_output_buffers_are_validated(antq_lib, dev):
    """out= of the wrong dtype / size / layout is refused by both bindings (compiled extension and ctypes), by the batch
    builder too; an unordered launch without a caller-owned buffer (or with an index output) is refused."""
    import torch
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.plan_for(g)
    x = torch.randn(8, 1024, device=dev)
    a = x.abs().amax(1).contiguous()
    bad = [torch.empty(8, 1024, device=dev, dtype=torch.bfloat16), torch.empty(8, 512, device=dev),
           torch.empty(1024, 8, device=dev).t(), torch.empty(8, 1024)]
    for o in bad:
        with pytest.raises(antq_lib.AntqError):
            antq_lib.fakequant(x, a, plan, 10.0, 8, 1024, True, out=o)
        with pytest.raises(antq_lib.AntqError):
            antq_lib.Batch([(x, o, a, plan, 10.0, 8, 1024, True)])
    with pytest.raises(antq_lib.AntqError):
        antq_lib.fakequant_dynamic(x, plan, 10.0, 8, 1024, out=bad[0])
    with pytest.raises(antq_lib.AntqError):
        antq_lib.fakequant(x, a, plan, 10.0, 8, 1024, True, unordered=True)
    with pytest.raises(antq_lib.AntqError):
        antq_lib.fakequant(x, a, plan, 10.0, 8, 1024, True, unordered=True, out=torch.empty_like(x), want_idx=True)
    e = antq_lib.ext()
    if e is not None:                    # the extension on its own refuses the same (it may be called directly)
        with pytest.raises(RuntimeError):
            e.fakequant(x, a, plan.host_addr, plan.dev(x.device).data_ptr(), 10.0, 8, 1024, True, antq_lib.FLAG_UNORDERED)
        with pytest.raises(RuntimeError):
            e.fakequant(x, a, plan.host_addr, plan.dev(x.device).data_ptr(), 10.0, 8, 1024, True, 0, bad[1])
    ok = torch.empty_like(x)
    assert antq_lib.fakequant(x, a, plan, 10.0, 8, 1024, True, out=ok) is ok


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_calibrate_one_call_equals_the_stepwise_calibration_and_the_oracle(antq_lib, oracle, dev, tree):
    """antq_calibrate (clip statistic + every type's clip search + per-row pick + type pick in ONE C call, no host sync)
    against (a) the step-by-step composition the modules use -- bit for bit -- and (b) the ORACLE's search_mse per type
    (same candidate except for reference near-ties; the type with the smallest oracle sum unless two sums are within 1e-4).
    Per channel and per tensor, fp32 and bf16, long rows (single-read kernel), short / ragged rows (per-type kernels),
    an empty candidate range, a caller-supplied x_max."""
    import torch
    from ant_quantization_amd import core
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(11)
    if tree == "ant":
        grids_ = [G["int_b4_s"], G["flint_b4_s"], G["pot_b4_s"], G["float_b4_s"], G["apot_b4_s"]]
        gmaxs, ovp, step, stat = [10.0] * 5, False, 1, "absmax"
        lb, ub = 75, 150
    else:
        grids_ = [np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])]
        gmaxs, ovp, step, stat = [float(O["int_b4_s"].max()), float(O["flint_b4_s"].max())], True, 2, "3sigma"
        lb, ub = 75, 250
    plans = [antq_lib.plan_for(g) for g in grids_]
    for (rows, K), per_row, bf16 in [((32, 1024), True, False), ((16, 2048), True, True), ((24, 147), True, False),
                                     ((8, 4096), False, False), ((4, 2048), False, True), ((40, 64), True, True)]:
        x = (rng.standard_normal((rows, K)) * 0.05).astype(np.float32)
        x[rng.random((rows, K)) < 0.01] *= 12
        xh = oracle.f32_to_bf16(x) if bf16 else x
        xf = oracle.bf16_to_f32(xh) if bf16 else xh
        xt = to_dev(xh, dev, bf16)
        r_, k_ = (rows, K) if per_row else (1, rows * K)
        alpha, score, typ, xm = antq_lib.calibrate(xt, rows, K, per_row, plans, gmaxs, lb, ub, step, xmax=stat, ovp=ovp)
        na = rows if per_row else 1
        assert alpha.shape == (len(plans), na) and score.shape == (len(plans),) and typ.numel() == 1
        # (a) the stepwise path: same kernels, same order -> identical bits
        if tree == "ant":
            xm_step = antq_lib.absmax(xt, r_, k_, per_row=per_row) if per_row else antq_lib.absmax(xt, rows, K, per_row=False)
        else:
            xm_step = antq_lib.xmax_3sigma(xt, rows, K, per_row=per_row)
        assert torch.equal(xm, xm_step.reshape(-1)), (tree, rows, K, per_row)
        ratios = core._ratios(lb, ub, step, xt.device)
        sums = []
        for t, (p, gm) in enumerate(zip(plans, gmaxs)):
            sse = antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios, p, gm, ovp=ovp)
            best, al = antq_lib.search_pick(sse, xm, ratios, k_)
            assert torch.equal(al.reshape(-1), alpha[t]), (tree, rows, K, per_row, t)
            sums.append(best.double().sum().item())
            np.testing.assert_allclose(score[t].item(), sums[-1], rtol=1e-6)
        # (b) the oracle
        osum = []
        for t, (g, gm) in enumerate(zip(grids_, gmaxs)):
            rb, ra, trace = oracle.search_mse(xf.reshape(r_, k_) if not per_row else xf, xm.cpu().numpy(), lb, ub, step, g, gm,
                                              ovp, per_row)
            check_alpha_picks("%s_%d" % (tree, t), alpha[t].cpu().numpy(), ra, trace, ratios_of(lb, ub, step), xmax_rtol=0.0)
            osum.append(float(rb.astype(np.float64).sum()))
        order = np.argsort(osum)
        if (osum[order[1]] - osum[order[0]]) > 1e-4 * osum[order[0]]:
            assert int(typ.item()) == int(order[0]), (tree, rows, K, per_row, osum, score.cpu().numpy())
        assert int(typ.item()) == int(np.argsort(score.cpu().numpy(), kind="stable")[0])
    # empty candidate range: alpha = x_max, score = rows * 1e10 for every type, type 0; and a caller-supplied x_max
    xt = to_dev((rng.standard_normal((8, 1024)) * 0.05).astype(np.float32), dev)
    given = torch.full((8,), 0.2, device=dev)
    alpha, score, typ, xm = antq_lib.calibrate(xt, 8, 1024, True, plans, gmaxs, 100, 100, step, xmax=given, ovp=ovp)
    assert torch.equal(alpha, given.expand(len(plans), 8)) and int(typ.item()) == 0
    np.testing.assert_allclose(score.cpu().numpy(), 8e10, rtol=1e-6)
    alpha, score, typ, xm = antq_lib.calibrate(xt, 8, 1024, True, plans[:1], gmaxs[:1], lb, ub, step, xmax=given, ovp=ovp)
    ratios = core._ratios(lb, ub, step, xt.device)
    best, al = antq_lib.search_pick(antq_lib.search_sse(xt, 8, 1024, given, True, ratios, plans[0], gmaxs[0], ovp=ovp), given, ratios, 1024)
    assert torch.equal(al, alpha[0]) and int(typ.item()) == 0


def test_bench_line_contract_on_the_gpu(antq_lib, dev):
    """`python bench.py` prints ONE JSON line with the driver's keys, the roofline object of the dominant kernel (live launch
    time, algorithmic bytes, copy ceiling of the same process) and -- unless switched off -- the CPU baseline; K steps are timed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "3", "--nbuf", "8"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "Gelem/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 8 * 4096 * 4096 * 4
    # value = elements per step / time per step, and the live launch time is (nearly) the step time of a one-launch step
    assert abs(d["value"] - 8 * 4096 * 4096 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    assert abs(r["launch_us"] - d["ms_per_step"] * 1e3) < 0.25 * r["launch_us"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Gelem/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 2))))
def test_calibration_fuzz_random_shapes_vs_oracle(antq_lib, oracle, dev, seed):
    """The calibration kernels on random shapes -- rows of one partial task, of many tasks with a partial last one (4- and
    8-vector tasks), ragged and unaligned rows (element kernel), per row and per tensor, fp32 / bf16, ANT types and OliVe's
    pair rule: every candidate's mean squared error against the oracle's trace (summation-order tolerance), the picks
    against the oracle's (near-ties excepted), through antq_search_sse, antq_search_sse_multi and antq_calibrate."""
    import torch
    from ant_quantization_amd import core
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(77000 + seed)
    for case in range(5):
        olive = bool(rng.random() < 0.4)
        if olive:
            grids_ = [np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])]
            gmaxs, ovp, step, lb, ub = [float(O["int_b4_s"].max()), float(O["flint_b4_s"].max())], True, 2, 75, 140
        else:
            names = list(rng.choice(["int_b4_s", "flint_b4_s", "pot_b4_s", "float_b4_s", "apot_b4_s", "flint_b3_s", "int_b6_s"], 3, replace=False))
            grids_ = [G[n] for n in names]
            gmaxs, ovp, step, lb, ub = [float(g.max()) for g in grids_], False, 1, 80, 115
        bf16 = bool(rng.random() < 0.5)
        epv = 8 if bf16 else 4
        kind = rng.integers(0, 4)
        if kind == 0:
            K = epv * int(rng.integers(128, 2200))              # long rows: 4- / 8-vector tasks, partial last task
        elif kind == 1:
            K = epv * int(rng.integers(1, 128))                 # short rows
        elif kind == 2:
            K = int(rng.integers(3, 3000))                      # ragged
        else:
            K = epv * int(rng.choice([128, 256, 512, 1024, 2048]))
        rows = max(1, min(int(rng.choice([1, 2, 5, 16, 40])), 200_000 // K))
        per_row = bool(rng.random() < 0.6)
        x = (rng.standard_normal((rows, K)) * 0.05).astype(np.float32)
        x[rng.random((rows, K)) < 0.01] *= 12
        xh = oracle.f32_to_bf16(x) if bf16 else x
        xf = oracle.bf16_to_f32(xh) if bf16 else xh
        xt = to_dev(xh, dev, bf16)
        r_, k_ = (rows, K) if per_row else (1, rows * K)
        tag = (seed, case, olive, rows, K, per_row, bf16)
        plans = [antq_lib.plan_for(g) for g in grids_]
        alpha, score, typ, xm = antq_lib.calibrate(xt, rows, K, per_row, plans, gmaxs, lb, ub, step,
                                                  xmax="3sigma" if olive else "absmax", ovp=ovp)
        ratios = core._ratios(lb, ub, step, xt.device)
        xm_np = xm.cpu().numpy()
        osum = []
        for t, (g, gm) in enumerate(zip(grids_, gmaxs)):
            rb, ra, trace = oracle.search_mse(xf.reshape(r_, k_), xm_np, lb, ub, step, g, gm, ovp, per_row)
            sse = antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios, plans[t], gm, ovp=ovp)
            np.testing.assert_allclose((sse / k_).float().cpu().numpy(), trace, rtol=3e-5, atol=1e-12, err_msg=str(tag + (t,)))
            check_alpha_picks("fuzz_%d_%d_%d" % (seed, case, t), alpha[t].cpu().numpy(), ra, trace, ratios_of(lb, ub, step), xmax_rtol=0.0)
            osum.append(float(rb.astype(np.float64).sum()))
        multi = antq_lib.search_sse_multi(xt, r_, k_, xm, per_row, ratios, plans, gmaxs, ovp=ovp)
        if multi is not None:
            for t, (p, gm) in enumerate(zip(plans, gmaxs)):
                one = antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios, p, gm, ovp=ovp)
                np.testing.assert_allclose(multi[t].cpu().numpy(), one.cpu().numpy(), rtol=1e-12, err_msg=str(tag + (t, "multi")))
        order = np.argsort(osum)
        if len(osum) > 1 and (osum[order[1]] - osum[order[0]]) > 1e-4 * max(osum[order[0]], 1e-30):
            assert int(typ.item()) == int(order[0]), (tag, osum, score.cpu().numpy())


@pytest.mark.parametrize("seed", range(int(os.environ.get("ANTQ_FUZZ_SEEDS", 2))))
def test_fp16_io_fuzz_every_launch_form(antq_lib, oracle, dev, seed):
    """fp16 tensors (what model.half() hands the quantisers): the kernels compute the fp32 path on the widened input and
    round the result to half ONCE -- i.e. half(oracle(float(x))) -- through every launch form: ordinary (+ indices),
    unordered, batched, in-kernel abs-max, 4-bit codec.  Reference codebooks, random row lengths, planted outliers."""
    import torch
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(91000 + seed)
    names = [k for k in G.files if not k.startswith("INVALID")]
    jobs = {}
    for case in range(8):
        if rng.random() < 0.5:
            gname = names[rng.integers(0, len(names))]
            g, ovp, n_normal = G[gname], False, 0
            gmax = float(g.max())
        else:
            t, b, sg = ["int", "flint"][rng.integers(0, 2)], int(rng.choice([3, 4, 4, 5])), "su"[rng.integers(0, 2)]
            gn = O["%s_b%d_%s" % (t, b, sg)]
            g, gmax, ovp, gname, n_normal = np.concatenate([gn, O["outlier_b%d_%s" % (b, sg)]]), float(gn.max()), True, "olive_" + sg, gn.size
        K = int(rng.choice([8 * int(rng.integers(128, 1500)), 8 * int(rng.integers(1, 128)), int(rng.integers(1, 5000)), 1024, 4096]))
        rows = max(1, min(int(rng.choice([1, 3, 8, 33])), 300_000 // K))
        x = make_x(rng, rows, K, unsigned=gname.endswith("_u"), specials=False) * np.float32(rng.uniform(0.2, 20))
        m = rng.random((rows, K)) < 0.02
        x[m] *= rng.uniform(8, 100, int(m.sum())).astype(np.float32)
        xh = x.astype(np.float16)
        xf = xh.astype(np.float32)
        alpha = (np.abs(xf).max(1) * rng.uniform(0.05, 1.2, rows) + 1e-4).astype(np.float32)
        per_row = bool(rng.random() < 0.75)
        a_np = alpha if per_row else np.float32(alpha.mean())
        with np.errstate(all="ignore"):
            ref32, ridx = oracle.forward(xf, a_np, g, gmax, ovp)
            ref = ref32.astype(np.float16)
        tag = (seed, case, gname, rows, K, ovp, per_row)
        plan = antq_lib.plan_for(g)
        xt = torch.from_numpy(xh).to(dev)
        a_t = torch.from_numpy(np.atleast_1d(a_np).astype(np.float32)).to(dev)

        def same(t, r=ref):
            got = t.cpu().numpy().view(np.uint16).reshape(-1)
            want = r.view(np.uint16).reshape(-1)
            return bool(np.all((got == want) | (np.isnan(t.cpu().numpy().reshape(-1)) & np.isnan(r.reshape(-1)))))

        out, idx = antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
        assert same(out), ("ordered", tag)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32), ridx), ("indices", tag)
        buf = torch.empty_like(xt)
        torch.cuda.synchronize()
        antq_lib.fakequant(xt, a_t, plan, gmax, rows, K, per_row, ovp=ovp, unordered=True, out=buf)
        assert same(buf), ("unordered", tag)
        jobs.setdefault(ovp, []).append((xt, torch.zeros_like(xt), a_t, plan, gmax, rows, K, per_row, ref, tag))
        if per_row:
            a_dyn = oracle.absmax(xf, True, 1.0)
            with np.errstate(all="ignore"):
                ref_d = oracle.forward(xf, a_dyn, g, gmax, ovp)[0].astype(np.float16)
            out_d, a_dev, _ = antq_lib.fakequant_dynamic(xt, plan, gmax, rows, K, ovp=ovp)
            assert np.array_equal(a_dev.cpu().numpy(), a_dyn) and same(out_d, ref_d), ("dynamic", tag)
        four_bit = (g.size <= 16 and not ovp) or (ovp and n_normal <= 15 and g.size - n_normal <= 8)
        if four_bit and K % 8 == 0:
            try:
                codes = antq_lib.encode4(xt, a_t, plan, gmax, rows, K, per_row, n_normal=n_normal, ovp=ovp)
            except antq_lib.AntqError:
                codes = None
            if codes is not None:
                zc = np.flatnonzero((g[:n_normal] if ovp else g) == 0)
                want = _oracle_codes(oracle, ridx, n_normal, ovp, int(zc[-1]) if zc.size else None)
                scanned = (ridx != oracle.IDX_NONE) | bool(zc.size)
                assert np.array_equal(_nibbles(codes, rows, K)[scanned], want[scanned]), ("codes", tag)
                dec = antq_lib.decode4(codes, a_t, plan, gmax, rows, K, per_row, torch.float16, n_normal=n_normal, ovp=ovp)
                a_rows = np.broadcast_to(np.atleast_1d(a_np).astype(np.float32)[:, None], (rows, K)) if per_row else np.float32(a_np)
                near = np.isfinite(ref32) & (np.abs(xf) <= 1.9 * np.abs(a_rows) * float(np.abs(g).max()) / gmax)
                bad = np.flatnonzero((dec.cpu().numpy().view(np.uint16) != ref.view(np.uint16)) & near)
                assert bad.size == 0, ("codec", tag, bad[:8])
    for ovp, js in jobs.items():
        antq_lib.Batch([j[:8] for j in js], ovp=ovp).run()
        for j in js:
            got = j[1].cpu().numpy()
            assert bool(np.all((got.view(np.uint16) == j[8].view(np.uint16)) | (np.isnan(got) & np.isnan(j[8])))), ("batched", j[9])


@pytest.mark.gpu
def test_16bit_domain_row_kernels_on_every_pattern(antq_lib, oracle, dev):
    """K1h on the device (csrc/antq_k_hrow.h: k_fq_hrow one tensor per launch, k_fq_hbatch many): rows that hold EVERY one of
    the 65 536 bf16 / f16 patterns (in order and shuffled: other pairs, other lane / vector positions), one row per scale --
    ordinary ones from 1e-9 to 1e9, zero, negative, NaN, Inf, denormal-range and overflowing ones -- against the oracle's
    fp32 sequence rounded to 16 bits, bit for bit, for the headline codebooks and a few 3- / 5-bit ones, with and without
    the pair rule; the same rows through the batched launch, through 3- and 2-vector tasks (rows of 576 / 128 vectors) and
    with the round-3 fp32-domain kernels (knob 9 = 0), which must agree as well."""
    import torch
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(41)
    allpat = np.arange(65536, dtype=np.uint16)
    books = [("flint_b4_s", G["flint_b4_s"], None, False), ("int_b4_s", G["int_b4_s"], None, False),
             ("flint_b4_u", G["flint_b4_u"], None, False), ("pot_b4_s", G["pot_b4_s"], None, False),
             ("float_b5_s", G["float_b5_s"], None, False), ("int_b3_u", G["int_b3_u"], None, False),
             ("olive_flint_b4_s", np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]), float(O["flint_b4_s"].max()), True),
             ("olive_int_b4_s", np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), float(O["int_b4_s"].max()), True),
             ("olive_flint_b4_u", np.concatenate([O["flint_b4_u"], O["outlier_b4_u"]]), float(O["flint_b4_u"].max()), True)]
    knob = antq_lib.lib().antq_debug_set
    for name, g, gmax, olive in books:
        g = np.ascontiguousarray(g, dtype=np.float32)
        gmax = float(g.max()) if gmax is None else gmax
        plan = antq_lib.plan_for(g)
        assert plan.is_table and int(plan.host[:128].view(np.uint32)[24]) == 3, name      # hdom for both dtypes
        alphas = np.concatenate([np.float32([1.0, 0.06, 0.0, -0.05, np.nan, np.inf, 1e-30, 1e30, 65504.0, 6e-8]),
                                 np.exp(rng.uniform(-20, 20, 6)).astype(np.float32)])
        rows = len(alphas)
        for tdt, npdt in ((torch.bfloat16, None), (torch.float16, np.float16)):
            for pats in (allpat, rng.permutation(allpat)):
                x16 = np.ascontiguousarray(np.broadcast_to(pats, (rows, 65536)))
                xf = oracle.bf16_to_f32(x16) if npdt is None else x16.view(np.float16).astype(np.float32)
                xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(tdt)
                at = torch.from_numpy(alphas).to(dev)
                for ovp in ((False, True) if olive else (False,)):
                    with np.errstate(all="ignore"):
                        ref, _ = oracle.forward(xf, alphas, g, gmax, ovp)
                        ref16 = oracle.f32_to_bf16(ref) if npdt is None else ref.astype(np.float16).view(np.uint16)

                    def same(t, what):
                        got = t.view(torch.int16).cpu().numpy().view(np.uint16).reshape(ref16.shape)
                        gf = oracle.bf16_to_f32(got) if npdt is None else got.view(np.float16).astype(np.float32)
                        rf = oracle.bf16_to_f32(ref16) if npdt is None else ref16.view(np.float16).astype(np.float32)
                        bad = ~((got == ref16) | (np.isnan(gf) & np.isnan(rf)))
                        assert not bad.any(), (name, str(tdt), ovp, what, int(bad.sum()), np.argwhere(bad)[:3].tolist(),
                                               x16[bad][:3], got[bad][:3], ref16[bad][:3])

                    same(antq_lib.fakequant(xt, at, plan, gmax, rows, 65536, True, ovp=ovp), "one launch")
                    knob(0, 8)      # (what an ordered launch of 1024 ... 4096 whole-row wavefronts takes by itself)
                    same(antq_lib.fakequant(xt, at, plan, gmax, rows, 65536, True, ovp=ovp), "one launch, 8-vector tasks")
                    knob(0, 0)
                    same(antq_lib.fakequant(xt, at, plan, gmax, rows, 65536, True, ovp=ovp, out=torch.empty_like(xt), unordered=True), "unordered")
                    knob(9, 0)
                    same(antq_lib.fakequant(xt, at, plan, gmax, rows, 65536, True, ovp=ovp), "round-3 kernels")
                    knob(9, 1)
                    out = torch.empty_like(xt)
                    antq_lib.Batch([(xt, out, at, plan, gmax, rows, 65536, True)], ovp=ovp).run()
                    same(out, "batched")
                    # the same elements as rows of 576 vectors (tasks of 3 vectors per lane) and of 128 vectors (2): per-row
                    # scales repeated so that every short row keeps the scale of the long row it came from
                    for rl in (4608, 1024):
                        if 65536 % rl:
                            n_keep = (65536 // rl) * rl
                            xs = xt[:, :n_keep].contiguous().view(-1, rl)
                            a2 = at.repeat_interleave(n_keep // rl)
                            keep = ref16[:, :n_keep]
                        else:
                            n_keep = 65536
                            xs, a2, keep = xt.view(-1, rl), at.repeat_interleave(65536 // rl), ref16
                        if ovp and rl % 2:
                            continue
                        o2 = torch.empty_like(xs)
                        antq_lib.Batch([(xs, o2, a2, plan, gmax, xs.shape[0], rl, True)], ovp=ovp).run()
                        got = o2.view(torch.int16).cpu().numpy().view(np.uint16).reshape(keep.shape)
                        gf = oracle.bf16_to_f32(got) if npdt is None else got.view(np.float16).astype(np.float32)
                        rf = oracle.bf16_to_f32(keep) if npdt is None else keep.view(np.float16).astype(np.float32)
                        assert np.all((got == keep) | (np.isnan(gf) & np.isnan(rf))), (name, str(tdt), ovp, rl)
                        o3 = antq_lib.fakequant(xs, a2, plan, gmax, xs.shape[0], rl, True, ovp=ovp)
                        assert torch.equal(o3.view(torch.int16), o2.view(torch.int16)), (name, rl)
    # a per-tensor scale (one row however the tensor is shaped) and an in-place launch
    g = np.ascontiguousarray(G["flint_b4_s"], dtype=np.float32)
    plan = antq_lib.plan_for(g)
    x16 = rng.permutation(allpat)
    xt = torch.from_numpy(np.tile(x16, 4).view(np.int16)).to(dev).view(torch.bfloat16).view(64, 4096)
    a1 = torch.tensor([0.37], device=dev)
    with np.errstate(all="ignore"):
        ref, _ = oracle.forward(oracle.bf16_to_f32(np.tile(x16, 4)).reshape(1, -1), np.float32([0.37]), g, 10.0, False)
    ref16 = oracle.f32_to_bf16(ref).reshape(-1)
    got = antq_lib.fakequant(xt, a1, plan, 10.0, 64, 4096, False)
    assert bf16_same(bf16_bits(got).reshape(-1), ref16, oracle)
    xi = xt.clone()
    antq_lib.fakequant(xi, a1, plan, 10.0, 64, 4096, False, out=xi)
    assert bf16_same(bf16_bits(xi).reshape(-1), ref16, oracle)


class _ThreadComm:
    """Two 'ranks' on one device: each runs in its own thread on its own stream; all_reduce meets at a barrier."""

    def __init__(self, world):
        import threading
        self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world

    def rank(self, r):
        comm = self

        class R:
            def all_reduce(self, t, op):
                import torch
                torch.cuda.current_stream().synchronize()
                comm.slots[r] = t.clone()
                torch.cuda.current_stream().synchronize()
                comm.barrier.wait()
                acc = comm.slots[0].clone()
                for v in comm.slots[1:]:
                    acc = torch.maximum(acc, v) if op == "max" else acc + v          # rank order, like a ring's fixed order
                torch.cuda.current_stream().synchronize()
                comm.barrier.wait()
                t.copy_(acc)
        return R()


@pytest.mark.gpu
def test_sharded_per_tensor_calibration_two_ranks_on_one_device(antq_lib, oracle, dev):
    """sharding.sharded_calibrate with the HIP kernels (GpuBlockOps): a per-tensor quantiser's tensor cut into two row
    blocks, each 'rank' reducing its block on its own stream, the three all-reduces done by an in-process communicator --
    against the unsharded calibration of the whole tensor (antq_calibrate, one C call) and against the oracle's search:
    same x_max (abs-max exactly; 3-sigma to the statistic's documented tolerance), same alpha per type, same type."""
    import threading
    import torch
    from ant_quantization_amd import grids, sharding
    rng = np.random.default_rng(5)
    cases = [("olive", torch.float32, (64 * 128, 768)), ("olive", torch.bfloat16, (2048, 3072)), ("ant", torch.float32, (999, 1024)),
             ("ant", torch.bfloat16, (4096, 768))]
    for tree, tdt, (rows, K) in cases:
        x = torch.from_numpy((rng.standard_normal((rows, K)) * 0.6).astype(np.float32)).to(dev)
        if tree == "olive":
            m = torch.from_numpy(rng.random((rows, K)) < 0.002).to(dev)
            x[m] *= 40.0
            fulls = [np.concatenate([grids.olive_grid(t, 4, True), grids.olive_outliers(4, True)]).astype(np.float32) for t in ("int", "flint")]
            gmaxs = [float(grids.olive_grid(t, 4, True).max()) for t in ("int", "flint")]
            lb, ub, step, stat, ovp = 75, 250, 2, "3sigma", True
        else:
            fulls = [np.ascontiguousarray(grids.ant_grid(t, 4, True), dtype=np.float32) for t in ("int", "pot", "flint")]
            gmaxs = [float(g.max()) for g in fulls]
            lb, ub, step, stat, ovp = 80, 150, 1, "absmax", False
        x = x.to(tdt).contiguous()
        plans = [antq_lib.plan_for(g) for g in fulls]
        # unsharded: the whole calibration as one C call
        a_ref, s_ref, t_ref, xm_ref = antq_lib.calibrate(x, rows, K, False, plans, gmaxs, lb, ub, step, xmax=stat, ovp=ovp)
        comm, res, errs = _ThreadComm(2), [None, None], []

        def worker(r):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    b, e = sharding.row_block(rows, r, 2, pair_safe_row_len=K)
                    res[r] = sharding.sharded_calibrate(x[b:e], rows * K, plans, gmaxs, lb, ub, step, statistic=stat, ovp=ovp,
                                                        group=comm.rank(r))
                    torch.cuda.current_stream().synchronize()
            except Exception as ex:           # noqa: BLE001
                errs.append(ex)
                comm.barrier.abort()

        th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not errs, errs
        r0, r1 = res
        assert torch.equal(r0["xmax"], r1["xmax"]) and torch.equal(r0["alpha"], r1["alpha"]) and r0["type"] == r1["type"]
        if stat == "absmax":
            assert torch.equal(r0["xmax"], xm_ref.reshape(1))
        else:
            np.testing.assert_allclose(r0["xmax"].cpu().numpy(), xm_ref.cpu().numpy(), rtol=2.0 ** -7 if tdt == torch.bfloat16 else 2e-6)
        if torch.equal(r0["xmax"], xm_ref.reshape(1)):
            # same statistic -> the same candidates: picks must agree unless two candidates tie to the last digit of the score
            sa, sr = r0["score"].cpu().numpy(), s_ref.cpu().numpy()
            np.testing.assert_allclose(sa, sr, rtol=1e-6)
            for t in range(len(plans)):
                if float(r0["alpha"][t]) != float(a_ref[t, 0]):
                    assert abs(sa[t] - sr[t]) <= 1e-6 * sr[t], (tree, t)
            assert r0["type"] == int(t_ref.item()) or abs(sr[r0["type"]] - sr.min()) <= 1e-6 * sr.min()
        # ... and the oracle on the whole tensor with the sharded statistic (fp32 cases: the oracle's search is fp32)
        if tdt == torch.float32 and rows * K <= 1 << 20:
            xn = x.cpu().numpy()
            for t, (g, gm) in enumerate(zip(fulls, gmaxs)):
                bs, ba, tr = oracle.search_mse(xn, r0["xmax"].cpu().numpy(), lb, ub, step, g, gm, ovp=ovp, per_row=False)
                if float(ba[0]) != float(r0["alpha"][t]):
                    ci = int(round((float(r0["alpha"][t]) / float(r0["xmax"][0]) * 100 - lb) / step))
                    assert abs(float(tr[ci, 0]) - float(tr.min())) <= 2e-6 * float(tr.min()), (tree, t)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["default", "resident"])
@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_auto_weight_bank_is_the_default_and_bit_identical(antq_lib, dev, tree, schedule, capsys):
    """enable_quantization(model) arms weight_bank.AutoBank: after the calibrating forward every later forward that needs no
    gradient serves ALL weight quantisers from one batched launch -- default schedule: ONE launch per forward (the
    reference re-quantises every weight on every forward); resident (set_weights_at_rest): no weight launch at all while
    weights and alphas are unchanged, ONE launch after they change -- with outputs bit-identical to the per-layer schedule
    (set_weight_bank(model, False) = the reference's).  A ResNet-shaped stack (conv rows of 27 / 576 / 1152 / 4608 elements, ragged conv1) and a
    BERT-shaped one (768 / 3072-wide Linear layers), fp32 and bf16; training-mode forwards with gradients bypass the bank."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4, w_up=150, a_up=150))
    torch.manual_seed(11)
    convs = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(),
                          nn.Conv2d(128, 512, 3, padding=1), nn.ReLU(), nn.Conv2d(512, 64, 3, padding=1), nn.ReLU(),
                          nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(64, 10))
    berts = nn.Sequential(nn.Linear(768, 768), nn.GELU(), nn.Linear(768, 3072), nn.GELU(), nn.Linear(3072, 768), nn.Linear(768, 2))
    for net, x in ((convs, torch.randn(8, 3, 16, 16, device=dev)), (berts, torch.randn(64, 768, device=dev))):
        for dt in (torch.float32, torch.bfloat16):
            model = qmod.quantize_model(net).to(dev).eval()
            qutil.enable_quantization(model)
            resident = schedule == "resident"
            per = 0 if resident else 1                          # bank launches per no-grad forward on unchanged weights
            if resident:
                qutil.set_weights_at_rest(model, True)
            ab = model._antq_auto_bank
            assert ab is not None and ab.enabled and ab.bank is None and ab.resident == resident
            with torch.no_grad():
                y0 = model(x)                                   # calibration: per-layer path, nothing attached yet
                assert ab.bank is None
                model = model.to(dt)
                xd = x.to(dt)
                y1 = model(xd)                                  # first steady forward: the bank attaches and refreshes once
                assert ab.bank is not None and ab.bank.launches == 1 and not ab.bank.skipped
                nq = sum(1 for m in model.modules() if hasattr(m, "quant_weight"))
                assert len(ab.bank.entries) == nq
                y2 = model(xd)
                y3 = model(xd)
                assert ab.bank.launches == 1 + 2 * per          # resident, unchanged weights: no weight launch at all
                assert torch.equal(y1, y2) and torch.equal(y2, y3)
                qutil.set_weight_bank(model, False)             # the reference's schedule
                assert ab.bank is None and all(m.quant_weight._bank is None for m in model.modules() if hasattr(m, "quant_weight"))
                y_ref = model(xd)
                assert torch.equal(y_ref, y1)
                qutil.set_weight_bank(model, True)
                lin = [m for m in model.modules() if hasattr(m, "quant_weight")]
                lin[0].weight.mul_(1.01)                        # a weight edit: ONE refresh, then quiet again
                y4 = model(xd)
                assert ab.bank is not None and ab.bank.launches == 1 and not torch.equal(y4, y1)
                y5 = model(xd)
                assert ab.bank.launches == 1 + per and torch.equal(y4, y5)
                lin[1].quant_weight.alpha.mul_(0.9)
                y6 = model(xd)
                assert ab.bank.launches == 2 + per
                qutil.set_weight_bank(model, False)
                assert torch.equal(model(xd), y6)
                qutil.set_weight_bank(model, True)
            # gradients through the quantisers: the autograd path, not the bank
            if dt == torch.float32 and tree == "ant":
                model.train()
                before = model._antq_auto_bank.bank.launches if model._antq_auto_bank.bank is not None else 0
                out = model(x)
                out.float().sum().backward()
                assert lin[0].weight.grad is not None
                assert (model._antq_auto_bank.bank.launches if model._antq_auto_bank.bank is not None else 0) == before
                model.eval()
    # copies and pickles of a model do not drag the bank along
    import copy
    import pickle
    m2 = copy.deepcopy(model)
    assert all(m.quant_weight._auto_bank is None and m.quant_weight._bank is None or True for m in m2.modules() if hasattr(m, "quant_weight"))
    pickle.dumps(model.state_dict())
    capsys.readouterr()


def test_search_pick_selection_rule(antq_lib, dev):
    """antq_search_pick (one wavefront per row, candidates over the lanes) against the reference's sequential loop
    (AQ:299-306: best = 1e10, ascending candidates, strict '<'): ties keep the earliest candidate, NaN scores and scores of
    1e10 or more are never taken (alpha stays x_max), candidate lists shorter and longer than a wavefront."""
    import torch
    rng = np.random.default_rng(11)
    for ncand, na in ((1, 1), (7, 5), (64, 3), (65, 9), (76, 1), (76, 300), (200, 17)):
        row_len = 768
        sse = rng.random((ncand, na)) * 50.0
        # ties (exact duplicates, some of them the minimum), NaNs, huge scores, whole rows without any eligible candidate
        for r in range(na):
            k = rng.integers(0, 6)
            if k == 0 and ncand > 2:
                i, j = sorted(rng.choice(ncand, 2, replace=False))
                sse[i, r] = sse[j, r] = 0.001
            elif k == 1:
                sse[rng.integers(0, ncand), r] = np.nan
            elif k == 2:
                sse[:, r] = 1e10 * row_len * (1.0 + rng.random(ncand))
            elif k == 3:
                sse[:, r] = np.nan
            elif k == 4 and ncand > 70:
                sse[:, r] = 5.0
                sse[[3, 67], r] = 1.0                     # the same minimum in two different lanes' strides and in one lane's
                sse[67 - 64, r] = 1.0
        xm = (rng.random(na) + 0.5).astype(np.float32)
        ratios = np.float32([np.float32((75 + c) * 0.01) for c in range(ncand)])
        best_ref = np.full(na, 1e10, np.float32)
        alpha_ref = xm.copy()
        for r in range(na):
            for c in range(ncand):
                with np.errstate(all="ignore"):
                    score = np.float32(sse[c, r] / float(row_len))
                if score < best_ref[r]:
                    best_ref[r] = score
                    alpha_ref[r] = np.float32(xm[r] * ratios[c])
        best, alpha = antq_lib.search_pick(torch.from_numpy(sse).to(dev), torch.from_numpy(xm).to(dev),
                                           torch.from_numpy(ratios).to(dev), row_len)
        assert f32_same(best.cpu().numpy(), best_ref), (ncand, na)
        assert f32_same(alpha.cpu().numpy(), alpha_ref), (ncand, na)


def test_search_sums_do_not_depend_on_the_grid_split(antq_lib, dev, oracle):
    """The clip-search launches split the candidate list over blockIdx.y by a cost model (rounds of resident workgroups x work
    per workgroup, antq_search.hip: search_grid); the sums of squared errors must not depend on that split: every candidate's
    sum is formed in one fixed order whatever chunk it rides in.  Checked by evaluating the same candidates as ONE list and
    as several shorter lists (each its own launch, hence its own split), per row and per tensor, fp32 and bf16, several sizes
    (few / many workgroups per round), one codebook and the multi-codebook kernel."""
    import torch
    from ant_quantization_amd import grids
    rng = np.random.default_rng(5)
    plans = [antq_lib.plan_for(grids.ant_flint(4, True)), antq_lib.plan_for(grids.ant_int(4, True))]
    ratios_np = np.float32([np.float32(i * 0.01) for i in range(75, 151)])
    for (rows, K), per_row, bf16 in (((768, 768), True, False), ((96, 3072), True, True), ((2048, 768), False, False),
                                     ((8192, 768), False, False), ((1024, 3072), False, True), ((3, 512), True, False)):
        x = (rng.standard_normal((rows, K)) * 0.05).astype(np.float32)
        xh = oracle.f32_to_bf16(x) if bf16 else x
        xt = to_dev(xh, dev, bf16)
        r_, k_ = (rows, K) if per_row else (1, rows * K)
        xm = antq_lib.absmax(xt, r_, k_, per_row=per_row).reshape(-1)
        ratios = torch.from_numpy(ratios_np).to(dev)
        whole = antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios, plans[0], 10.0)
        for cut in (1, 5, 19, 40):
            parts = [antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios[a:a + cut].contiguous(), plans[0], 10.0)
                     for a in range(0, 76, cut)]
            assert torch.equal(torch.cat(parts, 0), whole), (rows, K, per_row, bf16, cut)
        multi = antq_lib.search_sse_multi(xt, r_, k_, xm, per_row, ratios, plans, [10.0, 7.0])
        if multi is not None:
            assert torch.equal(multi[0], whole), (rows, K, per_row, bf16)
            second = antq_lib.search_sse(xt, r_, k_, xm, per_row, ratios, plans[1], 7.0)
            assert torch.equal(multi[1], second), (rows, K, per_row, bf16)
    # long candidate lists: 300 ratios (3 chunks of <= 128 at least), and 4 codebooks x 175 ratios = 700 flat entries on one read
    x = (rng.standard_normal((64, 1024)) * 0.05).astype(np.float32)
    xt = to_dev(x, dev)
    xm = antq_lib.absmax(xt, 64, 1024, per_row=True).reshape(-1)
    long_r = torch.from_numpy(np.float32([np.float32(i * 0.005) for i in range(100, 400)])).to(dev)
    whole = antq_lib.search_sse(xt, 64, 1024, xm, True, long_r, plans[0], 10.0)
    parts = [antq_lib.search_sse(xt, 64, 1024, xm, True, long_r[a:a + 50].contiguous(), plans[0], 10.0) for a in range(0, 300, 50)]
    # (round 6, the sorted-row search: this list starts at ratio 0.5, where a row's largest element sits at the edge of the
    #  codebook's step-function domain, |x / s| = 2 max|v| -- in the launches whose smallest scale is that one the element is
    #  evaluated on its own, (O - x)^2 in double, in the others inside the closed form: the same number to the double's
    #  rounding, not to the bit.  Lists from ratio 0.75 -- every case above -- have no such element and are equal to the bit;
    #  so are the direct kernels, knobs 19 = 20 = 0, on this list.)
    torch.testing.assert_close(torch.cat(parts, 0), whole, rtol=1e-12, atol=0)
    antq_lib.lib().antq_debug_set(19, 0)
    antq_lib.lib().antq_debug_set(20, 0)
    try:
        whole0 = antq_lib.search_sse(xt, 64, 1024, xm, True, long_r, plans[0], 10.0)
        parts0 = [antq_lib.search_sse(xt, 64, 1024, xm, True, long_r[a:a + 50].contiguous(), plans[0], 10.0) for a in range(0, 300, 50)]
    finally:
        antq_lib.lib().antq_debug_set(19, 1)
        antq_lib.lib().antq_debug_set(20, 1)
    assert torch.equal(torch.cat(parts0, 0), whole0)
    torch.testing.assert_close(whole, whole0, rtol=3e-7, atol=0)
    r175 = torch.from_numpy(np.float32([np.float32(i * 0.01) for i in range(75, 250)])).to(dev)
    four = [plans[0], plans[1], antq_lib.plan_for(grids.ant_pot(4, True)), antq_lib.plan_for(grids.ant_int(4, False))]
    multi = antq_lib.search_sse_multi(xt, 64, 1024, xm, True, r175, four, [10.0, 7.0, 10.0, 15.0])
    assert multi is not None and multi.shape[:2] == (4, 175)
    # (round 6, the sorted-row search: the UNSIGNED codebook's step-function domain ends below this signed tensor's largest
    #  elements at the small ratios; with it in the launch those elements are evaluated on their own for every codebook --
    #  (O - x)^2 in double, what the closed form gives them otherwise: equal to the double's rounding (measured 3e-14), and to
    #  the bit once the unsigned codebook is left out or the direct kernels run)
    for t, (p_, g_) in enumerate(zip(four, [10.0, 7.0, 10.0, 15.0])):
        torch.testing.assert_close(multi[t], antq_lib.search_sse(xt, 64, 1024, xm, True, r175, p_, g_), rtol=1e-12, atol=0)
    multi3 = antq_lib.search_sse_multi(xt, 64, 1024, xm, True, r175, four[:3], [10.0, 7.0, 10.0])
    for t, (p_, g_) in enumerate(zip(four[:3], [10.0, 7.0, 10.0])):
        assert torch.equal(multi3[t], antq_lib.search_sse(xt, 64, 1024, xm, True, r175, p_, g_)), t
    antq_lib.lib().antq_debug_set(19, 0)
    antq_lib.lib().antq_debug_set(20, 0)
    try:
        multi0 = antq_lib.search_sse_multi(xt, 64, 1024, xm, True, r175, four, [10.0, 7.0, 10.0, 15.0])
        for t, (p_, g_) in enumerate(zip(four, [10.0, 7.0, 10.0, 15.0])):
            assert torch.equal(multi0[t], antq_lib.search_sse(xt, 64, 1024, xm, True, r175, p_, g_)), t
    finally:
        antq_lib.lib().antq_debug_set(19, 1)
        antq_lib.lib().antq_debug_set(20, 1)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_sign_probe_and_lazy_log_values_on_the_calibration_pass(tree, dev, capsys):
    """The wrapper layers ask for the input's sign BEFORE they calibrate their weight (prefetch_sign: the minimum travels to
    pinned memory while the weight's calibration is issued) and `mse` is formed when read: same signedness, alphas, grids,
    outputs and `mse` as quantisers calibrated without the probe (update_signed reading on the spot), for inputs with and
    without negative values; a probe taken on ANOTHER tensor object, or on one edited in place since, is not trusted."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))

    def make():
        torch.manual_seed(9)
        net = nn.Sequential(nn.Linear(256, 512), nn.ReLU(), nn.Linear(512, 128))
        m = qmod.quantize_model(net).to(dev).eval()
        qutil.enable_quantization(m)
        return m

    torch.manual_seed(1)
    x = torch.randn(64, 256, device=dev)
    with torch.no_grad():
        a = make()
        ya = a(x)                                            # probes on
        b = make()
        for m in b.modules():
            if hasattr(m, "quant_input"):
                m.quant_input.prefetch_sign = lambda t: None     # reference behaviour: the read happens on the spot
        yb = b(x)
        assert torch.equal(ya, yb)
        la = [m for m in a.modules() if hasattr(m, "quant_input")]
        lb = [m for m in b.modules() if hasattr(m, "quant_input")]
        assert [m.quant_input.is_signed for m in la] == [m.quant_input.is_signed for m in lb] == [True, False]
        for ma, mb in zip(la, lb):
            for qa, qb in ((ma.quant_input, mb.quant_input), (ma.quant_weight, mb.quant_weight)):
                assert torch.equal(qa.alpha, qb.alpha) and torch.equal(qa.quant_grid, qb.quant_grid)
                assert qa._sign_probe is None
                assert torch.equal(qa.mse, qb.mse) and qa.mse.numel() == 1 and float(qa.mse) > 0
                assert qa.mse is qa.mse                      # formed once
        # a stale probe: taken on a tensor that is edited in place afterwards / on another object
        c = make()
        lin = [m for m in c.modules() if hasattr(m, "quant_input")][0]
        pos = torch.rand(64, 256, device=dev)
        lin.quant_input.prefetch_sign(pos)
        pos.sub_(0.5)                                        # now it has negative values; the probe saw none
        lin.quant_input.update_signed(pos)
        assert lin.quant_input.is_signed
        d = make()
        lin = [m for m in d.modules() if hasattr(m, "quant_input")][0]
        lin.quant_input.prefetch_sign(torch.rand(64, 256, device=dev))
        lin.quant_input.update_signed(-torch.rand(64, 256, device=dev))
        assert lin.quant_input.is_signed and lin.quant_input._sign_probe is None
    capsys.readouterr()


@pytest.mark.parametrize("tree,mode", [("ant", "flint"), ("ant", "ant-int-flint"), ("olive", "flint"), ("olive", "ant-int-flint")])
def test_parallel_branches_search_their_shared_input_once(tree, mode, dev, capsys):
    """Query / key / value projections calibrate their input quantisers on ONE tensor object: the per-tensor clip search
    (and the type selection's pass) runs once and is looked up twice (core.SearchMemo) -- same alphas, grids and outputs as
    with the memo switched off; an in-place edit of the tensor (version counter) or another tensor object is searched anew;
    every quantiser owns its alpha storage."""
    import importlib
    import torch
    import torch.nn as nn
    from ant_quantization_amd import core
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode=mode, wbit=4, abit=4))

    class QKV(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v, self.o = nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 64)

        def forward(self, x):
            return self.o(self.q(x) * torch.sigmoid(self.k(x)) + self.v(x))

    def make():
        torch.manual_seed(4)
        m = qmod.quantize_model(QKV()).to(dev).eval()
        qutil.enable_quantization(m)
        return m

    torch.manual_seed(2)
    x = torch.randn(128, 256, device=dev)
    with torch.no_grad():
        core.search_memo.clear()
        core.search_memo.hits = 0
        a = make()
        ya = a(x)
        assert core.search_memo.hits >= 2                        # k and v found q's search (type pass and / or clip search)
        hits = core.search_memo.hits
        core.search_memo.enabled = False
        try:
            b = make()
            yb = b(x)
        finally:
            core.search_memo.enabled = True
        assert core.search_memo.hits == hits and torch.equal(ya, yb)
        for (na, qa), (nb, qb) in zip(a.named_modules(), b.named_modules()):
            if hasattr(qa, "quant_input"):
                for u, v in ((qa.quant_input, qb.quant_input), (qa.quant_weight, qb.quant_weight)):
                    assert u.mode == v.mode and torch.equal(u.alpha, v.alpha) and torch.equal(u.quant_grid, v.quant_grid), na
        ptrs = [m.quant_input.alpha.data_ptr() for m in a.modules() if hasattr(m, "quant_input")]
        assert len(set(ptrs)) == len(ptrs)                       # nobody shares the memo's tensors
        # the same object edited in place, and a different object with the same values: searched anew
        core.search_memo.clear()
        core.search_memo.hits = 0
        c = make()
        lin = [m for m in c.modules() if hasattr(m, "quant_input")]
        x2 = x.clone()
        lin[0].quant_input(x2)
        x2.mul_(2.0)
        lin[1].quant_input(x2)
        lin[2].quant_input(x2.clone())
        assert core.search_memo.hits == 0
        assert torch.equal(lin[1].quant_input.alpha, lin[2].quant_input.alpha)
        assert not torch.equal(lin[0].quant_input.alpha, lin[1].quant_input.alpha)
    capsys.readouterr()


@pytest.mark.parametrize("tree,mode", [("ant", "flint"), ("ant", "ant-int-pot-flint"), ("ant", "ant-int-float2-flint"),
                                       ("olive", "flint"), ("olive", "ant-int-flint")])
def test_weights_calibrated_in_one_batch_before_the_first_layer(tree, mode, dev, capsys):
    """antq_calibrate_batch through the model path (weight_bank.AutoBank.precalibrate): every weight quantiser is calibrated
    by ONE call when the first layer's forward starts, the type picks come back in one copy -- and every quantiser ends up in
    exactly the state its own first forward would have produced (mode, codebook, alpha bits, mse), the model prints the same
    lines in the same order and returns the same output.  Conv rows that are not whole vectors (K = 27), Linear rows, an
    8-bit layer (window 95..), fp32 and bf16; quantisers the batch cannot take (float1-4 types) keep the per-layer path."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode=mode, wbit=4, abit=4))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.c2 = nn.Conv2d(3, 16, 3, padding=1), nn.Conv2d(16, 32, 3, padding=1)
            self.f1, self.f2 = nn.Linear(32 * 8 * 8, 256), nn.Linear(256, 10)

        def forward(self, x):
            x = torch.relu(self.c2(torch.relu(self.c1(x))))
            return self.f2(torch.relu(self.f1(x.flatten(1))))

    for dt in (torch.float32, torch.bfloat16):
        outs, states, logs, counts = [], [], [], []
        for batch in (True, False):
            torch.manual_seed(12)
            model = qmod.quantize_model(Net()).to(dev).to(dt).eval()
            qs = [m for m in model.modules() if hasattr(m, "quant_weight")]
            qs[1].quant_weight.bit.data = torch.tensor(8, device=dev)          # an 8-bit layer: 'int', window from 95
            qs[1].quant_weight.rearm()
            capsys.readouterr()
            qutil.enable_quantization(model)
            model._antq_auto_bank.batch_calibration = 2 if batch else 0        # (2: fixed modes too; the default batches only type selections)
            torch.manual_seed(13)
            x = torch.randn(8, 3, 8, 8, device=dev).to(dt)
            from ant_quantization_amd import core
            per_layer = []
            real = (core.clip_search, core.clip_search_types)
            core.clip_search = lambda t, xm, pc, *a, **k: (per_layer.append(pc), real[0](t, xm, pc, *a, **k))[1]
            core.clip_search_types = lambda t, xm, pc, *a, **k: (per_layer.append(pc), real[1](t, xm, pc, *a, **k))[1]
            try:
                with torch.no_grad():
                    outs.append(model(x))
            finally:
                core.clip_search, core.clip_search_types = real
            # (per-channel searches issued by layers: none for the weights the batch took)
            assert sum(per_layer) == 0 if (batch and "float2" not in mode) else sum(per_layer) >= 4, (per_layer, batch, mode)
            logs.append(capsys.readouterr().out)
            counts.append(model._antq_auto_bank.precalibrated)
            states.append([(q.quant_weight.mode, q.quant_weight.quant_grid.clone(), q.quant_weight.alpha.detach().clone(),
                            q.quant_weight.mse.clone(), q.quant_input.alpha.detach().clone(), q.quant_input.mode) for q in qs])
        takes = 0 if "float2" in mode else 4
        assert counts == [takes, 0], (counts, mode)
        assert logs[0] == logs[1] and logs[0].count("-bit") == 8, (tree, mode, dt)
        assert torch.equal(outs[0], outs[1]), (tree, mode, dt)
        for (m0, g0, a0, e0, ia0, im0), (m1, g1, a1, e1, ia1, im1) in zip(*states):
            assert m0 == m1 and im0 == im1 and torch.equal(g0, g1) and torch.equal(a0, a1) and torch.equal(ia0, ia1), (tree, mode, dt)
            np.testing.assert_allclose(e0.float().cpu().numpy(), e1.float().cpu().numpy(), rtol=1e-5)
        assert states[0][1][0] == "int"
    # default setting: a fixed mode keeps the (sync-free) per-layer path, a type selection is batched; a weight edited between
    # the batch's search and its layer's forward is searched again by the layer
    torch.manual_seed(12)
    model = qmod.quantize_model(Net()).to(dev).eval()
    qutil.enable_quantization(model)
    ab = model._antq_auto_bank
    assert ab.batch_calibration == 1
    qs = [m for m in model.modules() if hasattr(m, "quant_weight")]
    with torch.no_grad():
        ab.precalibrate()
        batched = mode.startswith("ant-") and "float2" not in mode
        assert ab.precalibrated == (4 if batched else 0)
        qs[2].weight.mul_(3.0)
        torch.manual_seed(13)
        x = torch.randn(8, 3, 8, 8, device=dev)
        y = model(x)
        torch.manual_seed(12)
        ref = qmod.quantize_model(Net()).to(dev).eval()
        qutil.enable_quantization(ref)
        ref._antq_auto_bank.batch_calibration = 0
        [m for m in ref.modules() if hasattr(m, "quant_weight")][2].weight.mul_(3.0)
        assert torch.equal(ref(x), y)
        for a, b in zip(qs, [m for m in ref.modules() if hasattr(m, "quant_weight")]):
            assert torch.equal(a.quant_weight.alpha, b.quant_weight.alpha) and a.quant_weight.mode == b.quant_weight.mode
    capsys.readouterr()


def test_calibrate_batch_equals_calibrate_per_job(antq_lib, dev):
    """antq_calibrate_batch(jobs) = antq_calibrate(job) for every job, bit for bit: mixed shapes (ragged rows, short rows,
    long rows), one and several candidate codebooks, per row and per tensor, ANT (abs-max) and OliVe (3 sigma, pair rule)."""
    import torch
    from ant_quantization_amd import grids
    rng = np.random.default_rng(21)
    plans_a = [antq_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
    on = grids.olive_grid("flint", 4, True)
    plan_o = antq_lib.plan_for(np.concatenate([on, grids.olive_outliers(4, True)]))
    shapes = [(16, 27, True), (64, 576, True), (8, 4096, True), (32, 768, False), (3, 100, True), (12, 2048, True)]
    for dt in (torch.float32, torch.bfloat16):
        xs = [torch.from_numpy((rng.standard_normal((r, k)) * 0.05).astype(np.float32)).to(dev).to(dt) for r, k, _ in shapes]
        for stat, ovp, plans, gm, lb, ub, step in (("absmax", False, plans_a, [7.0, 64.0, 10.0], 75, 150, 1),
                                                   ("absmax", False, plans_a[2:], [10.0], 95, 150, 1),
                                                   ("3sigma", True, [plan_o], [float(on.max())], 75, 250, 2)):
            jobs = [(x, r, k, pr, plans, gm, lb, ub, step) for x, (r, k, pr) in zip(xs, shapes)]
            res, types = antq_lib.calibrate_batch(jobs, xmax=stat, ovp=ovp)
            assert types.shape == (len(jobs),)
            for i, (x, (r, k, pr)) in enumerate(zip(xs, shapes)):
                a1, s1, t1, xm1 = antq_lib.calibrate(x, r, k, pr, plans, gm, lb, ub, step, xmax=stat, ovp=ovp)
                assert torch.equal(res[i][0], a1) and torch.equal(res[i][1], s1) and torch.equal(res[i][2], xm1.reshape(-1)), (dt, stat, i)
                assert int(types[i]) == int(t1[0])
    res, types = antq_lib.calibrate_batch([])
    assert res == [] and types is None


@pytest.mark.parametrize("tree,mode", [("ant", "ant-int-pot-flint"), ("olive", "ant-int-flint")])
def test_type_pick_stays_on_the_device_for_input_quantisers(tree, mode, dev, capsys):
    """`ant-...` modes, input quantisers of a model armed by enable_quantization: search, pick, alpha, codebook and this
    forward's output are chosen by a device-side index (antq_calibrate + gathers), the host learns the pick when the model's
    forward returns (forward hook) -- no read-back that drains the stream per quantiser.  Same modes, codebooks, alphas,
    `mse`, outputs and printed lines (in the same order) as with the pick read on the spot; a layer called outside the
    model's forward names its pick at its next call; a forward that wants gradients through the quantisers does not defer."""
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode=mode, wbit=4, abit=4))

    class QKV(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v, self.o = nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 64)

        def forward(self, x):
            return self.o(torch.relu(self.q(x)) * torch.sigmoid(self.k(x)) + self.v(x))

    def make(defer):
        torch.manual_seed(4)
        m = qmod.quantize_model(QKV()).to(dev).eval()
        capsys.readouterr()
        qutil.enable_quantization(m)
        m._antq_auto_bank.defer_types = defer
        return m

    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(2)
        x = torch.randn(128, 256, device=dev).to(dt)
        res = []
        for defer in (True, False):
            m = make(defer).to(dt)
            reads = []
            real_cpu, real_item = torch.Tensor.cpu, torch.Tensor.item
            torch.Tensor.cpu = lambda self, *a, **k: (reads.append("cpu"), real_cpu(self, *a, **k))[1]
            torch.Tensor.item = lambda self: (reads.append("item"), real_item(self))[1]
            try:
                with torch.no_grad():
                    y = m(x)
            finally:
                torch.Tensor.cpu, torch.Tensor.item = real_cpu, real_item
            log = capsys.readouterr().out
            qs = [l for l in m.modules() if hasattr(l, "quant_input")]
            assert all(l.quant_input._pending is None and l.quant_input._steady and l.quant_input.mode in ("int", "flint", "pot")
                       for l in qs)
            if defer:
                assert m._antq_auto_bank.deferred == 4 and len(reads) <= 1, reads        # (the weights' picks: one copy)
            else:
                assert m._antq_auto_bank.deferred == 0 and len(reads) >= 3
            with torch.no_grad():
                y2 = m(x)
            assert torch.equal(y, y2) and capsys.readouterr().out == ""
            res.append((y, log, [(l.quant_input.mode, l.quant_input.quant_grid.clone(), l.quant_input.alpha.detach().clone(),
                                  l.quant_input.mse.clone(), l.quant_weight.mode, l.quant_weight.alpha.detach().clone()) for l in qs]))
        (ya, la, sa), (yb, lb, sb) = res
        assert torch.equal(ya, yb) and la == lb and la.count("-bit") == 8, (tree, dt)
        for a, b in zip(sa, sb):
            assert a[0] == b[0] and a[4] == b[4] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[5], b[5])
            np.testing.assert_allclose(a[3].float().cpu().numpy(), b[3].float().cpu().numpy(), rtol=1e-5)
    # a layer used outside the model's forward: its pick is named at its next call; gradients wanted: read on the spot
    m = make(True)
    lin = [l for l in m.modules() if hasattr(l, "quant_input")]
    x = torch.randn(128, 256, device=dev)
    with torch.no_grad():
        y0 = lin[0](x)
        assert lin[0].quant_input._pending is not None and lin[0].quant_input.mode == mode
        y1 = lin[0](x)
        assert lin[0].quant_input._pending is None and lin[0].quant_input.mode in ("int", "flint", "pot") and torch.equal(y0, y1)
    if tree == "ant":
        out = lin[1](x.clone().requires_grad_(True))
        assert lin[1].quant_input._pending is None and lin[1].quant_input._steady and out.requires_grad
    capsys.readouterr()


def _fuzz_seeds(default):
    return int(os.environ.get("ANTQ_FUZZ_SEEDS", default))


@pytest.mark.parametrize("seed", range(_fuzz_seeds(3)))
def test_calibration_pass_fuzz_fast_schedule_equals_step_by_step(seed, dev, capsys):
    """Random small models (shared inputs, conv + linear, ragged conv rows, random widths), random tree / mode / bit widths /
    dtype / windows: the calibration pass with everything on (sign probes, search memo, weights searched in one batch, type
    picks kept on the device, lazy log values) against the step-by-step schedule (all of it off) -- same modes, codebooks,
    alphas, outputs of the calibrating and of the following forward, and the same printed lines in the same order."""
    import importlib
    import torch
    import torch.nn as nn
    from ant_quantization_amd import core
    rng = np.random.default_rng(1000 + seed)
    tree = ("ant", "olive")[int(rng.integers(0, 2))]
    modes = {"ant": ["flint", "int", "ant-int-flint", "ant-int-pot-flint", "ant-int-pot-float-flint", "ant-int-float2-flint"],
             "olive": ["flint", "int", "ant-int-flint"]}[tree]
    mode = modes[int(rng.integers(0, len(modes)))]
    dt = (torch.float32, torch.bfloat16, torch.float16)[int(rng.integers(0, 3))]
    wbit, abit = int(rng.choice([3, 4, 4, 5])), int(rng.choice([4, 4, 5]))
    lo, up = int(rng.integers(60, 96)), int(rng.integers(100, 180))
    c_in, c_mid = int(rng.choice([3, 4, 8])), int(rng.choice([8, 16, 24]))
    hw, feat = 6, int(rng.choice([64, 96, 128]))
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(_args(mode=mode, wbit=wbit, abit=abit, w_low=lo, a_low=lo, w_up=up, a_up=up,
                              no_outlier=bool(rng.integers(0, 4) == 0)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Conv2d(c_in, c_mid, 3, padding=1)
            self.q, self.k, self.v = nn.Linear(c_mid * hw * hw, feat), nn.Linear(c_mid * hw * hw, feat), nn.Linear(c_mid * hw * hw, feat)
            self.o = nn.Linear(feat, 10)

        def forward(self, x):
            h = torch.relu(self.c1(x)).flatten(1)
            return self.o(self.q(h) * torch.sigmoid(self.k(h)) + self.v(h))

    torch.manual_seed(seed)
    x = (torch.randn(16, c_in, hw, hw, device=dev) * float(rng.uniform(0.2, 3.0))).to(dt)
    res = []
    for fast in (True, False):
        torch.manual_seed(seed + 77)
        m = qmod.quantize_model(Net()).to(dev).to(dt).eval()
        capsys.readouterr()
        qutil.enable_quantization(m)
        ab = m._antq_auto_bank
        if not fast:
            ab.batch_calibration, ab.defer_types = 0, False
            for l in m.modules():
                if hasattr(l, "quant_input"):
                    l.quant_input.prefetch_sign = lambda t: None
        core.search_memo.clear()
        core.search_memo.enabled = fast
        try:
            with torch.no_grad():
                y1 = m(x)
                log = capsys.readouterr().out
                y2 = m(x)
        finally:
            core.search_memo.enabled = True
        st = []
        for l in m.modules():
            if hasattr(l, "quant_input"):
                for q in (l.quant_weight, l.quant_input):
                    assert q._steady and q._pending is None
                    st.append((q.mode, q.is_signed, q.quant_grid.clone(), q.alpha.detach().clone(), q.mse.clone()))
        res.append((y1, y2, log, st))
    (a1, a2, la, sa), (b1, b2, lb, sb) = res
    tag = (seed, tree, mode, str(dt), wbit, abit, lo, up)
    assert la == lb and la.count("-bit") == 10, tag
    for (m0, s0, g0, al0, e0), (m1, s1, g1, al1, e1) in zip(sa, sb):
        assert m0 == m1 and s0 == s1 and torch.equal(g0, g1) and torch.equal(al0, al1), tag
        np.testing.assert_allclose(e0.float().cpu().numpy(), e1.float().cpu().numpy(), rtol=1e-5, err_msg=str(tag))
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(a1, a2), tag
    capsys.readouterr()


def test_ordered_launch_with_whole_row_wavefronts(antq_lib, oracle, dev):
    """An ordinary (ordered) launch of 1024 ... 4096 rows of 512 / 1024 vectors takes 8 vectors per lane (antq_fq.hip:
    launch_hrow): 1024 x 4096 and 2048 x 8192 bf16 / f16, ANT and OliVe pairs, against the oracle; the unordered launch of the
    same tensor (4-vector tasks) and the forced 4-vector ordered one agree bit for bit."""
    import torch
    from ant_quantization_amd import grids
    rng = np.random.default_rng(77)
    knob = antq_lib.lib().antq_debug_set
    gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
    for (rows, K) in ((1024, 4096), (2048, 8192)):
        x = (rng.standard_normal((rows, K)) * 0.03).astype(np.float32)
        x.reshape(-1)[::4099] *= 25.0
        for f16 in (False, True):
            xh = x.astype(np.float16).view(np.uint16) if f16 else oracle.f32_to_bf16(x)
            xf = xh.view(np.float16).astype(np.float32) if f16 else oracle.bf16_to_f32(xh)
            xt = torch.from_numpy(xh.view(np.int16)).to(dev).view(torch.float16 if f16 else torch.bfloat16)
            for g, gmax, ovp, ratio in ((grids.ant_flint(4, True), 10.0, False, 0.9), (np.concatenate([gn, go]), float(gn.max()), True, 0.25)):
                alpha = (np.abs(xf).max(1) * np.float32(ratio)).astype(np.float32)
                ref, _ = oracle.forward(xf, alpha, np.ascontiguousarray(g, dtype=np.float32), gmax, ovp)
                ref16 = ref.astype(np.float16).view(np.uint16) if f16 else oracle.f32_to_bf16(ref)
                plan = antq_lib.plan_for(g)
                at = torch.from_numpy(alpha).to(dev)
                got = antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp)
                assert np.array_equal(got.view(torch.int16).cpu().numpy().view(np.uint16), ref16), (rows, K, f16, ovp)
                un = antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp, out=torch.empty_like(xt), unordered=True)
                knob(0, 4)
                four = antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp)
                knob(0, 0)
                assert torch.equal(un.view(torch.int16), got.view(torch.int16)) and torch.equal(four.view(torch.int16), got.view(torch.int16))


def test_absmax_into_accumulates(antq_lib, oracle, dev):
    """antq_absmax_into: max(initial, max|x|) in one launch -- a fresh maximum from a zeroed slot (what _lib.absmax does per
    tensor), a tensor fed in pieces, an initial value above the data, NaN, ragged / unaligned pieces; fp32, bf16, f16."""
    import ctypes
    import torch
    L = antq_lib.lib()
    rng = np.random.default_rng(3)
    for tdt, code in ((torch.float32, 0), (torch.bfloat16, 1), (torch.float16, 2)):
        x = (torch.from_numpy((rng.standard_normal(1 << 20) * 3).astype(np.float32)).to(dev)).to(tdt)
        want = x.float().abs().max()
        got = antq_lib.absmax(x, 1024, 1024, per_row=False)
        assert got.shape == (1,) and torch.equal(got[0], want)
        acc = torch.zeros(1, device=dev)
        for a, b in ((0, 1000), (1000, 300001), (300001, 1 << 20)):         # pieces that start at odd element offsets
            piece = x[a:b]
            assert L.antq_absmax_into(piece.data_ptr(), acc.data_ptr(), b - a, code, None) == 0
        assert torch.equal(acc[0], want)
        big = torch.full((1,), 1e6, device=dev)
        assert L.antq_absmax_into(x.data_ptr(), big.data_ptr(), x.numel(), code, None) == 0 and float(big) == 1e6
        xn = x.clone()
        xn[12345] = float("nan")
        acc = torch.zeros(1, device=dev)
        assert L.antq_absmax_into(xn.data_ptr(), acc.data_ptr(), xn.numel(), code, None) == 0 and torch.isnan(acc).all()
    assert L.antq_absmax_into(None, None, 0, 0, None) == 0 and L.antq_absmax_into(None, None, 8, 0, None) == -1
    # many calls: slots are never handed out twice
    seen = set()
    for _ in range(5000):
        t = antq_lib._zero_slot(dev)
        assert t.data_ptr() not in seen
        seen.add(t.data_ptr())
