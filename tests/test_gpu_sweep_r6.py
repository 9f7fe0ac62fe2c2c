"""Round 6: the threshold-sweep clip search (csrc/antq_k_sweep.h) against the direct kernels and the oracle.
(Since the sorted-row search, csrc/antq_k_sortsearch.h / tests/test_gpu_sort_r6.py, takes these launches first, every test here
switches it off -- knob 20 = 0 -- and the sweep is what runs behind it.)

A per-row clip search scores every candidate by the squared error of the whole row (AQ:287-326, OQ:189-233); the sweep
kernel forms the same sums from a histogram of threshold crossings instead of C evaluations per element.  Bar: the same
PICK for every row and codebook (certified by the oracle's own scores where they tie), sums equal to the direct kernels' to
their rounding (<= 5e-7: the direct kernels round each squared term to fp32 as the reference does, the sweep does not), the
same NaN pattern; literal elements (NaN / Inf / far-clipped / OliVe pairs that may hold an outlier) take the reference sequence.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _ratios(lb, ub, step, dev):
    return torch.tensor([np.float32(i * 0.01) for i in range(lb, ub, step)], dtype=torch.float32, device=dev)


def _both(L, x, rows, K, xm, rt, plans, gmaxs, ovp):
    out = []
    for knob in (0, 2):
        L.lib().antq_debug_set(19, knob); L.lib().antq_debug_set(20, 0)
        try:
            s = L.search_sse_multi(x, rows, K, xm, True, rt, plans, gmaxs, ovp=ovp) if len(plans) > 1 else None
            if s is None:
                s = torch.stack([L.search_sse(x, rows, K, xm, True, rt, p, g, ovp=ovp) for p, g in zip(plans, gmaxs)])
        finally:
            L.lib().antq_debug_set(19, 1); L.lib().antq_debug_set(20, 1)
        out.append(s.clone())
    return out


def _compare(a, b, what, rtol=5e-7):
    an, bn = torch.isnan(a), torch.isnan(b)
    assert torch.equal(an, bn), what
    ok = ~an & torch.isfinite(a)
    rel = ((a - b).abs() / a.abs().clamp_min(1e-300))[ok]
    assert rel.numel() == 0 or float(rel.max()) <= rtol, (what, float(rel.max()))
    # the pick of every (codebook, row): first strict minimum along the candidates
    fa = torch.where(torch.isnan(a), torch.full_like(a, float("inf")), a)
    fb = torch.where(torch.isnan(b), torch.full_like(b, float("inf")), b)
    pa, pb = fa.argmin(1), fb.argmin(1)
    diff = (pa != pb)
    if diff.any():        # only where the direct kernel's own two scores tie to its rounding
        ia, ib = pa[diff], pb[diff]
        t, r = diff.nonzero(as_tuple=True)
        gap = (a[t, ib, r] - a[t, ia, r]).abs() / a[t, ia, r].abs()
        assert float(gap.max()) <= 2e-7, (what, float(gap.max()))


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16", "float16"])
def test_sweep_equals_direct_ant_codebooks(dev, dtype_name):
    from ant_quantization_amd import _lib as L, grids
    dt = getattr(torch, dtype_name)
    torch.manual_seed(61)
    types = ("int", "pot", "flint", "float")
    plans = [L.plan_for(grids.ant_grid(t, 4, True)) for t in types]
    for rows, K, lb, ub in ((96, 4096, 80, 150), (33, 2048 + 64, 75, 150), (64, 512, 95, 101), (17, 8192, 75, 76), (40, 264, 75, 150)):
        x = (torch.randn(rows, K, device=dev) * 0.03)
        x[::7] *= 0.2
        x = x.to(dt)
        xm = L.absmax(x, rows, K)
        a, b = _both(L, x, rows, K, xm, _ratios(lb, ub, 1, dev), plans, [10.0] * 4, False)
        _compare(a, b, (dtype_name, rows, K))
    # unsigned codebooks on a non-negative tensor, one type at a time (antq_search_sse)
    pu = [L.plan_for(grids.ant_grid(t, 4, False)) for t in ("int", "flint")]
    x = torch.nn.functional.relu(torch.randn(48, 3072, device=dev)).to(dt)
    xm = L.absmax(x, 48, 3072)
    for p in pu:
        a, b = _both(L, x, 48, 3072, xm, _ratios(75, 150, 1, dev), [p], [10.0], False)
        _compare(a, b, (dtype_name, "unsigned"))


def test_sweep_literal_elements_and_unusable_rows(dev):
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(62)
    plans = [L.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "flint")]
    x = torch.randn(24, 1024, device=dev) * 0.02
    x[0] = 0.0                                   # x_max = 0: no usable scale, the literal sequence for every element
    x[1, 5] = float("nan")
    x[2, 7] = float("inf")
    x[3, 9] = -float("inf")
    x[4, 11] = 1e30                              # x_max = 1e30: every other element is tiny next to it
    x[5] = 0.5
    x[6, ::2] = 0.0
    x[7] = -x[7].abs()
    x[8, 100:140] *= 300.0                       # far-clipped against the given statistic below
    xm = L.absmax(x, 24, 1024)
    xm[8] = 0.05
    a, b = _both(L, x, 24, 1024, xm, _ratios(75, 150, 1, dev), plans, [10.0, 10.0], False)
    _compare(a, b, "edge rows")
    assert torch.isnan(b[:, :, 1]).all() and torch.isnan(b[:, :, 2]).all() and torch.isnan(b[:, :, 3]).all()
    # a ratio list that is not ascending: the kernel notices and evaluates literally
    rt = _ratios(75, 150, 1, dev).flip(0).contiguous()
    a, b = _both(L, x[9:], 15, 1024, xm[9:].contiguous(), rt, plans, [10.0, 10.0], False)
    _compare(a, b, "descending ratios")
    # ... and one that ascends irregularly (no arithmetic progression: the flip candidates are bisected)
    rt = torch.tensor([0.5, 0.51, 0.7, 0.71, 0.72, 0.9, 1.3, 1.31, 2.0], device=dev)
    a, b = _both(L, x[9:], 15, 1024, xm[9:].contiguous(), rt, plans, [10.0, 10.0], False)
    _compare(a, b, "irregular ratios")


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_sweep_olive_pairs_against_direct_and_oracle(dev, oracle, dtype_name):
    """OliVe codebooks (forced onto the sweep path: by default their 28 thresholds keep the direct kernels) with planted
    outliers: pairs that may hold an outlier under some candidate are evaluated literally with the pair rule (OQ:311-320)."""
    from ant_quantization_amd import _lib as L, grids
    dt = getattr(torch, dtype_name)
    torch.manual_seed(63)
    oo = grids.olive_outliers(4, True)
    cb = [(np.concatenate([grids.olive_grid(t, 4, True), oo]), float(grids.olive_grid(t, 4, True).max())) for t in ("int", "flint")]
    plans, gm = [L.plan_for(g) for g, _ in cb], [m for _, m in cb]
    rows, K = 40, 2048
    x = torch.randn(rows, K, device=dev) * 0.02
    idx = torch.randint(0, x.numel(), (x.numel() // 300,), device=dev)
    x.view(-1)[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
    x = x.to(dt)
    xm = L.xmax_3sigma(x, rows, K, per_row=True)
    rt = _ratios(75, 250, 2, dev)
    for ovp in (True, False):
        a, b = _both(L, x, rows, K, xm, rt, plans, gm, ovp)
        _compare(a, b, (dtype_name, "olive", ovp))
    # a few rows against the oracle's own search (the reference's op sequence on the fp32 image of the tensor)
    xn = x[:6].float().cpu().numpy()
    L.lib().antq_debug_set(19, 2); L.lib().antq_debug_set(20, 0)
    try:
        s = L.search_sse_multi(x[:6].contiguous(), 6, K, xm[:6].contiguous(), True, rt, plans, gm, ovp=True)
    finally:
        L.lib().antq_debug_set(19, 1); L.lib().antq_debug_set(20, 1)
    for t, (g, m) in enumerate(cb):
        best, alpha, trace = oracle.search_mse(xn, xm[:6].cpu().numpy(), 75, 250, 2, g, m, ovp=True, per_row=True)
        got = (s[t] / K).cpu().numpy()                       # [ncand, rows] mean squared error
        np.testing.assert_allclose(got, trace, rtol=3e-6)
        assert np.array_equal(got.argmin(0), trace.argmin(0)) or np.allclose(np.take_along_axis(trace, got.argmin(0)[None], 0), trace.min(0), rtol=6e-7)


def test_calibrate_picks_do_not_depend_on_the_search_path(dev):
    """antq_calibrate (x_max, every codebook's search, per-row picks, the type pick) with the sweep on (default rule: rows of
    2048 elements and more), forced everywhere, and off: the same alphas, scores to rounding and the same type."""
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(64)
    plans = [L.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
    x = torch.distributions.Laplace(0.0, 0.02).sample((128, 4096)).to(dev)
    res = []
    for knob in (0, 1, 2):
        L.lib().antq_debug_set(19, knob); L.lib().antq_debug_set(20, 0)
        try:
            alpha, score, typ, xmax = L.calibrate(x, 128, 4096, True, plans, [10.0] * 3, 75, 150, 1, xmax="absmax")
            res.append((alpha.clone(), score.clone(), int(typ)))
        finally:
            L.lib().antq_debug_set(19, 1); L.lib().antq_debug_set(20, 1)
    (a0, s0, t0), (a1, s1, t1), (a2, s2, t2) = res
    assert t0 == t1 == t2
    assert torch.equal(a1, a2) and torch.equal(s1, s2)                  # default rule == forced on these rows
    same = (a0 == a1)
    assert float(same.float().mean()) >= 0.995                          # (a differing row: a tie of the direct kernel's own scores)
    assert torch.allclose(s0, s1, rtol=1e-6)


def test_sweep_one_scale_fp32_tensors(dev, oracle):
    """A tensor with ONE scale (every activation quantiser, AQ:51-53, :308-324) in fp32 -- the only dtype the reference itself
    runs -- through the sweep spread over many workgroups (integer slabs, two-level totals, fixed-order doubles): sums equal to
    the direct kernels' to their rounding, the same picks, bit-identical from run to run; ReLU outputs (half the elements
    exactly zero: counted by ballot), GELU outputs, a tensor with specials (NaN wins), and a small one against the oracle."""
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(65)
    signed = [L.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
    unsigned = [L.plan_for(grids.ant_grid(t, 4, False)) for t in ("int", "pot", "flint")]
    rt = _ratios(80, 150, 1, dev)

    def both(x, plans, knob_on):
        n = x.numel()
        xm = L.absmax(x, 1, n, per_row=False)
        res = []
        for knob in (0, knob_on):
            L.lib().antq_debug_set(19, knob); L.lib().antq_debug_set(20, 0)
            try:
                s = L.search_sse_multi(x, 1, n, xm, False, rt, plans, [10.0] * 3)
                if s is None:
                    s = torch.stack([L.search_sse(x, 1, n, xm, False, rt, p, 10.0) for p in plans])
                res.append(s.clone())
            finally:
                L.lib().antq_debug_set(19, 1); L.lib().antq_debug_set(20, 1)
        return res

    for x, plans in ((torch.nn.functional.gelu(torch.randn(1 << 22, device=dev)), signed),
                     (torch.relu(torch.randn((1 << 22) + 4096, device=dev)), unsigned)):
        a, b = both(x, plans, 1)                                  # the default rule takes tensors of 4 M elements and more
        _compare(a, b, "one scale, default rule", rtol=2e-7)
        assert not torch.equal(a, b), "the sweep did not run: the comparison would prove nothing"
        assert torch.equal(b, both(x, plans, 1)[1])               # the same bits on every run
    x = torch.randn(70000 * 4, device=dev) * 0.3
    a, b = both(x, signed, 2)                                     # forced onto a small tensor
    _compare(a, b, "one scale, forced", rtol=2e-7)
    xn = x.cpu().numpy().reshape(1, -1)
    xm = np.float32([np.abs(xn).max()])
    for t, name in enumerate(("int", "pot", "flint")):
        best, alpha, trace = oracle.search_mse(xn, xm, 80, 150, 1, grids.ant_grid(name, 4, True), 10.0, ovp=False, per_row=False)
        np.testing.assert_allclose((b[t] / x.numel()).cpu().numpy().reshape(-1), trace.reshape(-1), rtol=3e-6)
    x[12345] = float("nan")
    a, b = both(x, signed, 2)
    assert torch.isnan(a).all() and torch.isnan(b).all()
