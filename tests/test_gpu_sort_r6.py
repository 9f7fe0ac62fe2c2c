"""Round 6: the sorted-row clip search (csrc/antq_k_sortsearch.h) against the direct kernels and the oracle.

A clip search scores every candidate by the squared error of the whole row (AQ:287-326, OQ:189-233), for every candidate
codebook of a type selection (AQ:328-415, OQ:235-256); the sorted-row kernel sorts the row once and reads every (codebook,
candidate, threshold) count and sum off its prefix sums.  Bar: the same PICK for every row and codebook (certified by the
direct kernels' own scores where they tie), sums equal to the direct kernels' to their rounding (<= 5e-7: the direct
kernels round each squared term to fp32 as the reference does, the closed form does not), the same NaN pattern; literal
elements (NaN / Inf / far-clipped) take the reference sequence; OliVe's pair rule through the victim corrections.
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


def _search(L, x, rows, K, xm, per_row, rt, plans, gmaxs, ovp):
    s = L.search_sse_multi(x, rows, K, xm, per_row, rt, plans, gmaxs, ovp=ovp) if len(plans) > 1 else None
    if s is None:
        s = torch.stack([L.search_sse(x, rows, K, xm, per_row, rt, p, g, ovp=ovp) for p, g in zip(plans, gmaxs)])
    return s.clone()


def _both(L, x, rows, K, xm, rt, plans, gmaxs, ovp, per_row=True, on=2):
    """(direct kernels, sorted-row search)"""
    out = []
    for k19, k20 in ((0, 0), (1, on)):
        L.lib().antq_debug_set(19, k19)
        L.lib().antq_debug_set(20, k20)
        try:
            out.append(_search(L, x, rows, K, xm, per_row, rt, plans, gmaxs, ovp))
        finally:
            L.lib().antq_debug_set(19, 1)
            L.lib().antq_debug_set(20, 1)
    return out


def _compare(a, b, what, rtol=5e-7, tie=2e-7):
    an, bn = torch.isnan(a), torch.isnan(b)
    assert torch.equal(an, bn), what
    ok = ~an & torch.isfinite(a)
    rel = ((a - b).abs() / a.abs().clamp_min(1e-300))[ok]
    assert rel.numel() == 0 or float(rel.max()) <= rtol, (what, float(rel.max()))
    fa = torch.where(torch.isnan(a), torch.full_like(a, float("inf")), a)
    fb = torch.where(torch.isnan(b), torch.full_like(b, float("inf")), b)
    pa, pb = fa.argmin(1), fb.argmin(1)
    diff = (pa != pb)
    if diff.any():        # only where the direct kernel's own two scores tie to its rounding
        ia, ib = pa[diff], pb[diff]
        t, r = diff.nonzero(as_tuple=True)
        gap = (a[t, ib, r] - a[t, ia, r]).abs() / a[t, ia, r].abs()
        assert float(gap.max()) <= tie, (what, float(gap.max()))


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16", "float16"])
def test_sorted_equals_direct_ant_codebooks(dev, dtype_name):
    from ant_quantization_amd import _lib as L, grids
    dt = getattr(torch, dtype_name)
    torch.manual_seed(71)
    types = ("int", "pot", "flint", "float")
    plans = [L.plan_for(grids.ant_grid(t, 4, True)) for t in types]
    # one chunk, a ragged second chunk, short rows, rows of several chunks, a row that is no multiple of anything
    for rows, K, lb, ub in ((96, 4096, 80, 150), (33, 4096 + 64, 75, 150), (64, 512, 95, 101), (70, 128, 75, 150), (17, 3 * 4096, 75, 76), (40, 264, 75, 150),
                            (9, 11008, 60, 150)):
        x = (torch.randn(rows, K, device=dev) * 0.03)
        x[::7] *= 0.2
        x = x.to(dt)
        xm = L.absmax(x, rows, K)
        a, b = _both(L, x, rows, K, xm, _ratios(lb, ub, 1, dev), plans, [10.0] * 4, False)
        assert not torch.equal(a, b), "the sorted search did not run: the comparison would prove nothing"
        _compare(a, b, (dtype_name, rows, K))
    # one type at a time (antq_search_sse): the same bits as the single-read type selection
    x = (torch.randn(48, 4096, device=dev) * 0.03).to(dt)
    xm = L.absmax(x, 48, 4096)
    rt = _ratios(75, 150, 1, dev)
    _, multi = _both(L, x, 48, 4096, xm, rt, plans[:3], [10.0] * 3, False)
    for t in range(3):
        _, one = _both(L, x, 48, 4096, xm, rt, [plans[t]], [10.0], False)
        assert torch.equal(one[0], multi[t]), "a codebook's sums depend on its company"
    # unsigned codebooks on a non-negative tensor (half the elements exactly zero)
    pu = [L.plan_for(grids.ant_grid(t, 4, False)) for t in ("int", "flint")]
    x = torch.nn.functional.relu(torch.randn(48, 3072, device=dev)).to(dt)
    xm = L.absmax(x, 48, 3072)
    a, b = _both(L, x, 48, 3072, xm, rt, pu, [10.0] * 2, False)
    _compare(a, b, (dtype_name, "unsigned"))
    # rows of <= 1024 elements run one row per wavefront (k_search_sorted_short); knob 21 = 0 sends them through the 4096-key
    # kernel instead: the same sums to the closed form's rounding (the sum of x^2 is grouped differently), the same picks
    x = (torch.randn(200, 768, device=dev) * 0.03).to(dt)
    xm = L.absmax(x, 200, 768)
    _, short = _both(L, x, 200, 768, xm, rt, plans[:3], [10.0] * 3, False)
    L.lib().antq_debug_set(21, 0)
    try:
        _, long_ = _both(L, x, 200, 768, xm, rt, plans[:3], [10.0] * 3, False)
    finally:
        L.lib().antq_debug_set(21, 1)
    torch.testing.assert_close(short, long_, rtol=1e-13, atol=0)      # (16-bit inputs: sums of exact squares -- to the bit)
    # a ratio list that starts at 0.3: each row's largest elements lie beyond the step function's domain for the small scales
    # (a dozen per row: the short list behind the sorted keys; knob 21 = 0: the 4096-key kernel's list)
    for k21 in (1, 0):
        L.lib().antq_debug_set(21, k21)
        try:
            a, b = _both(L, x, 200, 768, xm, _ratios(30, 150, 1, dev), plans[:3], [10.0] * 3, False)
        finally:
            L.lib().antq_debug_set(21, 1)
        _compare(a, b, (dtype_name, "ratios from 0.3, rows of 768", k21))
    # a long candidate list goes out in pieces
    a, b = _both(L, x, 48, 3072, xm, _ratios(20, 300, 1, dev), pu, [10.0] * 2, False)
    _compare(a, b, (dtype_name, "280 candidates"))


def test_sorted_literal_elements_and_unusable_rows(dev):
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(72)
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
    x[10, 3] = -0.0
    xm = L.absmax(x, 24, 1024)
    xm[8] = 0.05
    a, b = _both(L, x, 24, 1024, xm, _ratios(75, 150, 1, dev), plans, [10.0, 10.0], False)
    _compare(a, b, "edge rows")
    assert torch.isnan(b[:, :, 1]).all() and torch.isnan(b[:, :, 2]).all() and torch.isnan(b[:, :, 3]).all()
    # a ratio list that is not ascending: the kernel notices and evaluates literally
    rt = _ratios(75, 150, 1, dev).flip(0).contiguous()
    a, b = _both(L, x[9:], 15, 1024, xm[9:].contiguous(), rt, plans, [10.0, 10.0], False)
    _compare(a, b, "descending ratios")
    # ... and one that ascends irregularly
    rt = torch.tensor([0.5, 0.51, 0.7, 0.71, 0.72, 0.9, 1.3, 1.31, 2.0], device=dev)
    a, b = _both(L, x[9:], 15, 1024, xm[9:].contiguous(), rt, plans, [10.0, 10.0], False)
    _compare(a, b, "irregular ratios")


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_sorted_olive_pairs_against_direct_and_oracle(dev, oracle, dtype_name):
    """OliVe codebooks with planted outliers: victims through the correction list (OQ:311-320), also where BOTH members of a
    pair are outliers and where an outlier meets a far-clipped element."""
    from ant_quantization_amd import _lib as L, grids
    dt = getattr(torch, dtype_name)
    torch.manual_seed(73)
    oo = grids.olive_outliers(4, True)
    cb = [(np.concatenate([grids.olive_grid(t, 4, True), oo]), float(grids.olive_grid(t, 4, True).max())) for t in ("int", "flint")]
    plans, gm = [L.plan_for(g) for g, _ in cb], [m for _, m in cb]
    # (rows of <= 1024 elements: one row per wavefront, the outlier-capable pairs in a 32-pair list -- or, the 20 x 1024 case
    #  with an outlier in every fourth pair, read again from the row)
    for rows, K in ((40, 2048), (12, 4096 + 512), (6, 3 * 4096), (64, 768), (50, 1024), (33, 576), (20, 1024)):
        x = torch.randn(rows, K, device=dev) * 0.02
        idx = torch.randint(0, x.numel(), (x.numel() // 300,), device=dev)
        x.view(-1)[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
        x[0, 10:14] = torch.tensor([0.9, -1.1, 0.8, 0.7], device=dev)          # both members outliers
        x[1, 20] = 3.0e4                                                          # far beyond the codebook: a literal pair
        if (rows, K) == (20, 1024):
            x[:, ::8] *= 12.0                                                     # an outlier in every fourth pair: the list overflows
        x = x.to(dt)
        xm = L.xmax_3sigma(x, rows, K, per_row=True)
        rt = _ratios(75, 250, 2, dev)
        for ovp in (True, False):
            a, b = _both(L, x, rows, K, xm, rt, plans, gm, ovp)
            assert not torch.equal(a, b)
            # (rows with a planted 0.9 next to 0.02-sized elements: the direct kernels add a lane's squared terms in fp32 --
            #  noise of the size of the reference's own reduction noise, tests/calib_check.py -- the closed form does not)
            _compare(a, b, (dtype_name, "olive", ovp, rows, K), rtol=1.2e-6, tie=5.9e-7)
    # a few rows against the oracle's own search (the reference's op sequence on the fp32 image of the tensor)
    for rows, K in ((6, 2048), (6, 768)):
        _oracle_rows(L, oracle, dev, dt, cb, plans, gm, rows, K)


def _oracle_rows(L, oracle, dev, dt, cb, plans, gm, rows, K):
    import torch
    x = torch.randn(rows, K, device=dev) * 0.02
    idx = torch.randint(0, x.numel(), (x.numel() // 300,), device=dev)
    x.view(-1)[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
    x = x.to(dt)
    xm = L.xmax_3sigma(x, rows, K, per_row=True)
    rt = _ratios(75, 250, 2, dev)
    _, s = _both(L, x, rows, K, xm, rt, plans, gm, True)
    xn = x.float().cpu().numpy()
    for t, (g, m) in enumerate(cb):
        best, alpha, trace = oracle.search_mse(xn, xm.cpu().numpy(), 75, 250, 2, g, m, ovp=True, per_row=True)
        got = (s[t] / K).cpu().numpy()                       # [ncand, rows] mean squared error
        np.testing.assert_allclose(got, trace, rtol=3e-6)
        assert np.array_equal(got.argmin(0), trace.argmin(0)) or np.allclose(np.take_along_axis(trace, got.argmin(0)[None], 0), trace.min(0), rtol=6e-7)


def test_calibrate_picks_do_not_depend_on_the_sorted_search(dev):
    """antq_calibrate (x_max, every codebook's search, per-row picks, the type pick) with the sorted search under its default
    rule, forced everywhere, and off (the direct kernels): the same alphas, scores to rounding and the same type."""
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(74)
    plans = [L.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
    x = torch.distributions.Laplace(0.0, 0.02).sample((128, 4096)).to(dev)
    res = []
    for k19, k20 in ((0, 0), (1, 1), (1, 2)):
        L.lib().antq_debug_set(19, k19)
        L.lib().antq_debug_set(20, k20)
        try:
            alpha, score, typ, xmax = L.calibrate(x, 128, 4096, True, plans, [10.0] * 3, 75, 150, 1, xmax="absmax")
            res.append((alpha.clone(), score.clone(), int(typ)))
        finally:
            L.lib().antq_debug_set(19, 1)
            L.lib().antq_debug_set(20, 1)
    (a0, s0, t0), (a1, s1, t1), (a2, s2, t2) = res
    assert t0 == t1 == t2
    assert torch.equal(a1, a2) and torch.equal(s1, s2)                  # default rule == forced on these rows
    same = (a0 == a1)
    assert float(same.float().mean()) >= 0.995                          # (a differing row: a tie of the direct kernel's own scores)
    assert torch.allclose(s0, s1, rtol=1e-6)


def test_sorted_one_scale_fp32_tensors(dev, oracle):
    """A tensor with ONE scale (every activation quantiser, AQ:51-53, :308-324) in fp32 -- the only dtype the reference itself
    runs -- chunk by chunk over many workgroups (slabs of partial terms added in slab order): sums equal to the direct
    kernels' to their rounding, the same picks, bit-identical from run to run; ReLU and GELU outputs, a ragged size, a tensor
    with specials (NaN wins), and a small one against the oracle."""
    from ant_quantization_amd import _lib as L, grids
    torch.manual_seed(75)
    signed = [L.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
    unsigned = [L.plan_for(grids.ant_grid(t, 4, False)) for t in ("int", "pot", "flint")]
    rt = _ratios(80, 150, 1, dev)

    def both(x, plans, on):
        n = x.numel()
        xm = L.absmax(x, 1, n, per_row=False)
        return _both(L, x, 1, n, xm, rt, plans, [10.0] * 3, False, per_row=False, on=on)

    for x, plans in ((torch.nn.functional.gelu(torch.randn(1 << 22, device=dev)), signed),
                     (torch.relu(torch.randn((1 << 22) + 4096 + 8, device=dev)), unsigned)):
        a, b = both(x, plans, 1)                                  # the default rule takes tensors of 1 M elements and more
        _compare(a, b, "one scale, default rule", rtol=2e-7)
        assert not torch.equal(a, b), "the sorted search did not run: the comparison would prove nothing"
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
    # OliVe's codebooks with the pair rule on a tensor with one scale (3-sigma statistic, planted outliers)
    oo = grids.olive_outliers(4, True)
    cb = [(np.concatenate([grids.olive_grid(t, 4, True), oo]), float(grids.olive_grid(t, 4, True).max())) for t in ("int", "flint")]
    plans, gm = [L.plan_for(g) for g, _ in cb], [m for _, m in cb]
    x = torch.randn((1 << 20) + 4096 * 3 + 16, device=dev) * 0.02
    idx = torch.randint(0, x.numel(), (x.numel() // 300,), device=dev)
    x[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
    xm = L.xmax_3sigma(x, 1, x.numel(), per_row=False)
    rt2 = _ratios(75, 250, 2, dev)
    for ovp in (True, False):
        a, b = _both(L, x, 1, x.numel(), xm, rt2, plans, gm, ovp, per_row=False, on=1)
        assert not torch.equal(a, b)
        _compare(a, b, ("one scale, olive", ovp), rtol=1.2e-6, tie=5.9e-7)
