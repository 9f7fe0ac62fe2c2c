"""The threshold-sweep clip search (antq_k_sweep.h) against the direct kernels (knob 19 = 0): largest relative difference of the
sums, whether every row's pick agrees, and the time of the search -- ANT three codebooks x 70 candidates and OliVe two
codebooks x 88 candidates with the pair rule (0.1 % planted outliers), fp32 and bf16, per-row scales."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
L = _lib.lib()


def ratios(lb, ub, step):
    return torch.tensor([np.float32(i * 0.01) for i in range(lb, ub, step)], dtype=torch.float32, device=dev)


def run(name, x, rows, K, plans, gmaxs, xm, rt, ovp):
    out = {}
    for knob in (0, 1):
        L.antq_debug_set(19, 2 * knob)
        s = _lib.search_sse_multi(x, rows, K, xm, True, rt, plans, gmaxs, ovp=ovp)
        if s is None:
            s = torch.stack([_lib.search_sse(x, rows, K, xm, True, rt, p, g, ovp=ovp) for p, g in zip(plans, gmaxs)])
        torch.cuda.synchronize()
        t = timed(lambda: _lib.search_sse_multi(x, rows, K, xm, True, rt, plans, gmaxs, ovp=ovp), 3)
        out[knob] = (s.clone(), t)
    L.antq_debug_set(19, 1)
    a, b = out[0][0], out[1][0]
    rel = ((a - b).abs() / a.abs().clamp_min(1e-300))
    rel = torch.where(torch.isfinite(rel), rel, torch.zeros_like(rel))
    pa, pb = a.argmin(1), b.argmin(1)
    nan_same = bool((torch.isnan(a) == torch.isnan(b)).all())
    print("%-64s direct %8.3f ms  sweep %8.3f ms  (x %.1f)  max rel diff %.2e  picks differing %d / %d  nan pattern same %s" % (
        name, out[0][1] * 1e3, out[1][1] * 1e3, out[0][1] / out[1][1], float(rel.max()), int((pa != pb).sum()), pa.numel(), nan_same), flush=True)


torch.manual_seed(0)
ant = [(_lib.plan_for(grids.ant_grid(t, 4, True)), 10.0) for t in ("int", "pot", "flint")]
on, oo = grids.olive_grid("int", 4, True), grids.olive_outliers(4, True)
fn = grids.olive_grid("flint", 4, True)
oli = [(_lib.plan_for(np.concatenate([on, oo])), float(on.max())), (_lib.plan_for(np.concatenate([fn, oo])), float(fn.max()))]
for dt in (torch.float32, torch.bfloat16):
    nm = str(dt)[6:]
    for rows, K in ((4096, 4096), (768, 3072), (3072, 768), (16384, 4096), (4096, 16384)):
        x = (torch.randn(rows, K, device=dev) * 0.02).to(dt)
        xm = _lib.absmax(x, rows, K)
        run("ANT int/pot/flint x 70, %d x %d %s" % (rows, K, nm), x, rows, K, [p for p, _ in ant], [g for _, g in ant], xm, ratios(80, 150, 1), False)
    for rows, K in ((4096, 4096), (16384, 4096), (4096, 16384)):
        x = torch.randn(rows, K, device=dev) * 0.02
        idx = torch.arange(x.numel() // 1000, device=dev) * 1000 + torch.randint(0, 1000, (x.numel() // 1000,), device=dev)
        x.view(-1)[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
        x = x.to(dt)
        xm = _lib.xmax_3sigma(x, rows, K, per_row=True)
        run("OliVe int/flint + outliers x 88, pairs, %d x %d %s" % (rows, K, nm), x, rows, K, [p for p, _ in oli], [g for _, g in oli], xm, ratios(75, 250, 2), True)
        run("OliVe flint + outliers x 88, NO pairs, %d x %d %s" % (rows, K, nm), x, rows, K, [oli[1][0]], [oli[1][1]], xm, ratios(75, 250, 2), False)
# edge rows: zeros, a NaN, an Inf, a huge element, constant rows
x = torch.randn(64, 1024, device=dev) * 0.02
x[0] = 0.0
x[1, 5] = float("nan")
x[2, 7] = float("inf")
x[3, 9] = 1e30
x[4] = 0.5
x[5, ::2] = 0.0
xm = _lib.absmax(x, 64, 1024)
run("edge rows (zeros / NaN / Inf / 1e30 / constant), fp32", x, 64, 1024, [p for p, _ in ant], [g for _, g in ant], xm, ratios(75, 150, 1), False)

# ---- one scale per tensor (activations), fp32: the sweep over many workgroups against the direct kernels
print("per-tensor fp32:")
pu = [(_lib.plan_for(grids.ant_grid(t, 4, False)), 10.0) for t in ("int", "pot", "flint")]
for nelem, signed in ((64 * 128 * 3072, True), (64 * 128 * 768, True), (64 * 128 * 3072, False), (1 << 20, True)):
    x = torch.nn.functional.gelu(torch.randn(nelem, device=dev)) if signed else torch.relu(torch.randn(nelem, device=dev))
    pl = ant if signed else pu
    xm = _lib.absmax(x, 1, nelem, per_row=False)
    out = {}
    for knob in (0, 1):
        L.antq_debug_set(19, knob)
        s = _lib.search_sse_multi(x, 1, nelem, xm, False, ratios(80, 150, 1), [p for p, _ in pl], [g for _, g in pl])
        if s is None:
            s = torch.stack([_lib.search_sse(x, 1, nelem, xm, False, ratios(80, 150, 1), p, g) for p, g in pl])
        t = timed(lambda: _lib.search_sse_multi(x, 1, nelem, xm, False, ratios(80, 150, 1), [p for p, _ in pl], [g for _, g in pl]), 3)
        out[knob] = (s.clone(), t)
    L.antq_debug_set(19, 1)
    a, b = out[0][0], out[1][0]
    rel = ((a - b).abs() / a.abs()).max()
    print("ANT %s x 70, one scale, %9d elements fp32   direct %8.3f ms  sweep %8.3f ms (x %.1f)  max rel diff %.2e  picks %s" % (
        "int/pot/flint" if signed else "unsigned int/pot/flint", nelem, out[0][1] * 1e3, out[1][1] * 1e3, out[0][1] / out[1][1], float(rel),
        "same" if torch.equal(a.argmin(1), b.argmin(1)) else "DIFFER"), flush=True)
