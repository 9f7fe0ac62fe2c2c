"""The sorted-row clip search (antq_k_sortsearch.h, knob 20) against the direct kernels (knobs 19 = 20 = 0) and the threshold
sweep (knob 19 = 1, knob 20 = 0): largest relative difference of the sums, whether every row's pick agrees, and the time of
the search -- ANT three codebooks x 70 candidates and OliVe two codebooks x 88 candidates with the pair rule (0.1 % planted
outliers), fp32 and bf16, per-row scales; tensors with ONE scale in fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
L = _lib.lib()
QUICK = "--quick" in sys.argv


def ratios(lb, ub, step):
    return torch.tensor([np.float32(i * 0.01) for i in range(lb, ub, step)], dtype=torch.float32, device=dev)


def search(x, rows, K, xm, per_row, rt, plans, gmaxs, ovp):
    s = _lib.search_sse_multi(x, rows, K, xm, per_row, rt, plans, gmaxs, ovp=ovp) if len(plans) > 1 else None
    if s is None:
        s = torch.stack([_lib.search_sse(x, rows, K, xm, per_row, rt, p, g, ovp=ovp) for p, g in zip(plans, gmaxs)])
    return s


MODES = (("direct", 0, 0), ("sweep", 1, 0), ("sorted", 1, 2))
if "--long" in sys.argv:          # rows of <= 1024 elements through the 4096-key kernel (knob 21 = 0) instead of one row per wavefront
    L.antq_debug_set(21, 0)


def run(name, x, rows, K, plans, gmaxs, xm, rt, ovp, per_row=True):
    out = {}
    for mode, k19, k20 in MODES:
        L.antq_debug_set(19, k19); L.antq_debug_set(20, k20)
        s = search(x, rows, K, xm, per_row, rt, plans, gmaxs, ovp)
        torch.cuda.synchronize()
        t = timed(lambda: search(x, rows, K, xm, per_row, rt, plans, gmaxs, ovp), 3)
        out[mode] = (s.clone(), t)
    L.antq_debug_set(19, 1); L.antq_debug_set(20, 1)
    a, b = out["direct"][0], out["sorted"][0]
    rel = ((a - b).abs() / a.abs().clamp_min(1e-300))
    rel = torch.where(torch.isfinite(rel), rel, torch.zeros_like(rel))
    pa, pb = a.argmin(1), b.argmin(1)
    nan_same = bool((torch.isnan(a) == torch.isnan(b)).all())
    print("%-66s direct %8.3f  sweep %8.3f  sorted %8.3f ms (x %.1f)  max rel diff %.2e  picks differing %d / %d  nan same %s" % (
        name, out["direct"][1] * 1e3, out["sweep"][1] * 1e3, out["sorted"][1] * 1e3, out["direct"][1] / out["sorted"][1], float(rel.max()),
        int((pa != pb).sum()), pa.numel(), nan_same), flush=True)


torch.manual_seed(0)
SCAN = "--scan" in sys.argv
ant = [(_lib.plan_for(grids.ant_grid(t, 4, True)), 10.0) for t in ("int", "pot", "flint")]
on, oo = grids.olive_grid("int", 4, True), grids.olive_outliers(4, True)
fn = grids.olive_grid("flint", 4, True)
oli = [(_lib.plan_for(np.concatenate([on, oo])), float(on.max())), (_lib.plan_for(np.concatenate([fn, oo])), float(fn.max()))]
if SCAN:      # where does the sorted search start to pay?  rows of K elements, ~4 M elements per tensor
    for dt in (torch.float32, torch.bfloat16):
        for K in (128, 256, 512, 576, 768, 1024, 1152, 1536, 2048, 2304, 3072, 4096, 4608, 8192):
            rows = max(64, (1 << 22) // K)
            x = (torch.randn(rows, K, device=dev) * 0.02).to(dt)
            xm = _lib.absmax(x, rows, K)
            run("scan ANT int/pot/flint x 75, %d x %d %s" % (rows, K, str(dt)[6:]), x, rows, K, [p for p, _ in ant], [g for _, g in ant], xm, ratios(75, 150, 1), False)
            if K >= 256 and dt == torch.float32:
                xo = x.clone()
                idx = torch.arange(xo.numel() // 1000, device=dev) * 1000
                xo.view(-1)[idx] *= 30.0
                xm3 = _lib.xmax_3sigma(xo, rows, K, per_row=True)
                run("scan OliVe int/flint x 88 pairs, %d x %d %s" % (rows, K, str(dt)[6:]), xo, rows, K, [p for p, _ in oli], [g for _, g in oli], xm3, ratios(75, 250, 2), True)
    sys.exit(0)
for dt in (torch.float32, torch.bfloat16):
    nm = str(dt)[6:]
    for rows, K in ((4096, 4096), (768, 3072), (3072, 768), (16384, 4096), (4096, 16384), (4096, 11008))[: 3 if QUICK else 6]:
        x = (torch.randn(rows, K, device=dev) * 0.02).to(dt)
        xm = _lib.absmax(x, rows, K)
        run("ANT int/pot/flint x 70, %d x %d %s" % (rows, K, nm), x, rows, K, [p for p, _ in ant], [g for _, g in ant], xm, ratios(80, 150, 1), False)
    for rows, K in ((4096, 4096), (16384, 4096), (4096, 16384))[: 1 if QUICK else 3]:
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

print("one scale per tensor, fp32:")
pu = [(_lib.plan_for(grids.ant_grid(t, 4, False)), 10.0) for t in ("int", "pot", "flint")]
for nelem, signed in ((64 * 128 * 3072, True), (64 * 128 * 768, True), (64 * 128 * 3072, False), (1 << 20, True)):
    x = torch.nn.functional.gelu(torch.randn(nelem, device=dev)) if signed else torch.relu(torch.randn(nelem, device=dev))
    pl = ant if signed else pu
    xm = _lib.absmax(x, 1, nelem, per_row=False)
    run("ANT %s x 70, one scale, %d elements fp32" % ("int/pot/flint" if signed else "unsigned int/pot/flint", nelem), x, 1, nelem,
        [p for p, _ in pl], [g for _, g in pl], xm, ratios(80, 150, 1), False, per_row=False)
