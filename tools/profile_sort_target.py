"""Launches of the sorted-row clip search for a counter pass (tools/profile_sort.sh): ANT 3 codebooks x 70 candidates and OliVe
2 codebooks x 88 candidates with the pair rule on 4096 x 4096 fp32, and a 25 M-element fp32 tensor with one scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
torch.manual_seed(0)
def ratios(lb, ub, step):
    return torch.tensor([np.float32(i * 0.01) for i in range(lb, ub, step)], dtype=torch.float32, device=dev)
ant = [(_lib.plan_for(grids.ant_grid(t, 4, True)), 10.0) for t in ("int", "pot", "flint")]
on, oo = grids.olive_grid("int", 4, True), grids.olive_outliers(4, True)
fn = grids.olive_grid("flint", 4, True)
oli = [(_lib.plan_for(np.concatenate([on, oo])), float(on.max())), (_lib.plan_for(np.concatenate([fn, oo])), float(fn.max()))]
x = torch.randn(4096, 4096, device=dev) * 0.02
xm = _lib.absmax(x, 4096, 4096)
xo = x.clone()
idx = torch.arange(xo.numel() // 1000, device=dev) * 1000
xo.view(-1)[idx] *= 30.0
xm3 = _lib.xmax_3sigma(xo, 4096, 4096, per_row=True)
a = torch.nn.functional.gelu(torch.randn(64 * 128 * 3072, device=dev))
am = _lib.absmax(a, 1, a.numel(), per_row=False)
xs = torch.randn(16384, 768, device=dev) * 0.02
xsm = _lib.absmax(xs, 16384, 768)
for _ in range(3):
    _lib.search_sse_multi(xs, 16384, 768, xsm, True, ratios(80, 150, 1), [p for p, _ in ant], [g for _, g in ant])
    _lib.search_sse_multi(x, 4096, 4096, xm, True, ratios(80, 150, 1), [p for p, _ in ant], [g for _, g in ant])
    _lib.search_sse_multi(xo, 4096, 4096, xm3, True, ratios(75, 250, 2), [p for p, _ in oli], [g for _, g in oli], ovp=True)
    _lib.search_sse_multi(a, 1, a.numel(), am, False, ratios(80, 150, 1), [p for p, _ in ant], [g for _, g in ant])
torch.cuda.synchronize()
