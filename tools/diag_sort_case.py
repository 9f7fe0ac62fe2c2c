"""Where the sorted-row search and the direct kernels differ most on the OliVe test tensor: both against the exact (float64)
sum of the reference's per-element outputs (oracle.forward), with and without the reference's fp32 rounding of each term."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib as L, grids
from oracle import antq_oracle as orc
orc.build(); orc.lib()
dev = torch.device("cuda:0")
torch.manual_seed(73)
oo = grids.olive_outliers(4, True)
cb = [(np.concatenate([grids.olive_grid(t, 4, True), oo]), float(grids.olive_grid(t, 4, True).max())) for t in ("int", "flint")]
plans, gm = [L.plan_for(g) for g, _ in cb], [m for _, m in cb]
rt = torch.tensor([np.float32(i * 0.01) for i in range(75, 250, 2)], dtype=torch.float32, device=dev)
for dt in (torch.float32, torch.bfloat16):
  for rows, K in ((40, 2048), (12, 4096 + 512), (6, 3 * 4096)):
    x = torch.randn(rows, K, device=dev) * 0.02
    idx = torch.randint(0, x.numel(), (x.numel() // 300,), device=dev)
    x.view(-1)[idx] *= torch.empty(idx.numel(), device=dev).uniform_(8, 64)
    x[0, 10:14] = torch.tensor([0.9, -1.1, 0.8, 0.7], device=dev)
    x[1, 20] = 3.0e4
    x = x.to(dt)
    xm = L.xmax_3sigma(x, rows, K, per_row=True)
    for ovp in (True, False):
        res = []
        for k19, k20 in ((0, 0), (1, 2)):
            L.lib().antq_debug_set(19, k19); L.lib().antq_debug_set(20, k20)
            s = L.search_sse_multi(x, rows, K, xm, True, rt, plans, gm, ovp=ovp)
            if s is None:
                s = torch.stack([L.search_sse(x, rows, K, xm, True, rt, p, g, ovp=ovp) for p, g in zip(plans, gm)])
            res.append(s.clone())
        L.lib().antq_debug_set(19, 1); L.lib().antq_debug_set(20, 1)
        a, b = res
        rel = ((a - b).abs() / a.abs())
        rel = torch.where(torch.isfinite(rel), rel, torch.zeros_like(rel))
        t, c, r = np.unravel_index(int(rel.argmax()), rel.shape)
        xn = x[r:r + 1].float().cpu().numpy()
        alpha = (xm[r:r + 1] * rt[c]).cpu().numpy().astype(np.float32).reshape(1, 1)
        out = orc.forward(xn, alpha, cb[t][0], cb[t][1], ovp=ovp, want_idx=False)
        out = out[0] if isinstance(out, tuple) else out
        d32 = (out.astype(np.float32) - xn.astype(np.float32)).astype(np.float32)
        exact = float((d32.astype(np.float64) ** 2).sum())
        rounded = float(((d32 * d32).astype(np.float32)).astype(np.float64).sum())
        print("%s %dx%d ovp %d: worst rel %.2e at type %d cand %d row %d: direct %.12e sorted %.12e | exact-sum %.12e (direct %+.2e sorted %+.2e)  fp32-terms %.12e (direct %+.2e sorted %+.2e)" % (
            str(dt)[6:], rows, K, ovp, float(rel.max()), t, c, r, float(a[t, c, r]), float(b[t, c, r]), exact, float(a[t, c, r]) / exact - 1, float(b[t, c, r]) / exact - 1,
            rounded, float(a[t, c, r]) / rounded - 1, float(b[t, c, r]) / rounded - 1), flush=True)
