import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib as L, grids
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
plans = [L.plan_for(grids.ant_flint(4, True)), L.plan_for(grids.ant_int(4, True))]
x = (rng.standard_normal((64, 1024)) * 0.05).astype(np.float32)
xt = torch.from_numpy(x).to(dev)
xm = L.absmax(xt, 64, 1024, per_row=True).reshape(-1)
four = [plans[0], plans[1], L.plan_for(grids.ant_pot(4, True)), L.plan_for(grids.ant_int(4, False))]
gm = [10.0, 7.0, 10.0, 15.0]
for lo, hi in ((75, 250), (75, 163), (75, 120), (163, 250)):
    r = torch.from_numpy(np.float32([np.float32(i * 0.01) for i in range(lo, hi)])).to(dev)
    for sel in ((0, 1, 2, 3), (0, 1), (0, 3), (0, 2), (1, 3)):
        multi = L.search_sse_multi(xt, 64, 1024, xm, True, r, [four[i] for i in sel], [gm[i] for i in sel])
        for j, t in enumerate(sel):
            one = L.search_sse(xt, 64, 1024, xm, True, r, four[t], gm[t])
            d = (multi[j] != one)
            if d.any():
                rel = ((multi[j] - one).abs() / one.abs())
                c, rr = np.unravel_index(int(rel.argmax().cpu()), rel.shape)
                print("ratios %d..%d types %s: type %d differs in %d / %d entries, max rel %.2e at cand %d row %d; cands with a difference: %s" % (
                    lo, hi, sel, t, int(d.sum()), d.numel(), float(rel.max()), c, rr, sorted(set(d.nonzero()[:, 0].tolist()))[:12]))
            else:
                print("ratios %d..%d types %s: type %d equal" % (lo, hi, sel, t))
