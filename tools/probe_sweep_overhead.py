import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
L = _lib.lib()
def ratios(lb, ub, step):
    return torch.tensor([np.float32(i * 0.01) for i in range(lb, ub, step)], dtype=torch.float32, device=dev)
p = _lib.plan_for(grids.ant_grid("flint", 4, True))
L.antq_debug_set(19, 2)
for rows, K in ((65536, 256), (16384, 1024), (4096, 4096), (1024, 16384)):
    x = torch.randn(rows, K, device=dev) * 0.02
    xm = _lib.absmax(x, rows, K)
    for nc in (4, 16, 70, 128):
        rt = ratios(80, 80 + nc, 1)
        t = timed(lambda: _lib.search_sse(x, rows, K, xm, True, rt, p, 10.0), 3)
        print("rows %6d K %6d ncand %3d: %8.3f ms  = %7.2f us per row  %6.1f ns per element" % (rows, K, nc, t * 1e3, t * 1e6 / rows * 2304 if False else t*1e6/rows, t * 1e9 / (rows * K)), flush=True)
