import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, (t1 - t0) * 1e6 / reps
for shp in [(64, 3, 7, 7), (64, 64, 1, 1), (256, 64, 1, 1), (512, 512, 3, 3), (2048, 1024, 1, 1), (1000, 2048)]:
    w = torch.randn(*shp, device=dev) * 0.05
    o = torch.empty_like(w)
    n = w.numel()
    a_pc = _lib.absmax(w, shp[0], n // shp[0]); a16 = _lib.absmax(w, n // 16, 16)
    g, h = timed(lambda: _lib.fakequant(w, a_pc, plan, 10.0, shp[0], n // shp[0], True, out=o))
    g2, h2 = timed(lambda: _lib.fakequant(w, a16, plan, 10.0, n // 16, 16, True, out=o))
    g3, h3 = timed(lambda: _lib.fakequant_dynamic(w, plan, 10.0, n // 16, 16, out=o, want_alpha=False))
    print("%-20s n=%8d  per-channel gpu %.1f us host %.1f us | group16 static gpu %.1f host %.1f | group16 dyn gpu %.1f host %.1f" % (shp, n, g, h, g2, h2, g3, h3))
