import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from ant_quantization_amd import _lib, core, grids
dev = torch.device("cuda:0")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for shape in ((64,128,768),(64,128,3072)):
    x = torch.nn.functional.gelu(torch.randn(*shape, device=dev))
    for t in ("int","pot","flint"):
        plan = _lib.plan_for(grids.ant_grid(t, 4, True))
        xm = core.row_absmax(x, False)
        ms = timed(lambda: core.clip_search(x, xm, False, 75, 150, 1, plan, 10.0))
        print(shape, t, "per-tensor search %.3f ms  %.0f G evals/s" % (ms, x.numel()*75/ms/1e6))
    ms = timed(lambda: core.row_absmax(x, False))
    print(shape, "absmax per-tensor %.3f ms" % ms)
    ms = timed(lambda: x.min())
    print(shape, "torch min %.3f ms" % ms)
w = torch.randn(3072, 768, device=dev)*0.02
plan = _lib.plan_for(grids.ant_grid("flint", 4, True))
xm = core.row_absmax(w, True)
ms = timed(lambda: core.clip_search(w, xm, True, 75, 150, 1, plan, 10.0))
print("W[3072,768] per-row search %.3f ms %.0f G evals/s" % (ms, w.numel()*75/ms/1e6))
