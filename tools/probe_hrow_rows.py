"""Same-box A/B of the 16-bit-domain row kernels (K1h) against the round-3 fp32-domain row table (knob 9 = 0) over row lengths,
static and dynamic, batched: ~1 GiB of bf16 per case.   python tools/probe_hrow_rows.py [olive]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
olive = len(sys.argv) > 1 and sys.argv[1] == "olive"
if olive:
    gn = grids.olive_grid("flint", 4, True)
    plan, gmax = _lib.plan_for(np.concatenate([gn, grids.olive_outliers(4, True)])), float(gn.max())
else:
    plan, gmax = _lib.plan_for(grids.ant_flint(4, True)), 10.0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def bench(b):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): b.run()
        torch.cuda.synchronize()
    e0.record()
    for _ in range(30): b.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / 30

for K in (1024, 1152, 1536, 2048, 2304, 3072, 4096, 4608, 8192, 11008, 16384, 28672):
    rows = (1 << 29) // (16 * K) // 2 * 2
    xs = [(torch.randn(rows, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, rows, K) for x in xs]
    line = "K=%6d rows=%6d x16 " % (K, rows)
    nbytes = 16 * rows * K * 4
    for dyn in (False, True):
        res = []
        for h in (1, 0):
            knob(9, h)
            try:
                b = _lib.Batch([(x, o, (a if not dyn else torch.empty_like(a)), plan, gmax, rows, K, True) for x, o, a in zip(xs, outs, al)], ovp=olive, dynamic=dyn)
                res.append(nbytes / bench(b) / 8e12 * 100)
            except Exception as ex:
                res.append(float("nan"))
            knob(9, 1)
        line += " | %s: K1h %5.1f %%  round-3 %5.1f %%" % ("dynamic" if dyn else "static ", res[0], res[1])
    print(line, flush=True)
    del xs, outs, al
