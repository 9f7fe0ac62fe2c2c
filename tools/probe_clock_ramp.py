import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
nb = 32
xs, outs, al = [], [], []
for i in range(nb):
    x = (torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16); xs.append(x)
    al.append(_lib.absmax(x, 4096, 4096)); outs.append(torch.empty_like(x))
bt = _lib.Batch([(x, o, a, plan, 10.0, 4096, 4096, True) for x, a, o in zip(xs, al, outs)])
torch.cuda.synchronize()
t00 = time.perf_counter()
for rnd in range(14):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if rnd < 10 else 200
    e0.record()
    for _ in range(reps): bt.run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / reps
    print("t=%.2fs round %d: batched %.1f us  %.1f%%" % (time.perf_counter() - t00, rnd, t, nb * 67.108864e6 / (t * 1e-6) / 8e10))
