import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
nb = 16
for layout in ("interleaved", "grouped", "pool", "pool2", "interleaved"):
    torch.cuda.empty_cache()
    xs, outs, al = [], [], []
    if layout == "interleaved":
        for i in range(nb):
            x = (torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16); xs.append(x)
            al.append(_lib.absmax(x, 4096, 4096)); outs.append(torch.empty_like(x))
    elif layout in ("pool", "pool2"):
        n1 = 4096 * 4096
        if layout == "pool":
            pool = torch.empty(2 * nb * n1, device=dev, dtype=torch.bfloat16)
            px, po = pool[: nb * n1], pool[nb * n1:]
        else:
            px = torch.empty(nb * n1, device=dev, dtype=torch.bfloat16); po = torch.empty(nb * n1, device=dev, dtype=torch.bfloat16)
        for i in range(nb):
            x = px[i * n1:(i + 1) * n1].view(4096, 4096); x.copy_((torch.randn(4096, 4096, device=dev) * 0.02)); xs.append(x)
            al.append(_lib.absmax(x, 4096, 4096)); outs.append(po[i * n1:(i + 1) * n1].view(4096, 4096))
    else:
        for i in range(nb):
            xs.append((torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16))
        for i in range(nb): al.append(_lib.absmax(xs[i], 4096, 4096))
        for i in range(nb): outs.append(torch.empty_like(xs[i]))
    d = [(o.data_ptr() - x.data_ptr()) / 2**20 for x, o in zip(xs, outs)]
    bt = _lib.Batch([(x, o, a, plan, 10.0, 4096, 4096, True) for x, a, o in zip(xs, al, outs)])
    t = timed(bt.run)
    tp = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xs, al, outs)], 10)
    tc = timed(lambda: [_lib.copy(x, o) for x, o in zip(xs, outs)], 10)
    print("%-12s out-x offset MiB %s..: batched %.1f us (%.1f%%)  per-tensor %.1f us/launch (%.1f%%)  copy %.1f us/launch" % (
        layout, ["%.1f" % v for v in d[:3]], t, nb * 67.108864e6 / (t * 1e-6) / 8e10, tp / nb, 67.108864e6 / (tp / nb * 1e-6) / 8e10, tc / nb))
    del xs, outs, al, bt
    px = po = pool = None
