import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids, quant_cuda
dev = torch.device("cuda:0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import timed as _timed   # steady-clock timing (warm-up >= 60 ms, >= 40 ms measured)


def timed(fn, reps=10):
    return _timed(fn, reps)


xs = [torch.randn(4096 * 4096, device=dev) * 4 for _ in range(8)]
for name, g in (("flint4 (16)", grids.ant_flint(4, True)), ("olive flint4+outliers (29)", np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)])),
                ("int6 (64)", grids.ant_int(6, True)), ("int8 (256)", grids.ant_int(8, True))):
    gt = torch.from_numpy(g).to(dev)
    t = timed(lambda: [_lib.nearest(x, gt) for x in xs], 5) / len(xs)
    pl = _lib.plan_for(g)
    tp = timed(lambda: [_lib.nearest_plan(x, pl) for x in xs], 5) / len(xs)
    tq = timed(lambda: [quant_cuda.quant(x, gt) for x in xs], 5) / len(xs)
    print("%-28s antq_nearest (scan) %6.1f us %.2f TB/s | antq_nearest_plan (table) %6.1f us %.2f TB/s | quant_cuda.quant incl. "
          "its zero idx tensor %6.1f us" % (name, t * 1e6, 134.2e6 / t / 1e12, tp * 1e6, 134.2e6 / tp / 1e12, tq * 1e6))
# the reference's 7-op _forward around it (PyTorch ops + our nearest), vs the fused kernel
g = grids.ant_flint(4, True); gt = torch.from_numpy(g).to(dev); plan = _lib.plan_for(g)
w = [torch.randn(4096, 4096, device=dev) * 0.02 for _ in range(8)]
al = [x.abs().amax(1, keepdim=True) for x in w]
def ref_forward(x, alpha):
    scale = alpha / gt.max()
    d = (x.view(x.shape[0], -1) / scale).view(x.shape)
    q, _ = quant_cuda.quant(d.view(-1), gt)
    q = q.view(x.shape)
    t = (q - d).detach() + d
    return (t.view(t.shape[0], -1) * scale).view(x.shape)
t_ref = timed(lambda: [ref_forward(x, a) for x, a in zip(w, al)], 5) / len(w)
t_fus = timed(lambda: [_lib.fakequant(x, a.view(-1), plan, 10.0, 4096, 4096, True) for x, a in zip(w, al)], 5) / len(w)
same = torch.equal(ref_forward(w[0], al[0]), _lib.fakequant(w[0], al[0].view(-1), plan, 10.0, 4096, 4096, True))
print("reference op sequence (7 PyTorch launches around quant_cuda.quant): %.1f us ; fused antq_fakequant: %.1f us ; speed-up %.1fx ; bit-identical: %s" % (t_ref * 1e6, t_fus * 1e6, t_ref / t_fus, same))
