import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import timed as _timed   # steady-clock timing (warm-up >= 60 ms, >= 40 ms measured)


def timed(fn, reps=10):
    return _timed(fn, reps)


nb = 8
for dt, bpe in ((torch.float32, 8), (torch.bfloat16, 4)):
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(nb)]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, 4096, 4096) for x in xs]
    for name, g in (("int8 s (256)", grids.ant_int(8, True)), ("int8 u (256)", grids.ant_int(8, False)), ("int6 s", grids.ant_int(6, True)),
                    ("flint6 s", grids.ant_flint(6, True)), ("int4 s", grids.ant_int(4, True)), ("pot4 s", grids.ant_pot(4, True))):
        plan = _lib.plan_for(g)
        hdr = plan.host[:96].view(np.uint32)
        xx = [x.abs() for x in xs] if name.endswith("u (256)") else xs
        t = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xx, al, outs)]) / nb
        _lib.lib().antq_debug_set(4, 0)      # A/B: the same launches with the exact division per element
        t0 = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xx, al, outs)]) / nb
        _lib.lib().antq_debug_set(4, 1)
        jobs = [(x, o, a, plan, 10.0, 4096, 4096, True) for x, a, o in zip(xx, al, outs)]
        tb = timed(_lib.Batch(jobs).run) / nb
        print("%-8s %-14s kind=%d entries=%4d xdom=%d adom=%d : %7.1f us/launch  %6.1f Gelem/s  %.1f%% of 8 TB/s"
              "   [exact division: %.1f%%]   batched: %.1f%%" % (
            str(dt)[6:], name, plan.kind, hdr[11], hdr[16], hdr[22], t * 1e6, 16.777216e6 / t / 1e9, 16.777216e6 * bpe / t / 8e10,
            16.777216e6 * bpe / t0 / 8e10, 16.777216e6 * bpe / tb / 8e10))
    del xs, outs
