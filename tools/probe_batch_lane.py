"""The headline batch (32 x 4096^2 bf16 / 16 x fp32, per-row alpha, plain and with OliVe's pairs) through the per-row table
kernel (knob 5 = 0) and as lane jobs with the exact per-element decision (the default for rows of a power of two of vectors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
flint = _lib.plan_for(grids.ant_flint(4, True))
ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
    nt = 32 if dt == torch.bfloat16 else 16
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(nt)]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, 4096, 4096) for x in xs]
    n = nt * 4096 * 4096 * bpe
    for rnd in range(3):
        res = []
        for knob in (0, 1):
            _lib.lib().antq_debug_set(5, knob)
            b1 = _lib.Batch([(x, o, a, flint, 10.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)])
            b2 = _lib.Batch([(x, o, a * 0.25, ol, 32.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)], ovp=True)
            _lib.lib().antq_debug_set(5, 1)
            res += [n / timed(b1.run, 30) / 8e10, n / timed(b2.run, 30) / 8e10]
        print(str(dt)[6:], "row-table kernel plain/OVP %.2f %.2f   lane kernel plain/OVP %.2f %.2f" % tuple(res), flush=True)
    del xs, outs
