#!/usr/bin/env python3
"""One launch per tensor, static alpha, rows of 1024 ... 65536 elements (a power of two of vectors): the x-domain row kernel
(knob 5 = 0) against the lane kernel with the exact per-element decision (the default for such rows), same process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096      # tensors of R x R elements
NT = max(2, 16 * 4096 * 4096 // (R * R))
n = R * R
plans = (("flint-4", _lib.plan_for(grids.ant_flint(4, True)), 10.0, False),
         ("OliVe flint-4 + pairs", _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)])), 32.0, True))
for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
    xs = [(torch.randn(R, R, device=dev) * 0.02).to(dt) for _ in range(NT)]
    outs = [torch.empty_like(x) for x in xs]
    for name, plan, gmax, ovp in plans:
        for G in ((4096, n) if R != 4096 else (1024, 4096, 16384, 65536, n)):
            per_row = G != n
            al = [_lib.absmax(x, n // G, G) * (0.25 if ovp else 1.0) if per_row else x.float().abs().max().reshape(1) for x in xs]
            res = []
            for knob in (0, 1):
                _lib.lib().antq_debug_set(5, knob)
                t = timed(lambda: [_lib.fakequant(x, a, plan, gmax, n // G if per_row else 1, G, per_row, ovp=ovp, out=o)
                                   for x, a, o in zip(xs, al, outs)], 5)
                res.append(NT * n * bpe / t / 8e10)
            _lib.lib().antq_debug_set(5, 1)
            print("%-9s %5d^2 %-22s rows of %9d: row kernel %5.1f%%   lane kernel %5.1f%%  of 8 TB/s" % (str(dt)[6:], R, name, G, res[0], res[1]), flush=True)
    del xs, outs
