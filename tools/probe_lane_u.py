#!/usr/bin/env python3
"""A/B of the lane-job task size in BIG batches (16 x 4096^2, group-16 / 64): 4 (default) against 2 vectors per lane."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
n = 4096 * 4096
for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    outs = [torch.empty_like(x) for x in xs]
    for G in (16, 64, 128, 256, 512):
        al = [_lib.absmax(x, n // G, G) for x in xs]
        res = []
        for rnd in range(2):
            for knob in (4, 2):
                _lib.lib().antq_debug_set(0, knob)
                for dyn in (False, True):
                    bt = _lib.Batch([(x, o, (torch.empty_like(a) if dyn else a), plan, 10.0, n // G, G, True)
                                     for x, a, o in zip(xs, al, outs)], dynamic=dyn)
                    res.append("U=%d %s %.1f%%" % (knob, "dyn" if dyn else "static", 16 * n * bpe / timed(bt.run, 20) / 8e10))
        _lib.lib().antq_debug_set(0, 0)
        print(str(dt)[6:], "group-%d:" % G, "  ".join(res), flush=True)
    del xs, outs
