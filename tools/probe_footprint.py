#!/usr/bin/env python3
"""Batched launch over n x [4096, 4096] bf16 tensors (flint-4) as a function of the footprint: vectors per lane (knob 0) x
wavefronts per workgroup (knob 6), interleaved in one process.   python tools/probe_footprint.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
knob = _lib.lib().antq_debug_set
flint = _lib.plan_for(grids.ant_flint(4, True))
xs, outs, al = [], [], []
print("n tensors (GB in + GB out): % of 8 TB/s per round")
for n in (8, 16, 32, 64, 128, 256, 512):
    while len(xs) < n:
        x = (torch.randn(4096, 4096, device=dev) * 0.02).bfloat16()
        xs.append(x)
        outs.append(torch.empty_like(x))
        al.append(_lib.absmax(x, 4096, 4096))
    res = {}
    for rnd in range(rounds):
        for u in (2, 4):
            knob(0, u)
            bt = _lib.Batch([(x, o, a, flint, 10.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)])
            knob(0, 0)
            for w in (1, 4):
                knob(6, w)
                res.setdefault("U=%d W=%d" % (u, w), []).append(n * 4096 * 4096 * 4 / timed(bt.run, 5) / 8e10)
            knob(6, 0)
    print("%4d (%5.2f + %5.2f GB)   %s" % (n, n * 0.03355, n * 0.03355, "   ".join("%s %s" % (k, "/".join("%.1f" % v for v in vs)) for k, vs in res.items())), flush=True)
