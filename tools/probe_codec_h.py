#!/usr/bin/env python3
"""The 16-bit-domain encoder (k_encode4_hrow) on 16384 x 8192 bf16: vectors per lane and task (knob 0), occupancy (knob 10:
dynamic LDS bytes per one-wavefront workgroup), against the fp32-domain row encoder (knob 9 = 0).  Bytes: 2.5 per element."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
R, C = 16384, 8192
n = R * C
gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
cases = (("OliVe flint-4 + pairs", _lib.plan_for(np.concatenate([gn, go])), 32.0, True, gn.size),
         ("ANT flint-4", _lib.plan_for(grids.ant_flint(4, True)), 10.0, False, 0))
knob = _lib.lib().antq_debug_set
xs = [(torch.randn(R, C, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
am = [_lib.absmax(x, R, C) for x in xs]
for name, plan, gmax, ovp, nn in cases:
    for clip in (1.0, 0.25):
        al = [a * (0.25 if ovp else 1.0) * clip for a in am]

        def enc():
            return [_lib.encode4(x, a, plan, gmax, R, C, True, n_normal=nn, ovp=ovp) for x, a in zip(xs, al)]
        knob(9, 0)
        t = timed(enc, 3) / 4
        print("%-22s clip %.2f  fp32-domain row encoder        %6.1f us = %4.1f %%" % (name, clip, t * 1e6, n * 2.5 / t / 8e10), flush=True)
        knob(9, 1)
        for u, ldss in ((8, (0, 8192)), (4, (0,))):
            for lds in ldss:
                knob(0, u)
                knob(10, lds)
                t = timed(enc, 3) / 4
                print("%-22s clip %.2f  hrow U=%d lds pad %5d           %6.1f us = %4.1f %%" % (name, clip, u, lds, t * 1e6, n * 2.5 / t / 8e10), flush=True)
        if True:
            for u, lds in ((8, 8192), (8, 0)):
                for e in (1,):      # knob 13 = 1: nontemporal stores for the 8-vector tasks
                    knob(0, u); knob(10, lds); knob(13, e)
                    t = timed(enc, 3) / 4
                    print("%-22s clip %.2f  hrow U=%d lds pad %5d exp %d     %6.1f us = %4.1f %%" % (name, clip, u, lds, e, t * 1e6, n * 2.5 / t / 8e10), flush=True)
            knob(13, 0)
        knob(0, 0)
        knob(10, -1)
        t = timed(enc, 3) / 4
        print("%-22s clip %.2f  hrow default                    %6.1f us = %4.1f %%" % (name, clip, t * 1e6, n * 2.5 / t / 8e10), flush=True)
