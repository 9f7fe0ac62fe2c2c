#!/usr/bin/env python3
"""Packed 4-bit codec on 16384 x 8192 tensors (large enough that the launch, not the host call, is timed) (OliVe flint-4 + outlier-victim pairs and ANT flint-4), fp32 and bf16:
antq_encode4 (with the exact-decision element path, and -- knob 4 = 0 -- with the exact division) and antq_decode4.
Bytes counted: encode reads x and writes numel / 2; decode reads numel / 2 and writes out."""
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
for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2)):
    xs = [(torch.randn(R, C, device=dev) * 0.02).to(dt) for _ in range(4)]
    am = [_lib.absmax(x, R, C) for x in xs]
    for name, plan, gmax, ovp, nn in cases:
        al = [a * (0.25 if ovp else 1.0) for a in am]   # OliVe clips at ~3 sigma (outliers exist); ANT at the abs-max

        def enc():
            return [_lib.encode4(x, a, plan, gmax, R, C, True, n_normal=nn, ovp=ovp) for x, a in zip(xs, al)]
        te = timed(enc, 3) / 4
        _lib.lib().antq_debug_set(4, 0)
        te0 = timed(enc, 3) / 4
        _lib.lib().antq_debug_set(4, 1)
        codes = enc()
        td = timed(lambda: [_lib.decode4(c, a, plan, gmax, R, C, True, dt, n_normal=nn, ovp=ovp) for c, a in zip(codes, al)], 3) / 4
        print("%-9s %-22s encode %5.1f us = %4.1f%% of 8 TB/s (exact division: %5.1f us)   decode %5.1f us = %4.1f%%" % (
            str(dt)[6:], name, te * 1e6, n * (esz + 0.5) / te / 8e10, te0 * 1e6, td * 1e6, n * (esz + 0.5) / td / 8e10), flush=True)
        if ovp and len(sys.argv) > 1:   # persistent-workgroup sweep
            for wg in (512, 1024, 1536, 2048, 3072, 4096, 8192, 1 << 30):
                _lib.lib().antq_debug_set(1, wg)
                print("    %10d workgroups: %5.1f us" % (wg, timed(enc, 3) / 4 * 1e6), flush=True)
            _lib.lib().antq_debug_set(1, 0)
        if not ovp:   # heavy clipping (alpha = abs-max / 4: |x / s| beyond twice the outermost value for many elements)
            al = [a * 0.25 for a in am]
            print("%-9s %-22s encode %5.1f us with alpha = abs-max / 4" % (str(dt)[6:], name, timed(enc, 3) / 4 * 1e6), flush=True)
    del xs
