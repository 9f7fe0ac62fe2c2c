#!/usr/bin/env python3
"""antq_encode4 (row-table encoder): vectors per lane and task 2 / 4 / 8 (knob 0), plain and pair rule, bf16 / fp32."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
flint = _lib.plan_for(grids.ant_flint(4, True))
gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
ol = _lib.plan_for(np.concatenate([gn, go]))
for dtype in (torch.bfloat16, torch.float32):
    for rows, K in ((16384, 8192), (16384, 4096), (8192, 2048)):
        x = (torch.randn(rows, K, device=dev) * 0.02).to(dtype)
        a = _lib.absmax(x, rows, K)
        n = rows * K
        bpe = (2 if dtype == torch.bfloat16 else 4) + 0.5
        line = "%-9s %6d x %5d " % (str(dtype)[6:], rows, K)
        for u in (2, 4, 8):
            _lib.lib().antq_debug_set(0, u)
            t = timed(lambda: _lib.encode4(x, a, flint, 10.0, rows, K, True), 10)
            t2 = timed(lambda: _lib.encode4(x, a * 0.3, ol, 32.0, rows, K, True, n_normal=gn.size, ovp=True), 10)
            line += "  U=%d plain %5.1f%% pairs %5.1f%%" % (u, n * bpe / t / 8e10, n * bpe / t2 / 8e10)
        _lib.lib().antq_debug_set(0, 0)
        print(line, flush=True)
