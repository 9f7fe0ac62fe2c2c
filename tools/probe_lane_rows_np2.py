#!/usr/bin/env python3
"""One launch per tensor, static alpha, rows that are NOT a power of two of vectors (4608, 11008, 28672, 768, 3072 wide): the
per-row table kernel (knob 5 = 1, the default for such rows) against the lane kernel with the f64-reciprocal row index
(knob 5 = 2), same process; and the same tensors in one batched launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
    for rows, K in ((3640, 4608), (1536, 11008), (584, 28672), (3072, 768), (768, 3072), (21840, 768)):
        nt = 8
        xs = [(torch.randn(rows, K, device=dev) * 0.02).to(dt) for _ in range(nt)]
        outs = [torch.empty_like(x) for x in xs]
        al = [_lib.absmax(x, rows, K) for x in xs]
        res = []
        for knob in (1, 2):
            _lib.lib().antq_debug_set(5, knob)
            t = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, rows, K, True, out=o) for x, a, o in zip(xs, al, outs)], 5)
            res.append(nt * rows * K * bpe / t / 8e10)
        _lib.lib().antq_debug_set(5, 1)
        print("%-9s %6d x %6d (%5.1f MB): row kernel %5.1f%%   lane kernel %5.1f%%  of 8 TB/s" % (
            str(dt)[6:], rows, K, rows * K * bpe / 2 / 1e6, res[0], res[1]), flush=True)
        del xs, outs
