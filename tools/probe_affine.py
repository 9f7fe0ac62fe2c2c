#!/usr/bin/env python3
"""antq_affine (AsymmetricQuantFunction, BASELINE configs[0]'s operator) on 8 x 4096^2 fp32: per tensor and per row."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from ant_quantization_amd import _lib  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
xs = [torch.randn(4096, 4096, device=dev) for _ in range(8)]
for per_row in (False, True):
    mn = [(x.amin(1) if per_row else x.min().reshape(1)).contiguous() for x in xs]
    mx = [(x.amax(1) if per_row else x.max().reshape(1)).contiguous() for x in xs]
    secs = timed(lambda: [_lib.affine(x, 8, a, b, 4096, 4096, per_row) for x, a, b in zip(xs, mn, mx)], 5)
    print("antq_affine 8-bit fp32 4096^2 %-10s: %5.1f us/launch = %4.1f%% of 8 TB/s" % (
        "per row" if per_row else "per tensor", secs / 8 * 1e6, 8 * 4096 * 4096 * 8 / secs / 8e10), flush=True)
# one large tensor (16384 x 16384 fp32, 1.07 GB in + 1.07 GB out): the launch boundary amortised
x = torch.randn(16384, 16384, device=dev)
for per_row in (False, True):
    mn = (x.amin(1) if per_row else x.min().reshape(1)).contiguous()
    mx = (x.amax(1) if per_row else x.max().reshape(1)).contiguous()
    secs = timed(lambda: _lib.affine(x, 8, mn, mx, 16384, 16384, per_row), 5)
    print("antq_affine 8-bit fp32 16384^2 %-10s: %6.1f us/launch = %4.1f%% of 8 TB/s" % (
        "per row" if per_row else "per tensor", secs * 1e6, 16384 * 16384 * 8 / secs / 8e10), flush=True)
