#!/usr/bin/env python3
"""Throughput of the calibration kernels (candidate evaluations per second = elements x clip ratios x types / time):
k_search_sse (one type per read) and k_search_sse_multi (3 types on one read), per row and per tensor, fp32 and bf16,
on a 4096 x 4096 tensor and on BERT-base's 768 / 3072-wide rows; antq_calibrate end to end.
    python tools/probe_search.py            (ANTQ_LIB=... for an A/B of library builds)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from ant_quantization_amd import _lib, core, grids
from bench_configs import timed


def main():
    dev = torch.device("cuda:0")
    gs = [grids.ant_grid(t, 4, True) for t in ("int", "flint", "pot")]
    plans = [_lib.plan_for(g) for g in gs]
    gm = [10.0] * 3
    ratios = core._ratios(75, 145, 1, dev)
    nc = ratios.numel()
    print("library:", os.environ.get("ANTQ_LIB", "in-tree"))
    for name, shape, dtype, per_row in (("4096x4096 fp32 per row", (4096, 4096), torch.float32, True),
                                        ("4096x4096 bf16 per row", (4096, 4096), torch.bfloat16, True),
                                        ("4096x4096 fp32 per tensor", (4096, 4096), torch.float32, False),
                                        ("3072x768 fp32 per row", (3072, 768), torch.float32, True),
                                        ("768x3072 fp32 per row", (768, 3072), torch.float32, True),
                                        ("16384x4096 bf16 per row", (16384, 4096), torch.bfloat16, True)):
        x = (torch.randn(*shape, device=dev) * 0.02).to(dtype)
        rows, K = shape
        r_, k_ = (rows, K) if per_row else (1, rows * K)
        xm = _lib.absmax(x, rows, K, per_row=per_row)
        n = x.numel()
        t1 = timed(lambda: _lib.search_sse(x, r_, k_, xm, per_row, ratios, plans[1], 10.0), 5)
        t3 = timed(lambda: _lib.search_sse_multi(x, r_, k_, xm, per_row, ratios, plans, gm), 5)
        tc = timed(lambda: _lib.calibrate(x, rows, K, per_row, plans, gm, 75, 145, 1), 5)
        print("%-28s one type %7.1f G/s (%6.3f ms)   three types, one read %7.1f G/s (%6.3f ms)   antq_calibrate %6.3f ms" % (
            name, n * nc / t1 / 1e9, t1 * 1e3, 3 * n * nc / t3 / 1e9, t3 * 1e3, tc * 1e3), flush=True)


if __name__ == "__main__":
    main()
