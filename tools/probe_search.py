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
                                        ("768x768 bf16 per row", (768, 768), torch.bfloat16, True),
                                        ("3072x768 bf16 per row", (3072, 768), torch.bfloat16, True),
                                        ("512x576 bf16 per row (conv 3x3x64)", (512, 576), torch.bfloat16, True),
                                        ("768x3072 fp32 per row", (768, 3072), torch.float32, True),
                                        ("16384x4096 bf16 per row", (16384, 4096), torch.bfloat16, True)):
        x = (torch.randn(*shape, device=dev) * 0.02).to(dtype)
        rows, K = shape
        r_, k_ = (rows, K) if per_row else (1, rows * K)
        xm = _lib.absmax(x, rows, K, per_row=per_row)
        n = x.numel()
        t1 = timed(lambda: _lib.search_sse(x, r_, k_, xm, per_row, ratios, plans[1], 10.0), 5)
        t3 = timed(lambda: _lib.search_sse_multi(x, r_, k_, xm, per_row, ratios, plans, gm), 5) if K * x.element_size() >= 1024 else float("nan")
        tc = timed(lambda: _lib.calibrate(x, rows, K, per_row, plans, gm, 75, 145, 1), 5)
        print("%-28s one type %7.1f G/s (%6.3f ms)   three types, one read %7.1f G/s (%6.3f ms)   antq_calibrate %6.3f ms" % (
            name, n * nc / t1 / 1e9, t1 * 1e3, 3 * n * nc / t3 / 1e9, t3 * 1e3, tc * 1e3), flush=True)


def hist():
    """The histogram path (16-bit tensors, one scale, no pair rule; knob 14: 0 off / 2 always) against the direct kernels."""
    dev = torch.device("cuda:0")
    knob = _lib.lib().antq_debug_set
    gs = [grids.ant_grid(t, 4, True) for t in ("int", "flint", "pot")]
    plans = [_lib.plan_for(g) for g in gs]
    gm = [10.0] * 3
    for name, n, relu in (("BERT 64x128x3072 bf16 (GELU-like)", 64 * 128 * 3072, False), ("64x128x768 bf16", 64 * 128 * 768, False),
                          ("64x128x3072 bf16 after ReLU", 64 * 128 * 3072, True), ("1 M bf16", 1 << 20, False), ("256 K bf16", 1 << 18, False)):
        x = torch.randn(n, device=dev)
        if relu:
            x = torch.relu(x)
        x = x.to(torch.bfloat16)
        for mode in (0, 2):
            knob(14, mode)
            tc = timed(lambda: _lib.calibrate(x, 1, n, False, plans, gm, 75, 150, 1), 5)
            t1 = timed(lambda: _lib.calibrate(x, 1, n, False, plans[1:2], gm[:1], 75, 150, 1), 5)
            print("%-36s %s   antq_calibrate three types %7.3f ms   one type %7.3f ms" % (name, "histogram" if mode else "direct   ", tc * 1e3, t1 * 1e3), flush=True)
    # OliVe: two codebooks (int / flint + outliers), 3-sigma statistic, outlier-victim pairs, ratios range(95... as OQ: lb..ub step 2)
    on = [np.concatenate([grids.olive_grid(t, 4, True), grids.olive_outliers(4, True)]) for t in ("int", "flint")]
    oplans = [_lib.plan_for(g) for g in on]
    ogm = [float(grids.olive_grid(t, 4, True).max()) for t in ("int", "flint")]
    for name, n, frac in (("OliVe 64x128x3072 bf16, 0.3 % outliers", 64 * 128 * 3072, 0.003), ("OliVe 2048x4096 bf16, 1 % outliers", 2048 * 4096, 0.01),
                          ("OliVe 1 M bf16, 0.3 % outliers", 1 << 20, 0.003)):
        x = torch.randn(n, device=dev) * 0.05
        m = torch.rand(n, device=dev) < frac
        x[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 60)
        x = x.to(torch.bfloat16)
        for mode in (0, 2):
            knob(14, mode)
            tc = timed(lambda: _lib.calibrate(x, 1, n, False, oplans, ogm, 75, 250, 2, xmax="3sigma", ovp=True), 5)
            print("%-40s %s   antq_calibrate two types x 88 ratios, pairs %7.3f ms" % (name, "histogram" if mode else "direct   ", tc * 1e3), flush=True)
    knob(14, 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hist":
        hist()
        sys.exit(0)
    main()
