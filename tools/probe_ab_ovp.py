#!/usr/bin/env python3
"""Same-box A/B of library builds on the pair-rule (OliVe) batched kernel's variants: short rows (4 wavefronts per
workgroup), 2-vector tasks, and a big-footprint batch (4-vector tasks).  % of 8 TB/s, builds interleaved over rounds.
    python tools/probe_ab_ovp.py libantq.so libantq_base.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
res = []
for rows, K, n in ((4096, 1024, 64), (4096, 4096, 32), (16384, 4096, 80)):
    xs = [(torch.randn(rows, K, device=dev) * 0.02).bfloat16() for _ in range(n)]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, rows, K) * 0.25 for x in xs]
    b = _lib.Batch([(x, o, a, ol, 32.0, rows, K, True) for x, o, a in zip(xs, outs, al)], ovp=True)
    res.append(n * rows * K * 4 / timed(b.run, 10) / 8e10)
    del xs, outs, b
print(" ".join("%%.2f" %% r for r in res))
""" % (ROOT, ROOT)


def main():
    libs = sys.argv[1:] or ["libantq.so"]
    res = {l: [] for l in libs}
    for rnd in range(2):
        for l in libs:
            env = dict(os.environ, ANTQ_LIB=os.path.join(ROOT, "ant_quantization_amd", l))
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [x for x in out.stdout.strip().splitlines() if x and x[0].isdigit()]
            res[l].append(line[-1] if line else "failed: " + out.stderr[-300:])
    for l in libs:
        print("%-22s 64 x [4096,1024] (4 waves/WG) / 32 x 4096^2 (2-vector tasks) / 80 x [16384,4096] = 21.5 GB (4-vector tasks): %s" % (
            l, "   ".join(res[l])))


if __name__ == "__main__":
    main()
