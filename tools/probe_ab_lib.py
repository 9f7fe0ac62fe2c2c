#!/usr/bin/env python3
"""Same-box A/B of library builds (ANTQ_LIB): the headline batch (32 x 4096^2 bf16, flint-4), the same with OliVe's
outlier-victim pairs, and the same tensors as one launch each, each build in its own process, builds interleaved over several rounds (clock / thermal drift shows up
as a trend over rounds, not as a difference between builds).
    python tools/probe_ab_lib.py libantq.so libantq_OVP7.so ...        (paths relative to ant_quantization_amd/)"""
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
xs = [(torch.randn(4096, 4096, device=dev) * 0.02).bfloat16() for _ in range(32)]
outs = [torch.empty_like(x) for x in xs]
al = [_lib.absmax(x, 4096, 4096) for x in xs]
flint = _lib.plan_for(grids.ant_flint(4, True))
ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
b1 = _lib.Batch([(x, o, a, flint, 10.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)])
b2 = _lib.Batch([(x, o, a * 0.25, ol, 32.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)], ovp=True)
n = 32 * 4096 * 4096 * 4
pt = lambda: [_lib.fakequant(x, a, flint, 10.0, 4096, 4096, True, out=o) for x, o, a in zip(xs, outs, al)]
print("%%.2f %%.2f %%.2f" %% (n / timed(b1.run, 30) / 8e10, n / timed(b2.run, 30) / 8e10, n / timed(pt, 10) / 8e10))
""" % (ROOT, ROOT)


def main():
    libs = sys.argv[1:] or ["libantq.so"]
    res = {l: [] for l in libs}
    for rnd in range(3):
        for l in libs:
            env = dict(os.environ, ANTQ_LIB=os.path.join(ROOT, "ant_quantization_amd", l))
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [x for x in out.stdout.strip().splitlines() if x and x[0].isdigit()]
            res[l].append(line[-1] if line else "failed: " + out.stderr[-200:])
    for l in libs:
        print("%-22s batched plain / batched OVP / one launch per tensor, %% of 8 TB/s, per round: %s" % (l, "   ".join(res[l])))


if __name__ == "__main__":
    main()
