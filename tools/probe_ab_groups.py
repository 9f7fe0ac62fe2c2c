#!/usr/bin/env python3
"""Same-box A/B of library builds on the small-group launches: bf16 / fp32 static and dynamic for the group sizes in
ANTQ_AB_GROUPS (default 16), batched (16 x 4096^2), each build in its own process, builds interleaved over three rounds.
    python tools/probe_ab_groups.py libantq.so libantq_new.so        (paths relative to ant_quantization_amd/)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed
dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
n = 4096 * 4096
res = []
GS = [int(g) for g in os.environ.get("ANTQ_AB_GROUPS", "16").split(",")]
for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    outs = [torch.empty_like(x) for x in xs]
    for G in GS:
        al = [_lib.absmax(x, n // G, G) for x in xs]
        bs = _lib.Batch([(x, o, a, plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)])
        bd = _lib.Batch([(x, o, torch.empty_like(a), plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)], dynamic=True)
        res += [16 * n * bpe / timed(bs.run, 20) / 8e10, 16 * n * bpe / timed(bd.run, 20) / 8e10]
    del xs, outs
print(" ".join("%%.1f" %% r for r in res))
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
        print("%-20s bf16 then fp32: static / dynamic per group size (ANTQ_AB_GROUPS, default 16), batched, %% of 8 TB/s, per round: %s" % (l, "   ".join(res[l])))


if __name__ == "__main__":
    main()
