#!/usr/bin/env python3
"""Same-process A/B of the batched row-table kernel's task size (knob 0: vectors per lane) and workgroup size (knob 6:
wavefronts per workgroup) by row length: bf16, OliVe flint-4 + outlier-victim pairs (the C3 / C4 shapes) and plain flint-4.
    python tools/probe_batch_u.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
knob = _lib.lib().antq_debug_set
flint = _lib.plan_for(grids.ant_flint(4, True))
ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
cases = [("4096 x 4096", (4096, 4096), 32), ("8192 x 8192", (8192, 8192), 8), ("1024 x 8192", (1024, 8192), 32),
         ("28672 x 8192", (28672, 8192), 3), ("8192 x 28672", (8192, 28672), 3), ("16384 x 4096", (16384, 4096), 8),
         ("4096 x 16384", (4096, 16384), 8), ("4096 x 11008", (4096, 11008), 8), ("3072 x 768", (3072, 768), 64),
         ("768 x 3072", (768, 3072), 64), ("2048 x 1024", (2048, 1024), 64), ("4096 x 2048", (4096, 2048), 32)]
res = {}
for name, (r, c), n in cases:
    for dt in (torch.bfloat16, torch.float32) if c in (4096, 768) else (torch.bfloat16,):
        xs = [(torch.randn(r, c, device=dev) * 0.02).to(dt) for _ in range(n if dt == torch.bfloat16 else max(1, n // 2))]
        outs = [torch.empty_like(x) for x in xs]
        al = [_lib.absmax(x, r, c) for x in xs]
        nbytes = len(xs) * r * c * 2 * xs[0].element_size()
        for ovp in (False, True):
            for rnd in range(rounds):
                for u in (2, 3, 4):
                    knob(0, u)
                    if ovp:
                        bt = _lib.Batch([(x, o, a * 0.25, ol, 32.0, r, c, True) for x, o, a in zip(xs, outs, al)], ovp=True)
                    else:
                        bt = _lib.Batch([(x, o, a, flint, 10.0, r, c, True) for x, o, a in zip(xs, outs, al)])
                    knob(0, 0)
                    for w in (1, 4):
                        knob(6, w)
                        key = "%-14s %-8s %s  U=%d W=%d" % (name, str(dt)[6:], "OVP  " if ovp else "plain", u, w)
                        res.setdefault(key, []).append(nbytes / timed(bt.run, 10) / 8e10)
                    knob(6, 0)
        del xs, outs, al
        torch.cuda.empty_cache()
print("batched launch, % of 8 TB/s per round")
for k, v in res.items():
    print("%-52s %s" % (k, "  ".join("%5.1f" % x for x in v)))
