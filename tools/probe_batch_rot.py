#!/usr/bin/env python3
"""Same-process A/B: batched row-table kernel, bf16 flint-4: vectors per lane (knob 0) x wavefronts per workgroup (knob 6) x
rotation of the workgroup -> task map per group of 8 workgroups (knob 8), by row length.   python tools/probe_batch_rot.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
knob = _lib.lib().antq_debug_set
flint = _lib.plan_for(grids.ant_flint(4, True))
cases = [("4096 x 4096", (4096, 4096), 32), ("4096 x 2048", (4096, 2048), 32), ("768 x 3072", (768, 3072), 64),
         ("2048 x 1024", (2048, 1024), 64), ("4096 x 11008", (4096, 11008), 8), ("8192 x 28672", (8192, 28672), 3)]
res = {}
big_in = torch.empty(1 << 29, dtype=torch.int16, device=dev)
big_out = torch.empty_like(big_in)
for name, (r, c), n in cases:
    xs = [(torch.randn(r, c, device=dev) * 0.02).bfloat16() for _ in range(n)]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, r, c) for x in xs]
    nbytes = len(xs) * r * c * 4
    for rnd in range(rounds):
        res.setdefault("%-14s copy kernel, 1 GiB" % name, []).append(2 * big_in.numel() * 2 / timed(lambda: _lib.copy(big_in, big_out), 10) / 8e10)
        for u in (2, 4, 3):
            knob(0, u)
            bt = _lib.Batch([(x, o, a, flint, 10.0, r, c, True) for x, o, a in zip(xs, outs, al)])
            knob(0, 0)
            for w in (1, 2, 4):
                for rot in ((0, 1) if w < 4 else (0,)):
                    knob(6, w)
                    knob(8, rot)
                    res.setdefault("%-14s U=%d W=%d rot=%d" % (name, u, w, rot), []).append(nbytes / timed(bt.run, 10) / 8e10)
            knob(6, 0)
            knob(8, 0)
    del xs, outs, al
    torch.cuda.empty_cache()
print("batched launch bf16 flint-4, % of 8 TB/s per round")
for k, v in res.items():
    print("%-44s %s" % (k, "  ".join("%5.1f" % x for x in v)))
