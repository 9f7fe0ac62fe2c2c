#!/usr/bin/env python3
"""In-process, interleaved A/B of the batched launch over a WHOLE model's weights (large footprint): OPT-6.7B (192 tensors,
12.9 GB in + 12.9 GB out) or the 70 B stack in place (137 GB), OliVe flint-4 + pairs, bf16: vectors per lane (knob 0) x
wavefronts per workgroup (knob 6).    python tools/probe_sharded_ab.py [opt6.7b|llama70b] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import llama70b_shapes, opt67_shapes, timed  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "opt6.7b"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
shapes = opt67_shapes(32) if model.startswith("opt6.7b") else opt67_shapes(int(model[3:])) if model.startswith("opt") else llama70b_shapes(80)
if ":" in model:                      # e.g. opt6.7b:4096x4096 -- only the tensors of that shape
    want = tuple(int(v) for v in model.split(":")[1].split("x"))
    shapes = [s_ for s_ in shapes if tuple(s_) == want]
if os.environ.get("PROBE_LPT") == "1":           # the order tools/bench_sharded.py uses (largest tensors first)
    from ant_quantization_amd import sharding
    shapes = [shapes[i] for i in sharding.lpt_assign([2 * a * b for a, b in shapes], 1)[0]]
ovp = os.environ.get("PROBE_OVP", "1") == "1"
plan = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]) if ovp else grids.ant_flint(4, True))
gmax = 32.0 if ovp else 10.0
gen = torch.Generator(device=dev).manual_seed(4)
ws, alphas = [], []
for sh in shapes:
    w = torch.randn(*sh, device=dev, dtype=torch.bfloat16, generator=gen) * 0.02
    if os.environ.get("PROBE_PLANT") == "1":          # 0.1 % of the entries multiplied by U(8, 64)
        m = torch.rand(w.shape, device=dev, generator=gen) < 0.001
        w[m] *= torch.empty(int(m.sum()), device=dev, dtype=torch.bfloat16).uniform_(8, 64, generator=gen)
        del m
    ws.append(w)
    alphas.append(_lib.xmax_3sigma(w, sh[0], sh[1], per_row=True))
outs = ws if model.startswith("llama") else [torch.empty_like(w) for w in ws]
elems = sum(w.numel() for w in ws)
batches = {}
for u in (2, 4):
    knob(0, u)
    batches[u] = _lib.Batch([(w, o, a, plan, gmax, w.shape[0], w.shape[1], True) for w, o, a in zip(ws, outs, alphas)], ovp=ovp)
    knob(0, 0)
res = {}
for rnd in range(rounds):
    for u in (2, 4):
        for w_ in (1, 4):
            knob(6, w_)
            res.setdefault("U=%d W=%d" % (u, w_), []).append(elems * 4 / timed(batches[u].run, 5) / 8e10)
    knob(6, 0)
print("%s, %d tensors, %.1f GB %s: %% of 8 TB/s per round" % (model, len(ws), elems * 2 / 1e9, "in place" if outs is ws else "in + the same out"))
for k, v in res.items():
    print("%-12s %s" % (k, "  ".join("%5.1f" % x for x in v)))
