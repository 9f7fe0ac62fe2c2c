#!/usr/bin/env python3
"""Same-process, interleaved A/B of the launch shapes (round 3): wavefronts per workgroup of the batched row-table kernel
(knob 6), wavefronts per workgroup / vectors per lane of the one-launch-per-tensor lane kernel (knobs 6 / 7), ordinary
against unordered launches (ANTQ_FLAG_UNORDERED).  Headline tensors: 32 x 4096^2 bf16, flint-4 (+ OliVe pairs).
    python tools/probe_launch_shapes.py [rounds]"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
nb = 16 if dtype == torch.float32 else 32
xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dtype) for _ in range(nb)]
outs = [torch.empty_like(x) for x in xs]
al = [_lib.absmax(x, 4096, 4096) for x in xs]
flint = _lib.plan_for(grids.ant_flint(4, True))
ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
n_bytes = nb * 4096 * 4096 * 2 * xs[0].element_size()
knob = _lib.lib().antq_debug_set


def batch(ovp):
    if ovp:
        return _lib.Batch([(x, o, a * 0.25, ol, 32.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)], ovp=True)
    return _lib.Batch([(x, o, a, flint, 10.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)])


b_plain, b_ovp = batch(False), batch(True)
res = {}


def note(name, seconds):
    res.setdefault(name, []).append(n_bytes / seconds / 8e10)


copy_in = torch.stack(xs)
copy_out = torch.empty_like(copy_in)
for rnd in range(rounds):
    note("copy kernel (antq_copy), one launch", timed(lambda: _lib.copy(copy_in, copy_out), 20))
    for w in (4, 2, 1):
        knob(6, w)
        note("batched plain, %d wavefront(s) per workgroup" % w, timed(b_plain.run, 20))
        note("batched OVP,   %d wavefront(s) per workgroup" % w, timed(b_ovp.run, 20))
    knob(6, 0)
    for u in (2, 1):                       # (the descriptors fix the task size at build time: rebuild the batches)
        knob(0, u)
        bu_plain, bu_ovp = batch(False), batch(True)
        knob(0, 0)
        note("batched plain, 1 wavefront per workgroup, %d vectors per lane" % u, timed(bu_plain.run, 20))
        note("batched OVP,   1 wavefront per workgroup, %d vectors per lane" % u, timed(bu_ovp.run, 20))
    # one launch per tensor: (knob 5: 1 default / 0 row-table kernel / 2 lane kernel, knob 6, knob 7)
    for unordered in (False, True):
        for k5, w, u in ((1, 0, 0), (2, 4, 2), (2, 1, 4), (2, 1, 2), (0, 4, 0), (0, 1, 0)):
            knob(5, k5)
            knob(6, w)
            knob(7, u)
            for ovp in (False, True):
                def pt():
                    for x, o, a in zip(xs, outs, al):
                        if ovp:
                            _lib.fakequant(x, a, ol, 32.0, 4096, 4096, True, ovp=True, out=o, unordered=unordered)
                        else:
                            _lib.fakequant(x, a, flint, 10.0, 4096, 4096, True, out=o, unordered=unordered)
                kern = {1: "default", 2: "lane kernel", 0: "row-table kernel"}[k5]
                note("per tensor %s %s, %-16s wavefronts/workgroup %d, vectors/lane %d"
                     % ("OVP  " if ovp else "plain", "unordered" if unordered else "ordered  ", kern, w, u), timed(pt, 5))
    knob(5, 1)
    knob(6, 0)
    knob(7, 0)
print("%s, %% of 8 TB/s per round (algorithmic bytes / time)" % str(dtype))
for k, v in res.items():
    print("%-86s %s" % (k, "  ".join("%5.1f" % x for x in v)))
