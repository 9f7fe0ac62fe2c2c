#!/usr/bin/env python3
"""Which task size does THIS box prefer?  Box facts (partition modes, clocks) + the headline batch (32 x 4096^2 bf16 flint-4)
with 2 / 4 vectors per lane at 1 wavefront per workgroup, the copy kernel beside it.   python tools/probe_box.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

for cmd in (["rocm-smi", "--showmemorypartition", "--showcomputepartition"], ["rocm-smi", "--showclocks"],
            ["rocm-smi", "--showperflevel", "--showpower"]):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout
        print("\n".join(l for l in out.splitlines() if l.strip() and not set(l.strip()) <= set("=")))
    except Exception as e:      # noqa: BLE001
        print(cmd, "failed:", e)
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory >> 30, "GiB")
knob = _lib.lib().antq_debug_set
flint = _lib.plan_for(grids.ant_flint(4, True))
xs = [(torch.randn(4096, 4096, device=dev) * 0.02).bfloat16() for _ in range(32)]
outs = [torch.empty_like(x) for x in xs]
al = [_lib.absmax(x, 4096, 4096) for x in xs]
big_in, big_out = torch.stack(xs), torch.empty(32, 4096, 4096, dtype=torch.bfloat16, device=dev)
n = 32 * 4096 * 4096 * 4
res = {}
for rnd in range(3):
    res.setdefault("copy", []).append(n / timed(lambda: _lib.copy(big_in, big_out), 10) / 8e10)
    for u in (2, 4):
        knob(0, u)
        bt = _lib.Batch([(x, o, a, flint, 10.0, 4096, 4096, True) for x, o, a in zip(xs, outs, al)])
        knob(0, 0)
        for w in (1, 4):
            knob(6, w)
            res.setdefault("U=%d W=%d" % (u, w), []).append(n / timed(bt.run, 10) / 8e10)
        knob(6, 0)
for k, v in res.items():
    print("%-10s %s" % (k, "  ".join("%5.1f" % x for x in v)))
