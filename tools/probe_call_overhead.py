#!/usr/bin/env python3
"""Host time per call of the one-launch-per-tensor entry on small tensors (launch-bound): the compiled extension against
the ctypes binding (ANTQ_NO_EXT=1), and the bare launch cost for reference (an empty torch op)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ant_quantization_amd import _lib, grids, quant_cuda  # noqa: E402

dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
x = (torch.randn(64, 4096, device=dev) * 0.02).bfloat16()
a = _lib.absmax(x, 64, 4096)
o = torch.empty_like(x)
g = torch.from_numpy(grids.ant_flint(4, True)).to(dev)
xf = torch.randn(1 << 16, device=dev)


def per_call(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("extension loaded:", _lib.ext() is not None)
print("_lib.fakequant(out=...)            %.2f us per call" % per_call(lambda: _lib.fakequant(x, a, plan, 10.0, 64, 4096, True, out=o)))
print("_lib.fakequant(out=..., unordered) %.2f us per call" % per_call(lambda: _lib.fakequant(x, a, plan, 10.0, 64, 4096, True, out=o, unordered=True)))
xs = (torch.randn(8, 64, device=dev) * 0.02).bfloat16()
a_s = _lib.absmax(xs, 8, 64)
os_ = torch.empty_like(xs)
print("_lib.fakequant, 1 KiB tensor       %.2f us per call" % per_call(lambda: _lib.fakequant(xs, a_s, plan, 10.0, 8, 64, True, out=os_)))
print("_lib.fakequant, 1 KiB, unordered   %.2f us per call" % per_call(lambda: _lib.fakequant(xs, a_s, plan, 10.0, 8, 64, True, out=os_, unordered=True)))
e = _lib.ext()
if e is not None:
    pd = plan.dev(dev).data_ptr()
    print("ext.fakequant direct               %.2f us per call" % per_call(lambda: e.fakequant(x, a, plan.host_addr, pd, 10.0, 64, 4096, True, 0, o, False)))
print("quant_cuda.quant (z + zero idx)    %.2f us per call" % per_call(lambda: quant_cuda.quant(xf, g)))
print("torch: o.copy_(x) (one launch)     %.2f us per call" % per_call(lambda: o.copy_(x)))
print("torch: empty_like                  %.2f us per call" % per_call(lambda: torch.empty_like(x)))
