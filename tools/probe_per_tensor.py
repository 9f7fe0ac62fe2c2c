"""One launch per 4096 x 4096 bf16 tensor (antq_fakequant, the reference's granularity): launch shapes of the 16-bit-domain
row kernel (knob 0: vectors per lane and task, knob 10: dynamic LDS per workgroup = occupancy) against the round-3 paths
(knob 9 = 0), ordered and unordered.   python tools/probe_per_tensor.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
plan = _lib.plan_for(grids.ant_flint(4, True))
nb = 32
xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nb)]
al = [_lib.absmax(x, 4096, 4096) for x in xs]
outs = [torch.empty_like(x) for x in xs]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def run(unordered):
    for i in range(nb):
        _lib.fakequant(xs[i], al[i], plan, 10.0, 4096, 4096, True, out=outs[i], unordered=unordered)

def timeit(unordered):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(5): run(unordered)
        torch.cuda.synchronize()
    e0.record()
    for _ in range(10): run(unordered)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * nb)
    return us, 4096 * 4096 * 4 / (us * 1e-6) / 8e12 * 100

variants = [("round-3 paths (knob 9 = 0)", {9: 0})]
for u in (2, 4):
    for pad in (0, 2048, 4608, 8192, 12 * 1024):
        variants.append(("hrow U=%d lds pad %5d (%d wg/CU)" % (u, pad, min(32, 160 * 1024 // (2048 + pad))), {0: u, 10: pad}))
variants = [v for v in variants if "pad 12288" not in v[0] and "pad  8192" not in v[0]]
variants += [("hrow U=4 W=4", {0: 4, 6: 4}), ("hrow U=2 W=4", {0: 2, 6: 4}), ("hrow U=8 (one wavefront per 4096-element row)", {0: 8}),
             ("hrow U=8 lds pad  4608 (24 wg/CU)", {0: 8, 10: 4608}), ("hrow default", {})]
for rnd in range(2):
    for name, kn in variants:
        for k, v in kn.items(): knob(k, v)
        o = timeit(False); uo = timeit(True)
        knob(0, 0); knob(9, 1); knob(10, -1); knob(6, 0)
        print("%-40s ordered %6.2f us = %5.2f %% | unordered %6.2f us = %5.2f %%" % (name, o[0], o[1], uo[0], uo[1]), flush=True)
