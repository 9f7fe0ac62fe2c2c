"""Per-launch durations of the headline batched launch right after different preludes (round 4).

The driver times launches 6..25 of the batched kernel (bench.py --steps 20 --warmup 5); round 3's rocprof stats showed the
first ~20 launches ~9 % slower than steady state.  This probe records an event between every pair of consecutive
launches and prints the series for a few launch shapes and preludes, each after the GPU has idled for `idle` seconds.

    python tools/probe_transient.py [n_launches=60] [idle_s=1.5]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ant_quantization_amd import _lib, grids

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
IDLE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
plan = _lib.plan_for(grids.ant_flint(4, True))
nb = 32
gen = torch.Generator(device=dev)
gen.manual_seed(6)
x_slab = torch.empty(nb, 4096, 4096, dtype=torch.bfloat16, device=dev)
out_slab = torch.empty_like(x_slab)
xs, outs, al = [], [], []
for i in range(nb):
    x_slab[i] = (torch.randn(4096, 4096, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
    xs.append(x_slab[i])
    outs.append(out_slab[i])
    al.append(_lib.absmax(xs[i], 4096, 4096))
torch.cuda.synchronize()
BYTES = nb * 4096 * 4096 * 4


def make_batch(waves, u, h):
    knob(6, waves)
    knob(0, u)
    knob(9, h)
    b = _lib.Batch([(x, o, a, plan, 10.0, 4096, 4096, True) for x, a, o in zip(xs, al, outs)])
    knob(0, 0)
    knob(9, 1)
    return b


def series(fn, n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]


def per_tensor(unordered):
    for i in range(nb):
        _lib.fakequant(xs[i], al[i], plan, 10.0, 4096, 4096, True, out=outs[i], unordered=unordered)


def busy(fn, seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()


def show(tag, s):
    f = lambda v: BYTES / (v * 1e-6) / 8e12 * 100
    head = " ".join("%.0f" % v for v in s[:30])
    print("%-44s first5 %.1f us | 6..25 %.1f us = %.2f %% | last20 %.1f us = %.2f %% | max %.1f\n    %s" % (
        tag, sum(s[:5]) / 5, sum(s[5:25]) / 20, f(sum(s[5:25]) / 20), sum(s[-20:]) / 20, f(sum(s[-20:]) / 20), max(s), head),
        flush=True)


# (round 4: bf16 rows take the 16-bit-domain kernel -- one wavefront per workgroup always, knob 6 is moot for it; knob 0 forces
#  the vectors per lane and task; knob 9 = 0 brings back the round-3 kernel, for which knob 6 = wavefronts per workgroup)
shapes = [("K1h u=4 (shipped)", 0, 0, 1), ("K1h u=2 (forced)", 0, 2, 1), ("K1h u=3 (forced)", 0, 3, 1), ("round-3 kernel W=1 u=2", 1, 2, 0), ("round-3 kernel W=4 u=2", 4, 2, 0)]
for rnd in range(2):
    for name, w, u, h in shapes:
        bt = make_batch(w, u, h)
        knob(6, w)
        # (a) the bench's prelude: per-tensor launches 0.3 s ordered + 0.15 s unordered, then the batched launches
        time.sleep(IDLE)
        busy(lambda: per_tensor(False), 0.3)
        busy(lambda: per_tensor(True), 0.15)
        show("%s | prelude per-tensor 0.45 s" % name, series(bt.run, N))
        # (b) from idle, nothing before
        time.sleep(IDLE)
        show("%s | from idle" % name, series(bt.run, N))
        # (c) prelude = a plain copy of the same bytes for 0.45 s
        time.sleep(IDLE)
        busy(lambda: _lib.copy(x_slab, out_slab), 0.45)
        show("%s | prelude copy 0.45 s" % name, series(bt.run, N))
        # (d) prelude = the batched launch itself for 0.45 s
        time.sleep(IDLE)
        busy(bt.run, 0.45)
        show("%s | prelude batched 0.45 s" % name, series(bt.run, N))
        knob(6, 0)
