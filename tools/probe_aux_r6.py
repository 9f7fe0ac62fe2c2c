"""Round 6 read-side kernels, A/B on 16 x 4096^2 tensors (536 MB bf16 / 1 GB fp32: beyond the 256 MB
Infinity Cache), one launch per tensor, back to back:
  (a) whole-tensor abs-max: see tools/probe_absmax_fresh.py;
  (b) per-row abs-max of 4096-element rows: round-5 kernel (knob 16 = 0) vs one streaming wavefront per row;
  (c) alpha gradient per tensor: antq_alpha_grad (two launches) vs antq_alpha_grad_t (one)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ant_quantization_amd import _lib
from bench_configs import timed
dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: _lib._stream_int(dev)


def show(name, byt, secs, n):
    print("%-74s %6.2f us/launch  %5.2f TB/s (%4.1f%% of 8)" % (name, secs / n * 1e6, byt / secs / 1e12, byt / secs / 8e10), flush=True)


for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    code = _lib._DTYPES[dt]
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    byt = 16 * 4096 * 4096 * esz
    nm = str(dt)[6:]
    n = 4096 * 4096
    outs = [torch.empty(1, device=dev) for _ in xs]
    zeros = torch.zeros(4096, device=dev)
    red = _lib._reduce_ws(dev)
    # (a): tools/probe_absmax_fresh.py (a fresh zero slot per call: the closing atomics are real)
    # (b)
    rows_out = [torch.empty(4096, device=dev) for _ in xs]
    for knob in (0, 1):
        L.antq_debug_set(16, knob)
        show("abs-max per row %s, rows of 4096: %s" % (nm, "one streaming wavefront per row, whole row in flight" if knob else "round-5 kernel (4 waves / workgroup, plain loads)"), byt,
             timed(lambda: [L.antq_absmax(x.data_ptr(), o.data_ptr(), 4096, 4096, 1, code, st()) for x, o in zip(xs, rows_out)], 20), 16)
    assert torch.equal(rows_out[0], xs[0].float().abs().amax(1))
    # (c)
    og = [t + (torch.randn_like(t.float()) * 0.001).to(dt) for t in xs[:8]]
    gg = [torch.randn(4096, 4096, device=dev).to(dt) for _ in range(8)]
    gs = [torch.empty(1, dtype=torch.float64, device=dev) for _ in range(8)]
    ws = _lib._workspace(dev)
    b3 = 8 * 4096 * 4096 * 3 * esz
    show("alpha gradient per tensor %s, antq_alpha_grad (two launches)" % nm, b3,
         timed(lambda: [L.antq_alpha_grad(a.data_ptr(), b.data_ptr(), c.data_ptr(), 4096, 4096, 0, g.data_ptr(), ws.data_ptr(), code, st())
                        for a, b, c, g in zip(xs, og, gg, gs)], 10), 8)
    r_old = [float(g) for g in gs]
    show("alpha gradient per tensor %s, antq_alpha_grad_t (ONE launch)" % nm, b3,
         timed(lambda: [L.antq_alpha_grad_t(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, g.data_ptr(), code, red.data_ptr(), st())
                        for a, b, c, g in zip(xs, og, gg, gs)], 10), 8)
    r_new = [float(g) for g in gs]
    print("   (two-launch vs one-launch sums, relative difference: %.2e)" % max(abs(a - b) / max(abs(a), 1e-300) for a, b in zip(r_old, r_new)))
    gr = [torch.empty(4096, dtype=torch.float64, device=dev) for _ in range(8)]
    show("alpha gradient per row %s (for scale)" % nm, b3,
         timed(lambda: [L.antq_alpha_grad(a.data_ptr(), b.data_ptr(), c.data_ptr(), 4096, 4096, 1, g.data_ptr(), None, code, st())
                        for a, b, c, g in zip(xs, og, gg, gr)], 10), 8)
    assert int(red[:16384].count_nonzero()) == 0, "the ticket block was not left zeroed"
    del xs, og, gg
