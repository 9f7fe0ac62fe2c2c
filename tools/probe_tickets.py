"""Ticket shapes of the one-launch reductions (knob 17 = workgroups per group, 18 = workgroups): 16 x 4096^2, launch per tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ant_quantization_amd import _lib
from bench_configs import timed
dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: _lib._stream_int(dev)
red = _lib._reduce_ws(dev)
n = 4096 * 4096
for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    code = _lib._DTYPES[dt]
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    outs = [torch.empty(1, device=dev) for _ in xs]
    og = [t + (torch.randn_like(t.float()) * 0.001).to(dt) for t in xs[:8]]
    gg = [torch.randn(4096, 4096, device=dev).to(dt) for _ in range(8)]
    gs = [torch.empty(1, dtype=torch.float64, device=dev) for _ in range(8)]
    for blocks in (128, 256, 512, 1024):
        for group in (8, 16, 32, 64, 1024):
            if (blocks + group - 1) // group > 64:
                continue
            L.antq_debug_set(17, group); L.antq_debug_set(18, blocks)
            ta = timed(lambda: [L.antq_absmax_t(x.data_ptr(), o.data_ptr(), n, code, red.data_ptr(), st()) for x, o in zip(xs, outs)], 20) / 16
            tg = timed(lambda: [L.antq_alpha_grad_t(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, g.data_ptr(), code, red.data_ptr(), st())
                                for a, b, c, g in zip(xs, og, gg, gs)], 10) / 8
            print("%-9s blocks %4d group %4d: absmax_t %6.2f us (%4.1f %%)   alpha_grad_t %6.2f us (%4.1f %%)" % (
                str(dt)[6:], blocks, group, ta * 1e6, n * esz / ta / 8e10, tg * 1e6, 3 * n * esz / tg / 8e10), flush=True)
    L.antq_debug_set(17, 0); L.antq_debug_set(18, 0)
    assert int(red[:16384].count_nonzero()) == 0
