"""Whole-tensor abs-max launch shapes (knob 13; experiment builds only): 33.5 MB bf16, 67 MB fp32, 12.6 / 50 MB activations, 0.5 GB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ant_quantization_amd import _lib
from bench_configs import timed
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
cases = []
for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    for shape in ((4096, 4096), (64 * 128, 768), (64 * 128, 3072), (16384, 16384)):
        nb = 16 if shape[0] * shape[1] <= 1 << 25 else 1
        cases.append((dt, esz, shape, [(torch.randn(*shape, device=dev) * 0.02).to(dt) for _ in range(nb)]))
    for k in (0, 6, 7, 8, 9):
        knob(13, k)
        line = "knob13=%d %-9s" % (k, str(dt)[6:])
        for dt2, esz2, shape, xs in cases:
            if dt2 != dt:
                continue
            ref = [float(x.abs().max()) for x in xs[:2]]
            got = [float(_lib.absmax(x, shape[0], shape[1], per_row=False)) for x in xs[:2]]
            assert ref == got, (ref, got)
            secs = timed(lambda: [_lib.absmax(x, shape[0], shape[1], per_row=False) for x in xs], 5) / len(xs)
            byt = shape[0] * shape[1] * esz2
            line += "  %dx%d %6.1f us %4.1f%%" % (shape[0], shape[1], secs * 1e6, byt / secs / 8e10)
        print(line, flush=True)
    cases = [c for c in cases if c[0] != dt]
knob(13, 0)
