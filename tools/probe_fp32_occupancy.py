"""(pad -1 = the shipped rule, 0 = no cap.)  fp32 batched launches (lane jobs, k_fq_batch_d, 256-thread workgroups) with extra dynamic LDS per workgroup (knob 11): does
the bytes-in-flight rule of the 16-bit-domain kernels (about 64-96 KiB per CU) hold here too?  16 x 4096^2 fp32 and ResNet-50."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import resnet50_shapes
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
plan = _lib.plan_for(grids.ant_flint(4, True))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def bench(b, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): b.run()
        torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): b.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
def mk(shapes):
    xs = [torch.randn(r, k, device=dev) * 0.02 for r, k in shapes]
    outs = [torch.empty_like(x) for x in xs]
    al = [_lib.absmax(x, x.shape[0], x.shape[1]) for x in xs]
    return xs, outs, al
sets = {"16 x 4096^2 fp32": [(4096, 4096)] * 16, "ResNet-50 54 W fp32": [(s[0], int(np.prod(s[1:]))) for s in resnet50_shapes()]}
for name, shapes in sets.items():
    xs, outs, al = mk(shapes)
    nbytes = sum(x.numel() for x in xs) * 8
    for u in ((1, 2, 4) if "ResNet" in name else (2, 4)):
        for pad in (-1, 0, 18432, 20480, 22528, 24576, 28672, 32768, 40960, 53248):
            knob(0, u); knob(11, pad)
            b = _lib.Batch([(x, o, a, plan, 10.0, x.shape[0], x.shape[1], True) for x, o, a in zip(xs, outs, al)])
            t = bench(b, 50 if nbytes > 1e9 else 300)
            knob(0, 0); knob(11, -1)
            print("%-22s u=%d lds pad %5d: %7.1f us  %5.1f %%" % (name, u, pad, t * 1e6, nbytes / t / 8e12 * 100), flush=True)
