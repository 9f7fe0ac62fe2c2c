"""Whole-tensor abs-max into a FRESH zero slot per call (what a caller really does): the closing atomics are real here
(a slot that already holds the maximum makes every workgroup skip its atomic).  16 x 4096^2, one launch per tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ant_quantization_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: _lib._stream_int(dev)
red = _lib._reduce_ws(dev)
n = 4096 * 4096
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(fn, reps=40):
    """fn(k, slot_ptr) for k in 16 tensors; fresh zero slots every pass."""
    best = 1e9
    for _ in range(3):
        z = torch.zeros(16 * reps, device=dev)
        p0 = z.data_ptr()
        torch.cuda.synchronize()
        e0.record()
        for r in range(reps):
            for k in range(16):
                fn(k, p0 + 4 * (r * 16 + k))
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / (reps * 16))
    return best


for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    code = _lib._DTYPES[dt]
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    ptr = [x.data_ptr() for x in xs]
    s = st()
    t = run(lambda k, slot: L.antq_absmax_into(ptr[k], slot, n, code, s))
    print("%-9s antq_absmax_into (256 workgroups block-strided, 256 closing atomics)%11s %6.2f us  %4.1f %%" % (str(dt)[6:], "", t * 1e6, n * esz / t / 8e10), flush=True)
    import time
    for x in xs:
        _lib.absmax(x, 1, n, per_row=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        for x in xs:
            _lib.absmax(x, 1, n, per_row=False)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 1600
    print("%-9s _lib.absmax(per_row=False): the Python binding, wall clock, host time included%4s %6.2f us  %4.1f %%" % (str(dt)[6:], "", t * 1e6, n * esz / t / 8e10), flush=True)
    L.antq_debug_set(16, 1); L.antq_debug_set(18, 0)
    t = run(lambda k, slot: L.antq_absmax_t(ptr[k], slot, n, code, red.data_ptr(), s))
    print("%-9s antq_absmax_t (tickets, writes its result)%37s %6.2f us  %4.1f %%" % (str(dt)[6:], "", t * 1e6, n * esz / t / 8e10), flush=True)
    t = run(lambda k, slot: L.antq_absmax(ptr[k], slot, 1, n, 0, code, s))
    print("%-9s antq_absmax (zeroing launch + kernel)%42s %6.2f us  %4.1f %%" % (str(dt)[6:], "", t * 1e6, n * esz / t / 8e10), flush=True)
