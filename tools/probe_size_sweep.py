#!/usr/bin/env python3
"""Fraction of the HBM roofline vs tensor size: one launch per tensor (antq_fakequant) on a rotating set of distinct
buffers (>= 1 GiB in flight so the 256 MB Infinity Cache cannot hold them), and the same tensors through ONE batched
launch.  bf16 and fp32, ANT flint-4, per-row alpha, rows of 4096 elements."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    plan = _lib.plan_for(grids.ant_flint(4, True))
    warm = torch.randn(4096, 4096, device=dev)
    aw = _lib.absmax(warm, 4096, 4096)
    ow = torch.empty_like(warm)
    for _ in range(3000):                                   # ~75 ms of load: steady clocks
        _lib.fakequant(warm, aw, plan, 10.0, 4096, 4096, True, out=ow)
    print("%-10s %10s %6s | %12s %8s | %12s %8s" % ("dtype", "MB/tensor", "nbuf", "per-tensor us", "% of 8T", "batched us/t", "% of 8T"))
    for dt, bpe in ((torch.bfloat16, 2), (torch.float32, 4)):
        for rows in (64, 256, 1024, 4096, 16384, 65536):
            K = 4096
            mb = rows * K * bpe / 1e6
            nbuf = max(2, min(256, int(1.1e9 / (rows * K * bpe * 2))))
            xs = [(torch.randn(rows, K, device=dev) * 0.02).to(dt) for _ in range(nbuf)]
            al = [_lib.absmax(x, rows, K) for x in xs]
            outs = [torch.empty_like(x) for x in xs]

            def per_tensor():
                for x, a, o in zip(xs, al, outs):
                    _lib.fakequant(x, a, plan, 10.0, rows, K, True, out=o)

            bt = _lib.Batch([(x, o, a, plan, 10.0, rows, K, True) for x, a, o in zip(xs, al, outs)])
            reps = max(3, int(0.05 / (nbuf * mb * 2e6 / 5e12)))
            t_pt = timed(per_tensor, reps) / nbuf
            t_b = timed(bt.run, reps) / nbuf
            byt = rows * K * bpe * 2
            print("%-10s %10.2f %6d | %12.2f %7.1f%% | %12.2f %7.1f%%" % (
                str(dt)[6:], mb, nbuf, t_pt * 1e6, byt / t_pt / 8e10, t_b * 1e6, byt / t_b / 8e10), flush=True)
            del xs, outs, al, bt


if __name__ == "__main__":
    main()
