#!/usr/bin/env python3
"""Group-size sweep: 16 x [4096,4096] tensors viewed as groups of G elements (one alpha per group), ANT flint-4.
Static alpha through the batched launch and one launch per tensor, and the dynamic (abs-max in the kernel) variant -- batched
with the scales stored and without (alpha_dev = NULL), per tensor without (want_alpha=False).
Fractions count x / out bytes only; the alpha stream adds 4 / (G * element size) on top (6 % for bf16 group-16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from ant_quantization_amd import _lib, grids  # noqa: E402
from bench_configs import timed  # noqa: E402

dev = torch.device("cuda:0")


def main():
    plan = _lib.plan_for(grids.ant_flint(4, True))
    n = 4096 * 4096
    for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
        xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
        outs = [torch.empty_like(x) for x in xs]
        for G in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
            al = [_lib.absmax(x, n // G, G) for x in xs]
            bt = _lib.Batch([(x, o, a, plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)])
            tb = timed(bt.run, 20)
            tp = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, n // G, G, True, out=o) for x, a, o in zip(xs, al, outs)], 5)
            td = timed(lambda: [_lib.fakequant_dynamic(x, plan, 10.0, n // G, G, out=o, want_alpha=False)
                                for x, o in zip(xs, outs)], 5)
            try:                                   # rows of 2..16 KiB: alpha in the kernel, all tensors in one launch
                bd = _lib.Batch([(x, o, torch.empty_like(a), plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)],
                                dynamic=True)
                tdb = "%5.1f%%" % (16 * n * bpe / timed(bd.run, 20) / 8e10)
                bdn = _lib.Batch([(x, o, None, plan, 10.0, n // G, G, True) for x, o in zip(xs, outs)], dynamic=True)
                tdb += " (scales not stored: %5.1f%%)" % (16 * n * bpe / timed(bdn.run, 20) / 8e10)
            except _lib.AntqError:
                tdb = "   n/a"
            alt = []
            for knob in (1, 8):          # A/B of the dynamic row variants: knob 0 = 1 -> rows of <= 256 vectors through the per-row
                _lib.lib().antq_debug_set(0, knob)   # table kernel; = 8 -> rows of 257..512 vectors in ONE wavefront (8 per lane)
                try:
                    bd2 = _lib.Batch([(x, o, torch.empty_like(a), plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)],
                                     dynamic=True)
                    alt.append("%5.1f%%" % (16 * n * bpe / timed(bd2.run, 20) / 8e10))
                except _lib.AntqError:
                    alt.append("   n/a")
                _lib.lib().antq_debug_set(0, 0)
            # A/B on the same box: knob 4 = 0 turns the approximate-quotient element path off (exact division per element;
            # groups of 16 / 32 / 64 vectors then use per-group x-domain tables in the batched launch)
            _lib.lib().antq_debug_set(4, 0)
            bt0 = _lib.Batch([(x, o, a, plan, 10.0, n // G, G, True) for x, a, o in zip(xs, al, outs)])
            tb0 = timed(bt0.run, 20)
            tp0 = timed(lambda: [_lib.fakequant(x, a, plan, 10.0, n // G, G, True, out=o) for x, a, o in zip(xs, al, outs)], 5)
            _lib.lib().antq_debug_set(4, 1)
            print("%-9s group-%-5d static: batched %5.1f%%  per tensor %5.1f%%   dynamic: batched %s  per tensor %5.1f%%  of 8 TB/s"
                  "   [exact-division path: batched %5.1f%%  per tensor %5.1f%%]   [dynamic batched, other row variants: %s]" % (
                str(dt)[6:], G, 16 * n * bpe / tb / 8e10, 16 * n * bpe / tp / 8e10, tdb, 16 * n * bpe / td / 8e10,
                16 * n * bpe / tb0 / 8e10, 16 * n * bpe / tp0 / 8e10, " ".join(alt)), flush=True)
        del xs, outs


if __name__ == "__main__":
    main()
