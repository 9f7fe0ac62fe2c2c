#!/usr/bin/env python3
"""A few launches of every hot kernel (for rocprofv3 --pmc passes; see tools/profile_counters.sh).  Sizes are large enough
for steady-state counters (16 M elements per launch or more) and few enough launches for the serialised PMC passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ant_quantization_amd import _lib, core, grids  # noqa: E402

dev = torch.device("cuda:0")
REPS = 3


def main():
    torch.manual_seed(0)
    flint = _lib.plan_for(grids.ant_flint(4, True))
    int8 = _lib.plan_for(grids.ant_int(8, True))
    ol = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
    n = 4096 * 4096
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).bfloat16() for _ in range(8)]
    outs = [torch.empty_like(x) for x in xs]
    xf = [torch.randn(4096, 4096, device=dev) * 0.02 for _ in range(4)]
    of = [torch.empty_like(x) for x in xf]

    def batch(ts, os_, G, plan, gmax, ovp=False, dynamic=False):
        al = [_lib.absmax(x, n // G, G) for x in ts]
        if ovp:
            al = [a * 0.25 for a in al]
        return _lib.Batch([(x, o, a, plan, gmax, n // G, G, True) for x, o, a in zip(ts, os_, al)], ovp=ovp, dynamic=dynamic), al

    runs = []
    b, al = batch(xs, outs, 4096, flint, 10.0)
    runs.append(b.run)                                                                   # k_fq_batch<bf16,false>: headline
    runs.append(lambda: [_lib.fakequant(x, a, flint, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xs, al, outs)])   # k_fq_lane
    runs.append(lambda: [_lib.fakequant(x, a, flint, 10.0, 4096, 4096, True, out=o, unordered=True) for x, a, o in zip(xs, al, outs)])   # k_fq_xrow, 1 wavefront / workgroup (unordered launches)
    runs.append(lambda: [_lib.moments(x, 4096, 4096, True) for x in xs])                # k_moments (OliVe 3-sigma statistic), per row
    runs.append(lambda: [_lib.moments(x, 4096, 4096, False) for x in xs])               # ... per tensor
    # rows that are not a power of two of vectors (ResNet's 3x3 rows: 4608 = 576 bf16 vectors): the per-row table kernels
    x46 = [x.view(-1)[:3640 * 4608].view(3640, 4608) for x in xs]
    o46 = [o.view(-1)[:3640 * 4608].view(3640, 4608) for o in outs]
    a46 = [_lib.absmax(x, 3640, 4608) for x in x46]
    runs.append(_lib.Batch([(x, o, a, flint, 10.0, 3640, 4608, True) for x, o, a in zip(x46, o46, a46)]).run)          # k_fq_batch<bf16,false>
    runs.append(lambda: [_lib.fakequant(x, a, flint, 10.0, 3640, 4608, True, out=o) for x, a, o in zip(x46, a46, o46)])  # k_fq_xrow
    b16, al16 = batch(xs, outs, 16, flint, 10.0)
    runs.append(b16.run)                                                                 # k_fq_batch_d<bf16,.,AD>: group-16
    runs.append(lambda: [_lib.fakequant(x, a, flint, 10.0, n // 16, 16, True, out=o) for x, a, o in zip(xs, al16, outs)])  # k_fq_lane
    runs.append(batch(xs, outs, 16, flint, 10.0, dynamic=True)[0].run)                   # ... dynamic
    runs.append(batch(xs, outs, 256, flint, 10.0)[0].run)                                # group-256: lane jobs
    runs.append(batch(xs, outs, 4096, ol, 32.0, ovp=True)[0].run)                        # k_fq_batch<bf16,true>: OliVe pairs
    runs.append(batch(xs, outs, 4096, flint, 10.0, dynamic=True)[0].run)                 # k_fq_batch_dyn
    xl = [x.view(-1)[:512 * 28672].view(512, 28672) for x in xs[:4]]
    ol_ = [o.view(-1)[:512 * 28672].view(512, 28672) for o in outs[:4]]
    bl = _lib.Batch([(x, o, torch.empty(512, device=dev), flint, 10.0, 512, 28672, True) for x, o in zip(xl, ol_)], dynamic=True)
    runs.append(bl.run)                                                                  # k_fq_batch_dyn16: 28 672-wide rows
    runs.append(batch(xf, of, 16, flint, 10.0)[0].run)                                   # fp32 group-16
    a8 = [_lib.absmax(x, 4096, 4096) for x in xs[:4]]
    runs.append(lambda: [_lib.fakequant(x, a, int8, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xs[:4], a8, outs)])  # k_fq_uniform
    # calibration
    ratios = core._ratios(75, 150, 1, dev)
    xm = core.row_absmax(xf[0], True)
    runs.append(lambda: _lib.search_sse(xf[0], 4096, 4096, xm, True, ratios, flint, 10.0))               # k_search_sse
    plans = [_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "flint", "pot")]
    runs.append(lambda: _lib.search_sse_multi(xf[0], 4096, 4096, xm, True, ratios, plans, [10.0] * 3))   # k_search_sse_multi
    act = torch.nn.functional.gelu(torch.randn(64, 128, 3072, device=dev))
    am = core.row_absmax(act, False)
    runs.append(lambda: _lib.search_sse(act, 1, act.numel(), am, False, ratios, flint, 10.0))            # per-tensor (PT)
    act16 = act.to(torch.bfloat16)                                                                       # round 5: the histogram path
    am16 = core.row_absmax(act16, False)
    hist = [lambda: _lib.search_sse_multi(act16, 1, act16.numel(), am16, False, ratios, plans, [10.0] * 3)]         # k_hist16 / _reduce / _score
    hist.append(lambda: _lib.calibrate(act16, 1, act16.numel(), False, plans, [10.0] * 3, 75, 150, 1))                # k_hist16<.,false,true>: abs-max on the way
    op = [_lib.plan_for(np.concatenate([grids.olive_grid(t, 4, True), grids.olive_outliers(4, True)])) for t in ("int", "flint")]
    ogm = [float(grids.olive_grid(t, 4, True).max()) for t in ("int", "flint")]
    xo = torch.randn(64 * 128 * 3072, device=dev) * 0.05
    mo = torch.rand(xo.numel(), device=dev) < 0.003
    xo[mo] *= torch.empty(int(mo.sum()), device=dev).uniform_(8, 60)
    xo = xo.to(torch.bfloat16)
    hist.append(lambda: _lib.calibrate(xo, 1, xo.numel(), False, op, ogm, 75, 250, 2, xmax="3sigma", ovp=True))       # the pair-rule variants
    runs += hist
    # packed 4-bit codec (fp32 and bf16, OliVe pairs)
    gn = grids.olive_flint(4, True).size
    codec = []
    for x in (xf[0], xs[0]):
        a = _lib.absmax(x, 4096, 4096) * 0.25
        codes = _lib.encode4(x, a, ol, 32.0, 4096, 4096, True, n_normal=gn, ovp=True)
        codec.append(lambda x=x, a=a: _lib.encode4(x, a, ol, 32.0, 4096, 4096, True, n_normal=gn, ovp=True))      # k_encode4
        codec.append(lambda x=x, a=a, c=codes: _lib.decode4(c, a, ol, 32.0, 4096, 4096, True, x.dtype, n_normal=gn, ovp=True))
    a_ant = _lib.absmax(xs[0], 4096, 4096)
    codec.append(lambda: _lib.encode4(xs[0], a_ant, flint, 10.0, 4096, 4096, True))                     # k_encode4_hrow, ANT
    runs += codec
    if os.environ.get("ANTQ_TARGETS") == "codec":
        runs = codec
    if os.environ.get("ANTQ_TARGETS") == "hist":
        runs = hist
    for r in runs:
        for _ in range(REPS):
            r()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
