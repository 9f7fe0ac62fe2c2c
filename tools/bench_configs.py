#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs (C1-C3 + variants of the headline) on one MI355X.

Not the contract benchmark (that is bench.py); this script produces the per-config table quoted in
DESIGN.md.  Synthetic data of the stated shapes, everything resident before timing,
graph-free plain launches on the current stream, HIP events around `reps` passes over the config.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ant_quantization_amd import _lib, core, grids  # noqa: E402

dev = torch.device("cuda:0")


def resnet50_shapes():
    s = [(64, 3, 7, 7)]
    inp = 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            s += [(planes, inp, 1, 1), (planes, planes, 3, 3), (planes * 4, planes, 1, 1)]
            if b == 0:
                s.append((planes * 4, inp, 1, 1))
            inp = planes * 4
    s.append((1000, 2048))
    return s


def bert_base_shapes():
    s = []
    for _ in range(12):
        s += [(768, 768)] * 4 + [(3072, 768), (768, 3072)]
    s += [(768, 768), (2, 768)]
    return s


def opt67_shapes(layers=32):
    s = []
    for _ in range(layers):
        s += [(4096, 4096)] * 4 + [(16384, 4096), (4096, 16384)]
    return s


def llama70b_shapes(layers=80):
    """C4: synthetic 70B-parameter Linear stack: 80 x {[8192,8192] x2, [1024,8192] x2, [28672,8192] x2, [8192,28672]}."""
    s = []
    for _ in range(layers):
        s += [(8192, 8192)] * 2 + [(1024, 8192)] * 2 + [(28672, 8192)] * 2 + [(8192, 28672)]
    return s


def timed(fn, reps):
    """Seconds per call at steady clocks: warm up for >= 60 ms of GPU time first (an idle MI355X ramps for ~50 ms,
    tools/probe_clock_ramp.py), then time max(reps, enough calls for >= 40 ms) back-to-back calls with HIP events."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    once = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
    for _ in range(min(20000, int(0.06 / once) + 1)):
        fn()
    reps = max(reps, min(20000, int(0.04 / once) + 1))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def report(name, elems, bytes_per_elem, secs, launches, extra=""):
    print("%-58s %9.1f Gelem/s %7.3f TB/s (%4.1f%% of 8)  %8.1f us/pass %4d launches %s" % (
        name, elems / secs / 1e9, elems * bytes_per_elem / secs / 1e12, elems * bytes_per_elem / secs / 8e10,
        secs * 1e6, launches, extra), flush=True)


def main():
    torch.manual_seed(1)
    out = {}
    # ---------------- C1: ResNet-50 weights, ANT 4-bit flint: per-channel and group-16, fp32
    plan = _lib.plan_for(grids.ant_flint(4, True))
    ws = [torch.randn(*s, device=dev) * float(np.sqrt(2.0 / (s[0] * np.prod(s[2:], dtype=np.int64)))) for s in resnet50_shapes()]
    elems = sum(w.numel() for w in ws)
    outs = [torch.empty_like(w) for w in ws]
    a_pc = [_lib.absmax(w, w.shape[0], w.numel() // w.shape[0]) for w in ws]
    a_g16 = [_lib.absmax(w, w.numel() // 16, 16) for w in ws]

    def c1_pc():
        for w, a, o in zip(ws, a_pc, outs):
            _lib.fakequant(w, a, plan, 10.0, w.shape[0], w.numel() // w.shape[0], True, out=o)

    def c1_g16():
        for w, a, o in zip(ws, a_g16, outs):
            _lib.fakequant(w, a, plan, 10.0, w.numel() // 16, 16, True, out=o)

    def c1_g16_dyn():
        for w, o in zip(ws, outs):
            _lib.fakequant_dynamic(w, plan, 10.0, w.numel() // 16, 16, out=o, want_alpha=False)

    report("C1 ResNet-50 54 W, flint4 per-channel, fp32 (static alpha)", elems, 8, timed(c1_pc, 20), len(ws))
    report("C1 ResNet-50 54 W, flint4 group-16, fp32 (static alpha)", elems, 8, timed(c1_g16, 20), len(ws))
    report("C1 ResNet-50 54 W, flint4 group-16, fp32 (dynamic abs-max)", elems, 8, timed(c1_g16_dyn, 20), len(ws))
    for nm, al_, rl in (("per-channel", a_pc, None), ("group-16", a_g16, 16)):
        jobs = []
        for w, a, o in zip(ws, al_, outs):
            rows, K = (w.shape[0], w.numel() // w.shape[0]) if rl is None else (w.numel() // 16, 16)
            jobs.append((w, o, a, plan, 10.0, rows, K, True))
        bt = _lib.Batch(jobs)
        report("C1 ResNet-50 54 W, flint4 %s, fp32, BATCHED launch" % nm, elems, 8, timed(bt.run, 50),
               1 + len(bt.singles))
        _lib.lib().antq_debug_set(0, 4)      # A/B: lane jobs with 4 vectors per lane as in big batches (default here: 2)
        bt4 = _lib.Batch(jobs)
        _lib.lib().antq_debug_set(0, 0)
        report("   (the same with 4-vector lane tasks)", elems, 8, timed(bt4.run, 50), 1 + len(bt4.singles))
    # one big concatenated buffer: what a multi-tensor launch could reach for group-16
    flat = torch.cat([w.reshape(-1) for w in ws]).contiguous()
    fo = torch.empty_like(flat)
    af = _lib.absmax(flat, flat.numel() // 16, 16)
    report("   (same bytes as ONE group-16 launch over a flat buffer)", elems, 8,
           timed(lambda: _lib.fakequant(flat, af, plan, 10.0, flat.numel() // 16, 16, True, out=fo), 50), 1)
    # ... and what a plain copy of the same bytes reaches in a launch this short (the ceiling of a 33 us pass)
    report("   (the copy kernel over the same flat buffer, antq_copy)", elems, 8, timed(lambda: _lib.copy(flat, fo), 50), 1)
    del ws, outs, flat, fo

    # ---------------- C2: BERT-base Linear weights: steady state + calibration (ant-int-pot-flint)
    ws = [torch.randn(*s, device=dev) * 0.02 for s in bert_base_shapes()]
    elems = sum(w.numel() for w in ws)
    outs = [torch.empty_like(w) for w in ws]
    al = [_lib.absmax(w, w.shape[0], w.shape[1]) for w in ws]
    report("C2 BERT-base 74 Linear W, flint4 per-channel, fp32", elems, 8,
           timed(lambda: [_lib.fakequant(w, a, plan, 10.0, w.shape[0], w.shape[1], True, out=o) for w, a, o in zip(ws, al, outs)], 20), len(ws))
    bt = _lib.Batch([(w, o, a, plan, 10.0, w.shape[0], w.shape[1], True) for w, a, o in zip(ws, al, outs)])
    report("C2 BERT-base 74 Linear W, flint4 per-channel, fp32, BATCHED", elems, 8, timed(bt.run, 50), 1)
    plans = {t: _lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")}

    def c2_cal():
        for w, a in zip(ws, al):
            for t in ("int", "pot", "flint"):
                core.clip_search(w, a, True, 80, 150, 1, plans[t], 10.0)

    secs = timed(c2_cal, 2)
    evals = elems * 3 * 70
    print("%-58s %9.1f G candidate-evals/s  %8.1f ms/pass (3 types x 70 clip ratios, %d tensors)" % (
        "C2 BERT-base calibration (type select + clip search)", evals / secs / 1e9, secs * 1e3, len(ws)), flush=True)

    def c2_cal_multi():          # the same work as the quantiser issues it since round 2: all three types on ONE read
        for w, a in zip(ws, al):
            core.clip_search_types(w, a, True, 80, 150, 1, [plans[t] for t in ("int", "pot", "flint")], [10.0] * 3)

    secs = timed(c2_cal_multi, 2)
    print("%-58s %9.1f G candidate-evals/s  %8.1f ms/pass (the same, ONE launch and one read per tensor for all 3 types)" % (
        "   (antq_search_sse_multi)", evals / secs / 1e9, secs * 1e3), flush=True)
    big = torch.randn(4096, 4096, device=dev) * 0.02
    ab = _lib.absmax(big, 4096, 4096)
    secs = timed(lambda: core.clip_search(big, ab, True, 80, 150, 1, plans["flint"], 10.0), 3)
    print("%-58s %9.1f G candidate-evals/s  %8.2f ms (one 4096x4096 fp32 tensor, 70 clip ratios)" % (
        "clip search on a large tensor", big.numel() * 70 / secs / 1e9, secs * 1e3), flush=True)
    secs = timed(lambda: core.clip_search_types(big, ab, True, 80, 150, 1, [plans[t] for t in ("int", "pot", "flint")], [10.0] * 3), 3)
    print("%-58s %9.1f G candidate-evals/s  %8.2f ms (the same tensor, 3 types x 70 ratios on one read)" % (
        "   (antq_search_sse_multi)", big.numel() * 210 / secs / 1e9, secs * 1e3), flush=True)
    del big
    x = torch.nn.functional.gelu(torch.randn(64, 128, 3072, device=dev))
    pu = _lib.plan_for(grids.ant_flint(4, True))
    ax = x.abs().max().reshape(1)
    ox = torch.empty_like(x)
    report("C2 activation [64,128,3072] fp32, per-tensor flint4", x.numel(), 8,
           timed(lambda: _lib.fakequant(x, ax, pu, 10.0, 1, x.numel(), False, out=ox), 20), 1)
    # QAT backward w.r.t. alpha (N3): sum gout * (out - x) per row / per tensor -- three reads per element, no write
    for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2)):
        xg = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(8)]
        og = [t + (torch.randn_like(t.float()) * 0.001).to(dt) for t in xg]
        gg = [torch.randn(4096, 4096, device=dev).to(dt) for _ in range(8)]
        for per_row in (True, False):
            secs = timed(lambda: [_lib.alpha_grad(a, b, c, 4096, 4096, per_row=per_row) for a, b, c in zip(xg, og, gg)], 5)
            byt = 8 * 4096 * 4096 * 3 * esz
            print("%-58s %9.1f Gelem/s  %6.3f TB/s (%4.1f%% of 8)  %8.1f us/pass   %2d launches" % (
                "alpha gradient %s, %s, 8 x 4096^2" % (str(dt)[6:], "per row" if per_row else "per tensor"),
                8 * 4096 * 4096 / secs / 1e9, byt / secs / 1e12, byt / secs / 8e10, secs * 1e6, 8 if per_row else 16), flush=True)
        del xg, og, gg
    del ws, outs, x, ox

    # ---------------- C3: OPT-6.7B weights (4 of 32 layers resident), OliVe flint4 + outliers, OVP
    gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
    pol = _lib.plan_for(np.concatenate([gn, go]))
    for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
        ws = []
        for s in opt67_shapes(4 if dt == torch.bfloat16 else 2):
            w = torch.randn(*s, device=dev) * 0.02
            m = torch.rand_like(w) < 0.001
            w[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 64)
            ws.append(w.to(dt))
        elems = sum(w.numel() for w in ws)
        al = [(3 * w.float().std(1)).contiguous() for w in ws]
        outs = [torch.empty_like(w) for w in ws]
        report("C3 OPT-6.7B W (%d tensors), OliVe flint4 OVP, %s" % (len(ws), str(dt)[6:]), elems, bpe,
               timed(lambda: [_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], True, ovp=True, out=o) for w, a, o in zip(ws, al, outs)], 5), len(ws))
        report("C3 the same, launches marked UNORDERED (weights at rest)", elems, bpe,
               timed(lambda: [_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], True, ovp=True, out=o, unordered=True) for w, a, o in zip(ws, al, outs)], 5), len(ws))
        bt = _lib.Batch([(w, o, a, pol, 32.0, w.shape[0], w.shape[1], True) for w, a, o in zip(ws, al, outs)], ovp=True)
        report("C3 OPT-6.7B W (%d tensors), OliVe flint4 OVP, %s, BATCHED" % (len(ws), str(dt)[6:]), elems, bpe,
               timed(bt.run, 5), 1)
        if dt == torch.bfloat16:
            # OliVe's calibration of these tensors (OQ:189-256): x_max = max|mean +- 3 std| per row, then for int and flint
            # (+ outliers, pair rule) the clip search over range(75, 250, 2) -- 88 candidates x 2 types -- and the type pick.
            # "reference statistic": t.mean(1) / t.std(1) as torch ops (what OQ:193-197 runs: >= 3 reads of the tensor);
            # "one read": antq_moments + antq_xmax_3sigma.  The search is ours in both (the reference's: 176 full passes).
            gi = np.concatenate([grids.olive_int(4, True), go])
            pls, gms = [_lib.plan_for(gi), pol], [float(grids.olive_int(4, True).max()), 32.0]
            def stat_ref():
                for w in ws:
                    mu, sd = w.mean(1), w.std(1)
                    torch.maximum((mu + 3 * sd).abs(), (mu - 3 * sd).abs())
            def stat_one():
                for w in ws:
                    _lib.xmax_3sigma(w, w.shape[0], w.shape[1], per_row=True)
            def full():
                for w in ws:
                    _lib.calibrate(w, w.shape[0], w.shape[1], True, pls, gms, 75, 250, 2, xmax="3sigma", ovp=True)
            jobs = [(w, w.shape[0], w.shape[1], True, pls, gms, 75, 250, 2) for w in ws]
            t_batch = timed(lambda: _lib.calibrate_batch(jobs, xmax="3sigma", ovp=True), 2)
            t_ref, t_one, t_full = timed(stat_ref, 3), timed(stat_one, 3), timed(full, 2)
            evals = elems * 88 * 2
            scale = 192 / len(ws)
            print("%-58s %8.2f ms reference statistic (torch mean / std), %6.2f ms on one read (antq_moments)" % (
                "C3 OPT-6.7B calibration, clip statistic, %d tensors" % len(ws), t_ref * 1e3, t_one * 1e3), flush=True)
            print("%-58s %8.1f ms = %6.1f G candidate-evals/s (statistic + 2 types x 88 ratios + picks, antq_calibrate); x %d for the 192 tensors: %.0f ms" % (
                "C3 OPT-6.7B calibration, whole (these %d tensors)" % len(ws), t_full * 1e3, evals / t_full / 1e9, int(scale), t_full * 1e3 * scale), flush=True)
            print("%-58s %8.1f ms; x %d for the 192 tensors: %.0f ms" % ("   the same in ONE C call (antq_calibrate_batch)", t_batch * 1e3, int(scale), t_batch * 1e3 * scale), flush=True)
        del ws, outs, al, bt

    # ---------------- C4: 70B-parameter bf16 Linear stack, OliVe flint4 OVP: this rank's 1/8 share (LPT by bytes)
    from ant_quantization_amd import sharding
    shapes = llama70b_shapes()
    share = sharding.lpt_assign([2 * a * b for a, b in shapes], 8)[0]
    gen = torch.Generator(device=dev).manual_seed(5)
    ws = []
    for i in share:
        w = torch.randn(*shapes[i], device=dev, generator=gen) * 0.02
        m = torch.rand(w.shape, device=dev, generator=gen) < 0.001
        w[m] *= torch.empty(int(m.sum()), device=dev).uniform_(8, 64, generator=gen)
        ws.append(w.to(torch.bfloat16))
        del w, m
    elems = sum(w.numel() for w in ws)
    al = [(3 * w.float().std(1)).contiguous() for w in ws]
    outs = [torch.empty_like(w) for w in ws]
    bt = _lib.Batch([(w, o, a, pol, 32.0, w.shape[0], w.shape[1], True) for w, a, o in zip(ws, al, outs)], ovp=True)
    report("C4 70B bf16 W, rank 0 of 8 (%d tensors, %.1f GB in), OliVe flint4 OVP, BATCHED" % (len(ws), elems * 2 / 1e9),
           elems, 4, timed(bt.run, 5), 1)
    report("C4 same share, one launch per tensor", elems, 4,
           timed(lambda: [_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], True, ovp=True, out=o) for w, a, o in zip(ws, al, outs)], 3), len(ws))
    report("C4 same share, one launch per tensor, UNORDERED", elems, 4,
           timed(lambda: [_lib.fakequant(w, a, pol, 32.0, w.shape[0], w.shape[1], True, ovp=True, out=o, unordered=True) for w, a, o in zip(ws, al, outs)], 3), len(ws))
    del ws, outs, al, bt
    torch.cuda.empty_cache()

    # ---------------- headline variants
    for dt, bpe in ((torch.bfloat16, 4), (torch.float32, 8)):
        nb = 16
        xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(nb)]
        al = [_lib.absmax(x, 4096, 4096) for x in xs]
        outs = [torch.empty_like(x) for x in xs]
        report("headline 4096x4096 %s flint4 per-row (static)" % str(dt)[6:], nb * 4096 * 4096, bpe,
               timed(lambda: [_lib.fakequant(x, a, plan, 10.0, 4096, 4096, True, out=o) for x, a, o in zip(xs, al, outs)], 10), nb)
        report("headline 4096x4096 %s flint4 per-row (static), UNORDERED launches" % str(dt)[6:], nb * 4096 * 4096, bpe,
               timed(lambda: [_lib.fakequant(x, a, plan, 10.0, 4096, 4096, True, out=o, unordered=True) for x, a, o in zip(xs, al, outs)], 10), nb)
        report("headline 4096x4096 %s flint4 per-row (dynamic abs-max)" % str(dt)[6:], nb * 4096 * 4096, bpe,
               timed(lambda: [_lib.fakequant_dynamic(x, plan, 10.0, 4096, 4096, out=o, want_alpha=False) for x, o in zip(xs, outs)], 10), nb)
        bt = _lib.Batch([(x, o, a, plan, 10.0, 4096, 4096, True) for x, a, o in zip(xs, al, outs)])
        report("headline 16 x 4096x4096 %s, BATCHED (one launch)" % str(dt)[6:], nb * 4096 * 4096, bpe, timed(bt.run, 10), 1)
        report("copy (antq_copy) same buffers %s" % str(dt)[6:], nb * 4096 * 4096, bpe,
               timed(lambda: [_lib.copy(x, o) for x, o in zip(xs, outs)], 10), nb)
        del xs, outs


if __name__ == "__main__":
    main()
