#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fake-quant hot path on MI355X.

Metric (BASELINE.json): Gelements/s of quantize->dequantize, 4-bit ANT, plus the achieved HBM
bandwidth of the dominant kernel against the 8 TB/s roofline.

Workload (config.workload): `nbuf` distinct 4096x4096 bf16 weight tensors per GPU (synthetic
randn*0.02, seed 6+rank), signed 4-bit flint grid, calibrated per-row alpha (= row abs-max), steady
state Quantizer._forward (ant_quantization/antquant/quant_modules.py:535-551) through the C ABI.
One STEP = one pass over all `nbuf` tensors = ONE launch of the batched entry point
(antq_fakequant_batch); the set (nbuf x 33.5 MB in, same out) is far larger than the 256 MB
Infinity Cache, so everything streams from / to HBM.  The same pass issued as one launch per
tensor (antq_fakequant, the reference's granularity) is timed too and reported in
config.per_tensor_launches.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns its own
`nbuf` tensors, there is no data-path collective; ranks only meet at the barriers that
bracket the timed region and at the MAX-reduction of the elapsed time.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes of one launch (4 B/elem:
read bf16 + write bf16, x nbuf x 16.7 M elements) / average duration of that launch, measured
here with HIP events recorded on the launch stream around the timed region (which consists of
exactly `steps` launches of that kernel).  `cpu_baseline` times the CPU
oracle (oracle/antq_oracle.c, a literal restatement of the reference's op sequence: "port") on all
host cores (and on one) on rows of the same workload.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ROWS = COLS = 4096
BYTES_PER_ELEM = 4               # algorithmic: read one bf16 + write one bf16


def cpu_baseline(seconds_budget=12.0):
    """Oracle (port of the reference op sequence) on the host cores: 64 rows of 4096 bf16 elements per hardware thread
    (16384 rows = four headline tensors on a 256-thread host), every thread sweeping its rows `reps` times so that
    thread start-up does not count; about `seconds_budget` CPU-seconds in total.  Plus the same oracle on one thread."""
    import numpy as np
    from oracle import antq_oracle as orc
    from ant_quantization_amd import grids
    orc.build()
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(6)
    rows = 64 * cores
    x = orc.f32_to_bf16((rng.standard_normal((rows, COLS)) * 0.02).astype(np.float32))
    out = np.empty_like(x)
    g = grids.ant_flint(4, True)
    alpha = orc.absmax(orc.bf16_to_f32(x), True, 1.0)
    t0 = time.perf_counter()
    orc.forward_rows(x, out, 0, 64, alpha, g, 10.0)               # ONE host thread, 64 rows
    one_thread = 64 * COLS / (time.perf_counter() - t0)           # elements / s
    reps = max(1, int(seconds_budget * one_thread / (rows * COLS)))

    def work(b, e):
        for _ in range(reps):
            orc.forward_rows(x, out, b, e, alpha, g, 10.0)        # ctypes call: releases the GIL

    ts = [threading.Thread(target=work, args=(64 * t, 64 * (t + 1))) for t in range(cores)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(rows * COLS * reps / dt / 1e9, 5), "unit": "Gelem/s", "cores": cores, "kind": "port",
            "single_thread_gelem_per_s": round(one_thread / 1e9, 6),
            "sample": "%d rows x %d cols bf16 (64 rows per thread), flint 4-bit per-row alpha, %d sweeps, %d threads, "
                      "%.1f s wall" % (rows, COLS, reps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nbuf", type=int, default=32, help="distinct 4096x4096 bf16 tensors per GPU (one step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ant_quantization_amd import _lib, grids

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU fallback (the CPU oracle is only the reported baseline)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torchrun
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # RCCL; only barriers + one MAX reduce
        dist.barrier()                      # builds the communicator now, so later barriers cost microseconds

    # ---- workload: resident in HBM before the timed region -----------------------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(6 + rank)
    plan = _lib.plan_for(grids.ant_flint(4, True))
    xs, alphas, outs = [], [], []
    for _ in range(args.nbuf):
        x = (torch.randn(ROWS, COLS, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
        xs.append(x)
        alphas.append(_lib.absmax(x, ROWS, COLS, per_row=True))           # calibrated alpha = row abs-max
        outs.append(torch.empty_like(x))
    torch.cuda.synchronize()

    # One STEP = one pass of the hot path over the batch = ONE launch of the batched entry point
    # (antq_fakequant_batch: every workgroup looks its tensor up in a resident descriptor table).
    batch = _lib.Batch([(xs[i], outs[i], alphas[i], plan, 10.0, ROWS, COLS, True) for i in range(args.nbuf)])
    assert not batch.singles

    def step():
        batch.run()

    def step_per_tensor():          # the reference's granularity: one launch per tensor (reported beside it)
        for i in range(args.nbuf):
            _lib.fakequant(xs[i], alphas[i], plan, 10.0, ROWS, COLS, True, out=outs[i])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup pass: the same work as one launch per tensor (k_fq_xrow, the reference's granularity).
    # Reported beside the headline; it also keeps the GPU busy for >= 0.3 s before anything is timed,
    # which is what it takes for an idle MI355X to reach steady clocks (the first ~50 ms of load run
    # up to 20 % slower: tools/probe_clock_ramp.py).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pt_launch_s, t_setup = None, time.perf_counter()
    while time.perf_counter() - t_setup < 0.3:
        ev0.record()
        for _ in range(5):
            step_per_tensor()
        ev1.record()
        torch.cuda.synchronize()
        pt_launch_s = ev0.elapsed_time(ev1) * 1e-3 / (5 * args.nbuf)      # keep the last (steady-clock) reading

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ev0.record()                                  # HIP events on the launch stream (= torch's current stream)
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    # ---- roofline of the dominant kernel (antq::k_fq_batch): the timed region is nothing but
    # back-to-back launches of it on one stream, so its average launch duration = event time / launches.
    launch_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    algo_bytes = args.nbuf * ROWS * COLS * BYTES_PER_ELEM
    achieved = algo_bytes / launch_s / 1e9

    # ---- parity spot check of what was just measured (cheap, outside the timed region) --------
    ok = bool(torch.equal(_lib.fakequant(outs[0], alphas[0], plan, 10.0, ROWS, COLS, True), outs[0]))

    if use_dist:
        dist.barrier()
    if rank != 0:
        dist.destroy_process_group()
        return

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # rocprofv3 --pmc result, see profiles/README.md
    if os.path.exists(tpath):
        try:
            per_tensor = json.load(open(tpath)).get("k_fq_batch_bf16_bytes_per_tensor")
            traffic = int(per_tensor * args.nbuf) if per_tensor else None
        except Exception:
            traffic = None

    total_elems = world * args.nbuf * ROWS * COLS * args.steps
    res = {
        "metric": "Gelements/s quant-dequant + achieved HBM GB/s %peak, 4-bit ANT, 1/8 MI355X",
        "value": round(total_elems / elapsed / 1e9, 3),
        "unit": "Gelem/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "headline: %d x [4096,4096] bf16 weight tensors per GPU, ANT 4-bit signed flint grid, "
                               "calibrated per-row alpha, steady-state _forward; one step = ONE batched launch "
                               "(antq_fakequant_batch) over all %d tensors" % (args.nbuf, args.nbuf),
                   "io_dtype": "bf16", "elements_per_step_per_gpu": args.nbuf * ROWS * COLS,
                   "sharding": "independent tensors per rank, no data-path collective",
                   "idempotence_check": ok,
                   "per_tensor_launches": {"kernel": "antq::k_fq_xrow<bf16,...,U=4> (antq_fakequant, one launch per tensor)",
                                           "launch_us": round(pt_launch_s * 1e6, 2),
                                           "gelem_per_s": round(ROWS * COLS / pt_launch_s / 1e9, 1),
                                           "achieved_GBps": round(ROWS * COLS * BYTES_PER_ELEM / pt_launch_s / 1e9, 1),
                                           "frac": round(ROWS * COLS * BYTES_PER_ELEM / pt_launch_s / 1e9 / HBM_PEAK_GBPS, 4)}},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "kernel": "antq::k_fq_batch<bf16,false>", "launch_us": round(launch_s * 1e6, 2),
                     "algorithmic_bytes_per_launch": algo_bytes},
    }
    if world == 1 and not args.no_cpu_baseline:      # reported baseline, N=1 only
        res["cpu_baseline"] = cpu_baseline()
    print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
