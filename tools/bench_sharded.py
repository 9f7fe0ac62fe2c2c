#!/usr/bin/env python3
"""BASELINE configs 3 / 4 end to end: a whole model's Linear weights, OliVe 4-bit flint + outlier-victim pairs, sharded
over the ranks of one node by bytes (sharding.lpt_assign), every rank quantising its own tensors in ONE batched launch.
No data-path collective: ranks meet at the barriers around the timed region and at one MAX reduction of the time.

    python tools/bench_sharded.py --model opt6.7b                       # 1 GPU: all 192 tensors (12.9 GB in, 12.9 GB out)
    python tools/bench_sharded.py --model llama70b --inplace            # 1 GPU: 560 tensors, 137 GB, quantised in place
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_sharded.py --model llama70b

Synthetic weights (randn * 0.02, 0.1 % of the entries multiplied by U(8, 64)), alpha = 3 sigma per row (OQ:193-197)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["opt6.7b", "llama70b"], default="opt6.7b")
    ap.add_argument("--inplace", action="store_true", help="out = x (halves the footprint)")
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--knobs", default="", help="dev: comma-separated key=value pairs for antq_debug_set before the batch is built")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from ant_quantization_amd import _lib, grids, sharding
    from bench_configs import llama70b_shapes, opt67_shapes

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or "MASTER_PORT" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        dist.barrier()
    multi = dist.is_initialized()
    shapes = opt67_shapes(32) if args.model == "opt6.7b" else llama70b_shapes(80)
    mine = sharding.lpt_assign([2 * a * b for a, b in shapes], world)[rank]
    gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
    plan = _lib.plan_for(np.concatenate([gn, go]))
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    ws, alphas = [], []
    for i in mine:
        w = torch.randn(*shapes[i], device=dev, dtype=torch.bfloat16, generator=gen) * 0.02
        m = torch.rand(w.shape, device=dev, generator=gen) < 0.001
        w[m] *= torch.empty(int(m.sum()), device=dev, dtype=torch.bfloat16).uniform_(8, 64, generator=gen)
        del m
        ws.append(w)
        alphas.append(_lib.xmax_3sigma(w, w.shape[0], w.shape[1], per_row=True))      # OQ:193-197 on one read (antq_moments)
    for kv in [k for k in args.knobs.split(",") if k]:
        _lib.lib().antq_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    outs = ws if args.inplace else [torch.empty_like(w) for w in ws]
    elems = sum(w.numel() for w in ws)
    bt = _lib.Batch([(w, o, a, plan, 32.0, w.shape[0], w.shape[1], True) for w, o, a in zip(ws, outs, alphas)], ovp=True)
    # warm-up: at least 2 passes and at least 100 ms of them (an MI355X that was idle -- or busy with the short kernels of the
    # data generation above -- needs ~50 ms of load to reach steady clocks: tools/probe_clock_ramp.py); then at least
    # `--passes` passes and at least 200 ms of them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bt.run()
    torch.cuda.synchronize()
    once = max(time.perf_counter() - t0, 1e-6)
    for _ in range(max(1, int(0.1 / once) + 1)):
        bt.run()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    passes = max(args.passes, int(0.2 / once) + 1)
    t0 = time.perf_counter()
    for _ in range(passes):
        bt.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    tot = torch.tensor([float(elems)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"model": args.model, "n_gpus": world, "tensors_total": len(shapes), "tensors_rank0": len(mine),
                          "elements_total": int(tot.item()), "GB_resident_rank0": round(elems * (2 if args.inplace else 4) / 1e9, 1),
                          "ms_per_pass": round(tmax.item() * 1e3, 3), "Gelem_per_s": round(tot.item() / tmax.item() / 1e9, 1),
                          "frac_of_8TBps_per_gpu": round(tot.item() * 4 / tmax.item() / world / 8e12, 4),
                          "inplace": bool(args.inplace), "launches_per_pass_per_gpu": 1, "knobs": args.knobs}), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
