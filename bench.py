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
    python bench.py --gpus N ...                       (no launcher: starts the N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes of one launch (4 B/elem:
read bf16 + write bf16, x nbuf x 16.7 M elements) / average duration of that launch, measured
here with HIP events recorded on the launch stream around the timed region (which consists of
exactly `steps` launches of that kernel).  `roofline.copy_ceiling` is the empirical ceiling SURVEY 8d
asks for: a plain 16-byte-per-lane nontemporal copy kernel and hipMemcpyDtoD over the same buffers,
timed right after the timed region.  `cpu_baseline` times the CPU oracle (oracle/antq_oracle.c, a
literal restatement of the reference's op sequence: "port") with OpenMP on all host cores (and on
one); `cpu_baseline_torch` is the same op sequence as PyTorch CPU ops.

ANTQ_BENCH_SELFTEST=1 (tests/test_sharding_gloo.py): the rank / barrier / MAX-over-ranks / report
logic below runs unchanged on CPU over gloo with a stub in place of the GPU workload, so that the
multi-rank harness the driver launches at N = 2, 4, 8 is executed code and not only prose.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ROWS = COLS = 4096
BYTES_PER_ELEM = 4               # algorithmic: read one bf16 + write one bf16
METRIC = "Gelements/s quant-dequant + achieved HBM GB/s %peak, 4-bit ANT, 1/8 MI355X"


# ------------------------------------------------------------------------------------------------
# rank harness: one process per GPU, no data-path collective
# ------------------------------------------------------------------------------------------------
class Harness:
    """What every rank does around its own, independent work: join the job `torch.distributed.run` started, meet the
    others at a barrier on both sides of the timed region, agree on the slowest rank's time."""

    def __init__(self, gpus, selftest=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.selftest = selftest
        if self.world != gpus:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch N ranks with `python -m torch.distributed.run "
                     "--nnodes=1 --nproc-per-node N ... bench.py --gpus N` (one process per GPU) or run `python bench.py "
                     "--gpus N` without WORLD_SIZE set (it then starts the N ranks itself); refusing to report a number "
                     "for a job of another size" % (gpus, self.world))
        import torch
        self.torch = torch
        if selftest:
            self.dev = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                sys.exit("bench.py needs an MI355X; there is no CPU fallback (the CPU oracle is only the reported baseline)")
            if self.local_rank >= torch.cuda.device_count():
                sys.exit("bench.py: LOCAL_RANK %d but only %d visible GPU(s)" % (self.local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        if self.world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):      # launched by torchrun
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if selftest else "nccl", rank=self.rank, world_size=self.world)   # nccl = RCCL
            self.dist = dist
            self.barrier()                 # builds the communicator now, so later barriers cost microseconds

    def sync(self):
        if not self.selftest:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, value):
        """One float per rank, known to every rank (a one-hot vector summed over ranks: no collective beyond all_reduce)."""
        if self.dist is None:
            return [float(value)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev)
        t[self.rank] = float(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def timed(self, step, steps, warmup, on_start=None, on_stop=None):
        """`warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
        returns the MAX over ranks of the wall-clock time of the bracketed region."""
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        if on_start:
            on_start()
        for _ in range(steps):
            step()
        if on_stop:
            on_stop()
        self.sync()
        elapsed = time.perf_counter() - t0
        self.barrier()
        return self.max_over_ranks(elapsed)

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def per_rank_stats(values, digits=2):
    return {"min": round(min(values), digits), "mean": round(sum(values) / len(values), digits), "max": round(max(values), digits)}


def report(h, args, elems_per_step_all_ranks, elapsed, extra, scaling="weak"):
    """The one JSON line (rank 0).  value = units ALL ranks processed / the slowest rank's time."""
    total = elems_per_step_all_ranks * args.steps
    res = {
        "metric": METRIC,
        "value": round(total / elapsed / 1e9, 3),
        "unit": "Gelem/s",
        "n_gpus": h.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
    }
    res.update(extra)
    return res


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; a bounded sample of the same workload)
# ------------------------------------------------------------------------------------------------
def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo (None when it cannot be read)."""
    try:
        seen, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    seen.add((phys, line.split(":", 1)[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(seconds_budget=12.0, threads=None):
    """Oracle (port of the reference op sequence, oracle/antq_oracle.c) with OpenMP on every host core: 64 rows of 4096
    bf16 elements per hardware thread, swept `reps` times inside ONE parallel region (thread start-up is paid once,
    outside the sweeps that matter); about `seconds_budget` CPU-seconds per core.  Plus the same on one thread."""
    import numpy as np
    from oracle import antq_oracle as orc
    from ant_quantization_amd import grids
    orc.build()
    cores = threads or os.cpu_count() or 1
    rng = np.random.default_rng(6)
    rows = 64 * cores
    x = orc.f32_to_bf16((rng.standard_normal((rows, COLS)) * 0.02).astype(np.float32))
    out = np.empty_like(x)
    g = grids.ant_flint(4, True)
    alpha = orc.absmax(orc.bf16_to_f32(x), True, 1.0)
    t0 = time.perf_counter()
    orc.forward_omp(x[:64], out[:64], alpha[:64], g, 10.0, reps=1, threads=1)     # ONE host thread, 64 rows
    one_thread = 64 * COLS / (time.perf_counter() - t0)                            # elements / s
    orc.forward_omp(x, out, alpha, g, 10.0, reps=1, threads=cores)                 # spawns the OpenMP team
    t0 = time.perf_counter()
    orc.forward_omp(x, out, alpha, g, 10.0, reps=1, threads=cores)                 # one sweep, to size the sample: the
    sweep = time.perf_counter() - t0                                               # box may grant fewer cores than it shows
    reps = max(1, min(int(seconds_budget / max(sweep, 1e-4)), int(seconds_budget * one_thread / (64 * COLS))))
    t0 = time.perf_counter()
    used = orc.forward_omp(x, out, alpha, g, 10.0, reps=reps, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(rows * COLS * reps / dt / 1e9, 5), "unit": "Gelem/s", "cores": used, "kind": "port",
            "single_thread_gelem_per_s": round(one_thread / 1e9, 6), "cpu_model": _cpu_model(),
            "nproc": os.cpu_count(), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
            "sample": "%d rows x %d cols bf16 (64 rows per thread), flint 4-bit per-row alpha, %d sweeps, %d OpenMP "
                      "threads (static row split), %.1f s wall" % (rows, COLS, reps, used, dt)}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline_torch(seconds_budget=6.0):
    """The reference's own op sequence as PyTorch CPU ops (SURVEY 8d baseline 2; the reference has no CPU kernel, and its
    Python cannot travel to the GPU box): d = x / scale (row-broadcast); q = nearest grid value; t = (q - d) + d;
    out = t * scale (AQ:535-551), fp32 arithmetic on a bf16 tensor.  `nearest` = the stand-in for quant_cuda.quant
    written with torch ops (bucketize on the mid-points of the sorted grid: one pass, no [N, 16] distance matrix)."""
    import numpy as np
    import torch
    from ant_quantization_amd import grids
    g = torch.from_numpy(np.unique(grids.ant_flint(4, True)))
    mids = (g[:-1] + g[1:]) / 2
    x = (torch.randn(1024, COLS, generator=torch.Generator().manual_seed(6)) * 0.02).to(torch.bfloat16)
    alpha = x.float().abs().amax(1, keepdim=True)

    def fwd():
        xf = x.float()
        scale = alpha / 10.0
        d = xf / scale
        q = g[torch.bucketize(d, mids, right=True)]
        return (((q - d) + d) * scale).to(torch.bfloat16)

    fwd()
    best, t_all, n = 1e30, time.perf_counter(), 0
    while n < 5 or (time.perf_counter() - t_all < seconds_budget and n < 200):
        t0 = time.perf_counter()
        fwd()
        best = min(best, time.perf_counter() - t0)
        n += 1
    return {"value": round(x.numel() / best / 1e9, 5), "unit": "Gelem/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "PyTorch-CPU op sequence (div, bucketize-nearest, STE add, mul) on 1024 x %d bf16, "
                                      "best of %d, torch.get_num_threads() = %d" % (COLS, n, torch.get_num_threads())}


# ------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel, measured (rank 0, N = 1): two rocprofv3 --pmc child runs of this very file
# ------------------------------------------------------------------------------------------------
def measure_traffic(nbuf, kernels=("k_fq_hbatch",), timeout_s=240, child_args=()):
    """FETCH_SIZE and WRITE_SIZE of the batched kernel, one counter per pass as MI355X_MICROARCH.md's HBM section
    prescribes (`rocprofv3 --pmc <C> --kernel-trace`, nothing else), each pass a child `bench.py --traffic-child` that
    builds the same workload and issues 3 launches.  Units and the gfx950 correction as in that guide: both counters are
    KiB; FETCH_SIZE tallies the 128-byte requests of 16 B / lane streaming reads at 64 B and is doubled.
    `kernels`: regular expressions, one per kernel of interest (the child may launch several: the headline child also
    runs the OliVe twin of the batch); Returns ({pattern: bytes per launch}, {pattern: note}) or (None, why not)."""
    import re
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py is itself running under rocprofv3: no nested counter passes"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="antq_pmc_", dir="/tmp")
        try:
            # (the child is a plain one-GPU run on this rank's GPU, whatever launcher started us)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                                    "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                    "TORCHELASTIC_RUN_ID")}
            env["TMPDIR"] = "/tmp"
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--nbuf", str(nbuf)] + list(child_args)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [row for f in files for row in csv.DictReader(open(f)) if row.get("Counter_Name") == counter]
            for pat in kernels:
                got = [float(row["Counter_Value"]) for row in rows if re.search(pat, row.get("Kernel_Name", ""))]
                if not got:
                    return None, "rocprofv3 --pmc %s gave no row for %s (rc %d)" % (counter, pat, r.returncode)
                vals[(pat, counter)] = sum(got) / len(got)
        except Exception as ex:           # noqa: BLE001  (a missing profiler must not cost the bench its line)
            return None, "rocprofv3 --pmc %s failed: %r" % (counter, ex)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total, notes = {}, {}
    for pat in kernels:
        fs, ws = vals[(pat, "FETCH_SIZE")], vals[(pat, "WRITE_SIZE")]
        total[pat] = int(round((2.0 * fs + ws) * 1024.0))
        notes[pat] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate child passes of `bench.py "
                      "--traffic-child`, 3 launches each; FETCH_SIZE %.1f KiB x 2 per the guide's gfx950 correction + WRITE_SIZE "
                      "%.1f KiB)" % (fs, ws))
    return total, notes


# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves -- one fresh process per GPU, the same
    command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, exactly what
    `torch.distributed.run --nnodes=1 --nproc-per-node N` would have set -- and wait for them.  Rank 0 prints the JSON
    line on the inherited stdout.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def selftest_main(args):
    """Stub workload for the CPU test of the rank harness: rank r sleeps (r + 1) ms per step.  Everything around the
    workload is the real thing: the sharding of the named workload (host bookkeeping), barriers, MAX over ranks, the
    per-rank gather, the report."""
    h = Harness(args.gpus, selftest=True)
    scaling = "weak"
    if args.workload == "headline":
        units = 1000000
        cfg = {"workload": "stub: rank r sleeps (r+1) ms per step"}
    else:
        from ant_quantization_amd import sharding
        mine = sharding.shard_plan(args.workload, h.rank, h.world, layers=args.layers)
        units = sum((e - b) * c for _, b, e, c in mine)
        scaling = "strong"
        cfg = {"workload": "stub over the %s shard plan: rank r sleeps (r+1) ms per step" % args.workload,
               "units_of_rank0": len(mine)}

    def step():
        time.sleep(1e-3 * (h.rank + 1))

    elapsed = h.timed(step, args.steps, args.warmup)
    launch_us = h.gather(1e3 * (h.rank + 1))
    units_all = h.gather(units)
    if h.rank == 0:
        cfg["elements_per_step_per_rank"] = [int(u) for u in units_all]
        res = report(h, args, sum(units_all), elapsed, {
            "data": "stub", "selftest": True, "config": cfg,
            "roofline": {"bound": "hbm", "per_rank": {"launch_us": per_rank_stats(launch_us), "ranks": h.world}}}, scaling=scaling)
        print(json.dumps(res), flush=True)
    h.finish()


# ------------------------------------------------------------------------------------------------
# workloads: everything resident in HBM before the timed region; one STEP = one pass = ONE batched launch per rank
# ------------------------------------------------------------------------------------------------
def build_headline(args, h, _lib, grids):
    torch, dev, rank = h.torch, h.dev, h.rank
    gen = torch.Generator(device=dev)
    gen.manual_seed(6 + rank)
    plan = _lib.plan_for(grids.ant_flint(4, True))
    x_slab = torch.empty(args.nbuf, ROWS, COLS, dtype=torch.bfloat16, device=dev)      # contiguous: the copy ceiling below
    out_slab = torch.empty_like(x_slab)                                                # is measured on the very same bytes
    xs, alphas, outs = [], [], []
    for i in range(args.nbuf):
        x_slab[i] = (torch.randn(ROWS, COLS, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
        xs.append(x_slab[i])
        alphas.append(_lib.absmax(xs[i], ROWS, COLS, per_row=True))           # calibrated alpha = row abs-max
        outs.append(out_slab[i])
    h.sync()
    # One STEP = one pass of the hot path over the batch = ONE launch of the batched entry point
    # (antq_fakequant_batch: every workgroup looks its tensor up in a resident descriptor table).
    batch = _lib.Batch([(xs[i], outs[i], alphas[i], plan, 10.0, ROWS, COLS, True) for i in range(args.nbuf)])
    assert not batch.singles

    def step_per_tensor():          # the reference's granularity: one launch per tensor (reported beside it)
        for i in range(args.nbuf):
            _lib.fakequant(xs[i], alphas[i], plan, 10.0, ROWS, COLS, True, out=outs[i])

    def step_per_tensor_unordered():    # the same launches marked independent of their predecessors (weights at rest):
        for i in range(args.nbuf):      # ANTQ_FLAG_UNORDERED, no barrier bit on the dispatch packet
            _lib.fakequant(xs[i], alphas[i], plan, 10.0, ROWS, COLS, True, out=outs[i], unordered=True)

    def check():
        return bool(torch.equal(_lib.fakequant(outs[0], alphas[0], plan, 10.0, ROWS, COLS, True), outs[0]))

    return dict(step=batch.run, elems=args.nbuf * ROWS * COLS, kernel="antq::k_fq_hbatch<bf16,false>", scaling="weak",
                per_tensor=(step_per_tensor, step_per_tensor_unordered), check=check,
                copy=(lambda: _lib.copy(x_slab, out_slab), lambda: out_slab.copy_(x_slab)),
                workload="headline: %d x [4096,4096] bf16 weight tensors per GPU, ANT 4-bit signed flint grid, calibrated per-row "
                         "alpha, steady-state _forward; one step = ONE batched launch (antq_fakequant_batch) over all %d "
                         "tensors" % (args.nbuf, args.nbuf),
                sharding="independent tensors per rank, no data-path collective", keep=(x_slab, out_slab, xs, alphas, outs, batch))


def build_model(args, h, _lib, grids):
    """BASELINE configs[3] (`--workload opt6.7b`: 192 Linear weights, whole tensors packed onto the ranks by bytes) and
    configs[4] (`--workload llama70b`: the 70 B stack, a row block of every matrix per rank): OliVe 4-bit flint + outliers,
    outlier-victim pairs (OQ:294-330), alpha = 3 sigma per row (OQ:193-197), bf16.  The TOTAL work is fixed: strong scaling.
    Synthetic weights: randn * 0.02 with 0.1 % of the entries multiplied by U(8, 64)."""
    import numpy as np
    from ant_quantization_amd import sharding
    torch, dev, rank = h.torch, h.dev, h.rank
    mine = sharding.shard_plan(args.workload, rank, h.world, layers=args.layers)
    gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
    plan = _lib.plan_for(np.concatenate([gn, go]))
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    inplace = args.inplace or args.workload == "llama70b"
    ws, alphas = [], []
    for _, b, e, c in mine:
        w = torch.randn(e - b, c, device=dev, dtype=torch.bfloat16, generator=gen) * 0.02
        m = torch.rand(w.shape, device=dev, generator=gen) < 0.001
        w[m] *= torch.empty(int(m.sum()), device=dev, dtype=torch.bfloat16).uniform_(8, 64, generator=gen)
        del m
        ws.append(w)
        alphas.append(_lib.xmax_3sigma(w, w.shape[0], w.shape[1], per_row=True))      # OQ:193-197 on one read (antq_moments)
    outs = ws if inplace else [torch.empty_like(w) for w in ws]
    h.sync()
    batch = _lib.Batch([(w, o, a, plan, 32.0, w.shape[0], w.shape[1], True) for w, o, a in zip(ws, outs, alphas)], ovp=True)
    assert not batch.singles
    elems = sum(w.numel() for w in ws)

    def check():               # fake-quant of a fake-quantised tensor at the same scale is the tensor itself
        o = outs[0][:64].clone()
        return bool(torch.equal(_lib.fakequant(o, alphas[0][:64].contiguous(), plan, 32.0, 64, o.shape[1], True, ovp=True), o))

    src = ws[0].reshape(-1)
    dst = torch.empty_like(src)
    what = {"opt6.7b": "configs[3]: OPT-6.7B, 192 Linear weights (6.44 G elements), whole tensors packed onto the ranks by bytes "
                       "(sharding.lpt_assign)",
            "llama70b": "configs[4]: synthetic 70 B-parameter Linear stack, 560 matrices (68.5 G elements), a contiguous row block "
                        "of every matrix per rank (sharding.row_block)"}[args.workload]
    return dict(step=batch.run, elems=elems, kernel="antq::k_fq_hbatch<bf16,true>", scaling="strong", per_tensor=None, check=check,
                copy=(lambda: _lib.copy(src, dst), lambda: dst.copy_(src)), copy_bytes=2 * src.numel() * 2,
                workload="%s%s; OliVe 4-bit flint + outliers with outlier-victim pairs, alpha = 3 sigma per row, bf16%s; one step = "
                         "ONE batched launch per rank over its %d units" % (what, " (%d layers)" % args.layers if args.layers else "",
                                                                           ", quantised in place (from the second step on the input is the previous step's output -- a fixed point of the quantiser: same bytes moved, already-quantised values)" if inplace else "", len(ws)),
                sharding="%d units on rank 0 of %d ranks, no data-path collective" % (len(ws), h.world),
                keep=(ws, outs, alphas, batch, dst))


# ------------------------------------------------------------------------------------------------
# config.configs[]: the OliVe twin of the headline + every BASELINE config, each event-timed on its own after the
# headline's timed region (rank 0, N = 1).  Everything resident before its timing starts; HIP events on the launch stream.
# ------------------------------------------------------------------------------------------------
def plant_outliers(torch, w, gen, every=1000):
    """SURVEY 8d C3/C4 distribution: 0.1 % of the entries multiplied by U(8, 64) -- one entry in every run of `every`
    consecutive flat elements, at a random offset inside the run (distinct positions, no mask pass over the tensor)."""
    flat = w.view(-1)
    k = flat.numel() // every
    if k == 0:
        return
    idx = torch.arange(k, device=w.device) * every + torch.randint(0, every, (k,), device=w.device, generator=gen)
    f = torch.empty(k, device=w.device, dtype=torch.float32).uniform_(8, 64, generator=gen)
    flat[idx] = (flat[idx].float() * f).to(w.dtype)


def build_headline_olive(torch, dev, _lib, olive_plan, nbuf):
    """The OliVe twin of the headline batch: nbuf x [4096,4096] bf16, randn * 0.02 with 0.1 % of the entries multiplied by
    U(8, 64), alpha = max|mean +- 3 std| per row (OQ:193-197); ONE batched launch with the pair rule (OQ:311-320)."""
    gen = torch.Generator(device=dev).manual_seed(60)
    xs = torch.empty(nbuf, ROWS, COLS, dtype=torch.bfloat16, device=dev)
    os_ = torch.empty_like(xs)
    al = []
    for i in range(nbuf):
        xs[i] = (torch.randn(ROWS, COLS, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
        plant_outliers(torch, xs[i], gen)
        al.append(_lib.xmax_3sigma(xs[i], ROWS, COLS, per_row=True))
    bt = _lib.Batch([(xs[i], os_[i], al[i], olive_plan, 32.0, ROWS, COLS, True) for i in range(nbuf)], ovp=True)
    return xs, os_, al, bt


def resnet50_shapes():
    """The 54 weight tensors of torchvision's ResNet-50 (53 convolutions + fc), SURVEY 8a / Appendix C: 25 502 912 elements."""
    s, inp = [(64, 3, 7, 7)], 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            s += [(planes, inp, 1, 1), (planes, planes, 3, 3), (planes * 4, planes, 1, 1)]
            if b == 0:
                s.append((planes * 4, inp, 1, 1))
            inp = planes * 4
    return s + [(1000, 2048)]


def bert_base_shapes():
    """The 74 Linear weights of BERT-base (12 x {q, k, v, out, fc1, fc2} + pooler + classifier): 85.5 M elements."""
    return ([(768, 768)] * 4 + [(3072, 768), (768, 3072)]) * 12 + [(768, 768), (2, 768)]


def run_configs(h, _lib, grids, want):
    """-> list of entries {name, what, kernel, launches_per_pass, elements, bytes_per_elem, launch_us (= one pass),
    gelem_per_s, achieved_GBps, frac}.  `want`: names to run (None = all)."""
    import numpy as np
    from ant_quantization_amd import sharding
    torch, dev = h.torch, h.dev
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []

    def timed(fn):
        """Seconds per pass at steady clocks: >= 40 ms of warm-up passes, then >= 40 ms (and >= 10) timed passes."""
        fn()
        torch.cuda.synchronize()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        once = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
        for _ in range(min(5000, int(0.04 / once) + 1)):
            fn()
        reps = max(10, min(5000, int(0.04 / once) + 1))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps, reps

    def entry(name, what, kernels, elems, bpe, fn, **extra):
        secs, reps = timed(fn)
        e = {"name": name, "what": what, "kernel": " + ".join(k for k, _ in kernels), "launches_per_pass": len(kernels),
             "grid": [g for _, g in kernels], "elements": int(elems), "bytes_per_elem": bpe, "timed_passes": reps,
             "launch_us": round(secs * 1e6, 2), "gelem_per_s": round(elems / secs / 1e9, 1),
             "achieved_GBps": round(elems * bpe / secs / 1e9, 1), "algorithmic_bytes": int(elems * bpe),
             "frac": round(elems * bpe / secs / 1e9 / HBM_PEAK_GBPS, 4)}
        e.update(extra)
        out.append(e)
        return e

    def centry(name, what, kernel, launches, elems, ncand_total, fn, **extra):
        """A calibration pass (first-call clip search, AQ:287-415 / OQ:189-256): compute-bound (sort + binary searches per
        row), so no HBM fraction -- the time of one pass and the (element x candidate) evaluations per second it replaces."""
        secs, reps = timed(fn)
        e = {"name": name, "what": what, "kernel": kernel, "launches_per_pass": launches, "elements": int(elems),
             "candidates_per_element": int(ncand_total), "timed_passes": reps, "pass_ms": round(secs * 1e3, 3),
             "gcand_evals_per_s": round(elems * ncand_total / secs / 1e9, 1), "bound": "valu"}
        e.update(extra)
        out.append(e)
        return e

    def on(name):
        return want is None or name in want

    flint = _lib.plan_for(grids.ant_flint(4, True))
    gn, go = grids.olive_flint(4, True), grids.olive_outliers(4, True)
    olive = _lib.plan_for(np.concatenate([gn, go]))

    # (i) the headline batch through the OliVe outlier-victim kernel (OQ:294-330)
    if on("headline_olive"):
        xs, os_, al, bt = build_headline_olive(torch, dev, _lib, olive, 32)
        e = entry("headline_olive", "the headline batch (32 x [4096,4096] bf16) through the OliVe path: 4-bit flint + abfloat outliers, "
                  "outlier-victim pairs, 0.1 % planted outliers (x U(8,64)), alpha = 3 sigma per row; ONE batched launch",
                  bt.kernels(), 32 * ROWS * COLS, 4, bt.run)
        o0 = os_[0].clone()
        e["idempotence_check"] = bool(torch.equal(_lib.fakequant(o0, al[0], olive, 32.0, ROWS, COLS, True, ovp=True), o0))
        s0 = (al[0] / 32.0).view(-1, 1)
        e["outlier_frac_of_output"] = round(float((os_[0].float().abs() > s0 * 40.0).float().mean()), 6)
        e["victim_frac_of_output"] = round(float(((os_[0] == 0) & (xs[0].float().abs() > s0)).float().mean()), 6)
        del xs, os_, al, bt, o0

    # (ii) configs[1]: ResNet-50, all 54 weight tensors, ANT 4-bit flint, per-channel and group = 16, fp32, one batched launch
    if on("C1"):
        gen = torch.Generator(device=dev).manual_seed(1)
        ws = [torch.randn(*s, device=dev, generator=gen) * float(np.sqrt(2.0 / (s[0] * np.prod(s[2:], dtype=np.int64))))
              for s in resnet50_shapes()]
        outs = [torch.empty_like(w) for w in ws]
        elems = sum(w.numel() for w in ws)
        for nm, G in (("per-channel", 0), ("group-16", 16)):
            jobs = []
            for w, o in zip(ws, outs):
                rows, K = (w.shape[0], w.numel() // w.shape[0]) if not G else (w.numel() // G, G)
                jobs.append((w, o, _lib.absmax(w, rows, K), flint, 10.0, rows, K, True))
            bt = _lib.Batch(jobs)
            entry("C1_resnet50_%s_f32" % nm, "configs[1]: ResNet-50, 54 weight tensors (25.5 M elements), ANT 4-bit flint, %s scales "
                  "(alpha = abs-max), fp32; ONE batched launch over all 54" % nm, bt.kernels(), elems, 8, bt.run)
        del ws, outs, jobs, bt

    # (iii) configs[2]: BERT-base, 74 Linear weights batched + the [64,128,3072] post-GELU activation per tensor; fp32 and bf16
    if on("C2"):
        for dt, bpe, tag in ((torch.float32, 8, "f32"), (torch.bfloat16, 4, "bf16")):
            gen = torch.Generator(device=dev).manual_seed(2)
            ws = [(torch.randn(*s, device=dev, generator=gen) * 0.02).to(dt) for s in bert_base_shapes()]
            outs = [torch.empty_like(w) for w in ws]
            bt = _lib.Batch([(w, o, _lib.absmax(w, w.shape[0], w.shape[1]), flint, 10.0, w.shape[0], w.shape[1], True)
                             for w, o in zip(ws, outs)])
            entry("C2_bert_weights_%s" % tag, "configs[2]: BERT-base, 74 Linear weights (85.5 M elements), steady state after the type "
                  "pick (flint), per-channel alpha, %s; ONE batched launch" % tag, bt.kernels(), sum(w.numel() for w in ws), bpe, bt.run)
            gen = torch.Generator(device=dev).manual_seed(3)
            x = torch.nn.functional.gelu(torch.randn(64, 128, 3072, device=dev, generator=gen)).to(dt)
            ax = _lib.absmax(x, 1, x.numel(), per_row=False)
            ox = torch.empty_like(x)
            entry("C2_bert_activation_%s" % tag, "configs[2]: the [64,128,3072] post-GELU activation (25.2 M elements), one scale "
                  "for the tensor (signed flint after the sign flip), %s; one launch (antq_fakequant)" % tag,
                  [("antq_fakequant, per tensor", 0)], x.numel(), bpe,
                  lambda: _lib.fakequant(x, ax, flint, 10.0, 1, x.numel(), False, out=ox))
            del ws, outs, bt, x, ox

    # (iv) configs[3]: OPT-6.7B, rank 0's share of 8 (24 whole tensors by LPT), OliVe pairs, bf16 and fp32
    if on("C3"):
        mine = sharding.shard_plan("opt6.7b", 0, 8)
        for dt, bpe, tag in ((torch.bfloat16, 4, "bf16"), (torch.float32, 8, "f32")):
            gen = torch.Generator(device=dev).manual_seed(4)
            ws = []
            for _, b, e_, c in mine:
                w = (torch.randn(e_ - b, c, device=dev, generator=gen) * 0.02).to(dt)
                plant_outliers(torch, w, gen)
                ws.append(w)
            outs = [torch.empty_like(w) for w in ws]
            al = [_lib.xmax_3sigma(w, w.shape[0], w.shape[1], per_row=True) for w in ws]
            bt = _lib.Batch([(w, o, a, olive, 32.0, w.shape[0], w.shape[1], True) for w, o, a in zip(ws, outs, al)], ovp=True)
            entry("C3_opt6.7b_rank0of8_%s" % tag, "configs[3]: OPT-6.7B, the 24 weight tensors rank 0 of 8 owns (LPT by bytes, 805 M "
                  "elements), OliVe 4-bit flint + outliers, outlier-victim pairs, alpha = 3 sigma per row, %s; ONE batched launch" % tag,
                  bt.kernels(), sum(w.numel() for w in ws), bpe, bt.run)
            if tag == "bf16" and on("calib_C3"):
                # the first-call calibration of the same tensors (OQ:189-256): mean +- 3 sigma per row, int and flint (+ outliers)
                # x 88 clip ratios with the pair rule, per-row picks, the type pick -- one antq_calibrate per tensor
                oi = _lib.plan_for(np.concatenate([grids.olive_int(4, True), go]))
                pls, gms = [oi, olive], [float(grids.olive_int(4, True).max()), float(gn.max())]

                jobs = [(w, w.shape[0], w.shape[1], True, pls, gms, 75, 250, 2) for w in ws]

                def cal():          # (what enable_quantization's first forward does with a model's weights: ONE C call)
                    _lib.calibrate_batch(jobs, xmax="3sigma", ovp=True)

                def cal_each():
                    for w in ws:
                        _lib.calibrate(w, w.shape[0], w.shape[1], True, pls, gms, 75, 250, 2, xmax="3sigma", ovp=True)

                e = centry("calib_C3_opt6.7b_rank0of8_bf16", "first-call calibration of those 24 tensors: 3-sigma statistic, OliVe int / "
                           "flint + outliers x 88 clip ratios with the pair rule, per-row picks, type pick (antq_calibrate_batch: one C call, "
                           "per tensor k_moments + the sorted-row search + picks)", "k_search_sorted<bf16,true,false>", 24 * 6,
                           sum(w.numel() for w in ws), 176, cal)
                e["x8_for_the_192_tensors_ms"] = round(e["pass_ms"] * 8, 1)
                e["one_antq_calibrate_per_tensor_ms"] = round(timed(cal_each)[0] * 1e3, 3)
            del ws, outs, al, bt

    # (v) configs[4]: the 70 B-parameter bf16 stack, rank 0's row block of every matrix at 8 ranks (17.1 GB in, 17.1 GB out)
    if on("C4"):
        mine = sharding.shard_plan("llama70b", 0, 8)
        gen = torch.Generator(device=dev).manual_seed(5)
        ws = []
        for _, b, e_, c in mine:
            w = torch.randn(e_ - b, c, device=dev, dtype=torch.bfloat16, generator=gen) * 0.02
            plant_outliers(torch, w, gen)
            ws.append(w)
        outs = [torch.empty_like(w) for w in ws]
        al = [_lib.xmax_3sigma(w, w.shape[0], w.shape[1], per_row=True) for w in ws]
        bt = _lib.Batch([(w, o, a, olive, 32.0, w.shape[0], w.shape[1], True) for w, o, a in zip(ws, outs, al)], ovp=True)
        entry("C4_llama70b_rank0of8_bf16", "configs[4]: synthetic 70 B-parameter Linear stack, rank 0's row block of each of the 560 "
              "matrices at 8 ranks (%.2f G elements, %.1f GB in + the same out, out of place), OliVe 4-bit flint + outliers, "
              "outlier-victim pairs, alpha = 3 sigma per row, bf16; ONE batched launch"
              % (sum(w.numel() for w in ws) / 1e9, sum(w.numel() for w in ws) * 2 / 1e9), bt.kernels(),
              sum(w.numel() for w in ws), 4, bt.run)
        del ws, outs, al, bt
    # (vi) first-call calibration (N1): the type selection of a per-channel weight and of an fp32 activation
    if on("calib"):
        ant3 = [_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
        gen = torch.Generator(device=dev).manual_seed(6)
        w = torch.randn(ROWS, COLS, device=dev, generator=gen) * 0.02
        centry("calib_type_selection_4096x4096_f32", "antq_calibrate of one [4096,4096] fp32 weight, per-channel: abs-max, ANT int / pot / "
               "flint x 75 clip ratios (AQ:287-415), per-row picks, type pick", "k_search_sorted<float,false,false>", 1, w.numel(), 225,
               lambda: _lib.calibrate(w, ROWS, COLS, True, ant3, [10.0] * 3, 75, 150, 1, xmax="absmax"))
        a = torch.nn.functional.gelu(torch.randn(64 * 128 * 3072, device=dev, generator=gen))
        centry("calib_activation_64x128x3072_f32", "antq_calibrate of one [64,128,3072] fp32 activation with ONE scale (configs[2]'s "
               "largest): abs-max, ANT int / pot / flint x 75 clip ratios, picks", "k_search_sorted<float,false,true>", 1, a.numel(), 225,
               lambda: _lib.calibrate(a, 1, a.numel(), False, ant3, [10.0] * 3, 75, 150, 1, xmax="absmax"))
        ab = a.to(torch.bfloat16)
        centry("calib_activation_64x128x3072_bf16", "the same activation in bf16: the 65 536-bin histogram search", "k_hist16 + k_hist_score",
               1, ab.numel(), 225, lambda: _lib.calibrate(ab, 1, ab.numel(), False, ant3, [10.0] * 3, 75, 150, 1, xmax="absmax"))
        del w, a, ab
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nbuf", type=int, default=32, help="distinct 4096x4096 bf16 tensors per GPU (one step)")
    ap.add_argument("--workload", choices=["headline", "opt6.7b", "llama70b"], default="headline",
                    help="headline (BASELINE metric: weak scaling) | opt6.7b (configs[3]) | llama70b (configs[4]): a whole model's "
                         "Linear weights sharded over the ranks, strong scaling")
    ap.add_argument("--layers", type=int, default=0, help="model workloads: decoder layers (0 = the model's own depth)")
    ap.add_argument("--inplace", action="store_true", help="model workloads: out = x (llama70b always: 137 GB of weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic falls back "
                                                               "to the committed profiles/hbm_traffic.json value)")
    ap.add_argument("--configs", default="all", help="headline workload at N = 1: which config.configs[] entries to time after "
                                                     "the headline region: all | none | comma list of headline_olive,C1,C2,C3,C4")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))             # no launcher around us: one process per GPU, started here
    if os.environ.get("ANTQ_BENCH_SELFTEST") == "1":
        return selftest_main(args)

    h = Harness(args.gpus)
    torch, rank = h.torch, h.rank
    from ant_quantization_amd import _lib, grids

    W = build_headline(args, h, _lib, grids) if args.workload == "headline" else build_model(args, h, _lib, grids)
    step = W["step"]

    if args.traffic_child:              # a counter pass of measure_traffic(): the batched launch(es) and nothing else
        for _ in range(3):
            step()
        if args.workload == "headline":     # ... and the OliVe twin of the batch (config.configs[0]), same pass
            import numpy as np
            olive = _lib.plan_for(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))
            twin = build_headline_olive(torch, h.dev, _lib, olive, args.nbuf)
            for _ in range(3):
                twin[3].run()
        h.sync()
        return

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def event_time(fn, min_seconds, per_call):
        """Average HIP-event duration of fn() over back-to-back calls for at least `min_seconds` (last reading)."""
        t_all, last = time.perf_counter(), None
        while last is None or time.perf_counter() - t_all < min_seconds:
            ev0.record()
            for _ in range(per_call):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            last = ev0.elapsed_time(ev1) * 1e-3 / per_call
        return last

    # ---- setup pass: the same work as one launch per tensor (the reference's granularity).
    # Reported beside the headline; it also keeps the GPU busy for >= 0.3 s before anything is timed,
    # which is what it takes for an idle MI355X to reach steady clocks (the first ~50 ms of load run
    # up to 20 % slower: tools/probe_clock_ramp.py).
    if W["per_tensor"]:
        pt_launch_s = event_time(W["per_tensor"][0], 0.3, 5) / args.nbuf
        ptu_launch_s = event_time(W["per_tensor"][1], 0.15, 5) / args.nbuf
    else:
        event_time(step, 0.3, 1)

    # ---- the timed region: HIP events on the launch stream (= torch's current stream) inside the barriers
    elapsed = h.timed(step, args.steps, args.warmup, on_start=ev0.record, on_stop=ev1.record)
    # roofline of the dominant kernel: the timed region is nothing but back-to-back launches of it on one stream,
    # so its average launch duration = event time / launches -- on EVERY rank, gathered below
    launch_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps
    algo_bytes = W["elems"] * BYTES_PER_ELEM
    launch_all = h.gather(launch_s)
    bytes_all = h.gather(algo_bytes)
    fracs = [b / t / 1e9 / HBM_PEAK_GBPS for b, t in zip(bytes_all, launch_all)]
    worst = min(range(h.world), key=lambda r: fracs[r])
    achieved = bytes_all[worst] / launch_all[worst] / 1e9           # the SLOWEST rank's kernel (N = 1: the only one)

    # ---- parity spot check of what was just measured (cheap, outside the timed region) --------
    ok = W["check"]()

    h.finish()                          # every rank is done; what follows is rank 0 describing its own GPU
    if rank != 0:
        return

    # ---- empirical ceiling on the same bytes: plain copies in -> out, one launch each
    copy_bytes = W.get("copy_bytes", algo_bytes)
    copy_s = event_time(W["copy"][0], 0.15, 10)
    d2d_s = event_time(W["copy"][1], 0.15, 10)
    ceiling = max(copy_bytes / copy_s, copy_bytes / d2d_s) / 1e9

    want = None if args.configs == "all" else [c for c in args.configs.split(",") if c and c != "none"]
    do_configs = h.world == 1 and args.workload == "headline" and (want is None or want)
    K_ANT, K_OVP = r"k_fq_hbatch<[^>]*false>", r"k_fq_hbatch<[^>]*true>"
    traffic, traffic_note, traffic_olive = None, None, (None, None)
    if not args.no_traffic:             # (at N > 1 too: the child passes are one-GPU runs on rank 0's GPU, after the job is over)
        child = ["--workload", args.workload, "--layers", str(args.layers)] + (["--inplace"] if args.inplace else [])
        if args.workload != "headline" and h.world > 1:
            traffic_note = "not measured: a child pass cannot rebuild rank 0's share of a sharded model on its own"
        else:
            pats = (K_ANT, K_OVP) if args.workload == "headline" else ("k_fq_hbatch",)
            got, notes = measure_traffic(args.nbuf, kernels=pats, child_args=child,
                                         timeout_s=240 if args.workload == "headline" else 900)
            if got is None:
                traffic_note = notes
            else:
                traffic, traffic_note = got[pats[0]], notes[pats[0]]
                if len(pats) > 1:
                    traffic_olive = (got[pats[1]], notes[pats[1]])
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # rocprofv3 --pmc result, see profiles/README.md
    if traffic is None and args.workload == "headline" and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            per_tensor = tj.get("headline_bytes_per_tensor")
            traffic = int(per_tensor * args.nbuf) if per_tensor else None
            traffic_note = ("from profiles/hbm_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                            "command (FETCH_SIZE x2 per the guide's gfx950 correction), NOT measured in this run (%s)"
                            % (tj.get("headline_kernel"), traffic_note))
        except Exception:
            traffic = None

    config = {"workload": W["workload"], "io_dtype": "bf16", "elements_per_step_per_gpu": W["elems"],
              "elements_per_step_per_rank": [int(b // BYTES_PER_ELEM) for b in bytes_all],
              "sharding": W["sharding"], "idempotence_check": ok}
    if W["per_tensor"]:
        config["per_tensor_launches"] = {
            "what": "the same pass as ONE LAUNCH PER TENSOR (antq_fakequant, the reference's granularity), %d independent weight "
                    "tensors back to back on one stream, each launch marked ANTQ_FLAG_UNORDERED (inputs at rest: its dispatch "
                    "packet carries no barrier bit, so it may start while its predecessor drains); `ordered` = the same launches "
                    "without the flag" % args.nbuf,
            "kernel": "antq::k_fq_hrow<bf16,false,4> (16-bit-domain row table, one wavefront per workgroup)",
            "launch_us": round(ptu_launch_s * 1e6, 2),
            "gelem_per_s": round(ROWS * COLS / ptu_launch_s / 1e9, 1),
            "achieved_GBps": round(ROWS * COLS * BYTES_PER_ELEM / ptu_launch_s / 1e9, 1),
            "frac": round(ROWS * COLS * BYTES_PER_ELEM / ptu_launch_s / 1e9 / HBM_PEAK_GBPS, 4),
            "ordered": {"kernel": "antq::k_fq_hrow<bf16,false,8> (one wavefront per row)",
                        "launch_us": round(pt_launch_s * 1e6, 2),
                        "frac": round(ROWS * COLS * BYTES_PER_ELEM / pt_launch_s / 1e9 / HBM_PEAK_GBPS, 4)}}
    if do_configs:
        config["configs"] = run_configs(h, _lib, grids, want)
        for e in config["configs"]:
            if e["name"] == "headline_olive" and args.nbuf == 32:
                e["traffic"], e["traffic_source"] = traffic_olive
                if e["traffic"]:
                    e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_bytes"], 4)
    res = report(h, args, sum(bytes_all) / BYTES_PER_ELEM, elapsed, {
        "config": config,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": W["kernel"], "launch_us": round(launch_all[worst] * 1e6, 2),
                     "algorithmic_bytes_per_launch": int(bytes_all[worst]),
                     "per_rank": {"what": "every rank's own launch duration (HIP events around its timed region) and its "
                                          "algorithmic bytes / that duration / 8 TB/s; `achieved`, `frac`, `launch_us` above "
                                          "are the slowest rank's (rank %d)" % worst,
                                  "ranks": h.world, "launch_us": per_rank_stats([t * 1e6 for t in launch_all]),
                                  "frac": per_rank_stats(fracs, 4)},
                     "copy_ceiling": {"antq_copy_GBps": round(copy_bytes / copy_s / 1e9, 1),
                                      "hipMemcpyDtoD_GBps": round(copy_bytes / d2d_s / 1e9, 1),
                                      "frac_of_copy_ceiling": round(achieved / ceiling, 4),
                                      "what": "plain copy of %d bytes in -> out (one launch) on rank 0's GPU, timed after the "
                                              "timed region" % (copy_bytes // 2)}},
    }, scaling=W["scaling"])
    if h.world == 1 and not args.no_cpu_baseline and args.workload == "headline":      # reported baselines, N=1 only
        res["cpu_baseline"] = cpu_baseline()
        phys = _physical_cores()
        if phys and phys < (os.cpu_count() or 1):      # SMT box: the same port with one thread per physical core beside it
            res["cpu_baseline_physical_cores"] = cpu_baseline(seconds_budget=2.0, threads=phys)
        res["cpu_baseline_torch"] = cpu_baseline_torch()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
