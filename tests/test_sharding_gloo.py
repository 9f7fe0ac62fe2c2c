"""N>1 path on CPU: world_size-2 gloo.  The data path has no collective, so what must hold is that
the partition is exact (every unit owned once, pairs never cut) and that the only cross-rank
operations bench.py uses (barrier + MAX of elapsed) behave."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ant_quantization_amd import sharding
    from oracle import antq_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # OPT-6.7B-like tensor list (elements), LPT by size
    sizes = [4096 * 4096] * 8 + [16384 * 4096] * 2 + [4096 * 16384] * 2 + [1000, 7]
    mine = sharding.lpt_assign(sizes, world)[rank]
    # row-block sharding of ONE tensor with OliVe pairs: each rank quantises its block with the CPU oracle
    # (stand-in for the GPU kernel in this CPU-only test); concatenation must equal the unsharded result.
    rng = np.random.default_rng(0)
    rows, K = 10, 33                                     # odd row length: pairs straddle rows
    x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
    x[rng.random((rows, K)) < 0.05] *= 40
    alpha = (3 * x.std(1)).astype(np.float32)
    gn = orc.olive_flint_value(4, True)
    grid = np.concatenate([gn, orc.olive_outlier_value(4, True)])
    b, e = sharding.row_block(rows, rank, world, pair_safe_row_len=K)
    part, _ = orc.forward(x[b:e], alpha[b:e], grid, gmax=32.0, ovp=True)
    full, _ = orc.forward(x, alpha, grid, gmax=32.0, ovp=True)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b, e, part))          # test-only gather, not part of the data path
    dist.barrier()
    ok = bool(np.array_equal(np.concatenate([g[2] for g in gathered]), full))
    # ODD element count (odd rows x odd row length): the last element's partner is GLOBAL element 0 (torch.roll wrap,
    # OQ:313-318), which lives on rank 0 -- sharding.fix_odd_numel_wrap carries that one bit to the last rank.  Both
    # states of the bit, and a last element that the local launch pairs with an outlier of its own block.
    plan = types.SimpleNamespace(grid=grid)
    for first, blk_first in ((0.9, 0.0), (0.001, 0.9), (0.9, 0.9), (0.001, 0.001)):
        rows = 9
        x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
        alpha = np.full(rows, 0.06, np.float32)
        x[0, 0] = first                                   # outlier / normal at global element 0
        x[-1, -1] = 0.01
        b, e = sharding.row_block(rows, rank, world, pair_safe_row_len=K)
        b1 = sharding.row_block(rows, world - 1, world, pair_safe_row_len=K)[0]
        x[b1, 0] = blk_first                              # first element of the LAST rank's block (the wrong partner)
        full, _ = orc.forward(x, alpha, grid, gmax=32.0, ovp=True)
        part, _ = orc.forward(x[b:e], alpha[b:e], grid, gmax=32.0, ovp=True)
        xb, ob, ab = torch.from_numpy(x[b:e].copy()), torch.from_numpy(part.copy()), torch.from_numpy(alpha[b:e].copy())

        def plain(x1, a1):
            return torch.from_numpy(orc.forward(x1.numpy().reshape(1, 1), a1.numpy(), grid, gmax=32.0, ovp=False)[0].reshape(-1))

        sharding.fix_odd_numel_wrap(xb, ob, ab, plan, 32.0, rows, K, rank, world, plain_fn=plain)
        gathered = [None] * world
        dist.all_gather_object(gathered, ob.numpy())
        got = np.concatenate(gathered)
        ok = ok and bool(np.array_equal(got.view(np.uint32), full.view(np.uint32)))
        if blk_first > 0.5 and first < 0.5:               # the case the fix exists for: unfixed blocks differ
            unfixed = [None] * world
            dist.all_gather_object(unfixed, part)
            ok = ok and not np.array_equal(np.concatenate(unfixed), full)
    mx = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, mine, (b, e), mx, [g[:2] for g in gathered] if False else None, ok))
    dist.destroy_process_group()


def test_partition_is_exact_and_pair_safe_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(res[0][1] + res[1][1])
    assert owned == list(range(14))                                      # every tensor exactly once
    assert res[0][2][1] == res[1][2][0] and res[0][2][0] == 0 and res[1][2][1] == 9
    assert res[0][2][1] % 2 == 0                                         # odd row_len: cut on an even row
    assert res[0][3] == res[1][3] == 2.0                                 # MAX over ranks
    assert res[0][5] and res[1][5]                                       # sharded == unsharded, bit for bit


def test_lpt_balance_and_row_blocks():
    from ant_quantization_amd import sharding
    sizes = [4096 * 4096] * 128 + [16384 * 4096] * 32 + [4096 * 16384] * 32      # OPT-6.7B
    for world in (1, 2, 4, 8):
        parts = sharding.lpt_assign(sizes, world)
        assert sorted(i for p in parts for i in p) == list(range(192))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) == min(loads)                                   # 24 tensors per GPU at world 8
    for rows in (1, 7, 4096, 28672):
        for world in (1, 2, 3, 8):
            blocks = [sharding.row_block(rows, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == rows
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in blocks) - min(e - b for b, e in blocks) <= 1
    with pytest.raises(ValueError):
        sharding.row_block(8, 2, 2)


def _run_bench(tmp_path, nproc, gpus, steps=12, warmup=2, extra=()):
    env = dict(os.environ, ANTQ_BENCH_SELFTEST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus),
           "--steps", str(steps), "--warmup", str(warmup)] + list(extra)
    return subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)


def test_bench_rank_harness_under_torchrun_world2(tmp_path):
    """The command line the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`), executed end to end at world 2:
    bench.py's own Harness (env parsing, process group, barriers, MAX over ranks, rank-0 JSON line) over gloo with the
    GPU workload replaced by a stub in which rank r sleeps (r + 1) ms per step (ANTQ_BENCH_SELFTEST=1)."""
    r = _run_bench(tmp_path, 2, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                     # ONE JSON line, from rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 12 and res["warmup"] == 2 and res["scaling"] == "weak"
    assert res["higher_is_better"] is True and res["unit"] == "Gelem/s" and res["selftest"] is True
    assert 2.0 <= res["ms_per_step"] < 20.0                              # the SLOWER rank's 2 ms, not rank 0's 1 ms
    # value = units of ALL ranks / the slowest rank's time
    assert abs(res["value"] - 2 * 1e6 * 12 / (res["ms_per_step"] * 1e-3 * 12) / 1e9) < 2e-3 * res["value"] + 1e-3
    # every rank's own kernel time reaches the line (gathered, not rank 0's alone)
    pr = res["roofline"]["per_rank"]
    assert pr["ranks"] == 2 and pr["launch_us"] == {"min": 1000.0, "mean": 1500.0, "max": 2000.0}
    assert res["config"]["elements_per_step_per_rank"] == [1000000, 1000000]


@pytest.mark.parametrize("workload,layers,total", [("opt6.7b", 2, 2 * 100663296 * 2), ("llama70b", 1, 855638016)])
def test_bench_sharded_model_workloads_under_torchrun_world2(tmp_path, workload, layers, total):
    """BASELINE configs[3] / [4] as driver-runnable bench lines (`bench.py --workload opt6.7b | llama70b`): at world 2 the
    shard plan (sharding.shard_plan: LPT packing of whole tensors / a row block of every matrix) gives the two ranks disjoint
    shares that add up to the whole model, the work is fixed (`scaling: strong`) and `value` counts ALL ranks' elements
    against the slowest rank's time."""
    r = _run_bench(tmp_path, 2, 2, extra=["--workload", workload, "--layers", str(layers)])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    per = res["config"]["elements_per_step_per_rank"]
    assert res["scaling"] == "strong" and res["n_gpus"] == 2 and len(per) == 2 and sum(per) == total
    assert max(per) <= 1.01 * min(per)
    assert abs(res["value"] - total / (res["ms_per_step"] * 1e-3) / 1e9) < 2e-3 * res["value"] + 1e-3
    assert res["roofline"]["per_rank"]["ranks"] == 2


def test_shard_plan_covers_every_unit_exactly_once():
    """sharding.shard_plan for both sharded BASELINE workloads at world 1, 2, 3, 8: every (tensor, row) is owned by exactly one
    rank; opt6.7b ranks own whole tensors with loads within one tensor of each other; llama70b ranks own a row block of every
    matrix, balanced to within a row, cut so that no OliVe pair is split."""
    from ant_quantization_amd import sharding
    for name in ("opt6.7b", "llama70b"):
        shapes = sharding.model_linear_shapes(name)
        total = sum(r * c for r, c in shapes)
        assert total == {"opt6.7b": 6442450944, "llama70b": 68451041280}[name]
        for world in (1, 2, 3, 8):
            owned = {}
            loads = []
            for rank in range(world):
                units = sharding.shard_plan(name, rank, world)
                loads.append(sum((e - b) * c for _, b, e, c in units))
                for i, b, e, c in units:
                    assert c == shapes[i][1] and 0 <= b < e <= shapes[i][0] and ((b * c) % 2 == 0)
                    owned.setdefault(i, []).append((b, e))
            assert sum(loads) == total
            for i, blocks in owned.items():
                blocks.sort()
                assert blocks[0][0] == 0 and blocks[-1][1] == shapes[i][0]
                assert all(blocks[k][1] == blocks[k + 1][0] for k in range(len(blocks) - 1))
            assert sorted(owned) == list(range(len(shapes)))
            assert max(loads) - min(loads) <= max(r * c for r, c in shapes)


def test_bench_refuses_a_world_size_other_than_gpus(tmp_path):
    r = _run_bench(tmp_path, 2, 4)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    env = dict(os.environ, ANTQ_BENCH_SELFTEST="1", WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE in the environment): bench.py starts the two
    ranks itself -- one process per GPU, the environment torch.distributed.run would have set -- and the same harness
    produces the same single JSON line (gloo, stub workload)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ANTQ_BENCH_SELFTEST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 10 and res["selftest"] is True
    assert 2.0 <= res["ms_per_step"] < 20.0                              # the slower rank's time


def test_c3_shaped_list_lpt_split_and_odd_numel_fix_world2():
    """The sharded driver's host logic (tools/bench_sharded.py) for a C3-shaped list at world 2: LPT by bytes gives every
    tensor to exactly one rank with equal loads, and `fix_odd_numel_wrap` is a no-op for every C3 / C4 tensor (even
    element counts) -- checked by calling it with a process group of 2 over gloo on a small even tensor."""
    import torch.multiprocessing as mp
    from ant_quantization_amd import sharding
    shapes = []
    for _ in range(4):                                                   # 4 OPT-6.7B layers
        shapes += [(4096, 4096)] * 4 + [(16384, 4096), (4096, 16384)]
    sizes = [r * c for r, c in shapes]
    parts = sharding.lpt_assign(sizes, 2)
    assert sorted(i for p in parts for i in p) == list(range(len(shapes)))
    assert sum(sizes[i] for i in parts[0]) == sum(sizes[i] for i in parts[1])
    assert all((r * c) % 2 == 0 for r, c in shapes)                      # no odd element count: no wrap bit needed
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_even_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def _even_worker(rank, world, port, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    from ant_quantization_amd import _lib, grids, sharding
    from oracle import antq_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)])
    plan = _lib.plan_for(grid)
    rng = np.random.default_rng(5)
    rows, K = 8, 64
    x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
    x[0, 0] = 1.9                                                         # element 0 an outlier: irrelevant for an even count
    alpha = np.full(rows, 0.06, dtype=np.float32)
    b, e = sharding.row_block(rows, rank, world, pair_safe_row_len=K)
    full, _ = orc.forward(x, alpha, grid, gmax=32.0, ovp=True)
    part, _ = orc.forward(x[b:e], alpha[b:e], grid, gmax=32.0, ovp=True)
    ob = torch.from_numpy(part.copy())
    sharding.fix_odd_numel_wrap(torch.from_numpy(x[b:e].copy()), ob, torch.from_numpy(alpha[b:e].copy()), plan, 32.0, rows, K,
                                rank, world, plain_fn=None)
    q.put(bool(np.array_equal(ob.numpy().view(np.uint32), full[b:e].view(np.uint32))))
    dist.destroy_process_group()


# ---- SURVEY 8e's only collectives: a per-tensor quantiser on a row-sharded tensor --------------------------------------
class OracleBlockOps:
    """sharding.GpuBlockOps with the CPU oracle in place of the HIP kernels (test-only): the same per-block quantities,
    from numpy / oracle/antq_oracle.c, as torch CPU tensors -- so that the protocol (what is reduced, how, and what every
    rank computes from the reduced numbers) runs over gloo exactly as it runs over RCCL."""

    def __init__(self, orc, grids_of_plans, ovp_gmax=None):
        self.orc, self.grids = orc, grids_of_plans

    def absmax(self, xb):
        import torch
        return torch.from_numpy(np.float32([np.abs(xb.numpy()).max()]))

    def moments(self, xb):
        import torch
        x = xb.numpy().astype(np.float64)
        return torch.from_numpy(np.float64([[x.sum(), (x * x).sum()]]))

    def xmax_3sigma(self, xb, sums, n_total):
        import torch
        s1, s2 = float(sums[0, 0]), float(sums[0, 1])
        mean = s1 / n_total
        var = (s2 - n_total * mean * mean) / (n_total - 1)
        m32, sd32 = np.float32(mean), np.float32(np.sqrt(max(var, 0.0)))
        t3 = np.float32(3.0) * sd32
        return torch.from_numpy(np.float32([max(abs(np.float32(m32 + t3)), abs(np.float32(m32 - t3)))]))

    def ratios(self, lb, ub, step, device):
        import torch
        return torch.from_numpy(np.float32([np.float32(i * 0.01) for i in range(lb, ub, step)]))

    def search_sse(self, xb, xmax, ratios, plans, gmaxs, ovp):
        import torch
        x = xb.numpy()
        out = np.empty((len(plans), ratios.numel(), 1), np.float64)
        for t, (p, gm) in enumerate(zip(plans, gmaxs)):
            for c, r in enumerate(ratios.numpy()):
                alpha = np.float32(xmax.numpy()[0] * r)
                q, _ = self.orc.forward(x.reshape(1, -1), np.float32([alpha]), self.grids[p], gm, ovp, want_idx=False)
                out[t, c, 0] = ((q.reshape(-1).astype(np.float64) - x.reshape(-1)) ** 2).sum()
        return torch.from_numpy(out)

    def pick(self, sse, xmax, ratios, n_total):
        import torch
        best, alpha = np.float32(1e10), np.float32(xmax.numpy()[0])
        for c in range(sse.shape[0]):
            score = np.float32(float(sse[c, 0]) / float(n_total))
            if score < best:
                best, alpha = score, np.float32(xmax.numpy()[0] * ratios.numpy()[c])
        return torch.from_numpy(np.float32([best])), torch.from_numpy(np.float32([alpha]))


def _calib_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ant_quantization_amd import sharding
    from oracle import antq_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    rng = np.random.default_rng(77)
    cases = []
    # OliVe activation quantiser (per tensor, unsigned -> here signed data, ant-int-flint, 3-sigma rule, pairs) and an ANT one
    # (abs-max, ant-int-pot-flint); rows deliberately not divisible by the world size
    for rows, K, olive in ((37, 64, True), (37, 64, False), (5, 1536, True), (64, 33 * 2, False)):
        x = (rng.standard_normal((rows, K)) * 0.7).astype(np.float32)
        if olive:
            x[rng.random((rows, K)) < 0.01] *= 30.0
        cases.append((x, olive))
    for x, olive in cases:
        rows, K = x.shape
        if olive:
            types = {"int": np.concatenate([orc.olive_int_value(4, True), orc.olive_outlier_value(4, True)]),
                     "flint": np.concatenate([orc.olive_flint_value(4, True), orc.olive_outlier_value(4, True)])}
            gmaxs = [float(orc.olive_int_value(4, True).max()), float(orc.olive_flint_value(4, True).max())]
            lb, ub, step, stat = 75, 250, 2, "3sigma"
        else:
            types = {"int": orc.ant_grid("int", 4, True), "pot": orc.ant_grid("pot", 4, True), "flint": orc.ant_grid("flint", 4, True)}
            gmaxs = [float(np.max(g)) for g in types.values()]
            lb, ub, step, stat = 80, 150, 1, "absmax"
        names = list(types)
        ops = OracleBlockOps(orc, types)
        b, e = sharding.row_block(rows, rank, world, pair_safe_row_len=K)
        xb = torch.from_numpy(np.ascontiguousarray(x[b:e]))
        res = sharding.sharded_calibrate(xb, x.size, names, gmaxs, lb, ub, step, statistic=stat, ovp=olive, ops=ops)
        # the unsharded calibration of the same tensor, straight from the oracle
        xm = orc.three_sigma(x, per_row=False) if olive else orc.absmax(x, per_row=False)
        full = [orc.search_mse(x, xm, lb, ub, step, types[n], gm, ovp=olive, per_row=False) for n, gm in zip(names, gmaxs)]
        scores = np.float32([f[0][0] for f in full])
        out.append(dict(rank=rank, xmax=float(res["xmax"][0]), xmax_ref=float(xm[0]), alpha=res["alpha"].numpy().tolist(),
                        alpha_ref=[float(f[1][0]) for f in full], type=res["type"], type_ref=int(np.argsort(scores, kind="stable")[0]),
                        score=res["score"].numpy().tolist(), score_ref=scores.tolist(),
                        traces=[f[2][:, 0].tolist() for f in full], lb=lb, step=step))
    q.put(out)
    dist.destroy_process_group()


def test_sharded_per_tensor_calibration_world2():
    """A row-sharded PER-TENSOR quantiser (SURVEY 8e's only collectives: all_reduce(MAX) of the abs-max, all_reduce(SUM) of
    (sum x, sum x^2) and of the T x R squared-error sums) over gloo at world 2: both ranks end with the same x_max, the
    same clip alpha per type and the same type as the unsharded calibration of the whole tensor by the oracle (a differing
    alpha must be a tie by the oracle's own scores)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_calib_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res[0]) == len(res[1]) == 4
    for a, b in zip(*res):
        for k in ("xmax", "alpha", "type", "score"):
            assert a[k] == b[k], (k, a[k], b[k])                         # every rank holds identical numbers
        assert a["xmax"] == pytest.approx(a["xmax_ref"], rel=2e-6)       # (sum order of the statistic; exact for the abs-max)
        for t, (al, ar) in enumerate(zip(a["alpha"], a["alpha_ref"])):
            if al != pytest.approx(ar, rel=2e-6):
                tr = np.float32(a["traces"][t])                          # a different candidate: only as a tie of the oracle's scores
                ci = int(round((al / a["xmax"] * 100 - a["lb"]) / a["step"]))
                assert abs(float(tr[ci]) - float(tr.min())) <= 1e-6 * float(tr.min()), (t, al, ar)
        assert a["type"] == a["type_ref"] or abs(a["score_ref"][a["type"]] - min(a["score_ref"])) <= 1e-6 * min(a["score_ref"])
        np.testing.assert_allclose(a["score"], a["score_ref"], rtol=1e-5)


def _nan_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from ant_quantization_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = OracleBlockOps(orc_module(), {})
    out = []
    # (which rank holds what): NaN on one rank only, on the other, on both; +Inf against a finite block; -NaN payloads;
    # plain finite blocks (the larger wins exactly); an all-zero tensor
    cases = [([1.0, -3.5], [2.0, float("nan")]), ([float("nan"), 0.5], [7.0, 1.0]), ([float("nan")], [float("nan")]),
             ([float("inf"), 1.0], [3.0e38, 2.0]), ([-float("inf")], [float("nan")]), ([1.5, -2.25], [-2.5, 0.125]),
             ([0.0, -0.0], [0.0, 0.0])]
    for blocks in cases:
        xb = torch.tensor(blocks[rank], dtype=torch.float32).reshape(1, -1)
        if blocks is cases[4] and rank == 1:
            xb = torch.tensor([0xFFC00001 - (1 << 32)], dtype=torch.int32).view(torch.float32).reshape(1, 1)   # a NEGATIVE NaN with payload
        r = sharding.sharded_absmax(xb, ops=ops)
        out.append(float(r[0]))
    q.put((rank, out))
    dist.destroy_process_group()


def orc_module():
    from oracle import antq_oracle
    return antq_oracle


def test_sharded_absmax_keeps_nan_by_construction_world2():
    """the abs-max of a row-sharded tensor crosses the ranks as an int32 bit pattern under the INTEGER
    maximum (NaN above +Inf above every finite value: the kernel's own atomicMax ordering), so a NaN held by any one rank
    reaches every rank -- like torch.max on the whole tensor -- whatever a backend's float MAX does with NaN."""
    import math
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        v = res[r]
        assert math.isnan(v[0]) and math.isnan(v[1]) and math.isnan(v[2]) and math.isnan(v[4]), v
        assert v[3] == float("inf") and v[5] == 2.5 and v[6] == 0.0, v
