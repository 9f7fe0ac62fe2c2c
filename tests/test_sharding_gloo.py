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


def _run_bench(tmp_path, nproc, gpus, steps=12, warmup=2):
    env = dict(os.environ, ANTQ_BENCH_SELFTEST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus),
           "--steps", str(steps), "--warmup", str(warmup)]
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
