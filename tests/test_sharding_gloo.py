"""N>1 path on CPU: world_size-2 gloo.  The data path has no collective, so what must hold is that
the partition is exact (every unit owned once, pairs never cut) and that the only cross-rank
operations bench.py uses (barrier + MAX of elapsed) behave."""
import os
import socket
import sys

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
    mx = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, mine, (b, e), mx, [g[:2] for g in gathered],
           bool(np.array_equal(np.concatenate([g[2] for g in gathered]), full))))
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
    assert res[0][2][1] == res[1][2][0] and res[0][2][0] == 0 and res[1][2][1] == 10
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
