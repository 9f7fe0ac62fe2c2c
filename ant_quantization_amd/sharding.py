"""Partitioning of the fake-quant work across the GPUs of a node.

The path has no exchange step: tensors -- or contiguous row blocks of one huge
tensor -- are independent units, so sharding is pure bookkeeping and there is NO data-path
collective.  Only two things ever cross ranks: the barrier around a timed region and a MAX
reduction of elapsed times (bench.py), plus the reference's own one-shot calibration syncs
(AQ/quant_modules.py:525-531) when the quantiser runs under DDP.
"""
from typing import List, Sequence, Tuple


def lpt_assign(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time bin packing of tensors (by bytes / elements) onto `world` ranks.
    Returns, per rank, the indices of the tensors it owns.  Deterministic (ties by index), every
    index appears exactly once."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    for r in range(world):
        out[r].sort()
    return out


def row_block(rows: int, rank: int, world: int, pair_safe_row_len: int = 0) -> Tuple[int, int]:
    """[begin, end) rows of a [rows, row_len] tensor owned by `rank`: contiguous, balanced to within
    one row.  With `pair_safe_row_len` odd (OliVe outlier-victim pairs live on the FLAT tensor), block
    boundaries are moved to even row indices so that no (2k, 2k+1) pair is cut."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    b = rows * rank // world
    e = rows * (rank + 1) // world
    if pair_safe_row_len % 2 == 1:
        b -= b % 2
        e = rows if rank == world - 1 else e - e % 2
    return b, e


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a python float over the process group (identity without one)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
