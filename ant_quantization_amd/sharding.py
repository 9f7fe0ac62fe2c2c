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


# ---- OliVe pairs on a row-sharded tensor with an ODD element count ------------------------------------------------
# Pairs (2k, 2k+1) live on the FLAT tensor and `torch.roll` wraps (OQ:313-318): with an odd element count the last
# element has no partner of its own and is zeroed iff element 0 is an outlier.  When rows are sharded, element 0 lives
# on rank 0 and the last element on the last rank, whose local launch pairs it with the first element of ITS block
# instead.  That one bit is the only thing the sharded path ever exchanges; every other pair is whole inside a block
# because row_block() cuts at even flat offsets.
def wrap_flag(out_block0, alpha0, plan, gmax):
    """Rank 0: is global element 0 an outlier (|q| > 32, OQ:314)?  Read off its OUTPUT: out = fl(q * s), so
    |q| > 32 <=> |out| >= fl(vout * s) with vout the smallest outlier magnitude of the codebook (monotone rounding;
    32 and vout are a factor 1.5 apart, so the two sides cannot merge, in fp32 or after the bf16 / fp16 store).  An
    element-0 victim (zeroed by an outlier at element 1) is not an outlier itself.  Returns a 1-element int32 tensor."""
    import numpy as np
    import torch
    mag = np.abs(plan.grid)
    if not (mag > 32).any():
        return torch.zeros(1, dtype=torch.int32, device=out_block0.device)
    vout = float(mag[mag > 32].min())
    s0 = alpha0.reshape(-1)[:1].float() / gmax                       # fp32 division, as AQ:536 / OQ:296
    thr = (s0 * vout).to(out_block0.dtype)
    return (out_block0.reshape(-1)[:1].abs() >= thr).to(torch.int32)


def apply_wrap_flag(x_block, out_block, alpha_last, plan, gmax, flag, plain_fn=None):
    """Last rank: redo the LAST element of its block.  Zero if `flag` (element 0 is an outlier), else its plain
    fake-quant value (the local launch may have zeroed it against the wrong partner).  No host sync.
    plain_fn(x1, alpha1) -> fake-quant of one element without the pair rule; default: the HIP kernel."""
    import torch
    xl = x_block.reshape(-1)[-1:].contiguous()
    al = alpha_last.reshape(-1)[-1:].float().contiguous()
    if plain_fn is None:
        from . import _lib
        plain = _lib.fakequant(xl, al, plan, gmax, 1, 1, True, ovp=False)
    else:
        plain = plain_fn(xl, al)
    out_block.reshape(-1)[-1:] = torch.where(flag.to(torch.bool), torch.zeros_like(plain), plain)


def fix_odd_numel_wrap(x_block, out_block, alpha_block, plan, gmax, rows, row_len, rank, world, group=None, plain_fn=None):
    """Call after the per-rank OVP launch on a row block of a [rows, row_len] tensor: a no-op unless the tensor's element
    count is odd and it is actually sharded; then one int32 travels from rank 0 to the last rank (broadcast)."""
    if world == 1 or (rows * row_len) % 2 == 0:
        return
    import torch
    import torch.distributed as dist
    flag = wrap_flag(out_block, alpha_block, plan, gmax) if rank == 0 else \
        torch.zeros(1, dtype=torch.int32, device=out_block.device)
    dist.broadcast(flag, 0, group=group)
    if rank == world - 1:
        apply_wrap_flag(x_block, out_block, alpha_block, plan, gmax, flag, plain_fn)


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a python float over the process group (identity without one)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
