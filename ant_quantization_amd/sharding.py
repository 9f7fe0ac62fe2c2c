"""Partitioning of the fake-quant work across the GPUs of a node.

The steady-state path has no exchange step: tensors -- or contiguous row blocks of one
huge tensor -- are independent units, so sharding is pure bookkeeping and there is NO data-path
collective.  What crosses ranks: the barrier around a timed region and a MAX reduction of elapsed
times (bench.py); the reference's own one-shot calibration syncs (AQ/quant_modules.py:525-531) when
the quantiser runs under DDP; one bit for OliVe pairs on a row-sharded tensor with an odd element
count (fix_odd_numel_wrap); and -- the only collectives SURVEY 8e lists -- the first-call
calibration of a PER-TENSOR quantiser whose tensor is row-sharded (sharded_absmax,
sharded_three_sigma, sharded_calibrate: a float MAX, two doubles SUM, T x R doubles SUM).
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


# ---- BASELINE configs[3] / [4]: which rank owns what ----------------------------------------------------------------
def model_linear_shapes(name: str, layers: int = 0) -> List[Tuple[int, int]]:
    """[rows, cols] of the Linear weights of the two sharded BASELINE workloads:
    "opt6.7b"  -- OPT-6.7B, 32 decoder layers x {q, k, v, out: [4096, 4096]; fc1 [16384, 4096]; fc2 [4096, 16384]} = 192
                  tensors, 6.44 G elements;
    "llama70b" -- the synthetic 70 B-parameter stack, 80 x {[8192, 8192] x 2, [1024, 8192] x 2, [28672, 8192] x 2,
                  [8192, 28672]} = 560 tensors, 68.7 G elements.   layers: 0 = the model's own depth."""
    if name == "opt6.7b":
        per = [(4096, 4096)] * 4 + [(16384, 4096), (4096, 16384)]
        return per * (layers or 32)
    if name == "llama70b":
        per = [(8192, 8192)] * 2 + [(1024, 8192)] * 2 + [(28672, 8192)] * 2 + [(8192, 28672)]
        return per * (layers or 80)
    raise ValueError("unknown model %r" % (name,))


def shard_plan(name: str, rank: int, world: int, layers: int = 0) -> List[Tuple[int, int, int, int]]:
    """The units rank `rank` of `world` owns, as (tensor index, first row, end row, cols):
    "opt6.7b"  -- whole tensors, longest-processing-time packing by bytes (lpt_assign): configs[3], "sharded 8 x MI355X";
    "llama70b" -- a contiguous row block of EVERY tensor (row_block, pair-safe): configs[4], the 70 B row blocks -- each
                  rank streams 1/world of every matrix.
    No unit is shared and none is left out; nothing crosses ranks on the data path (per-row alpha: a row block carries its
    own scales; OliVe pairs live inside a row because every row length is even)."""
    shapes = model_linear_shapes(name, layers)
    if name == "opt6.7b":
        mine = lpt_assign([r * c for r, c in shapes], world)[rank]
        return [(i, 0, shapes[i][0], shapes[i][1]) for i in mine]
    out = []
    for i, (r, c) in enumerate(shapes):
        b, e = row_block(r, rank, world, pair_safe_row_len=c)
        if e > b:
            out.append((i, b, e, c))
    return out


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


# ---- a PER-TENSOR quantiser on a row-sharded tensor: the path's only collectives -----------------------
# A per-tensor (activation) quantiser whose tensor is split into row blocks across ranks calibrates on statistics of the
# WHOLE tensor.  Every one of them is a sum or a maximum of per-block terms, so each rank reduces its own block on one
# read (the kernels the unsharded calibration uses) and a few hundred bytes cross the ranks:
#     abs-max (ANT, AQ/quant_modules.py:308-324, :473-477)              1 float            all_reduce(MAX)
#     (sum x, sum x^2) for OliVe's 3-sigma rule (OQ:213-218)            2 doubles          all_reduce(SUM)
#     squared-error sums of the clip candidates, T types x R ratios     T x R doubles      all_reduce(SUM)
# after which every rank holds identical numbers and runs the reference's selection loop (strict '<', first best;
# type = first minimum, AQ:398-415 / OQ:250-256) on its own -- no broadcast of the result is needed.
# `ops` abstracts the per-block kernels (default: the HIP library); the CPU tests of the protocol pass an oracle-backed one.
class GpuBlockOps:
    """Per-block reductions on the GPU (ant_quantization_amd._lib).  x_block: [rows_here, row_len] contiguous device tensor."""

    @staticmethod
    def absmax(x_block):
        from . import _lib
        return _lib.absmax(x_block, x_block.shape[0], x_block.shape[1], per_row=False)              # [1] float32

    @staticmethod
    def moments(x_block):
        from . import _lib
        return _lib.moments(x_block, x_block.shape[0], x_block.shape[1], per_row=False)             # [1, 2] float64

    @staticmethod
    def xmax_3sigma(x_block, sums, n_total):
        from . import _lib
        return _lib.xmax_3sigma(x_block, 1, n_total, per_row=False, sums=sums)                      # [1] float32

    @staticmethod
    def ratios(lb, ub, step, device):
        from . import core
        return core._ratios(int(lb), int(ub), int(step), device)

    @staticmethod
    def search_sse(x_block, xmax, ratios, plans, gmaxs, ovp):
        """[T, R, 1] float64: every type's / ratio's sum of squared errors over THIS block."""
        import torch
        from . import _lib
        rows, row_len = x_block.shape
        out = []
        for b in range(0, len(plans), 4):
            sse = _lib.search_sse_multi(x_block, rows, row_len, xmax, False, ratios, plans[b:b + 4], gmaxs[b:b + 4], ovp=ovp) \
                if len(plans) > 1 else None
            if sse is None:
                sse = torch.stack([_lib.search_sse(x_block, rows, row_len, xmax, False, ratios, p, g, ovp=ovp)
                                   for p, g in zip(plans[b:b + 4], gmaxs[b:b + 4])])
            out.append(sse)
        return torch.cat(out)

    @staticmethod
    def pick(sse, xmax, ratios, n_total):
        from . import _lib
        return _lib.search_pick(sse, xmax, ratios, n_total)                                         # (best [1], alpha [1]) float32


def _all_reduce(t, op, group):
    """all_reduce of `t` in place over `group`: a torch.distributed process group (None = the default one; RCCL on the
    GPUs of a node, gloo in the CPU tests), or any object with an all_reduce(tensor, "max" | "sum") method (the tests'
    in-process communicator that stands in for two ranks on one device)."""
    import torch.distributed as dist
    if group is not None and hasattr(group, "all_reduce") and not isinstance(group, dist.ProcessGroup):
        group.all_reduce(t, "max" if op == dist.ReduceOp.MAX else "sum")
    elif dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=op, group=group)
    return t


def sharded_absmax(x_block, group=None, ops=GpuBlockOps):
    """abs-max of the whole row-sharded tensor as a 1-element float32 tensor on every rank: local abs-max, all_reduce(MAX)
    (AQ/quant_modules.py:308-324: x_max of a per-tensor quantiser).  NaN propagates like torch.max, BY CONSTRUCTION: what
    crosses the ranks is the float's bit pattern with the sign cleared, as int32, reduced with the INTEGER maximum -- the
    ordering the kernel's own atomicMax uses (csrc/antq_k_aux.h): non-negative floats order like their patterns, +Inf
    (0x7f800000) above every finite value, every NaN (0x7f800001 ..) above +Inf.  No backend's floating-point MAX is
    trusted with a NaN (IEEE maxNum drops it; RCCL's float MAX was never run on one here -- DESIGN section 7)."""
    import torch
    import torch.distributed as dist
    local = ops.absmax(x_block).reshape(1).to(torch.float32).clone()
    bits = local.view(torch.int32) & 0x7FFFFFFF
    _all_reduce(bits, dist.ReduceOp.MAX, group)
    return bits.view(torch.float32)


def sharded_three_sigma(x_block, n_total, group=None, ops=GpuBlockOps):
    """OliVe's clip statistic max(|mean + 3 std|, |mean - 3 std|) (OQ:213-218, unbiased std) of the whole row-sharded tensor:
    (sum x, sum x^2) of the block in double on one read, all_reduce(SUM) of the two doubles, then the same roundings the
    unsharded statistic applies (antq_xmax_3sigma).  n_total = element count of the WHOLE tensor."""
    import torch.distributed as dist
    sums = _all_reduce(ops.moments(x_block).clone(), dist.ReduceOp.SUM, group)
    return ops.xmax_3sigma(x_block, sums, int(n_total))


def sharded_calibrate(x_block, n_total, plans, gmaxs, lb, ub, step, statistic="absmax", ovp=False, group=None, ops=GpuBlockOps):
    """First-call calibration of a PER-TENSOR quantiser whose tensor is row-sharded (AQ:328-415 + :287-326, OQ:189-256):
    the clip statistic, every candidate codebook's clip search and the type choice from all-reduced block sums.  Every rank
    returns the same dict: xmax [1], ratios [R], alpha [T] (best clip per type), score [T] (its MSE), type (python int:
    index of the winning codebook; the reference's argsort()[0] = first minimum) -- bit-identical to what the unsharded
    calibration computes from the same sums (the all-reduce adds the blocks' doubles in rank order on every rank).
    statistic: "absmax" (ANT) or "3sigma" (OliVe).  An empty candidate range keeps alpha = xmax, score 1e10 (the reference's
    loop body never runs)."""
    import torch
    import torch.distributed as dist
    if statistic == "absmax":
        xmax = sharded_absmax(x_block, group, ops)
    elif statistic == "3sigma":
        xmax = sharded_three_sigma(x_block, n_total, group, ops)
    else:
        raise ValueError("statistic must be 'absmax' or '3sigma'")
    xmax = xmax.reshape(1).to(torch.float32).contiguous()
    nt = len(plans)
    if not list(range(int(lb), int(ub), int(step))):
        return dict(xmax=xmax, ratios=None, alpha=xmax.repeat(nt), score=torch.full((nt,), 1e10, dtype=torch.float32, device=xmax.device), type=0)
    ratios = ops.ratios(lb, ub, step, x_block.device)
    sse = _all_reduce(ops.search_sse(x_block, xmax, ratios, list(plans), list(gmaxs), ovp).contiguous(), dist.ReduceOp.SUM, group)
    best, alpha = [], []
    for t in range(nt):
        b, a = ops.pick(sse[t], xmax, ratios, int(n_total))
        best.append(b.reshape(()))
        alpha.append(a.reshape(()))
    score = torch.stack(best)
    # the reference: np.argsort(mse_list)[0] (AQ:411-415 / OQ:254-256): the FIRST minimum (NaN scores sort last)
    s = score.detach().cpu().numpy()
    import numpy as np
    return dict(xmax=xmax, ratios=ratios, alpha=torch.stack(alpha), score=score, type=int(np.argsort(s, kind="stable")[0]))
