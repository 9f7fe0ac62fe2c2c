"""Fused building blocks shared by the ANT and OliVe host mirrors.

Everything here launches hand-written gfx950 kernels through the C ABI (`_lib`); the only
torch ops are O(rows) bookkeeping on tiny tensors (arg-min over the clip candidates,
divisions of a few scalars).  No `.item()` / `if cuda_tensor:` in the steady-state
forward: the reference's hidden device->host syncs are gone.
"""
import numpy as np
import torch

from . import _lib


def view_rows(t, per_channel):
    """(rows, row_len) of the reference's `tensor.view(shape[0], -1)` (AQ:538-539) or the
    flat per-tensor view."""
    if per_channel and t.dim() >= 1 and t.shape[0] > 0:
        rows = t.shape[0]
        return rows, t.numel() // rows
    return 1, t.numel()


class FakeQuantSTE(torch.autograd.Function):
    """out = ((q - d).detach() + d) * s with d = x / s, s = alpha / gmax  (AQ:535-551).

    Forward: one fused kernel.  Backward (QAT, ANT only): the straight-through estimator
    the reference's autograd graph yields -- d out/d x = 1 (no clip mask, so the incoming gradient
    is passed through untouched), and d out/d alpha = sum_row g * (q - d) / gmax
    = sum_row g * (out - x) / alpha: one fused reduction kernel (`antq_alpha_grad`).
    """

    @staticmethod
    def forward(ctx, x, alpha, plan, gmax, per_channel, ovp):
        xc = x.contiguous()
        rows, row_len = view_rows(xc, per_channel)
        a = alpha.detach().reshape(-1).to(torch.float32).contiguous()
        out = _lib.fakequant(xc, a, plan, gmax, rows, row_len, per_channel, ovp=ovp)
        ctx.save_for_backward(xc, out, alpha)
        ctx.per_channel = per_channel
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, alpha = ctx.saved_tensors
        gx = g if ctx.needs_input_grad[0] else None
        ga = None
        if ctx.needs_input_grad[1]:
            rows, row_len = view_rows(x, ctx.per_channel)
            gc = g.contiguous()
            if gc.dtype != x.dtype:
                gc = gc.to(x.dtype)
            gsum = _lib.alpha_grad(x, out, gc, rows, row_len, ctx.per_channel)       # one fused reduction kernel
            ga = (gsum.reshape(alpha.shape) / alpha.detach().double()).to(alpha.dtype)
        return gx, ga, None, None, None, None


_grid64_cache = {}


def fake_quant_f64(x, alpha, plan, gmax, per_channel, ovp=False):
    """float64 tensors (a `.double()` model).  The reference's operator is dispatched for double and narrows to float
    INSIDE the kernel (KQ/quant_kernel.cu:51, :28), everything around it runs in double: that is exactly what happens
    here -- fused into one kernel when no gradient is wanted (antq_fakequant_f64), otherwise the reference's own op sequence
    (AQ:535-551 / OQ:294-330) in torch float64 around `antq_nearest` (F64: same narrowing), seven launches, through which
    autograd flows as it does in the reference (ANT's straight-through graph).  Both forms give the same bits."""
    if not (torch.is_grad_enabled() and (x.requires_grad or alpha.requires_grad)) and x.is_cuda:
        # no gradient wanted: the same sequence as ONE kernel (antq_fakequant_f64, round 5)
        xc = x.detach().contiguous()
        rows, row_len = view_rows(xc, per_channel)
        a = alpha.detach().reshape(-1).to(torch.float64).contiguous()
        return _lib.fakequant_f64(xc, a, plan, gmax, rows, row_len, per_channel, ovp=ovp).view(x.shape)
    key = (plan.grid.tobytes(), x.device.index)
    grid = _grid64_cache.get(key)
    if grid is None:
        grid = _grid64_cache[key] = torch.from_numpy(plan.grid.astype(np.float64)).to(x.device)
    gkey = ("gmax", float(gmax), x.device.index)
    gm = _grid64_cache.get(gkey)
    if gm is None:                                 # a 0-dim DEVICE tensor, like the reference's torch.max(quant_grid): a python
        gm = _grid64_cache[gkey] = torch.tensor(float(gmax), dtype=torch.float64, device=x.device)   # scalar would be multiplied by its reciprocal
    scale = alpha.to(torch.float64) / gm
    data = (x.view(x.shape[0], -1) / scale).view(x.shape) if per_channel else x / scale
    with torch.no_grad():
        q = _lib.nearest(data.detach().reshape(-1).contiguous(), grid)
        if ovp:                                   # OQ:311-320 on the flat tensor
            mask = q.abs() > 32
            victim_odd = torch.roll(mask, 1, -1)
            victim_odd[::2] = 0
            victim_even = torch.roll(mask & (~victim_odd), -1, -1)
            victim_even[1::2] = 0
            q = q * (~(victim_even | victim_odd))
        q = q.view(data.shape)
    t = (q - data).detach() + data
    return (t.view(t.shape[0], -1) * scale).view(x.shape) if per_channel else t * scale


def _calib_view(x):
    """Calibration kernels take fp32 / bf16 / fp16: a float64 tensor is calibrated on its float32 image (the statistics and
    the MSE scores of the clip search in single precision; the forward itself then runs in double, fake_quant_f64)."""
    return x.detach().float() if x.dtype == torch.float64 else x


def fake_quant(x, alpha, plan, gmax, per_channel, ovp=False, unordered=False, out=None):
    """Steady-state Quantizer._forward.  Uses autograd only when a gradient is wanted.
    unordered (inference only, needs `out`): x and alpha are at rest -- nothing still in flight on the stream writes them --
    and `out` is a buffer nothing in flight touches, so the launch may overlap the tail of the launches queued before it
    (ANTQ_FLAG_UNORDERED; see Quantizer.weights_at_rest)."""
    if x.dtype == torch.float64:
        return fake_quant_f64(x, alpha, plan, gmax, per_channel, ovp)
    if torch.is_grad_enabled() and (x.requires_grad or alpha.requires_grad):
        return FakeQuantSTE.apply(x, alpha, plan, gmax, per_channel, ovp)
    xc = x.detach().contiguous()
    rows, row_len = view_rows(xc, per_channel)
    a = alpha.detach().reshape(-1).to(torch.float32).contiguous()
    # (x or alpha converted on the way -- a non-contiguous weight, an alpha that model.bfloat16() turned into bf16 -- are
    #  tensors a kernel still in flight is writing: that launch must stay ordered)
    unordered = unordered and out is not None and xc.data_ptr() == x.data_ptr() and a.data_ptr() == alpha.data_ptr()
    return _lib.fakequant(xc, a, plan, gmax, rows, row_len, per_channel, ovp=ovp, unordered=unordered, out=out)


def clip_search(x, x_max, per_channel, lo, hi, step, plan, gmax, ovp=False):
    """search_mse (AQ:287-326 / OQ:189-233) without materialising a single quantised tensor.

    x_max: float32 device tensor with `rows` entries (per-channel) or 1.
    Returns (best_score [rows or 1] float32, best_alpha same shape, ratios float32 tensor).
    Candidates i in range(lo, hi, step) use alpha_i = x_max * fl32(i * 0.01); the first
    strict minimum wins, per row for weights and per tensor for activations.
    """
    xc = _calib_view(x).detach().contiguous()
    rows, row_len = view_rows(xc, per_channel)
    cand = list(range(int(lo), int(hi), int(step)))
    if not cand:
        # the reference's loop body never runs: best_score stays 1e10, alpha = x_max
        na = rows if per_channel else 1
        return torch.full((na,), 1e10, dtype=torch.float32, device=x.device), x_max.clone(), None
    ratios = _ratios(int(lo), int(hi), int(step), x.device)
    xm = x_max.reshape(-1).to(torch.float32).contiguous()
    sse = _lib.search_sse(xc, rows, row_len, xm, per_channel, ratios, plan, gmax, ovp=ovp)  # [ncand, na] f64
    # the reference's selection loop (best = 1e10, strict '<', candidates in ascending order) on the device
    best_score, best_alpha = _lib.search_pick(sse, xm, ratios, row_len)
    return best_score, best_alpha, ratios


def clip_search_types(x, x_max, per_channel, lo, hi, step, plans, gmaxs, ovp=False):
    """search_mse for SEVERAL codebooks (the type selection, AQ:328-415 / OQ:235-256) on one read of the tensor: a list
    of (best_score, best_alpha, ratios) per plan, exactly what clip_search returns for each -- or None when there is no
    single-read path for this shape / these plans (the caller then searches type by type)."""
    if len(plans) < 2 or not list(range(int(lo), int(hi), int(step))):
        return None
    xc = _calib_view(x).detach().contiguous()
    rows, row_len = view_rows(xc, per_channel)
    ratios = _ratios(int(lo), int(hi), int(step), x.device)
    xm = x_max.reshape(-1).to(torch.float32).contiguous()
    out = []
    for b in range(0, len(plans), 4):
        sse = _lib.search_sse_multi(xc, rows, row_len, xm, per_channel, ratios, plans[b:b + 4], gmaxs[b:b + 4], ovp=ovp)
        if sse is None:
            return None
        for t in range(sse.shape[0]):
            best_score, best_alpha = _lib.search_pick(sse[t], xm, ratios, row_len)
            out.append((best_score, best_alpha, ratios))
    return out


class SearchMemo:
    """Per-tensor clip searches of the SAME activation tensor are run once.

    Parallel branches calibrate on one tensor object: the query / key / value projections of an attention block share their
    input, a ResNet block's first convolution and its downsample path do too.  Each wrapper owns an input quantiser, all in
    the same state on the first forward, so the reference repeats the identical 75-candidate search three times (BERT-base:
    24 of its 72 activation searches).  The kernels are deterministic -- one fixed summation order -- so the repeat returns
    the same bits; it is looked up instead.

    An entry is keyed by the tensor OBJECT (held weakly), its storage address and version counter, and by everything else the
    search depends on (the caller's key: statistic, window, step, codebook bytes, gmax, pair rule).  Only per-tensor searches
    of tensors that are not Parameters are kept (a weight has one quantiser; and Parameters are what `.data` edits, which
    move no version counter, usually touch).  A handful of entries, newest first; the forward hook enable_quantization
    registers on the model clears them when the model's forward returns."""

    def __init__(self, keep=4):
        self.keep, self.entries, self.enabled, self.hits = keep, [], True, 0

    @staticmethod
    def _stamp(t):
        if not isinstance(t, torch.Tensor) or isinstance(t, torch.nn.Parameter) or torch.is_inference(t):
            return None
        return (t.data_ptr(), t._version, t.dtype, tuple(t.shape), tuple(t.stride()))

    def get(self, tensor, key):
        st = self._stamp(tensor) if self.enabled else None
        if st is None:
            return None
        for ref, stamp, k, value in self.entries:
            if ref() is tensor and stamp == st and k == key:
                self.hits += 1
                return value
        return None

    def put(self, tensor, key, value):
        st = self._stamp(tensor) if self.enabled else None
        if st is None:
            return
        import weakref
        self.entries = [e for e in self.entries if e[0]() is not None][:self.keep - 1]
        self.entries.insert(0, (weakref.ref(tensor), st, key, value))

    def clear(self):
        self.entries = []


search_memo = SearchMemo()


_ratio_cache = {}


def _ratios(lo, hi, step, device):
    """fl32(i * 0.01) for i in range(lo, hi, step) as a device tensor (cached: the ranges are fixed per run)."""
    key = (lo, hi, step, device.index)
    t = _ratio_cache.get(key)
    if t is None:
        arr = np.asarray([float(np.float32(i * 0.01)) for i in range(lo, hi, step)], dtype=np.float32)
        t = torch.from_numpy(arr).to(device)
        _ratio_cache[key] = t
    return t


_grid_cache = {}


def device_grid(values, device):
    """A fresh device tensor holding `values` (host float32 array) without a host->device copy per call: one resident
    master per (bytes, device), cloned on the device.  Calibration installs 2-5 codebooks per quantiser; a pageable
    H2D copy each would serialise the host with whatever the GPU is still doing."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    key = (v.tobytes(), device.type, device.index)
    t = _grid_cache.get(key)
    if t is None:
        t = torch.from_numpy(v.copy()).to(device)
        _grid_cache[key] = t
    return t.clone()


def device_grid_master(values, device):
    """The resident master copy itself (read-only use: a stack of candidate codebooks a kernel gathers a row from)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    key = (v.tobytes(), device.type, device.index)
    t = _grid_cache.get(key)
    if t is None:
        t = torch.from_numpy(v.copy()).to(device)
        _grid_cache[key] = t
    return t


_absmax_memo = [None]          # (tensor, (version, address, dtype, shape, strides), per_channel, result) of the last call


def forget_absmax():
    """End of a calibration: drop the memo (a later `.data` edit of the same tensor object would not move its version)."""
    _absmax_memo[0] = None


def row_absmax(x, per_channel):
    """max |x| per row / per tensor as float32 (AQ:473-477, :289 / :308).  The first-call calibration asks for it twice on
    the same tensor (the initial alpha, then search_mse's x_max): the second call is answered from a one-entry memo keyed by
    the tensor object and its version counter (a weak reference: nothing is kept alive)."""
    import weakref
    m = _absmax_memo[0]
    ver = None if torch.is_inference(x) else (x._version, x.data_ptr(), x.dtype, tuple(x.shape), tuple(x.stride()))
    if m is not None and m[0]() is x and m[1] == ver and ver is not None and m[2] == bool(per_channel):
        return m[3].clone()            # (callers store the result into alpha.data: never hand the memo's own tensor out twice)
    xc = _calib_view(x).detach().contiguous()
    rows, row_len = view_rows(xc, per_channel)
    out = _lib.absmax(xc, rows, row_len, per_row=per_channel)
    try:
        _absmax_memo[0] = (weakref.ref(x), ver, bool(per_channel), out.clone())
    except TypeError:
        _absmax_memo[0] = None
    return out
