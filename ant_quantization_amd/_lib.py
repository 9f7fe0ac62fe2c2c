"""ctypes binding of libantq.so (include/antq.h) for torch tensors.

PyTorch is plumbing here: it owns the HBM buffers and the HIP stream; every
computation is done by the hand-written gfx950 kernels behind the C ABI.
There is NO CPU or PyTorch fallback: if the shared library is missing, or a
tensor is not on a HIP device, the call raises.
"""
import ctypes
import os
import threading

import numpy as np
import torch  # must be imported before libantq.so so that ONE libamdhip64 is shared

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANTQ_LIB") or os.path.join(_HERE, "libantq.so")   # ANTQ_LIB: A/B builds (dev)

ABI_VERSION = 7         # include/antq.h ANTQ_ABI_VERSION this binding was written against
F32, BF16, F16, F64 = 0, 1, 2, 3
FLAG_OVP = 1
FLAG_DYNAMIC = 2
FLAG_UNORDERED = 4     # antq_fakequant: the launch may start while earlier launches of the stream drain (inputs at rest)
IDX_NONE = -1
IDX_VICTIM = -2
MAX_GRID = 1024
PLAN_MAX_BYTES = 128 + 4 * MAX_GRID + 16 * 3072 + 20 * 1024 + 16 * 64

_DTYPES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, torch.float64: F64}

_lib = None
_lock = threading.RLock()        # (re-entrant: ext() loads the library under it)


class AntqError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise AntqError(
                        "libantq.so not found at %s -- build it with `make -C %s/csrc` "
                        "(or __graft_entry__.build()); there is no CPU fallback" % (LIB_PATH, _HERE))
                L = ctypes.CDLL(LIB_PATH)
                L.antq_abi_version.restype = ctypes.c_int
                if L.antq_abi_version() != ABI_VERSION:
                    raise AntqError("%s speaks C ABI version %d, this binding version %d: the argument lists differ -- "
                                    "rebuild it (make -C %s/csrc)" % (LIB_PATH, L.antq_abi_version(), ABI_VERSION, _HERE))
                L.antq_strerror.restype = ctypes.c_char_p
                for name in ("antq_abi_version", "antq_nearest", "antq_plan_build", "antq_plan_kind",
                             "antq_plan_bytes", "antq_plan_eval_host", "antq_fakequant",
                             "antq_fakequant_dynamic", "antq_absmax", "antq_search_sse", "antq_affine",
                             "antq_copy", "antq_batch_build", "antq_fakequant_batch", "antq_encode4", "antq_decode4",
                             "antq_search_pick", "antq_alpha_grad", "antq_nearest_plan", "antq_nearest_hinted",
                             "antq_search_sse_multi", "antq_plan_eval_host_a", "antq_moments", "antq_xmax_3sigma",
                             "antq_calibrate", "antq_prefetch_kernels", "antq_plan_eval_host_h", "antq_calibrate_batch", "antq_absmax_into", "antq_fakequant_f64",
                             "antq_absmax_t", "antq_alpha_grad_t", "antq_calibrate_install"):
                    getattr(L, name).restype = ctypes.c_int
                L.antq_batch_capacity.restype = ctypes.c_size_t
                L.antq_search_workspace_bytes.restype = ctypes.c_size_t
                L.antq_calibrate_workspace_bytes.restype = ctypes.c_size_t
                L.antq_calibrate_batch_workspace_bytes.restype = ctypes.c_size_t
                # declared signatures: plain python ints go straight through (no per-call wrapper objects)
                vp, sz, ci, cf, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_uint
                L.antq_fakequant.argtypes = [vp, vp, vp, sz, sz, vp, ci, cf, vp, vp, cu, ci, vp]
                L.antq_fakequant_dynamic.argtypes = [vp, vp, vp, vp, sz, sz, cf, cf, vp, vp, cu, ci, vp]
                L.antq_nearest.argtypes = [vp, vp, vp, sz, vp, ci, ci, vp]
                L.antq_absmax.argtypes = [vp, vp, sz, sz, ci, ci, vp]
                L.antq_absmax_into.argtypes = [vp, vp, sz, ci, vp]
                L.antq_absmax_t.argtypes = [vp, vp, sz, ci, vp, vp]
                L.antq_alpha_grad_t.argtypes = [vp, vp, vp, sz, vp, ci, vp, vp]
                L.antq_alpha_grad.argtypes = [vp, vp, vp, sz, sz, ci, vp, vp, ci, vp]
                L.antq_calibrate_install.argtypes = [vp, vp, sz, ci, ci, vp, vp, vp, cu, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp]
                L.antq_copy.argtypes = [vp, vp, sz, vp]
                # development: ANTQ_DEBUG_KNOBS="14=2,0=8" applies antq_debug_set(key, value) pairs to the loading thread
                # (the knobs are thread-local; tools/fuzz_campaign.sh forces code paths with it)
                for kv in [k for k in os.environ.get("ANTQ_DEBUG_KNOBS", "").split(",") if "=" in k]:
                    L.antq_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
                _lib = L
    return _lib


_prefetched = set()


def prefetch_kernels(device=None):
    """Load the library's GPU code objects for `device` (default: the current one) now instead of inside the first
    calibrating forward (antq_prefetch_kernels; ~50 ms, once per process and device).  A no-op without a GPU."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx in _prefetched:
        return
    with torch.cuda.device(idx):
        lib().antq_prefetch_kernels()
    _prefetched.add(idx)


_prewarmed = set()


def prewarm(device, dtype=torch.float32):
    """Run the calibration and steady-state entry points once on a toy tensor (256 x 1024 of `dtype`, per row and per
    tensor, with and without the pair rule): HIP resolves every kernel at its FIRST launch (1-3 ms each, ~50 ms for the
    dozen a calibrating forward uses) and torch does the same for the few elementwise kernels around them -- paid here, at
    enable_quantization time, instead of inside the first forward.  Once per (device, dtype); a no-op without a GPU."""
    if not torch.cuda.is_available() or dtype not in _DTYPES or _DTYPES[dtype] == F64:
        return
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), dtype)
    if key in _prewarmed:
        return
    _prewarmed.add(key)
    from . import grids
    prefetch_kernels(device)
    with torch.no_grad():
        # (a deterministic toy tensor: the global RNG stream of a seeded run is not advanced by quantising the model)
        x = (torch.sin(torch.arange(256 * 1024, device=device, dtype=torch.float32) * 0.37) * 0.1).view(256, 1024).to(dtype)
        plans_a = [plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "flint")]
        on = grids.olive_grid("flint", 4, True)
        plan_o = plan_for(np.concatenate([on, grids.olive_outliers(4, True)]))
        for per_row in (True, False):
            a, _, _, xm = calibrate(x, 256, 1024, per_row, plans_a, [10.0, 10.0], 95, 100, 1, xmax="absmax")
            calibrate(x, 256, 1024, per_row, plans_a[:1], [10.0], 95, 100, 1, xmax="absmax")
            calibrate(x, 256, 1024, per_row, [plan_o], [float(on.max())], 95, 100, 2, xmax="3sigma", ovp=True)
            fakequant(x, a[0].contiguous(), plans_a[0], 10.0, 256, 1024, per_row)
            fakequant(x, a[0].contiguous(), plan_o, float(on.max()), 256, 1024, per_row, ovp=True)
            al = absmax(x, 256, 1024, per_row)
            (al / xm).mean()
        (x.min() < 0).item()
        # rows of >= 512 vectors take the 8-vector search tasks; the sign probe's pinned slots and its async copy
        xl = x.view(64, 4096)
        calibrate(xl, 64, 4096, True, plans_a[:1], [10.0], 95, 100, 1, xmax="absmax")
        calibrate(xl, 64, 4096, True, [plan_o], [float(on.max())], 95, 100, 2, xmax="3sigma", ovp=True)
        from . import _mirror
        s1 = _mirror._slots.take()
        s1.copy_(x.min().float().reshape(1), non_blocking=True)
        torch.ones_like(al), al.clone()
        # the device-side type pick of input quantisers (_mirror.CalibrationMixin._calibrate_deferred): its gathers and casts
        a3, s3, typ, _ = calibrate(x, 1, x.numel(), False, plans_a, [10.0, 10.0], 95, 100, 1, xmax="absmax")
        stack = torch.zeros(2, 16, device=device)
        calibrate_install(x, torch.empty_like(x), plans_a, [10.0, 10.0], typ, a3.reshape(-1), s3.reshape(-1), stack,
                          torch.empty(16, device=device), torch.empty(1, device=device), torch.empty(1, device=device))
        calibrate_install(x, torch.empty_like(x), [plan_o, plan_o], [float(on.max())] * 2, typ, a3.reshape(-1), s3.reshape(-1), stack,
                          torch.empty(16, device=device), torch.empty(1, device=device), torch.empty(1, device=device), ovp=True)
        idx = typ.long()
        a3.index_select(0, idx).reshape(())
        torch.stack([x, x]).index_select(0, idx)
        torch.zeros(2, 16, device=device).index_select(0, idx)[0]
        s2 = _mirror._slots.take()
        s2.copy_(typ.float(), non_blocking=True)
        torch.cuda.synchronize(device)
        _mirror._slots.give(s1)
        _mirror._slots.give(s2)


_ext_mod = False          # False: not tried yet; None: unavailable


def ext():
    """The compiled torch extension (csrc/antq_torch.cpp -> _antq_ext*.so: the counterpart of the reference's pybind11
    `quant_cuda` module plus compiled fast paths of the fused entry points), or None: not built, switched off with
    ANTQ_NO_EXT=1, or ANTQ_LIB points at another library build (the extension is linked against the in-tree one).  Every
    caller falls back to the ctypes binding of the same C ABI -- same kernels, ~4 us more host time per call."""
    global _ext_mod
    if _ext_mod is False:
        with _lock:
            if _ext_mod is False:
                mod = None
                if os.environ.get("ANTQ_NO_EXT") != "1" and not os.environ.get("ANTQ_LIB"):
                    lib()                                   # the library first: ONE libantq / libamdhip64 in the process
                    try:
                        from . import _antq_ext as mod
                    except ImportError:
                        try:
                            import importlib
                            mod = importlib.import_module("ant_quantization_amd._antq_ext")
                        except ImportError:
                            mod = None
                    if mod is not None and mod.abi_version() != ABI_VERSION:
                        mod = None
                _ext_mod = mod
    return _ext_mod


def _check(rc, what):
    if rc != 0:
        raise AntqError("%s failed: %s (%d)" % (what, lib().antq_strerror(rc).decode(), rc))


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_int(dev):
    """Raw hipStream_t of torch's current stream on `dev` as a python int (fast path)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (saves ~3 us per launch)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _require_out(out, x):
    """A caller-supplied output buffer: same device, dtype and element count as x, contiguous."""
    if not (out.is_cuda and out.device == x.device and out.dtype == x.dtype and out.numel() == x.numel()
            and out.is_contiguous()):
        raise AntqError("out must be a contiguous tensor of x's dtype, element count and device")


def _require_gpu(t, name):
    if not t.is_cuda:
        raise AntqError("%s must live on a HIP device (got %s); libantq has no CPU path" % (name, t.device))
    if not t.is_contiguous():
        raise AntqError("%s must be contiguous" % name)


# ---------------------------------------------------------------------------------
# plans
# ---------------------------------------------------------------------------------
class Plan:
    """Host blob + lazily uploaded per-device copies of one grid's decision table."""

    def __init__(self, grid):
        g = np.ascontiguousarray(np.asarray(grid, dtype=np.float32).reshape(-1))
        if g.size < 1 or g.size > MAX_GRID:
            raise AntqError("grid must have 1..%d entries" % MAX_GRID)
        buf = np.zeros(PLAN_MAX_BYTES, dtype=np.uint8)
        n = lib().antq_plan_build(g.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(g.size),
                                  buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
        if n <= 0:
            _check(n, "antq_plan_build")
        self.grid = g
        self.host = buf[:n].copy()
        self.host_addr = self.host.ctypes.data          # stays valid: self.host is never reallocated
        self.kind = int(lib().antq_plan_kind(self.host.ctypes.data_as(ctypes.c_void_p)))
        self._dev = {}
        self._dev_ptr = {}       # device index -> address of the device copy (the extension's fast path)

    @property
    def is_table(self):
        return self.kind == 1

    def host_ptr(self):
        return self.host.ctypes.data_as(ctypes.c_void_p)

    def dev(self, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        t = self._dev.get(key)
        if t is None:
            t = torch.from_numpy(self.host).to(device)
            self._dev[key] = t
            self._dev_ptr[key] = t.data_ptr()
        return t

    def eval_host(self, d):
        """CPU model of the device element path (test utility)."""
        d = np.ascontiguousarray(d, dtype=np.float32)
        q = np.empty_like(d)
        idx = np.empty(d.shape, dtype=np.int16)
        _check(lib().antq_plan_eval_host(self.host_ptr(), d.ctypes.data_as(ctypes.c_void_p),
                                         q.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.c_size_t(d.size)), "antq_plan_eval_host")
        return q, idx


_plan_cache = {}


def plan_for(grid):
    """Cached Plan keyed by the grid's bytes (grids are tiny; plans are immutable)."""
    g = np.ascontiguousarray(np.asarray(grid, dtype=np.float32).reshape(-1))
    key = g.tobytes()
    p = _plan_cache.get(key)
    if p is None:
        p = Plan(g)
        _plan_cache[key] = p
    return p


# ---------------------------------------------------------------------------------
# entry points
# ---------------------------------------------------------------------------------
def nearest(x, grid, want_idx=False):
    """quant_cuda.quant body: z = nearest grid value of every element of flat x."""
    _require_gpu(x, "x")
    _require_gpu(grid, "grid")
    dt = _DTYPES.get(x.dtype)
    if dt is None:
        raise AntqError("unsupported dtype %s" % x.dtype)
    want_grid = x.dtype if dt in (F32, F64) else torch.float32
    if grid.dtype != want_grid:
        raise AntqError("grid dtype %s, expected %s" % (grid.dtype, want_grid))
    z = torch.empty_like(x)
    idx = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_idx else None
    with torch.cuda.device(x.device):
        _check(lib().antq_nearest(_vp(x), _vp(z), _vp(idx), ctypes.c_size_t(x.numel()), _vp(grid),
                                  ctypes.c_int(grid.numel()), ctypes.c_int(dt), _stream(x.device)), "antq_nearest")
    return (z, idx) if want_idx else z


def nearest_plan(x, plan, want_idx=False):
    """quant_cuda.quant body through a plan (the grid is known on the host): table lookup instead of the scan.
    Falls back to nearest() (the literal scan on the plan's grid) for ragged / unaligned / float64 inputs."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None:
        raise AntqError("unsupported dtype %s" % x.dtype)
    epl = 4 if dt == F32 else 8
    if dt == F64 or x.numel() % epl or x.data_ptr() % 16:
        g = torch.from_numpy(plan.grid).to(x.device)
        return nearest(x, g.to(x.dtype) if dt in (F32, F64) else g, want_idx)
    z = torch.empty_like(x)
    idx = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_idx else None
    pd = plan.dev(x.device)
    with _on_device(x.device):
        _check(lib().antq_nearest_plan(_vp(x), _vp(z), _vp(idx), ctypes.c_size_t(x.numel()),
                                       ctypes.c_void_p(plan.host_addr), _vp(pd), ctypes.c_int(dt), _stream(x.device)),
               "antq_nearest_plan")
    return (z, idx) if want_idx else z


def hinted_ok(x, grid, plan):
    """Can antq_nearest_hinted take this call?  (fp32 / bf16 / fp16 x, fp32 grid of the plan's size, whole 16-byte
    vectors.)  Everything else goes through nearest(): the literal scan."""
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64 or grid.dtype != torch.float32 or grid.numel() != plan.grid.size:
        return False
    return x.numel() % (4 if dt == F32 else 8) == 0 and x.data_ptr() % 16 == 0


def nearest_hinted(x, grid, plan, stale=None, want_idx=False):
    """quant_cuda.quant body on the DEVICE grid `grid`, with `plan` as the host's belief of its contents: the kernel
    verifies the belief (bit compare of the m values) and scans `grid` literally when it is wrong, raising `stale`
    (a pinned host int32 tensor, polled by the caller without a sync)."""
    _require_gpu(x, "x")
    _require_gpu(grid, "grid")
    if not hinted_ok(x, grid, plan):
        raise AntqError("nearest_hinted: unsupported dtype / size / alignment (use nearest)")
    z = torch.empty_like(x)
    idx = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_idx else None
    pd = plan.dev(x.device)
    with _on_device(x.device):
        _check(lib().antq_nearest_hinted(_vp(x), _vp(z), _vp(idx), ctypes.c_size_t(x.numel()), _vp(grid),
                                         ctypes.c_int(grid.numel()), ctypes.c_void_p(plan.host_addr), _vp(pd),
                                         _vp(stale), ctypes.c_int(_DTYPES[x.dtype]), _stream(x.device)),
               "antq_nearest_hinted")
    return (z, idx) if want_idx else z


def fakequant(x, alpha, plan, gmax, rows, row_len, per_row, ovp=False, want_idx=False, out=None, unordered=False):
    """Fused Quantizer._forward on a contiguous tensor viewed as [rows, row_len].
    unordered: the caller promises that x / alpha are not produced by work still in flight on the stream (weights at rest);
    the launch may then overlap the tail of the launches queued before it (ANTQ_FLAG_UNORDERED)."""
    flags = (FLAG_OVP if ovp else 0) | (FLAG_UNORDERED if unordered else 0)
    if unordered and (out is None or want_idx):
        # torch's caching allocator recycles blocks in STREAM ORDER: a fresh tensor may sit on memory that a kernel still
        # in flight is reading or writing -- harmless for an ordered launch, fatal for one that may start early
        raise AntqError("an unordered launch needs a caller-owned output buffer (out=...) that nothing in flight touches")
    e = _ext_mod
    if e is False:
        e = ext()
    if e is not None:
        # the compiled path: every check below is made by the extension as well (TORCH_CHECK -> AntqError)
        pd = plan._dev_ptr.get(x.device.index)
        if pd is None:
            if not x.is_cuda:
                raise AntqError("x must live on a HIP device (got %s); libantq has no CPU path" % x.device)
            pd = plan.dev(x.device).data_ptr()
        try:
            return e.fakequant(x, alpha, plan.host_addr, pd, gmax, rows, row_len, per_row, flags, out, want_idx)
        except RuntimeError as err:
            raise AntqError(str(err).split("\n")[0]) from None
    _require_gpu(x, "x")
    _require_gpu(alpha, "alpha")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    if alpha.dtype != torch.float32:
        raise AntqError("alpha must be float32")
    if rows * row_len != x.numel():
        raise AntqError("rows*row_len != numel")
    if alpha.numel() != (rows if per_row else 1):
        raise AntqError("alpha has %d entries, expected %d" % (alpha.numel(), rows if per_row else 1))
    pd = plan.dev(x.device)
    if out is None:
        out = torch.empty_like(x)
    else:
        _require_out(out, x)
    idx = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_idx else None
    with _on_device(x.device):
        rc = lib().antq_fakequant(x.data_ptr(), out.data_ptr(), _ptr(idx), rows, row_len, alpha.data_ptr(),
                                  1 if per_row else 0, gmax, plan.host_addr, pd.data_ptr(), flags, dt,
                                  _stream_int(x.device))
    if rc:
        _check(rc, "antq_fakequant")
    return (out, idx) if want_idx else out


def fakequant_f64(x, alpha, plan, gmax, rows, row_len, per_row, ovp=False):
    """antq_fakequant_f64: the reference's float64 op sequence around its float-narrowing operator, fused.  x: float64,
    contiguous; alpha: float64 tensor with rows (per_row) / 1 entries."""
    _require_gpu(x, "x")
    if x.dtype != torch.float64 or alpha.dtype != torch.float64 or not x.is_contiguous() or not alpha.is_contiguous():
        raise AntqError("fakequant_f64 takes contiguous float64 tensors")
    if rows * row_len != x.numel() or alpha.numel() != (rows if per_row else 1) or alpha.device != x.device:
        raise AntqError("rows*row_len != numel, or alpha does not hold one value per row / one value")
    out = torch.empty_like(x)
    with _on_device(x.device):
        rc = lib().antq_fakequant_f64(_vp(x), _vp(out), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _vp(alpha),
                                      ctypes.c_int(1 if per_row else 0), ctypes.c_double(float(gmax)), ctypes.c_void_p(plan.host_addr),
                                      ctypes.c_void_p(plan.dev(x.device).data_ptr()), ctypes.c_uint(FLAG_OVP if ovp else 0),
                                      _stream(x.device))
    _check(rc, "antq_fakequant_f64")
    return out


def fakequant_dynamic(x, plan, gmax, rows, row_len, ratio=1.0, ovp=False, want_idx=False, out=None,
                      want_alpha=True):
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    if rows * row_len != x.numel():
        raise AntqError("rows*row_len != numel")
    if out is None:
        out = torch.empty_like(x)
    else:
        _require_out(out, x)
    idx = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want_idx else None
    pd = plan.dev(x.device)
    alpha, rc = None, -1
    with _on_device(x.device):
        if not want_alpha:
            # nobody wants the scales: the single-read kernels then skip the store; only the two-pass fallback for very long
            # or ragged rows needs the buffer as scratch and says so (ANTQ_ERR_ARG)
            rc = lib().antq_fakequant_dynamic(x.data_ptr(), out.data_ptr(), _ptr(idx), None, rows, row_len, ratio, gmax,
                                              plan.host_addr, pd.data_ptr(), FLAG_OVP if ovp else 0, dt, _stream_int(x.device))
        if rc == -1:
            alpha = torch.empty(rows, dtype=torch.float32, device=x.device)
            rc = lib().antq_fakequant_dynamic(x.data_ptr(), out.data_ptr(), _ptr(idx), alpha.data_ptr(), rows, row_len,
                                              ratio, gmax, plan.host_addr, pd.data_ptr(), FLAG_OVP if ovp else 0, dt,
                                              _stream_int(x.device))
    if rc:
        _check(rc, "antq_fakequant_dynamic")
    return out, (alpha if want_alpha else None), idx


REDUCE_WS_BYTES = 65536   # include/antq.h ANTQ_REDUCE_WS_BYTES
_reduce_blocks = {}       # (device index, raw stream) -> the ticket block of antq_absmax_t / antq_alpha_grad_t


def _reduce_ws(device):
    """The caller-owned ticket block of the one-launch whole-tensor reductions: zeroed ONCE here, left zeroed by every call;
    ONE per (device, stream) -- calls on one stream run one after the other and may share it, two streams never do."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _stream_int(device))
    ws = _reduce_blocks.get(key)
    if ws is None:
        with _lock:
            ws = _reduce_blocks.get(key)
            if ws is None:
                if len(_reduce_blocks) >= 64:        # streams come and go
                    _reduce_blocks.clear()
                ws = _reduce_blocks[key] = torch.zeros(REDUCE_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


_zero_pools = {}          # device index -> [float32 zeros, next free slot]


def _zero_slot(device):
    """A one-element float32 tensor holding 0 -- a slot of a pool zeroed once for 4096 calls, never handed out twice (the
    whole-tensor abs-max then needs no launch of its own for that zero: antq_absmax_into).  One pool per (device, stream):
    the fill is ordered before every use of its slots by the stream itself (a pool shared by streams could be used on one
    before the other had zeroed it).  Never reached while a stream is capturing (absmax below)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _stream_int(device))
    pool = _zero_pools.get(key)
    if pool is None or pool[1] >= len(pool[0]):
        if len(_zero_pools) >= 64:               # streams come and go
            _zero_pools.clear()
        # (the 4096 one-element views are cut once per refill, in C++: ~0.15 us per call instead of a Python slice)
        pool = _zero_pools[key] = [torch.zeros(4096, dtype=torch.float32, device=device).split(1), 0]
    pool[1] += 1
    return pool[0][pool[1] - 1]


def absmax(x, rows, row_len, per_row=True):
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    if not per_row and not torch.cuda.is_current_stream_capturing():
        # one launch: the maximum is accumulated into a slot that already holds 0 (a captured graph would replay into the
        # same slot, so captures keep the self-initialising entry below).  The streaming kernel of round 6 reads a 33.5 MB
        # bf16 tensor in 6.5 us -- less than the ctypes path's host time, hence the compiled call
        amax = _zero_slot(x.device)
        e = _ext_mod if _ext_mod is not False else ext()
        if e is not None and hasattr(e, "absmax_into") and x.is_contiguous():
            try:
                e.absmax_into(x, amax)
            except RuntimeError as err:
                raise AntqError(str(err).split("\n")[0]) from None
            return amax
        with _on_device(x.device):
            rc = lib().antq_absmax_into(x.data_ptr(), amax.data_ptr(), rows * row_len, dt, _stream_int(x.device))
        if rc:
            _check(rc, "antq_absmax_into")
        return amax
    amax = torch.empty(rows if per_row else 1, dtype=torch.float32, device=x.device)     # (the entry point initialises it)
    with _on_device(x.device):
        rc = lib().antq_absmax(x.data_ptr(), amax.data_ptr(), rows, row_len, 1 if per_row else 0, dt, _stream_int(x.device))
    if rc:
        _check(rc, "antq_absmax")
    return amax


def moments(x, rows, row_len, per_row=True):
    """[rows or 1, 2] float64: (sum x, sum x^2) per row or for the whole tensor, on ONE read, in a fixed order."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    if rows * row_len != x.numel():
        raise AntqError("rows*row_len != numel")
    sums = torch.empty((rows if per_row else 1, 2), dtype=torch.float64, device=x.device)
    ws = None if per_row else _workspace(x.device)
    with _on_device(x.device):
        _check(lib().antq_moments(_vp(x), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), ctypes.c_int(1 if per_row else 0),
                                  ctypes.c_int(dt), _vp(sums), _vp(ws), _stream(x.device)), "antq_moments")
    return sums


def xmax_3sigma(x, rows, row_len, per_row=True, sums=None):
    """OliVe's clip statistic max(|mean + 3 std|, |mean - 3 std|) (OQ:193-197, :213-218; unbiased std) per row or per
    tensor as float32, with the roundings of x's dtype: one read of x (`sums`: already reduced moments, e.g. all-reduced
    across the ranks of a row-sharded tensor -- then x is not read at all)."""
    dt = _DTYPES.get(x.dtype)
    if sums is None:
        sums = moments(x, rows, row_len, per_row)
    na = sums.shape[0]
    out = torch.empty(na, dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _check(lib().antq_xmax_3sigma(_vp(sums), ctypes.c_size_t(na), ctypes.c_size_t(row_len if per_row else rows * row_len),
                                      ctypes.c_int(dt), _vp(out), _stream(x.device)), "antq_xmax_3sigma")
    return out


def alpha_grad(x, out, gout, rows, row_len, per_row=True):
    """sum over each row (or the tensor) of gout * (out - x), float64: the alpha gradient of fakequant times alpha."""
    for name, t in (("x", x), ("out", out), ("gout", gout)):
        _require_gpu(t, name)
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64 or out.dtype != x.dtype or gout.dtype != x.dtype:
        raise AntqError("alpha_grad: x / out / gout must share one of float32 / bfloat16 / float16")
    gsum = torch.empty(rows if per_row else 1, dtype=torch.float64, device=x.device)
    with _on_device(x.device):
        if per_row or torch.cuda.is_current_stream_capturing():
            ws = None if per_row else _workspace(x.device)
            _check(lib().antq_alpha_grad(_vp(x), _vp(out), _vp(gout), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
                                         ctypes.c_int(1 if per_row else 0), _vp(gsum), _vp(ws), ctypes.c_int(dt),
                                         _stream(x.device)), "antq_alpha_grad")
        else:           # one scale for the tensor: ONE launch (fixed-order tree through the stream's ticket block, ABI 7)
            _check(lib().antq_alpha_grad_t(x.data_ptr(), out.data_ptr(), gout.data_ptr(), rows * row_len, gsum.data_ptr(), dt,
                                           _reduce_ws(x.device).data_ptr(), _stream_int(x.device)), "antq_alpha_grad_t")
    return gsum


_workspaces = {}            # (device index, raw stream) -> scratch of antq_search_workspace_bytes() (48.3 MiB); launches on one stream are ordered, so they may share it


def _workspace(device):
    """Scratch for the workgroup partials of a whole-tensor sum (antq_search_workspace_bytes), ONE per (device, stream):
    kernels that use it on the same stream run one after the other, so a QAT backward does not ask the caching allocator
    for a fresh block per quantiser and step; different streams never share one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _stream_int(device))
    ws = _workspaces.get(key)
    if ws is None:
        with _lock:
            ws = _workspaces.get(key)
            if ws is None:
                if len(_workspaces) >= 16:           # streams come and go: do not grow without bound (40 MiB each)
                    _workspaces.clear()
                ws = _workspaces[key] = torch.empty(int(lib().antq_search_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def _search_workspace(device, rows, per_row):
    if per_row and rows > 1:
        return None
    return _workspace(device)


def search_sse(x, rows, row_len, xmax, per_row, ratios, plan, gmax, ovp=False):
    """sum of squared errors of every clip candidate: [ncand, rows] (or [ncand, 1]) float64."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    ncand = ratios.numel()
    na = rows if per_row else 1
    sse = torch.empty(ncand, na, dtype=torch.float64, device=x.device)
    ws = _search_workspace(x.device, rows, per_row)
    pd = plan.dev(x.device)
    with torch.cuda.device(x.device):
        _check(lib().antq_search_sse(_vp(x), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _vp(xmax),
                                     ctypes.c_int(1 if per_row else 0), _vp(ratios), ctypes.c_int(ncand),
                                     ctypes.c_float(gmax), plan.host_ptr(), _vp(pd),
                                     ctypes.c_uint(FLAG_OVP if ovp else 0), ctypes.c_int(dt), _vp(sse),
                                     _vp(ws), _stream(x.device)), "antq_search_sse")
    return sse


def search_sse_multi(x, rows, row_len, xmax, per_row, ratios, plans, gmaxs, ovp=False):
    """The sums of search_sse for up to 4 codebooks on ONE read of the tensor: [ntypes, ncand, rows] (or [.., 1]) float64;
    None when the launch shape / a plan has no single-read path (the caller then issues one search_sse per type)."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    nt = len(plans)
    if not 1 <= nt <= 4:
        return None
    ncand = ratios.numel()
    na = rows if per_row else 1
    sse = torch.empty(nt, ncand, na, dtype=torch.float64, device=x.device)
    ws = _search_workspace(x.device, rows, per_row)
    ph = (ctypes.c_void_p * nt)(*[p.host_addr for p in plans])
    pd = (ctypes.c_void_p * nt)(*[p.dev(x.device).data_ptr() for p in plans])
    gm = (ctypes.c_float * nt)(*[float(g) for g in gmaxs])
    with _on_device(x.device):
        rc = lib().antq_search_sse_multi(_vp(x), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _vp(xmax),
                                         ctypes.c_int(1 if per_row else 0), _vp(ratios), ctypes.c_int(ncand),
                                         ctypes.c_int(nt), gm, ph, pd, ctypes.c_uint(FLAG_OVP if ovp else 0),
                                         ctypes.c_int(dt), _vp(sse), _vp(ws),
                                         _stream(x.device))
    if rc == -2:              # ANTQ_ERR_UNSUPPORTED
        return None
    _check(rc, "antq_search_sse_multi")
    return sse


XMAX_GIVEN, XMAX_ABSMAX, XMAX_3SIGMA = 0, 1, 2


def calibrate(x, rows, row_len, per_row, plans, gmaxs, lb, ub, step, xmax="absmax", ovp=False):
    """One quantiser's whole first-call calibration in one C call (antq_calibrate): the clip statistic, every candidate
    codebook's clip search, the per-row choice and the type choice -- no device->host sync.
    xmax: "absmax" (ANT), "3sigma" (OliVe) or a float32 device tensor with rows (per_row) / 1 entries.
    Returns (alpha [ntypes, na] float32, score [ntypes] float32, type int32[1], xmax [na] float32), all on x's device."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    if rows * row_len != x.numel():
        raise AntqError("rows*row_len != numel")
    nt = len(plans)
    if nt < 1 or len(gmaxs) != nt:
        raise AntqError("one gmax per plan, at least one plan")
    na = rows if per_row else 1
    if isinstance(xmax, str):
        mode = {"absmax": XMAX_ABSMAX, "3sigma": XMAX_3SIGMA}.get(xmax)
        if mode is None:
            raise AntqError("xmax must be 'absmax', '3sigma' or a tensor")
        xm = torch.empty(na, dtype=torch.float32, device=x.device)
    else:
        mode = XMAX_GIVEN
        _require_gpu(xmax, "xmax")
        xm = xmax.reshape(-1)
        if xm.dtype != torch.float32 or xm.numel() != na or xm.device != x.device:
            raise AntqError("xmax must hold %d float32 values on x's device" % na)
    nbytes = lib().antq_calibrate_workspace_bytes(ctypes.c_size_t(rows), ctypes.c_int(1 if per_row else 0), ctypes.c_int(lb),
                                                  ctypes.c_int(ub), ctypes.c_int(step), ctypes.c_int(nt))
    if nbytes == 0:
        raise AntqError("antq_calibrate: bad candidate range / type count")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    alpha = torch.empty(nt, na, dtype=torch.float32, device=x.device)
    score = torch.empty(nt, dtype=torch.float32, device=x.device)
    typ = torch.empty(1, dtype=torch.int32, device=x.device)
    ph = (ctypes.c_void_p * nt)(*[p.host_addr for p in plans])
    pd = (ctypes.c_void_p * nt)(*[p.dev(x.device).data_ptr() for p in plans])
    gm = (ctypes.c_float * nt)(*[float(g) for g in gmaxs])
    with _on_device(x.device):
        rc = lib().antq_calibrate(_vp(x), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), ctypes.c_int(1 if per_row else 0),
                                  ctypes.c_int(dt), ctypes.c_int(mode), _vp(xm), ctypes.c_int(lb), ctypes.c_int(ub),
                                  ctypes.c_int(step), ctypes.c_int(nt), gm, ph, pd, ctypes.c_uint(FLAG_OVP if ovp else 0),
                                  _vp(alpha), _vp(score), _vp(typ), _vp(ws), ctypes.c_size_t(nbytes), _stream(x.device))
    _check(rc, "antq_calibrate")
    return alpha, score, typ, xm


_install_args = {}        # (plans..., gmaxs...) -> the three small ctypes arrays of antq_calibrate_install (built once per type list)


def calibrate_install(x, out, plans, gmaxs, typ, alpha, score, grids_stack, grid_out, alpha_out, mse_out, ovp=False):
    """antq_calibrate_install: from the device-side pick `typ` of calibrate() -- alpha_out / mse_out / grid_out <- the winner's
    alpha, score and row of grids_stack [ntypes, grid_len] float32, and out <- the steady-state forward of x (one scale) with
    the winner's codebook.  Two launches, no read-back.  False: this tensor is not eligible (ragged / unaligned: the caller
    quantises with every candidate and gathers)."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    nt = len(plans)
    if dt is None or dt == F64 or nt > 4 or not x.is_contiguous() or grids_stack.dtype != torch.float32 or grids_stack.shape[0] != nt:
        return False
    key = (tuple(id(p) for p in plans), tuple(gmaxs), x.device.index)
    a = _install_args.get(key)
    if a is None:
        if len(_install_args) > 256:
            _install_args.clear()
        a = _install_args[key] = ((ctypes.c_float * nt)(*[float(g) for g in gmaxs]), (ctypes.c_void_p * nt)(*[p.host_addr for p in plans]),
                                  (ctypes.c_void_p * nt)(*[p.dev(x.device).data_ptr() for p in plans]), list(plans))
    with _on_device(x.device):
        rc = lib().antq_calibrate_install(x.data_ptr(), out.data_ptr(), x.numel(), dt, nt, ctypes.addressof(a[0]), ctypes.addressof(a[1]),
                                          ctypes.addressof(a[2]), FLAG_OVP if ovp else 0, typ.data_ptr(), alpha.data_ptr(), score.data_ptr(),
                                          grids_stack.data_ptr(), grids_stack.shape[1], grid_out.data_ptr(), None, 0, None,
                                          alpha_out.data_ptr(), mse_out.data_ptr(), _stream_int(x.device))
    if rc == -2:                    # ANTQ_ERR_UNSUPPORTED
        return False
    if rc:
        _check(rc, "antq_calibrate_install")
    return True


class _CalibJob(ctypes.Structure):           # include/antq.h: antq_calib_job
    _fields_ = [("x_dev", ctypes.c_void_p), ("rows", ctypes.c_size_t), ("row_len", ctypes.c_size_t),
                ("alpha_per_row", ctypes.c_int), ("xmax_mode", ctypes.c_int), ("xmax_dev", ctypes.c_void_p),
                ("lb", ctypes.c_int), ("ub", ctypes.c_int), ("step", ctypes.c_int), ("ntypes", ctypes.c_int),
                ("gmax_host", ctypes.POINTER(ctypes.c_float)), ("plan_host", ctypes.POINTER(ctypes.c_void_p)),
                ("plan_dev", ctypes.POINTER(ctypes.c_void_p)), ("alpha_dev", ctypes.c_void_p),
                ("score_dev", ctypes.c_void_p), ("type_dev", ctypes.c_void_p)]


def calibrate_batch(jobs, xmax="absmax", ovp=False):
    """antq_calibrate_batch: the calibrations of many quantisers (one model's weights) in one C call, stream-ordered, no
    device->host sync.  jobs: sequence of (x, rows, row_len, per_row, plans, gmaxs, lb, ub, step); one dtype and one device
    for the batch.  Returns (results, types): results[i] = (alpha [ntypes_i, na_i], score [ntypes_i], xmax [na_i]) -- views
    of two flat float32 buffers -- and types: ONE int32 tensor with the n type picks (a single read-back for the model)."""
    jobs = list(jobs)
    if not jobs:
        return [], None
    x0 = jobs[0][0]
    _require_gpu(x0, "x")
    dt = _DTYPES.get(x0.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x0.dtype)
    mode = {"absmax": XMAX_ABSMAX, "3sigma": XMAX_3SIGMA}.get(xmax)
    if mode is None:
        raise AntqError("xmax must be 'absmax' or '3sigma'")
    dev = x0.device
    n = len(jobs)
    arr = (_CalibJob * n)()
    keep, sizes = [], []
    total = 0
    for x, rows, row_len, per_row, plans, gmaxs, lb, ub, step in jobs:
        if x.device != dev or x.dtype != x0.dtype or not x.is_contiguous():
            raise AntqError("calibrate_batch: one device, one dtype, contiguous tensors")
        if rows * row_len != x.numel() or len(plans) < 1 or len(plans) != len(gmaxs):
            raise AntqError("calibrate_batch: rows*row_len != numel, or plans / gmaxs mismatch")
        na, nt = (rows if per_row else 1), len(plans)
        sizes.append((na, nt, total))
        total += nt * na + nt + na
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    types = torch.empty(n, dtype=torch.int32, device=dev)
    results = []
    for i, ((x, rows, row_len, per_row, plans, gmaxs, lb, ub, step), (na, nt, off)) in enumerate(zip(jobs, sizes)):
        alpha, score, xm = flat[off:off + nt * na].view(nt, na), flat[off + nt * na:off + nt * na + nt], flat[off + nt * na + nt:off + nt * na + nt + na]
        ph = (ctypes.c_void_p * nt)(*[p.host_addr for p in plans])
        pd = (ctypes.c_void_p * nt)(*[p.dev(dev).data_ptr() for p in plans])
        gm = (ctypes.c_float * nt)(*[float(g) for g in gmaxs])
        keep.append((ph, pd, gm))
        arr[i] = _CalibJob(x.data_ptr(), rows, row_len, 1 if per_row else 0, mode, xm.data_ptr(), int(lb), int(ub), int(step), nt,
                           gm, ph, pd, alpha.data_ptr(), score.data_ptr(), types.data_ptr() + 4 * i)
        results.append((alpha, score, xm))
    nbytes = lib().antq_calibrate_batch_workspace_bytes(arr, ctypes.c_int(n))
    if nbytes == 0:
        raise AntqError("antq_calibrate_batch: bad candidate range / type count")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _on_device(dev):
        rc = lib().antq_calibrate_batch(arr, ctypes.c_int(n), ctypes.c_int(dt), ctypes.c_uint(FLAG_OVP if ovp else 0), _vp(ws),
                                        ctypes.c_size_t(nbytes), _stream(dev))
    _check(rc, "antq_calibrate_batch")
    return results, types


def search_pick(sse, xmax, ratios, row_len):
    """(best_score, best_alpha) per row from the candidates' sums of squared errors (strict '<', first best)."""
    ncand, na = sse.shape
    best = torch.empty(na, dtype=torch.float32, device=sse.device)
    alpha = torch.empty(na, dtype=torch.float32, device=sse.device)
    with _on_device(sse.device):
        _check(lib().antq_search_pick(_vp(sse), _vp(xmax), _vp(ratios), ctypes.c_int(ncand), ctypes.c_size_t(na),
                                      ctypes.c_size_t(row_len), _vp(best), _vp(alpha), _stream(sse.device)),
               "antq_search_pick")
    return best, alpha


def affine(x, k, xmin, xmax, rows, row_len, per_row, want_q=False):
    _require_gpu(x, "x")
    if x.dtype != torch.float32:
        raise AntqError("antq_affine is fp32 only")
    out = torch.empty_like(x)
    q = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_q else None
    with torch.cuda.device(x.device):
        _check(lib().antq_affine(_vp(x), _vp(out), _vp(q), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
                                 ctypes.c_int(k), _vp(xmin), _vp(xmax), ctypes.c_int(1 if per_row else 0),
                                 _stream(x.device)), "antq_affine")
    return (out, q) if want_q else out


def copy(src, dst):
    _require_gpu(src, "src")
    _require_gpu(dst, "dst")
    nbytes = src.numel() * src.element_size()
    with torch.cuda.device(src.device):
        _check(lib().antq_copy(_vp(src), _vp(dst), ctypes.c_size_t(nbytes), _stream(src.device)), "antq_copy")
    return dst


def encode4(x, alpha, plan, gmax, rows, row_len, per_row, n_normal=0, ovp=False):
    """Packed 4-bit codes (two per byte) of the quantised tensor: uint8 tensor of numel/2 bytes."""
    _require_gpu(x, "x")
    dt = _DTYPES.get(x.dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % x.dtype)
    codes = torch.empty(x.numel() // 2, dtype=torch.uint8, device=x.device)
    pd = plan.dev(x.device)
    with _on_device(x.device):
        rc = lib().antq_encode4(_vp(x), _vp(codes), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _vp(alpha),
                                ctypes.c_int(1 if per_row else 0), ctypes.c_float(gmax), plan.host_ptr(), _vp(pd),
                                ctypes.c_int(n_normal), ctypes.c_uint(FLAG_OVP if ovp else 0), ctypes.c_int(dt),
                                _stream(x.device))
    _check(rc, "antq_encode4")
    return codes


def decode4(codes, alpha, plan, gmax, rows, row_len, per_row, dtype, n_normal=0, ovp=False):
    _require_gpu(codes, "codes")
    dt = _DTYPES.get(dtype)
    if dt is None or dt == F64:
        raise AntqError("unsupported dtype %s" % dtype)
    out = torch.empty(rows * row_len, dtype=dtype, device=codes.device)
    pd = plan.dev(codes.device)
    with _on_device(codes.device):
        rc = lib().antq_decode4(_vp(codes), _vp(out), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _vp(alpha),
                                ctypes.c_int(1 if per_row else 0), ctypes.c_float(gmax), plan.host_ptr(), _vp(pd),
                                ctypes.c_int(n_normal), ctypes.c_uint(FLAG_OVP if ovp else 0), ctypes.c_int(dt),
                                _stream(codes.device))
    _check(rc, "antq_decode4")
    return out.view(rows, row_len)


# ---------------------------------------------------------------------------------
# batched launch: many tensors, one kernel
# ---------------------------------------------------------------------------------
class _Job(ctypes.Structure):
    _fields_ = [("x_dev", ctypes.c_void_p), ("out_dev", ctypes.c_void_p), ("alpha_dev", ctypes.c_void_p),
                ("rows", ctypes.c_size_t), ("row_len", ctypes.c_size_t), ("alpha_per_row", ctypes.c_int),
                ("gmax", ctypes.c_float), ("plan_host", ctypes.c_void_p), ("plan_dev", ctypes.c_void_p)]


class Batch:
    """Fake-quant of many (static-alpha) tensors in ONE launch (antq_fakequant_batch).

    jobs: iterable of tuples (x, out, alpha, plan, gmax, rows, row_len, per_row); all tensors on one device.
    Ragged rows (conv1's K = 147) and unaligned buffers ride in the same launch (element-granular blocks); only
    a job whose dtype differs from the first one's is kept aside and launched on its own by run().  The
    descriptor table is built once and stays resident: weights and calibrated alphas do not move between
    forwards."""

    def __init__(self, jobs, ovp=False, dynamic=False):
        """dynamic=True: every job's alpha tensor is an OUTPUT (group / row abs-max computed in the kernel), or None when the
        caller has no use for the scales (the kernel then skips the store); per_row, groups of a power of two of 16-byte
        vectors or rows of 128 .. 8192 vectors -- otherwise AntqError."""
        jobs = [tuple(j) for j in jobs]
        if not jobs:
            raise AntqError("empty batch")
        x0 = jobs[0][0]
        _require_gpu(x0, "x")
        self.device, self.dtype, self.ovp = x0.device, _DTYPES[x0.dtype], ovp
        self._keep = jobs                       # keeps tensors and plans alive
        self.singles, batched = [], []
        for j in jobs:
            x, out, alpha, plan, gmax, rows, row_len, per_row = j
            _require_gpu(x, "x")
            if x.device != x0.device:
                raise AntqError("every tensor of a batch must live on one device")
            _require_out(out, x)
            if rows * row_len != x.numel():
                raise AntqError("rows*row_len != numel")
            if x.dtype != x0.dtype:
                self.singles.append(j)
            else:
                batched.append(j)
        self.n_batched = len(batched)
        self.host = self.dev = None
        if batched:
            arr = (_Job * len(batched))()
            for k, (x, out, alpha, plan, gmax, rows, row_len, per_row) in enumerate(batched):
                arr[k] = _Job(x.data_ptr(), out.data_ptr(), alpha.data_ptr() if alpha is not None else 0, rows, row_len,
                              1 if per_row else 0, gmax,
                              plan.host_addr, plan.dev(self.device).data_ptr())
            cap = lib().antq_batch_capacity(arr, len(batched), self.dtype)
            host = np.zeros(cap, dtype=np.uint8)
            n = lib().antq_batch_build(arr, len(batched), self.dtype,
                                       (FLAG_OVP if ovp else 0) | (FLAG_DYNAMIC if dynamic else 0),
                                       host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
            if n <= 0:
                _check(n, "antq_batch_build")
            self.host = host[:n].copy()
            self.dev = torch.from_numpy(self.host).to(self.device)

    def run(self):
        e = _ext_mod if _ext_mod is not False else ext()
        if self.host is not None and e is not None:
            e.batch_run(self.host.ctypes.data, self.dev)
        elif self.host is not None:
            with _on_device(self.device):
                rc = lib().antq_fakequant_batch(ctypes.c_void_p(self.host.ctypes.data), ctypes.c_void_p(self.dev.data_ptr()),
                                                ctypes.c_void_p(_stream_int(self.device)))
            if rc:
                _check(rc, "antq_fakequant_batch")
        for x, out, alpha, plan, gmax, rows, row_len, per_row in self.singles:
            fakequant(x, alpha, plan, gmax, rows, row_len, per_row, ovp=self.ovp, out=out)

    def kernels(self):
        """The launches run() issues, read off the descriptor header the library built (csrc/antq_k_batch.h BatchHeader,
        the dispatch of antq_fakequant_batch in csrc/antq_batch.hip): [(kernel name, workgroups)], + one entry per single.
        For reports (bench.py config.configs[].kernel) -- nothing on the launch path reads it."""
        out = []
        if self.host is not None:
            h = self.host[:80].view(np.uint32)
            fam, pad = [int(v) for v in h[8:17]], int(h[17])
            t = {F32: "float", BF16: "bf16", F16: "f16"}[self.dtype]
            o = "true" if self.ovp else "false"
            dyn = "true" if int(h[3]) & FLAG_DYNAMIC else "false"
            if pad & 1:
                out.append(("antq::k_fq_batch_all<%s,%s>" % (t, o), fam[0]))
            elif fam[0]:
                w = 1 if ((pad >> 8) & 7) == 1 else 4
                out.append(("antq::k_fq_batch<%s,%s,%d>" % (t, o, w), fam[0] * (4 // w)))
            if fam[5]:
                out.append(("antq::k_fq_hbatch<%s,%s>" % (t, o), fam[5] * 4))
            for f, v in ((6, 1), (7, 4), (8, 16)):
                if fam[f] and not (pad & 1):
                    out.append(("antq::k_fq_hbatch_dyn<%s,%s,%d>" % (t, o, v), fam[f]))
            if not (pad & 1):
                if fam[1]:
                    out.append(("antq::k_fq_batch_d<%s,%s,true,%s>" % (t, o, dyn), fam[1]))
                if fam[2]:
                    out.append(("antq::k_fq_batch_d<%s,%s,false,%s>" % (t, o, dyn), fam[2]))
                if fam[3]:
                    out.append(("antq::k_fq_batch_dyn<%s,%s>" % (t, o), fam[3]))
                if fam[4]:
                    out.append(("antq::k_fq_batch_dyn16<%s,%s>" % (t, o), fam[4]))
        out += [("antq_fakequant (own launch)", 0)] * len(self.singles)
        return out
