"""Drop-in for the reference's compiled extension module `quant_cuda`
(ant_quantization/quant/quant.cpp:27-29: `m.def("quant", ...)`).

    quant(x, grid) -> (z, idx)

x: 1-D contiguous float32 / float64 HIP tensor, grid: <= 1024 values already cast to x's
dtype (QuantBase._quantization does `grid.type_as(x)`).  z = nearest grid value of every
element under the reference scan's rule (last minimum wins, 0 beyond 102400); idx is the
reference's second output: a freshly allocated all-zero tensor shaped like x that its
kernel never writes (quant_kernel.cu:18,49) and its caller discards.

Put this package directory on sys.path (or `sys.modules['quant_cuda'] = this module`) and
the reference's unmodified quant_modules.py runs on MI355X.  Launches go to the CURRENT
torch stream of x's device (the reference used the legacy default stream).
"""
import torch

from . import _lib

# The reference calls quant(x, self.quant_grid) with the SAME grid buffer on every forward.  The first call with a
# given buffer state reads the grid back once and builds its plan; later calls go through the table kernel
# (antq_nearest_plan) -- about twice as fast as scanning.  Keyed by (data_ptr, _version, numel, dtype, device); the
# entry keeps the grid tensor alive, so its address cannot be recycled for another grid while the entry exists.
# (An edit through `.data` does not bump `_version`: PyTorch's own staleness rule applies.)
_plans = {}


def _plan_of(grid):
    key = (grid.data_ptr(), grid._version, grid.numel(), grid.dtype, grid.device)
    hit = _plans.get(key)
    if hit is None:
        if len(_plans) > 256:
            _plans.clear()
        hit = (_lib.plan_for(grid.detach().float().cpu().numpy()), grid)
        _plans[key] = hit
    return hit[0]


def quant(x, y):
    if x.dim() != 1:
        raise RuntimeError("quant_cuda.quant: x must be 1-D (got %d-D)" % x.dim())
    x = x.contiguous()
    if x.dtype == torch.float32 and y.numel() <= _lib.MAX_GRID:
        z = _lib.nearest_plan(x, _plan_of(y))
    else:
        z = _lib.nearest(x, y.contiguous())
    return z, torch.zeros_like(x)
