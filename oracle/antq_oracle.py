"""ctypes/numpy front-end of the CPU oracle (oracle/antq_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (ant_quantization_amd/) never
imports this module.  Parity pinning: see the header of antq_oracle.c -- the
reference holds no tests/golden vectors for this path, the pins are
tests/golden/*.npz produced by tests/golden/make_golden.py from the
reference's own Python.

Also holds numpy restatements of the reference's codebook generators
(AQ/quant_modules.py:75-278, OQ/quant_modules.py:72-179), each checked against
the golden dumps in tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libantq_oracle.so")

IDX_NONE = -1
IDX_VICTIM = -2


def build(force=False):
    """Compile libantq_oracle.so with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "antq_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "libantq_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.antq_oracle_search_mse_f32.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------
# a1/a2  quant_cuda.quant
# --------------------------------------------------------------------------
def nearest(x, grid):
    """(z, idx) = literal KQ/quant_kernel.cu:25-37 on a flat array.

    x float32 or float64 (the two dtypes AT_DISPATCH_FLOATING_TYPES accepts).
    idx is the last-minimum scan index (int32; -1 when no entry is within
    102400, z is then 0.0)."""
    x = np.ascontiguousarray(x)
    n = x.size
    idx = np.empty(n, dtype=np.int32)
    if x.dtype == np.float64:
        g = np.ascontiguousarray(grid, dtype=np.float64)
        z = np.empty(n, dtype=np.float64)
        lib().antq_oracle_nearest_f64(_p(x), _p(z), _p(idx), ctypes.c_size_t(n), _p(g), ctypes.c_int(g.size))
    else:
        x = _f32(x)
        g = _f32(grid)
        z = np.empty(n, dtype=np.float32)
        lib().antq_oracle_nearest_f32(_p(x), _p(z), _p(idx), ctypes.c_size_t(n), _p(g), ctypes.c_int(g.size))
    return z.reshape(x.shape), idx.reshape(x.shape)


def bf16_to_f32(u16):
    u16 = np.ascontiguousarray(u16, dtype=np.uint16)
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(f):
    f = _f32(f)
    out = np.empty(f.shape, dtype=np.uint16)
    lib().antq_oracle_f32_to_bf16(_p(f), _p(out), ctypes.c_size_t(f.size))
    return out


# --------------------------------------------------------------------------
# a4/a5  Quantizer._forward
# --------------------------------------------------------------------------
def forward(x2d, alpha, grid, gmax=None, ovp=False, want_idx=True):
    """Fake-quant of a [rows, row_len] array.

    x2d  float32, or uint16 holding bf16 bits (bf16 I/O extension).
    alpha: array of `rows` entries (per-channel) or a scalar / 1 entry.
    grid : array passed to the kernel (OliVe: cat(normal, outliers)).
    gmax : max of the NORMAL grid (defaults to max(grid); OliVe must pass it).
    Returns (out, idx) with out in x2d's dtype."""
    x2d = np.ascontiguousarray(x2d)
    assert x2d.ndim == 2
    rows, row_len = x2d.shape
    alpha = _f32(np.atleast_1d(alpha)).reshape(-1)
    if alpha.size not in (1, rows):
        raise ValueError("alpha must have 1 or rows entries")
    per_row = 1 if (alpha.size == rows and rows > 1) else 0
    g = _f32(grid)
    if gmax is None:
        gmax = float(np.max(g))
    idx = np.empty((rows, row_len), dtype=np.int32) if want_idx else None
    if x2d.dtype == np.uint16:
        out = np.empty_like(x2d)
        fn = lib().antq_oracle_forward_bf16
    else:
        x2d = _f32(x2d)
        out = np.empty_like(x2d)
        fn = lib().antq_oracle_forward_f32
    fn(_p(x2d), _p(out), _p(idx), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
       _p(alpha), ctypes.c_int(per_row), _p(g), ctypes.c_int(g.size),
       ctypes.c_float(gmax), ctypes.c_int(1 if ovp else 0))
    return out, idx


def absmax(x2d, per_row=True, ratio=1.0):
    x2d = _f32(x2d)
    rows, row_len = x2d.shape
    alpha = np.empty(rows if per_row else 1, dtype=np.float32)
    lib().antq_oracle_absmax_f32(_p(x2d), _p(alpha), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
                                 ctypes.c_int(1 if per_row else 0), ctypes.c_float(ratio))
    return alpha


def three_sigma(x2d, per_row=True):
    """OliVe's clip statistic (OQ/quant_modules.py:193-197 per channel, :213-218 per tensor):
    x_max = max(|mean + 3 std|, |mean - 3 std|), std unbiased (torch.std default).  x2d float32, or uint16 holding bf16
    bits: the reference then computes in the tensor's own dtype -- mean, std, 3 * std, the sum and the difference are each
    rounded to bf16 (torch's element-wise kernels compute in fp32 and round the result).  Sums in float64 here; torch's
    fp32 reductions agree to their summation-order noise."""
    bf16 = x2d.dtype == np.uint16
    xf = (bf16_to_f32(x2d) if bf16 else np.asarray(x2d, dtype=np.float32)).astype(np.float64)
    if not per_row:
        xf = xf.reshape(1, -1)
    rnd = (lambda v: bf16_to_f32(f32_to_bf16(np.asarray(v, dtype=np.float32)))) if bf16 else (lambda v: np.asarray(v, dtype=np.float32))
    with np.errstate(all="ignore"):
        mean = rnd(xf.mean(axis=1).astype(np.float32))
        std = rnd(xf.std(axis=1, ddof=1).astype(np.float32))
        t3 = rnd(np.float32(3.0) * std)
        a, b = np.abs(rnd(mean + t3)), np.abs(rnd(mean - t3))
    return np.maximum(a, b).astype(np.float32)


def mse(q2d, x2d, per_row=True):
    q2d = _f32(q2d)
    x2d = _f32(x2d)
    rows, row_len = x2d.shape
    out = np.empty(rows if per_row else 1, dtype=np.float32)
    lib().antq_oracle_mse_f32(_p(q2d), _p(x2d), _p(out), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
                              ctypes.c_int(1 if per_row else 0))
    return out


def search_mse(x2d, x_max, lb, ub, step, grid, gmax=None, ovp=False, per_row=True):
    """a10.  Returns (best_score, best_alpha, trace[ncand, na])."""
    x2d = _f32(x2d)
    rows, row_len = x2d.shape
    na = rows if per_row else 1
    x_max = _f32(np.atleast_1d(x_max)).reshape(-1)
    assert x_max.size == na
    g = _f32(grid)
    if gmax is None:
        gmax = float(np.max(g))
    ncand = len(range(lb, ub, step))
    best_score = np.empty(na, dtype=np.float32)
    best_alpha = np.empty(na, dtype=np.float32)
    trace = np.empty((ncand, na), dtype=np.float32)
    got = lib().antq_oracle_search_mse_f32(
        _p(x2d), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), ctypes.c_int(1 if per_row else 0),
        _p(x_max), ctypes.c_int(lb), ctypes.c_int(ub), ctypes.c_int(step),
        _p(g), ctypes.c_int(g.size), ctypes.c_float(gmax), ctypes.c_int(1 if ovp else 0),
        _p(best_score), _p(best_alpha), _p(trace))
    assert got == ncand
    return best_score, best_alpha, trace


def affine(x2d, k, x_min, x_max):
    """a14 AsymmetricQuantFunction.forward; x_min/x_max scalar or per-row."""
    x2d = _f32(x2d)
    rows, row_len = x2d.shape
    x_min = _f32(np.atleast_1d(x_min)).reshape(-1)
    x_max = _f32(np.atleast_1d(x_max)).reshape(-1)
    per_row = 1 if (x_min.size == rows and rows > 1) else 0
    out = np.empty_like(x2d)
    q = np.empty(x2d.shape, dtype=np.int32)
    lib().antq_oracle_affine_f32(_p(x2d), _p(out), _p(q), ctypes.c_size_t(rows), ctypes.c_size_t(row_len),
                                 ctypes.c_int(k), _p(x_min), _p(x_max), ctypes.c_int(per_row))
    return out, q


def forward_rows(x2d, out, row_begin, row_end, alpha, grid, gmax):
    """CPU-baseline leg (bench.py): rows [row_begin,row_end) of a4, releases the GIL."""
    rows, row_len = x2d.shape
    per_row = 1 if alpha.size == rows and rows > 1 else 0
    fn = lib().antq_oracle_forward_rows_bf16 if x2d.dtype == np.uint16 else lib().antq_oracle_forward_rows_f32
    fn(_p(x2d), _p(out), ctypes.c_size_t(row_begin), ctypes.c_size_t(row_end), ctypes.c_size_t(row_len),
       _p(alpha), ctypes.c_int(per_row), _p(grid), ctypes.c_int(grid.size), ctypes.c_float(gmax))


def forward_omp(x2d, out, alpha, grid, gmax, reps=1, threads=0):
    """CPU-baseline leg (bench.py): `reps` sweeps of a4 over all rows of a bf16 tensor on `threads` OpenMP threads
    (0: all).  Returns the number of threads used."""
    rows, row_len = x2d.shape
    assert x2d.dtype == np.uint16 and out.dtype == np.uint16
    per_row = 1 if alpha.size == rows and rows > 1 else 0
    fn = lib().antq_oracle_forward_omp_bf16
    fn.restype = ctypes.c_int
    return int(fn(_p(x2d), _p(out), ctypes.c_size_t(rows), ctypes.c_size_t(row_len), _p(alpha), ctypes.c_int(per_row),
                  _p(grid), ctypes.c_int(grid.size), ctypes.c_float(gmax), ctypes.c_int(reps), ctypes.c_int(threads)))


# --------------------------------------------------------------------------
# a7/a8 codebook generators (numpy restatement; float32 like torch.tensor())
# --------------------------------------------------------------------------
def _ant_convert(values, bit):
    """AQ/quant_modules.py:75-83 convert_tensor.

    pad one 0. if short, assert length, fp32, sort, then
    values.mul(10.0 / torch.max(values)): python-float / tensor is
    Tensor.__rtruediv__ = reciprocal(t) * 10.0, evaluated in fp32."""
    values = list(values)
    if 2 ** bit > len(values):
        values.append(0.)
    if 2 ** bit != len(values):
        raise AssertionError("codebook has %d entries, expected %d" % (len(values), 2 ** bit))
    v = np.sort(np.asarray(values, dtype=np.float32), kind="stable")
    c = np.float32(np.float32(1.0) / np.max(v)) * np.float32(10.0)
    return (v * np.float32(c)).astype(np.float32)


def ant_int_value(bit, signed):
    """AQ:204-221."""
    B = bit - 1 if signed else bit
    values = [0.]
    for i in range(1, 2 ** B):
        values.append(i)
        if signed:
            values.append(-i)
    if signed:
        values.append(-2 ** B)
    return _ant_convert(values, bit)


def ant_pot_value(bit, signed):
    """AQ:189-201."""
    B = bit - 1 if signed else bit
    values = [0.]
    for i in range(0, 2 ** B - 1):
        values.append(2 ** i)
        if signed:
            values.append(-2 ** i)
    return _ant_convert(values, bit)


def ant_float_value(bit, signed, eb=3):
    """AQ:157-187 (the live definition; :133-154 is shadowed)."""
    B = bit - 1 if signed else bit
    exp_bit = eb
    man_bit = B - exp_bit
    if B == 2:
        exp_bit = 2
        man_bit = 0
    values = []
    min_to_zero = True
    subnormal = True
    for i in range(2 ** exp_bit):
        for j in range(int(2 ** man_bit)):
            if min_to_zero:
                values.append(0.)
                min_to_zero = False
            else:
                if subnormal:
                    values.append((2 ** i) * (j * 2 ** (-man_bit)))
                else:
                    values.append((2 ** (i - 1)) * (1 + j * 2 ** (-man_bit)))
                if signed:
                    if subnormal:
                        values.append(-(2 ** i) * (j * 2 ** (-man_bit)))
                    else:
                        values.append(-(2 ** (i - 1)) * (1 + j * 2 ** (-man_bit)))
        subnormal = False
    return _ant_convert(values, bit)


def _flint_list(B, signed, exp_base=0):
    """Shared list builder of AQ:223-276 / OQ:94-146."""
    value_bit = B
    assert value_bit >= 2
    neg_exp_num = value_bit - 1
    pos_exp_num = value_bit - 1
    exp_max = pos_exp_num + exp_base
    values = [0.]
    for i in range(0, neg_exp_num + 1):
        exp_bit = i + 2
        exp_value = -(exp_bit - 1)
        mant_bit = value_bit - exp_bit
        for j in range(int(2 ** mant_bit)):
            v = 2 ** (exp_value + exp_base) * (1 + 2 ** (-mant_bit) * j)
            values.append(v)
            if signed:
                values.append(-v)
    exp_bit = 2
    mant_bit = value_bit - exp_bit
    for j in range(int(2 ** mant_bit)):
        v = 2 ** (0 + exp_base) * (1 + 2 ** (-mant_bit) * j)
        values.append(v)
        if signed:
            values.append(-v)
    for i in range(1, pos_exp_num):
        exp_bit = i + 2
        mant_bit = value_bit - exp_bit
        for j in range(int(2 ** mant_bit)):
            v = 2 ** (i + exp_base) * (1 + 2 ** (-mant_bit) * j)
            values.append(v)
            if signed:
                values.append(-v)
    values.append(2 ** exp_max)
    if signed:
        values.append(-2 ** exp_max)
    return values, exp_max


def ant_flint_value(bit, signed):
    """AQ:223-278."""
    B = bit - 1 if signed else bit
    values, _ = _flint_list(B, signed)
    return _ant_convert(values, bit)


def ant_apot_value(bit, signed):
    """AQ:85-131."""
    B = bit - 1 if signed else bit
    base_a, base_b, base_c = [0.], [0.], [0.]
    if B == 2:
        for i in range(3):
            base_a.append(2 ** (-i - 1))
    elif B == 4:
        for i in range(3):
            base_a.append(2 ** (-2 * i - 1))
            base_b.append(2 ** (-2 * i - 2))
    elif B == 6:
        for i in range(3):
            base_a.append(2 ** (-3 * i - 1))
            base_b.append(2 ** (-3 * i - 2))
            base_c.append(2 ** (-3 * i - 3))
    elif B == 3:
        for i in range(3):
            if i < 2:
                base_a.append(2 ** (-i - 1))
            else:
                base_b.append(2 ** (-i - 1))
                base_a.append(2 ** (-i - 2))
    elif B == 5:
        for i in range(3):
            if i < 2:
                base_a.append(2 ** (-2 * i - 1))
                base_b.append(2 ** (-2 * i - 2))
            else:
                base_c.append(2 ** (-2 * i - 1))
                base_a.append(2 ** (-2 * i - 2))
                base_b.append(2 ** (-2 * i - 3))
    values = []
    for a in base_a:
        for b in base_b:
            for c in base_c:
                values.append(a + b + c)
                if signed:
                    values.append(-(a + b + c))
    return _ant_convert(values, bit)


def ant_grid(mode, bit, signed):
    """Grid the reference installs for a resolved mode (AQ:488-511)."""
    if mode == "int":
        return ant_int_value(bit, signed)
    if mode == "flint":
        return ant_flint_value(bit, signed)
    if mode == "pot":
        return ant_pot_value(bit, signed)
    if mode == "apot":
        return ant_apot_value(bit, signed)
    if mode == "float":
        return ant_float_value(bit, signed, 3)
    if mode in ("float1", "float2", "float3", "float4"):
        return ant_float_value(bit, signed, int(mode[-1]))
    raise RuntimeError("Unsupported mode: " + mode)


def olive_int_value(bit, signed):
    """OQ:72-91: sorted ints, then `values *= 32 / 2**B` (python float scalar)."""
    B = bit - 1 if signed else bit
    values = [0.]
    for i in range(1, 2 ** B):
        values.append(i)
        if signed:
            values.append(-i)
    v = np.sort(np.asarray(values, dtype=np.float32), kind="stable")
    return (v * np.float32(32 / (2 ** B))).astype(np.float32)


def olive_flint_value(bit, signed):
    """OQ:93-153: flint list, sorted, `values *= 32 / 2**exp_max`."""
    B = bit - 1 if signed else bit
    values, exp_max = _flint_list(B, signed)
    v = np.sort(np.asarray(values, dtype=np.float32), kind="stable")
    return (v * np.float32(32 / (2 ** exp_max))).astype(np.float32)


def olive_outlier_value(bit, signed, exp_bit=2, exp_base=5):
    """OQ:155-179 abfloat outlier codebook."""
    B = bit - 1 if signed else bit
    mant_bit = B - exp_bit
    values = []
    for i in range(exp_base, exp_base + 2 ** exp_bit):
        for j in range(int(2 ** mant_bit)):
            if i == exp_base and j == 0:
                continue
            v = 2 ** i * (1 + 2 ** (-mant_bit) * j)
            values.append(v)
            if signed:
                values.append(-v)
    return np.sort(np.asarray(values, dtype=np.float32), kind="stable")


def olive_grid(mode, bit, signed):
    if mode == "int":
        return olive_int_value(bit, signed)
    if mode == "flint":
        return olive_flint_value(bit, signed)
    raise RuntimeError("Unsupported mode: " + mode)
