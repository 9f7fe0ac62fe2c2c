"""Value grids (codebooks) of the ANT and OliVe quantisers, built on the host.

What the reference builds with nested Python loops
(ant_quantization/antquant/quant_modules.py:75-278,
olive_quantization/antquant/quant_modules.py:72-179) is produced here from the closed form
of each number format: a set of magnitudes per binade, mirrored for signed grids.  The
values must be bit-identical to the reference's fp32 tensors, so the final normalisation
reproduces its arithmetic exactly:

  ANT   `values.mul(10.0 / torch.max(values))` -- python-float / tensor dispatches to
        Tensor.__rtruediv__ = reciprocal(max) * 10.0 in fp32, i.e. c = fl32(fl32(1/max)*10),
        then fl32(v * c) per entry (convert_tensor, AQ:75-83);
  OliVe `values *= 32 / 2**k`  -- an exact power-of-two (or 32/2**B) python scalar.

The host keeps the grid (numpy float32): it is needed to build the kernel's decision-table
plan without any device->host copy.
"""
import functools

import numpy as np

_F32 = np.float32


def _mirror(mags, signed, zeros_signed=2, zeros_unsigned=1):
    """sorted grid from positive magnitudes: negatives, zero(s), positives."""
    mags = np.sort(np.asarray(mags, dtype=np.float64))
    if signed:
        z = np.zeros(zeros_signed, dtype=np.float64)
        return np.concatenate([-mags[::-1], z, mags]).astype(_F32)
    z = np.zeros(zeros_unsigned, dtype=np.float64)
    return np.concatenate([z, mags]).astype(_F32)


def _ant_normalise(values, bit):
    """pad / length check / scale of convert_tensor (AQ:75-83)."""
    if values.size != 2 ** bit:
        raise AssertionError("codebook has %d entries, expected %d" % (values.size, 2 ** bit))
    with np.errstate(all="ignore"):
        c = _F32(_F32(1.0) / np.max(values)) * _F32(10.0)
        return (values * _F32(c)).astype(_F32)


def _value_bits(bit, signed):
    return bit - 1 if signed else bit


# ------------------------------------------------------------------------------------
# magnitude sets
# ------------------------------------------------------------------------------------
def _flint_magnitudes(B):
    """flint with B value bits: binade e carries mant(e) mantissa bits,
    mant(e) = B-1-|e| for e < 0, B-2-|e| for e >= 0, plus the single top value 2^(B-1)."""
    if B < 2:
        raise AssertionError("flint needs at least 2 value bits")
    mags = []
    for e in range(-(B - 1), B - 1):
        mant = (B - 1 + e) if e < 0 else (B - 2 - e)
        if mant < 0:
            continue
        for j in range(2 ** mant):
            mags.append(2.0 ** e * (1.0 + j * 2.0 ** (-mant)))
    mags.append(2.0 ** (B - 1))
    return mags, B - 1  # magnitudes, exponent of the maximum


def _float_magnitudes(B, eb):
    """minifloat with eb exponent bits and B-eb mantissa bits, subnormals in binade 0 (AQ:157-187)."""
    exp_bit, man_bit = eb, B - eb
    if B == 2:
        exp_bit, man_bit = 2, 0
    if man_bit < 0:
        raise TypeError("float grid: %d exponent bits do not fit %d value bits" % (eb, B))
    mags = []
    for j in range(1, 2 ** man_bit):
        mags.append(j * 2.0 ** (-man_bit))                      # subnormals (i = 0)
    for i in range(1, 2 ** exp_bit):
        for j in range(2 ** man_bit):
            mags.append(2.0 ** (i - 1) * (1.0 + j * 2.0 ** (-man_bit)))
    return mags


_APOT_TERMS = {
    # value bits -> additive power-of-two term sets (AQ:85-131)
    2: ([0.5, 0.25, 0.125], [], []),
    3: ([0.5, 0.25, 0.0625], [0.125], []),
    4: ([2.0 ** -1, 2.0 ** -3, 2.0 ** -5], [2.0 ** -2, 2.0 ** -4, 2.0 ** -6], []),
    5: ([2.0 ** -1, 2.0 ** -3, 2.0 ** -6], [2.0 ** -2, 2.0 ** -4, 2.0 ** -7], [2.0 ** -5]),
    6: ([2.0 ** -1, 2.0 ** -4, 2.0 ** -7], [2.0 ** -2, 2.0 ** -5, 2.0 ** -8], [2.0 ** -3, 2.0 ** -6, 2.0 ** -9]),
}


# ------------------------------------------------------------------------------------
# ANT grids
# ------------------------------------------------------------------------------------
def ant_int(bit, signed):
    B = _value_bits(bit, signed)
    if signed:
        v = np.arange(-(2 ** B), 2 ** B, dtype=np.float64).astype(_F32)  # -2^B .. 2^B-1  (AQ:217-219)
    else:
        v = np.arange(0, 2 ** B, dtype=np.float64).astype(_F32)
    return _ant_normalise(v, bit)


def ant_pot(bit, signed):
    B = _value_bits(bit, signed)
    with np.errstate(all="ignore"):
        mags = [2.0 ** i for i in range(0, 2 ** B - 1)]
        v = _mirror(mags, signed)        # signed: 2*(2^B-1) + one zero, padded with a second zero
    return _ant_normalise(v, bit)


def ant_flint(bit, signed):
    B = _value_bits(bit, signed)
    mags, _ = _flint_magnitudes(B)
    return _ant_normalise(_mirror(mags, signed), bit)


def ant_float(bit, signed, eb=3):
    B = _value_bits(bit, signed)
    return _ant_normalise(_mirror(_float_magnitudes(B, eb), signed), bit)


def ant_apot(bit, signed):
    B = _value_bits(bit, signed)
    ta, tb, tc = _APOT_TERMS.get(B, ([], [], []))
    sums = sorted({a + b + c for a in [0.0] + ta for b in [0.0] + tb for c in [0.0] + tc} - {0.0})
    if signed:
        # the reference emits +0.0 and -0.0 (a+b+c and its negation) and no padding zero
        v = np.concatenate([-np.asarray(sums[::-1]), [0.0, -0.0], np.asarray(sums)]).astype(_F32)
    else:
        v = np.concatenate([[0.0], np.asarray(sums)]).astype(_F32)
        if v.size + 1 == 2 ** bit:
            v = np.sort(np.concatenate([v, [0.0]]).astype(_F32))
    return _ant_normalise(v, bit)


def _frozen(fn):
    """The generators below are pure functions of tiny arguments and a calibration asks for the same codebook once per
    quantiser (146 times in BERT-base): memoised, the array handed out read-only so that no caller can edit the cache."""
    @functools.lru_cache(maxsize=None)
    def cached(*a):
        v = np.ascontiguousarray(fn(*a))
        v.setflags(write=False)
        return v
    return functools.wraps(fn)(lambda *a: cached(*a))


@_frozen
def ant_grid(mode, bit, signed):
    """Grid the reference installs for a resolved mode string (AQ:488-511)."""
    if mode == "int":
        return ant_int(bit, signed)
    if mode == "flint":
        return ant_flint(bit, signed)
    if mode == "pot":
        return ant_pot(bit, signed)
    if mode == "apot":
        return ant_apot(bit, signed)
    if mode == "float":
        return ant_float(bit, signed, 3)
    if mode in ("float1", "float2", "float3", "float4"):
        return ant_float(bit, signed, int(mode[-1]))
    raise RuntimeError("Unsupported mode: " + mode)


# ------------------------------------------------------------------------------------
# OliVe grids (outlier threshold normalised to 32)
# ------------------------------------------------------------------------------------
def olive_int(bit, signed):
    """+-(1 .. 2^B-1) and 0, times 32/2^B: 2^bit - 1 values when signed (OQ:72-91)."""
    B = _value_bits(bit, signed)
    mags = np.arange(1, 2 ** B, dtype=np.float64)
    v = _mirror(mags, signed, zeros_signed=1)
    return (v * _F32(32 / (2 ** B))).astype(_F32)


def olive_flint(bit, signed):
    B = _value_bits(bit, signed)
    mags, exp_max = _flint_magnitudes(B)
    v = _mirror(mags, signed, zeros_signed=1)
    return (v * _F32(32 / (2 ** exp_max))).astype(_F32)


@_frozen
def olive_outliers(bit, signed, exp_bit=2, exp_base=5):
    """abfloat outlier codebook 2^i (1 + j 2^-m), i = 5..8, without 32 itself (OQ:155-179)."""
    B = _value_bits(bit, signed)
    mant = B - exp_bit
    if mant < 0:
        # int(2 ** negative) == 0 in the reference: no values at all
        return np.zeros(0, dtype=_F32)
    mags = [2.0 ** i * (1.0 + j * 2.0 ** (-mant))
            for i in range(exp_base, exp_base + 2 ** exp_bit) for j in range(2 ** mant)
            if not (i == exp_base and j == 0)]
    if signed:
        mags = np.sort(np.asarray(mags))
        return np.concatenate([-mags[::-1], mags]).astype(_F32)
    return np.sort(np.asarray(mags)).astype(_F32)


@_frozen
def olive_grid(mode, bit, signed):
    if mode == "int":
        return olive_int(bit, signed)
    if mode == "flint":
        return olive_flint(bit, signed)
    raise RuntimeError("Unsupported mode: " + mode)
