"""Mirror of ant_quantization/antquant/quant_affine.py (generic asymmetric k-bit linear
fake-quant; star-imported but never called by the reference's quantiser).

`AsymmetricQuantFunction.forward` is one fused gfx950 kernel (antq_affine): the reference's
six element-wise PyTorch launches (scale*x, -zp, round, clamp, +zp, /scale; quant_affine.py
:95-115) become a single read and a single write.  The small helpers keep the reference's
names and broadcasting rules.  GPU tensors only (no CPU fallback in this package).
"""
import torch
from torch.autograd import Function

from .. import _lib

__all__ = ["clamp", "linear_quantize", "linear_dequantize", "asymmetric_linear_quantization_params",
           "AsymmetricQuantFunction"]


def clamp(input, min, max, inplace=False):
    return input.clamp_(min, max) if inplace else torch.clamp(input, min, max)


def _per_channel_view(input, t):
    # quant_affine.py:28-34 / :51-57: [-1,1,1,1] for conv weights/activations, [-1,1] for linear
    if not isinstance(t, torch.Tensor):
        return t
    if input.dim() == 4:
        return t.view(-1, 1, 1, 1)
    if input.dim() == 2:
        return t.view(-1, 1)
    return t


def linear_quantize(input, scale, zero_point, inplace=False):
    scale, zero_point = _per_channel_view(input, scale), _per_channel_view(input, zero_point)
    if inplace:
        return input.mul_(scale).sub_(zero_point).round_()
    return scale * input - zero_point          # NB: the reference does not round here (:39)


def linear_dequantize(input, scale, zero_point, inplace=False):
    scale, zero_point = _per_channel_view(input, scale), _per_channel_view(input, zero_point)
    if inplace:
        return input.add_(zero_point).div_(scale)
    return (input + zero_point) / scale


def asymmetric_linear_quantization_params(num_bits, saturation_min, saturation_max,
                                          integral_zero_point=True, signed=True):
    n = 2 ** num_bits - 1
    scale = n / torch.clamp((saturation_max - saturation_min), min=1e-8)
    zero_point = scale * saturation_min
    if integral_zero_point:
        zero_point = zero_point.round() if isinstance(zero_point, torch.Tensor) else float(round(zero_point))
    if signed:
        zero_point += 2 ** (num_bits - 1)
    return scale, zero_point


class AsymmetricQuantFunction(Function):
    """k-bit asymmetric fake-quant with a given range (inference only, like the reference)."""

    @staticmethod
    def forward(ctx, x, k, x_min=None, x_max=None):
        if x_min is None or x_max is None:
            raise ValueError("x_min / x_max are required (the reference's auto-range branch is commented out)")
        xc = x.detach().contiguous()
        if xc.dtype != torch.float32:
            raise _lib.AntqError("AsymmetricQuantFunction is fp32 only, like the reference path")
        mn = torch.as_tensor(x_min, dtype=torch.float32, device=xc.device).reshape(-1).contiguous()
        mx = torch.as_tensor(x_max, dtype=torch.float32, device=xc.device).reshape(-1).contiguous()
        per_row = mn.numel() > 1
        if per_row:
            if xc.dim() not in (2, 4) or mn.numel() != xc.shape[0] or mx.numel() != xc.shape[0]:
                raise ValueError("per-channel range needs a 2-D / 4-D input with one (min, max) per dim-0 slice")
            rows, row_len = xc.shape[0], xc.numel() // xc.shape[0]
        else:
            rows, row_len = 1, xc.numel()
        return _lib.affine(xc, int(k), mn, mx, rows, row_len, per_row)

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError
