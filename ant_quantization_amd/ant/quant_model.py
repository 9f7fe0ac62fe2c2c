"""ant_quantization/antquant/quant_model.py surface.  nn.MultiheadAttention is NOT wrapped: the
reference's MultiheadAttentionQuantizer is a vendored copy of torch's attention (out of scope);
such modules are left untouched by the exact-type match and their inner nn.Linear-like projections
(NonDynamicallyQuantizableLinear is a subclass, so not matched either) stay in full precision."""
import torch.nn as nn

from .._model import (load_ant_state_dict, make_quantize_model, make_set_8_bit_layer_l,  # noqa: F401
                      make_set_8_bit_layer_n, set_first_last_layer_impl)
from .quant_modules import Conv2dQuantizer, LinearQuantizer, TensorQuantizer
from .quant_utils import quant_args

quantize_model = make_quantize_model([(nn.Conv2d, Conv2dQuantizer), (nn.Linear, LinearQuantizer)], quant_args)
set_8_bit_layer_l = make_set_8_bit_layer_l(TensorQuantizer, verbose_rank0_only=True)
set_8_bit_layer_n = make_set_8_bit_layer_n(TensorQuantizer, verbose_rank0_only=True)


def set_first_last_layer(model):
    set_first_last_layer_impl(model, TensorQuantizer)
