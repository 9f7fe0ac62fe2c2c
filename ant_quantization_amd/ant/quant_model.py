"""ant_quantization/antquant/quant_model.py surface: nn.Conv2d / nn.Linear / nn.MultiheadAttention are
replaced by their quantised wrappers (exact-type match, quant_model.py:17-30)."""
import torch.nn as nn

from .._model import (load_ant_state_dict, make_quantize_model, make_set_8_bit_layer_l,  # noqa: F401
                      make_set_8_bit_layer_n, set_first_last_layer_impl)
from .multihead_attention import MultiheadAttentionQuantizer
from .quant_modules import Conv2dQuantizer, LinearQuantizer, TensorQuantizer
from .quant_utils import quant_args

quantize_model = make_quantize_model([(nn.Conv2d, Conv2dQuantizer), (nn.Linear, LinearQuantizer),
                                      (nn.MultiheadAttention, MultiheadAttentionQuantizer)], quant_args)
set_8_bit_layer_l = make_set_8_bit_layer_l(TensorQuantizer, verbose_rank0_only=True)
set_8_bit_layer_n = make_set_8_bit_layer_n(TensorQuantizer, verbose_rank0_only=True)


def set_first_last_layer(model):
    set_first_last_layer_impl(model, TensorQuantizer)
