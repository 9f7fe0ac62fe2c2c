"""olive_quantization/antquant/quant_model.py surface: additionally wraps HF `Conv1D` (GPT-2) and never
descends into `base_model` / `lm_head` (olive quant_model.py:26-29,50)."""
import torch.nn as nn

from .._model import (load_ant_state_dict, make_quantize_model, make_set_8_bit_layer_l,  # noqa: F401
                      make_set_8_bit_layer_n, set_first_last_layer_impl)
from .quant_modules import Conv1dQuantizer, Conv2dQuantizer, LinearQuantizer, TensorQuantizer
from .quant_utils import quant_args

_wrappers = [(nn.Conv2d, Conv2dQuantizer), (nn.Linear, LinearQuantizer)]
try:  # transformers is optional; without it there is no Conv1D to match
    from transformers import pytorch_utils as _pu
    _wrappers.append((_pu.Conv1D, Conv1dQuantizer))
except Exception:  # pragma: no cover
    pass

quantize_model = make_quantize_model(_wrappers, quant_args, skip_attrs=('base_model', 'lm_head'))
set_8_bit_layer_l = make_set_8_bit_layer_l(TensorQuantizer, verbose_rank0_only=False)
set_8_bit_layer_n = make_set_8_bit_layer_n(TensorQuantizer, verbose_rank0_only=False)


def set_first_last_layer(model):
    set_first_last_layer_impl(model, TensorQuantizer)
