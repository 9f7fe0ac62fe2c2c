"""olive_quantization/antquant/quant_utils.py surface (identical to the ANT one in the reference)."""
import logging  # noqa: F401
import os  # noqa: F401

import torch  # noqa: F401

from .._utils import (get_ckpt_filename, get_ckpt_path, get_model, logger, make_set_weights_at_rest, set_weight_bank,  # noqa: F401
                      make_walkers, set_util_logging, tag_info)
from .quant_modules import Quantizer as Q

quant_args = {}


def set_quantizer(args):
    global quant_args
    quant_args.update({'mode': args.mode, 'wbit': args.wbit, 'abit': args.abit, 'args': args})


disable_input_quantization, enable_quantization, disable_quantization = make_walkers(Q)
set_weights_at_rest = make_set_weights_at_rest(Q)       # ours, opt-in: unordered launches for frozen weights
