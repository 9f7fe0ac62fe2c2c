"""Model rewrite + mixed-precision policy shared by the ANT and OliVe mirrors.

Behaviour of the reference's quant_model.py (ant_quantization/antquant/quant_model.py:11-154,
olive_quantization/antquant/quant_model.py), kept because checkpoints and harnesses depend on
the resulting module tree:
  * exact-type matches only (`type(m) == nn.Linear`; subclasses are left alone);
  * nn.Sequential AND nn.ModuleList both come back as nn.Sequential;
  * other modules are deep-copied and every attribute that is an nn.Module is rewritten
    recursively (OliVe skips `base_model` and `lm_head`);
  * set_8_bit_layer_n / _l re-arm calibration of EVERY TensorQuantizer and raise selected
    (weight, input) pairs to 8 bit.
"""
import copy

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


def _rank0():
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0


def make_quantize_model(wrappers, quant_args, skip_attrs=()):
    """wrappers: list of (exact module type, wrapper class)."""

    def quantize_model(model):
        for src_type, wrapper in wrappers:
            if type(model) == src_type:
                quant_mod = wrapper(**quant_args)
                quant_mod.set_param(model)
                return quant_mod
        if type(model) in (nn.Sequential, nn.ModuleList) or isinstance(model, nn.Sequential):
            return nn.Sequential(*[quantize_model(m) for _, m in model.named_children()])
        q_model = copy.deepcopy(model)
        for attr in dir(model):
            if attr in skip_attrs or isinstance(getattr(type(model), attr, None), property):
                continue               # properties (HF `base_model`, ...) only alias children that are visited anyway
            try:
                mod = getattr(model, attr)
            except Exception:          # properties that need context (HF models have a few)
                continue
            if not isinstance(mod, nn.Module) or mod is model:   # `base_model` of a HF backbone is the module itself
                continue
            try:
                setattr(q_model, attr, quantize_model(mod))
            except AttributeError:     # read-only property aliasing a child that is rewritten under its real name
                pass
        return q_model

    return quantize_model


def _quantizers(model, cls, rearm=True):
    mods = []
    for m in model.modules():
        if isinstance(m, cls):
            mods.append(m)
            if rearm:
                m.has_inited_quant_para.data = torch.zeros_like(m.has_inited_quant_para)
                m.rearm()
    return mods


def _to_8bit(m):
    m.bit.data = torch.tensor(8, device=m.bit.device)
    if hasattr(m, "_hm_known"):
        m._hm_known("bit", 8)       # the host just wrote it: no read-back later
    m.rearm()


def make_set_8_bit_layer_l(cls, verbose_rank0_only):
    def set_8_bit_layer_l(model, layer_list):
        if layer_list == "None":
            return
        layers = [int(x) for x in layer_list.split(',')]
        module_list = _quantizers(model, cls)
        say = _rank0() if verbose_rank0_only else True
        if say:
            print("------------- 8-bit Re-SET -------------")
            print(len(layers))
        assert len(layers) > 0
        for i in range(len(module_list) // 2):
            if i in layers:
                if say:
                    print(module_list[i * 2].name, i)
                    print(module_list[i * 2 + 1].name, i)
                _to_8bit(module_list[i * 2])
                _to_8bit(module_list[i * 2 + 1])
        if say:
            print("------------- 8-bit Re-SET -------------")

    return set_8_bit_layer_l


def make_set_8_bit_layer_n(cls, verbose_rank0_only):
    def set_8_bit_layer_n(model, l_num):
        """Raise the l_num (weight, input) pairs to 8 bit: always the last two pairs (BERT's
        pooler / classifier), then the pairs with the largest summed calibration MSE."""
        module_list = _quantizers(model, cls, rearm=False)
        mse_list = [m.mse.item() for m in module_list]
        for m in module_list:
            m.has_inited_quant_para.data = torch.zeros_like(m.has_inited_quant_para)
            m.rearm()
        say = _rank0() if verbose_rank0_only else True
        if say:
            print("------------- 8-bit Re-SET -------------")
            print(l_num)
        assert l_num > 0
        l_num *= 2
        first_num, last_num = 0, 4
        for i in list(range(0, first_num)) + list(range(len(mse_list) - last_num, len(mse_list))):
            if say:
                print(module_list[i].name)
            _to_8bit(module_list[i])
        if say:
            print("------------- First and Last end -------------")
        module_list = module_list[first_num: len(mse_list) - last_num]
        mse_list = mse_list[first_num: len(mse_list) - last_num]
        pair = np.array([mse_list[2 * i] + mse_list[2 * i + 1] for i in range(len(mse_list) // 2)])
        order = np.argsort(-pair)
        n_pairs = (l_num - first_num - last_num) // 2
        if n_pairs > 0:
            for i in order[0:n_pairs]:
                if say:
                    print(module_list[i * 2].name, pair[i], i)
                    print(module_list[i * 2 + 1].name, pair[i], i)
                _to_8bit(module_list[i * 2])
                _to_8bit(module_list[i * 2 + 1])
        if say:
            print("------------- 8-bit Re-SET -------------")

    return set_8_bit_layer_n


def set_first_last_layer_impl(model, cls):
    # the reference only collects the two lists and does nothing with them (AQ quant_model.py:53-60)
    weights = [m for m in model.modules() if isinstance(m, cls) and not m.is_input]
    inputs = [m for m in model.modules() if isinstance(m, cls) and m.is_input]
    return weights, inputs


def load_ant_state_dict(model, checkpoint):
    """Pre-size every quant_grid buffer from the checkpoint so a strict load_state_dict succeeds
    (the grid length depends on the calibrated bit width, AQ quant_model.py:151-154)."""
    for name, module in model.named_modules():
        if name + ".quant_grid" in checkpoint.keys():
            module.quant_grid.data = checkpoint[name + ".quant_grid"]
            if hasattr(module, "rearm"):
                module.rearm()
        if name + ".outliers" in checkpoint.keys() and hasattr(module, "outliers"):
            module.outliers.data = checkpoint[name + ".outliers"]
        # Conv1dQuantizer.set_param (like the reference's, OQ:368-375) leaves quant_weight.alpha 0-dim until
        # calibration; pre-size it too so that calibrated GPT-2 checkpoints load with strict=True
        if name + ".alpha" in checkpoint.keys() and hasattr(module, "alpha") and hasattr(module, "quant_grid"):
            if module.alpha.shape != checkpoint[name + ".alpha"].shape:
                module.alpha.data = checkpoint[name + ".alpha"].detach().clone().to(module.alpha.device)
