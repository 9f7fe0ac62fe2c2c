"""quant_utils.py surface shared by both mirrors (ant_quantization/antquant/quant_utils.py)."""
import logging
import os
import uuid

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


def set_util_logging(filename):
    logging.basicConfig(
        format='%(asctime)s - %(levelname)s - %(name)s -   %(message)s',
        datefmt='%m/%d/%Y %H:%M:%S',
        level=logging.INFO,
        handlers=[logging.FileHandler(filename), logging.StreamHandler()],
    )


def tag_info(args):
    return "" if args.tag == "" else "_" + args.tag


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def get_ckpt_path(args):
    """output/<model>_<dataset>/<mode>_W<w>A<a>_<run id>/gpu_<rank>; the run id is drawn on rank 0 and
    broadcast (over RCCL when a process group exists) so that all ranks agree (quant_utils.py:34-60)."""
    multi = _dist_on()
    rank = dist.get_rank() if multi else 0
    run_id = int(uuid.uuid4().hex[:4], 16)
    if multi:
        t = torch.tensor(run_id, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.broadcast(t, 0)
        run_id = int(t.item())
    run_dir = os.path.join("output", f"{args.model}_{args.dataset}", f"{args.mode}_W{args.wbit}A{args.abit}_{run_id}")
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if multi:
        dist.barrier()
    mine = os.path.join(run_dir, f"gpu_{rank}")
    os.makedirs(mine, exist_ok=True)
    return mine


def get_ckpt_filename(path, epoch):
    return os.path.join(path, f"ckpt_{epoch}.pth")


def make_walkers(Q):
    """The three model walkers of quant_utils.py:62-78 for quantiser class Q."""
    def walker(method, with_name):
        def walk(model):
            if method == "enable_quantization":        # (the calibrating forward usually follows: first-launch costs paid now, _lib.prewarm)
                from . import _lib
                p0 = next(model.parameters(), None)
                if p0 is not None and p0.is_cuda:
                    try:
                        _lib.prewarm(p0.device, p0.dtype)
                    except Exception as ex:        # noqa: BLE001  (a warm-up, never a reason to fail quantising the model)
                        logger.warning("ant_quantization_amd: kernel prewarm skipped (%s)", ex)
            for name, module in model.named_modules():
                if isinstance(module, Q):
                    getattr(module, method)(*((name,) if with_name else ()))
            if method == "enable_quantization" and getattr(model, "_antq_auto_bank", None) is None:
                from .weight_bank import AutoBank
                try:
                    from .weight_bank import FlushHook
                    auto = AutoBank(model)
                    object.__setattr__(model, "_antq_auto_bank", auto)                 # (not a submodule, not state)
                    if not any(isinstance(h, FlushHook) for h in model._forward_hooks.values()):      # (a copy brings its own)
                        model.register_forward_hook(FlushHook())                       # (end of forward: picks named, bank spent)
                except Exception:              # noqa: BLE001  (exotic containers: the per-layer schedule simply stays)
                    pass
        walk.__name__ = method
        return walk

    return (walker("disable_input_quantization", False), walker("enable_quantization", True),
            walker("disable_quantization", True))


def make_set_weights_at_rest(Q):
    def set_weights_at_rest(model, flag=True, resident=True):
        """Ours, opt-in (inference on frozen weights): the caller promises that nothing writes the weights or alphas behind
        torch's back (no raw `.data` edits) while the flag is on.  Two things then happen:
          * resident=True (default): the model's weight bank keeps its quantised copies across forwards -- a no-grad forward
            on unchanged weights launches NOTHING for them (weight_bank, "RESIDENT"); version counters, train() / eval()
            calls and training forwards still trigger a refresh.  With flag=False the bank returns to the default,
            the reference's schedule: every forward re-quantises (one launch).
          * every weight quantiser the bank does not serve (bank switched off, no memory for the copies, or resident=False:
            the bank steps aside altogether) marks its own launch as independent of the work queued before it on the stream
            -- the kernel may start while earlier kernels drain (ANTQ_FLAG_UNORDERED: one launch per 33.5 MB tensor
            65 -> 77 % of the HBM roofline)."""
        for module in model.modules():
            if isinstance(module, Q):
                module.weights_at_rest = bool(flag)
        ab = getattr(model, "_antq_auto_bank", None)
        if flag and not resident:
            set_weight_bank(model, False)      # (an explicit choice of per-layer launches: the automatic bank steps aside)
        elif ab is not None:
            ab.resident = bool(flag)
            if ab.bank is not None:
                ab.bank.resident = bool(flag)
                ab.bank.dirty = True
    return set_weights_at_rest


def set_weight_bank(model, flag=True):
    """Ours: switch the automatic one-launch weight path (weight_bank.AutoBank, armed by enable_quantization) off -- the
    reference's per-layer schedule, a fresh tensor per layer and forward -- or back on."""
    from .weight_bank import AutoBank
    ab = getattr(model, "_antq_auto_bank", None)
    if not flag:
        if ab is not None:
            ab.disable()
        return
    if ab is None:
        from .weight_bank import FlushHook
        ab = AutoBank(model)
        object.__setattr__(model, "_antq_auto_bank", ab)
        if not any(isinstance(h, FlushHook) for h in model._forward_hooks.values()):
            model.register_forward_hook(FlushHook())
    else:
        ab.enabled = True


def get_model(args):
    import torchvision.models as models  # optional dependency, only the ImageNet harness needs it
    if args.model == "inception_v3":
        return models.inception_v3(aux_logits=False, pretrained=True)
    return models.__dict__[args.model](pretrained=True)
