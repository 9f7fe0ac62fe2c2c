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
    rank = dist.get_rank() if _dist_on() else 0

    def mk(p):
        if rank == 0 and not os.path.isdir(p):
            os.mkdir(p)

    path = 'output'
    mk(path)
    path = os.path.join(path, args.model + "_" + args.dataset)
    mk(path)
    num = int(uuid.uuid4().hex[0:4], 16)
    if _dist_on():
        t = torch.tensor(num, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.broadcast(t, 0)
        num = int(t.item())
    path = os.path.join(path, args.mode + '_W' + str(args.wbit) + 'A' + str(args.abit) + '_' + str(num))
    mk(path)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if _dist_on():
        dist.barrier()
    path = os.path.join(path, "gpu_" + str(rank))
    os.makedirs(path, exist_ok=True)
    return path


def get_ckpt_filename(path, epoch):
    return os.path.join(path, 'ckpt_' + str(epoch) + '.pth')


def make_walkers(Q):
    def disable_input_quantization(model):
        for _, module in model.named_modules():
            if isinstance(module, Q):
                module.disable_input_quantization()

    def enable_quantization(model):
        for name, module in model.named_modules():
            if isinstance(module, Q):
                module.enable_quantization(name)

    def disable_quantization(model):
        for name, module in model.named_modules():
            if isinstance(module, Q):
                module.disable_quantization(name)

    return disable_input_quantization, enable_quantization, disable_quantization


def get_model(args):
    import torchvision.models as models  # optional dependency, only the ImageNet harness needs it
    if args.model == "inception_v3":
        return models.inception_v3(aux_logits=False, pretrained=True)
    return models.__dict__[args.model](pretrained=True)
