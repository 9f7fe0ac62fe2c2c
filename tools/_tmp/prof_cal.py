import os, sys, types, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from transformers import BertConfig, BertModel
from ant_quantization_amd.ant import quant_model as qm, quant_utils as qu
dev = torch.device("cuda:0")
for mode in ("flint", "ant-int-pot-flint"):
    args = types.SimpleNamespace(mode=mode, wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False)
    qu.set_quantizer(args)
    torch.manual_seed(0)
    model = qm.quantize_model(BertModel(BertConfig()).eval()).to(dev).eval()
    ids = torch.randint(0, 30000, (64, 128), device=dev)
    qu.enable_quantization(model)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable(); model(ids); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(mode); print(s.getvalue()[:6000])
