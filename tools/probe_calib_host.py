#!/usr/bin/env python3
"""Where the host time of a model's FIRST (calibrating) forward goes: cProfile of HF BERT-base (fp32, batch 64 x 128,
mode ant-int-pot-flint = type selection + clip search per quantiser, 146 quantisers), top functions by cumulative time."""
import cProfile
import io
import os
import pstats
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from transformers import BertConfig, BertModel  # noqa: E402

from ant_quantization_amd.ant import quant_model as qm, quant_utils as qu  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "ant-int-pot-flint"
args = types.SimpleNamespace(mode=mode, wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False)
qu.set_quantizer(args)
torch.manual_seed(0)
ids = torch.randint(0, 30000, (64, 128), device=dev)
with torch.no_grad():
    warm = qm.quantize_model(BertModel(BertConfig(num_hidden_layers=1)).eval()).to(dev).eval()     # library / plan warm-up
    qu.enable_quantization(warm)
    warm(ids)
    torch.cuda.synchronize()
    model = qm.quantize_model(BertModel(BertConfig()).eval()).to(dev).eval()
    qu.enable_quantization(model)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    model(ids)
    torch.cuda.synchronize()
    pr.disable()
    print("mode %s: first forward %.1f ms" % (mode, (time.perf_counter() - t0) * 1e3))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines() if "site-packages/torch/nn" not in l))
