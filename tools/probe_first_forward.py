#!/usr/bin/env python3
"""Where the first (calibrating) forward of a quantised BERT-base spends its host time: cProfile of that one call.
    python tools/probe_first_forward.py [mode]        (default: flint; e.g. ant-int-pot-flint)"""
import cProfile
import os
import pstats
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from transformers import BertConfig, BertModel  # noqa: E402

TREE = os.environ.get("ANTQ_TREE", "ant")              # ant | olive
import importlib  # noqa: E402
qm = importlib.import_module("ant_quantization_amd.%s.quant_model" % TREE)
qu = importlib.import_module("ant_quantization_amd.%s.quant_utils" % TREE)

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "flint"
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(sys.argv[2] if len(sys.argv) > 2 else "", torch.float32)     # model dtype
args = types.SimpleNamespace(mode=mode, wbit=4, abit=4, w_up=150 if TREE == "ant" else 250, a_up=150 if TREE == "ant" else 250, w_low=75, a_low=75,
                             percent=100, search=False, no_outlier=False)
qu.set_quantizer(args)
if "ANTQ_HIST" in os.environ:                      # knob 14: 0 = the direct clip-search kernels only, 2 = the histogram path wherever eligible
    from ant_quantization_amd import _lib as _l
    _l.lib().antq_debug_set(14, int(os.environ["ANTQ_HIST"]))
    print("knob 14 =", os.environ["ANTQ_HIST"])
torch.manual_seed(0)
ids = torch.randint(0, 30000, (64, 128), device=dev)
if os.environ.get("ANTQ_PREWARM") == "1":
    # touch every kernel family once on toy tensors (module load, plan / grid caches) before the model exists
    import numpy as np
    from ant_quantization_amd import _lib, core, grids
    t0 = time.perf_counter()
    for dt in (torch.float32,):
        xw = torch.randn(256, 1024, device=dev, dtype=dt)
        for per in (True, False):
            pl = _lib.plan_for(grids.ant_flint(4, True))
            xm = core.row_absmax(xw, per)
            core.clip_search(xw, xm, per, 75, 150, 1, pl, 10.0)
            _lib.fakequant(xw, xm, pl, 10.0, 256, 1024, per)
        xw.min().item()
    torch.cuda.synchronize()
    print("prewarm %.1f ms" % ((time.perf_counter() - t0) * 1e3))
if os.environ.get("ANTQ_PREALLOC") == "1":
    # hand the caching allocator what the quantised forward will ask for, before it asks
    blocks = [torch.empty(n, device=dev, dtype=torch.uint8) for n in (100 << 20, 100 << 20, 100 << 20, 26 << 20, 26 << 20, 26 << 20, 10 << 20, 10 << 20, 10 << 20, 3 << 20, 3 << 20, 2 << 20, 2 << 20, 2 << 20, 2 << 20)]
    del blocks
for rep in range(2):
    model = qm.quantize_model(BertModel(BertConfig()).eval()).to(dev).to(DT).eval()
    with torch.no_grad():
        qu.disable_quantization(model)
        model(ids); model(ids)
        torch.cuda.synchronize()
        qu.enable_quantization(model)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if rep == 1:
            pr.enable()
        sys.stdout = open(os.devnull, "w")
        model(ids)
        t_host = time.perf_counter()
        torch.cuda.synchronize()
        sys.stdout = sys.__stdout__
        if rep == 1:
            pr.disable()
        print("%s mode %s %s: first forward %.1f ms%s (host returned after %.1f ms: the rest is the GPU finishing)" % (
            TREE, mode, str(DT)[6:], (time.perf_counter() - t0) * 1e3, " (under cProfile)" if rep else "", (t_host - t0) * 1e3))
        t0 = time.perf_counter()
        model(ids); torch.cuda.synchronize()
        print("   second forward %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats(os.environ.get("ANTQ_SORT", "cumulative")).print_stats(45)
