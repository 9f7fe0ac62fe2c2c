"""Batches of BERT-base / ResNet-50 / LLM-wide (11008, 28672) weight shapes, fp32 and bf16: the default job rules (knob 5 = 1)
against every long row as a lane job (knob 5 = 2), same process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import timed, resnet50_shapes
dev = torch.device("cuda:0")
flint = _lib.plan_for(grids.ant_flint(4, True))
bert = [(768, 768)] * 49 + [(3072, 768)] * 12 + [(768, 3072)] * 12
res50 = [(s[0], int(np.prod(s[1:]))) for s in resnet50_shapes()]
llm = [(4096, 11008)] * 4 + [(11008, 4096)] * 4 + [(2048, 28672)] * 2
for name, shapes in (("BERT-base 73 W", bert), ("ResNet-50 54 W", res50), ("11008 / 28672 wide", llm)):
    for dt, bpe in ((torch.float32, 8), (torch.bfloat16, 4)):
        xs = [(torch.randn(*s, device=dev) * 0.02).to(dt) for s in shapes]
        outs = [torch.empty_like(x) for x in xs]
        al = [_lib.absmax(x, x.shape[0], x.shape[1]) for x in xs]
        n = sum(x.numel() for x in xs) * bpe
        for rnd in range(2):
            res = []
            for knob in (1, 2):
                _lib.lib().antq_debug_set(5, knob)
                b = _lib.Batch([(x, o, a, flint, 10.0, x.shape[0], x.shape[1], True) for x, o, a in zip(xs, outs, al)])
                _lib.lib().antq_debug_set(5, 1)
                res.append(n / timed(b.run, 20) / 8e10)
            print("%-20s %-9s batch: default %.2f   all long rows as lane jobs %.2f" % (name, str(dt)[6:], res[0], res[1]), flush=True)
        del xs, outs
