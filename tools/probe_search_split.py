"""Clip-search launches: time against the number of candidate-list chunks (blockIdx.y; knob 12 forces it, 0 = the cost model of
antq_search.hip: search_grid).  Per tensor and per row, the shapes of a BERT-base / LLM calibration pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ant_quantization_amd import _lib, grids, core
dev = torch.device("cuda:0")
knob = _lib.lib().antq_debug_set
plan = _lib.plan_for(grids.ant_flint(4, True))
ratios = core._ratios(75, 151, 1, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
cases = [(8192, 768, False, torch.float32), (8192, 3072, False, torch.float32), (4096, 4096, False, torch.float32),
         (2048, 768, False, torch.float32), (768, 768, True, torch.float32), (3072, 768, True, torch.float32), (768, 3072, True, torch.float32),
         (4096, 4096, True, torch.float32), (4096, 4096, True, torch.bfloat16), (8192, 4096, False, torch.bfloat16), (11008, 4096, True, torch.bfloat16)]
for rows, K, per_row, dt in cases:
    x = (torch.randn(rows, K, device=dev) * 0.05).to(dt)
    r_, k_ = (rows, K) if per_row else (1, rows * K)
    xm = _lib.absmax(x, r_, k_, per_row=per_row).reshape(-1)
    line = []
    for c in (0, 1, 2, 4, 6, 7, 8, 10, 12, 19):
        knob(12, c)
        line.append("%s %6.1f" % ("auto" if c == 0 else "c=%d" % c, t(lambda: _lib.search_sse(x, r_, k_, xm, per_row, ratios, plan, 10.0))))
    knob(12, 0)
    print("%5d x %5d %-10s %-8s us: %s" % (rows, K, "per row" if per_row else "per tensor", str(dt)[6:], "  ".join(line)), flush=True)
